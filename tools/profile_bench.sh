#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats + separate PMC passes (HBM fetch / write, SQ instruction mix).
# Output tree: gpurun_out/prof/{stats,fetch,write,sq,sqw,l2}; condense with tools/summarize_prof.py gpurun_out/prof <tag> "<note>".
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/gpurun_out/prof
rm -rf $P && mkdir -p $P
ARGS="$R/bench.py --no-cpu-baseline --no-pose-legs"   # default --steps / --warmup: the command the driver runs, without the legs at other poses
                                                    # (so that every launch of the step kernel is a headline-pose launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $ARGS > $P/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 $ARGS > $P/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 $ARGS > $P/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $P/sq -- python3 $ARGS > $P/sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $P/sqw -- python3 $ARGS > $P/sqw.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $P/l2 -- python3 $ARGS > $P/l2.log 2>&1
ls $P/*/ | head -20
