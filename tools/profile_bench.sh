#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats + separate PMC passes (HBM fetch / write, SQ instruction mix).
# Output tree: gpurun_out/prof/{stats,fetch,write,sq}; condense with tools/summarize_prof.py gpurun_out/prof <tag> "<note>".
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/gpurun_out/prof
rm -rf $P && mkdir -p $P
ARGS="$R/bench.py --no-cpu-baseline"   # default --steps / --warmup: the same command the driver runs
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $ARGS > $P/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 $ARGS > $P/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 $ARGS > $P/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $P/sq -- python3 $ARGS > $P/sq.log 2>&1
ls $P/*/ | head -20
