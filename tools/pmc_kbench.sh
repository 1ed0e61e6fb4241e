cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $R/gpurun_out/sq_counters.txt
wc -l $R/gpurun_out/sq_counters.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc$i -o p -- $R/build/kbench4 8 256 > /dev/null 2>$R/gpurun_out/pmc$i.err
  tail -2 $R/gpurun_out/pmc$i.err
done
ls -R $R/gpurun_out/pmc1 | head
