cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in kbench17 kbench19; do
  rm -rf $R/gpurun_out/pmc_$b
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_$b -o p -- $R/build/$b 8 256 > /dev/null 2>&1
done
