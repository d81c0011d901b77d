export TMPDIR=/tmp
i=0
for set in "MfmaUtil MeanOccupancyPerActiveCU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "MemUnitStalled LDSBankConflict" "TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcg/p$i -o p -- ./tools/gemm_bench > /dev/null 2>gpurun_out/pmcg/err$i.txt || echo "pass $i failed"
done
ls gpurun_out/pmcg/*
