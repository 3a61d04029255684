# usage (GPU box): bash tools/pmc_gemm.sh   -> gpurun_out/gpmc{1..4}/pmc_results.db ; summarise with tools/pmc_report.py
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/gpmc$i -o pmc -- python $R/tools/pmc_gemm.py > $R/gpurun_out/gpmc$i.log 2>&1
  tail -2 $R/gpurun_out/gpmc$i.log
done
cd $R && python tools/pmc_report.py gpurun_out/gpmc*/pmc_results.db
