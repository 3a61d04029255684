# GPU-box check script (run through gpurun): [kbench] kernel microbench, parity tests, smoke, bench, [prof] rocprofv3.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
for a in "$@"; do case $a in
  kbench) timeout 600 python tools/kernel_bench.py > gpurun_out/kernel_bench.log 2>&1; tail -20 gpurun_out/kernel_bench.log;;
  tests) timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log;;
  smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log;;
  bench) timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log;;
  benchpab) timeout 900 python bench.py --steps 30 --warmup 30 --pab > gpurun_out/bench_pab.log 2>&1; tail -2 gpurun_out/bench_pab.log;;
  prof) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1);;
esac; done
# PMC passes (own runs, --kernel-trace only): usage  bash tools/run_gpu_checks.sh pmc
if [ "$1" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc1 -o pmc -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc1.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o pmc -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc2.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o pmc -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc4 -o pmc -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc4.log 2>&1
  cd $R; ls gpurun_out/pmc*/ | head; tail -3 gpurun_out/pmc1.log
fi
