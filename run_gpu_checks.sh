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
