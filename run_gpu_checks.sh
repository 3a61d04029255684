# GPU-box check script (run through gpurun): parity tests, smoke, bench, rocprofv3 kernel trace.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
rm -f gpurun_out/pytest_gpu.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
if [ "$1" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
  cd $R
  ls -la gpurun_out/prof/* | head -20 >> gpurun_out/prof.log
fi
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
