mkdir -p gpurun_out; export PYTHONUNBUFFERED=1; R=$PWD
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r2_pytest_gpu.log
tail -6 gpurun_out/r2_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1; tail -1 gpurun_out/r2_smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_bench.log 2>&1; tail -1 gpurun_out/r2_bench.log | cut -c1-600
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae --no-t5 > $R/gpurun_out/r2_prof.log 2>&1)
ls gpurun_out/r2prof | head
