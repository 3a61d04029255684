# round-3 GPU call 1: new parity tests + changed kernels' tests, per-rank DSP timing, bench
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_sp.py tests/test_gpu_fulldepth.py -q -x -s -p no:cacheprovider \
   -k "eight_ranks or thirty or stdit3 or exact" > gpurun_out/r3/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 gpurun_out/r3/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_cvx_kernels.py -q -x -p no:cacheprovider > gpurun_out/r3/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r3/pytest_parity.log
timeout 600 python tools/issue_time.py --dsp-rank 8 --sweep > gpurun_out/r3/issue_dsp8.log 2>&1; tail -5 gpurun_out/r3/issue_dsp8.log
timeout 300 python tools/issue_time.py > gpurun_out/r3/issue_1.log 2>&1; tail -1 gpurun_out/r3/issue_1.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench.log 2>&1; tail -1 gpurun_out/r3/bench.log | cut -c1-600
