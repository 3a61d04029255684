mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sp.py tests/test_gpu_pipeline.py tests/test_gpu_latte.py -q -x -p no:cacheprovider > gpurun_out/r3/pytest3.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3/pytest3.log
timeout 600 python tools/kernel_bench.py --reps 20 --rounds 2 > gpurun_out/r3/kernel_bench3.log 2>&1; tail -25 gpurun_out/r3/kernel_bench3.log
timeout 600 python tools/issue_time.py --dsp-rank 8 --sweep > gpurun_out/r3/issue_dsp8_prog.log 2>&1; cat gpurun_out/r3/issue_dsp8_prog.log | tail -4
timeout 600 python tools/issue_time.py --dsp-rank 8 --scatter flat --no-overlap --program 0 > gpurun_out/r3/issue_dsp8_noprog.log 2>&1; tail -1 gpurun_out/r3/issue_dsp8_noprog.log
timeout 300 python tools/issue_time.py > gpurun_out/r3/issue_1_prog.log 2>&1; tail -1 gpurun_out/r3/issue_1_prog.log
timeout 900 python tools/issue_time.py --geometry 720p128f --steps 3 > gpurun_out/r3/issue_720p_1.log 2>&1; tail -1 gpurun_out/r3/issue_720p_1.log
timeout 900 python tools/issue_time.py --geometry 720p128f --dsp-rank 8 --sweep --steps 3 > gpurun_out/r3/issue_720p_dsp8.log 2>&1; tail -4 gpurun_out/r3/issue_720p_dsp8.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench3.log 2>&1; tail -1 gpurun_out/r3/bench3.log | cut -c1-400
