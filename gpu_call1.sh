mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
(VSYS_TEST_LAB=1 timeout 300 python -m pytest tests/test_gpu_lab.py -q -x -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r2_lab.log
timeout 900 python tools/parity_full_depth.py > gpurun_out/r2_parity_full_depth.json 2> gpurun_out/r2_parity_full_depth.err
timeout 300 python tools/kernel_bench.py --reps 20 --flash-variants 0 > gpurun_out/r2_kbench0.log 2>&1
tail -5 gpurun_out/r2_lab.log; tail -3 gpurun_out/r2_parity_full_depth.err; grep -n verdict gpurun_out/r2_parity_full_depth.json; tail -22 gpurun_out/r2_kbench0.log
