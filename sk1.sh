mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "gemm" 2>&1 | tail -15) > gpurun_out/r2_sk_tests.log
tail -6 gpurun_out/r2_sk_tests.log
timeout 600 python tools/kernel_bench.py --reps 20 --rounds 3 --variants 8,70,80 --only gemm > gpurun_out/r2_kbench_sk.log 2>&1
tail -22 gpurun_out/r2_kbench_sk.log
