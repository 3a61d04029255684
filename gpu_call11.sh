mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_latte.py -q -x -p no:cacheprovider -k "attn or flash or latte or stdit3" 2>&1 | tail -8) > gpurun_out/r2_flash_tests.log
tail -4 gpurun_out/r2_flash_tests.log
timeout 600 python tools/kernel_bench.py --reps 30 --flash-variants 0,3 --variants 0 > gpurun_out/r2_kbench4.log 2>&1
tail -9 gpurun_out/r2_kbench4.log
