mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "gemm" 2>&1 | tail -15) > gpurun_out/r2_gemm4_tests.log
tail -5 gpurun_out/r2_gemm4_tests.log
timeout 600 python tools/gemm4_probe.py > gpurun_out/r2_gemm4_probe2.json 2> gpurun_out/r2_gemm4_probe2.err
tail -3 gpurun_out/r2_gemm4_probe2.err; cat gpurun_out/r2_gemm4_probe2.json
timeout 600 python tools/kernel_bench.py --reps 20 --variants 8,20,70 --only gemm > gpurun_out/r2_kbench2.log 2>&1
tail -16 gpurun_out/r2_kbench2.log
