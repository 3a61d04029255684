mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python tools/gemm_yardstick.py > gpurun_out/r2_gemm_yardstick.json 2> gpurun_out/r2_gemm_yardstick.err
tail -3 gpurun_out/r2_gemm_yardstick.err; cat gpurun_out/r2_gemm_yardstick.json
