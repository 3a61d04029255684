mkdir -p gpurun_out; export PYTHONUNBUFFERED=1; export VSYS_LAB=1
timeout 150 python tools/kernel_bench.py --reps 20 --rounds 3 --variants 70,80,81,82,83,84 --only gemm > gpurun_out/r2_kbench_sk_abl.log 2>&1
tail -26 gpurun_out/r2_kbench_sk_abl.log
