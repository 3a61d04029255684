mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r2_sp_tests.log
tail -12 gpurun_out/r2_sp_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-vae --no-t5 > gpurun_out/r2_bench_cpu.log 2>&1; tail -1 gpurun_out/r2_bench_cpu.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['cpu_baseline']))"
