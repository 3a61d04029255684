mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for fv in 0 3 0 3; do timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-vae --no-t5 --flash-variant $fv 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('flash variant $fv', j['ms_per_step'])"; done
