# round-4 GPU call: flash without the running max: parity, shape probes, in-situ bench with / without
mkdir -p gpurun_out/c22
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c22
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "w64" 2>&1 | tail -8 > $O/tests.log; tail -4 $O/tests.log
timeout 300 python tools/flash_shape_probe.py --frames 38 --tokens 1024 --variants 15,16,1018,15,16,1018 2>/dev/null > $O/flash_c2.json; grep -A4 "\"variant\"" $O/flash_c2.json | grep "variant\|ms_med\|diff" | paste - - - | cut -c1-120
timeout 300 python tools/flash_shape_probe.py --frames 76 --tokens 3600 --variants 15,141,1017,141,1017 --reps 6 2>/dev/null > $O/flash_720p.json; grep -A4 "\"variant\"" $O/flash_720p.json | grep "variant\|ms_med\|diff" | paste - - - | cut -c1-120
timeout 600 python bench.py --steps 10 --warmup 3 --no-vae --no-t5 --no-cpu-baseline > $O/bench_static.log 2>&1; tail -1 $O/bench_static.log | grep -o '"ms_per_step": [0-9.]*'
VSYS_FLASH_STATIC=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-vae --no-t5 --no-cpu-baseline > $O/bench_nostatic.log 2>&1; tail -1 $O/bench_nostatic.log | grep -o '"ms_per_step": [0-9.]*'
timeout 600 python bench.py --steps 10 --warmup 3 --no-vae --no-t5 --no-cpu-baseline > $O/bench_static2.log 2>&1; tail -1 $O/bench_static2.log | grep -o '"ms_per_step": [0-9.]*'
