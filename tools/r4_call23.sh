mkdir -p gpurun_out/c23
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c23
timeout 900 python -m pytest tests/test_gpu_cvx_kernels.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "flash_attn_d64 or without_running_max" 2>&1 | tail -8 > $O/tests.log; tail -4 $O/tests.log
timeout 300 python tools/flash_shape_probe.py --d64 --frames 2 --tokens 17776 --heads 48 --variants 15,144,1017,144,1017 --reps 5 2>/dev/null > $O/flash64.json; grep -A4 "\"variant\"" $O/flash64.json | grep "variant\|ms_med\|tflops" | paste - - - | cut -c1-120
for v in 0 19 0 19; do timeout 400 python tools/cogvideox_bench.py --steps 4 --flash-variant $v 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/cvx.jsonl; done
timeout 600 python bench.py --geometry 720p128f --steps 3 --warmup 1 > $O/bench_720p_static.log 2>&1; tail -1 $O/bench_720p_static.log | grep -o '"ms_per_step": [0-9.]*'
VSYS_FLASH_STATIC=0 timeout 600 python bench.py --geometry 720p128f --steps 3 --warmup 1 > $O/bench_720p_nostatic.log 2>&1; tail -1 $O/bench_720p_nostatic.log | grep -o '"ms_per_step": [0-9.]*'
VSYS_FLASH_W64=0 timeout 600 python bench.py --geometry 720p128f --steps 3 --warmup 1 > $O/bench_720p_old.log 2>&1; tail -1 $O/bench_720p_old.log | grep -o '"ms_per_step": [0-9.]*'
