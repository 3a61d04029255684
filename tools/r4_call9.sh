# round-4 GPU call 10: PAB broadcasts folded into the preceding GEMMs: parity, config-3 bench with / without, base step on the same box
mkdir -p gpurun_out/c9
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c9
timeout 900 python -m pytest tests/test_gpu_adaln_fold.py -q -p no:cacheprovider -x 2>&1 | tail -15 > $O/fold_tests.log; tail -6 $O/fold_tests.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "pab" 2>&1 | tail -15 > $O/pab_tests.log; tail -4 $O/pab_tests.log
timeout 600 python bench.py --steps 30 --warmup 30 --pab --no-vae --no-t5 > $O/bench_pab_fold.log 2>&1; tail -1 $O/bench_pab_fold.log | cut -c1-300
VSYS_PAB_FOLD=0 timeout 600 python bench.py --steps 30 --warmup 30 --pab --no-vae --no-t5 > $O/bench_pab_nofold.log 2>&1; tail -1 $O/bench_pab_nofold.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-vae --no-t5 --no-cpu-baseline > $O/bench_base.log 2>&1; tail -1 $O/bench_base.log | cut -c1-300
