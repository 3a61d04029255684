# round-4 GPU call 2: AdaLN fold with the LDS-staged epilogue: parity + A/B bench + kernel trace
mkdir -p gpurun_out/c2
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c2
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_adaln_fold.py -q -p no:cacheprovider -x 2>&1 | tail -30 > $O/fold_tests.log; tail -3 $O/fold_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sp.py tests/test_gpu_fulldepth.py -q -p no:cacheprovider -k "not thirty and not latte and not cogvideox" 2>&1 | tail -40 > $O/parity.log; tail -6 $O/parity.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_fold.log 2>&1; tail -1 $O/bench_fold.log | cut -c150-330
VSYS_ADALN_FOLD=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_nofold.log 2>&1; tail -1 $O/bench_nofold.log | cut -c150-330
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_fold2.log 2>&1; tail -1 $O/bench_fold2.log | cut -c150-330
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5 > $O/prof.log 2>&1)
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5 (round 4, AdaLN fold on, LDS-staged LN epilogue)" > $O/kernel_stats.txt 2>&1; head -14 $O/kernel_stats.txt | cut -c1-75,100-170
