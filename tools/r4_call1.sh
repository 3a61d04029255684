# round-4 GPU call 1: AdaLN fold parity + A/B bench on one box, GEMM raster probe (time + FETCH_SIZE + L2 hit counters)
mkdir -p gpurun_out/c1
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_adaln_fold.py -q -p no:cacheprovider -x 2>&1 | tail -30 > $O/fold_tests.log; tail -5 $O/fold_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fulldepth.py -q -p no:cacheprovider -k "not thirty and not latte and not cogvideox and not eight_ranks" 2>&1 | tail -30 > $O/parity.log; tail -5 $O/parity.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_fold.log 2>&1; tail -1 $O/bench_fold.log | cut -c1-400
VSYS_ADALN_FOLD=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_nofold.log 2>&1; tail -1 $O/bench_nofold.log | cut -c1-400
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_fold2.log 2>&1; tail -1 $O/bench_fold2.log | cut -c1-200
timeout 600 python tools/gemm_raster_probe.py > $O/raster_time.jsonl 2> $O/raster_time.err; cat $O/raster_time.jsonl | cut -c1-200
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/rf -o pmc -- python $R/tools/gemm_raster_probe.py --pmc > $O/rf.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rh -o pmc -- python $R/tools/gemm_raster_probe.py --pmc > $O/rh.log 2>&1)
python tools/gemm_raster_probe.py --report $(find $O/rf -name "*.db" | head -1) $(find $O/rh -name "*.db" | head -1) > $O/raster_pmc.jsonl 2> $O/raster_pmc.err; cat $O/raster_pmc.jsonl | cut -c1-300; tail -3 $O/raster_pmc.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5 > $O/prof.log 2>&1)
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5 (round 4, AdaLN fold on)" > $O/kernel_stats.txt 2>&1; head -16 $O/kernel_stats.txt | cut -c1-60,100-170
