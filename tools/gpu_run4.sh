mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sp.py -q -x -p no:cacheprovider -k "lanes or program or shards or dsp or temporal or eight" > gpurun_out/r3/pytest4.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3/pytest4.log
timeout 600 python tools/kernel_bench.py --reps 20 --rounds 2 --rows 4864 > gpurun_out/r3/kernel_bench_4864.log 2>&1; tail -12 gpurun_out/r3/kernel_bench_4864.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/prof_rank8b -o rank8 -- python $R/tools/issue_time.py --dsp-rank 8 --scatter flat --no-overlap --steps 5 > $R/gpurun_out/r3/prof_rank8b.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/r3/prof_rank8b -name "*.db" | head -1) > gpurun_out/r3/rank8b_kernel_stats.txt 2>&1; head -16 gpurun_out/r3/rank8b_kernel_stats.txt | cut -c1-60,100-170
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > gpurun_out/r3/bench4a.log 2>&1; tail -1 gpurun_out/r3/bench4a.log | cut -c1-330
VSYS_CFG_LANES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > gpurun_out/r3/bench4_lanes.log 2>&1; tail -1 gpurun_out/r3/bench4_lanes.log | cut -c1-330
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > gpurun_out/r3/bench4b.log 2>&1; tail -1 gpurun_out/r3/bench4b.log | cut -c1-330
