mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/prof_rank8 -o rank8 -- python $R/tools/issue_time.py --dsp-rank 8 --scatter flat --no-overlap --steps 5 > $R/gpurun_out/r3/prof_rank8.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/r3/prof_rank8 -name "*.db" | head -1) > gpurun_out/r3/rank8_kernel_stats.txt 2>&1; head -40 gpurun_out/r3/rank8_kernel_stats.txt
timeout 1500 python -m pytest tests/test_gpu_fulldepth.py -q -x -s -p no:cacheprovider -k "thirty or exact" > gpurun_out/r3/pytest_new2.log 2>&1; echo "new tests rc=$?"; grep "fulldepth\]" gpurun_out/r3/pytest_new2.log | cut -c1-300; tail -3 gpurun_out/r3/pytest_new2.log
