mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider -k "eight" > gpurun_out/r3/pytest5.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3/pytest5.log
timeout 600 python tools/issue_time.py --dsp-rank 8 --sweep > gpurun_out/r3/issue_dsp8_lanes.log 2>&1; cat gpurun_out/r3/issue_dsp8_lanes.log | tail -5
timeout 900 python tools/issue_time.py --geometry 720p128f --dsp-rank 8 --scatter flat --no-overlap --lanes --steps 3 > gpurun_out/r3/issue_720p_dsp8_lanes.log 2>&1; tail -1 gpurun_out/r3/issue_720p_dsp8_lanes.log
