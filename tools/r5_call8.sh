# round 5, call 8: the peer-to-peer exchange under Latte and CogVideoX (Ulysses), flags raised by the launch's last problem
mkdir -p gpurun_out/c8
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c8/build.log 2>&1; tail -1 gpurun_out/c8/build.log
timeout 1200 python -m pytest tests/test_gpu_sp.py -q -p no:cacheprovider --durations=8 2>&1 | tail -40 > gpurun_out/c8/t_sp.log; tail -16 gpurun_out/c8/t_sp.log
timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c8/issue_dsp8_noov.log 2>&1; tail -1 gpurun_out/c8/issue_dsp8_noov.log | cut -c1-300
