mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_cvx_kernels.py tests/test_gpu_cogvideox.py -q -x -p no:cacheprovider > gpurun_out/r3/pytest8.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/pytest8.log
for v in 0 13 12 0 13; do timeout 600 python tools/cogvideox_bench.py --steps 3 --flash-variant $v > gpurun_out/r3/cvx8_v$v.log 2>&1; echo "variant $v: $(tail -1 gpurun_out/r3/cvx8_v$v.log | cut -c1-120)"; done
