mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cvx_kernels.py tests/test_gpu_cogvideox.py -q -x -p no:cacheprovider -k "flash or temporal or attn or stdit3 or cogvideox" > gpurun_out/r3/pytest7.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3/pytest7.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > gpurun_out/r3/bench7a.log 2>&1; tail -1 gpurun_out/r3/bench7a.log | cut -c150-330
timeout 600 python bench.py --steps 30 --warmup 30 --pab --no-cpu-baseline --no-vae --no-t5 > gpurun_out/r3/bench7_pab.log 2>&1; tail -1 gpurun_out/r3/bench7_pab.log | cut -c150-330
timeout 600 python tools/cogvideox_bench.py --steps 3 > gpurun_out/r3/cvx7_ring3.log 2>&1; tail -1 gpurun_out/r3/cvx7_ring3.log | cut -c1-300
timeout 600 python tools/cogvideox_bench.py --steps 3 --flash-variant 12 > gpurun_out/r3/cvx7_ring2.log 2>&1; tail -1 gpurun_out/r3/cvx7_ring2.log | cut -c1-300
timeout 600 python tools/cogvideox_bench.py --steps 3 > gpurun_out/r3/cvx7_ring3b.log 2>&1; tail -1 gpurun_out/r3/cvx7_ring3b.log | cut -c1-300
