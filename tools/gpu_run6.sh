mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cvx_kernels.py -q -x -p no:cacheprovider -k "flash or temporal or attn" > gpurun_out/r3/pytest6.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3/pytest6.log
timeout 600 python tools/kernel_bench.py --reps 20 --rounds 1 --flash-variants 0,11,10,3,0,11 > gpurun_out/r3/kernel_bench6.log 2>&1; grep -E "flash|temporal|prep" gpurun_out/r3/kernel_bench6.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > gpurun_out/r3/bench6a.log 2>&1; tail -1 gpurun_out/r3/bench6a.log | cut -c150-330
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 --flash-variant 11 > gpurun_out/r3/bench6_ring3.log 2>&1; tail -1 gpurun_out/r3/bench6_ring3.log | cut -c150-330
timeout 600 python tools/cogvideox_bench.py --layers 10 --steps 3 > gpurun_out/r3/cvx6a.log 2>&1; tail -2 gpurun_out/r3/cvx6a.log | cut -c1-300
timeout 600 python tools/cogvideox_bench.py --layers 10 --steps 3 --flash-variant 12 > gpurun_out/r3/cvx6_ring3.log 2>&1; tail -2 gpurun_out/r3/cvx6_ring3.log | cut -c1-300
