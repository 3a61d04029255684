# round 5, call 13: the other BASELINE configurations on the final tree: Latte config 1, CogVideoX-5B config 5 (step, decode, sharded decode)
mkdir -p gpurun_out/c13
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c13/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c13/build.log 2>&1; tail -1 gpurun_out/c13/build.log
timeout 600 python tools/latte_config1.py --cpu-steps 0 > gpurun_out/c13/latte.log 2>&1; tail -1 gpurun_out/c13/latte.log | cut -c1-400
timeout 600 python tools/cogvideox_bench.py --steps 3 > gpurun_out/c13/cvx.log 2>&1; tail -1 gpurun_out/c13/cvx.log | cut -c1-400
timeout 600 python tools/cogvideox_bench.py --steps 4 --pab > gpurun_out/c13/cvx_pab.log 2>&1; tail -1 gpurun_out/c13/cvx_pab.log | cut -c1-400
timeout 600 python tools/cogvideox_vae_bench.py --shard 4 > gpurun_out/c13/cvx_vae.log 2>&1; tail -1 gpurun_out/c13/cvx_vae.log | cut -c1-500
