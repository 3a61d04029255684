# round 5, call 14: area-balanced tile dealing of the sharded CogVideoX decode; the configs[4] combination test
mkdir -p gpurun_out/c14
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c14/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c14/build.log 2>&1; tail -1 gpurun_out/c14/build.log
timeout 600 python tools/cogvideox_vae_bench.py --shard 4 > gpurun_out/c14/cvx_vae.log 2>&1; tail -1 gpurun_out/c14/cvx_vae.log | cut -c1-500
timeout 600 python -m pytest tests/test_gpu_sp.py tests/test_gpu_cogvideox_vae.py -q -x -p no:cacheprovider -k "cogvideox" 2>&1 | tail -3
