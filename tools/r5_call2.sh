# round 5, call 2: the one-kernel peer-to-peer exchange on hardware (in process, and two processes over HIP IPC), chunk-aligned VAE
# shards, per-rank step with both exchange forms
mkdir -p gpurun_out/c2
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c2/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c2/build.log 2>&1; tail -1 gpurun_out/c2/build.log
timeout 600 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider -k "peer_to_peer" 2>&1 | tail -30 > gpurun_out/c2/t_ipc.log; tail -4 gpurun_out/c2/t_ipc.log
timeout 900 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider -k "eight_ranks or bench_plain or two_ranks_equals" 2>&1 | tail -30 > gpurun_out/c2/t_sp.log; tail -4 gpurun_out/c2/t_sp.log
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "frame_ranges or keys_exact" 2>&1 | tail -5 > gpurun_out/c2/t_misc.log; tail -2 gpurun_out/c2/t_misc.log
timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/c2/issue_dsp8_p2p.log 2>&1; tail -1 gpurun_out/c2/issue_dsp8_p2p.log | cut -c1-600
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/c2/issue_dsp8_rccl.log 2>&1; tail -1 gpurun_out/c2/issue_dsp8_rccl.log | cut -c1-600
timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c2/issue_dsp8_p2p_noov.log 2>&1; tail -1 gpurun_out/c2/issue_dsp8_p2p_noov.log | cut -c1-400
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c2/issue_dsp8_rccl_noov.log 2>&1; tail -1 gpurun_out/c2/issue_dsp8_rccl_noov.log | cut -c1-400
timeout 600 python tools/vae_bench.py --shard 8 > gpurun_out/c2/vae_shard8.log 2>&1; tail -1 gpurun_out/c2/vae_shard8.log
timeout 300 python tools/issue_time.py > gpurun_out/c2/issue_1gpu.log 2>&1; tail -1 gpurun_out/c2/issue_1gpu.log | cut -c1-300
