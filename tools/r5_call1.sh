# round 5, call 1: new paths on hardware (exact-keys cross attention, sharded VAE decode, self-launching bench), a baseline bench line,
# the cross-attention A/B and the per-rank numbers of this box
mkdir -p gpurun_out/c1
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c1/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/c1/smoke.log 2>&1; tail -1 gpurun_out/c1/smoke.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "flash or cross or attn" 2>&1 | tail -5 > gpurun_out/c1/t_flash.log; tail -2 gpurun_out/c1/t_flash.log
timeout 600 python -m pytest tests/test_gpu_vae.py -q -x -p no:cacheprovider -k "frame_ranges or uint8 or decode_matches" 2>&1 | tail -5 > gpurun_out/c1/t_vae.log; tail -2 gpurun_out/c1/t_vae.log
timeout 900 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider -k "bench" 2>&1 | tail -15 > gpurun_out/c1/t_bench.log; tail -3 gpurun_out/c1/t_bench.log
timeout 600 python tools/kernel_bench.py --only flash --reps 30 > gpurun_out/c1/kb_flash.log 2>&1; grep -i "cross\|spatial" gpurun_out/c1/kb_flash.log | tail -8
timeout 600 python bench.py > gpurun_out/c1/bench.log 2>&1; tail -1 gpurun_out/c1/bench.log | cut -c1-400
VSYS_FLASH_EXACT=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-t5 > gpurun_out/c1/bench_noexact.log 2>&1; tail -1 gpurun_out/c1/bench_noexact.log | cut -c100-330
timeout 600 python tools/vae_bench.py --shard 8 > gpurun_out/c1/vae_shard8.log 2>&1; tail -1 gpurun_out/c1/vae_shard8.log
timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/c1/issue_dsp8.log 2>&1; tail -2 gpurun_out/c1/issue_dsp8.log | cut -c1-500
timeout 600 python tools/kernel_bench.py --only gemm --rows 4864 --reps 30 > gpurun_out/c1/kb_gemm_4864.log 2>&1; tail -12 gpurun_out/c1/kb_gemm_4864.log
