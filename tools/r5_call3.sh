# round 5, call 3: write-through p2p exchange re-timed, temporal attention with one rounding (A/B against the stage-by-stage form),
# the config-5 combination test, the DPM scheduler on the GPU, then the whole GPU suite and a bench line
mkdir -p gpurun_out/c3
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c3/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c3/build.log 2>&1; tail -1 gpurun_out/c3/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "temporal" 2>&1 | tail -12 > gpurun_out/c3/t_temporal.log; tail -3 gpurun_out/c3/t_temporal.log
timeout 900 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider -k "peer_to_peer or cogvideox_pab_ulysses" 2>&1 | tail -30 > gpurun_out/c3/t_sp.log; tail -3 gpurun_out/c3/t_sp.log
timeout 600 python -m pytest tests/test_gpu_cogvideox.py -q -x -p no:cacheprovider -k "dpm" 2>&1 | tail -12 > gpurun_out/c3/t_dpm.log; tail -3 gpurun_out/c3/t_dpm.log
timeout 600 python tools/kernel_bench.py --reps 30 > gpurun_out/c3/kb.log 2>&1; grep -i "temporal\|cross\|spatial" gpurun_out/c3/kb.log | tail -8
timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/c3/issue_dsp8_p2p.log 2>&1; tail -1 gpurun_out/c3/issue_dsp8_p2p.log | cut -c1-330
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/c3/issue_dsp8_rccl.log 2>&1; tail -1 gpurun_out/c3/issue_dsp8_rccl.log | cut -c1-330
timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c3/issue_dsp8_p2p_noov.log 2>&1; tail -1 gpurun_out/c3/issue_dsp8_p2p_noov.log | cut -c1-330
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c3/issue_dsp8_rccl_noov.log 2>&1; tail -1 gpurun_out/c3/issue_dsp8_rccl_noov.log | cut -c1-330
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/c3/bench.log 2>&1; tail -1 gpurun_out/c3/bench.log | cut -c100-330
VSYS_FLASH_VARIANT_NOTE=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-t5 --flash-variant 21 > gpurun_out/c3/bench_fv21.log 2>&1; tail -1 gpurun_out/c3/bench_fv21.log | cut -c100-330
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/c3/pytest_gpu.log; tail -4 gpurun_out/c3/pytest_gpu.log
