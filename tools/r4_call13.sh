# round-4 GPU call: head_dim 64 w64 stream: parity, shape probe at the CogVideoX-5B shape, CogVideoX step with / without
mkdir -p gpurun_out/c13
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c13
timeout 900 python -m pytest tests/test_gpu_cvx_kernels.py -q -p no:cacheprovider -x -k "flash_attn_d64" 2>&1 | tail -15 > $O/tests.log; tail -5 $O/tests.log
timeout 300 python tools/flash_shape_probe.py --d64 --frames 2 --tokens 17776 --heads 48 --variants 15,14,15,14 --reps 5 2>/dev/null > $O/flash64_cvx5b.json; cat $O/flash64_cvx5b.json
