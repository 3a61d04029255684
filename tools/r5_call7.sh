# round 5, call 7: the whole GPU suite on the final tree, with durations
mkdir -p gpurun_out/c7
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c7/box.txt 2>&1
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 2>&1 | tail -45 > gpurun_out/c7/pytest_gpu.log; tail -40 gpurun_out/c7/pytest_gpu.log
