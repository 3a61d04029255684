# round 5, call 11: the whole GPU suite on the final tree (after the Ulysses / Latte p2p exchange and the sharded tiled decode)
mkdir -p gpurun_out/c11
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c11/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/c11/smoke.log 2>&1; tail -1 gpurun_out/c11/smoke.log
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -30 > gpurun_out/c11/pytest_gpu.log; tail -24 gpurun_out/c11/pytest_gpu.log
