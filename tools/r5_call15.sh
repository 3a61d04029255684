# round 5, call 15: sequence-parallel GPU tests after the collective check of the exchange setting
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 420 python -m pytest tests/test_gpu_sp.py -q -p no:cacheprovider 2>&1 | tail -4
