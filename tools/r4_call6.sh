# round-4 GPU call 6: persistent w64 flash: parity + microbench
mkdir -p gpurun_out/c6
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c6
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "w64" 2>&1 | tail -12 > $O/w64_tests.log; tail -6 $O/w64_tests.log
timeout 600 python tools/kernel_bench.py --flash-variants 15,141,16,15,141,16 --only flash --rounds 3 > $O/kbench.log 2>&1; grep -i "flash\|check" $O/kbench.log | head -20
