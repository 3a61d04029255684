# round-4 GPU call 8: persistent w64 flash with the Q prefetch at the item tail: parity, stamps, microbench
mkdir -p gpurun_out/c7
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c7
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "w64" 2>&1 | tail -12 > $O/w64_tests.log; tail -4 $O/w64_tests.log
VSYS_LIB=$R/videosys_amd/libvideosys_amd_lab.so timeout 300 python tools/flash_w64_stamps.py > $O/stamps.json 2> $O/stamps.err; cat $O/stamps.json; tail -3 $O/stamps.err
timeout 600 python tools/kernel_bench.py --flash-variants 15,141,16,15,141,16 --only flash --rounds 3 > $O/kbench.log 2>&1; grep -i "flash\|check" $O/kbench.log | head -20
