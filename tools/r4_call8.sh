# round-4 GPU call 9: persistent w64 flash as default: parity, stamps, microbench, in-situ bench with and without
mkdir -p gpurun_out/c8
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c8
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "w64 or flash or attn" 2>&1 | tail -12 > $O/w64_tests.log; tail -4 $O/w64_tests.log
VSYS_LIB=$R/videosys_amd/libvideosys_amd_lab.so timeout 300 python tools/flash_w64_stamps.py > $O/stamps.json 2> $O/stamps.err; tail -22 $O/stamps.json; tail -3 $O/stamps.err
timeout 600 python tools/kernel_bench.py --flash-variants 15,16,15,16 --only flash --rounds 3 > $O/kbench.log 2>&1; grep -i "flash\|check" $O/kbench.log | head -20
VSYS_FLASH_W64=0 timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_old.log 2>&1; tail -1 $O/bench_old.log | cut -c1-400
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_w64p.log 2>&1; tail -1 $O/bench_w64p.log | cut -c1-400
VSYS_FLASH_W64=0 timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_old2.log 2>&1; tail -1 $O/bench_old2.log | cut -c1-400
