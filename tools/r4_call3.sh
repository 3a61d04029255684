# round-4 GPU call 3: the 64-rows-per-wave flash kernel (hand-allocated asm loop): parity, microbench A/B, bench A/B
mkdir -p gpurun_out/c3
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c3
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "w64 or x_mask or attn" 2>&1 | tail -30 > $O/w64_tests.log; tail -5 $O/w64_tests.log
timeout 600 python tools/kernel_bench.py --flash-variants 15,14,15,14 --only flash > $O/kbench.log 2>&1; grep -i "flash\|check" $O/kbench.log | head -20
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_w64.log 2>&1; tail -1 $O/bench_w64.log | cut -c150-330
VSYS_FLASH_W64=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_now64.log 2>&1; tail -1 $O/bench_now64.log | cut -c150-330
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-t5 > $O/bench_w64b.log 2>&1; tail -1 $O/bench_w64b.log | cut -c150-330
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5 > $O/prof.log 2>&1)
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5 (round 4: AdaLN fold + w64 flash)" > $O/kernel_stats.txt 2>&1; head -14 $O/kernel_stats.txt | cut -c1-75,100-170
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/pmc -o pmc -- python $R/tools/kernel_bench.py --flash-variants 14,15 --only flash --reps 3 > $O/pmc.log 2>&1)
python tools/pmc_report.py $(find $O/pmc -name "*.db" | head -1) > $O/pmc_flash.txt 2>&1; grep -A12 "flash_attn" $O/pmc_flash.txt | head -60
