mkdir -p gpurun_out/c27
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c27
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -1; } > $O/box.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o pmc -- python $R/tools/pmc_flash_long.py > $O/pmc.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/cvx -o cvx -- python $R/tools/cogvideox_bench.py --steps 3 > $O/cvx.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/p720 -o p720 -- python $R/bench.py --geometry 720p128f --steps 2 --warmup 1 > $O/p720.log 2>&1
cd $R
python tools/pmc_report.py $(find $O/pmc -name "*.db" | head -1) > $O/pmc_flash_long.txt 2>&1; grep -A12 "^flash" $O/pmc_flash_long.txt | grep "^flash\|derived: Mfma\|dur_us\|clock" | cut -c1-120
python tools/prof_summary.py $(find $O/cvx -name "*.db" | head -1) "# rocprofv3 --kernel-trace --stats -- python tools/cogvideox_bench.py --steps 3  (CogVideoX-5B geometry, config 5, 42 layers, one GPU; warm-up + 3 timed steps; round-4 final tree; box: $(head -1 $O/box.txt))" > $O/cvx_kernel_stats.txt 2>&1; head -8 $O/cvx_kernel_stats.txt | cut -c1-60,100-170
python tools/prof_summary.py $(find $O/p720 -name "*.db" | head -1) "# rocprofv3 --kernel-trace --stats -- python bench.py --geometry 720p128f --steps 2 --warmup 1  (configs[3] geometry on one GPU; round-4 final tree; box: $(head -1 $O/box.txt))" > $O/p720_kernel_stats.txt 2>&1; head -10 $O/p720_kernel_stats.txt | cut -c1-60,100-170
tail -1 $O/cvx.log | cut -c1-200
