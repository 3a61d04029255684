mkdir -p gpurun_out/c30
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c30
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/a -o a -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vae --no-t5 > $O/a.log 2>&1
VSYS_FLASH_W64=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/b -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vae --no-t5 > $O/b.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/a -name "*.db" | head -1) "# default dispatch" > $O/a.txt 2>&1; grep "flash\|prep_kv\|gemm2_kernel<3" $O/a.txt | cut -c1-60,100-170
python tools/prof_summary.py $(find $O/b -name "*.db" | head -1) "# VSYS_FLASH_W64=1 (persistent stream without running max for the spatial shape)" > $O/b.txt 2>&1; grep "flash\|prep_kv\|gemm2_kernel<3" $O/b.txt | cut -c1-60,100-170
tail -1 $O/a.log | grep -o '"ms_per_step": [0-9.]*'; tail -1 $O/b.log | grep -o '"ms_per_step": [0-9.]*'
