# round-4 GPU call 4: w64 flash placement variants + lab ablations (no DMA / no softmax VALU), PMC on the spatial shape
mkdir -p gpurun_out/c4
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c4
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "w64" 2>&1 | tail -5 > $O/w64_tests.log; tail -2 $O/w64_tests.log
timeout 600 python tools/kernel_bench.py --flash-variants 15,140,141,143 --only flash --rounds 4 > $O/kbench.log 2>&1; grep -i "flash\|check" $O/kbench.log | head -20
VSYS_LIB=$R/videosys_amd/libvideosys_amd_lab.so timeout 600 python tools/kernel_bench.py --flash-variants 141,148,149 --only flash --rounds 3 > $O/kbench_lab.log 2>&1; grep -i "flash_" $O/kbench_lab.log | head -20
(cd /tmp && export TMPDIR=/tmp && VSYS_KB_SPATIAL_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc -o pmc -- python $R/tools/kernel_bench.py --flash-variants 141 --only flash --reps 3 --rounds 1 > $O/pmc.log 2>&1)
python tools/pmc_report.py $(find $O/pmc -name "*.db" | head -1) > $O/pmc_flash.txt 2>&1; grep -A12 "w64" $O/pmc_flash.txt | head -30
(cd /tmp && export TMPDIR=/tmp && VSYS_KB_SPATIAL_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $O/pmc2 -o pmc -- python $R/tools/kernel_bench.py --flash-variants 141 --only flash --reps 3 --rounds 1 > $O/pmc2.log 2>&1)
python tools/pmc_report.py $(find $O/pmc2 -name "*.db" | head -1) > $O/pmc_flash2.txt 2>&1; grep -A12 "w64" $O/pmc_flash2.txt | head -30
