# round 5, call 12: the 128-row GEMM geometry on schedule 6 (DMA pieces interleaved with MFMA pairs) against the shipped schedule 3, at a rank's 4864 rows
mkdir -p gpurun_out/c12
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c12/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c12/build.log 2>&1; tail -1 gpurun_out/c12/build.log
timeout 600 python tools/kernel_bench.py --only gemm --rows 4864 --reps 30 --variants 103,106 > gpurun_out/c12/kb_4864.log 2>&1; tail -14 gpurun_out/c12/kb_4864.log
timeout 600 python tools/kernel_bench.py --only gemm --rows 5120 --reps 30 --variants 103,106 > gpurun_out/c12/kb_5120.log 2>&1; tail -11 gpurun_out/c12/kb_5120.log
