# Round-end evidence on ONE box (gpurun -- 'bash tools/evidence_run.sh'): box identity, smoke, bench line, rocprofv3 kernel trace of the
# bench command, MFMA-busy counters of the hot kernels (with derived MfmaUtil), GEMM traffic counters of the shipped dispatch, the
# per-kernel microbenchmark, config 3, one rank of 8 (peer-to-peer exchange and the RCCL-shaped path), the sharded VAE decode, the
# 720p x 128f step with its kernel trace.  Most important first: a cut-off run still leaves the top.
# Outputs under gpurun_out/ev/; the summaries that are evidence get copied into profiles/r06_* afterwards.
mkdir -p gpurun_out/ev
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
{ hostname; rocm-smi --showserial --showproductname --showuniqueid 2>/dev/null | grep -i "serial\|unique\|card series" | head -6; date -u; } > gpurun_out/ev/box.txt 2>&1; head -3 gpurun_out/ev/box.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/ev/smoke.log 2>&1; tail -1 gpurun_out/ev/smoke.log
timeout 900 python bench.py > gpurun_out/ev/bench.log 2>&1; tail -1 gpurun_out/ev/bench.log | cut -c150-330
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ev/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5 > $R/gpurun_out/ev/prof.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/ev/prof -name "*.db" | head -1) "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vae --no-t5   (config 2; 2 warm-up + 5 timed + 3 instrumented-replay steps = 10 denoise steps in the trace; round-6 final tree; box: $(grep -i unique gpurun_out/ev/box.txt | head -1))" > gpurun_out/ev/kernel_stats.txt 2>&1; head -14 gpurun_out/ev/kernel_stats.txt | cut -c1-60,100-170
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/ev/pmc2 -o pmc -- python $R/tools/pmc_probe.py > $R/gpurun_out/ev/pmc2.log 2>&1
VSYS_GEMM_ALL_SHAPES=1 VSYS_GEMM_VARIANTS=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/ev/tf -o pmc -- python $R/tools/pmc_gemm.py > $R/gpurun_out/ev/tf.log 2>&1
VSYS_GEMM_ALL_SHAPES=1 VSYS_GEMM_VARIANTS=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/ev/tw -o pmc -- python $R/tools/pmc_gemm.py > $R/gpurun_out/ev/tw.log 2>&1
cd $R
python tools/pmc_report.py $(find gpurun_out/ev/pmc2 -name "*.db" | head -1) > gpurun_out/ev/pmc_hot.txt 2>&1; grep -c . gpurun_out/ev/pmc_hot.txt
python tools/gemm_traffic.py $(find gpurun_out/ev/tf -name "*.db" | head -1) $(find gpurun_out/ev/tw -name "*.db" | head -1) > gpurun_out/ev/gemm_traffic.json 2> gpurun_out/ev/gemm_traffic.err; head -c 300 gpurun_out/ev/gemm_traffic.json; tail -2 gpurun_out/ev/gemm_traffic.err
timeout 600 python tools/kernel_bench.py --reps 30 --flash-variants 0,22 > gpurun_out/ev/kb.log 2>&1; grep -i "temporal\|cross\|spatial" gpurun_out/ev/kb.log | tail -8
timeout 600 python tools/kernel_bench.py --only gemm --rows 4864 --reps 50 > gpurun_out/ev/kb_4864.log 2>&1; tail -6 gpurun_out/ev/kb_4864.log
timeout 900 python bench.py --steps 30 --warmup 30 --pab --no-cpu-baseline --no-vae --no-t5 > gpurun_out/ev/bench_pab.log 2>&1; tail -1 gpurun_out/ev/bench_pab.log | cut -c150-330
timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/ev/issue_dsp8.log 2>&1; tail -1 gpurun_out/ev/issue_dsp8.log | cut -c1-300
VSYS_GEMM_MF16=0 VSYS_TEMPORAL_V5=0 timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/ev/issue_dsp8_r5kernels.log 2>&1; tail -1 gpurun_out/ev/issue_dsp8_r5kernels.log | cut -c1-300
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/ev/issue_dsp8_rccl.log 2>&1; tail -1 gpurun_out/ev/issue_dsp8_rccl.log | cut -c1-300
timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/ev/issue_dsp8_noov.log 2>&1; tail -1 gpurun_out/ev/issue_dsp8_noov.log | cut -c1-300
timeout 300 python tools/vae_bench.py --shard 8 > gpurun_out/ev/vae_shard8.log 2>&1; tail -1 gpurun_out/ev/vae_shard8.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ev/prof720 -o bench -- python $R/bench.py --geometry 720p128f --steps 3 --warmup 1 > $R/gpurun_out/ev/bench_720p.log 2>&1); tail -1 gpurun_out/ev/bench_720p.log | cut -c150-330
python tools/prof_summary.py $(find gpurun_out/ev/prof720 -name "*.db" | head -1) "# rocprofv3 --kernel-trace --stats -- python bench.py --geometry 720p128f --steps 3 --warmup 1   (1280x720x128f: 273 600 token rows on one GPU; 1 warm-up + 3 timed + 3 instrumented-replay steps in the trace; round-6 final tree; box: $(grep -i unique gpurun_out/ev/box.txt | head -1))" > gpurun_out/ev/kernel_stats_720p128f.txt 2>&1; head -8 gpurun_out/ev/kernel_stats_720p128f.txt | cut -c1-60,100-170
timeout 600 python tools/issue_time.py --dsp-rank 8 --geometry 720p128f --steps 2 > gpurun_out/ev/issue_dsp8_720p.log 2>&1; tail -1 gpurun_out/ev/issue_dsp8_720p.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ev/rk8 -o r -- python $R/tools/issue_time.py --dsp-rank 8 --no-overlap --steps 5 > $R/gpurun_out/ev/rk8.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/ev/rk8 -name "*.db" | head -1) "# rocprofv3 --kernel-trace of: python tools/issue_time.py --dsp-rank 8 --no-overlap --steps 5 (ONE rank of an 8-way DSP group at config 2, wire stubbed; round-6 final tree; box: $(grep -i unique gpurun_out/ev/box.txt | head -1))" > gpurun_out/ev/dsp_rank8_kernel_stats.txt 2>&1
VSYS_GEMM_ROWS128_RING=0 timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/ev/issue_dsp8_noov_noring.log 2>&1; tail -1 gpurun_out/ev/issue_dsp8_noov_noring.log | cut -c1-300
timeout 200 python tools/cold_weight_probe.py > gpurun_out/ev/cold_weight.log 2>&1
rm -rf gpurun_out/ev/rk8 gpurun_out/ev/prof gpurun_out/ev/prof720 gpurun_out/ev/pmc2 gpurun_out/ev/tf gpurun_out/ev/tw
