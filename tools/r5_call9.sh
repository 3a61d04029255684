# round 5, call 9: A/B of the per-rank step (p2p exchange vs pack + a2a stub + unpack) on one box, with the kernel trace of the p2p form
mkdir -p gpurun_out/c9
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c9/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c9/build.log 2>&1; tail -1 gpurun_out/c9/build.log
for i in 1 2; do
timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c9/issue_p2p_$i.log 2>&1; tail -1 gpurun_out/c9/issue_p2p_$i.log | cut -c60-130
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c9/issue_rccl_$i.log 2>&1; tail -1 gpurun_out/c9/issue_rccl_$i.log | cut -c60-130
done
timeout 300 python tools/issue_time.py > gpurun_out/c9/issue_1gpu.log 2>&1; tail -1 gpurun_out/c9/issue_1gpu.log | cut -c60-130
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c9/prof_rank8 -o rank8 -- python $GRAFT_REPO_ROOT/tools/issue_time.py --dsp-rank 8 --no-overlap --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/c9/prof_rank8.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/c9/prof_rank8 -name "*.db" | head -1) "# rocprofv3 --kernel-trace of: python tools/issue_time.py --dsp-rank 8 --no-overlap --steps 5 (ONE rank of an 8-way DSP group at config 2, one-kernel peer-to-peer exchange with every peer folded onto this rank, 7 steps in the trace; round 5 final tree)" > gpurun_out/c9/rank8_kernel_stats.txt 2>&1; head -14 gpurun_out/c9/rank8_kernel_stats.txt | cut -c1-60,100-170
rm -rf gpurun_out/c9/prof_rank8
