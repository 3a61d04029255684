# round 5, call 5: p2p exchange kernel with few fat workgroups / padded counters / no fences; config-5 combination test; test durations
mkdir -p gpurun_out/c5
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c5/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c5/build.log 2>&1; tail -1 gpurun_out/c5/build.log
timeout 900 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider --durations=12 2>&1 | tail -40 > gpurun_out/c5/t_sp.log; tail -18 gpurun_out/c5/t_sp.log
timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/c5/issue_dsp8_p2p.log 2>&1; tail -1 gpurun_out/c5/issue_dsp8_p2p.log | cut -c1-330
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 > gpurun_out/c5/issue_dsp8_rccl.log 2>&1; tail -1 gpurun_out/c5/issue_dsp8_rccl.log | cut -c1-330
timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c5/issue_dsp8_p2p_noov.log 2>&1; tail -1 gpurun_out/c5/issue_dsp8_p2p_noov.log | cut -c1-330
VSYS_DSP_P2P=0 timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c5/issue_dsp8_rccl_noov.log 2>&1; tail -1 gpurun_out/c5/issue_dsp8_rccl_noov.log | cut -c1-330
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c5/prof_rank8 -o rank8 -- python $GRAFT_REPO_ROOT/tools/issue_time.py --dsp-rank 8 --no-overlap --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/c5/prof_rank8.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/c5/prof_rank8 -name "*.db" | head -1) "# rocprofv3 --kernel-trace of: python tools/issue_time.py --dsp-rank 8 --no-overlap --steps 5 (ONE rank of an 8-way DSP group at config 2, one-kernel peer-to-peer exchange with every peer folded onto this rank, 7 steps in the trace; round 5)" > gpurun_out/c5/rank8_kernel_stats.txt 2>&1; head -12 gpurun_out/c5/rank8_kernel_stats.txt | cut -c1-60,100-170
rm -rf gpurun_out/c5/prof_rank8
