# round 5, call 10: weight-prefetch hint A/B — one rank of eight (auto = on below 16384 rows) and the whole step (opt-in), two alternating rounds
mkdir -p gpurun_out/c10
export PYTHONUNBUFFERED=1
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; date -u; } > gpurun_out/c10/box.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c10/build.log 2>&1; tail -1 gpurun_out/c10/build.log
for i in 1 2; do
VSYS_PREFETCH_WEIGHTS=0 timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c10/rank_off_$i.log 2>&1; tail -1 gpurun_out/c10/rank_off_$i.log | cut -c60-130
VSYS_PREFETCH_WEIGHTS=1 timeout 600 python tools/issue_time.py --dsp-rank 8 --no-overlap > gpurun_out/c10/rank_on_$i.log 2>&1; tail -1 gpurun_out/c10/rank_on_$i.log | cut -c60-130
done
VSYS_PREFETCH_WEIGHTS=0 timeout 300 python tools/issue_time.py > gpurun_out/c10/one_off.log 2>&1; tail -1 gpurun_out/c10/one_off.log | cut -c60-130
VSYS_PREFETCH_WEIGHTS=1 timeout 300 python tools/issue_time.py > gpurun_out/c10/one_on.log 2>&1; tail -1 gpurun_out/c10/one_on.log | cut -c60-130
VSYS_PREFETCH_WEIGHTS=0 timeout 300 python tools/issue_time.py > gpurun_out/c10/one_off2.log 2>&1; tail -1 gpurun_out/c10/one_off2.log | cut -c60-130
VSYS_PREFETCH_WEIGHTS=1 timeout 300 python tools/issue_time.py > gpurun_out/c10/one_on2.log 2>&1; tail -1 gpurun_out/c10/one_on2.log | cut -c60-130
timeout 300 python -m pytest tests/test_gpu_sp.py -q -x -p no:cacheprovider -k "eight_ranks and 64" 2>&1 | tail -3
