mkdir -p gpurun_out/c5
export PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c5
VSYS_LIB=$R/videosys_amd/libvideosys_amd_lab.so timeout 300 python tools/flash_w64_stamps.py > $O/stamps.json 2> $O/stamps.err; cat $O/stamps.json; tail -3 $O/stamps.err
