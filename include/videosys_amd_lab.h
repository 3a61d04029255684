/* Lab entry points of libvideosys_amd.so — present ONLY in -DVSYS_LAB builds (VSYS_LAB=1 python -c "import __graft_entry__ as
 * g; g.build(force=True)").  Such a build additionally accepts the ablation / stamp ids of vsys_tune_gemm_variant (18, 48: schedule
 * 8 without epilogue / without HBM stores; 31: cycle stamps of the wide-tile kernel; 40: 5-slot ring; 61-64, 71-74, 78: ping-pong
 * GEMM without in-loop DMA / fragment reads / stores, or with s_memtime stamps) and of vsys_tune_flash_variant (1: K/V tiles not
 * fetched; 2: phase timers).  THE OUTPUT OF THOSE VARIANTS IS NOT VALID; they exist to price parts of a kernel
 * (tools/gemm4_probe.py, tools/gemm_stamps.py, tools/pmc_flash.py).  Nothing in videosys_amd/ uses this header. */
#ifndef VIDEOSYS_AMD_LAB_H
#define VIDEOSYS_AMD_LAB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* device buffer the stamp variants write their int64 / uint64 accumulators to (NULL = off) */
int vsys_lab_flash_debug_buffer(void* dev_u64);

/* Host-only introspection of the stream-K segment plan of GEMM variant 80 for `ntiles` output tiles of `nt` K-tiles on a
 * persistent grid of `grid` workgroups (no GPU needed; tests/test_host_cpu.py checks coverage and dependency order with it).
 * The tiles [0, ntiles - ntiles % grid) run as whole tiles in the persistent kernel; the plan covers the remaining ones.
 * segs receives grid * (*nseg_max) rows of 4 ints, (*nseg_max) rows per workgroup: linear tile id (-1 ends the workgroup's list),
 * kb | ke << 16 (K-tile range), kind | nsrc << 8 (kind 0 = whole tile, 1 = partial sums dumped to the workgroup's workspace
 * slot, 2 = final range: adds the partial sums of workgroups b - 8 .. b - 8 nsrc, then the epilogue), 0.  Returns the number of
 * rows written, 0 when the shape is not split (no partial round, or a piece would be shorter than two K-tiles), VSYS_ERR_ARG when
 * cap_rows is too small. */
int vsys_gemm_streamk_plan(int ntiles, int nt, int grid, int32_t* segs, int cap_rows, int* nseg_max);

#ifdef __cplusplus
}
#endif
#endif
