/* Lab entry points of libvideosys_amd.so — present ONLY in -DVSYS_LAB builds (VSYS_LAB=1 python -c "import __graft_entry__ as
 * g; g.build(force=True)").  Such a build additionally accepts the ablation / stamp ids of vsys_tune_gemm_variant (18, 48: schedule
 * 8 without epilogue / without HBM stores; 31: cycle stamps of the wide-tile kernel; 40: 5-slot ring; 61-64, 71-74, 78: ping-pong
 * GEMM without in-loop DMA / fragment reads / stores, or with s_memtime stamps) and of vsys_tune_flash_variant (1: K/V tiles not
 * fetched; 2: phase timers).  THE OUTPUT OF THOSE VARIANTS IS NOT VALID; they exist to price parts of a kernel
 * (tools/gemm4_probe.py, tools/gemm_stamps.py, tools/pmc_flash.py).  Nothing in videosys_amd/ uses this header. */
#ifndef VIDEOSYS_AMD_LAB_H
#define VIDEOSYS_AMD_LAB_H
#ifdef __cplusplus
extern "C" {
#endif

/* device buffer the stamp variants write their int64 / uint64 accumulators to (NULL = off) */
int vsys_lab_flash_debug_buffer(void* dev_u64);

#ifdef __cplusplus
}
#endif
#endif
