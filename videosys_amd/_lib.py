"""ctypes binding of libvideosys_amd.so (the C ABI declared in include/videosys_amd.h).

The library is the product: there is NO fallback.  Importing this module on a machine where the shared object has
not been built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C videosys_amd/csrc``) raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VSYS_LIB: another build of the same library, e.g. the -DVSYS_LAB flavour the measurement tools load; never a fallback)
LIB_PATH = os.environ.get("VSYS_LIB") or os.path.join(_HERE, "libvideosys_amd.so")

_i64, _f32, _ptr, _int = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_int

# name -> argtypes, exactly as declared in include/videosys_amd.h
SIGNATURES = {
    "vsys_abi_version": [],
    "vsys_device_count": [],
    "vsys_tune_gemm_variant": [_int],
    "vsys_tune_flash_variant": [_int],
    "vsys_gemm_raster_probe": [_i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    "vsys_gemm_bf16": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _int, _ptr, _i64, _i64, _ptr, _i64,
                       _ptr, _i64, _ptr],
    "vsys_gemm_bf16_ln": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _int, _ptr, _i64, _f32, _ptr],
    "vsys_gemm_bf16_stats": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr],
    "vsys_gemm_bf16_gate_res_add": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr,
                                    _ptr, _ptr, _i64, _ptr],
    "vsys_adaln_prescale": [_ptr, _i64, _i64, _ptr, _ptr],
    "vsys_ln_row_stats": [_ptr, _i64, _i64, _ptr, _i64, _ptr],
    "vsys_linear_small": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _int, _int, _ptr],
    "vsys_adaln_modulate": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_mod_table": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "vsys_timestep_embedding": [_ptr, _ptr, _i64, _i64, _ptr],
    "vsys_patch_embed": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_final_layer": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                         _f32, _ptr],
    "vsys_cfg_euler_step": [_ptr, _ptr, _i64, _i64, _i64, _i64, _f32, _f32, _ptr],
    "vsys_add_rows": [_ptr, _ptr, _i64, _ptr],
    "vsys_cfg_linear_step": [_ptr, _ptr, _i64, _i64, _i64, _i64, _f32, _f32, _f32, _int, _ptr],
    "vsys_add_bcast_rows": [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    "vsys_copy_4d": [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_attn_prep_kv": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_flash_attn_d72": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_flash_attn_d72_exact": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_flash_attn_d72_kb": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _ptr],
    "vsys_gemm_bf16_gate2": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr, _i64,
                             _ptr, _i64, _ptr],
    "vsys_ln_modulate": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_gate_add_rows": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_im2col_patch": [_ptr, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_unpatchify_cvx": [_ptr, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_attn_prep_kv64": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _f32,
                            _ptr],
    "vsys_flash_attn_d64": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64,
                            _f32, _ptr],
    "vsys_flash_attn_d64_kb": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64,
                            _f32, _f32, _ptr],
    "vsys_attn_temporal_d72": [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_gather_rows": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "vsys_rms_norm_rows": [_ptr, _ptr, _ptr, _i64, _i64, _f32, _ptr],
    "vsys_geglu": [_ptr, _ptr, _i64, _i64, _ptr],
    "vsys_gemm_skinny_slices": [_ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_splitk_reduce": [_ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr],
    "vsys_t5_attention_mfma": [_ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "vsys_splitk_reduce_t": [_ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr],
    "vsys_t5_attention": [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    "vsys_copy_4d_batch": [_ptr, _ptr, _i64, _ptr, _ptr],
    "vsys_conv_bf16": [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                       _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_gn_stats": [_ptr, _ptr, _i64, _i64, _i64, _f32, _ptr, _i64, _ptr, _ptr],
    "vsys_gn_apply": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _int, _ptr],
    "vsys_regrid": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    "vsys_subsample": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_spatial_norm_apply": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "vsys_blend_edge": [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_d2s_time": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr],
    "vsys_vae_first_im2col": [_ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr],
    "vsys_extract_planar": [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _ptr],
    "vsys_softmax_rows": [_ptr, _ptr, _i64, _i64, _i64, _ptr],
    "vsys_patch_embed_shard": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_final_layer_tokens": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _f32, _ptr],
    "vsys_unpatchify_tokens": [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "vsys_p2p_exchange": [_ptr, _i64, _ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr],
    # set-up of the peer-to-peer exchange (host side; no stream)
    "vsys_p2p_alloc": [_i64, _i64, _ptr], "vsys_p2p_free": [_ptr], "vsys_p2p_ipc_export": [_ptr, _ptr],
    "vsys_p2p_ipc_open": [_ptr, _ptr], "vsys_p2p_ipc_close": [_ptr],
    # launch programs (no trailing stream: the stream table is an argument)
    "vsys_program_op_info": [_int, _ptr, _ptr],
    "vsys_program_run": [_ptr, _i64, _ptr, _i64, _ptr],
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
            "videosys_amd has no CPU/PyTorch fallback for its kernels."
        )
    lib = ctypes.CDLL(LIB_PATH)
    missing = [name for name in SIGNATURES if not hasattr(lib, name)]
    if missing:
        # (a shipped library from before the current sources — e.g. on a box without hipcc, where build() cannot rebuild it: never run
        #  old kernels against new host code)
        raise RuntimeError(f"{LIB_PATH} is stale: it does not export {missing[:6]}{'...' if len(missing) > 6 else ''} that this version of "
                           "videosys_amd binds; rebuild it on a box with the ROCm compiler (python -c 'import __graft_entry__ as g; g.build()')")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _int
    if hasattr(lib, "vsys_lab_flash_debug_buffer"):   # -DVSYS_LAB build (include/videosys_amd_lab.h)
        lib.vsys_lab_flash_debug_buffer.argtypes = [_ptr]
        lib.vsys_lab_flash_debug_buffer.restype = _int
        lib.vsys_gemm_streamk_plan.argtypes = [_int, _int, _int, _ptr, _int, _ptr]
        lib.vsys_gemm_streamk_plan.restype = _int
    lib.vsys_strerror.argtypes = [_int]
    lib.vsys_strerror.restype = ctypes.c_char_p
    _lib = lib
    return lib


class VsysError(RuntimeError):
    pass


# ---- PyTorch custom-op route (csrc/torch_binding.cpp -> libvideosys_torch.so): torch.ops.vsys.launch / torch.ops.vsys.program_run.
# The product path: ops._call and program.Program.run go through the dispatcher whenever the fragment is there (build() compiles
# it); VSYS_TORCH_OPS=0 or a tree without the fragment binds the same extern "C" functions through ctypes instead.
TORCH_LIB_PATH = os.path.join(_HERE, "libvideosys_torch.so")
_torch_ops = False   # False = not looked for yet, None = not available


def torch_ops():
    """``torch.ops.vsys`` with the fragment loaded, or None (ctypes route)."""
    global _torch_ops
    if _torch_ops is False:
        _torch_ops = None
        if os.environ.get("VSYS_TORCH_OPS", "1") != "0" and os.path.exists(TORCH_LIB_PATH) and not os.environ.get("VSYS_LIB"):
            import torch

            load()                                   # libvideosys_amd.so first: the fragment links against it
            try:
                torch.ops.load_library(TORCH_LIB_PATH)
                _torch_ops = torch.ops.vsys
            except OSError as e:                     # (a fragment built against another torch: say so once, use ctypes)
                import warnings

                warnings.warn(f"{TORCH_LIB_PATH} could not be loaded ({e}); the C ABI is bound through ctypes instead")
    return _torch_ops


def check(code: int, what: str):
    if code != 0:
        msg = load().vsys_strerror(code).decode()
        raise VsysError(f"{what}: {msg} (code {code})")
