"""CPU: cross-compile the hot kernels for gfx950 and check the resource usage the design depends on — no scratch
(a spilled staging register serialises the HBM->LDS pipeline), VGPRs within the occupancy the launch assumes."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "videosys_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


import functools


@functools.lru_cache(maxsize=None)
def _usage(src):
    """Per-kernel resource usage of one source (compiled once per test session: several tests read the same file's table)."""
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(CSRC, src), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=CSRC)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?:\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,pattern,max_vgpr", [("gemm_bf16.hip", "gemm_kernel", 256),
                                                  ("attention.hip", "flash_attn_d72_kernel", 256),
                                                  ("attention.hip", "attn_temporal_d72_kernel", 128),
                                                  ("attention_t3.hip", "attn_temporal_d72_v5_kernel", 128),     # four workgroups per CU
                                                  ("gemm2_bf16.hip", "gemm2_kernelILi3ELi2ELi0ELi1E", 256),       # qkv + folded LN on 16x16x32
                                                  ("gemm2_bf16.hip", "gemm2_kernelILi4ELi2ELi0ELi1E", 256)])      # fc1 + folded LN + GELU
def test_hot_kernels_have_no_scratch(src, pattern, max_vgpr):
    u = _usage(src)
    hits = {k: v for k, v in u.items() if pattern in k}
    assert hits, f"{pattern} not found in {src}"
    for name, res in hits.items():
        if "flash_attn_d72_kernelILi0ELi3ELb0E" in name or "flash_attn_d72_kernelILi4ELi3ELb0E" in name:   # (4 = the A/B partner, id 23)
            # three workgroups per CU (168 registers): the peeled MASKED last tile spills, as attention.hip says at the template; the
            # launcher selects this instantiation for unmasked key sequences only (and since round 4 only when the 64-rows-per-wave
            # kernel is switched off), so the spill code never executes
            assert res.get("VGPRs", 0) <= 168
            continue
        assert res.get("ScratchSize", 0) == 0, f"{name} uses scratch: {res}"
        # (SGPR spills are v_writelane / v_readlane into a VGPR, not memory: the resident-K/V flash instantiation carries 72 of them)
        assert res.get("VGPRs Spill", 0) == 0 and res.get("SGPRs Spill", 0) <= 96, f"{name} spills: {res}"
        assert res.get("VGPRs", 0) <= max_vgpr, f"{name}: {res.get('VGPRs')} VGPRs"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_two_workgroup_gemm_fits_two_per_cu():
    """gemm2_bf16.hip is dispatched for the store-only epilogues (bias, bias+GELU): those instantiations must run without
    scratch in 256 VGPRs (2 waves per SIMD) and in half the LDS of a CU."""
    u = _usage("gemm2_bf16.hip")
    hits = {k: v for k, v in u.items() if "gemm2_kernelILi0E" in k or "gemm2_kernelILi1E" in k}
    assert len(hits) >= 2, list(u)
    for name, res in hits.items():
        assert res.get("ScratchSize", 0) == 0, f"{name} uses scratch: {res}"
        assert res.get("VGPRs", 0) <= 256, f"{name}: {res.get('VGPRs')} VGPRs"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_conv_and_producer_gemm_resources():
    """conv_bf16.hip runs two workgroups per CU (<= 256 VGPRs, no scratch); the producer-wave GEMM variant needs three waves per
    SIMD (<= 168 VGPRs) for its store-only epilogues."""
    u = _usage("conv_bf16.hip")
    hits = {k: v for k, v in u.items() if "conv_kernel" in k}
    assert len(hits) == 4      # bf16 / fp32 output x the 32x32x16 and 16x16x32 (round 6, the default) forms
    for name, res in hits.items():
        assert res.get("ScratchSize", 0) == 0 and res.get("VGPRs", 0) <= 256, f"{name}: {res}"
    g = _usage("gemm_bf16.hip")
    prod = {k: v for k, v in g.items() if "ELi8ELi256ELi1ELi1E" in k}
    assert len(prod) >= 2, list(g)
    for name, res in prod.items():
        assert res.get("ScratchSize", 0) == 0 and res.get("VGPRs", 0) <= 168, f"{name}: {res}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_w64_flash_kernel_owns_its_accumulator_registers(tmp_path):
    """attention_w64.hip keeps O, Q and the K / Vt fragments in a[0:223] across several asm statements: the compiler-generated
    code around them must not touch those (no a-register operand below a224 outside ;;#ASMSTART .. ;;#ASMEND), must
    not spill, and the kernel must fit one wave per SIMD (<= 512 registers)."""
    import subprocess

    u = _usage("attention_w64.hip")
    hits = {k: v for k, v in u.items() if "flash_attn_d72_w64" in k}
    assert len(hits) >= 4      # the three placement variants + the persistent form
    u64 = {k: v for k, v in _usage("attention64_w64.hip").items() if "flash_attn_d64_w64" in k}
    assert len(u64) == 3       # the head_dim 64 forms of the same stream (placement 1, 4; 5 = no running max)
    hits.update(u64)
    for name, res in hits.items():
        # (a few SGPRs parked in VGPR lanes are fine — the persistent forms carry ~40 scalars across the statement; scratch is not)
        assert res.get("ScratchSize", 0) == 0 and res.get("VGPRs Spill", 0) == 0 and res.get("SGPRs Spill", 0) <= 16, (name, res)
        assert res.get("VGPRs", 0) <= 256 and 224 <= res.get("AGPRs", 0) <= 256, (name, res)
    # the register-ownership audit itself is part of the build (__graft_entry__.compile_library runs it on every compile of these
    # sources and fails the build); here it runs once more on the current tree
    import __graft_entry__ as ge

    for src in ge.ASM_OWNED:
        ge.audit_asm_register_ownership(os.path.join(CSRC, src))


def test_build_cache_is_keyed_on_content_not_mtime(tmp_path):
    """compile_library's object cache: a source whose mtime moves but whose bytes do not is NOT rebuilt; the key changes with the
    source bytes, the header bytes, the flags and the compiler."""
    import __graft_entry__ as ge

    ge.build()
    src = os.path.join(CSRC, "program.hip")
    st = os.stat(src)
    try:
        os.utime(src, None)                                   # fresh mtime, same bytes
        assert ge.compile_library(ge.LIB, lab=ge.LAB) is False
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    a = tmp_path / "a.hip"
    a.write_text("int x;")
    k0 = ge._digest(str(a), b"hdr", ["-O3"], "cc")
    assert k0 == ge._digest(str(a), b"hdr", ["-O3"], "cc")
    a.write_text("int y;")
    assert ge._digest(str(a), b"hdr", ["-O3"], "cc") != k0
    a.write_text("int x;")
    assert ge._digest(str(a), b"hdr2", ["-O3"], "cc") != k0 and ge._digest(str(a), b"hdr", ["-O2"], "cc") != k0
    assert ge._digest(str(a), b"hdr", ["-O3"], "cc2") != k0 and ge._digest(str(a), b"hdr", ["-O3"], "cc") == k0
