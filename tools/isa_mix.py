"""Static instruction mix of the hot loop of each kernel in a built object — no GPU needed (the objects are cross-compiled).

    python tools/isa_mix.py videosys_amd/csrc/build/ship/attention.o [--kernel flash_attn_d72] [--out profiles/<name>.txt]

For every kernel symbol of the gfx950 code object: the innermost loop that holds the most v_mfma instructions (a loop = a backward
branch and its target), its instructions by class, and the figures a kernel writer reads off them: VALU issues per MFMA (MFMA
32x32x16 bf16 occupies the matrix pipe for 8 passes = 32 cycles of a SIMD; what has to issue beside it has to fit those gaps,
MI355X_MICROARCH.md "one wave per SIMD" row), LDS and memory instructions per MFMA, waits and barriers per iteration.  The counts are
per LOOP ITERATION of one wave as the compiler scheduled it; they say nothing about stalls — PMC (profiles/*pmc*) does —
and they include the COLD blocks the compiler places inside a loop's address range (rare out-of-line branches): upper bounds on what executes."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"

CLASSES = [
    ("mfma", r"^v_mfma|^v_smfmac"),
    ("transcendental", r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_"),
    ("convert / pack", r"^v_cvt|^v_pack|^v_perm_b32|^v_bfe|^v_lshl_or|^v_and_or|^v_or3|^v_lshlrev_b32|^v_lshrrev_b32|^v_and_b32|^v_or_b32"),
    ("cross-lane", r"^v_readlane|^v_readfirstlane|^v_writelane|^ds_bpermute|^ds_permute|^v_permlane|^ds_swizzle|_dpp|^v_mov_b32_dpp"),
    ("max / min / cmp / select", r"^v_max|^v_min|^v_cmp|^v_cndmask|^v_med3"),
    ("accumulator moves", r"^v_accvgpr"),
    ("fp32 / packed arithmetic", r"^v_(pk_)?(fma|fmac|mul|add|sub|mad|dot2)"),
    ("other VALU", r"^v_"),
    ("LDS read", r"^ds_read|^ds_load"),
    ("LDS write", r"^ds_write|^ds_store"),
    ("global / buffer load", r"^(global|buffer|flat)_load|^buffer_load"),
    ("global / buffer store", r"^(global|buffer|flat)_(store|atomic)"),
    ("wait", r"^s_waitcnt"),
    ("barrier", r"^s_barrier"),
    ("scalar / branch / other", r"."),
]
VALU = {"transcendental", "convert / pack", "cross-lane", "max / min / cmp / select", "accumulator moves", "fp32 / packed arithmetic", "other VALU"}


def device_disassembly(obj):
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    local = os.path.join(tmp, os.path.basename(obj))
    with open(obj, "rb") as a, open(local, "wb") as b:
        b.write(a.read())
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)   # writes <local>.0.hipv4-...gfx950
    co = [f for f in os.listdir(tmp) if "amdgcn" in f]
    if not co:
        raise SystemExit(f"{obj}: no gfx950 code object inside")
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout


def kernels(asm):
    cur, out = None, collections.OrderedDict()
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if cur is not None and m:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))   # the branch target is in the tail
    return out


def demangle(name):
    try:
        for tool in (f"{LLVM}/llvm-cxxfilt", "c++filt"):
            try:
                return subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                continue
    except Exception:
        pass
    return name


def hot_loop(ins):
    """(start index, end index) of the innermost backward-branch loop holding the most MFMAs."""
    addr_to_i = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", args)
            tgt = None
            if m:
                base = ins[0][0]
                tgt = base + int(m.group(1), 16)
            if tgt is not None and tgt in addr_to_i and addr_to_i[tgt] <= i:
                loops.append((addr_to_i[tgt], i))
    best = None
    for s, e in loops:
        n = sum(1 for _, op, _ in ins[s:e + 1] if op.startswith(("v_mfma", "v_smfmac")))
        inner = not any(s <= s2 and e2 <= e and (s2, e2) != (s, e) and
                        sum(1 for _, op, _ in ins[s2:e2 + 1] if op.startswith("v_mfma")) >= max(1, n // 2) for s2, e2 in loops)
        key = (n, -((e - s)))
        if n and inner and (best is None or key > best[0]):
            best = (key, s, e)
    return None if best is None else best[1:]


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "scalar / branch / other"


def report(name, ins, out):
    loop = hot_loop(ins)
    total_mfma = sum(1 for _, op, _ in ins if op.startswith("v_mfma"))
    print(f"\n== {demangle(name)}", file=out)
    print(f"   {len(ins)} instructions, {total_mfma} MFMA in the whole kernel", file=out)
    if loop is None:
        print("   (no loop with MFMAs: a VALU kernel — the mix below is the WHOLE kernel, straight-line code and cold blocks included)", file=out)
        s, e = 0, len(ins) - 1
    else:
        s, e = loop
    body = ins[s:e + 1]
    cnt = collections.Counter(classify(op) for _, op, _ in body)
    ops = collections.Counter(op for _, op, _ in body)
    mfma = cnt["mfma"]
    valu = sum(v for k, v in cnt.items() if k in VALU)
    print(f"   {'hot loop' if loop is not None else 'kernel'}: {len(body)} instructions at +0x{body[0][0] - ins[0][0]:x} .. +0x{body[-1][0] - ins[0][0]:x}", file=out)
    for cname, _ in CLASSES:
        if cnt[cname]:
            top = ", ".join(f"{o} x{n}" for o, n in ops.most_common() if classify(o) == cname)[:150]
            print(f"   {cname:28s} {cnt[cname]:5d}   {top}", file=out)
    if mfma:
        print(f"   VALU issues per MFMA        {valu / mfma:5.2f}   (a 32x32x16 bf16 MFMA holds the matrix pipe 8 passes; ~5 fillers fit a gap)", file=out)
        print(f"   LDS reads per MFMA          {cnt['LDS read'] / mfma:5.2f}", file=out)
    else:
        print(f"   VALU issues                 {valu:5d}", file=out)
    print(f"   waits + barriers per iter   {cnt['wait'] + cnt['barrier']:5d}", file=out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("object")
    ap.add_argument("--kernel", default=None, help="substring of the (mangled or demangled) kernel name")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = open(a.out, "w") if a.out else sys.stdout
    print(f"# tools/isa_mix.py {a.object}: per-wave instruction mix of the hot loop of each kernel (static; gfx950 code object).\n"
          "# Counts cover every instruction between the loop head and its backward branch, cold out-of-line blocks included: upper bounds.", file=out)
    for name, ins in kernels(device_disassembly(a.object)).items():
        if a.kernel and a.kernel not in name and a.kernel not in demangle(name):
            continue
        if ins:
            report(name, ins, out)


if __name__ == "__main__":
    main()
