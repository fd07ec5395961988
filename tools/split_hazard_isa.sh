#!/bin/bash
# profiles/r04_split_hazard_isa.txt: the staging store of deform_infer_kernel<true, true> (csrc/mlp.hip) and the instructions before
# it, for the tree and the two variant sources of tools/variants/r04_split_hazard.py (CPU only: hipcc cross-compiles, llvm-objdump).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; LL=/opt/rocm/lib/llvm/bin; T="$(mktemp -d)"; trap 'rm -rf "$T"' EXIT
dump() { $LL/llvm-objcopy --dump-section .hip_fatbin=$T/fb "$1" && $LL/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb --output=$T/co --unbundle && $LL/llvm-objdump -d --no-show-raw-insn $T/co > "$2"; }
dump "$ROOT/s3gaussian_amd/lib/mlp.o" $T/tree.s
python - "$ROOT" "$T" <<'PY'
import os, runpy, subprocess, sys
root, T = sys.argv[1], sys.argv[2]
V = runpy.run_path(os.path.join(root, "tools/variants/r04_split_hazard.py"))["VARIANTS"]
src = open(os.path.join(root, "s3gaussian_amd/csrc/mlp.hip")).read()
for name, (_, edits) in V.items():
    s = src
    for old, new in edits:
        assert s.count(old) == 1
        s = s.replace(old, new)
    p = os.path.join(root, "s3gaussian_amd/csrc", f"_isa_{name}.hip")
    open(p, "w").write(s)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-munsafe-fp-atomics", "-c", p, "-o", os.path.join(T, name + ".o")])
    finally:
        os.remove(p)
PY
for v in split_nopad split_nops split_scalarized; do dump $T/$v.o $T/$v.s; done
python - "$T" > "$ROOT/profiles/r04_split_hazard_isa.txt" <<'PY'
import re, sys
T = sys.argv[1]
print("""# ISA of the split-arithmetic inference kernel deform_infer_kernel<true, true> around its staging store (VERDICT r3 Weak #6 / ADVICE r3).
# llvm-objdump (hipcc 7.2, gfx950) of four builds of csrc/mlp.hip; measured behaviour of each in profiles/r04_split_hazard.jsonl
# (tools/diag_split.py: every launch compared bit for bit with the first, 1000 launches at 1.2 M points, 400 at 70 001):
#   split_nopad  nothing between the products and the store           -> 110 872 wrong rows / 1000 launches (lanes 48..63 of a wave)
#   split_nops   16 wait states (round 3)                             -> 0
#   split_scalarized  an empty asm statement with per-component operands: products by v_mul_f32, no wait state, no copy -> 0
#   tree         4 x v_mov_b32 re-writing the stored registers on top of that, NO wait state (round 4) -> 0
# Without the pad the float4 handed to ds_write_b128 is the result of v_pk_mul_f32 (a packed, multi-pass fp32 instruction) issued a few
# slots earlier; the store reads it before the last quarter of the wave has been written when the OTHER wave of the SIMD is issuing
# v_mfma_f32_32x32x16_bf16 (never beside the fp32 MFMAs of the exact kernel, never with one wave per SIMD).  A register dependency on
# a single-pass VALU write (v_mov_b32) in front of the store is sufficient, independent of timing: the unsafe pair is
# "packed-fp32 VALU result -> DS store data" with XDL ops of another wave in flight.""")
for tag, f in (("split_nopad", "split_nopad.s"), ("split_nops", "split_nops.s"), ("split_scalarized", "split_scalarized.s"), ("tree (v_mov dependency)", "tree.s")):
    txt = open(f"{T}/{f}").read()
    name = "_ZN3s3g19deform_infer_kernelILb1ELb1EEEvNS_9InferArgsE"
    i = txt.index(name + ">:")
    lines = txt[i:txt.find("\n\n", i)].splitlines()
    st = [k for k, l in enumerate(lines) if "ds_write_b128" in l]
    print(f"\n==== {tag}: {len(lines)} instructions, {len(st)} ds_write_b128, {sum('v_pk_mul_f32' in l for l in lines)} v_pk_mul_f32, "
          f"{sum('s_nop 7' in l for l in lines)} s_nop 7")
    for k in st:              # (the first store parks the point coordinates, lanes 0..31 only; the next two the tap slots)
        win = lines[max(0, k - 9):k + 1]
        kind = ("PRODUCT store" if any(("v_pk_mul_f32" in l or "s_nop 7" in l) for l in win) or sum("v_mul_f32" in l for l in win) >= 3
                or sum("v_mov_b32" in l for l in win) >= 4 else "other store")
        if kind != "PRODUCT store":
            continue
        print(f"-- {kind} at instruction {k} and the 9 instructions before it")
        for l in win:
            print("   " + re.sub(r"\s+", " ", re.sub(r"//.*", "", l)).strip())
PY
echo written
