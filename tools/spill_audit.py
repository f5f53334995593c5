"""Where the register spills of the two big kernels sit (round-3 review items 1b, 2 and weak point 4): compiles both
translation units to gfx950 assembly and counts scratch instructions per function and per LOOP (a loop = a backward branch
to a label inside the function), next to the MFMAs in the same span; for the node factorisation of the batched QR (straight-
line code between workgroup barriers) per barrier segment.  usage: python tools/spill_audit.py > profiles/r04_spill_audit.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "matrixproductbp.jl_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value", "-fPIC", "-S", "--cuda-device-only"]


def asm(src):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["hipcc"] + FLAGS + ["-o", out, os.path.join(CSRC, src)], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def demangle(n):
    s = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    s = re.sub(r"\{lambda[^}]*\}[^,>]*", "lambda", s)
    s = re.sub(r"\(.*$", "", s)
    return s[:100]


def functions(lines):
    idx = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z[A-Za-z0-9_]+):", l)] if m]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    for i, n in idx:
        e = min(x for x in ends if x > i)
        yield n, lines[i:e]


def count(seg, pat):
    return sum(pat in l for l in seg)


def loops(body):
    lab = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB[0-9_]+):", l)] if m}
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            out.append((lab[m.group(1)], i))
    return sorted(set(out))


def main():
    print("# tools/spill_audit.py: scratch (spill) instructions per function of the shipped build, hipcc -O3 --offload-arch=gfx950 -S")
    lines = asm("mpbp_hip.hip")
    print("\n== csrc/mpbp_hip.hip: the workgroup-per-problem engine (v512::eng_kernel and the functions it calls; kernel metadata:")
    meta = "\n".join(lines)
    m = re.search(r"\.name:\s+_ZN4v51210eng_kernel.*?\n(.*?)\.wavefront_size", meta, re.S)
    if m:
        for key in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count"):
            mm = re.search(rf"\.{key}:\s+(\d+)", m.group(1))
            print(f"   {key} {mm.group(1) if mm else '?'}", end="")
        print(")")
    print(f"{'function':100s} {'lines':>6s} {'scratch ld':>10s} {'st':>5s} {'MFMA':>5s}   scratch loads inside MFMA loops (innermost loops with MFMAs: loads / MFMAs)")
    seen = set()
    for n, body in functions(lines):
        if not n.startswith("_ZN4v512"):
            continue
        d = demangle(n)
        if "st_" in d or "jac_bench" in d or d[:70] in seen:
            continue                                # self tests; the other instantiations of a template repeat the same picture
        seen.add(d[:70])
        ld, st, mf = count(body, "scratch_load"), count(body, "scratch_store"), count(body, "v_mfma")
        if ld + st == 0 and mf == 0:
            continue
        lp = loops(body)
        inner = [(s, e) for (s, e) in lp if count(body[s:e + 1], "v_mfma") and not any(s2 >= s and e2 <= e and (s2, e2) != (s, e) and count(body[s2:e2 + 1], "v_mfma") for (s2, e2) in lp)]
        desc = ", ".join(f"{count(body[s:e + 1], 'scratch_load')}/{count(body[s:e + 1], 'v_mfma')}" for s, e in inner[:14])
        print(f"{d:100s} {len(body):6d} {ld:10d} {st:5d} {mf:5d}   {desc}")
    lines = asm("v2_engine.hip")
    print("\n== csrc/v2_engine.hip: node factorisation of the batched QR (cq::k_cq_fac2, one workgroup per CU; the fused update + factorisation launch k_cq_updfac) and its update kernels:")
    print("   straight-line code between workgroup barriers; a COLUMN-STEP segment = no MFMA and at least eight DPP row broadcasts")
    for n, body in functions(lines):
        d = demangle(n)
        if "k_cq_" not in d:
            continue
        segs, cur = [], []
        for l in body:
            if "s_barrier" in l:
                segs.append(cur); cur = []
            cur.append(l)
        segs.append(cur)
        steps = [s for s in segs if count(s, "row_newbcast") >= 8 and count(s, "v_mfma") == 0]
        other = [s for s in segs if not (count(s, "row_newbcast") >= 8 and count(s, "v_mfma") == 0)]
        print(f"{d:40s} scratch ld {count(body, 'scratch_load'):5d} st {count(body, 'scratch_store'):5d} | {len(steps):3d} column-step segments: ld "
              f"{sum(count(s, 'scratch_load') for s in steps):4d} st {sum(count(s, 'scratch_store') for s in steps):4d} (worst segment "
              f"{max([count(s, 'scratch_') for s in steps] or [0])}) | {len(other):3d} other segments: ld {sum(count(s, 'scratch_load') for s in other):4d} st "
              f"{sum(count(s, 'scratch_store') for s in other):4d}")


if __name__ == "__main__":
    main()
