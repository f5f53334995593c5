"""Round 4: every VALU instruction of a SIMD takes its cycles from the fp64 MFMAs of the same SIMD (tools/probes/
mfma_operand_probe.hip).  Per function and per innermost LOOP of the shipped assembly that holds MFMAs: the
number of v_mfma, of other VALU instructions (v_* incl. v_accvgpr_*), of LDS and of global/scratch instructions, and the
VALU : MFMA ratio.  usage: python tools/valu_audit.py > profiles/r04_valu_audit.txt"""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spill_audit import asm, functions, loops, demangle  # noqa: E402


def mix(seg):
    ins = [l.split()[0] for l in seg if l.startswith("\t") and not l.strip().startswith((";", "."))]
    mf = sum(i.startswith("v_mfma") for i in ins)
    va = sum(i.startswith("v_") and not i.startswith("v_mfma") for i in ins)
    ds = sum(i.startswith("ds_") for i in ins)
    vm = sum(i.startswith(("global_", "scratch_", "buffer_", "flat_")) for i in ins)
    return mf, va, ds, vm


def main():
    print("# tools/valu_audit.py: VALU instructions beside the MFMAs, per innermost loop (hipcc -O3 --offload-arch=gfx950 -S of the shipped sources)")
    for src in ("mpbp_hip.hip", "v2_engine.hip"):
        print(f"\n== csrc/{src}")
        for name, body in functions(asm(src)):
            mf, va, ds, vm = mix(body)
            if mf < 16:
                continue
            print(f"{demangle(name)}: {mf} MFMA, {va} VALU ({va / mf:.2f} per MFMA), {ds} LDS, {vm} global/scratch")
            lp = loops(body)
            inner = [(a, b) for a, b in lp if not any((c > a or d < b) and c >= a and d <= b for c, d in lp if (c, d) != (a, b))]
            for a, b in inner:
                m2, v2, d2, g2 = mix(body[a:b + 1])
                if m2 >= 4:
                    print(f"     loop at +{a:5d} ({b - a + 1:4d} lines): {m2:4d} MFMA  {v2:4d} VALU ({v2 / m2:4.2f} per MFMA)  {d2:4d} LDS  {g2:3d} global/scratch")


if __name__ == "__main__":
    main()
