"""Assemble profiles/rNN_* from the raw output of one gpurun call of tools/rNN_profiles.sh (see profiles/README.md).
Run from the repository root:  python tools/profiles.py r03"""
import csv, json, shutil, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = "gpurun_out/"
R = f"profiles/{RND}_"
D = "" if RND == "r02" else RND + "_"          # round 2 wrote its raw directories without the round prefix
shutil.copy(G + f"{RND}_bench.json", R + "bench.json")
shutil.copy(G + f"{RND}_bench_under_rocprof.json", R + "bench_under_rocprof.json")
with open(G + f"{RND}_phase.txt") as f:
    lines = [l for l in f if "amdgpu.ids" not in l and not l.startswith("[bench]")]
open(R + "engine_phase_profile.txt", "w").writelines(lines)
rows = list(csv.reader(open(G + D + "profK/p_kernel_stats.csv")))
csv.writer(open(R + "bench_kernel_stats.csv", "w")).writerows([[c[:110] for c in r] for r in rows[:12]])
tr = list(csv.DictReader(open(G + D + "profK/p_kernel_trace.csv")))
eng = [r for r in tr if "eng_kernel" in r["Kernel_Name"]]
with open(R + "eng_kernel_dispatches.csv", "w") as f:
    f.write("dispatch,variant,grid_x,workgroup_x,duration_ms\n")
    for i, r in enumerate(eng):
        f.write(f"{i},{r['Kernel_Name'][:4].strip(':')},{r['Grid_Size_X']},{r['Workgroup_Size_X']},"
                f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.6f}\n")
big = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in eng if r["Kernel_Name"].startswith("v512")]
timed = big[-4:]          # the v512 cavity launches of the two timed sweeps
under = json.loads(open(G + f"{RND}_bench_under_rocprof.json").read().strip().splitlines()[-1])
print("rocprof avg of the timed v512 launches: %.3f ms; bench HIP events in the same process: %.3f ms" % (sum(timed) / len(timed), under["roofline"]["avg_launch_ms"]))
out, tot = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.reader(open(G + D + f"pmc_{c}/p_counter_collection.csv")))
    h = rows[0]
    kn, cv = h.index("Kernel_Name"), h.index("Counter_Value")
    keep = [rows[0]] + [r for r in rows[1:] if "eng_kernel" in r[kn]]
    csv.writer(open(R + f"pmc_{c}_eng_kernel_dispatches.csv", "w")).writerows(keep)
    v512 = [r for r in keep[1:] if r[kn].startswith("v512")]
    tot[c] = [float(r[cv]) for r in v512[-2:]]
    out[c + "_KiB_per_launch"] = tot[c]
fetch = sum(tot["FETCH_SIZE"]) * 1024 * 2      # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md, HBM section); verified for 8-, 16- and 32-byte-per-lane loads by tools/probes/fetch_probe.hip (profiles/r03_fetch_write_calibration.txt)
write = sum(tot["WRITE_SIZE"]) * 1024
out.update({"launches": 2, "bytes_per_launch_avg": (fetch + write) / 2, "fetch_bytes_corrected_total": fetch,
            "write_bytes_total": write,
            "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline",
            "note": "memory-side (L2 fabric) traffic of the two v512::eng_kernel launches of the timed sweep; Infinity-Cache hits are included in these counters"})
json.dump(out, open(R + "pmc_eng_kernel.json", "w"), indent=1)
print("traffic per launch (avg): %.3f TB" % (out["bytes_per_launch_avg"] / 1e12))
# MFMA counters of the same launches
rows = list(csv.DictReader(open(G + D + "pmc_MFMA/p_counter_collection.csv")))
v512 = [r for r in rows if r["Kernel_Name"].startswith("v512")]
disp = sorted(set(int(r["Dispatch_Id"]) for r in v512))[-2:]
m = {}
for r in v512:
    if int(r["Dispatch_Id"]) in disp:
        m.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {k: v for k, v in m.items()}
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
    # SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the 1024 SIMDs (= 64 per v_mfma_f64_16x16x4: it equals
    # SQ_INSTS_VALU_MFMA_F64 x 64 to the digit); SQ_BUSY_CYCLES = busy cycles summed over the 32 shader engines
    res["mfma_pipe_utilisation"] = [a / (b / 32.0 * 1024.0) for a, b in zip(m["SQ_VALU_MFMA_BUSY_CYCLES"], m["SQ_BUSY_CYCLES"])]
res["command"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 --output-format csv -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline"
res["note"] = ("the two v512::eng_kernel launches of the timed sweep; counter semantics as rocprofv3 reports them on gfx950 "
               "(summed over shader engines / XCDs); SQ_INSTS_VALU_MFMA_F64 x 2048 flop = executed MFMA flops")
if "SQ_INSTS_VALU_MFMA_F64" in m:
    res["mfma_flops_per_launch"] = [v * 2048 for v in m["SQ_INSTS_VALU_MFMA_F64"]]
json.dump(res, open(R + "pmc_mfma.json", "w"), indent=1)
print({k: v for k, v in res.items() if k not in ("command", "note")})
b = json.loads(open(R + "bench.json").read().strip().splitlines()[-1])
b["roofline"]["traffic"] = out["bytes_per_launch_avg"]
b["roofline"]["traffic_source"] = f"profiles/{RND}_pmc_eng_kernel.json (same command under rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE per launch)"
if b["roofline"].get("avg_launch_ms"):
    b["roofline"]["hbm_GBps"] = out["bytes_per_launch_avg"] / (b["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
    b["roofline"]["frac_hbm"] = b["roofline"]["hbm_GBps"] / 8000.0
json.dump(b, open(R + "bench.json", "w"), indent=1)

# counter calibration (tools/probes/fetch_probe.hip under rocprofv3 --pmc): bytes reported / bytes streamed per access shape
try:
    lines = ["# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- tools/_fetch_probe.bin   (4 GiB = 4194304 KiB streamed per kernel)\n"]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(G + f"{RND}_cal_{c}/p_counter_collection.csv")):
            if "per_lane" in r["Kernel_Name"]:
                v = float(r["Counter_Value"])
                lines.append(f"{c:10s} {r['Kernel_Name'].split('(')[0]:20s} {v:14.1f} KiB reported = {v / 4194304.0:.4f} x the bytes streamed\n")
    lines.append("# => FETCH_SIZE counts exactly half of the bytes for 8-, 16- and 32-byte-per-lane coalesced loads alike (x2 correction applies to the\n"
                 "#    engine's 8-byte and 32-byte-per-lane streams); WRITE_SIZE is exact for 8- and 32-byte-per-lane stores.\n")
    open(R + "fetch_write_calibration.txt", "w").writelines(lines)
    print("".join(lines))
except FileNotFoundError:
    pass
