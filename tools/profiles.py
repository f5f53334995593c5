"""Assemble profiles/r01_* from the raw rocprofv3 / bench output of one gpurun call (see profiles/README.md)."""
import csv, json, shutil, sys
G = "gpurun_out/"
shutil.copy(G + "r01_bench.json", "profiles/r01_bench.json")
shutil.copy(G + "r01_bench_under_rocprof.json", "profiles/r01_bench_under_rocprof.json")
with open(G + "r01_phase.txt") as f:
    lines = [l for l in f if "amdgpu.ids" not in l]
open("profiles/r01_engine_phase_profile.txt", "w").writelines(lines)
rows = list(csv.reader(open(G + "profK/p_kernel_stats.csv")))
csv.writer(open("profiles/r01_bench_kernel_stats.csv", "w")).writerows([[c[:110] for c in r] for r in rows[:12]])
tr = list(csv.DictReader(open(G + "profK/p_kernel_trace.csv")))
eng = [r for r in tr if "eng_kernel" in r["Kernel_Name"]]
with open("profiles/r01_eng_kernel_dispatches.csv", "w") as f:
    f.write("dispatch,variant,grid_x,workgroup_x,duration_ms\n")
    for i, r in enumerate(eng):
        f.write(f"{i},{r['Kernel_Name'][:4].strip(':')},{r['Grid_Size_X']},{r['Workgroup_Size_X']},"
                f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.6f}\n")
# timed cavity launches under rocprof: the v512 launches of the last two sweeps
big = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in eng if r["Kernel_Name"].startswith("v512")]
timed = big[-4:]
under = json.load(open(G + "r01_bench_under_rocprof.json"))
print("rocprof avg of timed v512 launches: %.3f ms; bench HIP events: %.3f ms" % (sum(timed) / len(timed), under["roofline"]["avg_launch_ms"]))
out, tot = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.reader(open(G + f"pmc_{c}/p_counter_collection.csv")))
    keep = [rows[0]] + [r for r in rows[1:] if "eng_kernel" in r[8]]
    csv.writer(open(f"profiles/r01_pmc_{c}_eng_kernel_dispatches.csv", "w")).writerows(keep)
    h = rows[0]
    v512 = [r for r in keep[1:] if r[h.index("Kernel_Name")].startswith("v512")]
    tot[c] = [float(r[h.index("Counter_Value")]) for r in v512[-2:]]
    out[c + "_KiB_per_launch"] = tot[c]
fetch = sum(tot["FETCH_SIZE"]) * 1024 * 2      # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md, HBM section)
write = sum(tot["WRITE_SIZE"]) * 1024
out.update({"launches": 2, "bytes_per_launch_avg": (fetch + write) / 2, "fetch_bytes_corrected_total": fetch,
            "write_bytes_total": write,
            "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline",
            "note": "memory-side (L2 fabric) traffic of the two v512::eng_kernel launches of the timed sweep; Infinity-Cache hits are included in these counters"})
json.dump(out, open("profiles/r01_pmc_eng_kernel.json", "w"), indent=1)
print("traffic per launch (avg): %.3f TB" % (out["bytes_per_launch_avg"] / 1e12))
# the bench line of this same call quoted the previous PMC file: make it carry this call's measurement
b = json.load(open("profiles/r01_bench.json"))
b["roofline"]["traffic"] = out["bytes_per_launch_avg"]
json.dump(b, open("profiles/r01_bench.json", "w"), indent=1)
