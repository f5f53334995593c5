"""Timeline summary of a rocprofv3 --kernel-trace of tools/qrbench3.py with the communication-avoiding batched QR:
busy time per kernel, and the launch timeline (start, duration, queue) of the first blocks of the last QR."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def grid(r):
    return tuple(int(r[k]) // max(1, int(r[w])) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), grid(r), r.get("Queue_Id", "?"), int(r["Workgroup_Size_X"])) for r in rows]
ks = sorted(k for k in ks if k[2].startswith("cq::"))
# the last QR: kernels after the largest gap between consecutive cq launches
gaps = [(ks[i + 1][0] - max(k[1] for k in ks[:i + 1][-8:]), i) for i in range(len(ks) - 1)]
cut = max(gaps)[1] + 1
ks = ks[cut:]
t0, t1 = ks[0][0], max(k[1] for k in ks)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print(f"{len(ks)} kernels, span {(t1 - t0) / 1e6:.2f} ms, some kernel running {union([(k[0], k[1]) for k in ks]) / 1e6:.2f} ms")
fam = {}
for s, e, n, g, q, wg in ks:
    f = fam.setdefault(n, [0, 0.0]); f[0] += 1; f[1] += (e - s) / 1e3
for n, (c, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:20s} {c:5d} launches  {us / 1e3:8.2f} ms busy  {us / c:7.1f} us each")
print("timeline of the first launches (us from the start):")
for s, e, n, g, q, wg in ks[:nshow]:
    print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  queue {q:>3s}  {n:14s} grid {str(g):16s} x {wg}")
