"""Timeline summary of a rocprofv3 --kernel-trace of tools/qrbench2.py 16384x4096x1: busy time per kernel family and how much
of the panel chain (stream B) runs beside the trailing update (stream A)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ks.sort()
# last QR only: take the kernels after the largest gap
gaps = [(ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1)]
cut = max(gaps)[1] + 1
ks = ks[cut:]
t0, t1 = ks[0][0], max(k[1] for k in ks)
print(f"{len(ks)} kernels, span {(t1 - t0) / 1e6:.2f} ms")
fam = {}
for s, e, n, q in ks:
    f = fam.setdefault((n[:40], q), [0, 0.0]); f[0] += 1; f[1] += (e - s) / 1e3
for (n, q), (c, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:40s} queue {q:>4s}  {c:5d} launches  {us / 1e3:8.2f} ms busy  {us / c:7.1f} us each")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
qs = sorted(set(k[3] for k in ks))
for q in qs:
    print(f"queue {q}: busy {union([(s, e) for s, e, n, qq in ks if qq == q]) / 1e6:.2f} ms")
print(f"all queues together busy {union([(s, e) for s, e, n, q in ks]) / 1e6:.2f} ms")
