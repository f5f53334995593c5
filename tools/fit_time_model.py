"""Regenerates matrixproductbp.jl_amd/time_model.json - the rates behind dist.node_times - from measurements:

  * the configs[2] shard runs of the CURRENT build: one JSON line per node block, as printed by
        python bench.py --config 2 --shard-of 8 --shard-index k --saturate --steps 1 --warmup 0 --no-cpu-baseline
    (gpurun_out/c2_shard*.json or any files given on the command line); time = flops / rate_grid + levels T level_latency is
    fitted to them by least squares;
  * optionally a configs[1] bench line (--bench profiles/rNN_bench.json): rate_wg = its roofline.achieved.

usage: python tools/fit_time_model.py [--bench FILE] shard0.json shard1.json ...      (run from the repository root)"""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = sys.argv[1:]
    bench = None
    if "--bench" in args:
        i = args.index("--bench")
        bench = args[i + 1]
        del args[i:i + 2]
    import networkx as nx
    import mpbp_amd as M
    from mpbp_amd import dist as D
    N, T, Mb = 2048, 100, 30
    g = M.IndexedBiDiGraph(nx.to_numpy_array(nx.gnp_random_graph(N, 4 / (N - 1), seed=0), nodelist=range(N)))
    ptr, _, _ = g.nbr_arrays()
    ny = lambda l: l + 1          # noqa: E731
    flops = D.node_costs(ptr, 2, Mb, T, nstates=ny)
    deg = np.diff(ptr)
    zmed = int(np.median(deg[deg > 0]))
    blocks = {}
    for f in args:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        m = re.search(r"node block \[(\d+),(\d+)\)", d["config"]["parallelism"])
        blocks[(int(m.group(1)), int(m.group(2)))] = d["ms_per_step"] / 1e3
    if len(blocks) < 3:
        raise SystemExit("need the shard runs of at least three node blocks")
    keys = sorted(blocks)
    A = np.array([[flops[lo:hi].sum(), max(3 * int(deg[lo:hi].max()) - 2 - max(3 * zmed - 2, 0), 0) * T] for lo, hi in keys])
    t = np.array([blocks[k] for k in keys])
    x, *_ = np.linalg.lstsq(A, t, rcond=None)
    rate_grid, lat = 1.0 / x[0], x[1]
    pred = A @ x
    path = os.path.join(ROOT, "matrixproductbp.jl_amd", "time_model.json")
    cur = json.load(open(path))
    out = {"rate_wg_flops": cur["rate_wg_flops"], "rate_grid_flops": float(rate_grid), "level_latency_s": float(lat),
           "source": "tools/fit_time_model.py " + " ".join(sys.argv[1:]),
           "config2_blocks_s": {f"{lo},{hi}": float(blocks[(lo, hi)]) for lo, hi in keys},
           "fit_residuals_rel": [float(v) for v in (pred - t) / t]}
    if bench:
        b = json.loads(open(bench).read().strip().splitlines()[-1]) if not open(bench).read().lstrip().startswith("{\n") else json.load(open(bench))
        out["rate_wg_flops"] = float(b["roofline"]["achieved"]) * 1e12
    json.dump(out, open(path, "w"), indent=1)
    print(f"rate_grid {rate_grid / 1e12:.2f} TFLOP/s, level latency {lat * 1e3:.1f} ms, rate_wg {out['rate_wg_flops'] / 1e12:.2f} TFLOP/s; "
          f"residuals {np.abs((pred - t) / t).max() * 100:.1f} % max; slowest / mean block {t.max() / t.mean():.3f}")


if __name__ == "__main__":
    main()
