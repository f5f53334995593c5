#!/usr/bin/env python3
"""Benchmark of the MPBP message-update hot path (BASELINE.json metric: edge-message updates/sec and
s/sweep, SIS on a 3-regular random graph, N=1024, T=50, max bond 20 = BASELINE configs[1]).

A "step" is one Jacobi sweep of onebpiter! over all nodes (E = 3072 edge-message updates), messages
resident in HBM.  Usage:  python bench.py --gpus N --steps K --warmup W
For N > 1 launch with  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one rank per GPU over RCCL; nodes are sharded, one all-gather of the message slots per sweep).
Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix (= vector) peak, public spec; BASELINE.md section 2


# ------------------------------------------------------------------------------------------------
# CPU baseline: the numpy oracle (a port of the reference algorithm: LAPACK gesdd SVD sweeps) timed on
# a bounded sample of the same workload.  Only this leg imports oracle/.
# ------------------------------------------------------------------------------------------------
def _cpu_heavy_op(args):
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    lam, rho, T, Mb, b1, b2 = args
    from oracle import mpbp as O
    from oracle.factors import SISFactor
    from oracle.tensor_trains import TensorTrain, TruncBond
    wi = [SISFactor(lam, rho)] * (T + 1)
    t0 = time.perf_counter()
    O.op_kron_compress(wi, (TensorTrain(b1), 1), (TensorTrain(b2), 1), T, TruncBond(Mb))
    return time.perf_counter() - t0


def cpu_baseline(msgs3, lam, rho, gam, T, Mb, E_per_sweep):
    """msgs3: the three incoming messages of one degree-3 node (lists of cores) from the device state.
    Sample: on every host core in parallel one heavy `op` (Kronecker 20x20 -> compress!, 100 SVDs up to
    400x1600); on one core the cheap parts (one light `op` with `init`, one message finalisation, the
    belief).  A degree-3 node update = 4 heavy + 3 light ops + 3 finalisations + 1 belief (cavity order,
    reference src/recursive_bp_factor.jl:140 with CavityTools.cavity)."""
    import multiprocessing as mp
    from oracle import mpbp as O
    from oracle.factors import SISFactor
    from oracle.tensor_trains import TensorTrain, TruncBond, compress, normalize, normalize_eachmatrix
    cores = min(len(os.sched_getaffinity(0)), 16)
    wi = [SISFactor(lam, rho)] * (T + 1)
    psi = [np.ones((2, 2))] * (T + 1)
    phi = [np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)]
    B = [O.prob_xy_apply(wi, 2, TensorTrain(m), psi, k, T) for k, m in enumerate(msgs3)]
    b1 = [c.copy() for c in B[0][0].tensors]
    b2 = [c.copy() for c in B[1][0].tensors]
    with mp.get_context("spawn").Pool(cores) as pool:
        th = pool.map(_cpu_heavy_op, [(lam, rho, T, Mb, b1, b2)] * cores)
    t_heavy = float(np.mean(th))
    tr = TruncBond(Mb)
    a = O.op_kron_compress(wi, B[0], B[1], T, tr)
    init = O.init_train(wi, 2, T)
    t0 = time.perf_counter()
    dest = O.op_kron_compress(wi, a, init, T, tr)
    t_light = time.perf_counter() - t0
    t0 = time.perf_counter()
    Bm = O.f_bp_partial_ij(dest[0], wi, phi, 2, 2, 1)
    mu = compress(O.mpem2(Bm), tr, is_orthogonal="left")
    normalize_eachmatrix(mu)
    normalize(mu)
    t_fin = time.perf_counter() - t0
    t0 = time.perf_counter()
    full = dest      # any bond-saturated ỹ-train: the belief step costs the same
    bb = O.marginalize(O.mpem2(O.f_bp_partial_i(full[0], wi, phi, 3)))
    normalize(bb)
    t_bel = time.perf_counter() - t0
    t_node = 4 * t_heavy + 3 * t_light + 3 * t_fin + t_bel
    rate = cores * 3.0 / t_node
    return {"value": rate, "unit": "edge-updates/s", "cores": cores, "kind": "port",
            "s_per_sweep": E_per_sweep / rate,
            "sample": (f"numpy oracle (LAPACK gesdd), {cores} processes x 1 heavy op (mean {t_heavy:.2f} s) + "
                       f"light op {t_light:.3f} s + finalisation {t_fin:.3f} s + belief {t_bel:.3f} s on the "
                       f"post-warm-up messages of node 0; node update = 4 heavy + 3 light + 3 fin + 1 belief "
                       f"= {t_node:.1f} s/core")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--bond", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (gloo = rehearsal on one GPU)")
    ap.add_argument("--dump-beliefs", default="", help="write rank-0 beliefs + f to this .npy file (parity checks)")
    ap.add_argument("--phase-profile", action="store_true", help="print the engine phase profile to stderr")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import networkx as nx
    import mpbp_amd as M
    from mpbp_amd import dist as D
    from mpbp_amd import flops as F

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # rehearsal: several ranks may share one GPU (gloo backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    # ---- BASELINE configs[1]: SIS, 3-regular random graph
    N, T, Mb = args.nodes, args.T, args.bond
    lam, rho, gam = 0.1, 0.05, 0.1
    G = nx.random_regular_graph(3, N, seed=0)
    g = M.IndexedBiDiGraph(nx.to_numpy_array(G, nodelist=range(N)))
    E = g.ne()
    ptr, ine, oute = g.nbr_arrays()
    slot, S, shards = D.slot_map(ptr, oute, E, world)
    slot_doubles = (T + 1) * Mb * Mb * 4
    cores_t = torch.zeros(world * S, slot_doubles, dtype=torch.float64, device=dev)
    bonds_t = torch.zeros(world * S, T + 2, dtype=torch.int32, device=dev)
    w = M.SISFactor(lam, rho)
    phi = [[np.array([1 - gam, gam]) if t == 0 else np.ones(2) for t in range(T + 1)] for _ in range(N)]
    bp = M.mpbp(g, [[w] * (T + 1)] * N, 2, T, phi=phi, max_bond=Mb, device=local, slot_of_edge=slot,
                n_slots=world * S, ext_cores=cores_t.data_ptr(), ext_bonds=bonds_t.data_ptr())
    bp._L.mpbp_set_profiling(bp._h, 2 if args.phase_profile else 1)
    lo, hi = shards[rank]
    owned = np.arange(lo, hi, dtype=np.int32)
    trunc = M.TruncBond(Mb)
    ms_orth, n_orth, ms_dev = [], [], []

    def step(record):
        M.onebpiter(bp, owned, trunc)
        if record:
            st = bp.last_stats
            ms_orth.append(st.ms_orth); n_orth.append(st.n_orth_launches); ms_dev.append(st.ms_total)
        if world > 1:
            D.allgather_slots(cores_t, bonds_t, S, rank, world)
            torch.cuda.synchronize()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    import ctypes as C
    ph = np.zeros(24)
    bp._L.mpbp_phase_profile(bp._h, ph.ctypes.data_as(C.POINTER(C.c_double)), 24, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if args.dump_beliefs:
        # beliefs / f of the nodes this rank owns; rank 0 gathers them
        b = np.array(M.beliefs(bp))[lo:hi]
        f = np.zeros(N)
        bp._L.mpbp_free_energy(bp._h, f.ctypes.data_as(C.POINTER(C.c_double)))
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, (lo, hi, b, f[lo:hi]))
        else:
            parts = [(lo, hi, b, f[lo:hi])]
        if rank == 0:
            ball = np.concatenate([p_[2] for p_ in parts]); fall = np.concatenate([p_[3] for p_ in parts])
            np.save(args.dump_beliefs, {"beliefs": ball, "f": fall}, allow_pickle=True)
    if rank == 0 and args.phase_profile:
        bp._L.mpbp_phase_profile(bp._h, ph.ctypes.data_as(C.POINTER(C.c_double)), 24, 0)
        names = ["stage", "Y1", "Y2", "qr1_panel", "qr1_trail", "Lf", "N", "Mt", "qr2_panel", "qr2_trail", "jacobi",
                 "trunc", "carry", "norm"]
        tot = ph[:len(names)].sum()
        st = bp.last_stats
        print(f"jacobi: {st.jacobi_sweeps} sweeps in {st.jacobi_calls} calls (last sweep, all engine launches)", file=sys.stderr)
        print("engine phase profile (workgroup-seconds, cavity-op launches):", file=sys.stderr)
        for n_, v in zip(names, ph):
            print(f"  {n_:10s} {v:10.3f} s  {100 * v / max(tot, 1e-30):5.1f} %", file=sys.stderr)
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = E * args.steps / dt
        # ---- roofline of the dominant kernel (eng_kernel launches of the cavity `op` levels, rank 0)
        b = bp.bonds()
        prof = [int(v) for v in b.max(axis=0)]
        fl = F.node_update_flops(prof, 3, 2, lambda l: 1 if l == 0 else 2)
        n_local = hi - lo
        launches = int(np.sum(n_orth))
        t_orth = float(np.sum(ms_orth)) * 1e-3
        ex = fl["executed_ops"] * n_local * args.steps
        refalg = fl["reference_ops"] * n_local * args.steps
        achieved = ex / t_orth / 1e12 if t_orth > 0 else 0.0
        # HBM-side bytes per launch come from the committed rocprofv3 --pmc passes of this same command
        # (bench.py cannot run the profiler on itself); only quoted for the configuration they were taken on
        traffic, traffic_src = None, None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_eng_kernel.json")
        if world == 1 and (N, T, Mb) == (1024, 50, 20) and os.path.exists(pmc):
            with open(pmc) as fh:
                traffic = json.load(fh)["bytes_per_launch_avg"]
            traffic_src = "profiles/r01_pmc_eng_kernel.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, per launch)"
        roofline = {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "eng_kernel (cavity op levels)", "launches": launches,
                    "avg_launch_ms": (t_orth / launches * 1e3) if launches else None,
                    "flops_per_launch_executed": ex / launches if launches else None,
                    "achieved_reference_algorithmic": refalg / t_orth / 1e12 if t_orth > 0 else 0.0,
                    "frac_reference_algorithmic": (refalg / t_orth / 1e12 / FP64_PEAK_TFLOPS) if t_orth > 0 else 0.0,
                    "note": "achieved / frac count the flops the engine executes (structured contractions + R-only "
                            "Householder QR + Jacobi on the small factor): the conservative, hardware-utilisation "
                            "reading; *_reference_algorithmic price the same launches with SURVEY.md 8(d)'s count for "
                            "the reference algorithm (SVD-based, 601.6 Gflop per node), which the device path "
                            "legitimately undercuts and which can therefore exceed the peak"}
        out = {"metric": "edge-message updates/sec (and s/sweep), SIS 3-regular N=1024 T=50 d=20",
               "value": value, "unit": "edge-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "s_per_sweep": ms_per_step / 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"SIS lambda=0.1 rho=0.05 gamma=0.1 on networkx.random_regular_graph(3,{N},seed=0), "
                                      f"T={T}, TruncBond({Mb}), Jacobi sweeps (BASELINE configs[1])",
                          "edges": E, "parallelism": f"nodes sharded over {world} GPU(s), 1 all-gather/sweep" if world > 1 else "1 GPU",
                          "bond_profile_max": prof[:5] + ["..."] + prof[-4:]},
               "device_ms_per_step": float(np.mean(ms_dev)), "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            e_in = [int(ine[p]) for p in range(ptr[0], ptr[1])]
            msgs = bp.get_messages(edges=e_in)
            out["cpu_baseline"] = cpu_baseline([msgs[e] for e in e_in], lam, rho, gam, T, Mb, E)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
