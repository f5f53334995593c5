"""Multi-GPU sharding of the sweep: one process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in CPU tests).  Nodes - and with them their out-edges - are split into contiguous,
edge-balanced blocks; every rank updates only its nodes (a Jacobi sweep reads nothing written in the same
sweep), then ONE all-gather of the fixed-stride message slots makes every rank hold all E messages again
(the only exchange step of the path: `bp.μ[idx(e)] = μj` visibility, reference
src/recursive_bp_factor.jl:177).  Slots are laid out rank-major and padded to the largest shard so that
the collective is a plain equal-size all-gather."""
from __future__ import annotations

import numpy as np


def node_costs(nbr_ptr, q, max_bond, T, nstates=None):
    """Predicted cost of updating each node (executed flops of its cavity products and finalisations,
    ``flops.node_update_flops`` on the saturated bond profile): grows like (3z-2) M^6-ish with the degree z and with
    nstates(l) along the cavity, so edge counts are a poor proxy on heterogeneous graphs."""
    from . import flops as F
    nbr_ptr = np.asarray(nbr_ptr, dtype=np.int64)
    deg = np.diff(nbr_ptr)
    L = T + 1
    prof = [min(max_bond, (q * q) ** min(t, L - t)) if min(t, L - t) < 32 else max_bond for t in range(L + 1)]
    ny = (lambda l: 1 if l == 0 else 2) if nstates is None else nstates
    by_deg = {}
    for z in sorted(set(int(d) for d in deg)):
        by_deg[z] = 1.0 if z == 0 else float(F.node_update_flops(prof, z, q, ny)["executed_total"])
    return np.array([by_deg[int(d)] for d in deg])


# Measured rates behind `node_times` - NOT constants of the source: they live in time_model.json next to this file and are
# regenerated from measurements by tools/fit_time_model.py (round-3 review item 7: a fit to one build drifts with every kernel
# change).  The file holds
#   rate_wg_flops     executed flop rate of the workgroup engine at configs[1]-sized batches (bench.py's roofline.achieved)
#   rate_grid_flops   executed flop rate of the batched gauge sweep when many problems share a launch (configs[2] levels)
#   level_latency_s   seconds per time step of a dependency level that holds ONE hub's problem alone: the levels of a node's
#                     3z-2 cavity products (CavityTools order) deeper than a typical node's are not filled by other nodes, and a
#                     single problem advances at the latency of its launch sequence, not at the flop rate.  Levels of different
#                     hubs of one rank run in the SAME launches, so the rank pays for its deepest node only (max, not sum).
#   config2_blocks_s  the measurements behind the last two: every node block of the shipped 8-way cut of configs[2], alone on
#                     one MI355X, seconds per saturated sweep (bench.py --config 2 --shard-of 8 --shard-index k --saturate);
#                     rate_grid_flops and level_latency_s are the least-squares fit of
#                         time = flops / rate_grid + levels(z_max) T level_latency
#                     to them (tests/test_dist.py pins the fit to the table at 4 %).
def _load_time_model():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_model.json")) as fh:
        return json.load(fh)


TIME_MODEL = _load_time_model()
RATE_WG = float(TIME_MODEL["rate_wg_flops"])
RATE_GRID = float(TIME_MODEL["rate_grid_flops"])
LEVEL_LATENCY = float(TIME_MODEL["level_latency_s"])


def config2_measured_blocks():
    """{(lo, hi): seconds per saturated sweep} of the node blocks the time model was fitted to."""
    return {tuple(int(v) for v in k.split(",")): float(s) for k, s in TIME_MODEL["config2_blocks_s"].items()}


def node_times(nbr_ptr, q, max_bond, T, nstates=None, grid=None):
    """Predicted seconds of one node update: (work, tail).
    `work` = executed flops / measured rate: it shares its launches with the other nodes of the rank, a rank's work is the SUM.
    `tail` = what the node adds if it is the DEEPEST of its rank: its dependency levels beyond those of a typical node
    (median degree) hold its problem alone (see LEVEL_LATENCY); a rank's tail is the MAX over its nodes.
    Predicted time of a rank = sum(work) + max(tail)  (`rank_times`, `shard_nodes_by_time`).
    `grid`: whether the batched gauge sweep is taken (default: the library's rule, Y_t rows > 2048)."""
    nbr_ptr = np.asarray(nbr_ptr, dtype=np.int64)
    deg = np.diff(nbr_ptr)
    ny = (lambda l: 1 if l == 0 else 2) if nstates is None else nstates
    flops = node_costs(nbr_ptr, q, max_bond, T, nstates)
    cols = min(max_bond, (q * q) ** min((T + 1) // 2, 32)) ** 2
    zmed = int(np.median(deg[deg > 0])) if (deg > 0).any() else 0
    d0 = max(3 * zmed - 2, 0)
    work = np.zeros(len(deg))
    tail = np.zeros(len(deg))
    for i, z in enumerate(deg):
        z = int(z)
        if z == 0:
            continue
        rows = cols * ny(z) * q
        g = (rows > 2048) if grid is None else grid
        work[i] = flops[i] / (RATE_GRID if g else RATE_WG)
        if g:
            tail[i] = max(3 * z - 2 - d0, 0) * T * LEVEL_LATENCY
    return work, tail


def rank_times(shards, work, tail):
    """Predicted seconds per rank: sum of the work of its nodes + the tail of its deepest node."""
    return np.array([float(np.sum(work[lo:hi])) + (float(np.max(tail[lo:hi])) if hi > lo else 0.0) for lo, hi in shards])


def shard_nodes_by_time(nbr_ptr, world, work, tail):
    """Contiguous node blocks that minimise the largest predicted rank time sum(work) + max(tail): bisection on the bound,
    greedy packing from the left for a given bound (optimal for contiguous blocks: both terms grow with the block)."""
    work = np.asarray(work, dtype=np.float64)
    tail = np.asarray(tail, dtype=np.float64)
    N = work.size

    def pack(bound):
        blocks, lo, w, t = [], 0, 0.0, 0.0
        for i in range(N):
            w2, t2 = w + work[i], max(t, tail[i])
            if i > lo and w2 + t2 > bound:
                blocks.append((lo, i))
                lo, w2, t2 = i, work[i], tail[i]
            w, t = w2, t2
        blocks.append((lo, N))
        return blocks

    lo_b = float(np.max(work + tail)) if N else 0.0
    hi_b = float(np.sum(work) + (np.max(tail) if N else 0.0))
    for _ in range(60):
        mid = 0.5 * (lo_b + hi_b)
        if len(pack(mid)) <= world:
            hi_b = mid
        else:
            lo_b = mid
    blocks = pack(hi_b)
    while len(blocks) < world:                 # fewer blocks than ranks (tiny graphs): split the largest
        k = int(np.argmax([hi - lo for lo, hi in blocks]))
        lo, hi = blocks[k]
        if hi - lo < 2:
            blocks.append((hi, hi))
            continue
        mid = (lo + hi) // 2
        blocks[k:k + 1] = [(lo, mid), (mid, hi)]
    blocks.sort()
    return blocks


def memory_plan(nbr_ptr, q, max_bond, T, shards, nstates=None, hbm_bytes=288e9, frac=0.85):
    """Bytes every rank must be able to hold at once, checked BEFORE anything is allocated (raises MemoryError):
      slab      all E message slots + bond tables (every rank keeps all messages: the all-gather restores them)
      snapshot  the in-edge slots of the rank's nodes (a split Jacobi sweep reads them from a copy, mpbp_sweep)
      node      the work trains of the rank's most expensive node (3z-2 cavity trains [b, b, nstates, q] + 2z embeddings):
                the list of a sweep is split until a pass fits, but one node is the floor
      gauge     the batched gauge sweep's buffers of that node's largest product (Y, Z, E, the packed Lf stack)
    Everything else (arena beyond one node, batch size of the gauge sweep) adapts to what is free.  Returns a list of dicts."""
    nbr_ptr = np.asarray(nbr_ptr, dtype=np.int64)
    deg = np.diff(nbr_ptr)
    E = int(nbr_ptr[-1])
    L = T + 1
    ny = (lambda l: 1 if l == 0 else 2) if nstates is None else nstates
    slot = 8 * L * max_bond * max_bond * q * q + 4 * (L + 1)
    out = []
    for r, (lo, hi) in enumerate(shards):
        zmax = int(deg[lo:hi].max()) if hi > lo else 0
        n_in = int(nbr_ptr[hi] - nbr_ptr[lo])
        b2 = max_bond * max_bond
        node = 8 * L * ((3 * zmax - 2) * b2 * ny(zmax) * q + 2 * zmax * (q * max_bond) ** 2 * q * q) if zmax else 0
        rows = b2 * ny(zmax) * q
        gauge = 8 * (2 * rows * (b2 + 32) + (T * b2 * min(rows, b2))) if zmax else 0
        need = E * slot + n_in * slot + node + gauge
        plan = {"rank": r, "slab": E * slot, "snapshot": n_in * slot, "node": node, "gauge": gauge, "total": need,
                "limit": frac * hbm_bytes}
        if need > frac * hbm_bytes:
            raise MemoryError(f"rank {r}: {need / 1e9:.1f} GB needed (slab {E * slot / 1e9:.1f}, snapshot "
                              f"{n_in * slot / 1e9:.1f}, largest node {node / 1e9:.1f}, gauge sweep {gauge / 1e9:.1f}) "
                              f"> {frac:.2f} x {hbm_bytes / 1e9:.0f} GB")
        out.append(plan)
    return out


def shard_nodes(nbr_ptr, world, cost=None):
    """Contiguous node blocks of (nearly) equal total cost.  ``cost``: per-node weights (``node_costs``); default =
    out-edge counts (exact for regular graphs).  Returns list of (lo, hi)."""
    nbr_ptr = np.asarray(nbr_ptr, dtype=np.int64)
    N = nbr_ptr.size - 1
    cum = nbr_ptr.astype(np.float64) if cost is None else np.concatenate([[0.0], np.cumsum(np.asarray(cost, dtype=np.float64))])
    tot = float(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = tot * r / world
        j = int(np.searchsorted(cum, target, side="left"))
        # the boundary that leaves the prefix closest to the target
        if j > 0 and abs(cum[j - 1] - target) < abs(cum[min(j, N)] - target):
            j -= 1
        bounds.append(min(max(j, bounds[-1]), N))
    bounds.append(N)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def slot_map(nbr_ptr, out_edge, n_edges, world, cost=None, shards=None):
    """slot_of_edge[e]: rank-major, padded.  Edge e is owned by the rank that owns its source node
    (= the node that has e among its out-edges).  `shards`: node blocks to use (e.g. `shard_nodes_by_time`), default
    `shard_nodes(nbr_ptr, world, cost)`.  Returns (slot_of_edge, slots_per_rank, shards)."""
    shards = shard_nodes(nbr_ptr, world, cost) if shards is None else shards
    nbr_ptr = np.asarray(nbr_ptr)
    out_edge = np.asarray(out_edge)
    owner_edges = []
    for (lo, hi) in shards:
        es = []
        seen = set()
        for p in range(int(nbr_ptr[lo]), int(nbr_ptr[hi])):
            e = int(out_edge[p])
            if e not in seen:
                seen.add(e)
                es.append(e)
        owner_edges.append(es)
    S = max(1, max(len(es) for es in owner_edges))
    slot = -np.ones(n_edges, dtype=np.int32)
    for r, es in enumerate(owner_edges):
        for k, e in enumerate(es):
            if slot[e] >= 0:
                raise ValueError(f"edge {e} is an out-edge of nodes on two ranks (aliased graphs do not shard)")
            slot[e] = r * S + k
    if (slot < 0).any():
        raise ValueError("some edge is nobody's out-edge")
    return slot, S, shards


def allgather_slots(cores, bonds, S, rank, world, group=None):
    """One exchange step: all-gather the rank-owned slot ranges of the two slab tensors in place.
    `cores`: [world*S, slot_doubles] float64, `bonds`: [world*S, T+2] int32 (torch tensors)."""
    import torch
    import torch.distributed as dist
    for t in (cores, bonds):
        mine = t[rank * S:(rank + 1) * S]
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(t, mine, group=group)       # in place: send = recv + rank*count
        elif t.is_cuda:
            # gloo rehearsal with device slabs (e.g. two ranks sharing one GPU): stage through the host
            host = mine.cpu()
            outs = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(outs, host, group=group)
            for r in range(world):
                if r != rank:
                    t[r * S:(r + 1) * S].copy_(outs[r])
        else:
            outs = [t[r * S:(r + 1) * S] for r in range(world)]
            dist.all_gather(outs, mine.clone(), group=group)
