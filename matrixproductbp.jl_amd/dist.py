"""Multi-GPU sharding of the sweep: one process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in CPU tests).  Nodes - and with them their out-edges - are split into contiguous,
edge-balanced blocks; every rank updates only its nodes (a Jacobi sweep reads nothing written in the same
sweep), then ONE all-gather of the fixed-stride message slots makes every rank hold all E messages again
(the only exchange step of the path: `bp.μ[idx(e)] = μj` visibility, reference
src/recursive_bp_factor.jl:177).  Slots are laid out rank-major and padded to the largest shard so that
the collective is a plain equal-size all-gather."""
from __future__ import annotations

import numpy as np


def shard_nodes(nbr_ptr, world):
    """Contiguous node blocks with (nearly) equal numbers of out-edges.  Returns list of (lo, hi)."""
    nbr_ptr = np.asarray(nbr_ptr, dtype=np.int64)
    N = nbr_ptr.size - 1
    tot = int(nbr_ptr[-1])
    bounds = [0]
    for r in range(1, world):
        target = tot * r / world
        j = int(np.searchsorted(nbr_ptr, target, side="left"))
        bounds.append(min(max(j, bounds[-1]), N))
    bounds.append(N)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def slot_map(nbr_ptr, out_edge, n_edges, world):
    """slot_of_edge[e]: rank-major, padded.  Edge e is owned by the rank that owns its source node
    (= the node that has e among its out-edges).  Returns (slot_of_edge, slots_per_rank, shards)."""
    shards = shard_nodes(nbr_ptr, world)
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
