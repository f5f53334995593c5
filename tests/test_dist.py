"""Sharding + exchange step of the multi-GPU path on CPU: world_size-2 gloo processes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(N=20, seed=0):
    import networkx as nx
    import mpbp_amd as M
    G = nx.random_regular_graph(3, N, seed=seed)
    return M.IndexedBiDiGraph(nx.to_numpy_array(G))


def test_shards_cover_and_balance():
    from mpbp_amd import dist as D
    g = _graph(64)
    ptr, ine, oute = g.nbr_arrays()
    for world in (1, 2, 3, 8):
        slot, S, shards = D.slot_map(ptr, oute, g.E, world)
        assert shards[0][0] == 0 and shards[-1][1] == g.N
        assert all(a[1] == b[0] for a, b in zip(shards[:-1], shards[1:]))
        assert len(set(slot.tolist())) == g.E and slot.max() < world * S
        for r, (lo, hi) in enumerate(shards):
            own = {int(oute[p]) for p in range(ptr[lo], ptr[hi])}
            assert all(r * S <= slot[e] < (r + 1) * S for e in own)
        sizes = [ptr[hi] - ptr[lo] for lo, hi in shards]
        assert max(sizes) - min(sizes) <= 3 + 3


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpbp_amd import dist as D
    g = _graph(20)
    ptr, ine, oute = g.nbr_arrays()
    slot, S, shards = D.slot_map(ptr, oute, g.E, world)
    slot_doubles, nb = 12, 5
    cores = torch.zeros(world * S, slot_doubles, dtype=torch.float64)
    bonds = torch.zeros(world * S, nb, dtype=torch.int32)
    # "sweep": every rank writes its own out-messages only (value encodes the edge id)
    lo, hi = shards[rank]
    for p in range(ptr[lo], ptr[hi]):
        e = int(oute[p])
        cores[slot[e]] = float(e) + 0.5
        bonds[slot[e]] = e
    D.allgather_slots(cores, bonds, S, rank, world)
    ok = all(float(cores[slot[e], 0]) == e + 0.5 and int(bonds[slot[e], 0]) == e for e in range(g.E))
    q.put((rank, ok, float(cores.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_exchange_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]
