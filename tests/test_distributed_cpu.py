"""CPU, world_size 2, gloo: the N > 1 path of the pipeline — contiguous query sharding with no data-path
collective and the single all-gather of the fixed-size result records."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pram_amd.pipeline import gather_records, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_queries, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_queries, rank, world)
        # a rank's record depends only on its own query ids (embarrassingly parallel)
        rec = (torch.stack([torch.full((4, 6), float(i)) + torch.arange(6.0) for i in range(lo, hi)]) if hi > lo
               else torch.zeros(0, 4, 6))
        sizes = [b - a for a, b in (shard_range(n_queries, r, world) for r in range(world))]
        full = gather_records(rec, sizes if min(sizes) != max(sizes) else None)
        want = torch.stack([torch.full((4, 6), float(i)) + torch.arange(6.0) for i in range(n_queries)])
        q.put((rank, bool(torch.equal(full, want)), tuple(full.shape)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


import pytest


@pytest.mark.parametrize("n", [6, 7, 1])
def test_gather_records_world2(n):
    """even shards (3 + 3), uneven (4 + 3) and a rank without queries (1 + 0): always the n records in query order"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (n, 4, 6), (rank, ok, shape)


def test_gather_records_single_process_is_identity():
    rec = torch.randn(3, 5, 6)
    assert gather_records(rec) is rec
