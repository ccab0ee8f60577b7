"""N > 1 path on CPU: two gloo ranks shard a frame batch, decode their shard (the
oracle stands in for the per-rank GPU decode -- this test is about the sharding
and the gather), gather the bytes, and must reproduce the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from myldpccppapi_amd import channel, codes, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, dst, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    rate, N = codes.RATE_3_4_B, 576
    K, M, z = codes.wimax_dims(rate, N)
    rows, cols = codes.wimax_edges(rate, N)
    g = oracle.Graph(rows, cols, M, N, K)

    def decode_fn(lo, hi):
        y = channel.awgn_frames(N, lo, hi - lo, 0.55, seed=99)
        return torch.from_numpy(oracle.decode(g, y, "ms")["out"].copy())

    out = sharding.decode_sharded(decode_fn, total, K, dst=dst)
    if out is not None:
        q.put((rank, out.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,dst", [(10, None), (7, 0), (1, None)])
def test_two_rank_shard_and_gather(total, dst):
    import oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, dst, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world if dst is None else 1)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rate, N = codes.RATE_3_4_B, 576
    K, M, z = codes.wimax_dims(rate, N)
    rows, cols = codes.wimax_edges(rate, N)
    g = oracle.Graph(rows, cols, M, N, K)
    want = oracle.decode(g, channel.awgn_frames(N, 0, total, 0.55, seed=99), "ms")["out"].tobytes()
    for rank, got in results:
        assert got == want


def test_shard_ranges_cover_and_order():
    for total in (0, 1, 5, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
