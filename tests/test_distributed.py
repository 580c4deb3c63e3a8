"""CPU, world_size 2 over gloo: the N>1 host logic (batch sharding, result scatter, weight-blob broadcast)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _pkg import load_package  # noqa: E402  (spawned ranks re-import this module without conftest)

load_package()
from gpu_ai_inference_server_amd import sharding  # noqa: E402


def test_shard_batch_partitions():
    assert sharding.shard_batch(32, 8) == [(4 * r, 4) for r in range(8)]
    assert sharding.shard_batch(1024, 8) == [(128 * r, 128) for r in range(8)]
    assert sharding.shard_batch(5, 4) == [(0, 2), (2, 2), (4, 1), (5, 0)]
    assert sharding.shard_batch(0, 3) == [(0, 0)] * 3
    assert sharding.shard_batch(1, 1) == [(0, 1)]
    for b in range(0, 70):
        for w in range(1, 9):
            sh = sharding.shard_batch(b, w)
            assert sum(c for _, c in sh) == b and all(s2 == s1 + c1 for (s1, c1), (s2, _) in zip(sh, sh[1:]))
    with pytest.raises(ValueError):
        sharding.shard_batch(4, 0)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # load-time weight exchange: rank 0's blob replaces whatever the others hold
        ref = np.arange(100003, dtype=np.uint8) * 7
        blob = torch.from_numpy(ref.copy() if rank == 0 else np.zeros_like(ref))
        sharding.broadcast_weights(dist, blob, src=0)
        ok_blob = sharding.blob_checksum(blob.numpy()) == sharding.blob_checksum(ref)
        # data path: each rank "infers" its contiguous shard independently (no collective), results scatter by offset
        batch = 11
        shards = sharding.shard_batch(batch, world)
        x = np.arange(batch * 3, dtype=np.float32).reshape(batch, 3)
        s, c = shards[rank]
        part = x[s:s + c] * 2 + 1                                  # stand-in forward
        gathered = [None] * world
        dist.all_gather_object(gathered, part)                      # test-only gather to check the scatter on rank 0
        out = sharding.scatter_outputs(gathered, shards, batch)
        q.put((rank, ok_blob, bool(np.array_equal(out, x * 2 + 1))))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_scatter():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=90) for _ in range(2))
    [p.join(60) for p in procs]
    assert res == [(0, True, True), (1, True, True)]
    assert all(p.exitcode == 0 for p in procs)
