"""Multi-rank path on CPU: world_size-2 `gloo` run of the sharding plumbing (partition, counter all-reduce, ordered
gather of variable-size per-read records).  The GPU box runs the same code over RCCL."""
import os
import socket

import numpy as np
import pytest

from dnascent_amd import shard


def test_partition_properties():
    rng = np.random.default_rng(5)
    n = np.clip(np.exp(rng.normal(np.log(20000), 0.9, 500)), 1000, 200000).astype(np.int64) * 12   # config 5 length law
    for world in (1, 2, 4, 8):
        parts = shard.assign_reads(n, world)
        allidx = np.concatenate(parts)
        assert sorted(allidx.tolist()) == list(range(500))              # disjoint and complete
        loads = np.array([n[p].sum() for p in parts], dtype=np.float64)
        assert loads.max() / loads.mean() < 1.05                        # LPT keeps the ranks balanced
        assert all(np.all(np.diff(p) > 0) for p in parts)
    assert [p.tolist() for p in shard.assign_reads([5, 5, 5, 5], 2)] == [[0, 2], [1, 3]]   # deterministic tie-break
    assert [p.tolist() for p in shard.assign_reads([], 3)] == [[], [], []]                 # empty batch


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = np.arange(1, 41, dtype=np.int64) * 1000
        mine = shard.assign_reads(n, world)[rank]
        tot = shard.reduce_counters(dist, [len(mine), int(n[mine].sum()), rank])
        mx = shard.reduce_max(dist, float(rank + 1))
        recs = [(">read%d\n" % i).encode() + b"x" * int(i % 7) for i in mine]          # ragged, some nearly empty
        merged = shard.gather_records(dist, mine.tolist(), recs, dst=0)
        q.put((rank, tot, mx, None if merged is None else [(o, bytes(b)) for o, b in merged]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    for _ in ps:
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r][1] == [40.0, float(sum(range(1, 41)) * 1000), 1.0]     # counters agree on every rank
        assert res[r][2] == 2.0
    assert res[1][3] is None
    merged = res[0][3]
    assert [o for o, _ in merged] == list(range(40))                          # input order restored on the writer rank
    assert all(b == (">read%d\n" % o).encode() + b"x" * (o % 7) for o, b in merged)
