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


def test_length_bucketed_batches():
    """make_batches: every read exactly once, budgets respected, and inside a batch the reads are of similar length (the band
    fill is a serial chain as long as the read: a batch lasts as long as its longest read)."""
    rng = np.random.default_rng(7)
    n = np.clip(np.exp(rng.normal(np.log(20000), 0.9, 3000)), 1000, 200000).astype(np.int64) * 12   # config 5 length law
    budget = 230_000_000
    batches = shard.make_batches(n, budget, max_reads=1000)
    allidx = np.concatenate(batches)
    assert sorted(allidx.tolist()) == list(range(3000))
    for b in batches:
        assert len(b) <= 1000 and (n[b].sum() <= budget or len(b) == 1) and np.all(np.diff(b) > 0)
    # utilisation of the one-wavefront-per-read kernels = mean length / longest length of the batch
    util = np.array([n[b].mean() / n[b].max() for b in batches])
    naive = [np.arange(i, min(i + 1000, 3000)) for i in range(0, 3000, 1000)]
    util_naive = np.array([n[b].mean() / n[b].max() for b in naive])
    assert np.average(util, weights=[n[b].sum() for b in batches]) > 0.65 > 0.2 > util_naive.max()    # 0.71 vs 0.13-0.15
    assert shard.make_batches([], 10) == [] and [b.tolist() for b in shard.make_batches([5, 50, 5], 12)] == [[1], [0, 2]]


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
