"""Read-level sharding across the GPUs of a node.

Reads are independent (detect.cpp:852-907 touches only per-read state), so the path shards with NO data-path
collective: every rank runs the whole hot path on its own reads.  The only exchanges are
  * a tiny all-reduce of counters (reads ok / failed, samples), and
  * the gather of variable-size per-read outputs to the writer rank, which emits them in INPUT order so that the
    output is identical to the reference run with one thread (detect.cpp:902-906 writes in completion order).
`torch.distributed` is plumbing here (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import heapq

import numpy as np


def assign_reads(sample_counts, world):
    """Longest-processing-time-first partition of reads over `world` ranks by raw sample count.

    Deterministic: ties are broken by input ordinal.  Returns a list of index arrays (ascending ordinals) per rank.
    """
    n = np.asarray(sample_counts, dtype=np.int64)
    order = sorted(range(n.shape[0]), key=lambda i: (-int(n[i]), i))
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    out = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + int(n[i]), r))
    return [np.array(sorted(v), dtype=np.int64) for v in out]


def make_batches(sample_counts, max_samples, max_reads=4096):
    """Group the reads of one rank into batches of SIMILAR length ("bucketed by length class to limit tail
    divergence").  Most kernels of the path run one wavefront or workgroup per read, and the band fill is a serial chain as long
    as the read, so a batch takes as long as its longest read: one 200 kb read among 20 kb reads leaves the SIMDs of the other
    999 idle for 90 % of the step.  Reads are therefore sorted by sample count (descending, ties by ordinal) and cut into
    consecutive groups of at most `max_samples` samples / `max_reads` reads; inside a group the longest / shortest ratio is
    small.  Output order is restored downstream by ordinal (gather_records).  Returns a list of ascending index arrays.
    """
    n = np.asarray(sample_counts, dtype=np.int64)
    order = sorted(range(n.shape[0]), key=lambda i: (-int(n[i]), i))
    out, cur, load = [], [], 0
    for i in order:
        if cur and (load + int(n[i]) > max_samples or len(cur) >= max_reads):
            out.append(np.array(sorted(cur), dtype=np.int64)); cur, load = [], 0
        cur.append(i); load += int(n[i])
    if cur:
        out.append(np.array(sorted(cur), dtype=np.int64))
    return out


def reduce_counters(dist, values, device="cpu"):
    """SUM all-reduce of a small vector of counters (reads ok, reads failed, samples, ...)."""
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def reduce_max(dist, value, device="cpu"):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(dist, ordinals, records, dst=0, device="cpu"):
    """Gather per-read byte records to rank `dst`, returned there sorted by input ordinal (None elsewhere).

    Each rank sends ONE buffer: [n, (ordinal, length)*n, payload].  all_gather of the buffer lengths, then padded
    all_gather of the buffers (gloo and RCCL both support it; the payload of a 1000-read batch is a few MB).
    """
    import torch
    assert len(ordinals) == len(records)
    head = np.zeros(1 + 2 * len(records), dtype=np.int64)
    head[0] = len(records)
    for j, (o, rec) in enumerate(zip(ordinals, records)):
        head[1 + 2 * j] = int(o)
        head[2 + 2 * j] = len(rec)
    blob = head.tobytes() + b"".join(records)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        blobs = [blob]
        rank = 0
    else:
        world, rank = dist.get_world_size(), dist.get_rank()
        ln = torch.tensor([len(blob)], dtype=torch.int64, device=device)
        lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(lens, ln)
        mx = int(max(int(x.item()) for x in lens))
        buf = torch.zeros(mx, dtype=torch.uint8, device=device)
        buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        bufs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(bufs, buf)
        blobs = [bytes(b[:int(l.item())].cpu().numpy().tobytes()) for b, l in zip(bufs, lens)]
    if rank != dst:
        return None
    merged = []
    for bl in blobs:
        n = int(np.frombuffer(bl[:8], dtype=np.int64)[0])
        hd = np.frombuffer(bl[8:8 + 16 * n], dtype=np.int64).reshape(n, 2)
        pos = 8 + 16 * n
        for o, l in hd:
            merged.append((int(o), bl[pos:pos + int(l)]))
            pos += int(l)
    merged.sort(key=lambda t: t[0])
    return merged
