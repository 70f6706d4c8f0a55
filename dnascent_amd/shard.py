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


def gather_bytes(dist, blob, dst=0, device="cpu"):
    """Variable-size byte blocks of every rank to the writer rank `dst`: a tiny all_gather of the lengths, then ONE grouped
    send / recv (batch_isend_irecv) -- every rank sends its block straight to the writer, the writer posts one receive per peer, so
    over RCCL all of its xGMI links are used at once and nobody receives what it does not need (a padded all_gather would move
    world x max bytes to every rank).  blob: uint8 numpy array.  Returns the list of per-rank uint8 arrays on `dst`, None elsewhere."""
    import torch
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [blob]
    world, rank = dist.get_world_size(), dist.get_rank()
    ln = torch.tensor([blob.shape[0]], dtype=torch.int64, device=device)
    lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lens, ln)
    lens = [int(x.item()) for x in lens]
    ops, bufs = [], {}
    if rank == dst:
        for r in range(world):
            if r != dst and lens[r]:
                bufs[r] = torch.empty(lens[r], dtype=torch.uint8, device=device)
                ops.append(dist.P2POp(dist.irecv, bufs[r], r))
    elif blob.shape[0]:
        mine = torch.from_numpy(blob).to(device)
        ops.append(dist.P2POp(dist.isend, mine, dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    return [blob if r == dst else (bufs[r].cpu().numpy() if r in bufs else np.zeros(0, np.uint8)) for r in range(world)]


def gather_records(dist, ordinals, records, dst=0, device="cpu"):
    """Gather per-read byte records to rank `dst`, returned there sorted by input ordinal (None elsewhere).
    Each rank sends ONE block: [n, (ordinal, length) * n, payload] (gather_bytes)."""
    assert len(ordinals) == len(records)
    head = np.zeros(1 + 2 * len(records), dtype=np.int64)
    head[0] = len(records)
    for j, (o, rec) in enumerate(zip(ordinals, records)):
        head[1 + 2 * j] = int(o)
        head[2 + 2 * j] = len(rec)
    blob = np.frombuffer(head.tobytes() + b"".join(records), dtype=np.uint8)
    blobs = gather_bytes(dist, blob, dst=dst, device=device)
    if blobs is None:
        return None
    merged = []
    for b in blobs:
        bl = b.tobytes()
        n = int(np.frombuffer(bl[:8], dtype=np.int64)[0])
        hd = np.frombuffer(bl[8:8 + 16 * n], dtype=np.int64).reshape(n, 2)
        pos = 8 + 16 * n
        for o, l in hd:
            merged.append((int(o), bl[pos:pos + int(l)]))
            pos += int(l)
    merged.sort(key=lambda t: t[0])
    return merged


def gather_calls(dist, read_calls, coord, p_edu, p_brdu, dst=0, device="cpu"):
    """The binary per-call results of a rank's reads -- calls per read (uint64), and per call the reference coordinate (uint32),
    P(EdU), P(BrdU) (float32): 12 bytes per call, SURVEY s8e -- to the writer rank.  Returns there a list of per-rank tuples
    (read_calls, coord, p_edu, p_brdu), None elsewhere."""
    rc = np.ascontiguousarray(read_calls, np.uint64); c = np.ascontiguousarray(coord, np.uint32)
    e = np.ascontiguousarray(p_edu, np.float32); b = np.ascontiguousarray(p_brdu, np.float32)
    head = np.array([rc.shape[0], c.shape[0]], dtype=np.uint64)
    blob = np.concatenate([x.view(np.uint8) for x in (head, rc, c, e, b)])
    blobs = gather_bytes(dist, blob, dst=dst, device=device)
    if blobs is None:
        return None
    out = []
    for bl in blobs:
        nr, nc = (int(x) for x in bl[:16].view(np.uint64))
        o = 16
        r_ = bl[o:o + 8 * nr].view(np.uint64); o += 8 * nr
        c_ = bl[o:o + 4 * nc].view(np.uint32); o += 4 * nc
        e_ = bl[o:o + 4 * nc].view(np.float32); o += 4 * nc
        b_ = bl[o:o + 4 * nc].view(np.float32)
        out.append((r_, c_, e_, b_))
    return out
