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


def gather_stats(dist, obj, device="cpu"):
    """a small JSON-able dict of every rank to every rank (run statistics: a few hundred bytes through gather_bytes' all_gather path)"""
    import json
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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
        while hi < n.shape[0] and (hi == lo or load + int(n[hi]) <= window_samples):
            load += int(n[hi]); hi += 1
        for idx in make_batches(n[lo:hi], batch_samples, batch_reads):
            batches.append(idx + lo); window_of.append(w)
        lo = hi; w += 1
    return batches, np.asarray(window_of, dtype=np.int64)


class WorkCounter:
    """One shared counter the ranks pull batch ids from: `next()` returns 0, 1, 2, ... exactly once each across all ranks.
    Backed by the process group's TCPStore (`add` is atomic on the store's server: no collective, nobody waits for anybody);
    a plain local counter for one rank.  `abort()` / `aborted()`: a rank that hit a fatal error tells the others to stop pulling."""

    def __init__(self, dist=None, name="dn_work"):
        self.local = 0
        self.store = None
        self.name = name
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed.distributed_c10d import _get_default_store
            self.store = _get_default_store()

    def next(self):
        if self.store is None:
            v = self.local; self.local += 1
            return v
        return int(self.store.add(self.name, 1)) - 1

    def abort(self):
        if self.store is not None:
            self.store.add(self.name + "_abort", 1)
        self.local_abort = True

    def aborted(self):
        if getattr(self, "local_abort", False):
            return True
        return self.store is not None and int(self.store.add(self.name + "_abort", 0)) > 0


def gather_window(dist, ordinals, records, dst=0, device="cpu", chunk_bytes=64 << 20, error=False, pending=None):
    """The records of ONE window to the writer rank: per peer a 24-byte header {n records, payload bytes, error flag} and the payload
    [(ordinal, length) * n, text] in pieces of at most `chunk_bytes` (the staging tensor on the device is bounded by the chunk, not by
    the window: round-2 advisor).  Point-to-point only: peers send, the writer receives peer by peer; no rank other than the writer
    ever holds another rank's text.  Returns (merged [(ordinal, bytes)] sorted by ordinal, any_error) on `dst`, (None, None) elsewhere.
    pending (a list): the peers' sends are posted with isend and their (work, tensor) pairs appended to it instead of being waited for
    -- a rank that is ahead of the writer then goes on pulling batches; the caller bounds how many windows it lets pile up."""
    import torch
    head = np.zeros(2 * len(records), dtype=np.int64)
    for j, (o, rec) in enumerate(zip(ordinals, records)):
        head[2 * j] = int(o); head[2 * j + 1] = len(rec)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(zip((int(o) for o in ordinals), records), key=lambda t: t[0]), bool(error)
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank != dst:
        payload = np.frombuffer(head.tobytes() + b"".join(records), dtype=np.uint8)
        hd = torch.tensor([len(records), payload.shape[0], int(bool(error))], dtype=torch.int64, device=device)
        parts = [hd] + [torch.from_numpy(payload[a:a + chunk_bytes].copy()).to(device) for a in range(0, payload.shape[0], chunk_bytes)]
        for t in parts:
            if pending is None:
                dist.send(t, dst)
            else:
                pending.append((dist.isend(t, dst), t))       # the tensor stays referenced until its send has completed
        return None, None
    merged = list(zip((int(o) for o in ordinals), records))
    any_err = bool(error)
    for r in range(world):
        if r == dst:
            continue
        hd = torch.zeros(3, dtype=torch.int64, device=device)
        dist.recv(hd, r)
        n, nbytes, err = (int(x) for x in hd.tolist())
        any_err = any_err or bool(err)
        parts = []
        for a in range(0, nbytes, chunk_bytes):
            t = torch.empty(min(chunk_bytes, nbytes - a), dtype=torch.uint8, device=device)
            dist.recv(t, r)
            parts.append(t.cpu().numpy().tobytes())
        bl = b"".join(parts)
        hdr = np.frombuffer(bl[:16 * n], dtype=np.int64).reshape(n, 2)
        pos = 16 * n
        for o, ln in hdr:
            merged.append((int(o), bl[pos:pos + int(ln)])); pos += int(ln)
    merged.sort(key=lambda t: t[0])
    return merged, any_err


class StreamDriver:
    """One rank's loop of the streamed, dynamically balanced run (the product driver, dnascent_amd/run_detect.py, and the CPU tests
    with a stand-in engine):

        pull a batch id from the shared counter -> load ITS reads (bounded: at most depth + 1 batches exist on the host) ->
        submit to the engine -> collect the oldest batch when the engine is full -> file its records under its window ->
        a window whose batches this rank has all collected, and past which the counter has moved, is GATHERED to the writer and
        written in input order -- windows in ascending order on every rank, so the point-to-point gathers pair up without a
        collective -- while the engine keeps working on the next window's batches.

    engine:  .full() .in_flight() .submit(batch_obj, tag) .collect() -> dict(tag, batch, status, record_bytes, text)
    load(ordinals) -> (batch_obj, accepted mask) or raises IOError (fatal: the run is aborted on every rank, nothing hangs)
    write(merged records of one window) is called on the writer rank only, windows ascending.
    """

    def __init__(self, dist, batches, window_of, engine, load, write, release=None, dst=0, device="cpu", chunk_bytes=64 << 20, counter=None,
                 max_pending_windows=2):
        self.dist, self.batches, self.window_of = dist, batches, window_of
        self.engine, self.load, self.write, self.release = engine, load, write, release
        self.dst, self.device, self.chunk = dst, device, chunk_bytes
        self.counter = counter or WorkCounter(dist)
        self.n_windows = int(window_of[-1]) + 1 if len(window_of) else 0
        self.pending = {}            # window -> [(ordinal, record bytes)] of this rank
        self.open = {}               # window -> batches of it this rank still has in flight
        self.flushed = 0             # windows [0, flushed) are gathered
        self.frontier = 0            # the counter has moved past every batch of windows < frontier (as seen by this rank)
        self.n_ok = self.n_fail = self.samples = 0
        self.peak_pending_bytes = 0; self.max_gather_bytes = 0
        self.batches_done = 0; self.busy_s = 0.0; self.gather_s = 0.0
        self.error = False
        self.rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
        # a peer posts its window with isend and goes on (dynamic balance would otherwise stop at every window: the writer only receives
        # window k when its own batches of k are done); at most max_pending_windows windows of text wait in a rank's send queue
        self.max_pending_windows = max_pending_windows
        self._sends = []             # per posted window: [(work, tensor), ...]

    def _collect_one(self):
        r = self.engine.collect()
        b = int(r["tag"]); w = int(self.window_of[b])
        ords = r["ordinals"] if "ordinals" in r else self._tag_ords.pop(b)
        pos = 0
        recs = self.pending.setdefault(w, [])
        for o, st, ln in zip(ords, r["status"], r["record_bytes"]):
            if int(st) == 0:
                recs.append((int(o), r["text"][pos:pos + int(ln)])); pos += int(ln); self.n_ok += 1
            else:
                self.n_fail += 1
        self.open[w] -= 1
        self.batches_done += 1
        if self.release:
            self.release(r["batch"])
        self.peak_pending_bytes = max(self.peak_pending_bytes, sum(len(x[1]) for v in self.pending.values() for x in v))

    def _flush_ready(self, final=False):
        import time
        while self.flushed < self.n_windows and (final or self.flushed < self.frontier) and self.open.get(self.flushed, 0) == 0:
            w = self.flushed
            recs = self.pending.pop(w, [])
            t0 = time.perf_counter()
            nbytes = sum(len(x[1]) for x in recs)
            while len(self._sends) >= self.max_pending_windows:          # bound the text parked in this rank's send queue
                for wk, _ in self._sends.pop(0):
                    wk.wait()
            posted = []
            merged, err = gather_window(self.dist, [x[0] for x in recs], [x[1] for x in recs], dst=self.dst, device=self.device,
                                        chunk_bytes=self.chunk, error=self.error, pending=posted if self.rank != self.dst else None)
            if posted:
                self._sends.append(posted)
            self.gather_s += time.perf_counter() - t0
            if merged is not None:
                self.max_gather_bytes = max(self.max_gather_bytes, sum(len(x[1]) for x in merged))
                self.error = self.error or bool(err)
                if not self.error:
                    self.write(merged)
            else:
                self.max_gather_bytes = max(self.max_gather_bytes, nbytes)
            self.flushed += 1

    def run(self, prefetch=True):
        """prefetch: the NEXT batch is pulled and loaded by a helper thread while this one drives the engine (the loader reads the
        container with all host cores; without it the GPU idles while a 600 MB batch comes off the disk)"""
        import time
        from concurrent.futures import ThreadPoolExecutor
        self._tag_ords = {}
        t_busy0 = time.perf_counter()
        pool = ThreadPoolExecutor(1) if prefetch else None

        def pull():
            """next batch id of this rank -> (id, future / result of its load), or None when the counter is exhausted / the run aborted"""
            if self.error or self.counter.aborted():
                self.error = True
                return None
            b = self.counter.next()
            if b >= len(self.batches):
                return None
            self.frontier = max(self.frontier, int(self.window_of[b]))      # every batch of an earlier window has been handed out
            return b, (pool.submit(self.load, self.batches[b]) if pool else None)

        nxt = pull()
        while nxt is not None:
            b, fut = nxt
            w = int(self.window_of[b])
            ords = self.batches[b]
            try:
                obj, accepted = fut.result() if fut is not None else self.load(ords)
            except IOError:
                self.error = True
                self.counter.abort()
                break
            nxt = pull()                                       # ... and its load runs while this batch is submitted / older ones collected
            keep = [int(o) for o, a in zip(ords, accepted) if a]
            self.n_fail += len(ords) - len(keep)               # rejected by the reference's own filters: failed reads, not errors
            if not keep:
                if self.release:
                    self.release(obj)
                continue
            while self.engine.full():
                self._collect_one()
                self._flush_ready()
            self._tag_ords[b] = keep
            self.open[w] = self.open.get(w, 0) + 1
            self.engine.submit(obj, b)
            self._flush_ready()
        if nxt is not None and nxt[1] is not None:             # aborted with a load in flight: let it finish, drop it
            try:
                obj, _ = nxt[1].result()
                if self.release:
                    self.release(obj)
            except IOError:
                pass
        if pool:
            pool.shutdown(wait=True)
        self.frontier = self.n_windows
        while self.engine.in_flight():
            self._collect_one()
            self._flush_ready()
        self.busy_s = time.perf_counter() - t_busy0 - self.gather_s
        self._flush_ready(final=True)                          # every rank walks ALL windows: the gathers always pair up
        for posted in self._sends:
            for wk, _ in posted:
                wk.wait()
        self._sends = []
        return not self.error
