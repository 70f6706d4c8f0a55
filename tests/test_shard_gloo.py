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


# ---------------------------------------------------------------------------------------------------------------------------------
# dynamic balancing + streamed windowed gather (shard.plan_windows / WorkCounter / exchange_window / StreamDriver), world_size 2 and 3,
# gloo.  The engine is a stand-in with the DetectStream interface (the real one needs a GPU: tests/test_gpu_run_detect.py): a read's
# record is a pure function of its ordinal, "failed" reads have none, rejected reads never reach the engine.  Its records are finished
# text (wire flag PACK_TEXT); the packed 16-byte-per-call form is covered by test_packed_calls_format_like_text below.
# ---------------------------------------------------------------------------------------------------------------------------------
def _record(o):
    return (">read%d\n" % o).encode() + b"y" * (o % 11)


class _FakeEngine:
    """DetectStream's interface over a FIFO; every batch costs `delay(rank)` seconds so that ranks run at different speeds"""

    def __init__(self, depth, delay, fail_at=None):
        self.depth, self.delay, self.q, self.fail_at, self.n = depth, delay, [], fail_at, 0

    def full(self):
        return len(self.q) >= self.depth

    def in_flight(self):
        return len(self.q)

    def submit(self, batch, tag):
        self.n += 1
        if self.fail_at is not None and self.n == self.fail_at:
            raise RuntimeError("DN_ERR_HIP: stand-in engine failure")          # what a DnError of dn_batch_upload looks like to the driver
        self.q.append((batch, tag))

    def collect(self):
        import time
        time.sleep(self.delay)
        batch, tag = self.q.pop(0)
        status = np.array([1 if o % 13 == 5 else 0 for o in batch], np.int32)          # every 13th read fails QC
        recs = [_record(o) if s == 0 else b"" for o, s in zip(batch, status)]
        return dict(tag=tag, batch=batch, status=status, record_bytes=np.array([len(r) for r in recs], np.uint64), text=b"".join(recs))


def _sizes(n):
    rng = np.random.default_rng(11)
    return np.clip(np.exp(rng.normal(np.log(20000), 0.9, n)), 1000, 200000).astype(np.int64) * 12     # config 5 length law


def _sink(written):
    """write(text, ordinals, record_bytes) -> appends (ordinal, record) pairs"""
    def write(text, ordinals, record_bytes):
        pos = 0
        for o, ln in zip(ordinals, record_bytes):
            written.append((int(o), bytes(text[pos:pos + int(ln)]))); pos += int(ln)
        assert pos == len(text)
    return write


def test_plan_windows_properties():
    n = _sizes(2000)
    batches, window_of = shard.plan_windows(n, window_samples=8 * 30e6, batch_samples=30e6, batch_reads=500)
    assert sorted(np.concatenate(batches).tolist()) == list(range(2000))                # every read exactly once
    assert np.all(np.diff(window_of) >= 0)                                               # batch ids ascend with the window
    lo = 0
    for w in range(int(window_of[-1]) + 1):                                              # a window is a range of CONSECUTIVE reads
        idx = np.sort(np.concatenate([b for b, ww in zip(batches, window_of) if ww == w]))
        assert idx[0] == lo and np.all(np.diff(idx) == 1)
        assert n[idx].sum() <= 8 * 30e6 or len(idx) == 1
        lo = idx[-1] + 1
    assert lo == 2000
    eb, ew = shard.plan_windows([], 10, 10)
    assert len(eb) == 0 and len(ew) == 0                                                 # empty input: no batches, no windows


def test_plan_has_no_stub_batches():
    """a window of k sample budgets is cut into k batches of similar size: no 48-read remainder whose kernels cannot fill the chip (round 4: the first
    mixed-length plan had one in every window, the 10 000 x 50 kb plan 5 stubs among 24 batches)"""
    rng = np.random.default_rng(2025)
    lens = np.clip(np.exp(rng.normal(np.log(20000.0), 0.9, 36000)), 1000, 200000).astype(np.int64)
    for n, wb in ((lens * 12.5, 4.0), (np.full(10000, 575000.0), 2.0)):
        batches, window_of = shard.plan_windows(n, wb * 300e6, 300e6, 4096)
        assert sorted(np.concatenate(batches).tolist()) == list(range(len(n)))
        for w in range(int(window_of[-1])):                                          # every window but the last (which holds what is left of the input)
            sz = [float(n[b].sum()) for b, ww in zip(batches, window_of) if ww == w]
            assert len(sz) == int(wb) and max(sz) <= 300e6 and min(sz) >= 0.9 * max(sz), (w, sz)


def _scatter_worker(rank, world, port, q, path, bad_at):
    """the product driver's form (round 5): no writer rank -- every rank formats its own records and pwrite()s them into the shared file"""
    import torch.distributed as dist
    from dnascent_amd import host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = _sizes(600)
        batches, window_of = shard.plan_windows(n, window_samples=2 * world * 8e6, batch_samples=8e6, batch_reads=64)
        head = b"#header line\n"
        if rank == 0:
            with open(path, "wb") as f:
                f.write(head)
        dist.barrier()
        f = open(path, "r+b")
        mine = []

        def load(ords):
            if bad_at is not None and bad_at in ords.tolist():
                raise IOError("truncated record")
            acc = np.array([0 if o % 17 == 3 else 1 for o in ords], np.uint8)
            return [int(o) for o, a in zip(ords, acc) if a], acc

        def write_at(text, src_off, lens, file_off, ordinals):
            host.pwrite_scatter(f.fileno(), text, src_off, lens, file_off)
            mine.extend(int(o) for o in ordinals)

        eng = _FakeEngine(depth=3, delay=0.03 if rank == 0 else 0.003)
        drv = shard.StreamDriver(dist, batches, window_of, eng, load, None, dst=0, write_at=write_at, file_base=len(head), group_bytes=2000)
        dist.barrier()
        ok = drv.run()
        f.close()
        tot = shard.reduce_counters(dist, [drv.n_ok, drv.n_fail, drv.batches_done, drv.records_written])
        dist.barrier()
        import traceback
        fail = "".join(traceback.format_exception(type(drv.failure), drv.failure, drv.failure.__traceback__)) if drv.failure is not None else "None"
        if drv.failure is not None and drv.failure.__cause__ is not None:
            c_ = drv.failure.__cause__
            fail += "caused by: " + "".join(traceback.format_exception(type(c_), c_, c_.__traceback__))
        q.put((rank, ok, tot, sorted(mine), drv.max_gather_bytes, drv.format_s, drv.file_pos, drv.n_windows, fail))
    finally:
        dist.destroy_process_group()


def _run_scatter(world, path, bad_at=None):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_scatter_worker, args=(r, world, port, q, path, bad_at)) for r in range(world)]
    for p in ps:
        p.start()
    res = {}
    for _ in ps:
        r = q.get(timeout=240)
        res[r[0]] = r
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_no_writer_rank_three_ranks(tmp_path):
    """round-4 verdict item 2: every rank formats its own records and writes them in place -- the file of 3 ranks is byte-identical to what one
    rank writes (header + records in input order), only LENGTHS cross between the ranks, and every rank did format"""
    expect = b"#header line\n" + b"".join(_record(o) for o in range(600) if not (o % 17 == 3 or o % 13 == 5))
    n_expect = sum(1 for o in range(600) if not (o % 17 == 3 or o % 13 == 5))
    path = str(tmp_path / "three.detect")
    res = _run_scatter(3, path)
    assert open(path, "rb").read() == expect
    assert all(res[r][1] for r in range(3))
    assert res[0][2][0] == n_expect and res[0][2][3] == n_expect                     # every record written exactly once, by somebody
    owners = [set(res[r][3]) for r in range(3)]
    assert all(owners) and not (owners[0] & owners[1]) and not (owners[0] & owners[2]) and not (owners[1] & owners[2])     # each by the rank that computed it
    total_text = len(expect)
    for r in range(3):
        assert 0 < res[r][4] < 0.05 * total_text + 16 * 600 + 4096                   # what crossed: 16 bytes per read and a header per window, not the records
        assert res[r][5] > 0                                                          # format_s: every rank formatted
        assert res[r][6] == len(expect)                                               # every rank's running file position ends at the file's size
    # one rank, no process group: same bytes
    path1 = str(tmp_path / "one.detect")
    with open(path1, "wb") as f:
        f.write(b"#header line\n")
    f = open(path1, "r+b")
    from dnascent_amd import host
    n = _sizes(600)
    batches, window_of = shard.plan_windows(n, window_samples=2 * 20e6, batch_samples=20e6, batch_reads=64)
    drv = shard.StreamDriver(None, batches, window_of, _FakeEngine(depth=2, delay=0.0),
                             lambda ords: ([int(o) for o in ords if o % 17 != 3], np.array([o % 17 != 3 for o in ords], np.uint8)), None,
                             write_at=lambda text, so, ln, fo, o: host.pwrite_scatter(f.fileno(), text, so, ln, fo), file_base=13)
    assert drv.run() and drv.max_gather_bytes == 0
    f.close()
    assert open(path1, "rb").read() == expect


def test_no_writer_rank_abort(tmp_path):
    """a loader failure on one rank: every rank sees the error flag of that window's announcement, nobody writes from there on, nothing hangs"""
    path = str(tmp_path / "abort.detect")
    res = _run_scatter(2, path, bad_at=301)
    assert not res[0][1] and not res[1][1]
    assert res[0][2][3] < 560                                                          # not a complete file


def _stream_worker(rank, world, port, q, bad_at, depth, fail_rank):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = _sizes(600)
        batches, window_of = shard.plan_windows(n, window_samples=2 * world * 8e6, batch_samples=8e6, batch_reads=64)
        results = []
        for run in range(2 if (bad_at is None and fail_rank is None) else 1):       # two runs on ONE process group: the store keys of the first must not leak into the second
            written = []
            live = [0, 0]                                            # batches alive on the host now / at most

            def load(ords):
                if bad_at is not None and bad_at in ords.tolist():
                    raise IOError("truncated record")
                live[0] += 1; live[1] = max(live[1], live[0])
                acc = np.array([0 if o % 17 == 3 else 1 for o in ords], np.uint8)            # rejected by the reader's filters
                return [int(o) for o, a in zip(ords, acc) if a], acc

            def release(_):
                live[0] -= 1

            eng = _FakeEngine(depth=depth, delay=0.03 if rank == 0 else 0.003,               # the other ranks are 10x faster: they must take more batches
                              fail_at=5 if fail_rank == rank else None)
            drv = shard.StreamDriver(dist, batches, window_of, eng, load, _sink(written), release=release, dst=0, chunk_bytes=700)
            dist.barrier()                                           # all ranks start pulling together (process start-up skew is not what is tested)
            ok = drv.run()
            tot = shard.reduce_counters(dist, [drv.n_ok, drv.n_fail, drv.batches_done])
            results.append((rank, ok, tot, written, drv.batches_done, drv.peak_pending_bytes, drv.max_gather_bytes, live[1], len(batches), drv.n_windows,
                            drv.stats.get("recv_groups", []), repr(drv.failure)))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


def _run_stream(world, bad_at=None, depth=3, fail_rank=None):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_stream_worker, args=(r, world, port, q, bad_at, depth, fail_rank)) for r in range(world)]
    for p in ps:
        p.start()
    res = {}
    for _ in ps:
        r = q.get(timeout=240)
        res[r[0]] = r[1]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _check_complete(res, world, run):
    expect_fail = sum(1 for o in range(600) if o % 17 == 3 or o % 13 == 5)
    expect = [(o, _record(o)) for o in range(600) if not (o % 17 == 3 or o % 13 == 5)]
    r0 = res[0][run]
    assert all(res[r][run][1] for r in range(world))
    assert all(res[r][run][2] == r0[2] for r in range(world)) and r0[2][0] == len(expect) and r0[2][1] == expect_fail     # counters: ok / failed (rejected + QC)
    assert all(res[r][run][3] == [] for r in range(1, world)) and r0[3] == expect      # the writer got every record, in INPUT order, nobody else anything
    n_batches, n_windows = r0[8], r0[9]
    assert n_windows >= 3 and r0[2][2] == n_batches                         # every batch processed exactly once across the ranks
    return expect, n_windows


@pytest.mark.parametrize("depth", [1, 3])
def test_streamed_dynamic_two_ranks(depth):
    """depth 1 = `run_detect --inflight 1`: the prefetched batch's window must not be gathered without it (round-3 advisor, high)"""
    res = _run_stream(2, depth=depth)
    for run in (0, 1):                                                      # the second run on the same process group: fresh store keys
        expect, n_windows = _check_complete(res, 2, run)
        assert res[1][run][4] > res[0][run][4]                              # the faster rank pulled more batches (dynamic balance)
        total_text = sum(len(r) for _, r in expect)
        for r in (0, 1):
            assert res[r][run][7] <= depth + 3                              # at most depth + 1 + the look-ahead (2) batches alive on a host
            assert res[r][run][5] < 0.6 * total_text and res[r][run][6] < 0.75 * total_text   # buffered / gathered bytes are bounded by the window, not the run


def test_streamed_three_ranks_grouped_receive():
    """the writer takes a window from ALL its peers in one grouped call (batch_isend_irecv), never peer by peer"""
    res = _run_stream(3)
    _check_complete(res, 3, 0)
    groups = res[0][0][10]
    assert groups and max(groups) == 2                                      # some call carried a receive from both peers ...
    n_windows = res[0][0][9]
    assert sum(1 for g in groups if g == 2) >= n_windows // 2               # ... and that is the rule (a peer with nothing in a window posts nothing)
    assert res[1][0][10] == [] and res[2][0][10] == []                      # peers never receive


def test_streamed_one_rank_matches():
    """world 1 (no process group): same records, same order, windows flushed as they complete; depth 1 and 2"""
    n = _sizes(600)
    batches, window_of = shard.plan_windows(n, window_samples=2 * 20e6, batch_samples=20e6, batch_reads=64)
    for depth, prefetch in ((2, 0), (1, 2), (1, 1), (1, 0)):
        written = []
        eng = _FakeEngine(depth=depth, delay=0.0)
        drv = shard.StreamDriver(None, batches, window_of, eng, lambda ords: ([int(o) for o in ords if o % 17 != 3], np.array([o % 17 != 3 for o in ords], np.uint8)),
                                 _sink(written))
        assert drv.run(prefetch=prefetch)
        assert written == [(o, _record(o)) for o in range(600) if not (o % 17 == 3 or o % 13 == 5)]
        assert drv.records_written == drv.n_ok == len(written) and not drv.pending


def test_streamed_abort_does_not_hang():
    """a rank that cannot read a record raises the shared abort flag: every rank stops pulling, walks the remaining gathers (so the
    exchanges pair up) and reports failure; the writer writes nothing after the error"""
    res = _run_stream(2, bad_at=301)
    assert not res[0][0][1] and not res[1][0][1]
    assert len(res[0][0][3]) < 560                                          # not a complete file


def test_engine_failure_aborts_cooperatively():
    """an exception of engine.submit / collect (a DnError: HIP failure, out of memory) on ONE rank: that rank raises the abort flag and
    still walks every window, the writer does not wait for it, both report failure (round-3 advisor, medium)"""
    res = _run_stream(2, fail_rank=1)
    assert not res[0][0][1] and not res[1][0][1]
    assert "stand-in engine failure" in res[1][0][11]
    assert len(res[0][0][3]) < 560


def test_packed_calls_format_like_text():
    """packCalls -> formatPacked (what a peer sends / the writer formats) gives the bytes formatDetectRecord gives, for forward and reverse
    reads, a failed read in between, a read without calls, and a read with an IUPAC base in a 9-mer (which travels as text)"""
    import ctypes as C
    from dnascent_amd import hip, host, synth
    model = synth.pore_model()
    reads = [synth.make_read(41, 300, model=model), synth.make_read(42, 400, model=model, is_reverse=True), synth.make_read(43, 250, model=model),
             synth.make_read(44, 200, model=model), synth.make_read(45, 350, model=model, is_reverse=True)]
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    rng = np.random.default_rng(3)
    summ = np.zeros(5, hip.SUMMARY_DTYPE)
    summ["status"][2] = 3                                                   # read 2 failed QC
    ncalls = [57, 80, 0, 0, 33]
    off = np.concatenate([[0], np.cumsum(ncalls)]).astype(np.uint64)
    k = int(off[-1])
    coord = rng.integers(0, 2 ** 31, k).astype(np.uint32)
    pe = rng.random(k).astype(np.float32); pb = rng.random(k).astype(np.float32)
    pe[:4] = [0.0, 1.0, 0.9999995, 1e-7]
    pe[4:10] = [np.nan, -0.0, 1.5, 1e30, -3.25, np.inf]                      # outside [0, 1]: "%f" of any width -- the formatter sizes every record before it writes it
    pb[60:63] = [np.nan, 2.0 ** 100, -1e-9]; coord[:10] = [0, 9, 10, 99, 100, 999999, 1000000, 2 ** 31 - 1, 4294967295, 12345]
    km = rng.choice(list(b"ACGT"), (k, 9)).astype(np.uint8); km[:, 4] = ord("T")
    km[3, 0] = ord("N")                                                     # N packs
    km[int(off[4]) + 5, 7] = ord("R")                                       # an IUPAC code: read 4 must travel as text
    rb = hip.ResultBatch(5, summ.ctypes.data, off.ctypes.data, k, coord.ctypes.data, coord.ctypes.data, coord.ctypes.data, pe.ctypes.data, pb.ctypes.data,
                         km.ctypes.data)
    res = C.c_void_p(host.lib().dnh_result_new())
    try:
        n = int(host.lib().dnh_pack_calls(b.h, C.byref(rb), res))
        mp = C.c_void_p(); pp = C.c_void_p(); nb = C.c_uint64()
        assert int(host.lib().dnh_result_packed(res, C.byref(mp), C.byref(pp), C.byref(nb))) == n == 4
        meta = np.frombuffer((C.c_char * (32 * n)).from_address(mp.value), np.uint64).reshape(n, 4).copy()
        pay = np.frombuffer((C.c_char * nb.value).from_address(pp.value), np.uint8).copy()
    finally:
        host.lib().dnh_result_free(res)
    assert meta[:, 0].tolist() == [0, 1, 3, 4]
    assert [int(f) & shard.PACK_TEXT for f in meta[:, 3]] == [0, 0, 0, shard.PACK_TEXT]
    sz = shard.payload_sizes(meta[:, 1:4])
    assert int(sz.sum()) == pay.shape[0] and int(sz[0]) == int(meta[0, 2]) + 16 * 57
    ptr = np.uint64(pay.ctypes.data) + np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64)
    order = np.array([3, 0, 2, 1])                                          # any output order
    text, rbytes = host.format_packed(meta[order][:, 1:4], ptr[order])
    want = []
    for j in order:
        i = int(meta[j, 0]); sr = reads[i]; lo, hi = int(off[i]), int(off[i + 1])
        probs = np.zeros((hi - lo, 3), np.float32); probs[:, 2] = pe[lo:hi]; probs[:, 1] = pb[lo:hi]
        want.append(host.format_detect(sr.read_id, sr.contig, sr.ref_start, sr.ref_start + len(sr.refseq), sr.is_reverse, coord[lo:hi],
                                       km[lo:hi].view("S9").ravel(), probs))
    assert [int(x) for x in rbytes] == [len(w) for w in want]
    assert text == b"".join(want)
    assert 16 * k + 400 > pay.shape[0] - int(sz[3]) > 16 * (k - 33)        # 16 bytes per call + the header lines
