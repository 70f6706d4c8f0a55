"""GPU parity: eventalign (windowed Viterbi + feature fill + tensor packing) through the C-ABI against the oracle.

Bar: every index / integer output (window walk, arg-max labels hence positions, coordinates, query/reference indices,
indel scores, sample counts, core / residual indices) bit-exact; the fp32 signal features bit-exact (their arithmetic
is restated cast by cast); the fp64 window Viterbi SCORE within 1e-9 relative, because the device exp/log are the ROCm
device-library functions where the reference calls glibc (north_star: log-likelihoods within 1e-3 relative).
"""
import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import hip, host, synth

pytestmark = pytest.mark.gpu

SPECS = [
    (201, 3000, dict()),
    (202, 5000, dict(is_reverse=True)),
    (203, 5000, dict(sub_rate=0.003, ins_rate=0.001, del_rate=0.001)),
    (204, 4000, dict(is_reverse=True, sub_rate=0.003, ins_rate=0.002, del_rate=0.002, soft_clip_head=25, soft_clip_tail=40)),
    (205, 4000, dict(n_unknown=3)),            # windows with N are skipped (alignment.cpp:599-604)
    (206, 3000, dict(noise_pa=6.5)),           # fails the banded QC: eventalign must not run
    (207, 20000, dict(sub_rate=0.002)),
    (208, 2500, dict(noise_pa=3.5)),
]


@pytest.fixture(scope="module")
def run(model):
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in SPECS]
    batch = host.ReadBatch()
    for r in reads:
        assert batch.add_synth(r) >= 0
    batch.upload(ctx)
    ctx.run("normalise")
    ctx.run("eventalign")
    ctx.sync()
    summ = ctx.summaries()
    oracles = []
    for r in reads:
        o = po.OracleRead(r, model)
        if o.normalise() == 0:
            o.eventalign()
        oracles.append(o)
    yield ctx, reads, summ, oracles
    for o in oracles:
        o.free()
    ctx.close()


def test_window_walk_and_scores(run):
    ctx, reads, summ, oracles = run
    n_win = 0
    for i, o in enumerate(oracles):
        if o.norm.status != 0:
            assert summ["status"][i] != 0 and summ["n_positions"][i] == 0
            continue
        wr, wl, wt, ws = o.windows()
        assert summ["n_windows"][i] == wr.shape[0]
        gr, gl, gt, gs = ctx.windows(i, wr.shape[0])
        assert np.array_equal(gr, wr) and np.array_equal(gl, wl) and np.array_equal(gt, wt)
        assert np.allclose(gs, ws, rtol=1e-9, atol=0.0, equal_nan=True)
        n_win += wr.shape[0]
    assert n_win > 500


def test_positions_and_features_bit_exact(run):
    ctx, reads, summ, oracles = run
    checked = 0
    for i, o in enumerate(oracles):
        if o.norm.status != 0:
            continue
        p = o.positions()
        n = p["coord"].shape[0]
        assert summ["n_positions"][i] == n
        g = ctx.positions(i, n)
        for f in ("coord", "query_idx", "ref_idx", "indel", "n_signal"):
            assert np.array_equal(g[f], p[f]), (i, f)
        assert np.array_equal(g["kmer"], p["kmer"])
        assert g["core"].tobytes() == p["core"].tobytes() and g["residual"].tobytes() == p["residual"].tobytes()
        assert g["signal"].tobytes() == p["signal"].tobytes()
        checked += n
    assert checked > 30000


def test_reverse_reads_descend(run):
    ctx, reads, summ, oracles = run
    g = ctx.positions(1, int(summ["n_positions"][1]))
    assert np.all(np.diff(g["coord"].astype(np.int64)) < 0)       # creation order == sequencing direction (reads.h:321)
    g = ctx.positions(0, int(summ["n_positions"][0]))
    assert np.all(np.diff(g["coord"].astype(np.int64)) > 0)


def test_align_table_and_record_text(run, model, tmp_path):
    """`DNAscent align` (alignment.cpp:697-733, :747-898): the per-sample table eventalign prints.  Rows (coordinate, k-mer start,
    kind) are index work: exact; the scaled samples are fp64 values restated cast by cast: bit-exact.  The file written by the
    host C++ layer equals the oracle's text for the passing reads, and asking for the table does not change the positions."""
    ctx, reads, summ, oracles = run
    before = [ctx.positions(i, int(summ["n_positions"][i])) for i in range(len(reads))]
    batch = host.ReadBatch()
    for r in reads:
        assert batch.add_synth(r) >= 0
    batch.upload(ctx)
    ctx.run("normalise")
    path = str(tmp_path / "out.align")
    written = batch.align_write(ctx, path, model)
    s2 = ctx.summaries()
    assert written == int((s2["status"] == 0).sum())
    ctx.set_align_table(True); ctx.run("eventalign"); ctx.sync(); ctx.set_align_table(False)
    rows = ctx.align_rows(len(reads))
    want_text = b""
    for i, (r, o) in enumerate(zip(reads, oracles)):
        if s2["status"][i] != 0:
            assert rows[i] == 0
            continue
        w = o.align_table()
        assert int(rows[i]) == w["coord"].shape[0] > 1000, i
        g = ctx.align_table(i, int(rows[i]))
        for k in ("coord", "ref_pos", "kind"):
            assert np.array_equal(g[k], w[k]), (i, k)
        assert g["value"].tobytes() == w["value"].tobytes(), i
        assert (w["kind"] == 1).any() or i > 0                     # the set exercises insertions
        txt = o.format_align()
        assert host.format_align(r.read_id, r.contig, r.ref_start, r.ref_end, r.is_reverse, r.refseq.tobytes(), model, g) == txt
        want_text += txt
        after = ctx.positions(i, int(s2["n_positions"][i]))
        for k in ("coord", "n_signal", "core"):
            assert np.array_equal(after[k], before[i][k]), (i, k)
        assert after["signal"].tobytes() == before[i]["signal"].tobytes()
    assert open(path, "rb").read() == want_text
    # without the switch the getters refuse instead of returning stale rows
    ctx.run("eventalign"); ctx.sync()
    with pytest.raises(Exception):
        ctx.align_rows(len(reads))


def test_oversized_windows_resume_where_they_stopped(model):
    """Windows holding more observations than the fast lattice (224) are handed to the 512-observation lattice ONE AT A TIME and the
    walk continues where it stopped (k2b_launch: four passes; round 2 redid such a read from its first window).  Slow translocation
    (long dwells) makes events per base rise: a read with exactly one oversized window (passes 0, 1, 2), one with two (all four
    passes), one with many, ordinary neighbours in the same batch -- window walk, positions, features and the `align` table bit for
    bit against the oracle, which has no such limit; a read with windows beyond 512 observations is reported DN_READ_FAIL_WINDOW_EVENTS
    (the documented divergence) without disturbing the others."""
    specs = [(4243, 6000, dict(mean_dwell=16.0, sub_rate=0.002, is_reverse=True)),      # 1 window > 224
             (201, 3000, dict()),
             (4242, 6000, dict(mean_dwell=16.0, sub_rate=0.002)),                       # 2
             (4242, 6000, dict(mean_dwell=20.0, sub_rate=0.002)),                       # 16 of 114
             (4242, 6000, dict(mean_dwell=50.0, sub_rate=0.002)),                       # some beyond 512
             (202, 5000, dict(is_reverse=True))]
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in specs]
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx)
    ctx.set_align_table(True)
    ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    s = ctx.summaries()
    rows = ctx.align_rows(len(reads))
    big = []
    for i, r in enumerate(reads):
        o = po.OracleRead(r, model)
        assert o.normalise() == 0 and o.eventalign() == 0
        wr, wl, wt, ws = o.windows()
        big.append(int((wt > 224).sum()))
        if i == 4:
            assert (wt > 512).any() and (wt <= 8192).all()                               # the global-memory lattice's read
        assert s["status"][i] == 0 and s["n_windows"][i] == wr.shape[0]
        gr, gl, gt, gs = ctx.windows(i, wr.shape[0])
        assert np.array_equal(gr, wr) and np.array_equal(gl, wl) and np.array_equal(gt, wt)
        assert np.allclose(gs, ws, rtol=1e-9, atol=0.0, equal_nan=True)
        p = o.positions()
        n = p["coord"].shape[0]
        assert s["n_positions"][i] == n
        g = ctx.positions(i, n)
        for f in ("coord", "query_idx", "ref_idx", "indel", "n_signal"):
            assert np.array_equal(g[f], p[f]), (i, f)
        assert g["signal"].tobytes() == p["signal"].tobytes() and g["core"].tobytes() == p["core"].tobytes()
        t = o.align_table()
        assert rows[i] == t["coord"].shape[0]
        gtab = ctx.align_table(i, int(rows[i]))
        for f in ("coord", "ref_pos", "kind"):
            assert np.array_equal(gtab[f], t[f]), (i, f)
        assert gtab["value"].tobytes() == t["value"].tobytes()
        o.free()
    assert big[0] == 1 and big[1] == 0 and big[2] == 2 and big[3] >= 10 and big[5] == 0
    ctx.close()
