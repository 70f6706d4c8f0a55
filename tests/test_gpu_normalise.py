"""GPU parity: the HIP normaliseEvents path (through the C-ABI) against the oracle, tap by tap.

Bar: bit-exact for every integer / index output (events, ranks, alignment pairs, trace-derived path, QC flags) and for
every fp64 / fp32 value whose arithmetic is restated cast by cast (prefix sums, t-stats, event means, rough and refined
scalings).  No tolerance is used anywhere in this file.
"""
import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import hip, host, synth

pytestmark = pytest.mark.gpu

SPECS = [
    # seed, bases, kwargs
    (101, 1500, dict()),
    (102, 3000, dict(is_reverse=True)),
    (103, 5000, dict(sub_rate=0.003, ins_rate=0.001, del_rate=0.001)),
    (104, 5000, dict(is_reverse=True, sub_rate=0.003, ins_rate=0.002, del_rate=0.002, soft_clip_head=25, soft_clip_tail=40)),
    (111, 4000, dict(sub_rate=0.03, ins_rate=0.02, del_rate=0.02)),   # heavy basecall errors: fails the banded QC (max_gap)
    (105, 4000, dict(n_unknown=3)),
    (106, 3000, dict(noise_pa=6.5)),            # banded QC failure (~6 pA is where reads start to fail)
    (107, 900, dict()),                         # < 1000 cleaned points: fails :438, Theil-Sen skipped
    (108, 20000, dict(sub_rate=0.002)),
    (109, 2500, dict(noise_pa=3.5, is_reverse=True)),
    (110, 12000, dict(ins_rate=0.004, del_rate=0.004)),
]


@pytest.fixture(scope="module")
def ctx(model):
    c = hip.Context(0)
    c.load_pore_model(model, 0.14)
    c.keep_k1(True)        # prefix sums / t-statistics normally never leave the segmentation kernels: ask for the taps
    yield c
    c.close()


@pytest.fixture(scope="module")
def run(ctx, model):
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in SPECS]
    batch = host.ReadBatch()
    for r in reads:
        assert batch.add_synth(r) >= 0
    batch.upload(ctx)
    ctx.run("normalise")
    ctx.sync()
    summ = ctx.summaries()
    oracles = []
    for r in reads:
        o = po.OracleRead(r, model)
        o.normalise()
        oracles.append(o)
    yield reads, summ, oracles
    for o in oracles:
        o.free()


def _beq(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_host_cigar_maps_match_oracle(run, model):
    reads, _, oracles = run
    batch = host.ReadBatch()
    for r in reads:
        batch.add_synth(r)
    for i, (r, o) in enumerate(zip(reads, oracles)):
        r2q, q2r, r2d = batch.maps(i, r.refseq.shape[0], r.basecall.shape[0])
        assert _beq(r2q, o.r2q) and _beq(q2r, o.q2r) and _beq(r2d, o.r2d)


def test_prefix_sums_and_tstats_bit_exact(ctx, run):
    reads, summ, oracles = run
    for i, o in enumerate(oracles):
        n = o.raw.shape[0]
        assert summ["n_samples"][i] == n
        s, q = ctx.prefix_sums(i, n)
        want_s = np.concatenate([[0.0], np.cumsum(o.raw)])          # numpy cumsum is the same left-to-right fp64 chain
        assert _beq(s, want_s)
        ev, t1, t2, pk = po.detect_events(o.raw, want_intermediates=True)
        g1, g2 = ctx.tstats(i, n)
        assert _beq(g1, t1) and _beq(g2, t2)
        st, ln, mn = ctx.scrappie_events(i, int(summ["n_scrappie"][i]))
        assert summ["n_scrappie"][i] == ev.shape[0]
        assert _beq(st.astype(np.uint64), ev["start"]) and _beq(ln, ev["length"]) and _beq(mn, ev["mean"])


def test_events_and_ranks_bit_exact(ctx, run):
    reads, summ, oracles = run
    for i, o in enumerate(oracles):
        oe = o.events()
        assert summ["n_events"][i] == oe.shape[0]
        mean, st, ln = ctx.events(i, oe.shape[0])
        assert _beq(mean, oe["mean"]) and _beq(st, oe["raw_start"]) and _beq(ln, oe["raw_len"])
        rq, rr = o.ranks()
        gq, gr = ctx.kmer_ranks(i, rq.shape[0], rr.shape[0])
        assert _beq(gq, rq) and _beq(gr, rr)


def test_rough_scaling_bit_exact(run):
    _, summ, oracles = run
    for i, o in enumerate(oracles):
        assert np.float64(summ["rough_shift"][i]).tobytes() == np.float64(o.norm.q_shift).tobytes()
        assert np.float64(summ["rough_scale"][i]).tobytes() == np.float64(o.norm.q_scale).tobytes()


def test_banded_alignment_bit_exact(ctx, run):
    reads, summ, oracles = run
    n_ok = 0
    for i, o in enumerate(oracles):
        n = o.norm
        assert summ["n_bands"][i] == n.n_bands
        assert summ["end_event"][i] == n.end_event
        oe, ok = o.alignment()
        assert summ["n_aligned"][i] == oe.shape[0]
        ge, gk = ctx.alignment(i, oe.shape[0])
        assert _beq(ge, oe) and _beq(gk, ok)
        assert summ["spanned"][i] == n.spanned and summ["max_gap"][i] == n.max_gap
        assert np.float64(summ["avg_log_emission"][i]).tobytes() == np.float64(n.avg_log_emission).tobytes()
        cs, cr = o.cleaned()
        assert summ["n_cleaned"][i] == cs.shape[0]
        gs, gr = ctx.cleaned(i, cs.shape[0])
        assert _beq(gs, cs) and _beq(gr, cr)
        n_ok += int(n.status == 0)
    assert n_ok >= 6


def test_status_and_final_scaling_bit_exact(run):
    _, summ, oracles = run
    statuses = set()
    for i, o in enumerate(oracles):
        n = o.norm
        assert summ["status"][i] == n.status, (i, summ["status"][i], n.status)
        statuses.add(int(n.status))
        for f, v in (("ts_slope", n.ts_slope), ("ts_intercept", n.ts_intercept), ("shift", n.shift), ("scale", n.scale),
                     ("events_per_base", n.events_per_base)):
            assert np.float64(summ[f][i]).tobytes() == np.float64(v).tobytes(), (i, f, summ[f][i], v)
    assert 0 in statuses and 1 in statuses           # the batch holds passing reads and QC failures


def test_trace_rows_consistent(ctx, run):
    """Size-independent property of the stored trace: every band's corner moves by exactly one step and the path
    re-walked on the host from the device trace reproduces the device alignment."""
    reads, summ, oracles = run
    i = 2
    nb = int(summ["n_bands"][i])
    tr, be, bk = ctx.trace(i, nb)
    assert be[0] == 49 and bk[0] == -51 and be[1] == 50 and bk[1] == -51
    de, dk = np.diff(be), np.diff(bk)
    assert np.all((de + dk) == 1) and np.all((de == 0) | (de == 1))
    ge, gk = ctx.alignment(i, int(summ["n_aligned"][i]))
    e, k = int(summ["end_event"][i]), int(summ["n_kmers_query"][i]) - 1
    path = []
    while e >= 0 and k >= 0:
        path.append((e, k))
        b = e + k + 2
        f = tr[b, be[b] - e]
        if f == 0:
            e -= 1; k -= 1
        elif f == 1:
            e -= 1
        else:
            k -= 1
    path.reverse()
    assert [p[0] for p in path] == list(ge) and [p[1] for p in path] == list(gk)
