"""Edge cases of the whole path on the GPU against the oracle: the config-5 maximum read length (200 kb) next to minimum-size
reads in one ragged batch, an empty batch, a batch in which every read fails, stages called out of order."""
import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import hip, host, synth

pytestmark = pytest.mark.gpu


def _batch(ctx, reads):
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx)
    return b


def test_ragged_batch_with_200kb_read(model):
    specs = [(501, 200, dict()), (502, 200000, dict(sub_rate=0.002)), (503, 64, dict()), (504, 1200, dict(is_reverse=True)),
             (505, 2500, dict())]
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in specs]
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    _batch(ctx, reads)
    ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    s = ctx.summaries()
    for i, r in enumerate(reads):
        o = po.OracleRead(r, model)
        st = o.normalise()
        assert (s["status"][i] != 0) == (st != 0), (i, s["status"][i], st)
        assert s["n_scrappie"][i] == o.norm.n_scrappie and s["n_events"][i] == o.norm.n_events
        assert s["n_aligned"][i] == o.norm.n_aln and s["n_cleaned"][i] == o.norm.n_cleaned
        if st == 0:
            assert s["shift"][i] == o.norm.shift and s["scale"][i] == o.norm.scale          # fp64, bit-exact
            ae, ak = ctx.alignment(i, int(s["n_aligned"][i]))
            we, wk = o.alignment()
            assert np.array_equal(ae, we) and np.array_equal(ak, wk)
            assert o.eventalign() == 0
            n = int(s["n_positions"][i])
            assert n == o.align.n_pos
            got = ctx.positions(i, n); want = o.positions()
            for k in ("coord", "query_idx", "ref_idx", "indel", "n_signal", "core", "residual", "kmer"):
                assert np.array_equal(got[k], want[k]), (i, k)
            assert got["signal"].tobytes() == want["signal"].tobytes()
        o.free()
    assert s["status"][1] == 0 and s["n_positions"][1] > 150000          # the 200 kb read went all the way through
    ctx.close()


def test_ultra_long_read_beyond_4096_detector_chunks(model):
    """A 400 kb read (4.6 M samples = 4 492 detector chunks): until round 6 a read of more than 4 096 chunks made the WHOLE batch fail with DN_ERR_OVERFLOW (k1_events cached
    the chunks' peak counts in a 4 096-entry LDS array); the reference has no such limit.  Now: normaliseEvents + eventalign bit-exact against the oracle, a short read
    beside it untouched."""
    reads = [synth.make_read(8401, 1500, model=model), synth.make_read(8400, 400000, model=model, sub_rate=0.002, ins_rate=0.001, del_rate=0.001)]
    assert (reads[1].adc.shape[0] + 1023) // 1024 > 4096
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    _batch(ctx, reads)
    ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    s = ctx.summaries()
    for i, r in enumerate(reads):
        o = po.OracleRead(r, model)
        assert o.normalise() == 0 and s["status"][i] == 0
        assert s["n_scrappie"][i] == o.norm.n_scrappie and s["n_events"][i] == o.norm.n_events and s["n_aligned"][i] == o.norm.n_aln
        assert s["shift"][i] == o.norm.shift and s["scale"][i] == o.norm.scale
        ae, ak = ctx.alignment(i, int(s["n_aligned"][i]))
        we, wk = o.alignment()
        assert np.array_equal(ae, we) and np.array_equal(ak, wk)
        assert o.eventalign() == 0 and int(s["n_positions"][i]) == o.align.n_pos
        got = ctx.positions(i, int(s["n_positions"][i])); want = o.positions()
        for k in ("coord", "query_idx", "ref_idx", "indel", "n_signal", "core", "residual", "kmer"):
            assert np.array_equal(got[k], want[k]), (i, k)
        assert got["signal"].tobytes() == want["signal"].tobytes()
        o.free()
    assert s["n_positions"][1] > 350000
    ctx.close()


def test_empty_batch_and_all_failed(model):
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    b = host.ReadBatch()
    b.upload(ctx)                                        # zero reads: every stage is a no-op, nothing is launched out of bounds
    ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    assert ctx.summaries().shape[0] == 0
    reads = [synth.make_read(600 + i, 300, model=model) for i in range(3)]          # all too short for the QC (:438)
    _batch(ctx, reads)
    ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    s = ctx.summaries()
    assert (s["status"] != 0).all() and (s["n_positions"] == 0).all()
    ctx.close()


def test_stage_order_is_enforced(model):
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    with pytest.raises(hip.DnError):
        ctx.run("normalise")                             # nothing uploaded
    _batch(ctx, [synth.make_read(700, 2500, model=model)])
    with pytest.raises(hip.DnError):
        ctx.run("eventalign")                            # normalise has not run
    with pytest.raises(hip.DnError):
        ctx.run("cnn")                                   # no model description loaded, eventalign has not run
    ctx.run("normalise"); ctx.run("eventalign")
    with pytest.raises(hip.DnError):
        ctx.run("hmm")                                   # fit models not loaded
    ctx.close()


def test_kernel_variants_behind_switches_are_bit_identical():
    """The alternative kernel forms kept for A/B measurements (offset-keyed band fill, one-read-per-wavefront scan, 128-row long-K
    convolutions, single-role / unfused separable layers) give the same summaries, alignment pairs, prefix sums and probabilities
    as the defaults, bit for bit (tools/variant_check.py: one process per switch)."""
    import os, subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "variant_check.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "variants agree: True" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_event_bound_overflow_is_reported_and_retried(model):
    """ABI 5 (dn_ctx_set_event_bound): a context whose event workspaces are sized tighter than the detector's own bound must never truncate silently --
    dn_collect reports DN_ERR_OVERFLOW for the batch -- and DNAscent::DetectStream runs that batch again at the safe bound: same records as a context that
    never had the tight bound, one retry counted, the context left at the safe bound."""
    from dnascent_amd import cnn_model
    desc, blob, _ = cnn_model.default_model()
    reads = [synth.make_read(7300 + i, 3000 + 400 * i, model=model, is_reverse=bool(i & 1)) for i in range(5)]

    def stream(bound):
        ctx = hip.Context(0)
        ctx.load_pore_model(model, 0.14); ctx.load_cnn(desc, blob)
        if bound:
            ctx.set_event_bound(bound)
        b = host.ReadBatch()
        for r in reads:
            assert b.add_synth(r) >= 0
        ds = host.DetectStream([ctx], emit=True)
        ds.submit(b, 7)
        out = ds.collect()
        st = ds.stats()
        ds.close()
        left = int(hip.lib().dn_ctx_get_event_bound(ctx.h))
        ctx.close()
        return out["text"], out["status"].tolist(), int(st.overflow_retries), left
    want, st_w, r_w, b_w = stream(0)
    assert r_w == 0 and b_w == 2 and want.count(b">") == 5
    got, st_g, r_g, b_g = stream(16)                        # samples / 16 + 64 events: these reads carry one event per ~5 samples
    assert r_g == 1 and b_g == 2 and st_g == st_w and got == want
    got4, _, r_4, b_4 = stream(4)                           # the drivers' bound: holds
    assert r_4 == 0 and b_4 == 4 and got4 == want
    # without a retrying host the overflow is an error, never a truncated result
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14); ctx.load_cnn(desc, blob); ctx.set_event_bound(16)
    _batch(ctx, reads)
    ctx.run("detect")
    with pytest.raises(hip.DnError):
        ctx.collect()
    ctx.close()


def test_cnn_lane_reserved_ahead_of_the_first_pass(model, monkeypatch):
    """ABI 6 (dn_cnn_reserve): a lane's activation buffers taken before any batch -- an error before dn_load_cnn, and the stream's records are the same with a
    lane sized ahead (smaller than, equal to and larger than what the batch needs: the pass grows what it lacks) as without the call."""
    from dnascent_amd import cnn_model
    monkeypatch.setenv("DN_CNN_ROWS", str(1 << 16))          # the cap the call rounds to (rows = 0 means a full pass)
    desc, blob, _ = cnn_model.default_model()
    reads = [synth.make_read(7400 + i, 2500 + 300 * i, model=model, is_reverse=bool(i & 1)) for i in range(4)]
    bare = hip.Context(0)
    with pytest.raises(hip.DnError):
        bare.cnn_reserve(0)
    bare.close()

    def stream(rows):
        ctx = hip.Context(0)
        ctx.load_pore_model(model, 0.14); ctx.load_cnn(desc, blob)
        if rows is not None:
            ctx.cnn_reserve(rows)
        b = host.ReadBatch()
        for r in reads:
            assert b.add_synth(r) >= 0
        ds = host.DetectStream([ctx], emit=True)
        ds.submit(b, 3)
        out = ds.collect()
        ds.close(); ctx.close()
        return out["text"]
    want = stream(None)
    assert want.count(b">") == 4
    for rows in (256, 0, 1 << 20):
        assert stream(rows) == want
