"""GPU parity at the read length of the headline workload (BASELINE.json configs[2]-[4]: 50 kb reads).

  * eight 50 kb reads -- forward / reverse / indels / soft clips / N in the reference / a QC failure -- through the WHOLE HIP path
    against the oracle: normaliseEvents bit-exact (events, alignment pairs, QC triple, final scaling), eventalign bit-exact
    (positions, coordinates, indices, fp32 signal features), the CNN within 1e-4 of the stock-PyTorch fp32 rendering, and the
    .detect records written by the host layer equal to the oracle's byte for byte;
  * a 1 000 x 50 kb batch (575 M samples, where the oracle would take an hour): idempotence and permutation invariance of the
    whole pipeline through dn_collect -- a read's calls are bit-identical whatever its place in the batch, its CNN pass or its
    neighbours -- plus probability rows in [0, 1] and monotone alignment paths."""
import hashlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import cnn_torch_ref
import pyoracle as po
from dnascent_amd import cnn_model, hip, host, synth

pytestmark = pytest.mark.gpu
NB = 50000
MILD = dict(sub_rate=0.002, ins_rate=0.001, del_rate=0.001)
SPECS = [
    (7001, NB, dict()),
    (7002, NB, dict(is_reverse=True)),
    (7003, NB, dict(**MILD)),
    (7004, NB, dict(is_reverse=True, **MILD)),
    (7005, NB, dict(sub_rate=0.003, ins_rate=0.002, del_rate=0.002, soft_clip_head=60, soft_clip_tail=35)),
    (7006, NB, dict(is_reverse=True, n_unknown=5, **MILD)),          # windows containing N are skipped (alignment.cpp:599-604)
    (7007, NB, dict(noise_pa=6.5)),                                   # fails the banded QC (event_handling.cpp:433-441)
    (7008, NB, dict(noise_pa=3.0, **MILD)),
]


@pytest.fixture(scope="module")
def run(model):
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in SPECS]
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    desc, blob, ref = cnn_model.default_model()
    ctx.load_cnn(desc, blob)
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx)
    ctx.run("detect")
    res = ctx.collect()

    def orc(r):                                                       # ctypes releases the GIL inside the oracle
        o = po.OracleRead(r, model)
        if o.normalise() == 0:
            o.eventalign()
        return o
    with ThreadPoolExecutor(8) as ex:
        oracles = list(ex.map(orc, reads))
    yield ctx, b, reads, res, oracles, ref
    for o in oracles:
        o.free()
    ctx.close()


def test_normalise_bit_exact_at_50kb(run):
    ctx, _, reads, res, oracles, _ = run
    s = res["summary"]
    assert (s["n_samples"] > 550000).all()
    for i, o in enumerate(oracles):
        n = o.norm
        assert s["status"][i] == n.status, (i, s["status"][i], n.status)
        assert s["n_scrappie"][i] == n.n_scrappie and s["n_events"][i] == n.n_events and s["n_bands"][i] == n.n_bands, i
        ev = o.events()
        gm, gs, gl = ctx.events(i, int(s["n_events"][i]))
        assert gm.tobytes() == ev["mean"].tobytes() and np.array_equal(gs, ev["raw_start"]) and np.array_equal(gl, ev["raw_len"]), i
        assert np.float64(s["rough_shift"][i]).tobytes() == np.float64(n.q_shift).tobytes(), i
        assert s["end_event"][i] == n.end_event and s["max_gap"][i] == n.max_gap and s["spanned"][i] == n.spanned, i
        assert np.float64(s["avg_log_emission"][i]).tobytes() == np.float64(n.avg_log_emission).tobytes(), i
        if n.status == 0:
            ge, gk = ctx.alignment(i, int(s["n_aligned"][i])); we, wk = o.alignment()
            assert np.array_equal(ge, we) and np.array_equal(gk, wk), i
            for f, v in (("shift", n.shift), ("scale", n.scale), ("events_per_base", n.events_per_base)):
                assert np.float64(s[f][i]).tobytes() == np.float64(v).tobytes(), (i, f)
    assert s["status"][6] == 1 and (np.delete(s["status"], 6) == 0).sum() >= 6     # the noisy read fails the QC, the others (nearly all) pass


def test_eventalign_bit_exact_and_cnn_within_tolerance_at_50kb(run):
    ctx, _, reads, res, oracles, ref = run
    s = res["summary"]
    worst = 0.0
    for i, o in enumerate(oracles):
        if o.norm.status != 0:
            assert s["n_positions"][i] == 0
            continue
        want = o.positions()
        n = int(s["n_positions"][i])
        assert n == want["coord"].shape[0] > 40000, i
        got = ctx.positions(i, n)
        for k in ("coord", "query_idx", "ref_idx", "indel", "n_signal", "core", "residual", "kmer"):
            assert np.array_equal(got[k], want[k]), (i, k)
        assert got["signal"].tobytes() == want["signal"].tobytes(), i
        if i in (0, 5):                                               # the PyTorch CPU rendering takes ~10 s per 50 kb read
            p = ctx.probabilities(i, n)
            w = cnn_torch_ref.run(ref, want["core"], want["residual"], want["signal"])
            worst = max(worst, float(np.abs(p - w).max()))
    assert worst < 1e-4, worst                                        # BASELINE.json north_star: 1e-4 absolute


def test_detect_records_at_50kb(run, tmp_path):
    ctx, b, reads, res, oracles, _ = run
    path = str(tmp_path / "r.detect")
    assert b.detect_write(ctx, path) == int((res["summary"]["status"] == 0).sum())
    want = b""
    s = res["summary"]
    for i, o in enumerate(oracles):
        if o.norm.status == 0:
            want += o.format_detect(ctx.probabilities(i, int(s["n_positions"][i])))
    assert open(path, "rb").read() == want


def _digests(res):
    out = []
    for r in range(res["summary"].shape[0]):
        lo, hi = int(res["call_off"][r]), int(res["call_off"][r + 1])
        h = hashlib.sha256()
        for k in ("status", "n_events", "n_aligned", "n_cleaned", "shift", "scale", "n_positions", "n_windows"):
            h.update(np.asarray(res["summary"][k][r]).tobytes())
        for k in ("ref_coord", "query_idx", "ref_idx", "p_edu", "p_brdu", "kmer"):
            h.update(np.ascontiguousarray(res[k][lo:hi]).tobytes())
        out.append(h.hexdigest())
    return out


def test_thousand_50kb_reads_idempotent_and_permutation_invariant(model):
    n = 1000
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    desc, blob, _ = cnn_model.default_model()
    ctx.load_cnn(desc, blob)
    fwd = host.ReadBatch()
    assert fwd.fill_synth(model, 900000, n, NB) == n
    assert fwd.samples() > 550e6
    fwd.upload(ctx); ctx.run("detect"); r1 = ctx.collect()
    s = r1["summary"]
    assert (s["status"] == 0).sum() >= 0.97 * n and int(r1["call_off"][-1]) > 10e6
    assert ((r1["p_edu"] >= 0) & (r1["p_edu"] <= 1) & (r1["p_brdu"] >= 0) & (r1["p_edu"] + r1["p_brdu"] <= 1.00001)).all()
    for i in range(0, n, 100):                                        # monotone alignment paths (D, U, L moves only)
        if s["status"][i] == 0:
            ae, ak = ctx.alignment(i, int(s["n_aligned"][i]))
            de, dk = np.diff(ae.astype(np.int64)), np.diff(ak.astype(np.int64))
            assert ((de == 1) & (dk == 1) | (de == 1) & (dk == 0) | (de == 0) & (dk == 1)).all(), i
    d1 = _digests(r1)
    ctx.run("detect"); r2 = ctx.collect()                            # idempotence of the resident batch
    assert _digests(r2) == d1
    # the same reads in two differently composed batches: the second half first (other neighbours, other CNN passes, other rows)
    a = host.ReadBatch(); assert a.fill_synth(model, 900000 + n // 2, n // 2, NB) == n // 2
    assert a.fill_synth(model, 900000, n // 2, NB) == n // 2
    a.upload(ctx); ctx.run("detect"); r3 = ctx.collect()
    d3 = _digests(r3)
    assert d3[:n // 2] == d1[n // 2:] and d3[n // 2:] == d1[:n // 2]
    assert ctx.cnn_range_escalations() == 0
    print("1000 x 50 kb digest", hashlib.sha256("".join(d1).encode()).hexdigest()[:16], "passing", int((s["status"] == 0).sum()),
          "calls", int(r1["call_off"][-1]))
    ctx.close()
