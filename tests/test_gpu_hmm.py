"""`detect --HMM` parity: llAcrossRead + sequenceProbability (detect.cpp:235-574) through the C-ABI against the oracle.

Bar: the set of calls, their order, coordinates and event counts are index work -> exact; forward log-likelihoods and
the log-likelihood ratio within 1e-3 relative (north_star), observed far tighter (device exp/log vs glibc)."""
import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import hip, host, synth

pytestmark = pytest.mark.gpu

SPECS = [
    (401, 3000, dict()),
    (402, 3000, dict(is_reverse=True, sub_rate=0.003, ins_rate=0.001, del_rate=0.002)),
    (403, 3000, dict(sub_rate=0.002, ins_rate=0.001, del_rate=0.001, soft_clip_head=20, soft_clip_tail=15)),
    (404, 3000, dict(n_unknown=4)),                 # windows containing N are skipped (:442)
    (405, 3000, dict(noise_pa=6.5)),                # fails normalise: no calls
    (406, 2600, dict(is_reverse=True)),
]


def test_hmm_calls_match_oracle(model):
    fit = synth.fit_models()
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    ctx.load_fit_models(*fit)
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in SPECS]
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx)
    ctx.run("normalise"); ctx.run("hmm"); ctx.sync()
    s = ctx.summaries()
    total = 0; worst = 0.0
    for i, r in enumerate(reads):
        o = po.OracleRead(r, model)
        if o.normalise() != 0:
            assert s["status"][i] != 0 and s["n_hmm_calls"][i] == 0
            o.free(); continue
        want = o.hmm(fit)
        n = int(s["n_hmm_calls"][i])
        assert n == want["llr"].shape[0] and n > 300
        got = ctx.hmm_calls(i, n)
        for k in ("pos_on_ref", "pos_on_query", "global_pos", "n_events"):
            assert np.array_equal(got[k], want[k]), k
        for k in ("log_analogue", "log_thymidine"):
            assert np.all(np.isnan(got[k]) == np.isnan(want[k]))
            f = ~np.isnan(want[k])
            rel = np.abs(got[k][f] - want[k][f]) / np.abs(want[k][f])
            worst = max(worst, float(rel.max()))
        f = ~np.isnan(want["llr"])
        assert np.allclose(got["llr"][f], want["llr"][f], rtol=1e-3, atol=1e-6)
        total += n
        o.free()
    assert worst < 1e-9, worst          # bar 1e-3; the device path is ~1e-13 away
    assert total > 2000
    # eventalign + CNN inputs are still available after the HMM pass (same normalised batch)
    ctx.run("eventalign")
    assert (ctx.summaries()["n_positions"][[0, 1]] > 1000).all()


def test_hmm_detect_file_matches_oracle(model, tmp_path):
    """The `--HMM` .detect file of a buffer of reads (host C++ layer: llAcrossRead + writer) vs the oracle's records.
    Lines are compared field by field: position and both 9-mers exact, the "%f" log-likelihood ratio within 2e-6
    (printed with 6 decimals from values that agree to ~1e-13)."""
    fit = synth.fit_models()
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14); ctx.load_fit_models(*fit)
    specs = [SPECS[0], SPECS[1], SPECS[4]]
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in specs]
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx); ctx.run("normalise")
    path = str(tmp_path / "hmm.detect")
    assert b.hmm_write(ctx, path, "#Mode HMM\n") == 2
    got = open(path).read().splitlines()
    want = ["#Mode HMM"]
    for r in reads:
        o = po.OracleRead(r, model)
        if o.normalise() == 0:
            o.hmm(fit)
            want += o.format_hmm().decode().splitlines()
        o.free()
    assert len(got) == len(want) and len(got) > 1000
    for g, w in zip(got, want):
        if g.startswith(">") or g.startswith("#"):
            assert g == w
            continue
        gf, wf = g.split("\t"), w.split("\t")
        assert gf[0] == wf[0] and gf[2:] == wf[2:]
        assert abs(float(gf[1]) - float(wf[1])) <= 2e-6
