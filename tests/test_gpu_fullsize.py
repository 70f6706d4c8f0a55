"""Parity at the FULL size of the headline workload (BASELINE.json configs[1]: 1 000 reads x 20 kb, 230 M samples), where
running the oracle over everything would take minutes.  Size-independent properties of the path instead:

  * idempotence      the same resident batch run twice gives bit-identical per-read results;
  * batch independence / permutation invariance: a read's result does not depend on its neighbours -- the batch uploaded in
                     reversed order gives the same per-read results (checksum over every per-read field, matched by read);
  * monotonicity     the rough alignment of every passing read is a monotone path (event and k-mer indices never decrease,
                     consecutive pairs differ by one of the three moves), event spans ascend without overlap;
  * spot parity      a sample of the reads -- both strands, first / middle / last of the batch -- against the oracle, bit-exact;
  * eventalign (incl. the `align` table), --HMM and the CNN at full size (20 M positions, several CNN passes): every probability row sums to 1, and a
                     read's tensors, HMM calls and probabilities are bit-identical whatever its place in the batch (other
                     pass, other neighbours, other rows).
"""
import hashlib

import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import cnn_model, hip, host, synth

pytestmark = pytest.mark.gpu
N_READS, N_BASES = 1000, 20000
FIELDS = ("status", "n_scrappie", "n_events", "n_aligned", "n_cleaned", "max_gap", "spanned", "avg_log_emission", "shift", "scale")


def _reads(model):
    return [synth.make_read(100000 + i, N_BASES, model=model, is_reverse=bool(i & 1), sub_rate=0.002, ins_rate=0.001, del_rate=0.001)
            for i in range(N_READS)]


def _run(ctx, reads, order):
    b = host.ReadBatch()
    for i in order:
        assert b.add_synth(reads[i]) >= 0
    b.upload(ctx)
    ctx.run("normalise"); ctx.sync()
    return b


def _per_read_digest(s, i):
    h = hashlib.sha256()
    for k in FIELDS:
        h.update(np.asarray(s[k][i]).tobytes())
    return h.hexdigest()


def _aux_digest(ctx, s, i):
    """eventalign tensors and --HMM calls of read i (bit patterns)"""
    h = hashlib.sha256()
    pos = ctx.positions(i, int(s["n_positions"][i]))
    for k in ("coord", "query_idx", "ref_idx", "indel", "n_signal", "core", "residual", "signal"):
        h.update(np.ascontiguousarray(pos[k]).tobytes())
    calls = ctx.hmm_calls(i, int(s["n_hmm_calls"][i]))
    for k in ("pos_on_ref", "n_events", "llr"):
        h.update(np.ascontiguousarray(calls[k]).tobytes())
    t = ctx.align_table(i, int(ctx.align_rows(len(s))[i]))
    assert t["coord"].shape[0] > 100000 and set(np.unique(t["kind"]).tolist()) <= {0, 1}
    for k in ("coord", "ref_pos", "value", "kind"):
        h.update(np.ascontiguousarray(t[k]).tobytes())
    return h.hexdigest()


def test_full_size_batch_properties(model):
    reads = _reads(model)
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    fwd = list(range(N_READS))
    b = _run(ctx, reads, fwd)
    assert b.samples() > 200e6
    s1 = ctx.summaries().copy()
    assert (s1["status"] == 0).sum() >= 0.97 * N_READS
    # ---- idempotence ----
    ctx.run("normalise"); ctx.sync()
    s2 = ctx.summaries()
    for k in FIELDS:
        assert np.asarray(s1[k]).tobytes() == np.asarray(s2[k]).tobytes(), k
    # ---- monotone alignment paths, tiling events (every 25th read: the getters copy whole arrays) ----
    for i in range(0, N_READS, 25):
        if s1["status"][i] != 0:
            continue
        ae, ak = ctx.alignment(i, int(s1["n_aligned"][i]))
        de, dk = np.diff(ae.astype(np.int64)), np.diff(ak.astype(np.int64))
        assert ((de == 1) & (dk == 1) | (de == 1) & (dk == 0) | (de == 0) & (dk == 1)).all(), i   # D, U, L moves only
        _, st, ln = ctx.events(i, int(s1["n_events"][i]))
        st, ln = st.astype(np.int64), ln.astype(np.int64)
        assert (ln > 0).all() and (st[1:] >= st[:-1] + ln[:-1]).all(), i
    # ---- spot parity against the oracle ----
    for i in (0, 1, 498, 499, 998, 999):
        o = po.OracleRead(reads[i], model)
        st = o.normalise()
        assert (s1["status"][i] != 0) == (st != 0), i
        assert s1["n_events"][i] == o.norm.n_events and s1["n_aligned"][i] == o.norm.n_aln, i
        if st == 0:
            assert np.float64(s1["shift"][i]).tobytes() == np.float64(o.norm.shift).tobytes(), i
            assert np.float64(s1["scale"][i]).tobytes() == np.float64(o.norm.scale).tobytes(), i
            ae, ak = ctx.alignment(i, int(s1["n_aligned"][i])); we, wk = o.alignment()
            assert np.array_equal(ae, we) and np.array_equal(ak, wk), i
        o.free()
    d_fwd = [_per_read_digest(s1, i) for i in range(N_READS)]
    desc, blob, _ = cnn_model.default_model()
    ctx.load_cnn(desc, blob); ctx.load_fit_models(*synth.fit_models())
    ctx.set_align_table(True)                                # `DNAscent align` table too: 230 M rows at this size
    ctx.run("hmm"); ctx.run("eventalign"); ctx.run("cnn"); ctx.sync()
    rows_fwd = ctx.align_rows(N_READS)
    assert int(rows_fwd.sum()) > 200e6
    sp = ctx.summaries().copy()
    probe = [i for i in range(0, N_READS, 10) if sp["status"][i] == 0]
    p_fwd = {}
    for i in probe:
        p = ctx.probabilities(i, int(sp["n_positions"][i]))
        assert p.shape[0] > 15000 and np.allclose(p.sum(1), 1.0, atol=1e-5) and (p >= 0).all(), i
        p_fwd[i] = hashlib.sha256(p.tobytes()).hexdigest() + _aux_digest(ctx, sp, i)
    # ---- permutation invariance: reversed batch order, same per-read digests ----
    rev = fwd[::-1]
    _run(ctx, reads, rev)
    s3 = ctx.summaries()
    d_rev = [_per_read_digest(s3, j) for j in range(N_READS)]
    assert d_rev == d_fwd[::-1]
    ctx.run("hmm"); ctx.run("eventalign"); ctx.run("cnn"); ctx.sync()
    sq = ctx.summaries()
    for i in probe:
        j = N_READS - 1 - i
        assert int(sq["n_positions"][j]) == int(sp["n_positions"][i])
        assert hashlib.sha256(ctx.probabilities(j, int(sq["n_positions"][j])).tobytes()).hexdigest() + _aux_digest(ctx, sq, j) == p_fwd[i], i
    assert ctx.cnn_range_escalations() == 0
    # checksum of checksums, for the log
    print("full-size digest", hashlib.sha256("".join(d_fwd).encode()).hexdigest()[:16], "passing", int((s1["status"] == 0).sum()))
    ctx.close()
