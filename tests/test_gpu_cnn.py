"""K3 parity: HIP CNN executor (dn_run_cnn, C-ABI) vs the stock-PyTorch fp32 rendering of the same description.
Tolerance: 1e-4 absolute on the class probabilities (BASELINE.json north_star)."""
import numpy as np
import pytest

from dnascent_amd import cnn_model, hip, host, synth
import cnn_torch_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run(model, specs, env_rows=None, monkeypatch=None):
    if env_rows is not None:
        monkeypatch.setenv("DN_CNN_ROWS", str(env_rows))
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in specs]
    desc, blob, ref = cnn_model.default_model()
    ctx = hip.Context(0)
    ctx.load_pore_model(model, 0.14)
    ctx.load_cnn(desc, blob)
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx)
    ctx.run("normalise"); ctx.run("eventalign"); ctx.run("cnn")
    ctx.sync()
    return ctx, ref


def _check(ctx, ref, n_reads):
    s = ctx.summaries()
    worst = 0.0
    for r in range(n_reads):
        if s["status"][r] != 0:
            continue
        n = int(s["n_positions"][r])
        pos = ctx.positions(r, n)
        got = ctx.probabilities(r, n)
        want = cnn_torch_ref.run(ref, pos["core"], pos["residual"], pos["signal"])
        assert got.shape == want.shape
        worst = max(worst, float(np.abs(got - want).max()))
        assert np.allclose(got.sum(1), 1.0, atol=1e-5)
    return worst, s


GOOD = dict(sub_rate=0.003, ins_rate=0.001, del_rate=0.001)


def test_cnn_matches_torch_reference(model):
    specs = [(100, 2500, GOOD), (101, 3600, dict(is_reverse=True)), (102, 3000, GOOD)]
    ctx, ref = _run(model, specs)
    worst, s = _check(ctx, ref, len(specs))
    assert (s["status"] == 0).all() and (s["n_positions"] > 1000).all()
    assert worst < TOL, worst


def test_cnn_failed_read_and_multiple_passes(model, monkeypatch):
    """A read that fails QC produces no rows; a small row cap forces several passes with the same answers."""
    specs = [(7, 2500, GOOD), (8, 3000, dict(noise_pa=6.5)), (9, 2700, GOOD), (10, 2400, dict(is_reverse=True))]
    ctx, ref = _run(model, specs, env_rows=4096, monkeypatch=monkeypatch)
    worst, s = _check(ctx, ref, len(specs))
    assert s["status"][1] != 0 and s["status"][0] == 0 and s["status"][2] == 0 and s["status"][3] == 0
    assert worst < TOL, worst


def test_detect_file_matches_oracle_records(model, tmp_path):
    """detect.cpp:852-907 for a buffer of reads through the host C++ layer: normalise -> eventalign -> runCNN -> writer.
    Expected text = the oracle's record formatter fed with the oracle's own positions and the GPU's probabilities
    (positions are bit-exact, test_gpu_eventalign.py), failed reads are skipped as the reference skips them."""
    import pyoracle as po
    specs = [(21, 2500, GOOD), (22, 3000, dict(noise_pa=6.5)), (23, 2600, dict(is_reverse=True, sub_rate=0.003, del_rate=0.003))]
    ctx, ref = _run(model, specs)
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in specs]
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0
    b.upload(ctx)
    ctx.run("normalise"); ctx.run("eventalign")
    path = str(tmp_path / "out.detect")
    header = "#Alignment a.bam\n#Genome g.fa\n"
    written = b.detect_write(ctx, path, header)
    s = ctx.summaries()
    assert written == int((s["status"] == 0).sum()) == 2
    want = header.encode()
    for i, r in enumerate(reads):
        if s["status"][i] != 0:
            continue
        o = po.OracleRead(r, model)
        assert o.normalise() == 0 and o.eventalign() == 0
        want += o.format_detect(ctx.probabilities(i, int(s["n_positions"][i])))
        o.free()
    assert open(path, "rb").read() == want


@pytest.mark.parametrize("math", ["bf16x6", "fp32", "f16x3"])
def test_cnn_infer_matches_golden_vectors(math):
    """dn_cnn_infer (the TF_SessionRun seam: three host tensors in, probabilities out) against the committed vectors;
    ragged lengths incl. a 1-position sequence and positions with no signal at all.  All three ways of multiplying: exact fp32
    MFMA, the three-piece bf16 split on the bf16 matrix cores (the default), the two-piece fp16 split."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_default_model.npz"))
    desc, blob, _ = cnn_model.default_model()
    ctx = hip.Context(0)
    ctx.load_cnn(desc, blob)
    ctx.cnn_set_math(math)
    got = ctx.cnn_infer(g["lens"], g["core"], g["resid"], g["signal"])
    err = float(np.abs(got - g["probs"]).max())
    print("math %s: max |dp| vs torch fp32 = %.3e" % (math, err))
    assert err < TOL / 5                                  # bar 1e-4; both paths sit near 2e-6
    # a sequence is independent of its neighbours in the batch: same answer alone
    n0 = int(g["lens"][0])
    alone = ctx.cnn_infer(g["lens"][:1], g["core"][:n0], g["resid"][:n0], g["signal"][:n0])
    assert np.array_equal(alone, got[:n0])
    assert ctx.cnn_infer(np.zeros(0, np.uint32), np.zeros(0), np.zeros(0), np.zeros((0, 20))).shape == (0, 3)
    assert ctx.cnn_range_escalations() == 0


def test_f16_split_escalates_when_an_activation_leaves_fp16_range():
    """The two-piece fp16 split cannot represent |x| > 65504.  A model whose first convolution is blown up by 2^20 makes the
    second one see such inputs: the device flags it and the pass is repeated with bf16 pieces -- the answer is then bit-identical
    to asking for bf16x6 in the first place, and the repeat is counted."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_default_model.npz"))
    desc, blob, _ = cnn_model.default_model()
    blob = blob.copy()
    first = next(o for o in desc["ops"] if o["op"] == "conv")
    blob[first["scale"]:first["scale"] + first["cout"]] *= 2.0 ** 20
    ctx = hip.Context(0)
    ctx.load_cnn(desc, blob)
    ctx.cnn_set_math("bf16x6")
    want = ctx.cnn_infer(g["lens"], g["core"], g["resid"], g["signal"])
    assert ctx.cnn_range_escalations() == 0
    ctx.cnn_set_math("f16x3")
    got = ctx.cnn_infer(g["lens"], g["core"], g["resid"], g["signal"])
    assert ctx.cnn_range_escalations() == 1
    assert np.array_equal(got, want, equal_nan=True)


def test_fused_separable_conv_is_bit_identical_to_two_kernels():
    """SeparableConv1D runs as one kernel (depthwise filter applied while the pointwise GEMM's A tile is staged; the 17-tap
    256-channel layers in the wave-specialised variant).  Same taps in the same order, same pieces, same MFMA order: the
    probabilities must be bit-identical to the depthwise + pointwise kernel pair (DN_CNN_FUSE=0), in both split modes."""
    import os, subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "cnn_fuse_check.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "fused == unfused" in l]
    assert len(lines) == 2 and all("True" in l for l in lines), out.stdout


def test_in_place_convolution_is_rejected():
    """Workgroups of one launch read rows / channels that others write: a description with src == dst on a convolution is an error,
    not a silent race."""
    import copy
    desc, blob, _ = cnn_model.default_model()
    d = copy.deepcopy(desc)
    first = next(o for o in d["ops"] if o["op"] == "conv")
    first["dst"] = first["src"]
    ctx = hip.Context(0)
    with pytest.raises(Exception, match="in place"):
        ctx.load_cnn(d, blob)
    ctx.load_cnn(desc, blob)                                # the context is still usable


def _blown_up_model():
    desc, blob, _ = cnn_model.default_model()
    blob = blob.copy()
    first = next(o for o in desc["ops"] if o["op"] == "conv")
    blob[first["scale"]:first["scale"] + first["cout"]] *= 2.0 ** 20
    return desc, blob


def test_deferred_escalation_through_run_detect_and_collect(model):
    """The ASYNCHRONOUS path of the same safeguard (round-2 advisor: untested): dn_run_detect only enqueues, so the fp16 range flag of
    a batch is looked at by whoever synchronises next -- dn_collect -- which then repeats the batch's network with bf16 pieces.  The
    collected calls must equal a context that was told to use bf16x6 from the start, bit for bit, the repeat is counted once, and the
    context stays on bf16 pieces (a second batch is not repeated).  A batch that raised the flag and was never collected must not leave
    it to the next batch of the context."""
    desc, blob = _blown_up_model()
    specs = [(7601, 2500, dict()), (7602, 3000, dict(is_reverse=True, **GOOD)), (7603, 2000, dict(noise_pa=6.5))]
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in specs]
    b = host.ReadBatch()
    for r in reads:
        assert b.add_synth(r) >= 0

    def fresh(math):
        c = hip.Context(0)
        c.load_pore_model(model, 0.14); c.load_cnn(desc, blob); c.cnn_set_math(math)
        return c
    ref = fresh("bf16x6")
    b.upload(ref); ref.run("detect"); want = ref.collect()
    assert ref.cnn_range_escalations() == 0 and int(want["call_off"][-1]) > 500
    ctx = fresh("f16x3")
    b.upload(ctx); ctx.run("detect")
    got = ctx.collect()                                     # the flag is seen here: the network of the batch runs again in bf16x6
    assert ctx.cnn_range_escalations() == 1
    for k in ("call_off", "ref_coord", "p_edu", "p_brdu", "kmer"):
        assert got[k].tobytes() == want[k].tobytes(), k
    b.upload(ctx); ctx.run("detect"); again = ctx.collect()  # stays on bf16 pieces: no second repeat, same answer
    assert ctx.cnn_range_escalations() == 1 and again["p_edu"].tobytes() == want["p_edu"].tobytes()
    # dropped batch: flag raised, never collected; the next upload must start clean (and remember that the model does not fit fp16)
    drop = fresh("f16x3")
    b.upload(drop); drop.run("detect"); drop.sync()
    b.upload(drop); drop.run("detect"); after = drop.collect()
    assert after["p_edu"].tobytes() == want["p_edu"].tobytes() and drop.cnn_range_escalations() == 1
    for c in (ref, ctx, drop):
        c.close()


def test_cnn_description_must_start_with_the_encoder():
    """the encoder writes the row validity mask every later epilogue reads: a description without it first is rejected, not run on
    stale lane state (round-2 advisor)"""
    import copy
    desc, blob, _ = cnn_model.default_model()
    d = copy.deepcopy(desc)
    d["ops"] = d["ops"][1:]
    ctx = hip.Context(0)
    with pytest.raises(Exception, match="ENCODE_GRU"):
        ctx.load_cnn(d, blob)
    ctx.close()


def _hbm_used():
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert rt.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return total.value - free.value


def test_cnn_lanes_are_released_with_the_last_context_and_by_dn_shutdown(monkeypatch):
    """The CNN lanes (stream + activation buffers, shared by the contexts of a device) used to outlive every context (round-2 advisor: ~64 GB
    per device in a long-lived host).  They are counted in dn_device_bytes, go with the LAST context of their device, and dn_shutdown
    frees them under live contexts (they come back on the next pass)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_default_model.npz"))
    desc, blob, _ = cnn_model.default_model()
    monkeypatch.setenv("DN_CNN_ROWS", str(1 << 20))
    lens = np.tile(g["lens"], 40); core = np.tile(g["core"], 40); resid = np.tile(g["resid"], 40); sig = np.tile(g["signal"], (40, 1))
    base = _hbm_used()
    a = hip.Context(0); b = hip.Context(0)
    a.load_cnn(desc, blob); b.load_cnn(desc, blob)
    before = a.device_bytes()
    want = a.cnn_infer(lens, core, resid, sig)
    lane_bytes = a.device_bytes() - before
    assert lane_bytes > 16 * int(lens.sum()) * 256                       # 4 activation buffers x 256 floats per row are now counted
    held = _hbm_used()
    assert held - base > lane_bytes
    assert hip.lib().dn_shutdown() == 0                                  # lanes freed under live contexts ...
    assert _hbm_used() < held - 0.9 * lane_bytes and a.device_bytes() < before + (8 << 20)
    again = b.cnn_infer(lens, core, resid, sig)                          # ... and come back
    assert np.array_equal(again, want)
    a.close()
    mid = _hbm_used()
    b.close()                                                            # the last context of the device takes the lanes with it
    assert _hbm_used() < mid - 0.9 * lane_bytes and _hbm_used() - base < (256 << 20)
