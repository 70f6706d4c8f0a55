"""CPU checks of the CNN description (no GPU): recovered layer inventory and the torch rendering."""
import numpy as np

from dnascent_amd import cnn_model, hip
import cnn_torch_ref


def test_inventory_matches_variables_index():
    desc, blob, ref = cnn_model.default_model()
    assert desc["n_weighted_layers"] == 80                  # variables.index: 80 weighted layers
    assert desc["keras_parameters"] == 1817459              # ... holding 1 817 459 fp32 parameters
    assert blob.dtype == np.float32 and blob.shape[0] == desc["n_weights"]
    assert desc["ops"][0]["op"] == "encode_gru" and desc["ops"][-1]["op"] == "dense_softmax"
    for o in desc["ops"]:
        if o["op"] == "conv":
            assert o["cin"] % 32 == 0 and o["cout"] % 64 == 0 and o["k"] % 2 == 1
            assert o["w"] + o["k"] * o["cin"] * o["cout"] <= blob.shape[0]


def test_struct_conversion_roundtrip():
    desc, blob, _ = cnn_model.default_model()
    ops = hip.cnn_ops_from_description(desc)
    assert len(ops) == len(desc["ops"]) == 71
    assert sum(1 for o in ops if o.op == hip.CNN_OPCODE["conv_add"]) == 5
    import ctypes
    assert ctypes.sizeof(hip.CnnOp) == 40 + 24 + 48
    assert ops[1].op == hip.CNN_OPCODE["conv"] and ops[1].k == 3 and ops[1].cin == 64 and ops[1].relu == 1
    assert ops[0].aux[5] == desc["ops"][0]["g2_bias"]


def test_torch_rendering_is_a_distribution_and_local():
    _, _, ref = cnn_model.default_model()
    rng = np.random.default_rng(5)
    L = 150
    core = rng.integers(1, 1025, L).astype(np.float32); resid = rng.integers(1, 257, L).astype(np.float32)
    sig = rng.normal(0, 1, (L, 20)).astype(np.float32)
    sig[rng.random((L, 20)) < 0.3] = 0.0
    p = cnn_torch_ref.run(ref, core, resid, sig)
    assert p.shape == (L, 3) and np.all(p >= 0) and np.allclose(p.sum(1), 1.0, atol=1e-5)
    # finite receptive field: a change at the far end must not move the first positions
    sig2 = sig.copy(); sig2[-1] += 1.0
    p2 = cnn_torch_ref.run(ref, core, resid, sig2)
    assert np.array_equal(p[:10], p2[:10]) and not np.array_equal(p[-1], p2[-1])


def test_torch_rendering_reproduces_golden():
    """The committed vectors (tests/golden/make_cnn_golden.py) still come out of description + rendering: guards the model
    builder, the weight draw order and the rendering against silent drift."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_default_model.npz"))
    _, _, ref = cnn_model.default_model()
    o = 0
    for n in g["lens"]:
        n = int(n)
        p = cnn_torch_ref.run(ref, g["core"][o:o + n], g["resid"][o:o + n], g["signal"][o:o + n])
        assert np.abs(p - g["probs"][o:o + n]).max() < 1e-6
        o += n
    assert 0.02 < g["probs"][:, 2].mean() and g["probs"].std(0).min() > 0.01      # not a saturated softmax
