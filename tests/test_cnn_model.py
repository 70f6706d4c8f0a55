"""CPU checks of the CNN description (no GPU): recovered layer inventory and the torch rendering."""
import numpy as np

from dnascent_amd import cnn_model, hip
import cnn_torch_ref


def _index():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cnn_variables_index.json")))["entries"]


def test_topology_reads_exactly_the_variables_of_the_reference_checkpoint():
    """tests/golden/cnn_variables_index.json is the reference's own dnn_models/.../variables/variables.index decoded in the build
    container (tools/parse_variables_index.py): every checkpoint variable with dtype, shape, offset and size.  The topology of
    cnn_model.build_model must read every float variable of it, by name, with exactly its shape -- layer by layer -- and nothing else."""
    ent = {k: e for k, e in _index().items() if e["dtype"] == "float32"}
    want = cnn_model.expected_checkpoint_variables()
    assert sorted(want) == sorted(ent) and len(ent) == 268
    for k, shape in want.items():
        assert list(ent[k]["shape"]) == list(shape), k
        assert ent[k]["size"] == 4 * int(np.prod(shape)), k
    assert sum(e["size"] for e in ent.values()) == 4 * 1817459
    # data-file offsets are contiguous: nothing is hidden between the tensors
    spans = sorted((e["offset"], e["size"]) for e in ent.values())
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))
    layers = {int(k.split("/")[0].split("-")[1]) for k in ent if k.startswith("layer_with_weights-")}
    assert layers == set(range(2, 79))                      # + layers 0, 1, 79 under trainable_variables/{0-5, 190, 191}


def test_converter_on_a_checkpoint_laid_out_like_the_reference(tmp_path):
    """tools/convert_savedmodel.py on a synthetic checkpoint: the REAL index's names / shapes / offsets (the fixture, re-encoded as
    an SSTable by the test) and seeded values in the data file.  The converted description must be the topology with exactly those
    values: BatchNorm folding, depthwise reshape, GRU blocks, and every tensor read from its own offset."""
    import importlib.util
    import os
    import struct
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("convert_savedmodel", os.path.join(root, "tools", "convert_savedmodel.py"))
    conv = importlib.util.module_from_spec(spec); spec.loader.exec_module(conv)
    ent = _index()
    # ---- a data file: every float tensor gets seeded values at its real offset ----
    rng = np.random.default_rng(11)
    total = max(e["offset"] + e["size"] for e in ent.values())
    data = np.zeros(total, np.uint8)
    vals = {}
    for k in sorted(ent):
        e = ent[k]
        if e["dtype"] != "float32":
            continue
        v = rng.normal(0, 0.3, e["size"] // 4).astype("<f4")
        if "moving_variance" in k:
            v = np.abs(v) + np.float32(0.5)
        vals[k] = v.reshape(e["shape"])
        data[e["offset"]:e["offset"] + e["size"]] = np.frombuffer(v.tobytes(), np.uint8)
    vdir = tmp_path / "variables"; vdir.mkdir()
    data.tofile(str(vdir / "variables.data-00000-of-00001"))
    # ---- the index as an SSTable (LevelDB table format, one uncompressed data block, no prefix sharing) ----

    def varint(x):
        out = b""
        while True:
            b = x & 0x7f; x >>= 7
            out += bytes([b | (0x80 if x else 0)])
            if not x:
                return out

    def entry_proto(e):
        dt = {"float32": 1, "string": 7}[e["dtype"]]
        shape = b"".join(b"\x12" + varint(len(d)) + d for d in [b"\x08" + varint(s) for s in e["shape"]])
        return b"\x08" + varint(dt) + b"\x12" + varint(len(shape)) + shape + b"\x20" + varint(e["offset"]) + b"\x28" + varint(e["size"])

    def block(items):
        body = b"".join(varint(0) + varint(len(k)) + varint(len(v)) + k + v for k, v in items)
        return body + struct.pack("<I", 0) + struct.pack("<I", 1)        # one restart point at offset 0
    items = [(b"", b"\x08\x01")] + [(k.encode(), entry_proto(ent[k])) for k in sorted(ent)]
    blk = block(items)
    f = blk + b"\x00" + b"\x00\x00\x00\x00"                          # block + type (uncompressed) + crc (unchecked)
    meta_off = len(f)
    meta = block([])
    f += meta + b"\x00" + b"\x00\x00\x00\x00"
    idx_off = len(f)
    idx = block([(b"\xff", varint(0) + varint(len(blk)))])
    f += idx + b"\x00" + b"\x00\x00\x00\x00"
    foot = varint(meta_off) + varint(len(meta)) + varint(idx_off) + varint(len(idx))
    f += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", 0xdb4775248b80fb57)
    (vdir / "variables.index").write_bytes(f)
    # the parser reads back what the fixture holds
    back = conv.parse_index(str(vdir / "variables.index"))
    assert {k: (e["shape"], e["offset"], e["size"]) for k, e in back.items()} == {k: (e["shape"], e["offset"], e["size"]) for k, e in ent.items()}
    # ---- convert, save, load ----
    desc, blob, ref = conv.convert(str(tmp_path), str(tmp_path / "model"))
    d2, b2 = cnn_model.load(str(tmp_path / "model"))
    assert d2 == desc and b2.tobytes() == blob.tobytes() and desc["synthetic_weights"] is False
    assert desc["n_weighted_layers"] == 80 and desc["keras_parameters"] == 1817459
    nm = cnn_model.ckpt_name
    # GRU blocks, untouched
    g = desc["ops"][0]
    assert np.array_equal(blob[g["g1_kernel"]:g["g1_kernel"] + 48], vals[nm(0, "kernel")].ravel())
    assert np.array_equal(blob[g["g2_recurrent"]:g["g2_recurrent"] + 768], vals[nm(1, "recurrent_kernel")].ravel())
    # stem conv (layer 2) with BatchNorm 3 folded into scale / shift
    c = desc["ops"][1]
    assert np.array_equal(blob[c["w"]:c["w"] + 3 * 64 * 64], vals[nm(2, "kernel")].ravel())
    s = vals[nm(3, "gamma")] / np.sqrt(vals[nm(3, "moving_variance")] + np.float32(1e-3))
    assert np.allclose(blob[c["scale"]:c["scale"] + 64], s, rtol=1e-6)
    assert np.allclose(blob[c["shift"]:c["shift"] + 64], (vals[nm(2, "bias")] - vals[nm(3, "moving_mean")]) * s + vals[nm(3, "beta")], rtol=1e-5, atol=1e-6)
    # first separable layer of block B1 (layer 32): depthwise [9, 64, 1] -> [9, 64], pointwise [1, 64, 128], BatchNorm 33
    ops = desc["ops"]
    dw = [o for o in ops if o["op"] == "dwconv" and o["k"] == 9 and o["c"] == 64][0]
    pw = ops[ops.index(dw) + 1]
    assert np.array_equal(blob[dw["w"]:dw["w"] + 9 * 64], vals[nm(32, "depthwise_kernel")].reshape(9, 64).ravel())
    assert pw["k"] == 1 and np.array_equal(blob[pw["w"]:pw["w"] + 64 * 128], vals[nm(32, "pointwise_kernel")].ravel())
    s = vals[nm(33, "gamma")] / np.sqrt(vals[nm(33, "moving_variance")] + np.float32(1e-3))
    assert np.allclose(blob[pw["scale"]:pw["scale"] + 128], s, rtol=1e-6)
    # dense head (layer 79)
    d = ops[-1]
    assert np.array_equal(blob[d["w"]:d["w"] + 192], vals[nm(79, "kernel")].ravel()) and np.array_equal(blob[d["b"]:d["b"] + 3], vals[nm(79, "bias")])
    # and the converted model is a working network for the independent rendering
    p = cnn_torch_ref.run(ref, np.arange(1, 41, dtype=np.float32), np.arange(1, 41, dtype=np.float32), rng.normal(0, 1, (40, 20)).astype(np.float32))
    assert p.shape == (40, 3) and np.allclose(p.sum(1), 1.0, atol=1e-5)


def test_inventory_matches_variables_index():
    desc, blob, ref = cnn_model.default_model()
    assert desc["synthetic_weights"] is True                # seeded random values: not the trained network
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
