"""Model description of the BrdU/EdU detect CNN -- topology as DATA.

The reference ships the network as a TensorFlow SavedModel (`dnn_models/detect_model_BrdUEdU_DNAr10_4_1`); in the
reference checkout only `variables/variables.index` survives, which pins every weight SHAPE (80 weighted layers,
1 817 459 fp32 parameters) but neither the graph wiring nor any value.  The HIP executor (csrc/k3_cnn.hip) therefore
runs a model DESCRIPTION: an ordered list of ops over named activation buffers plus one flat fp32 weight blob.
`default_model()` builds a description whose weighted layers have exactly the recovered shapes, in the recovered order;
the parts `variables.index` cannot tell are explicit assumptions (ASSUMED below) and are data, not code:

  ASSUMED  sequence encoding: there is no Embedding variable, so the core (5-mer, 1..1024) and residual (4-mer,
           1..256) indices are expanded to one-hot base digits (20 + 16 channels) and concatenated with the 16-wide
           GRU output and 12 zero channels to the 64 channels layer 2 expects;
  ASSUMED  ReLU activations, "same" padding, BatchNorm eps 1e-3 (Keras defaults), softmax head;
  ASSUMED  GRU: Keras defaults (tanh / sigmoid, reset_after, gate order z r h), zero samples masked (reads.h:161).

A converter for the real SavedModel only has to emit this description + blob (tools/convert_savedmodel.py documents
the mapping); nothing in the executor changes.
"""
import json

import numpy as np

OPS = ("encode_gru", "conv", "dwconv", "add_relu", "dense_softmax")


class ModelBuilder:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.w = []
        self.n = 0
        self.ops = []
        self.n_weighted = 0
        self.keras_params = 0

    def _put(self, arr):
        arr = np.ascontiguousarray(arr, np.float32).ravel()
        off = self.n
        self.w.append(arr)
        self.n += arr.shape[0]
        return off

    def _he(self, shape, fan_in):
        return self.rng.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)

    def bn(self, c, gain=1.0):
        """BatchNorm(c): gamma, beta, moving mean, moving variance (4 x [c]) -- returned raw, folded by the caller."""
        self.n_weighted += 1
        self.keras_params += 4 * c
        return dict(gamma=(gain * self.rng.uniform(0.8, 1.2, c)).astype(np.float32), beta=self.rng.normal(0, 0.05, c).astype(np.float32),
                    mean=self.rng.normal(0, 0.05, c).astype(np.float32), var=self.rng.uniform(0.5, 1.5, c).astype(np.float32))

    # ---- ops -------------------------------------------------------------------------------------------------
    def encode_gru(self, dst):
        g = {}
        for name, din in (("g1", 1), ("g2", 16)):
            self.n_weighted += 1
            self.keras_params += din * 48 + 16 * 48 + 96
            g[name] = dict(kernel=self.rng.normal(0, 0.5, (din, 48)).astype(np.float32),
                           recurrent=self.rng.normal(0, 0.25, (16, 48)).astype(np.float32),
                           bias=self.rng.normal(0, 0.1, (2, 48)).astype(np.float32))
        op = dict(op="encode_gru", dst=dst, cout=64, units=16, steps=20)
        for name in ("g1", "g2"):
            for k in ("kernel", "recurrent", "bias"):
                op["%s_%s" % (name, k)] = self._put(g[name][k])
        self.ops.append(op)
        return g

    def conv(self, src, dst, k, cin, cout, bias=True, bn=None, relu=False, pointwise_of=None, add=-1):
        """Conv1D [k, cin, cout] (+bias) with an optional folded BatchNorm and ReLU in the epilogue."""
        if pointwise_of is None:
            self.n_weighted += 1
        self.keras_params += k * cin * cout + (cout if bias else 0)
        w = self._he((k, cin, cout), k * cin)
        b = self.rng.normal(0, 0.05, cout).astype(np.float32) if bias else np.zeros(cout, np.float32)
        scale = np.ones(cout, np.float32); shift = b.copy()
        eps = 1e-3
        if bn is not None:
            s = bn["gamma"] / np.sqrt(bn["var"] + np.float32(eps))
            scale = s.astype(np.float32)
            shift = ((b - bn["mean"]) * s + bn["beta"]).astype(np.float32)
        op = dict(op="conv", src=src, dst=dst, k=k, cin=cin, cout=cout, relu=bool(relu), add=add, w=self._put(w), scale=self._put(scale),
                  shift=self._put(shift))            # add >= 0: residual join fused into the epilogue, y = act(conv + buf[add])
        self.ops.append(op)
        return dict(w=w, b=b, bn=bn, relu=relu)

    def dwconv(self, src, dst, k, c):
        self.n_weighted += 1                      # a SeparableConv1D is ONE weighted Keras layer (depthwise + pointwise + bias)
        self.keras_params += k * c
        w = self.rng.normal(0.0, np.sqrt(1.0 / k), (k, c)).astype(np.float32)   # no activation between depthwise and pointwise
        self.ops.append(dict(op="dwconv", src=src, dst=dst, k=k, c=c, w=self._put(w)))
        return w

    def add_relu(self, a, b, dst, c):
        self.ops.append(dict(op="add_relu", a=a, b=b, dst=dst, c=c))

    def dense_softmax(self, src, cin, cout):
        self.n_weighted += 1
        self.keras_params += cin * cout + cout
        w = self.rng.normal(0, np.sqrt(4.0 / cin), (cin, cout)).astype(np.float32)
        b = self.rng.normal(0, 0.05, cout).astype(np.float32)
        self.ops.append(dict(op="dense_softmax", src=src, cin=cin, cout=cout, w=self._put(w), b=self._put(b)))
        return w, b

    def finish(self):
        blob = np.concatenate(self.w) if self.w else np.zeros(0, np.float32)
        return dict(version=1, n_buffers=4, ops=self.ops, n_weights=int(blob.shape[0])), blob


def default_model(seed=2025):
    """The SURVEY s2.3 architecture: 2 GRUs, stem conv, residual blocks A1 A2 (k5, 64), B1 B2 (k9, 128), C1 (k17, 256),
    three head convs, TimeDistributed Dense(3) + softmax.  Returns (description dict, fp32 blob, reference-parameter dict)."""
    mb = ModelBuilder(seed)
    ref = {"ops": []}
    ref["gru"] = mb.encode_gru(dst=0)
    # layers 2-3: Conv1D k3 64->64 (+bias), BatchNorm, ReLU
    bn = mb.bn(64)
    ref["ops"].append(("conv", mb.conv(0, 1, 3, 64, 64, bias=True, bn=bn, relu=True)))
    cur = 1

    def block(cur, k, cin, cout):
        # shortcut: Conv1D k [k,cin,cout] + BN ; main: 6 x SeparableConv1D k (BN+ReLU between, BN after the last); add; ReLU
        free = [b for b in range(4) if b != cur]
        sc_buf, t1, t2 = free
        chain = []
        c_in = cin
        src = cur
        for j in range(6):
            dw = mb.dwconv(src, t1, k, c_in)
            bnj = mb.bn(cout, gain=1.0 if j < 5 else 0.7)     # random-init values chosen so activations stay O(1) through the joins
            pw = mb.conv(t1, t2, 1, c_in, cout, bias=True, bn=bnj, relu=(j < 5), pointwise_of="sep")
            chain.append((dw, pw))
            src = t2                              # the next depthwise reads t2 and overwrites t1
            c_in = cout
        bns = mb.bn(cout, gain=0.7)
        sc = mb.conv(cur, sc_buf, k, cin, cout, bias=True, bn=bns, relu=True, add=t2)     # Add + ReLU ride in the epilogue
        sc["relu"] = False                        # the RAW shortcut layer has no activation of its own (reference rendering)
        ref["ops"].append(("block", dict(k=k, cin=cin, cout=cout, chain=chain, shortcut=sc)))
        return sc_buf

    # NOTE on weighted-layer order: 6 separable convs with 5 BNs between them, then the shortcut conv, then
    # the two BNs (main tail, shortcut) = 14 weighted layers per block.  The builder draws them in a different order, which
    # only matters to a converter (it maps by name, not by draw order).
    cur = block(cur, 5, 64, 64)      # A1
    cur = block(cur, 5, 64, 64)      # A2
    cur = block(cur, 9, 64, 128)     # B1
    cur = block(cur, 9, 128, 128)    # B2
    cur = block(cur, 17, 128, 256)   # C1
    nxt = [b for b in range(4) if b != cur]
    bn = mb.bn(256); ref["ops"].append(("conv", mb.conv(cur, nxt[0], 3, 256, 256, bias=True, bn=bn, relu=True)))
    bn = mb.bn(128); ref["ops"].append(("conv", mb.conv(nxt[0], nxt[1], 3, 256, 128, bias=True, bn=bn, relu=True)))
    ref["ops"].append(("conv", mb.conv(nxt[1], nxt[2], 3, 128, 64, bias=True, bn=None, relu=True)))
    ref["dense"] = mb.dense_softmax(nxt[2], 64, 3)
    desc, blob = mb.finish()
    desc["n_weighted_layers"] = mb.n_weighted
    desc["keras_parameters"] = mb.keras_params
    return desc, blob, ref


def dumps(desc):
    return json.dumps(desc)
