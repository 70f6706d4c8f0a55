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

Where the parameters come from is a SOURCE object: `RandomSource` (seeded values, `default_model()` -- NOT the trained network:
every output computed with it is synthetic) or `tools/convert_savedmodel.py`'s checkpoint source, which reads the variables of the
real `variables.data-00000-of-00001` by their checkpoint names (tests/golden/cnn_variables_index.json lists them as parsed from the
reference's variables.index).  Both walk the same topology (`build_model`), so a converted model differs from the default one only
in its numbers; nothing in the executor changes.
"""
import json

import numpy as np

OPS = ("encode_gru", "conv", "dwconv", "add_relu", "dense_softmax")


def ckpt_name(layer, var):
    """checkpoint key of a variable: layers 0, 1 and 79 keep theirs at depth >= 3 and therefore appear as
    trainable_variables/{0-5, 190, 191}; every other layer as layer_with_weights-<n>/<var>"""
    special = {(0, "kernel"): 0, (0, "recurrent_kernel"): 1, (0, "bias"): 2, (1, "kernel"): 3, (1, "recurrent_kernel"): 4, (1, "bias"): 5,
               (79, "kernel"): 190, (79, "bias"): 191}
    if (layer, var) in special:
        return "trainable_variables/%d/.ATTRIBUTES/VARIABLE_VALUE" % special[(layer, var)]
    return "layer_with_weights-%d/%s/.ATTRIBUTES/VARIABLE_VALUE" % (layer, var)


class RandomSource:
    """Seeded random parameters with the recovered shapes (values chosen so activations stay O(1) through the joins).
    The draw order is part of the committed golden vectors (tests/golden/cnn_default_model.npz): do not reorder."""
    synthetic = True

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def gru(self, layer, din):
        return dict(kernel=self.rng.normal(0, 0.5, (din, 48)).astype(np.float32), recurrent=self.rng.normal(0, 0.25, (16, 48)).astype(np.float32),
                    bias=self.rng.normal(0, 0.1, (2, 48)).astype(np.float32))

    def bn(self, layer, c, gain):
        return dict(gamma=(gain * self.rng.uniform(0.8, 1.2, c)).astype(np.float32), beta=self.rng.normal(0, 0.05, c).astype(np.float32),
                    mean=self.rng.normal(0, 0.05, c).astype(np.float32), var=self.rng.uniform(0.5, 1.5, c).astype(np.float32))

    def conv(self, layer, var, k, cin, cout, bias):
        w = self.rng.normal(0.0, np.sqrt(2.0 / (k * cin)), (k, cin, cout)).astype(np.float32)
        b = self.rng.normal(0, 0.05, cout).astype(np.float32) if bias else np.zeros(cout, np.float32)
        return w, b

    def depthwise(self, layer, k, c):
        return self.rng.normal(0.0, np.sqrt(1.0 / k), (k, c)).astype(np.float32)   # no activation between depthwise and pointwise

    def dense(self, layer, cin, cout):
        return self.rng.normal(0, np.sqrt(4.0 / cin), (cin, cout)).astype(np.float32), self.rng.normal(0, 0.05, cout).astype(np.float32)


class ModelBuilder:
    def __init__(self, source):
        self.src = source
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

    def bn(self, c, gain=1.0, layer=-1):
        """BatchNorm(c): gamma, beta, moving mean, moving variance (4 x [c]) -- returned raw, folded by the caller."""
        self.n_weighted += 1
        self.keras_params += 4 * c
        return self.src.bn(layer, c, gain)

    # ---- ops -------------------------------------------------------------------------------------------------
    def encode_gru(self, dst):
        g = {}
        for layer, (name, din) in enumerate((("g1", 1), ("g2", 16))):
            self.n_weighted += 1
            self.keras_params += din * 48 + 16 * 48 + 96
            g[name] = self.src.gru(layer, din)
        op = dict(op="encode_gru", dst=dst, cout=64, units=16, steps=20)
        for name in ("g1", "g2"):
            for k in ("kernel", "recurrent", "bias"):
                op["%s_%s" % (name, k)] = self._put(g[name][k])
        self.ops.append(op)
        return g

    def conv(self, src, dst, k, cin, cout, bias=True, bn=None, relu=False, pointwise_of=None, add=-1, layer=-1):
        """Conv1D [k, cin, cout] (+bias) with an optional folded BatchNorm and ReLU in the epilogue."""
        if pointwise_of is None:
            self.n_weighted += 1
        self.keras_params += k * cin * cout + (cout if bias else 0)
        w, b = self.src.conv(layer, "pointwise_kernel" if pointwise_of is not None else "kernel", k, cin, cout, bias)
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

    def dwconv(self, src, dst, k, c, layer=-1):
        self.n_weighted += 1                      # a SeparableConv1D is ONE weighted Keras layer (depthwise + pointwise + bias)
        self.keras_params += k * c
        w = self.src.depthwise(layer, k, c)
        self.ops.append(dict(op="dwconv", src=src, dst=dst, k=k, c=c, w=self._put(w)))
        return w

    def add_relu(self, a, b, dst, c):
        self.ops.append(dict(op="add_relu", a=a, b=b, dst=dst, c=c))

    def dense_softmax(self, src, cin, cout, layer=-1):
        self.n_weighted += 1
        self.keras_params += cin * cout + cout
        w, b = self.src.dense(layer, cin, cout)
        self.ops.append(dict(op="dense_softmax", src=src, cin=cin, cout=cout, w=self._put(w), b=self._put(b)))
        return w, b

    def finish(self):
        blob = np.concatenate(self.w) if self.w else np.zeros(0, np.float32)
        return dict(version=1, n_buffers=4, ops=self.ops, n_weights=int(blob.shape[0])), blob


def build_model(source):
    """The SURVEY s2.3 architecture: 2 GRUs (layers 0, 1), stem conv + BN (2, 3), residual blocks A1 A2 (k5, 64; layers 4-17, 18-31),
    B1 B2 (k9, 128; 32-45, 46-59), C1 (k17, 256; 60-73), three head convs (74-78), TimeDistributed Dense(3) + softmax (79).
    Inside a block of 14 weighted layers starting at L: separable convs L, L+2, ..., L+10 with BatchNorms L+1, ..., L+9 between them,
    shortcut Conv1D L+11, BatchNorm L+12 after the last separable conv (ASSUMED: main branch first) and L+13 on the shortcut.
    Returns (description dict, fp32 blob, reference-parameter dict)."""
    mb = ModelBuilder(source)
    ref = {"ops": []}
    ref["gru"] = mb.encode_gru(dst=0)
    # layers 2-3: Conv1D k3 64->64 (+bias), BatchNorm, ReLU
    bn = mb.bn(64, layer=3)
    ref["ops"].append(("conv", mb.conv(0, 1, 3, 64, 64, bias=True, bn=bn, relu=True, layer=2)))
    cur = 1

    def block(cur, k, cin, cout, L):
        # shortcut: Conv1D k [k,cin,cout] + BN ; main: 6 x SeparableConv1D k (BN+ReLU between, BN after the last); add; ReLU
        free = [b for b in range(4) if b != cur]
        sc_buf, t1, t2 = free
        chain = []
        c_in = cin
        src = cur
        for j in range(6):
            dw = mb.dwconv(src, t1, k, c_in, layer=L + 2 * j)
            bnj = mb.bn(cout, gain=1.0 if j < 5 else 0.7, layer=L + 2 * j + 1 if j < 5 else L + 12)   # random-init gains keep activations O(1) through the joins
            pw = mb.conv(t1, t2, 1, c_in, cout, bias=True, bn=bnj, relu=(j < 5), pointwise_of="sep", layer=L + 2 * j)
            chain.append((dw, pw))
            src = t2                              # the next depthwise reads t2 and overwrites t1
            c_in = cout
        bns = mb.bn(cout, gain=0.7, layer=L + 13)
        sc = mb.conv(cur, sc_buf, k, cin, cout, bias=True, bn=bns, relu=True, add=t2, layer=L + 11)     # Add + ReLU ride in the epilogue
        sc["relu"] = False                        # the RAW shortcut layer has no activation of its own (reference rendering)
        ref["ops"].append(("block", dict(k=k, cin=cin, cout=cout, chain=chain, shortcut=sc)))
        return sc_buf

    cur = block(cur, 5, 64, 64, 4)       # A1
    cur = block(cur, 5, 64, 64, 18)      # A2
    cur = block(cur, 9, 64, 128, 32)     # B1
    cur = block(cur, 9, 128, 128, 46)    # B2
    cur = block(cur, 17, 128, 256, 60)   # C1
    nxt = [b for b in range(4) if b != cur]
    bn = mb.bn(256, layer=75); ref["ops"].append(("conv", mb.conv(cur, nxt[0], 3, 256, 256, bias=True, bn=bn, relu=True, layer=74)))
    bn = mb.bn(128, layer=77); ref["ops"].append(("conv", mb.conv(nxt[0], nxt[1], 3, 256, 128, bias=True, bn=bn, relu=True, layer=76)))
    ref["ops"].append(("conv", mb.conv(nxt[1], nxt[2], 3, 128, 64, bias=True, bn=None, relu=True, layer=78)))
    ref["dense"] = mb.dense_softmax(nxt[2], 64, 3, layer=79)
    desc, blob = mb.finish()
    desc["n_weighted_layers"] = mb.n_weighted
    desc["keras_parameters"] = mb.keras_params
    desc["synthetic_weights"] = bool(getattr(source, "synthetic", False))
    return desc, blob, ref


def default_model(seed=2025):
    """The architecture with SEEDED RANDOM weights (RandomSource): what the tests, smoke() and bench.py run.  Not the trained
    BrdU / EdU network -- its probabilities mean nothing biologically (description key `synthetic_weights`)."""
    return build_model(RandomSource(seed))


def expected_checkpoint_variables():
    """Every variable the topology reads, as {checkpoint key: shape}: what tests compare with the reference's variables.index."""
    class Rec:
        synthetic = False

        def __init__(self):
            self.v = {}

        def gru(self, layer, din):
            self.v[ckpt_name(layer, "kernel")] = [din, 48]; self.v[ckpt_name(layer, "recurrent_kernel")] = [16, 48]; self.v[ckpt_name(layer, "bias")] = [2, 48]
            return dict(kernel=np.zeros((din, 48), np.float32), recurrent=np.zeros((16, 48), np.float32), bias=np.zeros((2, 48), np.float32))

        def bn(self, layer, c, gain):
            for n in ("gamma", "beta", "moving_mean", "moving_variance"):
                self.v[ckpt_name(layer, n)] = [c]
            return dict(gamma=np.ones(c, np.float32), beta=np.zeros(c, np.float32), mean=np.zeros(c, np.float32), var=np.ones(c, np.float32))

        def conv(self, layer, var, k, cin, cout, bias):
            self.v[ckpt_name(layer, var)] = [k, cin, cout]
            if bias:
                self.v[ckpt_name(layer, "bias")] = [cout]
            return np.zeros((k, cin, cout), np.float32), np.zeros(cout, np.float32)

        def depthwise(self, layer, k, c):
            self.v[ckpt_name(layer, "depthwise_kernel")] = [k, c, 1]
            return np.zeros((k, c), np.float32)

        def dense(self, layer, cin, cout):
            self.v[ckpt_name(layer, "kernel")] = [cin, cout]; self.v[ckpt_name(layer, "bias")] = [cout]
            return np.zeros((cin, cout), np.float32), np.zeros(cout, np.float32)
    r = Rec()
    build_model(r)
    return r.v


def dumps(desc):
    return json.dumps(desc)


def save(prefix, desc, blob):
    """<prefix>.json (description) + <prefix>.f32 (little-endian fp32 blob): what tools/convert_savedmodel.py writes"""
    with open(prefix + ".json", "w") as f:
        json.dump(desc, f)
    np.ascontiguousarray(blob, "<f4").tofile(prefix + ".f32")


def load(prefix):
    desc = json.load(open(prefix + ".json"))
    blob = np.fromfile(prefix + ".f32", dtype="<f4")
    if blob.shape[0] != desc["n_weights"]:
        raise ValueError("%s.f32 holds %d floats, the description expects %d" % (prefix, blob.shape[0], desc["n_weights"]))
    return desc, blob
