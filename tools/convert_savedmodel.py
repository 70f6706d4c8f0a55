#!/usr/bin/env python3
"""convert_savedmodel.py -- the reference's trained detect CNN -> this framework's model description + weight blob.

  python tools/convert_savedmodel.py <SavedModel dir or its variables/ dir> <out prefix>
      -> <out prefix>.json (description: op list, buffer plan, offsets)  +  <out prefix>.f32 (flat little-endian fp32 blob)
  load with dnascent_amd.cnn_model.load(prefix) and hand to dn_load_cnn (hip.Context.load_cnn).

Input: `variables/variables.index` (SSTable of BundleEntryProto: name -> dtype, shape, offset, size; tools/parse_variables_index.py)
and `variables/variables.data-00000-of-00001` (the raw little-endian tensors at those offsets).  TensorFlow is NOT needed.  The
reference checkout ships only the index (the data file and saved_model.pb are listed in .MISSING_LARGE_BLOBS), so the conversion
can only run where the real model exists; tests/test_cnn_model.py runs it on a synthetic data file laid out per the real index.

Mapping (dnascent_amd/cnn_model.py build_model walks the topology; this file only supplies the numbers):
  GRU layers 0, 1          trainable_variables/{0,1,2} and {3,4,5}: kernel [in, 48], recurrent_kernel [16, 48], bias [2, 48]
                           (Keras reset_after: row 0 input bias, row 1 recurrent bias); Keras gate order z | r | h is the
                           executor's order: no permutation
  Conv1D                   kernel [k, cin, cout] as is (the device re-lays it at load), bias [cout]
  SeparableConv1D          depthwise_kernel [k, c, 1] -> [k, c]; pointwise_kernel [1, cin, cout] -> a 1-tap conv; bias [cout]
  BatchNormalization       gamma, beta, moving_mean, moving_variance folded into the preceding conv's epilogue:
                           scale = gamma / sqrt(var + 1e-3), shift = (bias - mean) * scale + beta
  Dense (layer 79)         trainable_variables/{190,191}: kernel [64, 3], bias [3]
What the index cannot tell -- the encoding of the two sequence inputs, activations, padding, BN epsilon, which of a block's last two
BatchNorms sits on the shortcut -- is the ASSUMED part of cnn_model.py; a SavedModel graph would settle it.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from parse_variables_index import parse_index  # noqa: E402


class CheckpointSource:
    """Parameters of the topology read from a TensorFlow checkpoint (index + one data shard) by variable name."""
    synthetic = False

    def __init__(self, index_path, data_path):
        from dnascent_amd.cnn_model import ckpt_name
        self.name = ckpt_name
        self.entries = parse_index(index_path)
        self.data = np.memmap(data_path, dtype=np.uint8, mode="r")
        self.used = set()

    def get(self, layer, var, shape):
        key = self.name(layer, var)
        e = self.entries.get(key)
        if e is None:
            raise KeyError("checkpoint has no variable %s" % key)
        if e["dtype"] != "float32" or list(e["shape"]) != list(shape):
            raise ValueError("%s: expected float32 %s, checkpoint has %s %s" % (key, list(shape), e["dtype"], e["shape"]))
        if e["shard"] != 0 or e["offset"] + e["size"] > self.data.shape[0] or e["size"] != 4 * int(np.prod(shape)):
            raise ValueError("%s: offset / size outside the data file" % key)
        self.used.add(key)
        return np.frombuffer(self.data[e["offset"]:e["offset"] + e["size"]].tobytes(), dtype="<f4").reshape(shape).copy()

    def gru(self, layer, din):
        return dict(kernel=self.get(layer, "kernel", (din, 48)), recurrent=self.get(layer, "recurrent_kernel", (16, 48)), bias=self.get(layer, "bias", (2, 48)))

    def bn(self, layer, c, gain):
        return dict(gamma=self.get(layer, "gamma", (c,)), beta=self.get(layer, "beta", (c,)), mean=self.get(layer, "moving_mean", (c,)),
                    var=self.get(layer, "moving_variance", (c,)))

    def conv(self, layer, var, k, cin, cout, bias):
        return self.get(layer, var, (k, cin, cout)), (self.get(layer, "bias", (cout,)) if bias else np.zeros(cout, np.float32))

    def depthwise(self, layer, k, c):
        return self.get(layer, "depthwise_kernel", (k, c, 1)).reshape(k, c)

    def dense(self, layer, cin, cout):
        return self.get(layer, "kernel", (cin, cout)), self.get(layer, "bias", (cout,))


def convert(model_dir, out_prefix=None):
    from dnascent_amd import cnn_model
    vdir = model_dir if os.path.exists(os.path.join(model_dir, "variables.index")) else os.path.join(model_dir, "variables")
    src = CheckpointSource(os.path.join(vdir, "variables.index"), os.path.join(vdir, "variables.data-00000-of-00001"))
    desc, blob, ref = cnn_model.build_model(src)
    unused = sorted(k for k, e in src.entries.items() if e["dtype"] == "float32" and k not in src.used)
    if unused:
        raise ValueError("checkpoint variables the topology does not read: %s" % unused[:5])
    if out_prefix:
        cnn_model.save(out_prefix, desc, blob)
    return desc, blob, ref


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    d, b, _ = convert(sys.argv[1], sys.argv[2])
    print("%d ops, %d weights (%d Keras parameters in %d weighted layers) -> %s.json / .f32" %
          (len(d["ops"]), b.shape[0], d["keras_parameters"], d["n_weighted_layers"], sys.argv[2]))
