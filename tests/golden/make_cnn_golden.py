"""Generates tests/golden/cnn_default_model.npz: seeded input tensors for the detect CNN (three sequences, ragged lengths,
zero-padded signal rows as reads.h:147-172 produces them) and the class probabilities of the stock-PyTorch fp32 rendering
(tests/cnn_torch_ref.py) of dnascent_amd.cnn_model.default_model(seed=2025).

The reference's own network (TensorFlow SavedModel) is absent from its checkout, so these vectors pin OUR description
+ executor against an independent implementation, not against the reference's trained weights ("CNN parity unpinned",
DESIGN.md).  Run:  python tests/golden/make_cnn_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import cnn_torch_ref  # noqa: E402
from dnascent_amd import cnn_model  # noqa: E402


def inputs(seed=77, lens=(140, 1, 333)):
    rng = np.random.default_rng(seed)
    L = int(sum(lens))
    core = rng.integers(1, 1025, L).astype(np.float32)
    resid = rng.integers(1, 257, L).astype(np.float32)
    sig = rng.normal(0.0, 1.0, (L, 20)).astype(np.float32)
    nsig = rng.integers(0, 21, L)                 # samples present at each position; the rest of the row is zero padding
    sig[np.arange(20)[None, :] >= nsig[:, None]] = 0.0
    return np.asarray(lens, np.uint32), core, resid, sig


def main():
    _, _, ref = cnn_model.default_model()
    lens, core, resid, sig = inputs()
    out = []
    o = 0
    for n in lens:
        n = int(n)
        out.append(cnn_torch_ref.run(ref, core[o:o + n], resid[o:o + n], sig[o:o + n]))
        o += n
    probs = np.concatenate(out).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "cnn_default_model.npz"), lens=lens, core=core, resid=resid, signal=sig, probs=probs)
    print("wrote", probs.shape, probs[:2])


if __name__ == "__main__":
    main()
