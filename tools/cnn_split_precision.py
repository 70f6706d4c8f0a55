"""CPU experiment: error of the class probabilities when every conv / pointwise product is done with split low-precision
operands (x = hi + lo, three products hi*hi + hi*lo + lo*hi, fp32 accumulation) -- what a 3 x bf16 / 3 x fp16 MFMA path would
compute -- against the plain fp32 rendering.  Decides whether such a path can meet the 1e-4 bar."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import torch.nn.functional as F
import cnn_torch_ref as T
from dnascent_amd import cnn_model

MODE = None


def split(x, dt):
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    return hi, lo


_orig_conv1d = F.conv1d


def conv1d(x, w, b=None, padding=0, groups=1, **kw):
    if MODE is None or groups != 1:
        return _orig_conv1d(x, w, b, padding=padding, groups=groups, **kw)
    dt = torch.bfloat16 if MODE.startswith("bf16") else torch.float16
    xh, xl = split(x, dt); wh, wl = split(w, dt)
    c = lambda a, b: _orig_conv1d(a, b, None, padding=padding)
    y = c(xh, wh)
    if MODE.endswith("x3"):
        y = y + c(xh, wl) + c(xl, wh)
    if MODE.endswith("x6"):                       # three-way split: hi, mid, lo; products down to 2^-24
        xm, xr = split(x - xh, dt); wm, wr = split(w - wh, dt)
        y = y + (c(xh, wm) + c(xm, wh)) + (c(xh, wr) + c(xr, wh) + c(xm, wm))
    if b is not None:
        y = y + b.view(1, -1, 1)
    return y


F.conv1d = conv1d
_, _, ref = cnn_model.default_model()
rng = np.random.default_rng(11); L = 2000
core = rng.integers(1, 1025, L).astype(np.float32); resid = rng.integers(1, 257, L).astype(np.float32)
sig = rng.normal(0, 1, (L, 20)).astype(np.float32); sig[rng.random((L, 20)) < 0.4] = 0
base = T.run(ref, core, resid, sig)
for m in ("bf16x1", "bf16x3", "bf16x6", "fp16x1", "fp16x3"):
    MODE = m
    p = T.run(ref, core, resid, sig)
    print("%-7s max |dp| %.3e   mean |dp| %.3e" % (m, np.abs(p - base).max(), np.abs(p - base).mean()))
