"""Activation scale through the random-init CNN (torch rendering) -- used to pick init gains that keep the softmax unsaturated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import torch.nn.functional as F
from dnascent_amd import cnn_model
import cnn_torch_ref as T
_, _, ref = cnn_model.default_model()
rng = np.random.default_rng(5); L = 400
core = rng.integers(1, 1025, L).astype(np.float32); resid = rng.integers(1, 257, L).astype(np.float32)
sig = rng.normal(0, 1, (L, 20)).astype(np.float32)
with torch.no_grad():
    s = torch.from_numpy(sig).reshape(L, 20, 1); mask = s[:, :, 0] != 0
    h1 = T._gru_layer(s, mask, ref["gru"]["g1"], True); h2 = T._gru_layer(h1, mask, ref["gru"]["g2"], False)
    x = np.concatenate([h2.numpy(), T._onehot_digits(core, 5), T._onehot_digits(resid, 4), np.zeros((L, 12), np.float32)], 1)
    x = torch.from_numpy(x).t().unsqueeze(0)
    for kind, p in ref["ops"]:
        if kind == "conv":
            x = T._conv(x, p)
        else:
            y = x
            for dw, pw in p["chain"]:
                y = T._conv(T._dw(y, dw), pw)
            x = F.relu(y + T._conv(x, p["shortcut"]))
        print(kind, "std %.3f max %.2f" % (x.std().item(), x.abs().max().item()))
    w, b = ref["dense"]; z = x[0].t() @ torch.from_numpy(w) + torch.from_numpy(b)
    pr = torch.softmax(z, 1)
    print("logit std", z.std(0).numpy(), "prob std", pr.std(0).numpy(), "prob mean", pr.mean(0).numpy())
