"""Independent CPU rendering of the CNN description (dnascent_amd/cnn_model.py) with stock PyTorch fp32 ops.

TEST INFRASTRUCTURE.  Built from the RAW parameters `default_model()` returns (unfolded BatchNorm, separate bias), so it
also checks the folding the HIP executor relies on.  One read at a time, [1, C, L] tensors, "same" zero padding == the zero
rows csrc/k3_cnn.hip keeps between reads."""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-3
_DT = torch.float32          # run(..., dtype=torch.float64) renders in double (the precision fuzz separates the fp32 reference's own error)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_DT)


def _gru_layer(x_seq, mask, p, return_sequences):
    """Keras GRU (reset_after, gate order z r h) with a Masking layer in front; x_seq [N, T, D], mask [N, T] bool."""
    K = _t(p["kernel"]); R = _t(p["recurrent"]); b = _t(p["bias"])
    N, T, _ = x_seq.shape
    h = torch.zeros(N, 16, dtype=_DT)
    out = []
    for t in range(T):
        xg = x_seq[:, t, :] @ K + b[0]
        hg = h @ R + b[1]
        z = torch.sigmoid(xg[:, 0:16] + hg[:, 0:16])
        r = torch.sigmoid(xg[:, 16:32] + hg[:, 16:32])
        c = torch.tanh(xg[:, 32:48] + r * hg[:, 32:48])
        hn = z * h + (1.0 - z) * c
        h = torch.where(mask[:, t:t + 1], hn, h)
        out.append(h)
    return torch.stack(out, 1) if return_sequences else h


def _onehot_digits(idx1, n_digits):
    v = idx1.astype(np.int64) - 1
    cols = []
    for j in range(n_digits):
        d = (v >> (2 * (n_digits - 1 - j))) & 3
        cols.append(np.eye(4, dtype=np.float32)[d])
    return np.concatenate(cols, 1)


def _bn(x, bn):
    g, b, m, v = (_t(bn[k]) for k in ("gamma", "beta", "mean", "var"))
    return F.batch_norm(x, m, v, g, b, training=False, eps=EPS)


def _conv(x, c):
    w = _t(c["w"]).permute(2, 1, 0).contiguous()          # [k, cin, cout] -> [cout, cin, k]
    y = F.conv1d(x, w, _t(c["b"]), padding=(w.shape[2] - 1) // 2)
    if c["bn"] is not None:
        y = _bn(y, c["bn"])
    return F.relu(y) if c["relu"] else y


def _dw(x, w):
    wt = _t(w).t().contiguous().unsqueeze(1)              # [k, c] -> [c, 1, k]
    return F.conv1d(x, wt, None, padding=(wt.shape[2] - 1) // 2, groups=wt.shape[0])


def run(ref, core, resid, sig20, dtype=None):
    """core, resid [L] (1-based indices as floats), sig20 [L, 20] -> probabilities [L, 3] (numpy, in `dtype`: float32 unless told otherwise)."""
    global _DT
    old, _DT = _DT, (dtype or torch.float32)
    try:
        return _run(ref, core, resid, sig20)
    finally:
        _DT = old


def _run(ref, core, resid, sig20):
    L = core.shape[0]
    with torch.no_grad():
        s = _t(np.ascontiguousarray(sig20, np.float32)).reshape(L, 20, 1)
        mask = s[:, :, 0] != 0.0
        h1 = _gru_layer(s, mask, ref["gru"]["g1"], True)
        h2 = _gru_layer(h1, mask, ref["gru"]["g2"], False)
        x = np.concatenate([h2.numpy().astype(np.float64), _onehot_digits(core, 5), _onehot_digits(resid, 4), np.zeros((L, 12), np.float32)], 1)
        x = _t(x).t().unsqueeze(0)                        # [1, 64, L]
        for kind, p in ref["ops"]:
            if kind == "conv":
                x = _conv(x, p)
            else:
                y = x
                for dw, pw in p["chain"]:
                    y = _conv(_dw(y, dw), pw)
                x = F.relu(y + _conv(x, p["shortcut"]))
        w, b = ref["dense"]
        z = x[0].t() @ _t(w) + _t(b)
        return torch.softmax(z, 1).numpy()
