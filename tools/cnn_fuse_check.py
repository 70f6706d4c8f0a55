"""GPU: the fused SeparableConv1D kernel must give bit-identical probabilities to the two-kernel path (DN_CNN_FUSE=0)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np

if len(sys.argv) > 1:
    from dnascent_amd import cnn_model, hip
    g = np.load(os.path.join(ROOT, "tests", "golden", "cnn_default_model.npz"))
    desc, blob, _ = cnn_model.default_model()
    ctx = hip.Context(0); ctx.load_cnn(desc, blob); ctx.cnn_set_math(sys.argv[2])
    small = ctx.cnn_infer(g["lens"], g["core"], g["resid"], g["signal"])
    # ... and a set large enough (40 sequences, 200 k positions, ~1 600 row tiles) that workgroups of neighbouring tiles really run
    # at different times: the fused kernel reads halo rows that belong to other tiles
    rng = np.random.default_rng(7)
    lens = rng.integers(3000, 7000, 40).astype(np.uint32); L = int(lens.sum())
    core = rng.integers(1, 4 ** 5 + 1, L).astype(np.float32); resid = rng.integers(1, 4 ** 4 + 1, L).astype(np.float32)
    sig = rng.normal(0.0, 1.0, (L, 20)).astype(np.float32)
    sig[np.arange(20)[None, :] >= rng.integers(3, 21, L)[:, None]] = 0.0
    big = ctx.cnn_infer(lens, core, resid, sig)
    np.save(sys.argv[1], np.concatenate([small, big]))
    sys.exit(0)
for math in ("f16x3", "bf16x6"):
    out = []
    for fuse in ("0", "1"):
        path = "/tmp/fuse_%s_%s.npy" % (math, fuse)
        subprocess.check_call([sys.executable, __file__, path, math], env=dict(os.environ, DN_CNN_FUSE=fuse))
        out.append(np.load(path))
    print(math, "fused == unfused:", np.array_equal(out[0], out[1]), "max |d| = %.3e" % float(np.abs(out[0] - out[1]).max()))
