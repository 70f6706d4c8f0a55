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
    np.save(sys.argv[1], ctx.cnn_infer(g["lens"], g["core"], g["resid"], g["signal"]))
    sys.exit(0)
for math in ("f16x3", "bf16x6"):
    out = []
    for fuse in ("0", "1"):
        path = "/tmp/fuse_%s_%s.npy" % (math, fuse)
        subprocess.check_call([sys.executable, __file__, path, math], env=dict(os.environ, DN_CNN_FUSE=fuse))
        out.append(np.load(path))
    print(math, "fused == unfused:", np.array_equal(out[0], out[1]), "max |d| = %.3e" % float(np.abs(out[0] - out[1]).max()))
