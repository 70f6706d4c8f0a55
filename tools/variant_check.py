"""GPU: the alternative kernel forms kept behind environment switches must give bit-identical results to the defaults:
DN_CNN_BM256=0 (128-row
workgroups on the long-K convolutions), DN_CNN_BLOCK64=0 / 1 (the 64-channel residual blocks layer by layer / only their separable layers in one launch), DN_CNN_SEP_WS=0 (single-role fused separable kernel for the 17-tap layers), DN_TS_FULL=1
(Theil-Sen: the general first-level histogram path that a median slope outside [0.5, 2) takes).
Each variant runs in its own process (the switches are read once per process)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np

SPECS = [(901, 6000, dict()), (902, 9000, dict(is_reverse=True, sub_rate=0.003, ins_rate=0.002, del_rate=0.002)), (903, 3000, dict(noise_pa=6.5)),
         (904, 12000, dict(sub_rate=0.002)), (905, 2500, dict(is_reverse=True)), (906, 4000, dict(n_unknown=2)), (907, 7000, dict())]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from dnascent_amd import cnn_model, hip, host, synth
    model = synth.pore_model()
    desc, blob, _ = cnn_model.default_model()
    ctx = hip.Context(0); ctx.load_pore_model(model, 0.14); ctx.load_cnn(desc, blob); ctx.keep_k1(True)
    b = host.ReadBatch()
    for seed, n, kw in SPECS:
        assert b.add_synth(synth.make_read(seed, n, model=model, **kw)) >= 0
    b.upload(ctx)
    ctx.run("normalise"); ctx.run("eventalign"); ctx.run("cnn"); ctx.sync()
    s = ctx.summaries()
    h = hashlib.sha256(s.tobytes())
    for i in range(len(SPECS)):
        n = int(s["n_positions"][i])
        if s["status"][i] == 0:
            ae, ak = ctx.alignment(i, int(s["n_aligned"][i])); h.update(ae.tobytes()); h.update(ak.tobytes())
            h.update(ctx.prefix_sums(i, int(s["n_samples"][i]))[0].tobytes())
            h.update(ctx.probabilities(i, n).tobytes())
    print("DIGEST", h.hexdigest(), int((s["status"] == 0).sum()))
    sys.exit(0)


def run(env):
    out = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")]
    assert out.returncode == 0 and line, out.stderr[-2000:]
    return line[0]


base = run({})
print("default        ", base)
ok = True
for name, env in (("conv BM=128", {"DN_CNN_BM256": "0"}),
                  ("sep no-ws", {"DN_CNN_SEP_WS": "0"}), ("sep unfused", {"DN_CNN_FUSE": "0"}), ("theilsen full", {"DN_TS_FULL": "1"}),
                  ("block64 off", {"DN_CNN_BLOCK64": "0"}), ("block64 sep", {"DN_CNN_BLOCK64": "1"}), ("pair128 off", {"DN_CNN_PAIR128": "0"}), ("pair128 on", {"DN_CNN_PAIR128": "1"})):
    d = run(env)
    print("%-15s" % name, d, "same" if d == base else "DIFFERENT")
    ok = ok and d == base
print("variants agree:", ok)
sys.exit(0 if ok else 1)
