"""Explain one mismatch of tools/gpu_shape_fuzz.py: rebuild the read of a seed, run the device and the oracle in both emission modes, print where the positions differ, the
sequence around it and whether the oracle with the device's emission formula agrees with the device (then it is a tie the emission's last bits decide: DESIGN.md s3).
    python tools/gpu_fuzz_explain.py <seed> [<seed> ...]      (DN_FUZZ_BASES / DN_FUZZ_CAL as in gpu_shape_fuzz.py)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import adversarial_signals as adv, pyoracle as po
from dnascent_amd import hip, host, synth

model = synth.pore_model()
L = po.oracle(); L.dno_set_device_emission.argtypes = [ctypes.c_int]
for seed in [int(a) for a in sys.argv[1:]]:
    sizes = [int(x) for x in os.environ.get("DN_FUZZ_BASES", "800,1500,2500,4000,6000,9000").split(",")]
    nb = sizes[seed % len(sizes)]
    r = synth.make_read(seed, nb, model=model, is_reverse=bool(seed & 1), sub_rate=0.002, ins_rate=0.001, del_rate=0.001, noise_pa=[1.6, 1.0, 2.5][seed % 3])
    r.adc, done = adv.mutate(r.adc, seed)
    print("seed %d: %d bases, %s, edits %s" % (seed, nb, "rev" if r.is_reverse else "fwd", done))
    ctx = hip.Context(0); ctx.load_pore_model(model, 0.14)
    b = host.ReadBatch(); assert b.add_synth(r) >= 0
    b.upload(ctx); ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    s = ctx.summaries()
    got = ctx.positions(0, int(s["n_positions"][0]))
    for mode in (0, 1):
        L.dno_set_device_emission(mode)
        o = po.OracleRead(r, model)
        st = o.normalise(); ea = o.eventalign() if st == 0 else None
        want = o.positions() if st == 0 else None
        name = "oracle with the DEVICE's emission formula" if mode else "oracle, reference arithmetic"
        if want is None or want["coord"].shape != got["coord"].shape:
            print("  %s: status %s / %s, positions %s vs device %d" % (name, st, ea, None if want is None else want["coord"].shape[0], got["coord"].shape[0]))
        else:
            d = np.flatnonzero((got["ref_idx"] != want["ref_idx"]) | (got["n_signal"] != want["n_signal"]))
            print("  %s: %d of %d positions differ" % (name, d.shape[0], got["coord"].shape[0]))
            for k in d[:6]:
                a = int(want["ref_idx"][k])
                print("     k %d: ref_idx device %d oracle %d, n_signal device %s oracle %s, sequence %s" % (k, got["ref_idx"][k], a, got["n_signal"][max(0, k - 1):k + 2], want["n_signal"][max(0, k - 1):k + 2],
                                                                                                                 r.refseq[max(0, a - 10):a + 11].tobytes().decode()))
        o.free()
    L.dno_set_device_emission(0)
    ctx.close()
