"""Exploration beyond tests/test_gpu_fuzz.py::test_signal_shape_fuzz_matches_oracle: many more mutated reads (tests/adversarial_signals.py mutate()), batch after batch,
device against oracle; prints the first mismatch of every failing batch instead of stopping.   python tools/gpu_shape_fuzz.py [batches] [seed0]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import adversarial_signals as adv
import test_gpu_fuzz as tf
from dnascent_amd import synth

nb_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
model = synth.pore_model()
bad = ties = 0
for b in range(nb_batches):
    reads, specs = [], []
    for i in range(int(os.environ.get("DN_FUZZ_READS", "40"))):
        seed = seed0 + 40 * b + i
        sizes = [int(x) for x in os.environ.get("DN_FUZZ_BASES", "800,1500,2500,4000,6000,9000").split(",")]
        nb = sizes[seed % len(sizes)]
        r = synth.make_read(seed, nb, model=model, is_reverse=bool(seed & 1), sub_rate=0.002, ins_rate=0.001, del_rate=0.001, noise_pa=[1.6, 1.0, 2.5][seed % 3])
        r.adc, done = adv.mutate(r.adc, seed)
        if os.environ.get("DN_FUZZ_CAL"):                    # another digitisation: a random calibration (offset, scale), the signal re-quantised to it
            rng = np.random.default_rng(seed)
            sc = np.float32(rng.uniform(0.04, 0.45)); off = np.float32(rng.uniform(-600, 600) if rng.random() < 0.5 else rng.integers(-600, 600))
            pa = (r.adc.astype(np.float64) + r.cal_offset) * r.cal_scale
            r.adc = np.clip(np.rint(pa / float(sc) - float(off)), -32768, 32767).astype(np.int16)
            r.cal_offset, r.cal_scale = float(off), float(sc)
            done = done + [("cal", float(off), float(sc))]
        reads.append(r); specs.append((seed, nb, done))
    try:
        tf._compare_batch(model, reads, specs, 0, 0)
        print("batch %d ok" % b, flush=True)
    except AssertionError:
        first = traceback.format_exc()[-600:]
        # a tie the emission's last bits decide?  Then the oracle with the DEVICE's emission formula agrees with the device
        import ctypes, pyoracle as po
        L = po.oracle(); L.dno_set_device_emission.argtypes = [ctypes.c_int]
        L.dno_set_device_emission(1)
        try:
            tf._compare_batch(model, reads, specs, 0, 0)
            ties += 1
            print("batch %d: a label differs from the reference-arithmetic oracle and agrees with the oracle that uses the device's emission formula (a tie): %s" % (b, first.strip().splitlines()[-1][:300]), flush=True)
        except AssertionError:
            bad += 1
            print("batch %d MISMATCH (also with the device's emission formula):" % b, traceback.format_exc()[-1500:], flush=True)
        finally:
            L.dno_set_device_emission(0)
print("batches with a real mismatch: %d of %d; batches with a libm-decided tie: %d" % (bad, nb_batches, ties))
