"""Hostile raw signals for the segmentation (scrappie/event_detection.c:60-115,122-198,213-266): what real nanopore signal has
and the synthetic generator (csrc/host/dn_synth.c: geometric dwell + Gaussian noise) never produces -- stalls, runs of identical
samples, open-pore spikes, square waves, drift, the densest event rate the detector can emit, reads shorter than a window.

Shared by tests/golden/make_golden.py (which runs the REFERENCE over them and stores its event tables) and the tests (oracle on
CPU, the HIP segmentation on the GPU).  Everything here is integer arithmetic on a splitmix64 stream or a splice of dn_synth
reads: no numpy generator whose stream could change between numpy versions.  The fixture stores a SHA-256 of every signal.
"""
import numpy as np

CAL = (-240.0, 0.1755)         # calibration offset / scale of the synthetic reads


def _splitmix(seed, n):
    """n uint64 draws of splitmix64, vectorised"""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _levels(seed, n, lo=300, hi=1100):
    """n pseudo-random ADC levels in [lo, hi) with consecutive levels at least 40 counts (7 pA) apart"""
    u = (_splitmix(seed, n) % np.uint64(hi - lo)).astype(np.int64) + lo
    for i in range(1, n):
        if abs(int(u[i]) - int(u[i - 1])) < 40:
            u[i] = lo + (int(u[i - 1]) - lo + 200) % (hi - lo)
    return u


def _steps(seed, run, n_samples):
    """piecewise-constant, noise-free: a new level every `run` samples"""
    nl = (n_samples + run - 1) // run
    return np.repeat(_levels(seed, nl), run)[:n_samples].astype(np.int16)


def _square(period, n_samples, lo=500, hi=900):
    half = period // 2
    return np.where((np.arange(n_samples) // half) % 2 == 0, lo, hi).astype(np.int16)


def _noise16(seed, n, amp):
    """integer noise in [-amp, amp], triangular (sum of two uniform draws)"""
    a = (_splitmix(seed, n) % np.uint64(amp + 1)).astype(np.int64)
    b = (_splitmix(seed ^ 0x5DEECE66D, n) % np.uint64(amp + 1)).astype(np.int64)
    return a - b


def normal_read(seed, n_bases, model):
    from dnascent_amd import synth
    return synth.make_read(seed, n_bases, model=model).adc.astype(np.int64)


def splice(base, at, piece):
    return np.concatenate([base[:at], piece, base[at:]])


def stall(seed, length, level, noise_amp):
    """a stalled pore: `length` samples at one level, exactly flat (noise_amp 0) or with a few counts of noise"""
    if noise_amp == 0:
        return np.full(length, level, np.int64)
    return level + _noise16(seed, length, noise_amp)


def cases(model):
    """name -> int16 signal.  Every case the reference's detect_events accepts (at least one peak)."""
    c = {}
    base = normal_read(7001, 2500, model)                                  # ~31 k samples
    n0 = base.shape[0]
    # a 6 000-sample stall inside a normal read: exactly flat (zero-variance windows: eta = FLT_MIN, event_detection.c:104) and noisy
    c["stall6000_flat"] = splice(base, n0 // 2, stall(1, 6000, 640, 0))
    c["stall6000_noisy"] = splice(base, n0 // 2, stall(2, 6000, 640, 9))
    # the flat stall entered right after a sub-threshold bump, so that a detector carries a pending peak across every chunk of it
    c["stall3000_after_bump"] = splice(base, 5000, np.concatenate([np.full(40, 700), np.full(7, 703), stall(3, 3000, 700, 0)]))
    # a NOISY stall running straight into an exactly flat one: in pure noise a detector usually holds a sub-threshold peak, and on flat signal (both
    # t-statistics 0) nothing ever replaces it -- the state at every later chunk start depends on history more than a warm-up back
    c["stall_noisy_then_flat"] = splice(base, 9000, np.concatenate([stall(4, 900, 620, 8), stall(5, 4200, 620, 0)]))
    # a bare full-range ramp: the t-statistics are constant up to integer rounding (6.36 / 15.5: above both thresholds), peaks come from rounding alone
    c["bare_ramp"] = -32768 + (np.arange(4000) * 65535) // 3999
    # runs of 3 / 6 / 10 identical samples, steps of 2 / 3 / 4 samples (3: one event per 3.0 samples, the densest the detector emits)
    for run in (2, 3, 4, 6, 10):
        c["steps_%d" % run] = _steps(100 + run, run, 6000)
    # square waves (period 2 and 4 give no peak at all: see no_peak_cases)
    for period in (6, 8, 12):
        c["square_%d" % period] = _square(period, 5000)
    # +-full-scale spikes (open pore / saturation) in a normal read
    sp = base.copy()
    for k, at in enumerate(range(1500, n0 - 100, 2111)):
        sp[at:at + 1 + (k % 3)] = 32767 if k % 2 == 0 else -32768
    c["spikes"] = sp
    # slow drift on a normal read (+-2 000 counts over the read)
    c["drift_up"] = base + (np.arange(n0) * 2000) // n0
    c["drift_down"] = base - (np.arange(n0) * 2000) // n0
    # uniform int16 noise
    c["uniform_int16"] = (_splitmix(77, 20000) % np.uint64(65536)).astype(np.int64) - 32768
    # a ramp over the full range with a normal read's texture on top (a bare ramp has a constant t-statistic: no peak)
    ramp = -30000 + (np.arange(n0) * 60000) // n0
    c["ramp_textured"] = ramp + (base - 600)
    # reads around the window lengths (w = 3: n >= 6; w = 6: n >= 12, event_detection.c:76-81)
    tiny = _steps(500, 5, 64)
    for n in (16, 40):
        c["tiny_%d" % n] = tiny[:n]
    # saturation plateaus: clipped at +-full scale for hundreds of samples
    sat = base.copy()
    sat[3000:3400] = 32767; sat[9000:9800] = -32768
    c["saturated_plateaus"] = sat
    return {k: np.clip(v, -32768, 32767).astype(np.int16) for k, v in c.items()}


STALLS50 = ((60011, 300, 655, 0), (150017, 1200, 540, 7), (260003, 5000, 700, 0), (400009, 2200, 610, 11), (520019, 800, 590, 0))


def read50kb_with_stalls(model):
    """ONE 50 kb synthetic read (the headline length) with five stalls of 300-5 000 samples spliced in (three flat, two noisy)"""
    x = normal_read(7050, 50000, model)
    for at, length, level, amp in sorted(STALLS50, reverse=True):
        x = splice(x, at, stall(at, length, level, amp))
    return np.clip(x, -32768, 32767).astype(np.int16)


def no_peak_cases():
    """Signals on which the detector emits NO peak.  The reference then reads peaks[-1] and aborts (create_events :262 ->
    assert(start < nsample), event_detection.c:215); this repo returns the one event [0, n) (documented divergence)."""
    c = {}
    c["flat"] = np.full(3000, 640, np.int16)
    c["saturated_hi"] = np.full(3000, 32767, np.int16)
    c["saturated_lo"] = np.full(2000, -32768, np.int16)
    c["square_2"] = _square(2, 3000)
    c["square_4"] = _square(4, 3000)
    c["shorter_than_a_window"] = _steps(501, 2, 5)
    return c


def mutate(adc, seed):
    """1-3 hostile edits of a read's signal, chosen and placed by a splitmix64 stream: a flat or noisy stall spliced in (50-4 000 samples, at the level found there or
    somewhere else), a burst of full-scale spikes, a saturated plateau, a dropout to the ADC's zero, a linear drift over a stretch, a stretch with three times the noise.
    Returns (int16 signal, list of what was done)."""
    x = np.asarray(adc, np.int64).copy()
    u = _splitmix(seed, 64).astype(np.uint64)
    k = 0

    def draw(n):
        nonlocal k
        v = int(u[k] % np.uint64(n)); k += 1
        return v
    done = []
    for _ in range(1 + draw(3)):
        op = draw(7)
        n = x.shape[0]
        at = 200 + draw(max(1, n - 400))
        if op == 0 or op == 1:
            ln = 50 + draw(3950)
            level = int(np.median(x[at:at + 8])) if draw(2) else 300 + draw(800)
            x = splice(x, at, stall(seed * 31 + k, ln, level, 0 if op == 0 else 3 + draw(12)))
            done.append(("stall_flat" if op == 0 else "stall_noisy", at, ln, level))
        elif op == 2:
            cnt = 1 + draw(6)
            for j in range(cnt):
                p_ = min(n - 4, at + 37 * j)
                x[p_:p_ + 1 + draw(3)] = 32767 if draw(2) else -32768
            done.append(("spikes", at, cnt))
        elif op == 3:
            ln = 20 + draw(900)
            x[at:at + ln] = 32767 if draw(2) else -32768
            done.append(("plateau", at, ln))
        elif op == 4:
            ln = 10 + draw(600)
            x[at:at + ln] = 0
            done.append(("dropout", at, ln))
        elif op == 5:
            ln = min(n - at, 500 + draw(6000))
            amp = (draw(2) * 2 - 1) * (100 + draw(900))
            x[at:at + ln] += (np.arange(ln) * amp) // max(ln, 1)
            x[at + ln:] += amp
            done.append(("drift", at, ln, amp))
        else:
            ln = min(n - at, 300 + draw(3000))
            x[at:at + ln] += 3 * _noise16(seed * 17 + k, ln, 12)
            done.append(("noisy_stretch", at, ln))
    return np.clip(x, -32768, 32767).astype(np.int16), done


# ---- low-complexity REFERENCE sequences (round 6): homopolymer runs of 5-40, di- / tri-nucleotide and longer tandem repeats, two-letter stretches, the odd N, with a signal
# that follows the pore model (geometric dwell, Gaussian noise) -- forward reads, perfect CIGAR.  numpy's Generator is used here (exploration + one pinned test whose
# expectations do not depend on the exact stream: device == oracle on whatever it draws).
ACGT = np.frombuffer(b"ACGT", np.uint8)


def low_complexity(rng, n):
    out = []
    while sum(len(x) for x in out) < n:
        kind = rng.integers(0, 6)
        if kind == 0:
            out.append(ACGT[rng.integers(0, 4, rng.integers(20, 200))])                       # ordinary stretch
        elif kind == 1:
            out.append(np.full(rng.integers(5, 41), ACGT[rng.integers(0, 4)], np.uint8))      # homopolymer
        elif kind == 2:
            out.append(np.tile(ACGT[rng.integers(0, 4, 2)], rng.integers(4, 25)))             # dinucleotide repeat
        elif kind == 3:
            out.append(np.tile(ACGT[rng.integers(0, 4, 3)], rng.integers(4, 20)))             # trinucleotide repeat
        elif kind == 4:
            two = ACGT[rng.choice(4, 2, replace=False)]
            out.append(two[rng.integers(0, 2, rng.integers(20, 120))])                        # two-letter stretch
        else:
            u = np.tile(ACGT[rng.integers(0, 4, rng.integers(5, 12))], rng.integers(2, 6))    # longer tandem repeat
            out.append(u)
    return np.concatenate(out)[:n]


def low_complexity_read(model, seed, n):
    rng = np.random.default_rng(seed)
    from dnascent_amd import synth
    r = synth.make_read(seed, n, model=model)               # carrier for the fields; everything that matters is replaced
    seq = low_complexity(rng, n)
    code = np.zeros(256, np.int64); code[ord("T")] = 1; code[ord("G")] = 2; code[ord("C")] = 3
    c = code[seq]
    rank = np.zeros(n - 8, np.int64)
    for j in range(9):
        rank = rank * 4 + c[j:j + n - 8]
    dwell = 1 + rng.geometric(1.0 / 11.5, n - 8)
    pa = np.repeat(model[rank] * 14.0 + 95.0, dwell) + rng.normal(0, 1.6, int(dwell.sum()))
    r.adc = np.clip(np.rint(pa / 0.1755 + 240.0), -32768, 32767).astype(np.int16)
    r.cal_offset, r.cal_scale = -240.0, 0.1755
    if rng.random() < 0.3:
        seq = seq.copy(); seq[rng.integers(50, n - 50, rng.integers(1, 4))] = ord("N")         # the signal keeps the base that was there
    r.refseq = seq.copy(); r.basecall = seq.copy()
    r.cigar_op = np.array([0], np.uint32); r.cigar_len = np.array([n], np.uint32)
    r.is_reverse = bool(seed % 3 == 1); r.ref_end = r.ref_start + n       # (strand-direction sequences: a reverse read differs only in how the host orients them)
    return r




def big_indel_read(model, seed, n):
    """a read against an iid reference with LARGE indels in its mapping -- deletions and insertions of 6-120 bases, reference skips (CIGAR N), soft clips of up to 150 at both
    ends -- forward or reverse; the signal follows the READ's sequence (geometric dwell, Gaussian noise).  Stresses the reference <-> query maps (htsInterface.cpp:59-157) on
    the host and everything on the device that looks through them (cleaned pairs, eventalign's window bounds and indel scores, --HMM)."""
    from dnascent_amd import synth
    rng = np.random.default_rng(seed)
    r = synth.make_read(seed, max(n, 600), model=model)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    ref = acgt[rng.integers(0, 4, n)]                          # strand direction
    ops, q = [], []                                            # ops in strand direction: (op code, length); BAM codes M0 I1 D2 N3 S4
    head = int(rng.integers(0, 151)) if rng.random() < 0.5 else 0
    if head:
        ops.append((4, head)); q.append(acgt[rng.integers(0, 4, head)])
    pos = 0
    while pos < n:
        m = int(min(n - pos, rng.integers(150, 900)))
        ops.append((0, m)); q.append(ref[pos:pos + m]); pos += m
        if pos >= n - 200:
            if pos < n:
                ops.append((0, n - pos)); q.append(ref[pos:]); pos = n
            break
        kind = rng.integers(0, 3)
        ln = int(rng.integers(6, 121))
        if kind == 0:
            ops.append((1, ln)); q.append(acgt[rng.integers(0, 4, ln)])
        else:
            ln = min(ln, n - pos - 150)
            ops.append((2 if kind == 1 else 3, ln)); pos += ln
    tail = int(rng.integers(0, 151)) if rng.random() < 0.5 else 0
    if tail:
        ops.append((4, tail)); q.append(acgt[rng.integers(0, 4, tail)])
    merged = []
    for o, l in ops:
        if merged and merged[-1][0] == o:
            merged[-1][1] += l
        else:
            merged.append([o, l])
    query = np.concatenate(q)
    code = np.zeros(256, np.int64); code[ord("T")] = 1; code[ord("G")] = 2; code[ord("C")] = 3
    c = code[query]
    nk = query.shape[0] - 8
    rank = np.zeros(nk, np.int64)
    for j in range(9):
        rank = rank * 4 + c[j:j + nk]
    dwell = 1 + rng.geometric(1.0 / 11.5, nk)
    pa = np.repeat(model[rank] * 14.0 + 95.0, dwell) + rng.normal(0, 1.6, int(dwell.sum()))
    r.adc = np.clip(np.rint(pa / 0.1755 + 240.0), -32768, 32767).astype(np.int16)
    r.cal_offset, r.cal_scale = -240.0, 0.1755
    r.is_reverse = bool(rng.integers(0, 2))
    if r.is_reverse:
        merged = merged[::-1]                                  # BAM keeps the CIGAR in reference-forward order (dn_synth.c does the same)
    r.refseq = ref.copy(); r.basecall = query.copy()
    r.cigar_op = np.array([o for o, _ in merged], np.uint32); r.cigar_len = np.array([l for _, l in merged], np.uint32)
    r.ref_end = r.ref_start + n
    return r
