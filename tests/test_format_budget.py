"""The per-rank CPU budget of an 8-GPU run (round-5 verdict, weak 15): with no writer rank every rank formats and writes ITS OWN records on its share of the
host's cores -- usable_cpus() // 8, i.e. two threads under the GPU pool's 16-CPU quota.  One rank at the bench's rate (BENCH_r05: 877.8 Msamples/s, 3.95 GB of
.detect text per 5.75 G samples) produces 0.60 GB/s of text; this test times DNAscent::formatPacked + dnh_pwrite_scatter on a full 500 x 50 kb batch's calls
(11 600 thymidine calls per read, 16 bytes each packed, 196 MB of text) at exactly that thread count, in a process of its own (the library reads DN_HOST_THREADS
once), and asserts 1.5 x the rank's rate.  Round 6 made it hold: 0.65 GB/s at two threads before (two digits per table lookup, cvtsd2si instead of rint(),
three bases per lookup in either orientation, recycled text buffers: 1.4 GB/s on the build container's 2.1 GHz Xeon)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANK_MSAMPLES_PER_S = 885.0            # one GPU's rate (profiles/r06_*: 878-886)
TEXT_BYTES_PER_SAMPLE = 3.95e9 / 5.749e9

CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from dnascent_amd import host
n_reads, k = 500, 11600
rng = np.random.default_rng(1)
hdrs = [(">synth-%%016x chrSynth %%d %%d %%s\n" %% (i, 1000, 51000, "rev" if i & 1 else "fwd")).encode() for i in range(n_reads)]
calls = np.zeros((n_reads, k, 4), np.uint32)
calls[:, :, 0] = np.sort(rng.integers(1000, 51000, (n_reads, k)), axis=1)
calls[:, :, 1] = rng.random((n_reads, k)).astype(np.float32).view(np.uint32)
calls[:, :, 2] = rng.random((n_reads, k)).astype(np.float32).view(np.uint32)
km = rng.integers(0, 4, (n_reads, k, 9)).astype(np.uint32)
calls[:, :, 3] = sum(km[:, :, z] << (3 * z) for z in range(9))
parts = []
for i in range(n_reads):
    parts.append(np.frombuffer(hdrs[i], np.uint8)); parts.append(calls[i].reshape(-1).view(np.uint8))
pay = np.concatenate(parts)
meta3 = np.array([[k, len(h), i & 1] for i, h in enumerate(hdrs)], np.uint64)          # flag 1 = DN_PACK_REVERSE
sz = np.array([len(h) + 16 * k for h in hdrs], np.uint64)
ptr = np.uint64(pay.ctypes.data) + np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64)
fd = os.open(sys.argv[1], os.O_RDWR | os.O_CREAT | os.O_TRUNC)
best, nbytes = 1e9, 0
for it in range(7):                      # best of six after the first (a loaded host: the budget is about what the code can do, not about the neighbours)
    t0 = time.perf_counter()
    text, rb = host.format_packed(meta3, ptr)
    src = np.concatenate([[0], np.cumsum(rb)[:-1]]).astype(np.uint64)
    host.pwrite_scatter(fd, text, src, rb, src)
    dt = time.perf_counter() - t0
    nbytes = len(text)
    if it:
        best = min(best, dt)               # the first round pays the buffers' page faults once per process
    del text
os.close(fd)
first = open(sys.argv[1], "rb").read(200).split(b"\n")
print(json.dumps(dict(threads=host.host_threads(), usable=host.usable_cpus(), seconds=best, text_bytes=nbytes, first_lines=[l.decode() for l in first[:2]],
                      file_bytes=os.path.getsize(sys.argv[1]))))
"""


def test_formatter_holds_one_gpus_record_rate_on_a_ranks_share_of_the_cores(tmp_path):
    sys.path.insert(0, ROOT)
    from dnascent_amd import host
    threads = max(2, host.usable_cpus() // 8)
    env = dict(os.environ, DN_HOST_THREADS=str(threads), OMP_WAIT_POLICY="passive")
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), str(tmp_path / "budget.detect")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["threads"] == threads and r["file_bytes"] == r["text_bytes"] > 190e6
    assert r["first_lines"][0].startswith(">synth-0000000000000000 chrSynth 1000 51000 fwd") and r["first_lines"][1].count("\t") == 3
    rate = r["text_bytes"] / r["seconds"] / 1e9
    need = RANK_MSAMPLES_PER_S * 1e6 * TEXT_BYTES_PER_SAMPLE / 1e9
    print("formatPacked + pwrite_scatter at %d threads (usable CPUs %d): %.2f GB/s of .detect text = %.2f x one rank's %.2f GB/s at %.0f Msamples/s" % (
        threads, r["usable"], rate, rate / need, need, RANK_MSAMPLES_PER_S))
    assert rate >= 1.5 * need, (rate, need)
