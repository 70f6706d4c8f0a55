"""GPU: the PRODUCT driver timed at BASELINE configs[2] size (round-3 verdict item 3): a binary read container of N synthetic 50 kb reads is
written to --dir (11.5 GB at N = 10 000), then `python -m dnascent_amd.run_detect` ingests it (index, plan, loader thread, DetectStream, packed
gather to the gather thread, C++ formatter, ordered write) into a .detect file.  Prints run_detect's own summary line and writes its --stats JSON.
    python tools/time_run_detect.py --reads 10000 --stats gpurun_out/run_detect_stats.json"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10000)
ap.add_argument("--bases", type=int, default=50000)
ap.add_argument("--dir", default="/tmp")
ap.add_argument("--inflight", type=int, default=5)
ap.add_argument("--stats", default=None)
ap.add_argument("--keep", action="store_true")
ap.add_argument("--reuse", action="store_true", help="do not rewrite the container if --dir already holds one (repeated timings in one session; implies --keep)")
ap.add_argument("--ranks", type=int, default=1, help="> 1: under torch.distributed.run with the gloo backend, the ranks SHARING this GPU (what the 8-GPU node runs over RCCL, "
                                                     "exercised at full payload size; not a throughput figure)")
ap.add_argument("--rccl-group-of-one", action="store_true", help="ONE rank under torch.distributed.run with the nccl backend (DN_RUN_DETECT_FORCE_DIST=1): RCCL executed on a 1-GPU box")
ap.add_argument("--sha", action="store_true", help="print the SHA-256 of the .detect file (1- and N-rank runs must agree)")
a, extra = ap.parse_known_args()
from dnascent_amd import host, synth
cont = os.path.join(a.dir, "bench_reads.dnrc"); out = os.path.join(a.dir, "bench_reads.detect")
t0 = time.time()
if a.reuse and os.path.exists(cont) and host.container_count(cont) == a.reads:
    print("container: reused (%d reads, %.2f GB)" % (a.reads, os.path.getsize(cont) / 1e9), flush=True)
else:
    n = host.write_synth_container(cont, synth.pore_model(), 1000003, a.reads, a.bases)
    print("container: %d reads, %.2f GB, written in %.1f s" % (n, os.path.getsize(cont) / 1e9, time.time() - t0), flush=True)
env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))       # DN_CNN_ROWS: run_detect's own default unless the caller exports one
tail = ["--container", cont, "--out", out, "--inflight", str(a.inflight)] + (["--stats", a.stats] if a.stats else []) + extra
if a.ranks > 1:
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.ranks), "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "dnascent_amd.run_detect", "--backend", "gloo"] + tail
elif a.rccl_group_of_one:
    env["DN_RUN_DETECT_FORCE_DIST"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29577",
           "-m", "dnascent_amd.run_detect", "--backend", "nccl"] + tail
else:
    cmd = [sys.executable, "-m", "dnascent_amd.run_detect"] + tail
t0 = time.time()
r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True)
print(r.stdout[-3000:]); print(r.stderr[-3000:], file=sys.stderr)
print("run_detect wall (process start to exit, incl. library load, context + CNN-lane allocation): %.1f s, rc %d, output %.2f GB" % (
    time.time() - t0, r.returncode, os.path.getsize(out) / 1e9 if os.path.exists(out) else 0.0))
if a.sha and os.path.exists(out):
    import hashlib
    h = hashlib.sha256()
    with open(out, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    print("sha256 %s  %d bytes  (%d rank(s))" % (h.hexdigest(), os.path.getsize(out), a.ranks))
if not a.keep and not a.reuse:
    for f in (cont, out):
        if os.path.exists(f):
            os.unlink(f)
elif os.path.exists(out) and not a.keep:
    os.unlink(out)
sys.exit(r.returncode)
