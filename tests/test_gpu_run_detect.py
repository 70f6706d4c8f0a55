"""GPU: the product path end to end across ranks: container in, .detect out.  `python -m dnascent_amd.run_detect` with one
rank, and under torch.distributed.run with two ranks sharing the one GPU (gloo; the 8-GPU node runs the same code over RCCL):
plan (plan_windows) -> length-bucketed batches pulled from a shared counter -> streamed pipeline -> every rank formats its own records and
writes them at their offsets in the shared file (round 5: no writer rank; --central-writer = round 4's gather to rank 0).  The files must be
byte-identical, and equal to a plain single-context run of the same reads."""
import os
import socket
import subprocess
import sys

import pytest

from dnascent_amd import cnn_model, hip, host, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_write_the_same_file_as_one(model, tmp_path):
    reads = [synth.make_read(8800 + i, 2000 + 700 * (i % 5), model=model, is_reverse=bool(i & 1), sub_rate=0.002, ins_rate=0.001, del_rate=0.001,
                             noise_pa=6.5 if i == 4 else 1.6) for i in range(14)]
    cont = str(tmp_path / "reads.dnrc")
    host.write_container(cont, reads)
    assert host.container_sizes(cont).tolist() == [r.n_samples() for r in reads]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    one = str(tmp_path / "one.detect"); two = str(tmp_path / "two.detect")
    common = ["--container", cont, "--batch-samples", "60000", "--inflight", "2", "--window-batches", "2", "--header", "#hdr\n"]
    r = subprocess.run([sys.executable, "-m", "dnascent_amd.run_detect", "--out", one] + common, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "dnascent_amd.run_detect", "--out", two, "--backend", "gloo"] + common,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, b = open(one, "rb").read(), open(two, "rb").read()
    assert a == b and a.startswith(b"#hdr\n") and a.count(b">") == 13                  # the noisy read fails the QC and is not written
    # round 4's form (packed calls gathered to rank 0, formatted there) stays selectable and writes the same bytes
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cen = str(tmp_path / "central.detect")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "dnascent_amd.run_detect", "--out", cen, "--backend", "gloo", "--central-writer"] + common,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert open(cen, "rb").read() == a
    # and both equal one plain batch on one context, records in input order
    ctx = hip.Context(0); ctx.load_pore_model(model, 0.14)
    desc, blob, _ = cnn_model.default_model(); ctx.load_cnn(desc, blob)
    bt = host.ReadBatch()
    for rd in reads:
        assert bt.add_synth(rd) >= 0
    bt.upload(ctx); ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    ref = str(tmp_path / "ref.detect")
    assert bt.detect_write(ctx, ref, header="#hdr\n") == 13
    assert open(ref, "rb").read() == a
    ctx.close()


def test_inflight_one_loses_nothing(model, tmp_path):
    """`run_detect --inflight 1` (round-3 advisor: the prefetched batch's window used to be gathered without it) writes the file the default depth writes"""
    reads = [synth.make_read(8900 + i, 2000 + 500 * (i % 4), model=model, is_reverse=bool(i & 1), sub_rate=0.002) for i in range(12)]
    cont = str(tmp_path / "reads.dnrc")
    host.write_container(cont, reads)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    outs = []
    for depth in ("1", "3"):
        out = str(tmp_path / ("d%s.detect" % depth)); outs.append(out)
        r = subprocess.run([sys.executable, "-m", "dnascent_amd.run_detect", "--container", cont, "--out", out, "--batch-samples", "50000", "--inflight", depth,
                            "--window-batches", "2"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, b = open(outs[0], "rb").read(), open(outs[1], "rb").read()
    assert a == b and a.count(b">") == 12


def test_bench_two_ranks_gloo():
    """bench.py's N > 1 path -- static shard, per-rank streams, gather_calls / gather_stats / reduce_max inside and after the timed region -- on the
    one GPU with the gloo backend (DN_BENCH_BACKEND): what the driver's 8-GPU SCALE command runs over RCCL must not execute for the first time there"""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), DN_BENCH_BACKEND="gloo", DN_CNN_ROWS=str(1 << 20))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reads-per-step", "40", "--bases", "5000", "--inflight", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 3 and d["scaling"] == "weak"
    assert len(d["ranks"]["per_rank"]) == 2 and d["ranks"]["gather_s_max"] >= 0.0 and "gather_s" in d["host"]
    assert d["config"]["samples_per_gpu"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" not in d          # the CPU leg runs at N = 1 only
    # round-4 verdict item 6: K1's own roofline entry beside the banded one; at N > 1 a rank cycles through inflight + 2 distinct batches (4 steps here:
    # all distinct) and says what generating them cost
    assert d["roofline_k1"]["frac"] > 0 and d["roofline_k1"]["algorithmic_bytes_per_launch"] > 0 and 2.0 < d["roofline_k1"]["bytes_per_sample"] < 6.0
    assert all(p["datagen_s"] > 0 and p["distinct_batches"] == 4 for p in d["ranks"]["per_rank"])


def test_hostile_reads_through_the_product_driver(model, tmp_path):
    """Round 6: the reads that broke something on the way -- a stalled read whose eventalign window exceeds both LDS lattices, the dense-event read that overflows the
    drivers' event workspaces (retried inside DetectStream), a 400 kb read (more than 4 096 detector chunks: it used to fail its whole batch), flat and saturated signals,
    a 16-sample read -- mixed with ordinary reads in one container, through `python -m dnascent_amd.run_detect` at its default event bound: exit code 0, the file equals a
    plain single-context run at the detector's own bound, the passing reads are exactly those the oracle passes, in input order."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, "tests")); _sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import adversarial_signals as adv
    import pyoracle as po
    import test_segmentation_adversarial as tsa
    sigs = adv.cases(model)
    nop = adv.no_peak_cases()
    stalled = synth.make_read(7001, 2500, model=model); stalled.adc = sigs["stall6000_noisy"]; stalled.cal_offset, stalled.cal_scale = adv.CAL; stalled.read_id = "stalled"
    reads = [synth.make_read(9901, 3000, model=model), stalled, tsa._dense_read(model, 9902, 4000), tsa._carrier(model, 9903, nop["flat"]),
             synth.make_read(9904, 2500, model=model, is_reverse=True), tsa._carrier(model, 9905, nop["saturated_hi"]), tsa._carrier(model, 9906, sigs["tiny_16"]),
             synth.make_read(8400, 400000, model=model, sub_rate=0.002, ins_rate=0.001, del_rate=0.001), tsa._carrier(model, 9907, sigs["spikes"]),
             synth.make_read(9908, 3500, model=model)]
    for k, r in enumerate(reads):
        r.read_id = "hostile-%02d" % k
    cont = str(tmp_path / "hostile.dnrc")
    host.write_container(cont, reads)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = str(tmp_path / "hostile.detect")
    r = subprocess.run([_sys.executable, "-m", "dnascent_amd.run_detect", "--container", cont, "--out", out, "--inflight", "2", "--header", "#hdr\n"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = open(out, "rb").read()
    want_ids = []
    for q in reads:
        o = po.OracleRead(q, model)
        if o.normalise() == 0 and o.eventalign() == 0:
            want_ids.append(q.read_id)
        o.free()
    ids = [l.split()[0][1:].decode() for l in got.split(b"\n") if l.startswith(b">")]
    assert ids == want_ids and ids == ["hostile-00", "hostile-01", "hostile-02", "hostile-04", "hostile-07", "hostile-09"]
    ctx = hip.Context(0); ctx.load_pore_model(model, 0.14)
    desc, blob, _ = cnn_model.default_model(); ctx.load_cnn(desc, blob)
    bt = host.ReadBatch()
    for rd in reads:
        assert bt.add_synth(rd) >= 0
    bt.upload(ctx); ctx.run("normalise"); ctx.run("eventalign"); ctx.sync()
    ref = str(tmp_path / "ref.detect")
    assert bt.detect_write(ctx, ref, header="#hdr\n") == 6
    assert open(ref, "rb").read() == got
    ctx.close()
