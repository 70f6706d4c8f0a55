#!/usr/bin/env python3
"""bench.py -- raw-signal Msamples/s of the `detect` hot path on MI355X (BASELINE.json metric).

Default workload (BASELINE.json configs[2]): a stream of synthetic 50 kb R10.4.1 reads through the FULL pipeline on one
MI355X.  One STEP = one batch of --reads-per-step reads (default 500): host -> HBM upload, normaliseEvents (segmentation,
rough scaling, adaptive banded alignment + backtrack + QC, Theil-Sen), eventalign (windowed Viterbi + feature fill), the
BrdU/EdU CNN, the bulk result back on the host (dn_collect) and -- unless --emit 0 -- the .detect records formatted and
written.  `--steps 20` (the default) is exactly the 10 000 reads of configs[2].  The timed region starts before the first
upload of the first timed batch and ends when the last batch's records are on the host ("first H2D to last
result on host"); every batch is distinct (own seeds), nothing is resident beforehand and nothing is skipped.  ONE host
thread drives --inflight contexts (DNAscent::streamDetect).  For N > 1 every rank owns its own stream of reads (reads shard
with no data-path collective: weak scaling) and the per-call results {coordinate, P(EdU), P(BrdU)} of all ranks are
gathered to rank 0 inside the timed region; value = samples of all ranks / max-over-ranks time.

`--scope banded` is BASELINE.json configs[1] (1 000 x 20 kb, normaliseEvents only, batch resident in HBM, CNN stubbed): the
scope the adaptive-banded kernel's HBM roofline is quoted on.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel of the run: the CNN (all layers of one batch = one
"launch" of dn_run_cnn) against the dense fp16 MFMA peak in the full scope, k2_fill against the HBM peak in the banded
scope; achieved = ALGORITHMIC flops / bytes per launch / mean launch duration from HIP events on the
library's streams inside the timed region.  `cpu_baseline` times the oracle (CPU restatement, kind "port") with OpenMP
schedule(dynamic) on all host cores over a bounded sample of the same reads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Every in-flight batch has its own HIP stream; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and
# streams that share a queue serialise.  Must be set before the HIP runtime initialises (torch or libdnascent_hip).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK = 2500.0         # dense fp16 / bf16 MFMA TFLOP/s (MI355X_MICROARCH.md; the 2:1-sparsity figure is not used)
MFMA_F32_PEAK = 157.0


def cnn_macs(desc):
    return sum(o["k"] * o["cin"] * o["cout"] for o in desc["ops"] if o["op"] == "conv") + \
           sum(o["k"] * o["c"] for o in desc["ops"] if o["op"] == "dwconv") + 47040 + 64 * 3


def cpu_baseline(model, n_bases, seed0, full, budget_reads):
    """The oracle (CPU restatement of the reference path) with OpenMP, one read per thread, schedule(dynamic) -- the shape of the
    reference's own loop (detect.cpp:852) -- over a bounded sample of the workload's reads.  full: + eventalign, and the CNN's
    cost from the stock-PyTorch CPU rendering of the same model description (fp32, all cores) on a sample of the positions."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from dnascent_amd import synth
    cores = os.cpu_count() or 1
    n = max(1, min(budget_reads, cores))
    reads = [synth.make_read(seed0 + i, n_bases, model=model, is_reverse=bool(i & 1), sub_rate=0.002, ins_rate=0.001, del_rate=0.001)
             for i in range(n)]
    secs, samples, positions, ok = po.bench_reads(reads, model, full, cores)
    what = "oracle normaliseEvents%s, OpenMP schedule(dynamic), %d threads: %.1f s" % (" + eventalign" if full else "", cores, secs)
    cnn_s = 0.0
    if full and positions:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import cnn_torch_ref
        import torch
        from dnascent_amd import cnn_model
        # one read per thread like the reference (one TF_SessionRun per read from every OpenMP thread, detect.cpp:653): the
        # rendering is timed on ONE thread and credited with perfect scaling over the cores -- generous to the CPU (a 256-thread
        # intra-op run of these small convolutions is two orders of magnitude slower than that)
        torch.set_num_threads(1)
        o = po.OracleRead(reads[0], model)
        assert o.normalise() == 0 and o.eventalign() == 0
        pos = o.positions()
        o.free()
        k = min(len(pos["core"]), 20000)
        ref = cnn_model.default_model()[2]
        t0 = time.time()
        reps = 0
        while time.time() - t0 < 4.0 or reps < 2:
            cnn_torch_ref.run(ref, pos["core"][:k], pos["residual"][:k], pos["signal"][:k])
            reps += 1
        pos_per_s = reps * k / (time.time() - t0)
        cnn_s = positions / (pos_per_s * cores)
        what += "; CNN: PyTorch CPU fp32 rendering, %.0f positions/s on one thread, credited x %d cores: %d positions of the sample -> %.1f s" % (
            pos_per_s, cores, positions, cnn_s)
    return {"value": samples / (secs + cnn_s) / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d-base reads of the workload (%d pass QC); %s" % (n, n_bases, ok, what)}


def _hbm_info():
    """device memory in use at the end of the run (every context, every CNN lane allocated): hipMemGetInfo through ctypes"""
    import ctypes
    try:
        rt = ctypes.CDLL("libamdhip64.so")
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        if rt.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) != 0:
            return None
        return {"used_GB": round((total.value - free.value) / 1e9, 1), "total_GB": round(total.value / 1e9, 1)}
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scope", choices=["full", "banded"], default="full",
                    help="full = BASELINE configs[2] (default): streamed 50 kb reads, whole pipeline, H2D to results on host; "
                         "banded = configs[1]: 1 000 x 20 kb resident batch, normaliseEvents only (CNN stubbed)")
    ap.add_argument("--reads-per-step", type=int, default=None, help="reads per batch (default 500 full / 1000 banded)")
    ap.add_argument("--bases", type=int, default=None, help="bases per read (default 50000 full / 20000 banded)")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight per GPU, each on its own context / stream / workspace (default 6 full / 8 banded)")
    ap.add_argument("--emit", type=int, default=1, help="full scope: format + write the .detect records inside the timed region")
    ap.add_argument("--out", default=None, help="full scope: .detect output path (default: formatted and counted, not written)")
    ap.add_argument("--cnn-math", choices=["f16x3", "bf16x6", "fp32"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    full = args.scope == "full"
    rps = args.reads_per_step or (500 if full else 1000)
    bases = args.bases or (50000 if full else 20000)
    inflight = args.inflight or (6 if full else 8)         # full: 3 -> 548, 4 -> 587, 6 -> 617 Msamples/s in one session; 8 does not fit beside the CNN lanes
    os.environ.setdefault("DN_CNN_ROWS", str(4 << 20))      # activation rows resident per CNN pass and lane: 4 Mi rows = 16 GiB (2 Mi: -6 %, 8 Mi: -3 %)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1:
        # the host side (read generation, record formatting) is OpenMP over reads: the ranks of a node share its cores
        # (torch.distributed.run exports OMP_NUM_THREADS=1 for its workers: the per-rank share replaces that launcher default)
        if os.environ.get("OMP_NUM_THREADS", "1") == "1":
            os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // world))
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # DN_BENCH_BACKEND=gloo (+ ranks sharing a device) exists only to exercise the N > 1 code path on a 1-GPU box
        backend = os.environ.get("DN_BENCH_BACKEND", "nccl")
        ndev = max(1, torch.cuda.device_count())
        if backend == "nccl":
            torch.cuda.set_device(local_rank % ndev)
        dist.init_process_group(backend, rank=rank, world_size=world)

    from dnascent_amd import cnn_model, hip, host, shard, synth
    model = synth.pore_model()
    dev = (local_rank % max(1, hip.lib().dn_device_count())) if world > 1 else 0
    red_dev = "cuda" if (dist is not None and os.environ.get("DN_BENCH_BACKEND", "nccl") == "nccl") else "cpu"
    cnn_desc, cnn_blob, _ = cnn_model.default_model()
    n_batches = (args.warmup + args.steps) if full else 1
    nctx = max(1, min(inflight, args.steps if not full else n_batches))
    ctxs = [hip.Context(dev) for _ in range(nctx)]
    for c in ctxs:
        c.load_pore_model(model, 0.14)
        if full:
            c.load_cnn(cnn_desc, cnn_blob)
            if args.cnn_math:
                c.cnn_set_math(args.cnn_math)
    cnn_math = args.cnn_math or os.environ.get("DN_CNN_MATH", "f16x3")

    # ---- the reads: every batch of the stream is distinct (seeds by rank / batch / read) ----
    t_gen = time.perf_counter()
    seed_base = 1000003 * (rank + 1)
    batches = []
    for b in range(n_batches):
        B = host.ReadBatch()
        got = B.fill_synth(model, seed_base + b * rps, rps, bases)
        assert got == rps, (got, rps)
        batches.append(B)
    t_gen = time.perf_counter() - t_gen

    def barrier():
        for c in ctxs:
            c.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    gather_s = 0.0
    if full:
        warm, timed = batches[:args.warmup], batches[args.warmup:]
        # set-up, not a step: every context the warm-up batches will not reach gets its workspace now (a first upload is a
        # 10+ GB hipMalloc; with --warmup 2 and 4 contexts two of those used to land inside the timed region: 513 against 605)
        for c in ctxs[len(warm):]:
            batches[0].upload(c)
            c.sync()
        if warm:
            host.stream_detect(ctxs, warm, emit=bool(args.emit), out_path=None)
        barrier()
        for c in ctxs:
            c.profile(True)
            c.profile_reset()
        t0 = time.perf_counter()
        if dist is not None:
            # the path's only exchange: the binary per-call results of every rank to the writer rank, one grouped
            # send / recv over RCCL, inside the timed region
            st, kept = host.stream_detect(ctxs, timed, emit=bool(args.emit), out_path=args.out, keep=True)
            tg = time.perf_counter()
            got = shard.gather_calls(dist, kept["read_calls"], kept["coord"], kept["p_edu"], kept["p_brdu"], dst=0, device=red_dev)
            if rank == 0:
                assert len(got) == world and all(int(g[0].sum()) == g[1].shape[0] for g in got)
            gather_s = time.perf_counter() - tg
            torch.cuda.synchronize()
        else:
            st = host.stream_detect(ctxs, timed, emit=bool(args.emit), out_path=args.out)
        dt = time.perf_counter() - t0
        barrier()
        samples_total = float(st.samples)
        last = ctxs[(len(timed) - 1) % nctx]
        last.n_reads = timed[-1].size()
        summ = last.summaries()
    else:
        B = batches[0]
        for c in ctxs:
            B.upload(c)                    # configs[1]: inputs resident in HBM before the timed region (one copy per in-flight slot)
        samples_step = B.samples()

        def run_steps(k):
            for i in range(k):
                c = ctxs[i % nctx]
                if i >= nctx:
                    c.sync()               # the slot's previous step; every dn_run_* only enqueues
                c.run("normalise")
            for c in ctxs:
                c.sync()
        run_steps(max(args.warmup, nctx if args.warmup else 0))
        barrier()
        for c in ctxs:
            c.profile(True)
            c.profile_reset()
        t0 = time.perf_counter()
        run_steps(args.steps)
        if dist is not None:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        samples_total = float(samples_step) * args.steps
        st = None
        summ = ctxs[0].summaries()
        for j, c in enumerate(ctxs[1:], 1):       # every slot ran the same batch concurrently: results must agree bit for bit
            if c.summaries().tobytes() != summ.tobytes():
                raise SystemExit("bench: in-flight slot %d disagrees with slot 0 on the per-read results" % j)

    prof = {}
    for c in ctxs:
        for k, v in c.profile_get().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
        c.profile(False)

    # the same kernels with the GPU to itself (outside the timed region): one batch, nothing else in flight
    solo = {}
    if rank == 0:
        c = ctxs[0]
        c.profile(True); c.profile_reset()
        sb = batches[-1]
        sb.upload(c)
        c.run("detect" if full else "normalise")
        c.sync()
        if full:
            c.collect()
        for k, v in c.profile_get().items():
            if v[1]:
                solo[k] = v[0] / v[1]
        c.profile(False)
        summ = c.summaries()

    # the only collectives of the path: MAX of the elapsed time, SUM of the counters (dnascent_amd/shard.py)
    if dist is not None:
        dt = shard.reduce_max(dist, dt, device=red_dev)
        samples_total = shard.reduce_counters(dist, [samples_total], device=red_dev)[0]

    if rank == 0:
        ok = summ["status"] != 5
        alg_bytes = float(np.sum(summ["n_bands"][ok].astype(np.float64) * 100.0 +
                                 4.0 * (summ["n_events"][ok].astype(np.float64) + summ["n_kmers_query"][ok]) +
                                 9.0 * summ["n_aligned"][ok]))           # trace + inputs + backtrack, per launch (= one batch)
        fill_ms, fill_n = prof.get("k2_fill", (0.0, 0))
        fill_s = (fill_ms / max(fill_n, 1)) / 1e3
        roof_banded = {"bound": "hbm", "kernel": "k2_fill", "achieved": alg_bytes / fill_s / 1e9 if fill_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": (alg_bytes / fill_s / 1e9 / HBM_PEAK_GBS) if fill_s > 0 else 0.0, "traffic": None,
                       "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_ms": fill_ms / max(fill_n, 1)}
        if "k2_fill" in solo:
            roof_banded["solo_launch_ms"] = solo["k2_fill"]
            roof_banded["solo_frac"] = alg_bytes / (solo["k2_fill"] / 1e3) / 1e9 / HBM_PEAK_GBS
        pm_path = os.path.join(ROOT, "profiles", "r02_pmc_k2_fill.json")
        if os.path.exists(pm_path):          # HBM bytes per launch from the committed PMC passes (cannot be collected inside a timed run)
            pm = json.load(open(pm_path))
            if pm.get("workload") == {"reads": rps, "bases": bases}:
                roof_banded["traffic"] = pm["write_bytes"] + pm["fetch_bytes_corrected"]
        out = {
            "metric": "raw-signal Msamples/sec (whole node) on `detect`" + ("" if full else " -- banded-HMM scope only (configs[1])"),
            "value": samples_total / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32/f64 (CNN: fp32 products as %s on the 16-bit matrix cores)" % cnn_math) if full else "f32/f64",
            "data": "synthetic",
            "kernel_ms_per_launch": {k: v[0] / v[1] for k, v in prof.items() if v[1]},
            "kernel_ms_solo": solo,
        }
        if full:
            pos = float(np.sum(summ["n_positions"][summ["status"] == 0]))
            mac = cnn_macs(cnn_desc)
            cnn_ms, cnn_n = prof.get("k3_cnn", (0.0, 0))
            flops = 2.0 * mac * pos
            peak = MFMA_F32_PEAK if cnn_math == "fp32" else MFMA_F16_PEAK
            ach = flops / ((cnn_ms / max(cnn_n, 1)) / 1e3) / 1e12 if cnn_ms > 0 else 0.0
            issued = {"f16x3": 3.0, "bf16x6": 6.0, "fp32": 1.0}[cnn_math]
            out["config"] = {
                "workload": "%d synthetic %d kb R10.4.1 reads per GPU (BASELINE configs[2] = 10 000 at --steps 20), full pipeline: upload, "
                            "normaliseEvents, eventalign, CNN (%s), results on host%s" % (args.steps * rps, bases // 1000, cnn_math,
                                                                                          ", .detect records formatted" + (" and written" if args.out else "") if args.emit else ""),
                "reads_per_step": rps, "bases_per_read": bases, "reads_per_gpu": args.steps * rps, "samples_per_gpu": int(st.samples),
                "reads_passing_qc_per_gpu": int(st.reads_ok), "calls_per_gpu": int(st.calls),
                "parallelism": "reads sharded, %d rank(s), %d batches in flight per GPU, one host thread per rank" % (world, nctx)}
            # ---- the two largest kernels of the run (profiles/r02_kernel_stats.csv): k3_sep_split<128, 9> (the eleven 9-tap separable
            #      layers 128 -> 128) and k3_sep_ws<256, 17> (the six 17-tap ones -> 256 channels).  HIP events bracket EVERY launch of
            #      both on the CNN lane's stream; the one with the larger summed time is `roofline`, the other `roofline_second`.
            #      A fused separable layer does 2 (k cin + cin cout) flops per position for 4 (cin + cout) bytes of activation I/O:
            #      34 flop/B (9 x 128 -> 128) and 68 flop/B (17 x 256 -> 256) against a ridge of 104 at the f16x3 rate (833 TFLOP/s
            #      over 8 TB/s) -- both sit on the HBM side of the roofline, so `bound` is "hbm" and `achieved` the layer I/O per
            #      second (PMC: HBM traffic = 1.03-1.05 x that, profiles/r02_pmc_k3_traffic.csv); the matrix-core view is kept beside it.
            ops = cnn_desc["ops"]
            def sep_roofline(name, label, pairs):
                ms, n = prof.get(name, (0.0, 0))
                if not n or cnn_math != "f16x3":
                    return None, 0.0
                fl = 2.0 * sum(d["k"] * d["c"] + p["cin"] * p["cout"] for d, p in pairs) * float(st.positions)
                by = 4.0 * sum(p["cin"] + p["cout"] + (p["cout"] if p.get("add", -1) >= 0 else 0) for d, p in pairs) * float(st.positions)   # + the residual read
                gbs = by / (ms / 1e3) / 1e9
                tf = fl / (ms / 1e3) / 1e12
                r = {"bound": "hbm", "kernel": label, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": None, "launches": n, "mean_launch_ms": ms / n, "algorithmic_bytes_per_launch": by / n,
                        "algorithmic_flops_per_launch": fl / n, "mfma_TFLOPs": tf, "mfma_frac": tf / MFMA_F16_PEAK, "mfma_issued_frac": 3.0 * tf / MFMA_F16_PEAK,
                        "note": "achieved = activation I/O of the layer (4 B x (cin + cout [+ cout residual]) x positions) / launch time, HIP events around every launch "
                                "in the timed region; a launch shares the chip with the other CNN lanes and the per-read stages of the batches in "
                                "flight; mfma_*: the same launches as algorithmic fp32 flops against the dense fp16 MFMA peak (x 3 issued); "
                                "solo_*: the same kernel in one batch that has the chip to itself (untimed pass after the run)"}
                if name in solo and solo[name] > 0:
                    r["solo_launch_ms"] = solo[name]
                    r["solo_frac"] = (by / n) / (solo[name] / 1e3) / 1e9 / HBM_PEAK_GBS
                return r, ms
            ws_pairs = [(o, ops[i + 1]) for i, o in enumerate(ops) if o["op"] == "dwconv" and o["k"] == 17 and ops[i + 1]["cout"] == 256]
            s9_pairs = [(o, ops[i + 1]) for i, o in enumerate(ops) if o["op"] == "dwconv" and o["k"] == 9 and o["c"] == 128 and ops[i + 1]["cout"] == 128]
            r_ws, t_ws = sep_roofline("k3_sep_ws", "k3_sep_ws<256, 17> (SeparableConv1D 17 taps -> 256 channels, depthwise fused into the pointwise GEMM, persistent)", ws_pairs)
            r_s9, t_s9 = sep_roofline("k3_sep9", "k3_sep_split<128, 9> (SeparableConv1D 9 taps 128 -> 128 channels, depthwise fused into the pointwise GEMM)", s9_pairs)
            first, second = (r_s9, r_ws) if t_s9 >= t_ws else (r_ws, r_s9)
            if first:
                out["roofline"] = first
            if second:
                out["roofline_second"] = second
            roof_net = {"bound": "mfma", "kernel": "k3_cnn (all layers of one batch = one dn_run_cnn)", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": None, "algorithmic_flops_per_launch": flops, "mean_launch_ms": cnn_ms / max(cnn_n, 1),
                        "issued_frac": ach * issued / peak,
                        "note": "achieved = algorithmic fp32 flops (2 x MACs x positions) / launch time; every fp32 product is issued as %g "
                                "16-bit MFMA products (issued_frac counts those)" % issued}
            if "k3_cnn" in solo:
                roof_net["solo_launch_ms"] = solo["k3_cnn"]
                roof_net["solo_frac"] = flops / (solo["k3_cnn"] / 1e3) / 1e12 / peak
            out["roofline_network"] = roof_net
            if "roofline" not in out:
                out["roofline"] = roof_net
            out["roofline_banded"] = roof_banded
            out["hbm"] = _hbm_info()
            out["host"] = {"datagen_s": t_gen, "upload_s": st.seconds_upload, "collect_wait_s": st.seconds_collect, "emit_s": st.seconds_emit,
                           "gather_s": gather_s, "emission": {"records_per_s": st.calls / st.seconds_emit if st.seconds_emit > 0 else None,
                                                              "MB_per_s": st.bytes_out / st.seconds_emit / 1e6 if st.seconds_emit > 0 else None,
                                                              "bytes": int(st.bytes_out)}}
        else:
            out["config"] = {"workload": "%d synthetic %d kb R10.4.1 reads per GPU, banded-HMM scope (segmentation + rough scaling + adaptive "
                                         "banded alignment + backtrack/QC + Theil-Sen), batch resident in HBM, CNN stubbed" % (rps, bases // 1000),
                             "reads_per_gpu": rps, "bases_per_read": bases, "samples_per_gpu_step": int(samples_step),
                             "reads_passing_qc": int(np.sum(summ["status"] == 0)),
                             "parallelism": "reads sharded, %d rank(s), %d batches in flight per GPU, one host thread per rank" % (world, nctx)}
            out["roofline"] = roof_banded
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, bases, seed_base, full, budget_reads=256)
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
