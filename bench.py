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

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel FAMILY of the run, found from the run itself: HIP events
bracket every launch of every layer of the network (dn_profile_get_layer) and every per-read stage on the streams they run on; the
family with the largest summed time inside the timed region is `roofline`, all of them are listed in `roofline_families`.  achieved =
ALGORITHMIC flops / bytes (per-position figures x the positions the launches processed) / summed launch time; `bound`
follows from the family's arithmetic intensity against the ridge of the arithmetic in use.  `traffic` (HBM bytes per launch, PMC) and
`roofline_chip` (whole-chip HBM / MFMA fractions of a step) come from the committed counter passes of tools/r03_profile.sh when they
were taken at this workload's shape -- counters cannot be collected inside a timed run.  k2_fill against the HBM peak is
`roofline_banded` (and `roofline` in the banded scope).  `cpu_baseline` times the oracle (CPU restatement, kind "port") with OpenMP
schedule(dynamic) on all host cores over a bounded sample of the same reads, several reads per thread.
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
# the host library's OpenMP teams must SLEEP between their loops: spinning threads starve the HIP runtime's callback thread (dn_host.cpp hostThreads)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK = 2500.0         # dense fp16 / bf16 MFMA TFLOP/s (MI355X_MICROARCH.md; the 2:1-sparsity figure is not used)
# what the chip DELIVERS of that on operands that toggle: every SIMD streaming v_mfma_f32_32x32x16_f16 on random fp16 values holds 1.47-1.53 GHz = 1 500 TFLOP/s
# (all-ones operands: 2.36 GHz, 2 390): tools/ubench_mfma_lds.hip, profiles/r05_ubench_mfma_lds.txt.  Reported beside the contract's fractions, never instead of them.
MFMA_F16_MEASURED_RANDOM = 1500.0
MFMA_F32_PEAK = 157.0


def cnn_macs(desc):
    return sum(o["k"] * o["cin"] * o["cout"] for o in desc["ops"] if o["op"] == "conv") + \
           sum(o["k"] * o["c"] for o in desc["ops"] if o["op"] == "dwconv") + 47040 + 64 * 3


def _cnn_worker(args):
    """one process of the CPU baseline's CNN leg: render the same slice of positions again and again for `seconds`; returns the positions done"""
    ref, core, resid, sig, seconds = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cnn_torch_ref
    import torch
    torch.set_num_threads(1)
    t0 = time.time(); done = 0
    while time.time() - t0 < seconds or done == 0:
        cnn_torch_ref.run(ref, core, resid, sig)
        done += core.shape[0]
    return done, time.time() - t0


def cpu_baseline(model, n_bases, seed0, full, reads_per_thread=None, cnn_seconds=6.0):
    """The oracle (CPU restatement of the reference path) on ALL host cores, measured, nothing extrapolated (round-3 verdict):
      * normaliseEvents (+ eventalign): OpenMP, one read per thread, schedule(dynamic) -- the shape of the reference's own loop
        (detect.cpp:852) -- over reads_per_thread x cores reads of the workload;
      * the CNN (full scope): the stock-PyTorch CPU fp32 rendering of the same model description, ONE PROCESS PER CORE at once, each
        rendering one read's positions single-threaded (the reference runs one TF_SessionRun per read from every OpenMP thread,
        detect.cpp:653) for ~cnn_seconds: the aggregate positions/s of the loaded box, memory bandwidth contention included.
    value = samples of the sample / (oracle seconds + its positions / measured aggregate CNN rate).  Must run BEFORE the process touches
    the GPU (the CNN leg forks)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from concurrent.futures import ThreadPoolExecutor
    from dnascent_amd import host as _host, synth
    # the CPUs this process can keep busy, not the hardware threads it can see: the pool's hosts show 256 and grant 16 (cgroup cpu.max); threads
    # beyond the quota only get the whole process frozen for the rest of each 100 ms period (host.usable_cpus; round 4)
    cores = _host.usable_cpus()
    if reads_per_thread is None:
        reads_per_thread = max(2, min(16, 256 // cores))     # ~10-30 s of oracle work whatever the quota
    cnn_rate = None
    what_cnn = ""
    first = synth.make_read(seed0, n_bases, model=model, is_reverse=False, sub_rate=0.002, ins_rate=0.001, del_rate=0.001)
    if full:
        import multiprocessing as mp
        from dnascent_amd import cnn_model
        o = po.OracleRead(first, model)
        assert o.normalise() == 0 and o.eventalign() == 0
        pos = o.positions()
        o.free()
        k = min(len(pos["core"]), 4000)                     # ~2 s per call per core with every core loaded
        ref = cnn_model.default_model()[2]
        job = (ref, np.ascontiguousarray(pos["core"][:k]), np.ascontiguousarray(pos["residual"][:k]), np.ascontiguousarray(pos["signal"][:k]), cnn_seconds)
        with mp.get_context("fork").Pool(cores) as pool:               # before any OpenMP team exists in this process
            t0 = time.time()
            res = pool.map(_cnn_worker, [job] * cores, chunksize=1)
            wall = time.time() - t0
        cnn_rate = sum(r[0] for r in res) / max(r[1] for r in res)
        what_cnn = "; CNN: PyTorch CPU fp32 rendering, %d processes x 1 thread at once, %d positions per call, %.1f s: %.0f positions/s aggregate (%.0f per core)" % (
            cores, k, wall, cnn_rate, cnn_rate / cores)
    n = cores * reads_per_thread
    with ThreadPoolExecutor(min(cores, 32)) as ex:                       # the generator is C behind ctypes: threads do run in parallel
        reads = [first] + list(ex.map(lambda i: synth.make_read(seed0 + i, n_bases, model=model, is_reverse=bool(i & 1), sub_rate=0.002, ins_rate=0.001,
                                                                  del_rate=0.001), range(1, n)))
    secs, samples, positions, ok = po.bench_reads(reads, model, full, cores)
    what = "oracle normaliseEvents%s, OpenMP schedule(dynamic), %d reads on %d threads (%d per thread): %.1f s" % (
        " + eventalign" if full else "", n, cores, reads_per_thread, secs)
    cnn_s = positions / cnn_rate if (full and cnn_rate) else 0.0
    if full:
        what_cnn += "; the sample's %d positions at that rate: %.1f s" % (positions, cnn_s)
    return {"value": samples / (secs + cnn_s) / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "threads_timed": cores,
            "Msamples_per_s_per_thread": samples / (secs + cnn_s) / 1e6 / cores, "per_thread_without_cnn": samples / secs / 1e6 / cores,
            "oracle_s": secs, "cnn_s": cnn_s, "cnn_positions_per_s_all_cores": cnn_rate, "hardware_threads_visible": os.cpu_count(),
            "reference_probe_per_thread": "0.15-0.20 Msamples/s (SURVEY.md s6: the real reference, one thread, normaliseEvents + eventalign)",
            "sample": "%d of the %d-base reads of the workload (%d pass QC); %s%s; value = samples / (oracle s + CNN s), every leg measured with all %d "
                      "usable CPUs loaded (%d hardware threads visible; the cgroup's CPU quota is the limit)" % (n, n_bases, ok, what, what_cnn, cores, os.cpu_count() or 0)}


def load_pmc(reads_per_step, bases, what, inflight=None):
    """The committed counter passes (tools/r05_profile.sh -> tools/r05_collect.py -> profiles/r05_pmc_*.json; earlier rounds' as a fall-back): HBM
    bytes per launch and per step, vector instructions of k2_fill, MFMA busy cycles.  Only used when they were taken at THIS workload's shape;
    the number of batches in flight during the counter pass is part of the shape for the chip-level figures (round-3 advisor): the caller
    gets it back as d["workload"]["inflight"] and labels `roofline_chip` with it."""
    for rnd in ("r06", "r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (rnd, "banded" if what == "banded" else "bench"))
        if not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
        except ValueError:
            continue
        w = d.get("workload", {})
        if w.get("reads_per_step") != reads_per_step or w.get("bases") != bases or (what != "banded" and w.get("cnn_math") != what):
            continue
        d["source"] = os.path.basename(path)
        d["inflight_matches"] = inflight is not None and w.get("inflight") == inflight
        return d
    return None


def shard_plan(lens, args):
    """the product driver's plan (shard.plan_windows) for reads of the given lengths in bases; sizes planned at 12.5 samples per base"""
    from dnascent_amd import shard
    return shard.plan_windows(lens * 12.5, args.window_batches * args.batch_samples, args.batch_samples, 4096)


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
    ap.add_argument("--scope", choices=["full", "banded", "mixed"], default="full",
                    help="full = BASELINE configs[2] (default): streamed 50 kb reads, whole pipeline, H2D to results on host; "
                         "banded = configs[1]: 1 000 x 20 kb resident batch, normaliseEvents only (CNN stubbed); "
                         "mixed = configs[4]'s read-length law on one GPU: --reads reads of clip(exp(N(ln 20 000, 0.9^2)), 1 000, 200 000) bases "
                         "(seed 2025) cut into length-bucketed batches by shard.plan_windows (the product driver's plan), whole pipeline")
    ap.add_argument("--reads", type=int, default=36000, help="mixed scope: reads of the workload")
    ap.add_argument("--batch-samples", type=float, default=300e6, help="mixed scope: sample budget of a batch (run_detect's default)")
    ap.add_argument("--window-batches", type=float, default=4.0, help="mixed scope: batches per window of consecutive reads")
    ap.add_argument("--order", choices=["plan", "long-first"], default="plan",
                    help="mixed scope: batch order -- plan = window by window (input order of the windows, longest batch of a window first: what the "
                         "ordered writer needs); long-first = all batches of the run by descending read length (the LPT order of a shared queue)")
    ap.add_argument("--reads-per-step", type=int, default=None, help="reads per batch (default 500 full / 1000 banded)")
    ap.add_argument("--bases", type=int, default=None, help="bases per read (default 50000 full / 20000 banded)")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight per GPU, each on its own context / stream / workspace (default 8)")
    ap.add_argument("--emit", type=int, default=1, help="full scope: format + write the .detect records inside the timed region")
    ap.add_argument("--out", default=None, help="full scope: .detect output path (default: formatted and counted, not written)")
    ap.add_argument("--cnn-math", choices=["f16x3", "bf16x6", "fp32"], default=None)
    ap.add_argument("--pin", type=int, default=0, help="page-lock the input batches (dn_host_register, before the timed region): uploads become asynchronous")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16-steps", type=int, default=3, help="full scope, 1 GPU: after the run, this many steps again with the CNN on bf16 pieces (value_bf16x6: what "
                                                              "the pipeline delivers after an fp16 range / canary escalation); 0 = off")
    ap.add_argument("--fp32-steps", type=int, default=4, help="full scope, 1 GPU: after the run, this many steps again with the CNN in exact fp32 MFMA "
                                                              "arithmetic (value_fp32: the headline metric without the 16-bit split); 0 = off")
    args = ap.parse_args()
    mixed = args.scope == "mixed"
    full = args.scope in ("full", "mixed")
    rps = args.reads_per_step or (500 if full else 1000)
    bases = args.bases or (50000 if full else 20000)
    # contexts (batches) in flight.  Full pipeline, one session of round 5 (gpurun_out/r5w, --steps 14): 4 -> 808 / 805, 5 -> 813 / 813, 6 -> 821 / 822, 8 -> 816 / 814
    # Msamples/s; 6 x 14.5 GB of workspaces + 4 CNN lanes x 16 GiB = 159 of 309 GB (8: 189).  The banded scope (no network) keeps 8.
    inflight = args.inflight or (6 if full else 8)
    os.environ.setdefault("DN_CNN_ROWS", str(8 << 20))      # activation rows resident per CNN pass and lane: 8 Mi rows = 32 GiB, three passes per 500 x 50 kb batch (round 5, with the persistent
    # whole-CU kernels in the network: 4 Mi 853-857, 6 Mi 862-863, 8 Mi 870-874 Msamples/s, profiles/r05_cnn_pass_rows_ab.txt; round 4: 2 Mi -6 %, 8 Mi -3 %)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_base = None
    # DN_BENCH_FORCE_DIST=1: a process group of ONE over RCCL (torchrun --nproc-per-node 1): the N > 1 code path -- torch.cuda beside the library's own contexts,
    # communicator set-up, device-tensor collectives -- executed on a 1-GPU box
    force_dist = os.environ.get("DN_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ["DN_SHARD_FORCE_COLLECTIVES"] = "1"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not force_dist:
        # FIRST, before this process touches the GPU: its CNN leg forks one worker per core
        from dnascent_amd import synth as _synth
        cpu_base = cpu_baseline(_synth.pore_model(), 30000 if mixed else bases, 1000003, full)
        if mixed:
            cpu_base["sample"] += " (mixed scope: 30 kb reads, the mean of the length law)"
    dist = None
    torch = None
    if world > 1 or force_dist:
        # the host side (read generation, record formatting) is OpenMP over reads: the ranks of a node share its cores
        # (torch.distributed.run exports OMP_NUM_THREADS=1 for its workers: the per-rank share replaces that launcher default)
        if os.environ.get("OMP_NUM_THREADS", "1") == "1":
            from dnascent_amd import host as _h
            os.environ["OMP_NUM_THREADS"] = str(max(1, _h.usable_cpus() // world))
        if "DN_HOST_THREADS" not in os.environ:
            # said explicitly, as run_detect does (a share of ONE would otherwise fall through to "all cores" in dn_host.cpp hostThreads); at least two: the
            # formatter at two threads holds 2 x one GPU's record rate (tests/test_format_budget.py)
            from dnascent_amd import host as _h
            os.environ["DN_HOST_THREADS"] = str(max(2, min(64, _h.usable_cpus() // int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
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
    mix = None
    if mixed:
        # BASELINE configs[4]'s length law on ONE GPU through the product driver's plan (shard.plan_windows: windows of consecutive reads, each cut
        # into length-bucketed batches, longest first); sizes are planned from the bases (x the generator's 12.5 samples per base)
        rng = np.random.default_rng(2025 + rank)
        lens = np.clip(np.exp(rng.normal(np.log(20000.0), 0.9, args.reads)), 1000, 200000).astype(np.int64)
        plan, window_of = shard_plan(lens, args)
        if args.order == "long-first":
            order = sorted(range(len(plan)), key=lambda i: -int(lens[plan[i]].max()))
            plan = [plan[i] for i in order]; window_of = window_of[order]
        mix = dict(lens=lens, plan=plan, window_of=window_of)
        args.steps = len(plan)
        args.warmup = min(args.warmup, len(plan))
    n_batches = (args.warmup + args.steps) if full else 1
    nctx = max(1, min(inflight, args.steps if not full else n_batches))
    ctxs = [hip.Context(dev) for _ in range(nctx)]
    for c in ctxs:
        c.load_pore_model(model, 0.14)
        if full:
            # event workspaces sized for one event per 4 samples instead of the detector's own bound of 2 (the synthetic R10.4.1 signal carries one per ~5.2;
            # a batch that overflowed would be run again at 2 by DetectStream: `host.overflow_retries` below): 8 x 14.5 instead of 8 x 21 GB of HBM
            c.set_event_bound(int(os.environ.get("DN_BENCH_EVENT_BOUND", "4")))
            c.load_cnn(cnn_desc, cnn_blob)
            if args.cnn_math:
                c.cnn_set_math(args.cnn_math)
    cnn_math = args.cnn_math or os.environ.get("DN_CNN_MATH", "f16x3")

    # ---- the reads: every batch of the stream is distinct (seeds by rank / batch / read) ----
    t_gen = time.perf_counter()
    seed_base = 1000003 * (rank + 1)
    batches = []
    # N > 1 (the driver's scaling runs): a rank generates inflight + 2 DISTINCT batches and cycles through them -- as many as are ever alive at once in
    # the stream, so no batch object is in flight twice; every submission uploads and computes in full (nothing is cached between steps).  All
    # warm-up + steps batches up front were 21 GB of host memory and 26 s of generation per rank on the CPUs of one rank: 170 GB and ~3.5 min before
    # the first kernel for eight ranks sharing a 16-CPU quota (round-4 verdict, weak 16).  N = 1 keeps every batch distinct.
    n_distinct = n_batches if (world == 1 or mixed or not full) else min(n_batches, inflight + 2)
    for b in range(n_batches):
        if b >= n_distinct:
            batches.append(batches[b % n_distinct])
            continue
        B = host.ReadBatch()
        if mixed:                                              # the warm-up replays the plan's first batches (same objects: an upload does not modify a batch)
            if b < args.warmup:
                continue
            idx = mix["plan"][b - args.warmup]
            got = B.fill_synth_list(model, 2 * (seed_base + idx.astype(np.int64)) + (idx & 1), mix["lens"][idx])
            assert got == len(idx), (got, len(idx))
        else:
            got = B.fill_synth(model, seed_base + b * rps, rps, bases)
            assert got == rps, (got, rps)
        batches.append(B)
    if mixed:
        batches = batches[:args.warmup] + batches
    t_gen = time.perf_counter() - t_gen
    if args.pin and full:
        for B in batches:
            B.pin()

    def barrier():
        for c in ctxs:
            c.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier(device_ids=[local_rank % ndev]) if red_dev == "cuda" else dist.barrier()   # RCCL: name the device instead of letting it guess from the rank

    gather_s = 0.0
    if full:
        warm, timed = batches[:args.warmup], batches[args.warmup:]
        # set-up, not a step: every context the warm-up batches will not reach gets its workspace now (a first upload is a
        # 10+ GB hipMalloc; with --warmup 2 and 4 contexts two of those used to land inside the timed region: 513 against 605)
        if mixed:
            # the batches differ in shape (28 .. 1 900 reads, 2 .. 200 kb): every context gets the workspace of the LARGEST of them now
            # (dn_ctx_reserve) -- regrowing a slab in the middle of the stream frees the old one, and hipFree waits for the whole device
            need = max(ctxs[0].workspace_bytes(B.desc()) for B in timed)
            for c in ctxs:
                c.reserve(int(need * 1.02), collect_bytes=int(args.batch_samples / 12.5 * 0.3 * 29 * 1.3))
        for c in ctxs[len(warm):] if not mixed else ctxs:
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
        cnn_guard = {"canaries_run": int(sum(c.cnn_canaries() for c in ctxs)), "escalations": int(sum(c.cnn_range_escalations() for c in ctxs)),
                     "note": "warm-up + timed steps: per batch its first sequences (>= 4 096 positions) run a second time with bf16 pieces and are compared on the "
                             "device (1e-4); an escalation repeats the batch with bf16 pieces.  Inside the timed region."}
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
    ctxs_layers = []
    for c in ctxs:
        for k, v in c.profile_get().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
        if full:
            ctxs_layers.append(c.profile_layers(len(cnn_desc["ops"])))
        c.profile(False)

    # the same kernels with the GPU to itself (outside the timed region): one batch, nothing else in flight
    solo = {}
    solo_layers, solo_positions = None, 0.0
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
        if full:
            solo_layers = c.profile_layers(len(cnn_desc["ops"]))
        c.profile(False)
        summ = c.summaries()
        solo_positions = float(np.sum(summ["n_positions"][summ["status"] == 0]))

    # the same pipeline in the network's two other arithmetics, a few steps each, timed the same way, after the run: bf16x6 (fp32 operands split EXACTLY into three
    # bf16 pieces, six products: what a range / canary escalation switches a context to) and exact fp32 (v_mfma_f32_32x32x2_f32)
    fp32_leg = bf16_leg = None
    if full and world == 1:
        for mode, steps in (("bf16x6", args.bf16_steps), ("fp32", args.fp32_steps)):
            if steps <= 0 or cnn_math == mode:
                continue
            for c in ctxs:
                c.cnn_set_math(mode)
            leg = batches[-min(steps, len(batches)):]
            host.stream_detect(ctxs, leg[:1], emit=bool(args.emit), out_path=None)          # first pass in this mode: its kernels' code objects load here
            for c in ctxs:
                c.sync()
            t1 = time.perf_counter()
            stm = host.stream_detect(ctxs, leg, emit=bool(args.emit), out_path=None)
            dtm = time.perf_counter() - t1
            res = {"value_" + mode: float(stm.samples) / dtm / 1e6, "steps": len(leg), "ms_per_step": dtm / len(leg) * 1e3,
                   "note": "the same full pipeline with --cnn-math %s (%s), %d steps of %d reads after the main run" % (
                       mode, "exact fp32 MFMA products" if mode == "fp32" else "three bf16 pieces per operand, six products: fp32-equivalent, fp32's exponent range -- "
                       "the arithmetic a context switches to after a range or canary escalation", len(leg), rps)}
            if mode == "fp32":
                fp32_leg = res
            else:
                bf16_leg = res
        for c in ctxs:
            c.cnn_set_math(cnn_math)

    # the only collectives of the path: MAX of the elapsed time, SUM of the counters (dnascent_amd/shard.py)
    rank_stats = None
    if dist is not None:
        busy_local = float(st.seconds_total) if st is not None else dt
        rank_stats = shard.gather_stats(dist, dict(rank=rank, busy_s=busy_local, gather_s=gather_s, elapsed_s=dt, datagen_s=t_gen, distinct_batches=n_distinct,
                                                   format_threads=host.host_threads(), emit_s=float(st.seconds_emit) if st is not None else None),
                                        device=red_dev)
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
        if not full:
            pmb = load_pmc(rps, bases, "banded")
            if pmb and pmb.get("k2_fill"):
                kf = pmb["k2_fill"]
                roof_banded["traffic"] = kf["write_bytes_per_launch"] + kf["fetch_bytes_per_launch_corrected"]
                if kf.get("valu_insts_per_launch") and fill_s > 0:
                    t_issue = kf["valu_insts_per_launch"] * 4.0 / (1024.0 * 2.4e9)
                    roof_banded["issue_bound_frac"] = alg_bytes / t_issue / 1e9 / HBM_PEAK_GBS
                    roof_banded["valu_insts_per_launch"] = kf["valu_insts_per_launch"]
        # K1 (segmentation: k1_scan + k1_detect + k1_events) against the HBM roof, as SURVEY s8d names it: algorithmic 2 B per sample in (int16) +
        # 8 B per event out (start u32 + mean f32) over the summed launch time of the three kernels of a batch.  It is a serial fp64 chain per read
        # (event_detection.c:35-48 order-exact), not a bandwidth kernel: the number is small and is printed because the contract asks for it.
        k1_names = ("k1_scan", "k1_tstat", "k1_detect", "k1_events")
        k1_ms = sum(prof.get(k, (0.0, 0))[0] / max(prof.get(k, (0.0, 0))[1], 1) for k in k1_names if prof.get(k, (0.0, 0))[1])
        k1_bytes = float(np.sum(summ["n_samples"][ok].astype(np.float64) * 2.0 + summ["n_events"][ok].astype(np.float64) * 8.0))
        roof_k1 = {"bound": "hbm", "kernel": "k1_scan + k1_detect + k1_events (one batch: summed launch time)", "achieved": k1_bytes / (k1_ms / 1e3) / 1e9 if k1_ms > 0 else 0.0,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (k1_bytes / (k1_ms / 1e3) / 1e9 / HBM_PEAK_GBS) if k1_ms > 0 else 0.0, "traffic": None,
                   "algorithmic_bytes_per_launch": k1_bytes, "bytes_per_sample": k1_bytes / max(float(np.sum(summ["n_samples"][ok])), 1.0), "mean_launch_ms": k1_ms}
        k1_solo = sum(solo.get(k, 0.0) for k in k1_names)
        if k1_solo > 0:
            roof_k1["solo_launch_ms"] = k1_solo
            roof_k1["solo_frac"] = k1_bytes / (k1_solo / 1e3) / 1e9 / HBM_PEAK_GBS
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
                "parallelism": "reads sharded, %d rank(s), %d batches in flight per GPU, one host thread per rank" % (world, nctx),
                "cnn_rows_per_pass": int(os.environ.get("DN_CNN_ROWS", 4 << 20)), "cnn_lanes": int(os.environ.get("DN_CNN_LANES", "4"))}
            # ---- the network, layer by layer, from THIS run: HIP events bracket every launch of every op on the CNN lane's stream
            #      (dn_profile_get_layer); ops are grouped by the kernel they take (named as rocprofv3 prints it) and kernels by family
            #      (the template name).  A fused separable layer is one op (its pointwise half reports nothing).
            ops = cnn_desc["ops"]
            lay = {}
            for c in ctxs_layers:
                for i, (ms, n, name) in enumerate(c):
                    if n:
                        a_ = lay.setdefault(i, [0.0, 0, name]); a_[0] += ms; a_[1] += n
            positions = float(st.positions)
            issued = {"f16x3": 3.0, "bf16x6": 6.0, "fp32": 1.0}[cnn_math]
            peak = MFMA_F32_PEAK if cnn_math == "fp32" else MFMA_F16_PEAK
            ridge = (peak / issued) * 1e12 / (HBM_PEAK_GBS * 1e9)          # flop per byte at which the two roofs meet for the arithmetic in use

            def op_cost(i):
                """algorithmic (flops, HBM bytes) per position of op i (fp32 activations in and out, + the residual read)"""
                o = ops[i]
                name_i = lay.get(i, [0, 0, ""])[2]
                if name_i.startswith("k3_block64"):        # one launch = a whole residual block (13 ops), or its six separable layers (12): the work of all of them, the
                    span = 13 if "true" in name_i else 12  # bytes of ONE pass over the activations (block input in, block output out; + the shortcut's second read of the input)
                    fl = 0.0
                    for j in range(i, i + span):
                        q = ops[j]
                        fl += 2.0 * (q["k"] * q["c"] if q["op"] == "dwconv" else q["k"] * q["cin"] * q["cout"])
                    return fl, 4.0 * (64 + 64 + (64 if span == 13 else 0))
                if name_i.startswith("k3_pair128"):        # two separable layers in one launch: the work of both, the bytes of ONE pass (first layer's input in, second layer's output out)
                    fl = 0.0
                    for j in range(i, i + 4):
                        q = ops[j]
                        fl += 2.0 * (q["k"] * q["c"] if q["op"] == "dwconv" else q["k"] * q["cin"] * q["cout"])
                    return fl, 4.0 * (ops[i + 1]["cin"] + ops[i + 3]["cout"])
                if o["op"] == "dwconv" and i + 1 < len(ops) and lay.get(i, [0, 0, ""])[2].startswith("k3_sep"):
                    p = ops[i + 1]
                    return 2.0 * (o["k"] * o["c"] + p["cin"] * p["cout"]), 4.0 * (p["cin"] + p["cout"] + (p["cout"] if p.get("add", -1) >= 0 else 0))
                if o["op"] == "conv":
                    return 2.0 * o["k"] * o["cin"] * o["cout"], 4.0 * (o["cin"] + o["cout"] + (o["cout"] if o.get("add", -1) >= 0 else 0))
                if o["op"] == "dwconv":
                    return 2.0 * o["k"] * o["c"], 8.0 * o["c"]
                if o["op"] == "encode_gru":
                    return 2.0 * 47040, 4.0 * (20 + 2 + 64)
                if o["op"] == "dense_softmax":
                    return 2.0 * o["cin"] * o["cout"], 4.0 * (o["cin"] + o["cout"])
                if o["op"] == "add_relu":
                    return float(o["c"]), 12.0 * o["c"]
                return 0.0, 0.0
            kern = {}
            for i, (ms, n, name) in lay.items():
                fl, by = op_cost(i)
                k_ = kern.setdefault(name, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, layers=0))
                k_["ms"] += ms; k_["launches"] += n; k_["flops"] += fl * positions; k_["bytes"] += by * positions; k_["layers"] += 1
            fam = {}
            for name, k_ in kern.items():
                f_ = fam.setdefault(name.split("<")[0], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, kernels={}))
                for q in ("ms", "launches", "flops", "bytes"):
                    f_[q] += k_[q]
                f_["kernels"][name] = k_
            k3_ms = sum(f_["ms"] for f_ in fam.values()) or 1.0
            pmc = None if mixed else load_pmc(rps, bases, cnn_math, inflight=nctx)

            def roof(label, d, solo_key=None):
                secs = d["ms"] / 1e3
                inten = d["flops"] / d["bytes"] if d["bytes"] else 0.0
                tf, gbs = d["flops"] / secs / 1e12, d["bytes"] / secs / 1e9
                r = {"kernel": label, "launches": d["launches"], "mean_launch_ms": d["ms"] / d["launches"], "share_of_network_time": d["ms"] / k3_ms,
                     "algorithmic_flops_per_launch": d["flops"] / d["launches"], "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                     "flop_per_byte": inten, "ridge_flop_per_byte": ridge, "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "mfma_TFLOPs": tf,
                     "mfma_frac": tf / peak, "mfma_issued_frac": issued * tf / peak,
                     "mfma_issued_over_measured_ceiling": (issued * tf / MFMA_F16_MEASURED_RANDOM) if peak == MFMA_F16_PEAK else None, "traffic": None,
                     # the family's algorithmic work of the WHOLE timed run over the run's wall time: the launches of four CNN lanes overlap, so
                     # their summed launch time exceeds the step and the per-launch `frac` is diluted; this one cannot be (round-3 verdict)
                     "work_per_step_over_step_time": {"TFLOPs": d["flops"] / dt / 1e12, "GBs": d["bytes"] / dt / 1e9,
                                                      "frac": (d["flops"] / dt / 1e12 / peak) if inten >= ridge else (d["bytes"] / dt / 1e9 / HBM_PEAK_GBS)}}
                if inten >= ridge:
                    r.update(bound="mfma", achieved=tf, peak=peak, unit="TFLOP/s", frac=tf / peak)
                else:
                    r.update(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
                return r
            fams = []
            for fname, f_ in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
                r = roof(fname + " (%d kernel instance(s), %d layers of the network)" % (len(f_["kernels"]), sum(k_["layers"] for k_ in f_["kernels"].values())), f_)
                r["instances"] = {}
                tr_w = tr_f = 0.0; tr_n = 0
                for name, k_ in sorted(f_["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
                    ri = roof(name, k_)
                    if pmc and name in pmc.get("kernels", {}):
                        pk = pmc["kernels"][name]
                        ri["traffic"] = pk["write_bytes_per_launch"] + pk["fetch_bytes_per_launch_corrected"]
                        ri["traffic_over_algorithmic"] = ri["traffic"] / ri["algorithmic_bytes_per_launch"] * (pk.get("positions_per_launch_ratio", 1.0))
                        tr_w += pk["write_bytes_per_launch"] * pk["launches"]; tr_f += pk["fetch_bytes_per_launch_corrected"] * pk["launches"]; tr_n += pk["launches"]
                    r["instances"][name] = {q: ri[q] for q in ("launches", "mean_launch_ms", "bound", "frac", "hbm_frac", "mfma_frac", "mfma_issued_frac", "traffic",
                                                               "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch")}
                if tr_n:
                    r["traffic"] = (tr_w + tr_f) / tr_n                      # HBM bytes per launch of the family, PMC (launch-weighted over its instances)
                fams.append(r)
            note = ("achieved = algorithmic work of the launches (per-position figures x positions processed) / summed launch time, HIP events around every launch "
                    "in the timed region on the stream it runs on; a launch shares the chip with the other CNN lanes and the per-read stages of the batches in flight. "
                    "bound: flop/byte of the family against the ridge (%.0f flop/B for %s at %g issued products per fp32 product). traffic: HBM bytes per launch from the "
                    "committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/r0N_pmc_bench.json) when they match this workload, else null" % (ridge, cnn_math, issued))
            if fams:
                out["roofline"] = dict(fams[0], note=note)
                out["roofline_families"] = [{q: f_[q] for q in ("kernel", "share_of_network_time", "launches", "mean_launch_ms", "bound", "achieved", "peak", "unit", "frac",
                                                               "hbm_frac", "mfma_frac", "mfma_issued_frac", "traffic", "work_per_step_over_step_time")} for f_ in fams]
            mac = cnn_macs(cnn_desc)
            cnn_ms, cnn_n = prof.get("k3_cnn", (0.0, 0))
            flops = 2.0 * mac * positions / max(cnn_n, 1)
            ach = flops / ((cnn_ms / max(cnn_n, 1)) / 1e3) / 1e12 if cnn_ms > 0 else 0.0
            roof_net = {"bound": "mfma", "kernel": "k3_cnn (all layers of one batch = one dn_run_cnn)", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": None, "algorithmic_flops_per_launch": flops, "mean_launch_ms": cnn_ms / max(cnn_n, 1),
                        "issued_frac": ach * issued / peak,
                        "note": "achieved = algorithmic fp32 flops (2 x MACs x positions) / launch time; every fp32 product is issued as %g "
                                "16-bit MFMA products (issued_frac counts those)" % issued}
            if "k3_cnn" in solo:
                roof_net["solo_launch_ms"] = solo["k3_cnn"]
                roof_net["solo_frac"] = flops / (solo["k3_cnn"] / 1e3) / 1e12 / peak
            if solo_layers:                                                    # the same families with the chip to themselves (one untimed batch after the run)
                sfam = {}
                for i, (ms, n, name) in enumerate(solo_layers):
                    if n:
                        fl, by = op_cost(i)
                        f_ = sfam.setdefault(name.split("<")[0], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                        f_["ms"] += ms; f_["launches"] += n; f_["flops"] += fl * solo_positions; f_["bytes"] += by * solo_positions
                for r in [out.get("roofline")] + out.get("roofline_families", []):
                    f_ = sfam.get(r["kernel"].split(" ")[0]) if r else None
                    if f_ and f_["ms"] > 0:
                        r["solo_mean_launch_ms"] = f_["ms"] / f_["launches"]
                        r["solo_frac"] = (f_["flops"] / (f_["ms"] / 1e3) / 1e12 / peak) if r["bound"] == "mfma" else (f_["bytes"] / (f_["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS)
            out["roofline_network"] = roof_net
            if "roofline" not in out:
                out["roofline"] = roof_net
            if pmc and pmc.get("step"):
                stp = pmc["step"]
                hbm = (stp["write_bytes"] + stp["fetch_bytes_corrected"]) / (dt / args.steps) / 1e9
                mf = issued * 2.0 * mac * positions / args.steps / (dt / args.steps) / 1e12
                out["roofline_chip"] = {"hbm_bytes_per_step": stp["write_bytes"] + stp["fetch_bytes_corrected"], "hbm_GBs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS,
                                        "mfma_issued_TFLOPs": mf, "mfma_issued_frac": mf / peak,
                                        "mfma_issued_over_measured_ceiling": (mf / MFMA_F16_MEASURED_RANDOM) if peak == MFMA_F16_PEAK else None, "mfma_util_counter": stp.get("mfma_util"),
                                        "counter_pass": {"source": pmc.get("source"), "inflight": pmc.get("workload", {}).get("inflight"), "this_run_inflight": nctx,
                                                         "same_inflight": bool(pmc.get("inflight_matches"))},
                                        "note": "whole chip over one step of THIS run: HBM bytes of all kernels of a step (PMC passes at this workload's shape: "
                                                "FETCH_SIZE x 2 + WRITE_SIZE) / ms_per_step / 8 TB/s; issued 16-bit MFMA flops of the network per step / ms_per_step / "
                                                "the dense peak; mfma_util_counter = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of the counter pass"}
            if pmc and pmc.get("k2_fill"):
                kf = pmc["k2_fill"]
                roof_banded["traffic"] = kf["write_bytes_per_launch"] + kf["fetch_bytes_per_launch_corrected"]
                if kf.get("valu_insts_per_launch") and fill_s > 0:
                    # the fill is a serial chain per read: its ceiling is vector ISSUE, not bytes.  SQ_INSTS_VALU of the launch at one instruction per
                    # 4 cycles on every SIMD of the chip gives the shortest time this instruction stream can take; the algorithmic bytes over THAT
                    # time is the fraction of the HBM roof the formulation can reach at all
                    t_issue = kf["valu_insts_per_launch"] * 4.0 / (1024.0 * 2.4e9)
                    roof_banded["issue_bound_frac"] = alg_bytes / t_issue / 1e9 / HBM_PEAK_GBS
                    roof_banded["valu_insts_per_launch"] = kf["valu_insts_per_launch"]
            out["roofline_banded"] = roof_banded
            out["roofline_k1"] = roof_k1
            if bf16_leg:
                out["value_bf16x6"] = bf16_leg["value_bf16x6"]
                out["bf16x6_leg"] = bf16_leg
            if fp32_leg:
                out["value_fp32"] = fp32_leg["value_fp32"]
                out["fp32_leg"] = fp32_leg
            if rank_stats:
                out["ranks"] = {"busy_s_min": min(r["busy_s"] for r in rank_stats), "busy_s_max": max(r["busy_s"] for r in rank_stats),
                                "gather_s_max": max(r["gather_s"] for r in rank_stats), "per_rank": rank_stats,
                                "note": "busy_s: a rank's own stream (first upload to its last records on the host); gather_s: its part of the RCCL gather of the "
                                        "per-call results to rank 0, inside the timed region"}
            out["cnn_guard"] = cnn_guard
            out["hbm"] = _hbm_info()
            out["host"] = {"datagen_s": t_gen, "upload_s": st.seconds_upload, "enqueue_s": st.seconds_run, "collect_wait_s": st.seconds_collect, "emit_s": st.seconds_emit,
                           "gather_s": gather_s, "overflow_retries": int(st.overflow_retries), "emission": {"records_per_s": st.calls / st.seconds_emit if st.seconds_emit > 0 else None,
                                                              "MB_per_s": st.bytes_out / st.seconds_emit / 1e6 if st.seconds_emit > 0 else None,
                                                              "bytes": int(st.bytes_out)}}
        else:
            out["config"] = {"workload": "%d synthetic %d kb R10.4.1 reads per GPU, banded-HMM scope (segmentation + rough scaling + adaptive "
                                         "banded alignment + backtrack/QC + Theil-Sen), batch resident in HBM, CNN stubbed" % (rps, bases // 1000),
                             "reads_per_gpu": rps, "bases_per_read": bases, "samples_per_gpu_step": int(samples_step),
                             "reads_passing_qc": int(np.sum(summ["status"] == 0)),
                             "parallelism": "reads sharded, %d rank(s), %d batches in flight per GPU, one host thread per rank" % (world, nctx),
                "cnn_rows_per_pass": int(os.environ.get("DN_CNN_ROWS", 4 << 20)), "cnn_lanes": int(os.environ.get("DN_CNN_LANES", "4"))}
            out["roofline"] = roof_banded
            out["roofline_k1"] = roof_k1
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        if mixed:
            lens = mix["lens"]
            per_batch = []
            for idx in mix["plan"]:
                l_ = lens[idx].astype(np.float64)
                per_batch.append(dict(reads=int(len(idx)), bases=int(l_.sum()), longest=int(l_.max()), shortest=int(l_.min()),
                                      lane_utilisation=float(l_.mean() / l_.max())))
            wsum = float(sum(b_["bases"] for b_ in per_batch))
            out["metric"] += " -- mixed read lengths (configs[4]'s law) on one GPU"
            out["config"]["workload"] = ("%d synthetic R10.4.1 reads of clip(exp(N(ln 20 000, 0.9^2)), 1 000, 200 000) bases (seed 2025; %d .. %d, median %d, mean %d), "
                                         "%d length-bucketed batches of <= %.0f M samples in %d windows (shard.plan_windows, order: %s), full pipeline (%s)" % (
                                             args.reads, int(lens.min()), int(lens.max()), int(np.median(lens)), int(lens.mean()), len(per_batch), args.batch_samples / 1e6,
                                             int(mix["window_of"].max()) + 1, args.order, cnn_math))
            out["mixed"] = {"reads": int(args.reads), "batches": len(per_batch), "order": args.order,
                            "lane_utilisation_weighted": float(sum(b_["lane_utilisation"] * b_["bases"] for b_ in per_batch) / wsum),
                            "longest_read_share_of_serial_kernels": float(1.0 - sum(b_["lane_utilisation"] * b_["bases"] for b_ in per_batch) / wsum),
                            "note": "k1 / k2_fill / k2_chase / k2b_eventalign run one wavefront (or workgroup) per read and are serial chains as long as the read: a batch "
                                    "holds its SIMDs for its LONGEST read.  lane_utilisation = mean / longest read length of a batch (bases-weighted over the batches); "
                                    "1 - that = the share of those kernels' wavefront-time spent waiting for the longest read",
                            "per_batch": per_batch}
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
