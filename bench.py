#!/usr/bin/env python3
"""bench.py -- raw-signal Msamples/s of the `detect` hot path on MI355X (BASELINE.json metric).

Workload at N=1 (BASELINE.json configs[1]): 1 000 synthetic 20 kb R10.4.1 reads on one MI355X, banded-HMM scope
(CNN stubbed): one STEP = one pass of normaliseEvents (segmentation -> rough scaling -> adaptive banded alignment +
backtrack + QC -> Theil-Sen) over the batch, inputs already resident in HBM.  For N>1 every rank owns its own 1 000
reads (reads shard with no data-path collective: weak scaling); value = samples of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.  The `roofline` object is for the dominant kernel (k2_fill, the banded DP): achieved =
ALGORITHMIC bytes per launch (n_bands*100 trace bytes + 4*(E+K) input bytes + 9*n_aligned backtrack
bytes, summed over the reads of the launch) / that kernel's mean launch duration measured with HIP events on the
library's stream inside the timed region.  `cpu_baseline` times the oracle (our CPU restatement, kind "port") on a
bounded sample of the same reads on the host cores.
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


def make_batch(n_reads, n_bases, seed0, model):
    from dnascent_amd import host, synth
    batch = host.ReadBatch()
    reads = []
    for i in range(n_reads):
        r = synth.make_read(seed0 + i, n_bases, model=model, is_reverse=bool(i & 1), sub_rate=0.002, ins_rate=0.001,
                            del_rate=0.001)
        assert batch.add_synth(r) >= 0
        reads.append(r)
    return batch, reads


def cpu_baseline(reads, model, budget_s=20.0, full=False):
    """Oracle (CPU restatement of the reference path) on a bounded sample, one read per thread like detect.cpp:852.
    full: + eventalign, and the CNN through the stock-PyTorch CPU rendering of the same model description (fp32)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    cnn = None
    if full:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import cnn_torch_ref
        import torch
        from dnascent_amd import cnn_model
        torch.set_num_threads(1)                        # one read per thread, like the reference's OpenMP loop
        cnn = (cnn_torch_ref, cnn_model.default_model()[2])

    def one(r):
        o = po.OracleRead(r, model)      # includes int16 -> pA and CIGAR flattening, as the GPU path does
        st = o.normalise()
        if full and st == 0 and o.eventalign() == 0:
            pos = o.positions()
            cnn[0].run(cnn[1], pos["core"], pos["residual"], pos["signal"])
        n = r.n_samples()
        o.free()
        return n

    t0 = time.time()
    one(reads[0])
    per = max(time.time() - t0, 1e-3)
    n = int(max(cores, min(len(reads), budget_s / per * cores)))
    n = min(n, len(reads))
    sample = reads[:n]
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:      # ctypes releases the GIL inside the oracle
        samples = sum(ex.map(one, sample))
    dt = time.time() - t0
    what = "oracle normaliseEvents + eventalign + PyTorch CPU fp32 CNN" if full else "oracle normaliseEvents (CNN excluded)"
    return {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d-base reads of the workload, %s, %.1f s" % (n, reads[0].refseq.shape[0], what, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--bases", type=int, default=20000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scope", choices=["banded", "full"], default="banded",
                    help="banded = BASELINE configs[1] (the default, CNN stubbed); full = configs[2]'s pipeline at this batch size: "
                         "normalise + eventalign + CNN, probabilities left in HBM")
    ap.add_argument("--inflight", type=int, default=8,
                    help="batches in flight per GPU (each on its own context/stream/workspace); every stage is latency-bound "
                         "at <= 1 wavefront per SIMD for a 1000-read batch, so consecutive steps are overlapped")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # DN_BENCH_BACKEND=gloo (+ ranks sharing a device) exists only to exercise the N > 1 code path on a 1-GPU box
        backend = os.environ.get("DN_BENCH_BACKEND", "nccl")
        ndev = max(1, torch.cuda.device_count())
        if backend == "nccl":
            torch.cuda.set_device(local_rank % ndev)
        dist.init_process_group(backend, rank=rank, world_size=world)

    from dnascent_amd import hip, synth
    model = synth.pore_model()
    dev = (local_rank % max(1, hip.lib().dn_device_count())) if world > 1 else 0
    red_dev = "cuda" if (dist is not None and os.environ.get("DN_BENCH_BACKEND", "nccl") == "nccl") else "cpu"
    nctx = max(1, min(args.inflight, args.steps))
    ctxs = [hip.Context(dev) for _ in range(nctx)]
    batch, reads = make_batch(args.reads, args.bases, 1000003 * (rank + 1), model)
    stages = ["normalise"] if args.scope == "banded" else ["normalise", "eventalign", "cnn"]
    cnn_desc = None
    if args.scope == "full":
        from dnascent_amd import cnn_model
        cnn_desc, cnn_blob, _ = cnn_model.default_model()
    for c in ctxs:
        c.load_pore_model(model, 0.14)
        if cnn_desc is not None:
            c.load_cnn(cnn_desc, cnn_blob)
        batch.upload(c)                    # inputs resident in HBM before the timed region (one copy per in-flight slot)
    ctx = ctxs[0]
    samples_per_step = batch.samples()

    def sync_all():
        for c in ctxs:
            c.sync()

    def barrier():
        sync_all()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    def run_steps(k):
        """k steps in total; slot j runs steps j, j+nctx, ... on its own host thread (dn_run_banded syncs its stream)."""
        if nctx == 1:
            for _ in range(k):
                for st in stages:
                    ctx.run(st)
            ctx.sync()
            return
        import threading

        errors = []

        def worker(j):
            try:
                for _ in range(j, k, nctx):
                    for st in stages:
                        ctxs[j].run(st)
                ctxs[j].sync()
            except BaseException as e:          # a failed step must fail the run, not shorten it
                errors.append(e)
        th = [threading.Thread(target=worker, args=(j,)) for j in range(nctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]

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
    prof = {}
    for c in ctxs:
        for k, v in c.profile_get().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
        c.profile(False)
    summ = ctx.summaries()
    # every in-flight slot ran the same batch concurrently with the others: their per-read results must be identical bit for bit
    # (outside the timed region; a disagreement fails the run instead of reporting a number)
    for j, c in enumerate(ctxs[1:], 1):
        if c.summaries().tobytes() != summ.tobytes():
            raise SystemExit("bench: in-flight slot %d disagrees with slot 0 on the per-read results" % j)
    # the same kernel with the GPU to itself (not part of the timed region): one batch, nothing else in flight
    solo_fill_ms = None
    if nctx > 1 and rank == 0:
        ctx.profile(True); ctx.profile_reset()
        for _ in range(2):
            ctx.run("normalise")
        ctx.sync()
        pf = ctx.profile_get().get("k2_fill")
        if pf and pf[1]:
            solo_fill_ms = pf[0] / pf[1]
        ctx.profile(False)

    # the only collectives of the path: MAX of the elapsed time, SUM of the counters (dnascent_amd/shard.py)
    from dnascent_amd import shard
    total_samples = float(samples_per_step)
    if dist is not None:
        dt = shard.reduce_max(dist, dt, device=red_dev)
        total_samples = shard.reduce_counters(dist, [total_samples], device=red_dev)[0]

    if rank == 0:
        fill_ms, fill_n = prof["k2_fill"]
        ok = summ["status"] != 5
        alg_bytes = float(np.sum(summ["n_bands"][ok].astype(np.float64) * 100.0 +
                                 4.0 * (summ["n_events"][ok].astype(np.float64) + summ["n_kmers_query"][ok]) +
                                 9.0 * summ["n_aligned"][ok]))
        fill_s = (fill_ms / max(fill_n, 1)) / 1e3
        achieved = alg_bytes / fill_s / 1e9 if fill_s > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (cannot be collected inside a timed run)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_e_pmc_k2_fill.json")))
            if pm["workload"] == {"reads": args.reads, "bases": args.bases}:
                traffic = pm["write_bytes"] + pm["fetch_bytes_corrected"]
        except Exception:
            traffic = None
        out = {
            "metric": "raw-signal Msamples/sec (whole node) on `detect`",
            "value": total_samples * args.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32/f64",
            "data": "synthetic",
            "config": {"workload": "%d synthetic %d kb R10.4.1 reads per GPU, banded-HMM scope (segmentation + rough scaling + "
                                   "adaptive banded alignment + backtrack/QC + Theil-Sen), CNN stubbed" % (args.reads, args.bases // 1000),
                       "reads_per_gpu": args.reads, "bases_per_read": args.bases, "samples_per_gpu_step": int(samples_per_step),
                       "reads_passing_qc": int(np.sum(summ["status"] == 0)), "parallelism": "reads sharded, %d rank(s), %d batches in flight per GPU" % (world, nctx)},
            "roofline": {"bound": "hbm", "kernel": "k2_fill", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_ms": fill_ms / max(fill_n, 1)},
            "kernel_ms_per_launch": {k: v[0] / v[1] for k, v in prof.items() if v[1]},
        }
        if solo_fill_ms:
            # launch durations in the timed region are stretched by the other batches in flight (the kernels time-share the
            # SIMDs); the kernel's own rate is what it reaches with one batch on the GPU
            out["roofline_solo"] = {"bound": "hbm", "kernel": "k2_fill", "achieved": alg_bytes / (solo_fill_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": alg_bytes / (solo_fill_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "mean_launch_ms": solo_fill_ms,
                                    "note": "one batch in flight, outside the timed region"}
        if args.scope == "banded" and (args.reads, args.bases) == (1000, 20000):
            # what actually binds this scope: instruction issue.  Wave-instructions per step from the committed PMC
            # pass of the same workload (profiles/r01_e_pmc_instruction_mix.csv, counters cannot be read inside a timed run),
            # rate = that count / the measured step time; peak = 1024 SIMDs x one VALU wave-instruction per 4 cycles at 2.4 GHz
            try:
                valu = salu = 0.0
                for line in open(os.path.join(ROOT, "profiles", "r01_e_pmc_instruction_mix.csv")):
                    f = line.strip().split(",")
                    if len(f) >= 3 and not line.startswith("#") and f[0] != "kernel":
                        valu += float(f[1]); salu += float(f[2])
                step_s = out["ms_per_step"] / 1e3
                out["roofline_issue"] = {"bound": "valu_issue", "achieved": valu / step_s / 1e9, "peak": 1024 * 2.4 / 4.0, "unit": "G wave-instr/s",
                                         "frac": valu / step_s / 1e9 / (1024 * 2.4 / 4.0), "valu_per_step": valu, "salu_per_step": salu,
                                         "note": "instruction counts from the committed PMC pass; informational, `roofline` above follows the contract"}
            except Exception:
                pass
        if args.scope == "full":
            # dominant stage = the CNN: algorithmic flops = 2 x MACs of the description x positions (3.7 MFLOP x L);
            # peak = the dense 16-bit MFMA rate (2.5 PFLOP/s) / the products per fp32 product of the split in use: 3 for the two-piece
            # fp16 split (default), 6 for the three-piece bf16 split; 157 for exact fp32 MFMA.  See DESIGN.md s4b
            cnn_math = os.environ.get("DN_CNN_MATH", "f16x3")
            cnn_peak = {"f16x3": 2500.0 / 3, "bf16x6": 2500.0 / 6, "fp32": 157.0}[cnn_math]
            mac = sum(o["k"] * o["cin"] * o["cout"] for o in cnn_desc["ops"] if o["op"] == "conv") + 47040
            pos = float(np.sum(summ["n_positions"][summ["status"] == 0]))
            cnn_ms, cnn_n = prof["k3_cnn"]
            ach = 2.0 * mac * pos / ((cnn_ms / max(cnn_n, 1)) / 1e3) / 1e12
            out["config"]["workload"] = out["config"]["workload"].replace("banded-HMM scope (segmentation + rough scaling + adaptive banded "
                                                                          "alignment + backtrack/QC + Theil-Sen), CNN stubbed",
                                                                          "full pipeline (normalise + eventalign + CNN, %s math)" % cnn_math)
            out["config"]["cnn_positions_per_gpu_step"] = int(pos)
            out["roofline_banded"] = out["roofline"]
            out["roofline"] = {"bound": "mfma", "kernel": "k3_cnn (all layers of one pass)", "achieved": ach, "peak": cnn_peak, "unit": "TFLOP/s",
                               "frac": ach / cnn_peak, "traffic": None, "mean_launch_ms": cnn_ms / max(cnn_n, 1),
                               "algorithmic_flops_per_launch": 2.0 * mac * pos}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(reads, model, full=(args.scope == "full"))
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
