#!/usr/bin/env python3
"""run_detect -- the reads of a binary read container through the whole HIP path into one .detect file, on 1..N GPUs.

    python -m dnascent_amd.run_detect --container reads.dnrc --out out.detect
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        -m dnascent_amd.run_detect --container reads.dnrc --out out.detect

The product path of SURVEY.md s8e end to end, one process per GPU (`torch.distributed` is plumbing: backend nccl == RCCL over xGMI;
--backend gloo with ranks sharing a device exists for the tests).  It is a STREAM, like the reference's loop over a bounded buffer of
reads (detect.cpp:821-907), never "load everything, run, gather everything":

  1. every rank indexes the container (record sizes + offsets: seeks, no payload) and derives the same PLAN from it: windows of
     consecutive reads (--window-batches x world x --batch-samples samples), each cut into length-bucketed batches (shard.plan_windows);
  2. the ranks PULL batch ids from one shared counter (shard.WorkCounter: a TCPStore add, no collective) -- the reference's
     `schedule(dynamic)` (detect.cpp:852): a rank whose reads fail QC early, or are short, takes more batches;
  3. a rank holds at most --inflight + 3 batches on the host: the next two are loaded (all cores, direct seeks) while --inflight of
     them are on the GPU (DNAscent::DetectStream: upload, normaliseEvents, eventalign, CNN, dn_collect, records formatted);
  4. THERE IS NO WRITER RANK (round 5; the reference funnels every record through one critical section, detect.cpp:902-906 -- the thing not to
     copy: a central formatter has the CPU share of ONE rank for the text of all of them).  When a rank has collected its last batch of a window
     it announces {ordinal, exact text length} of its reads through the process group's store (16 bytes per READ: a few KB per window, no
     collective, nothing on the GPU); with every rank's announcement an exclusive scan in input order gives each record its offset in the
     file; the rank then formats ITS OWN records (the C++ formatter on its share of the host's threads) and pwrite()s them in place -- one
     node, one file system.  All of it on a gather thread, while the GPU works on the next window.  The file is byte-identical whatever the
     number of ranks (input order is what the reference produces with one thread); --central-writer keeps round 4's form (packed calls
     gathered to rank 0 in one grouped receive per window and formatted there);
  5. one all-reduce of the counters (reads ok / failed, samples) at the end.

A read the reference's own filters reject (empty / too short signal, detect.cpp:839, pod5.cpp:64) counts as FAILED, as there.  A
truncated container is fatal: the rank raises the shared abort flag, every rank stops pulling, walks the remaining (empty) window
gathers -- so nothing hangs -- and the run exits non-zero without a half-written claim of success.

The CNN is loaded from --model PREFIX (tools/convert_savedmodel.py output) or, without it, is the seeded-random default model --
whose probabilities are synthetic (a warning says so).  Pore model: --pore-model FILE (text: kmer<TAB>mean, data_IO.cpp:160-175) or
the synthetic table of the tests.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the host library's OpenMP teams must SLEEP between their loops: spinning threads starve the HIP runtime's callback thread (dn_host.cpp hostThreads)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# activation rows resident per CNN pass and lane: 4 Mi = 16 GiB per lane.  bench.py takes 8 Mi (steady state +2 %); here the footprint decides: a run that starts right
# behind another process's exit gets memory the driver is still clearing, and its first uploads into every fresh allocation wait for that -- seconds that grow with
# what the process allocates (profiles/r05_run_detect_back_to_back.txt: 149 GB at 4 Mi, 218 GB at 8 Mi; cold, the two differ by 0.2 s of a 7.4 s stream)
os.environ.setdefault("DN_CNN_ROWS", str(4 << 20))
if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "DN_HOST_THREADS" not in os.environ:
    # N ranks share the host's cores: each rank's loader / packer / formatter loops take their share (dn_host.cpp hostThreads)
    from dnascent_amd.host import usable_cpus as _usable                   # the cgroup's CPU quota, not the hardware threads in sight
    os.environ["DN_HOST_THREADS"] = str(max(2, min(64, _usable() // int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])))))


def load_pore_model(path):
    """import_poreModel_staticStdv (data_IO.cpp:144-190): '#' lines skipped, kmer<TAB>mean, table in kmer2index order (A0 T1 G2 C3)"""
    code = {"A": 0, "T": 1, "G": 2, "C": 3}
    m = np.zeros(262144, np.float64)
    seen = 0
    for line in open(path):
        if not line.strip() or line[0] == "#":
            continue
        k, v = line.split()[:2]
        if len(k) != 9:
            continue
        idx = 0
        for ch in k:
            idx = idx * 4 + code[ch]
        m[idx] = float(v); seen += 1
    if seen != 262144:
        raise ValueError("%s: %d of 262144 9-mers" % (path, seen))
    return m


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--container", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--model", default=None, help="CNN description prefix (tools/convert_savedmodel.py); default: synthetic weights")
    ap.add_argument("--pore-model", default=None)
    ap.add_argument("--inflight", type=int, default=5, help="batches in flight on the GPU (one context each: ~14.5 GB of HBM per 300 M samples at the default event bound).  "
                    "4-8 stream alike (bench.py: within 1.5 %%); from 6 up the contexts' 160+ GB of hipMalloc, which run beside the first batches, took 3.5 s instead of 1 "
                    "in most runs (gpurun_out/r5v: 10 000 x 50 kb in 7.8-7.9 s at 4 and 5, 10.0-10.6 s at 6, 7.9-10.0 s at 8)")
    ap.add_argument("--batch-samples", type=float, default=300e6, help="sample budget of one batch")
    ap.add_argument("--batch-reads", type=int, default=4096)
    ap.add_argument("--window-batches", type=float, default=2.0,
                    help="a window (the unit of the ordered gather + write) holds about this many batches PER RANK")
    ap.add_argument("--gather-chunk-mb", type=int, default=256)
    ap.add_argument("--prefetch", type=int, default=3, help="batches loaded ahead of the one being submitted (loader threads)")
    ap.add_argument("--backend", default=os.environ.get("DN_BACKEND", "nccl"))
    ap.add_argument("--central-writer", action="store_true", help="round 4's form: packed results gathered to rank 0 and formatted there")
    ap.add_argument("--event-bound", type=int, default=4, help="samples per event the workspaces are sized for (2 = the detector's own bound; a batch that overflows a tighter one is re-run at 2)")
    ap.add_argument("--header", default=None, help="text written before the records (e.g. DNAscent::writeDetectHeader)")
    ap.add_argument("--stats", default=None, help="rank 0 writes a JSON with per-rank busy / gather seconds, batches, peak buffered bytes")
    a = ap.parse_args(argv)

    t_proc = time.time()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dev_t = "cpu"
    # DN_RUN_DETECT_FORCE_DIST=1: a process group of ONE (under torch.distributed.run --nproc-per-node 1): the N > 1 code path -- torch.cuda and a RCCL communicator
    # beside the library's contexts, the counters and statistics as device-tensor collectives -- executed on a 1-GPU box
    force_dist = os.environ.get("DN_RUN_DETECT_FORCE_DIST") == "1"
    if force_dist:
        os.environ["DN_SHARD_FORCE_COLLECTIVES"] = "1"
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
            dev_t = "cuda:%d" % (local % max(1, torch.cuda.device_count()))
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    def _barrier():
        # RCCL: name the device instead of letting the backend guess it from the rank
        dist.barrier(device_ids=[int(dev_t.split(":")[1])]) if dev_t != "cpu" else dist.barrier()

    from dnascent_amd import cnn_model, hip, host, shard, synth
    t0 = time.time()
    sizes, offsets = host.container_index(a.container)
    t_index = time.time() - t0
    batches, window_of = shard.plan_windows(sizes, a.window_batches * world * a.batch_samples, a.batch_samples, a.batch_reads)
    pore = load_pore_model(a.pore_model) if a.pore_model else synth.pore_model()
    if a.model:
        desc, blob = cnn_model.load(a.model)
    else:
        desc, blob, _ = cnn_model.default_model()
        if rank == 0:
            print("run_detect: NO --model given: the CNN runs seeded RANDOM weights; the probabilities are synthetic", file=sys.stderr)
    ndev = max(1, hip.lib().dn_device_count())
    t_ctx = time.time()
    ctxs = [hip.Context(local % ndev) for _ in range(max(1, min(a.inflight, max(1, len(batches)))))]
    for c in ctxs:
        c.load_pore_model(pore, 0.14)
        c.load_cnn(desc, blob)
        # a read's event workspace: samples / 4 + 64 instead of the detector's own bound samples / 2 (R10.4.1: one event per 5-8 samples) -- 14.5 instead of
        # 21 GB per 300 M-sample context, less to hipMalloc before the first batch; a batch that does overflow is run again at the safe bound by DetectStream
        c.set_event_bound(a.event_bound)
    t_ctx = time.time() - t_ctx
    import threading
    ready = [threading.Event() for _ in ctxs]

    class GatedStream:
        """DetectStream whose later contexts may still be getting their workspace (a helper thread, below): submit() waits for the context the batch
        lands on -- the k-th submission goes to context k mod n"""

        def __init__(self, inner):
            self.inner, self.k = inner, 0
            self.full, self.in_flight, self.collect, self.stats, self.close = inner.full, inner.in_flight, inner.collect, inner.stats, inner.close

        def submit(self, batch, tag):
            ready[self.k % len(ready)].wait()
            self.k += 1
            self.inner.submit(batch, tag)
    engine = GatedStream(host.DetectStream(ctxs, emit="packed"))
    free = []

    def load(ords):                                            # runs on the driver's loader threads (two at a time)
        try:
            b = free.pop()
        except IndexError:
            b = host.ReadBatch()
        b.clear()
        return b, b.add_container_at(a.container, offsets[ords])

    out_f = None
    head = (a.header or "").encode()
    out_pos = [len(head)]
    write_s = [0.0]
    if rank == 0:
        out_f = open(a.out, "wb")
        if head:
            out_f.write(head); out_f.flush()
    warm = None
    if dist is not None:
        # the file exists (and is empty but for the header) before anybody else opens it: a key in the group's store, not a barrier -- over RCCL the first
        # collective pays the communicator's set-up (seconds), which instead rides beside the stream on a helper thread (shard.warm_collectives)
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        ready_key = "dn_run_detect/file_ready/%d" % int(store.add("dn_run_detect/runs/r%d" % rank, 1))
        if rank == 0:
            fst = os.fstat(out_f.fileno())
            store.set(ready_key, "%d:%d" % (fst.st_dev, fst.st_ino))       # which file it is: the other ranks must see the SAME one (below)
        else:
            t_w = time.time()
            while not store.check([ready_key]):                # polled: TCPStore.wait() would hold this client's lock for the whole wait
                if time.time() - t_w > 600:
                    raise RuntimeError("run_detect: rank 0 never announced the output file")
                time.sleep(0.005)
        warm = shard.warm_collectives(dist, dev_t)
        if warm is not None and a.central_writer:
            warm.join()                                        # that form's gather thread issues point-to-point operations during the stream
    if rank != 0 and not a.central_writer:
        # every rank writes its own records in place: same file, own descriptor.  That needs ONE file system under all ranks (one node, or a shared mount): a
        # rank that cannot open the path -- or opens a different file of the same name -- says so now instead of dying later or writing where nobody looks
        # (round-5 advisor).  st_dev is compared only between ranks of one host (device numbers of a network mount differ from host to host).
        try:
            out_f = open(a.out, "r+b")
        except OSError as e:
            raise SystemExit("run_detect: rank %d cannot open %s (%s): without a writer rank every rank writes into the output file itself -- put --out on a file "
                             "system all ranks share, or use --central-writer" % (rank, a.out, e))
        want = store.get(ready_key).decode().split(":")
        fst = os.fstat(out_f.fileno())
        same_host = os.environ.get("LOCAL_WORLD_SIZE", str(world)) == str(world)
        if str(fst.st_ino) != want[1] or (same_host and str(fst.st_dev) != want[0]):
            raise SystemExit("run_detect: rank %d sees a different file at %s than rank 0 (inode %s vs %s): use a shared file system or --central-writer" % (
                rank, a.out, fst.st_ino, want[1]))

    def write(text, ordinals, record_bytes):                   # --central-writer: on the writer's gather thread, the group's text by several threads at once
        t_w = time.time()
        host.pwrite_parallel(out_f.fileno(), text, out_pos[0])
        out_pos[0] += len(text)
        write_s[0] += time.time() - t_w

    def write_at(text, src_off, lens, file_off, ordinals):     # on EVERY rank's gather thread: its own records to their places in the file
        t_w = time.time()
        host.pwrite_scatter(out_f.fileno(), text, src_off, lens, file_off)
        write_s[0] += time.time() - t_w

    drv = shard.StreamDriver(dist, batches, window_of, engine, load, write, release=free.append, dst=0, device=dev_t,
                             chunk_bytes=a.gather_chunk_mb << 20, write_at=None if a.central_writer else write_at, file_base=len(head))
    # set-up, not part of the stream: every context gets its workspace now (a 10+ GB hipMalloc), sized from the plan's first batch -- window 0's
    # longest reads at the full sample budget -- scaled to the largest planned batch, + 4 % for batches of other composition (more, shorter
    # reads): regrowing a slab later frees the old one, and hipFree waits for the WHOLE device, i.e. drains every batch in flight
    # Round 5: only the FIRST context is prepared before the stream starts; the others get theirs from a helper thread while the first batches already run
    # (hipMalloc does not drain the device -- only the hipFree of a regrowth did).  The stream's k-th submission waits for context k mod n (GatedStream).
    pre = {}
    setup_exc = None
    bg_exc = []
    bg = None
    try:
        if len(batches):
            t_l = time.time()
            b0, acc0 = load(batches[0])
            if os.environ.get("DN_RUN_DETECT_TIMING") == "1":
                print("run_detect: rank %d first batch loaded in %.3f s" % (rank, time.time() - t_l), file=sys.stderr)
            if b0.size():
                per_sample = ctxs[0].workspace_bytes(b0.desc()) / max(1, b0.samples())
                biggest = max(int(sizes[b].sum()) for b in batches)
                want = (int(per_sample * biggest * 1.04), int(biggest / 12.5 * 0.3 * 29 * 1.3))

                timing = os.environ.get("DN_RUN_DETECT_TIMING") == "1"

                def prepare(c, bt):
                    ta = time.time()
                    c.reserve(want[0], collect_bytes=want[1])
                    c.cnn_reserve(0)                           # its CNN lane's activation buffers (shared by the contexts of a lane: the first one pays)
                    tb = time.time()
                    bt.upload(c)                               # the side tables (per-read mirrors) take their size from a real batch
                    tc = time.time()
                    c.sync()
                    if timing:
                        print("run_detect: rank %d context set-up: reserve %.3f s, upload %.3f s, sync %.3f s" % (rank, tb - ta, tc - tb, time.time() - tc), file=sys.stderr)
                prepare(ctxs[0], b0)

                def prepare_rest():
                    try:
                        bb, _ = load(batches[0])               # its own copy: batch 0 itself may be collected, released and refilled by the loader meanwhile
                        for k in range(1, len(ctxs)):
                            prepare(ctxs[k], bb)
                            ready[k].set()
                        free.append(bb)
                    except BaseException as e:                 # noqa: BLE001
                        bg_exc.append(e)
                        try:
                            drv.counter.abort()
                        except Exception:
                            pass
                    finally:
                        for ev in ready:
                            ev.set()
                if len(ctxs) > 1:
                    bg = threading.Thread(target=prepare_rest, name="dn-prepare", daemon=True)
                    bg.start()
            pre[0] = (b0, acc0)                                # whichever rank pulls batch 0 submits this copy instead of reading it again
    except BaseException as e:                                 # noqa: BLE001 -- a rank that cannot even set up raises the shared abort flag and still walks the windows:
        setup_exc = e                                          # its peers stop at once instead of waiting for the store's timeout (round-4 advisor)
        try:
            drv.counter.abort()
        except Exception:
            pass
    if bg is None:
        for ev in ready:
            ev.set()
    else:
        ready[0].set()
    t_setup = time.time() - t0
    drv.preloaded = pre
    t_stream = time.time()
    ok = drv.run(prefetch=a.prefetch)
    t_stream = time.time() - t_stream
    for bl, _ in drv.preloaded.values():                       # batch 0 as loaded for the set-up, on the ranks that did not pull it: back to the pool
        free.append(bl)
    drv.preloaded = {}
    if bg is not None:
        bg.join()
    if setup_exc is None and bg_exc:
        setup_exc = bg_exc[0]
    if setup_exc is not None:
        ok = False
        drv.failure = drv.failure or setup_exc
    st = engine.stats()
    hbm_info = None
    try:                                                       # device memory in use at the end of the stream (every context, every CNN lane): hipMemGetInfo
        import ctypes
        rt = ctypes.CDLL("libamdhip64.so")
        fr, totl = ctypes.c_size_t(0), ctypes.c_size_t(0)
        if rt.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(totl)) == 0:
            hbm_info = {"used_GB": round((totl.value - fr.value) / 1e9, 1), "total_GB": round(totl.value / 1e9, 1)}
    except OSError:
        pass
    if warm is not None:
        warm.join()
    tot = shard.reduce_counters(dist, [drv.n_ok, drv.n_fail, int(st.samples), 0 if ok else 1], device=dev_t)
    if drv.failure is not None:
        print("run_detect: rank %d aborted: %r" % (rank, drv.failure), file=sys.stderr)
    per_rank = shard.gather_stats(dist, dict(rank=rank, batches=drv.batches_done, busy_s=round(drv.busy_s, 3), gather_s=round(drv.gather_s, 3),
                                             format_s=round(drv.format_s - write_s[0], 3), write_s=round(write_s[0], 3), peak_buffered_bytes=int(drv.peak_pending_bytes),
                                             max_gather_bytes=int(drv.max_gather_bytes), reads_ok=drv.n_ok, reads_failed=drv.n_fail,
                                             upload_s=round(st.seconds_upload, 3), enqueue_s=round(st.seconds_run, 3), collect_wait_s=round(st.seconds_collect, 3), load_wait_s=round(drv.load_wait_s, 3), load_s=round(drv.load_s, 3), driver_submit_s=round(drv.t_submit, 3), driver_collect_s=round(drv.t_collect, 3), driver_engine_collect_s=round(drv.t_engine_collect, 3), driver_hand_over_s=round(drv.t_hand_over, 3),
                                             pack_s=round(st.seconds_emit, 3)), device=dev_t)
    failed = tot[3] > 0
    # the file's record text: without a writer rank drv.text_bytes is what THIS rank formatted, while every rank keeps the same running file position
    text_bytes_all = int(drv.text_bytes) if a.central_writer else int(drv.file_pos) - len(head)
    if out_f is not None:
        out_f.close()
    if rank == 0:
        dt = time.time() - t0
        if failed:
            os.unlink(a.out)                                   # a partial file must not pass for a result
            print("run_detect: ABORTED: a rank failed on its share of %s (loader or engine error; see its message above)" % a.container, file=sys.stderr)
        else:
            busy = [p["busy_s"] for p in per_rank]
            print("run_detect: %d reads ok, %d failed, %.1f M samples, %d rank(s), %d batches in %d window(s), %.2f s (%.1f Msamples/s incl. "
                  "indexing, context set-up and ingestion; the stream itself -- first batch loaded to last record written -- %.2f s = %.1f Msamples/s); per-rank busy %.2f .. %.2f s, gather %.2f s max, formatting %.2f s (rank 0's share), at most %.1f MB of packed results "
                  "buffered on a rank" %
                  (tot[0], tot[1], tot[2] / 1e6, world, len(batches), drv.n_windows, dt, tot[2] / 1e6 / dt, t_stream, tot[2] / 1e6 / t_stream, min(busy), max(busy),
                   max(p["gather_s"] for p in per_rank), drv.format_s, max(p["peak_buffered_bytes"] for p in per_rank) / 1e6))
        if a.stats:
          with open(a.stats, "w") as stats_f:
            json.dump(dict(world=world, batches=len(batches), windows=drv.n_windows, seconds=dt, samples=tot[2], Msamples_per_s=tot[2] / 1e6 / dt,
                           imports_s=round(t0 - t_proc, 3), contexts_s=round(t_ctx, 3), event_bound=a.event_bound, overflow_retries=int(st.overflow_retries),
                           hbm=hbm_info,
                           reads_ok=tot[0], reads_failed=tot[1], text_bytes=int(text_bytes_all), index_s=round(t_index, 3), inflight=len(ctxs),
                           setup_s=round(t_setup, 3), stream_s=round(t_stream, 3), Msamples_per_s_stream=tot[2] / 1e6 / t_stream,
                           recv_groups=drv.stats.get("recv_groups", []), ranks=per_rank, failed=failed), stats_f)
    t_close = time.time()
    engine.close()
    for c in ctxs:
        c.close()
    if dist is not None:
        _barrier()
        dist.destroy_process_group()
    if rank == 0 and not failed:
        print("run_detect: process %.2f s = imports %.2f + index / plan %.2f + contexts %.2f + first workspace %.2f + stream %.2f + close %.2f (+ counters, stats)" % (
            time.time() - t_proc, t0 - t_proc, t_index, t_ctx, t_setup - t_index - t_ctx, t_stream, time.time() - t_close))
    return 1 if failed else 0


if __name__ == "__main__":
    rc = main()
    # Everything this process owes anyone is on disk and closed by now (the output, --stats), the contexts are released.  What is left between here and the
    # shell's prompt is the interpreter's and the HIP runtime's own tear-down (module unloading, code objects, the allocator's pools) -- seconds of it measured
    # from outside (tools/time_run_detect.py: process wall against the in-process account) -- which the kernel does for an exiting process anyway.
    # NOT when somebody else has work to do at exit (round-5 advisor): a profiler's preloaded tool library (rocprofv3: ROCP_TOOL_LIBRARIES / LD_PRELOAD), coverage,
    # torchrun's elastic agent expecting finalisers -- they write their results from atexit handlers / destructors, which os._exit skips.
    sys.stdout.flush(); sys.stderr.flush()
    tool_attached = any(os.environ.get(k) for k in ("ROCP_TOOL_LIBRARIES", "LD_PRELOAD", "ROCPROFILER_REGISTER_FORCE_LOAD", "COVERAGE_PROCESS_START", "COV_CORE_SOURCE",
                                                    "TORCHELASTIC_RUN_ID"))
    slow = os.environ.get("DN_RUN_DETECT_SLOW_EXIT")
    if slow == "1" or (slow is None and tool_attached):
        sys.exit(rc)
    os._exit(rc)
