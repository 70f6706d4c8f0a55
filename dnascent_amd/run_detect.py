#!/usr/bin/env python3
"""run_detect -- the reads of a binary read container through the whole HIP path into one .detect file, on 1..N GPUs.

    python -m dnascent_amd.run_detect --container reads.dnrc --out out.detect
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        -m dnascent_amd.run_detect --container reads.dnrc --out out.detect

The product path of SURVEY.md s8e end to end, one process per GPU (`torch.distributed` is plumbing: backend nccl == RCCL over xGMI;
--backend gloo with ranks sharing a device exists for the tests):

  1. every rank scans the container's record sizes (seeks, no payload) and takes ITS reads: shard.assign_reads, longest-processing-
     time-first by sample count, deterministic -- no data-path collective (detect.cpp:852: reads are independent);
  2. its reads are cut into length-bucketed batches (shard.make_batches) and streamed through --inflight contexts by one host thread
     (DNAscent::streamDetect: upload, normaliseEvents, eventalign, CNN, dn_collect, records formatted in parallel);
  3. the per-read records go to the writer rank in ONE grouped send / recv (shard.gather_records) and are written in INPUT order,
     so the file is byte-identical whatever the number of ranks (the reference writes in completion order, detect.cpp:902-906;
     input order is what it produces with one thread);
  4. one all-reduce of the counters (reads ok / failed, samples).

The CNN is loaded from --model PREFIX (tools/convert_savedmodel.py output) or, without it, is the seeded-random default model --
whose probabilities are synthetic (a warning says so).  Pore model: --pore-model FILE (text: kmer<TAB>mean, data_IO.cpp:160-175) or
the synthetic table of the tests.
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


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
    ap.add_argument("--inflight", type=int, default=4)
    ap.add_argument("--batch-samples", type=float, default=300e6, help="sample budget of one batch")
    ap.add_argument("--batch-reads", type=int, default=2000)
    ap.add_argument("--backend", default=os.environ.get("DN_BACKEND", "nccl"))
    ap.add_argument("--header", default=None, help="text written before the records (e.g. DNAscent::writeDetectHeader)")
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dev_t = "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
            dev_t = "cuda"
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    from dnascent_amd import cnn_model, hip, host, shard, synth
    t0 = time.time()
    sizes = host.container_sizes(a.container)
    mine = shard.assign_reads(sizes, world)[rank]                        # ascending input ordinals of this rank
    local_batches = shard.make_batches(sizes[mine], a.batch_samples, a.batch_reads)
    pore = load_pore_model(a.pore_model) if a.pore_model else synth.pore_model()
    if a.model:
        desc, blob = cnn_model.load(a.model)
    else:
        desc, blob, _ = cnn_model.default_model()
        if rank == 0:
            print("run_detect: NO --model given: the CNN runs seeded RANDOM weights; the probabilities are synthetic", file=sys.stderr)
    ndev = max(1, hip.lib().dn_device_count())
    ctxs = [hip.Context(local % ndev) for _ in range(max(1, min(a.inflight, max(1, len(local_batches)))))]
    for c in ctxs:
        c.load_pore_model(pore, 0.14)
        c.load_cnn(desc, blob)
    batches, ordinals = [], []
    for idx in local_batches:
        b = host.ReadBatch()
        ords = mine[idx]                                                  # ascending within a batch
        got = b.add_container_list(a.container, ords)
        if got != len(ords):
            raise SystemExit("run_detect: container read failed / a read was rejected (%d of %d)" % (got, len(ords)))
        batches.append(b); ordinals += [int(o) for o in ords]
    records, n_ok, n_fail, samples = [], 0, 0, 0
    if batches:
        with tempfile.NamedTemporaryFile(prefix="dn_rank%d_" % rank, suffix=".detect", delete=False) as tf:
            tmp = tf.name
        try:
            st, kept = host.stream_detect(ctxs, batches, emit=True, out_path=tmp, keep=True)
            blob_txt = open(tmp, "rb").read()
        finally:
            os.unlink(tmp)
        o = 0
        for ln in kept["record_bytes"]:
            records.append(blob_txt[o:o + int(ln)]); o += int(ln)
        assert o == len(blob_txt) and len(records) == len(ordinals)
        n_ok, n_fail, samples = int(st.reads_ok), int(st.reads - st.reads_ok), int(st.samples)
    merged = shard.gather_records(dist, ordinals, records, dst=0, device=dev_t)
    tot = shard.reduce_counters(dist, [n_ok, n_fail, samples], device=dev_t)
    if rank == 0:
        with open(a.out, "wb") as f:
            if a.header:
                f.write(a.header.encode())
            for _, rec in merged:
                f.write(rec)
        dt = time.time() - t0
        print("run_detect: %d reads ok, %d failed, %.1f M samples, %d rank(s), %.2f s (%.1f Msamples/s incl. ingestion)" %
              (tot[0], tot[1], tot[2] / 1e6, world, dt, tot[2] / 1e6 / dt))
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
