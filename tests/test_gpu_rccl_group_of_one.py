"""RCCL executed: a process group of ONE rank over the `nccl` backend (= RCCL on ROCm) on the GPU box.

The path shards with no data-path collective (detect.cpp:852-907): what crosses ranks is a MAX / SUM all-reduce of counters and the gather of
per-read results (dnascent_amd/shard.py).  Those helpers have run over gloo with CPU tensors (tests/test_shard.py, world_size 2); a 1-GPU box cannot hold two
RCCL ranks, so this test sends a group of one through the SAME helpers with device tensors (DN_SHARD_FORCE_COLLECTIVES=1), and -- the part that matters for
`bench.py --gpus N` -- with torch.cuda initialised and a RCCL communicator alive BESIDE the library's own HIP contexts, streams and kernels in one process.
Runs in a child process: a process group and torch's HIP state do not belong in the test session."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from dnascent_amd import hip, host, shard, synth, cnn_model
out = {}
out["max"] = shard.reduce_max(dist, 1.25, device="cuda")
out["sum"] = shard.reduce_counters(dist, [3, 4.5, 1e15 + 1], device="cuda")
out["stats"] = shard.gather_stats(dist, dict(rank=0, busy_s=0.5), device="cuda")
rc = np.array([2, 0, 3], np.uint64); co = np.arange(5, dtype=np.uint32); e = np.linspace(0, 1, 5).astype(np.float32); b = e[::-1].copy()
got = shard.gather_calls(dist, rc, co, e, b, dst=0, device="cuda")
out["calls_ok"] = bool(len(got) == 1 and (got[0][0] == rc).all() and (got[0][1] == co).all() and (got[0][2] == e).all() and (got[0][3] == b).all())
recs = shard.gather_records(dist, [7, 2], [b"seven\n", b"two\n"], dst=0, device="cuda")
out["records"] = [[o, r.decode()] for o, r in recs]
# the library's own context beside torch.cuda + the communicator: one small batch through the whole hot path, before and after a collective
model = synth.pore_model()
desc, blob, _ = cnn_model.default_model()
ctx = hip.Context(0)
ctx.load_pore_model(model, 0.14)
ctx.load_cnn(desc, blob)
B = host.ReadBatch()
assert B.fill_synth(model, 4100, 6, 3000) == 6
import hashlib
digests = []
for it in range(2):
    st, kept = host.stream_detect([ctx], [B], emit=True, out_path=None, keep=True)
    h = hashlib.sha256(b"".join(kept[k].tobytes() for k in ("read_calls", "coord", "p_edu", "p_brdu"))).hexdigest()
    digests.append([int(st.reads_ok), int(st.calls), h])
    torch.cuda.synchronize(); dist.barrier()
    out["max%d" % it] = shard.reduce_max(dist, float(st.calls), device="cuda")
out["detect"] = digests
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_rccl_group_of_one_beside_the_library_contexts():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", DN_SHARD_FORCE_COLLECTIVES="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert out["max"] == 1.25
    assert out["sum"] == [3.0, 4.5, 1e15 + 1]
    assert out["stats"] == [dict(rank=0, busy_s=0.5)]
    assert out["calls_ok"]
    assert out["records"] == [[2, "two\n"], [7, "seven\n"]]
    assert out["detect"][0] == out["detect"][1] and out["detect"][0][0] > 0 and out["detect"][0][1] > 0
    assert out["max0"] == out["max1"] == float(out["detect"][0][1])
