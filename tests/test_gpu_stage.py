"""GPU: the page-locked staging shard.py's gathers use under RCCL (shard.to_device / to_host; round-5 verdict weak 16) -- blocks of several sizes cross to cuda:0 and
back unchanged through ONE grow-only pinned buffer per thread (it regrows once for the larger block), from two threads at once.  (The two-rank exchange itself cannot
run on a 1-GPU box; the helpers are what it moves its blocks with.)  In a child process, torch first -- as tests/test_gpu_rccl_group_of_one.py does."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import threading
import numpy as np
import torch
assert torch.cuda.is_available()
from dnascent_amd import shard
dev = torch.device("cuda", 0)
errs = []

def work(seed):
    try:
        rng = np.random.default_rng(seed)
        for n in (1, 4097, 3 << 20, 40 << 20, 1 << 20):          # the fourth regrows the stage; the fifth reuses it
            a = rng.integers(0, 256, n, dtype=np.uint8)
            t = shard.to_device(a, dev)
            assert t.device.type == "cuda" and t.shape[0] == n
            back = np.empty(n, np.uint8)
            shard.to_host(t, back)
            assert np.array_equal(a, back)
        assert shard._stage_tls.buf.is_pinned() and shard._stage_tls.buf.shape[0] >= 40 << 20
    except Exception as e:
        errs.append(repr(e))
th = [threading.Thread(target=work, args=(s,)) for s in (1, 2)]
for t in th: t.start()
for t in th: t.join()
assert not errs, errs
a = np.arange(10, dtype=np.uint8)
assert shard.to_device(a, "cpu").numpy().ctypes.data == a.ctypes.data        # the CPU path is the array itself
print("STAGE OK")
"""


@pytest.mark.gpu
def test_blocks_cross_the_pinned_stage_unchanged():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "STAGE OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
