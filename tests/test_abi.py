"""CPU: the C-ABI library loads and exports every symbol include/dnascent_hip.h declares; host logic (CIGAR maps,
read orientation, trimming) matches the oracle.  No compute call is made here."""
import os
import re

import numpy as np
import pytest

import pyoracle as po
from dnascent_amd import build, hip, host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    if not os.path.exists(build.HIP_SO):
        build.build_hip()
    hdr = open(os.path.join(ROOT, "include", "dnascent_hip.h")).read()
    declared = set(re.findall(r"\b(dn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 28
    L = hip.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert set(hip.SYMBOLS) == declared
    assert L.dn_abi_version() == 7


def test_no_cpu_fallback():
    if hip.lib().dn_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hip.DnError):
        hip.Context(0)


@pytest.mark.parametrize("kw", [dict(), dict(is_reverse=True), dict(ins_rate=0.05, del_rate=0.05),
                                dict(is_reverse=True, ins_rate=0.05, del_rate=0.05, soft_clip_head=30, soft_clip_tail=7)])
def test_cigar_flattening_matches_oracle(model, kw):
    r = synth.make_read(5, 2000, model=model, **kw)
    b = host.ReadBatch()
    assert b.add_synth(r) == 0
    r2q, q2r, r2d = b.maps(0, r.refseq.shape[0], r.basecall.shape[0])
    o_r2q, o_q2r, o_r2d = po.parse_cigar(r.cigar_op, r.cigar_len, r.is_reverse, r.basecall.shape[0])
    assert np.array_equal(r2q, o_r2q) and np.array_equal(q2r, o_q2r) and np.array_equal(r2d, o_r2d)


def test_batch_keeps_sequencing_direction(model):
    r = synth.make_read(6, 1500, model=model, is_reverse=True)
    b = host.ReadBatch()
    b.add_synth(r)
    d = b.desc()
    import ctypes as C
    got = C.string_at(d.refseq, r.refseq.shape[0])
    assert got == r.refseq.tobytes()           # reverse-complemented back into sequencing direction (reads.h:280-286)
    got = C.string_at(d.basecall, r.basecall.shape[0])
    assert got == r.basecall.tobytes()


def test_dorado_trimming(model):
    r = synth.make_read(7, 1500, model=model)
    n = r.adc.shape[0]
    b = host.ReadBatch()
    b.add_synth(r, signal_length=n - 100, signal_trim=50)                       # pod5.cpp:88-92
    assert b.samples() == n - 150
    b.add_synth(r, signal_length=2000, signal_trim=10, signal_start=300, is_split=True)   # pod5.cpp:79-86
    assert b.samples() == (n - 150) + 1990
    assert b.add_synth(r, signal_length=5, signal_trim=0) == -1                 # nothing left: rejected


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes / numpy mirrors of the header's structs have the C compiler's sizes and field offsets (plain C, the
    header must stay includable from C)."""
    import ctypes as C
    import subprocess
    src = tmp_path / "layout.c"
    fields_summary = [n for n in hip.SUMMARY_DTYPE.names]
    fields_op = [f[0] for f in hip.CnnOp._fields_]
    fields_batch = [f[0] for f in hip.BatchDesc._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dnascent_hip.h"', 'int main(void) {',
             'printf("%zu %zu %zu\\n", sizeof(dn_read_summary), sizeof(dn_cnn_op), sizeof(dn_batch_desc));']
    for f in fields_summary:
        lines.append('printf("%%zu\\n", offsetof(dn_read_summary, %s));' % f)
    for f in fields_op:
        lines.append('printf("%%zu\\n", offsetof(dn_cnn_op, %s));' % f)
    for f in fields_batch:
        lines.append('printf("%%zu\\n", offsetof(dn_batch_desc, %s));' % f)
    lines += ['return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes, offs = [int(x) for x in out[:3]], [int(x) for x in out[3:]]
    assert sizes == [hip.SUMMARY_DTYPE.itemsize, C.sizeof(hip.CnnOp), C.sizeof(hip.BatchDesc)]
    want = [hip.SUMMARY_DTYPE.fields[f][1] for f in fields_summary] + [getattr(hip.CnnOp, f).offset for f in fields_op] + \
           [getattr(hip.BatchDesc, f).offset for f in fields_batch]
    assert offs == want


def test_host_threads_follow_the_cpus_the_process_may_use():
    """hostThreads(): an explicit DN_HOST_THREADS is taken as it is; otherwise min(64, cores, cgroup CPU quota), and OMP_NUM_THREADS=1 -- what
    torch.distributed.run exports to every worker -- does NOT reduce a rank's loader / formatter to one thread (round 4: it did).  Each case in its
    own process (the value is read once)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def ask(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("DN_HOST_THREADS", "OMP_NUM_THREADS")}
        e.update(env, PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", "from dnascent_amd import host; print(host.host_threads(), host.usable_cpus())"], env=e, capture_output=True,
                             text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        return [int(x) for x in out.stdout.split()]
    n, usable = ask()
    assert 1 <= n <= 64 and n <= usable <= (os.cpu_count() or 1)
    assert ask(OMP_NUM_THREADS="1")[0] == n                     # the launcher's default is not a request
    assert ask(DN_HOST_THREADS="3", OMP_NUM_THREADS="1")[0] == 3
    if n >= 2:
        assert ask(OMP_NUM_THREADS="2")[0] == 2                 # an explicit team size is
