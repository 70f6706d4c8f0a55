"""Binary read container (ingestion without htslib / libpod5).  CPU only: reads written to a container and loaded
back into a ReadBatch give byte-identical SoA arrays to reads added directly; ranges, counts and malformed files behave."""
import ctypes as C

import numpy as np
import pytest

from dnascent_amd import host, synth

SPECS = [(501, 1500, dict()), (502, 1800, dict(is_reverse=True)), (503, 1200, dict(sub_rate=0.01, ins_rate=0.004, del_rate=0.004)),
         (504, 900, dict(is_reverse=True, soft_clip_head=20, soft_clip_tail=11))]


def _soa(batch):
    d = batch.desc()
    n = batch.size()
    out = {"n": n}
    def arr(ptr, count, dt):
        if count == 0:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dt).itemsize,)).view(dt).copy()
    adc_off = arr(d.adc_off, n + 1, np.uint64); out["adc_off"] = adc_off
    out["adc"] = arr(d.adc, int(adc_off[-1]), np.int16)
    out["cal_offset"] = arr(d.cal_offset, n, np.float32); out["cal_scale"] = arr(d.cal_scale, n, np.float32)
    bo = arr(d.basecall_off, n + 1, np.uint64); ro = arr(d.refseq_off, n + 1, np.uint64)
    out["basecall"] = arr(d.basecall, int(bo[-1]), np.uint8); out["refseq"] = arr(d.refseq, int(ro[-1]), np.uint8)
    out["basecall_off"] = bo; out["refseq_off"] = ro
    return out


def test_container_round_trip(model, tmp_path):
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in SPECS]
    path = str(tmp_path / "reads.dnr")
    host.write_container(path, reads)
    assert host.container_count(path) == len(reads)
    direct = host.ReadBatch()
    for r in reads:
        assert direct.add_synth(r) >= 0
    loaded = host.ReadBatch()
    assert loaded.add_container(path) == len(reads)
    a, b = _soa(direct), _soa(loaded)
    assert a["n"] == b["n"] == len(reads)
    for k in a:
        if k != "n":
            assert np.array_equal(a[k], b[k]), k
    for i, r in enumerate(reads):                            # CIGAR maps are rebuilt on load: same maps as the direct path
        ma = direct.maps(i, r.refseq.shape[0], r.basecall.shape[0]); mb = loaded.maps(i, r.refseq.shape[0], r.basecall.shape[0])
        for x, y in zip(ma, mb):
            assert np.array_equal(x, y)
    # a range
    part = host.ReadBatch()
    assert part.add_container(path, 1, 2) == 2 and part.size() == 2
    p = _soa(part)
    assert np.array_equal(p["refseq"], a["refseq"][int(a["refseq_off"][1]):int(a["refseq_off"][3])])
    # Dorado trimming tags travel with the record (pod5.cpp:75-93): signalLength / signalTrim cut the stored signal
    tpath = str(tmp_path / "trim.dnr")
    host.write_container(tpath, reads[:1], signal_length=reads[0].adc.shape[0], signal_trim=100)
    t = host.ReadBatch(); d = host.ReadBatch()
    assert t.add_container(tpath) == 1 and d.add_synth(reads[0], signal_length=reads[0].adc.shape[0], signal_trim=100) >= 0
    assert np.array_equal(_soa(t)["adc"], _soa(d)["adc"]) and _soa(t)["adc"].shape[0] == reads[0].adc.shape[0] - 100


def test_container_rejects_malformed_files(model, tmp_path):
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in SPECS[:2]]
    path = str(tmp_path / "reads.dnr")
    host.write_container(path, reads)
    raw = open(path, "rb").read()
    assert host.container_count(str(tmp_path / "missing.dnr")) == -1
    bad = str(tmp_path / "bad.dnr")
    open(bad, "wb").write(b"XXXX" + raw[4:])
    assert host.container_count(bad) == -1 and host.ReadBatch().add_container(bad) == -1
    open(bad, "wb").write(raw[: len(raw) // 2])             # truncated inside a record
    assert host.container_count(bad) == len(reads) and host.ReadBatch().add_container(bad) == -1


def test_container_index_and_direct_loads(model, tmp_path):
    """container_index (sizes + offsets from one pass of seeks) and add_container_at (records by offset, any order, read by all host
    cores): the streamed product driver's loader.  Same SoA as the sequential loader; a record the reference's own filters reject
    (here: a signal trimmed to nothing, pod5.cpp:64) is reported as rejected, not as an error; a truncated record IS an error; and a
    cleared batch object can be reused."""
    reads = [synth.make_read(seed, n, model=model, **kw) for seed, n, kw in SPECS]
    path = str(tmp_path / "reads.dnr")
    host.write_container(path, reads)
    sizes, offs = host.container_index(path)
    assert sizes.tolist() == [r.n_samples() for r in reads] == host.container_sizes(path).tolist()
    assert np.all(np.diff(offs.astype(np.int64)) > 0) and offs[0] == 16                  # "DNRC" + version + count
    seq = host.ReadBatch(); assert seq.add_container(path) == len(reads)
    a = _soa(seq)
    order = [2, 0, 3]
    b = host.ReadBatch()
    acc = b.add_container_at(path, offs[order])
    assert acc.tolist() == [1, 1, 1] and b.size() == 3
    p = _soa(b)
    for j, i in enumerate(order):
        assert np.array_equal(p["adc"][int(p["adc_off"][j]):int(p["adc_off"][j + 1])], a["adc"][int(a["adc_off"][i]):int(a["adc_off"][i + 1])])
        assert np.array_equal(p["refseq"][int(p["refseq_off"][j]):int(p["refseq_off"][j + 1])], a["refseq"][int(a["refseq_off"][i]):int(a["refseq_off"][i + 1])])
    b.clear()
    assert b.size() == 0 and b.samples() == 0
    assert b.add_container_at(path, offs).tolist() == [1] * len(reads)
    q = _soa(b)
    for k in a:
        if k != "n":
            assert np.array_equal(a[k], q[k]), k
    # a rejected read among accepted ones: trimmed to 10 samples (< 16)
    tpath = str(tmp_path / "trim.dnr")
    w = host.lib().dnh_container_create(tpath.encode())
    for i, sr in enumerate(reads[:3]):
        qq = np.ascontiguousarray(host.revcomp(sr.basecall) if sr.is_reverse else sr.basecall); f = np.ascontiguousarray(host.revcomp(sr.refseq) if sr.is_reverse else sr.refseq)
        adc = np.ascontiguousarray(sr.adc)
        assert host.lib().dnh_container_add(w, sr.read_id.encode(), sr.contig.encode(), adc.ctypes.data, adc.shape[0], sr.cal_offset, sr.cal_scale,
                                            10 if i == 1 else -1, 0, 0, 0, qq.ctypes.data, qq.shape[0], f.ctypes.data, f.shape[0], sr.cigar_op.ctypes.data,
                                            sr.cigar_len.ctypes.data, sr.cigar_op.shape[0], sr.ref_start, int(sr.is_reverse)) == 0
    assert host.lib().dnh_container_close(w) == 0
    _, toffs = host.container_index(tpath)
    t = host.ReadBatch()
    assert t.add_container_at(tpath, toffs).tolist() == [1, 0, 1] and t.size() == 2
    # truncated inside the last record: fatal, and the batch is left empty
    raw = open(path, "rb").read()
    bad = str(tmp_path / "bad.dnr"); open(bad, "wb").write(raw[: int(offs[-1]) + 40])
    e = host.ReadBatch()
    with pytest.raises(IOError):
        e.add_container_at(bad, offs)
    assert e.size() == 0
    with pytest.raises(IOError):
        host.container_index(str(tmp_path / "missing.dnr"))
