"""GPU box probe: what a blocking pageable H2D copy of a 600 MB buffer costs the calling thread -- fresh 4 KB-page buffer, the same buffer again, a buffer
the kernel was asked to back with huge pages (madvise MADV_HUGEPAGE; the pool's hosts run THP in 'madvise' mode), a page-locked one.  run_detect's uploads are
copies of the first kind and cost 20 ms a batch on some boxes and 150 ms on others (gpurun_out/r5r, r5s).
    python tools/h2d_probe.py"""
import ctypes as C, mmap, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
libc = C.CDLL("libc.so.6", use_errno=True)
n = 600 << 20
dev = C.c_void_p(); assert hip.hipMalloc(C.byref(dev), C.c_size_t(n)) == 0
stream = C.c_void_p(); assert hip.hipStreamCreate(C.byref(stream)) == 0
def copy(ptr, asyn=False):
    t0 = time.perf_counter()
    if asyn:
        assert hip.hipMemcpyAsync(dev, C.c_void_p(ptr), C.c_size_t(n), 1, stream) == 0
        t1 = time.perf_counter(); hip.hipStreamSynchronize(stream)
        return (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3
    assert hip.hipMemcpy(dev, C.c_void_p(ptr), C.c_size_t(n), 1) == 0
    return (time.perf_counter() - t0) * 1e3
warm = np.ones(1 << 20, np.uint8); assert hip.hipMemcpy(dev, C.c_void_p(warm.ctypes.data), C.c_size_t(1 << 20), 1) == 0
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
for rep in range(3):
    a = np.empty(n, np.uint8); a[:] = rep
    t1 = copy(a.ctypes.data); t2 = copy(a.ctypes.data)
    ta = copy(a.ctypes.data, True)
    m = mmap.mmap(-1, n + (2 << 20)); addr = C.addressof(C.c_char.from_buffer(m)); al = (addr + (2 << 20) - 1) & ~((2 << 20) - 1)
    rc = libc.madvise(C.c_void_p(al), C.c_size_t(n), 14)                         # MADV_HUGEPAGE
    C.memset(C.c_void_p(al), rep + 1, n)
    t3 = copy(al); t4 = copy(al)
    print("rep %d: fresh 4 KB pages %.0f ms, again %.0f ms, async: call returns after %.0f ms, done after %.0f | MADV_HUGEPAGE (rc %d) %.0f ms, again %.0f ms" % (rep, t1, t2, ta[0], ta[1], rc, t3, t4))
    del a
hp = C.c_void_p(); t0 = time.perf_counter(); assert hip.hipHostMalloc(C.byref(hp), C.c_size_t(n), 0) == 0; tm = (time.perf_counter() - t0) * 1e3
C.memset(hp, 3, n); t5 = copy(hp.value); ta = copy(hp.value, True)
print("page-locked: hipHostMalloc %.0f ms, copy %.0f ms, async: call returns after %.1f ms, done after %.0f" % (tm, t5, ta[0], ta[1]))
print("AnonHugePages now:", [l.split()[1] for l in open("/proc/meminfo") if l.startswith("AnonHugePages")][0], "kB")
