"""GPU box probe: does the host NUMA node a pageable buffer was first touched on change what a blocking H2D copy of it costs?
(run_detect's upload_s was 0.7 s in some runs and 4.3 s in others on the same box, same command: profiles/r05_run_detect_stats*.json)
    python tools/numa_probe.py"""
import ctypes as C, glob, os, time
import numpy as np

def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out

nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
print("NUMA nodes:", {k: "%d cpus (%d..%d)" % (len(v), v[0], v[-1]) for k, v in nodes.items() if v})
print("allowed cpus:", len(os.sched_getaffinity(0)))
for f in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")) + sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
    try:
        t = open(f).read()
        if f.endswith("numa_node"):
            print(f, t.strip())
    except OSError:
        pass
allowed0 = os.sched_getaffinity(0)
hip = C.CDLL("libamdhip64.so")
dev = C.c_void_p()
n = 1 << 30
assert hip.hipMalloc(C.byref(dev), C.c_size_t(n)) == 0
for rep in range(2):
    for node, cpus in nodes.items():
        use = set(cpus) & allowed0
        if not use:
            continue
        os.sched_setaffinity(0, use)
        buf = np.empty(n, np.uint8); buf[:] = 1                     # first touch on this node
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            assert hip.hipMemcpy(dev, C.c_void_p(buf.ctypes.data), C.c_size_t(n), 1) == 0
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); b2 = buf.copy(); tc = time.perf_counter() - t0
        print("buffer touched on node %d, copying thread on node %d: pageable H2D %.1f GB/s (best of 3; %s), host memcpy %.1f GB/s" % (node, node, n / min(ts) / 1e9, " ".join("%.0f ms" % (t * 1e3) for t in ts), n / tc / 1e9))
        # cross: copy issued from the OTHER node's cpus
        for other, oc in nodes.items():
            ou = set(oc) & allowed0
            if other != node and ou:
                os.sched_setaffinity(0, ou)
                t0 = time.perf_counter(); assert hip.hipMemcpy(dev, C.c_void_p(buf.ctypes.data), C.c_size_t(n), 1) == 0; t1 = time.perf_counter() - t0
                print("    same buffer, copying thread on node %d: %.1f GB/s" % (other, n / t1 / 1e9))
        del buf, b2
os.sched_setaffinity(0, allowed0)
