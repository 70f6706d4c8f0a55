"""Build the native libraries in-tree (no JIT cache: the .so files travel with the gpurun snapshot).

  dnascent_amd/lib/libdnascent_hip.so   HIP kernels + the C-ABI of include/dnascent_hip.h   (hipcc, gfx950)
  dnascent_amd/lib/libdnascent_host.so  host side above the C-ABI (g++): synthetic generator, read
                                        model, CIGAR flattening, .detect writer, batch driver
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIP_SO = os.path.join(LIB, "libdnascent_hip.so")
HOST_SO = os.path.join(LIB, "libdnascent_host.so")

HIP_SOURCES = ["dn_capi.hip", "k1_segment.hip", "k2_banded.hip", "k_scaling.hip", "k2b_viterbi.hip", "k3_cnn.hip", "k_hmm.hip", "k_collect.hip"]
HOST_SOURCES = ["host/dn_synth.c", "host/dn_host.cpp", "host/dn_bam.cpp", "host/dn_vbz.cpp"]       # dn_bam.cpp: BGZF / BAM over zlib (libz is in the image; htslib is not); dn_vbz.cpp: POD5's signal codec (zstd through dlopen)


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _all_deps(srcs):
    deps = list(srcs)
    for d, _, files in os.walk(CSRC):
        deps += [os.path.join(d, f) for f in files if f.endswith((".h", ".hpp", ".cuh"))]
    deps.append(os.path.join(ROOT, "include", "dnascent_hip.h"))
    return deps


def build_hip(force=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _newer(HIP_SO, _all_deps(srcs)):
        return HIP_SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -ffp-contract=off: the reference arithmetic has no FMA contraction (Makefile:6 plain -O2 on x86-64);
    # every fused multiply-add in the kernels is written explicitly.
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
           "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", HIP_SO] + os.environ.get("DN_HIPCC_FLAGS", "").split() + srcs   # DN_HIPCC_FLAGS: experiment builds (-DDN_WS_TRACE=<workgroup>)
    _run(cmd)
    return HIP_SO


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    build_hip(False)
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _newer(HOST_SO, _all_deps(srcs)):
        return HOST_SO
    objs = []
    for s in srcs:
        o = os.path.join(LIB, os.path.basename(s) + ".o")
        if s.endswith(".c"):
            _run(["gcc", "-std=c99", "-O2", "-fPIC", "-ffp-contract=off", "-Wall", "-I", CSRC, "-c", s, "-o", o])
        else:
            _run(["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fopenmp", "-Wall",
                  "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", s, "-o", o])
        objs.append(o)
    # the host side calls the C-ABI: link it against libdnascent_hip.so sitting next to it
    _run(["g++", "-shared", "-fopenmp", "-o", HOST_SO] + objs +
         ["-L", LIB, "-ldnascent_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-ldl", "-lz"])
    return HOST_SO


def build_io(force=False):
    """OPTIONAL: BAM / POD5 ingestion through htslib and libpod5 (contrib/dn_io_htslib.cpp: never compiled in this image) -> lib/libdnascent_io.so.  Built only
    where the headers exist: DN_HTSLIB_INC / DN_POD5_INC (or /usr/include/htslib/sam.h, /usr/include/pod5_format/c_api.h).  Neither
    library is in this image; the function then returns None and the binary read container remains the ingestion path."""
    hts = os.environ.get("DN_HTSLIB_INC") or ("/usr/include" if os.path.exists("/usr/include/htslib/sam.h") else None)
    pod = os.environ.get("DN_POD5_INC") or ("/usr/include" if os.path.exists("/usr/include/pod5_format/c_api.h") else None)
    if not hts and not pod:
        return None
    out = os.path.join(LIB, "libdnascent_io.so")
    src = os.path.join(ROOT, "contrib", "dn_io_htslib.cpp")
    if not force and not _newer(out, [src]):
        return out
    build_host(False)
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", out, src]
    if hts:
        cmd += ["-DDN_WITH_HTSLIB", "-I", hts]
    if pod:
        cmd += ["-DDN_WITH_POD5", "-I", pod]
    cmd += ["-L", LIB, "-ldnascent_host", "-Wl,-rpath,$ORIGIN"] + (["-lhts"] if hts else []) + (["-lpod5_format"] if pod else [])
    _run(cmd)
    return out


def build_oracle():
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_all(force=False):
    build_hip(force)
    build_host(force)
    build_io(force)          # no-op where htslib / libpod5 are absent (this image)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
