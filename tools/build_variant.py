"""Experiment build of libdnascent_hip.so with extra hipcc flags into tools/_bin/lib_<name>/ (swapped in on the GPU box for same-session A / B runs):
    python tools/build_variant.py <name> -DK2B_W=4 -DK2_FILL_W=4"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dnascent_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "_bin", "lib_" + name)
os.makedirs(out, exist_ok=True)
srcs = [os.path.join(b.CSRC, s) for s in b.HIP_SOURCES]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value",
       "-Wno-unused-result", "-fhip-fp32-correctly-rounded-divide-sqrt"] + flags + ["-I", os.path.join(ROOT, "include"), "-I", b.CSRC, "-o", os.path.join(out, "libdnascent_hip.so")] + srcs
r = subprocess.run(cmd, capture_output=True, text=True)
print(name, "built" if r.returncode == 0 else "FAILED\n" + r.stderr[-800:])
sys.exit(r.returncode)
