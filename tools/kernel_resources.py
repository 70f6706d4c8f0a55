"""Per-kernel resource table (VGPRs, AGPRs, SGPRs, spills, scratch, LDS, occupancy) of the product kernels, from the compiler's own
remarks (-Rpass-analysis=kernel-resource-usage) with build.py's flags.  CPU only: hipcc cross-compiles gfx950.
    python tools/kernel_resources.py [source.hip ...]  > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dnascent_amd import build as b
srcs = sys.argv[1:] or [os.path.join(b.CSRC, s) for s in b.HIP_SOURCES]
rows = []
for src in srcs:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-I", os.path.join(ROOT, "include"), "-I", b.CSRC, "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src]
    cmd += os.environ.get("DN_HIPCC_FLAGS", "").split()
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = dict(name=subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]); rows.append(cur); continue
        m = re.search(r"remark: .*?\s+([A-Za-z ]+\[?[A-Za-z/ ]*\]?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
print("%-58s %5s %5s %5s %6s %6s %8s %8s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "scratchB", "LDS B", "occ"))
for r in rows:
    print("%-58s %5d %5d %5d %6d %6d %8d %8d %4d" % (r["name"][:58], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs", -1), r.get("VGPRs Spill", r.get("VGPR Spill", -1)),
                                                     r.get("SGPRs Spill", r.get("SGPR Spill", -1)), r.get("ScratchSize [bytes/lane]", -1), r.get("LDS Size [bytes/block]", -1),
                                                     r.get("Occupancy [waves/SIMD]", -1)))
