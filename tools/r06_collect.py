"""Turn gpurun_out/r06/ (tools/r06_profile.sh) into the committed summaries under profiles/ (r06_*):
  r06_bench_default.json / _under_rocprof.json / _banded.json      the bench lines
  r06_kernel_stats.csv                                            rocprofv3 --stats of the bench
  r06_pmc_bench.json, r06_pmc_banded.json                         what bench.py reads back (load_pmc): HBM bytes per launch of every kernel and
                                                                  per step (WRITE_SIZE + 2 x FETCH_SIZE: MI355X_MICROARCH.md, HBM), vector / MFMA
                                                                  instruction counts, MFMA busy cycles; taken at the bench's own launch shape
  r06_pmc_k3_mfma.json                                            MfmaUtil of every K3 kernel of the round (SQ_VALU_MFMA_BUSY_CYCLES)
  r06_k3_layers.txt, r06_k3_math_modes.txt                        K3 alone, layer by layer"""
import collections, csv, glob, json, os, shutil, sys
src = "gpurun_out/r06"


def counters(d):
    """{kernel: {counter: sum}}, {kernel: launches} of one counter pass"""
    fs = glob.glob(src + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        return None, None
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("__amd") or "selftest" in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    return agg, n


def build(tag, line, batches_per_run):
    """line: the bench JSON of the counter run (workload); batches_per_run: batches the run pushed through the pipeline"""
    w, nw = counters("pmc_%s_WRITE_SIZE" % tag); f, nf = counters("pmc_%s_FETCH_SIZE" % tag); m, nm = counters("pmc_%s_MFMA" % tag)
    if not w or not f:
        return None
    out = {"kernels": {}, "note": "rocprofv3 --pmc, separate passes (WRITE_SIZE | FETCH_SIZE | SQ_*), submitted at the workload's own depth (full scope: 8 batches in flight; the counter collection itself runs one dispatch at a time); bytes = KiB counters x 1024; FETCH_SIZE doubled "
                                  "(gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM); per launch = mean over the launches of the pass"}
    tw = tf = 0.0
    for k in sorted(w):
        wb = w[k]["WRITE_SIZE"] * 1024.0; fb = f.get(k, {}).get("FETCH_SIZE", 0.0) * 2048.0
        tw += wb; tf += fb
        e = {"launches": nw[k], "write_bytes_per_launch": wb / nw[k], "fetch_bytes_per_launch_corrected": fb / max(1, nf.get(k, 0))}
        if m and k in m:
            c = m[k]
            e["valu_insts_per_launch"] = c.get("SQ_INSTS_VALU", 0.0) / nm[k]
            e["mfma_insts_per_launch"] = c.get("SQ_INSTS_MFMA", 0.0) / nm[k]
            if c.get("GRBM_GUI_ACTIVE", 0) > 0:
                e["mfma_util"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        out["kernels"][k] = e
    out["step"] = {"write_bytes": tw / batches_per_run, "fetch_bytes_corrected": tf / batches_per_run, "batches_in_the_counter_run": batches_per_run}
    if m:
        busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in m.values()); act = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for c in m.values())
        if act > 0:
            out["step"]["mfma_util"] = busy / (act / 8.0 * 1024.0)
    kf = [k for k in out["kernels"] if k.startswith("k2_fill")]
    if kf:
        out["k2_fill"] = dict(out["kernels"][kf[0]], kernel=kf[0])
    return out


def copy(n, dst=None):
    p = "%s/%s" % (src, n)
    if os.path.exists(p) and os.path.getsize(p) > 2:
        shutil.copy(p, "profiles/r06_" + (dst or n))
        return True
    return False


for n in ("bench_default.json", "bench_under_rocprof.json", "bench_banded.json", "k3_layers.txt", "k3_math_modes.txt", "bench_mixed.json", "bench_mixed_longfirst.json",
          "run_detect_stats.json", "run_detect.log", "run_detect_stats_warm.json", "run_detect_warm.log", "run_detect_2ranks_gloo_stats.json", "run_detect_2ranks.log",
          "run_detect_1rank_sha.log", "run_detect_slow_exit.log", "pytest_gpu.log", "bench_default.log"):
    copy(n)
st = glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True)
if st:
    shutil.copy(st[0], "profiles/r06_kernel_stats.csv")
# ---- full scope: --steps 8 --warmup 1 --inflight 6 + the untimed solo batch = 10 batches through the whole pipeline (the set-up uploads run nothing)
bd = json.load(open(src + "/bench_default.json")) if os.path.exists(src + "/bench_default.json") else None
full = build("full", bd, 10)
if full and bd:
    full["workload"] = {"reads_per_step": bd["config"]["reads_per_step"], "bases": bd["config"]["bases_per_read"], "cnn_math": "f16x3", "inflight": 6, "steps": 8,
                        "command": "rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --steps 8 --warmup 1 --inflight 6 --no-cpu-baseline --fp32-steps 0 --bf16-steps 0"}
    json.dump(full, open("profiles/r06_pmc_bench.json", "w"), indent=1)
    k3 = {k: v for k, v in full["kernels"].items() if k.startswith("k3_")}
    json.dump({"command": full["workload"]["command"].replace("<counter>", "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"),
               "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), per kernel over all its launches of the pass (500 x 50 kb reads per batch, "
                       "4 Mi-row CNN passes: the bench's launch shape)", "kernels": {k: {"launches": v["launches"], "MfmaUtil": v.get("mfma_util"),
                       "mfma_insts_per_launch": v.get("mfma_insts_per_launch"), "valu_insts_per_launch": v.get("valu_insts_per_launch")} for k, v in k3.items()},
               "whole_step_MfmaUtil": full["step"].get("mfma_util")}, open("profiles/r06_pmc_k3_mfma.json", "w"), indent=1)
    for k, v in sorted(k3.items(), key=lambda kv: -(kv[1]["write_bytes_per_launch"] + kv[1]["fetch_bytes_per_launch_corrected"]) * kv[1]["launches"])[:12]:
        print("%-46s x%-4d  W %8.1f MB  F %8.1f MB  MfmaUtil %s" % (k[:46], v["launches"], v["write_bytes_per_launch"] / 1e6, v["fetch_bytes_per_launch_corrected"] / 1e6,
                                                                    None if v.get("mfma_util") is None else round(v["mfma_util"], 3)))
    print("step: HBM %.1f GB (W %.1f + F %.1f), MfmaUtil %s" % ((full["step"]["write_bytes"] + full["step"]["fetch_bytes_corrected"]) / 1e9, full["step"]["write_bytes"] / 1e9,
                                                               full["step"]["fetch_bytes_corrected"] / 1e9, full["step"].get("mfma_util")))
# ---- banded scope: --steps 2 --warmup 1 (warm-up runs max(1, inflight) steps) + the solo step = 4 normalise passes of the resident batch
bb = json.load(open(src + "/bench_banded.json")) if os.path.exists(src + "/bench_banded.json") else None
band = build("banded", bb, 4)
if band and bb:
    band["workload"] = {"reads_per_step": bb["config"]["reads_per_gpu"], "bases": bb["config"]["bases_per_read"], "inflight": 1, "steps": 2,
                        "command": "rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --scope banded --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline"}
    json.dump(band, open("profiles/r06_pmc_banded.json", "w"), indent=1)
    kf = band.get("k2_fill")
    if kf:
        print("k2_fill: W %.2f GB F %.2f GB per launch, %.3g vector instructions" % (kf["write_bytes_per_launch"] / 1e9, kf["fetch_bytes_per_launch_corrected"] / 1e9,
                                                                                      kf.get("valu_insts_per_launch", 0)))
for n in ("bench_default", "bench_banded"):
    p = "profiles/r06_%s.json" % n
    if os.path.exists(p):
        d = json.load(open(p))
        print(n, round(d["value"], 1), d["unit"], "| roofline", d["roofline"]["kernel"][:40], round(d["roofline"]["frac"], 4), "| cpu", d.get("cpu_baseline", {}).get("value"))
