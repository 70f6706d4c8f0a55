"""Turn gpurun_out/r02/ (tools/r02_profile.sh) into the committed summaries under profiles/ (r02_*)."""
import collections, csv, glob, json, os, shutil
src = "gpurun_out/r02"
shutil.copy(src + "/stats/bench_kernel_stats.csv", "profiles/r02_kernel_stats.csv")
for n in ("bench_default", "bench_under_rocprof", "bench_banded", "bench_fp32"):
    if os.path.exists("%s/%s.json" % (src, n)) and os.path.getsize("%s/%s.json" % (src, n)) > 2:
        shutil.copy("%s/%s.json" % (src, n), "profiles/r02_%s.json" % n)


for n in ("k3_layers.txt", "k3_math_modes.txt"):
    if os.path.exists(src + "/" + n):
        shutil.copy(src + "/" + n, "profiles/r02_" + n)


def per_kernel(d, short=True):
    f = glob.glob(src + "/" + d + "/*counter_collection.csv")[0]
    agg = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    return agg, n


w, nw = per_kernel("pmc_WRITE_SIZE"); f, nf = per_kernel("pmc_FETCH_SIZE")
rows = [(k, w[k] / nw[k], f[k] / nf[k]) for k in sorted(w) if not k.startswith("__amd") and "selftest" not in k]
k1 = sum((a + 2 * b) for k, a, b in rows if k.startswith("k1_")) * 1024
k2 = sum((a + 2 * b) for k, a, b in rows if k.startswith("k2_") or k == "k_prep") * 1024
bd = json.load(open(src + "/bench_banded.json"))
alg = bd["roofline"]["algorithmic_bytes_per_launch"]
samples = bd["config"]["samples_per_gpu_step"]
with open("profiles/r02_pmc_hbm_traffic.csv", "w") as o:
    o.write("# rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE (two separate passes), bench.py --scope banded --steps 2 --warmup 1 --inflight 1: 1000 x 20 kb reads, %d samples\n" % samples)
    o.write("# unit: KiB per launch, mean over launches; FETCH_SIZE on gfx950 reads 1/2 of wide coalesced streams (MI355X_MICROARCH.md, HBM): the corrected column doubles it\n")
    o.write("kernel,WRITE_SIZE_KiB,FETCH_SIZE_KiB,total_corrected_MB\n")
    for k, a, b in rows:
        o.write("%s,%.1f,%.1f,%.1f\n" % (k, a, b, (a + 2 * b) / 1024))
    o.write("# K1 (k1_scan4 + k1_detect + k1_events): %.2f GB per launch = %.1f bytes per sample (round 1: 20.8 GB, 90 B/sample; algorithmic 3.5 B/sample)\n" % (k1 / 1e9, k1 / samples))
    o.write("# K2 stage (k_prep + k2_fill6 + k2_chase + k2_expand + k2_post): %.2f GB per launch = %.2f x the algorithmic %.2f GB (round 1: 25.2 GB, 3.6 x)\n" % (k2 / 1e9, k2 / alg, alg / 1e9))
fk = [r for r in rows if "k2_fill" in r[0]][0]
json.dump({"workload": {"reads": 1000, "bases": 20000}, "kernel": fk[0], "write_bytes": fk[1] * 1024.0, "fetch_bytes_raw": fk[2] * 1024.0,
           "fetch_bytes_corrected": fk[2] * 2048.0,
           "note": "WRITE_SIZE/FETCH_SIZE from rocprofv3 --pmc (separate passes); fetch doubled per the gfx950 FETCH_SIZE correction"},
          open("profiles/r02_pmc_k2_fill.json", "w"), indent=1)
print(open("profiles/r02_pmc_hbm_traffic.csv").read())
for n in ("bench_default", "bench_banded", "bench_fp32"):
    d = json.load(open("profiles/r02_%s.json" % n))
    print(n, round(d["value"], 1), d["unit"], "| roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), "| cpu", d.get("cpu_baseline", {}).get("value"))

# ---- K3 alone (tools/gpu_cnn_time.py 64 20000: 1.2 M positions): HBM traffic per kernel, summed over the launches of ONE network pass ----
if glob.glob(src + "/k3pmc_WRITE_SIZE/*counter_collection.csv"):
    def per_kernel_sum(d):
        f = glob.glob(src + "/" + d + "/*counter_collection.csv")[0]
        agg = collections.defaultdict(float); n = collections.Counter(); seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith("k3_"):
                continue
            agg[k] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); n[k] += 1
        return agg, n
    w3, n3 = per_kernel_sum("k3pmc_WRITE_SIZE"); f3, m3 = per_kernel_sum("k3pmc_FETCH_SIZE")
    passes = 5.0            # gpu_cnn_time.py runs the network once to warm up and 4 more times (1 + 3 timed + 1 first)
    with open("profiles/r02_pmc_k3_traffic.csv", "w") as o:
        o.write("# rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes), tools/gpu_cnn_time.py 64 20000 f16x3: K3 alone, 1 200 057 positions per network pass\n")
        o.write("# MB per launch (mean); FETCH corrected x 2 (gfx950); layer I/O = rows x (cin + cout) x 4 B\n")
        o.write("kernel,launches,WRITE_MB_per_launch,FETCH_MB_per_launch_corrected\n")
        for k in sorted(w3, key=lambda k: -(w3[k] + 2 * f3.get(k, 0))):
            o.write("%s,%d,%.1f,%.1f\n" % (k, n3[k], w3[k] * 1024 / n3[k] / 1e6, 2 * f3.get(k, 0) * 1024 / max(1, m3.get(k, 1)) / 1e6))
    print(open("profiles/r02_pmc_k3_traffic.csv").read())
