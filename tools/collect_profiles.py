"""Turn gpurun_out/final/ (tools/final_profile.sh) into the committed summaries under profiles/ (tag from argv[1])."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src = "gpurun_out/final"
shutil.copy(src + "/stats/bench_kernel_stats.csv", "profiles/%s_kernel_stats.csv" % tag)
for n in ("bench_default", "bench_under_rocprof", "bench_full"):
    if os.path.exists("%s/%s.json" % (src, n)) and os.path.getsize("%s/%s.json" % (src, n)) > 2:
        shutil.copy("%s/%s.json" % (src, n), "profiles/%s_%s.json" % (tag, n))


def per_kernel(d):
    f = glob.glob(src + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
    return agg, n


w, nw = per_kernel("pmc_WRITE_SIZE"); f, nf = per_kernel("pmc_FETCH_SIZE"); ins, ni = per_kernel("pmc_insts")
rows = []
for k in sorted(w):
    if k.startswith("__amd") or "selftest" in k: continue
    rows.append((k, w[k]["WRITE_SIZE"] / nw[k], f[k]["FETCH_SIZE"] / nf[k]))
with open("profiles/%s_pmc_hbm_traffic.csv" % tag, "w") as o:
    o.write("# rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE (two separate passes), bench.py --steps 2 --warmup 1 --inflight 1, 1000 x 20 kb reads\n")
    o.write("# unit: KiB per launch, mean over launches; FETCH_SIZE on gfx950 reads 1/2 of wide coalesced streams (MI355X_MICROARCH.md, HBM): double it\n")
    o.write("kernel,WRITE_SIZE_KiB,FETCH_SIZE_KiB\n")
    for k, a, b in rows: o.write("%s,%.1f,%.1f\n" % (k, a, b))
fk = [r for r in rows if "k2_fill" in r[0]][0]
json.dump({"workload": {"reads": 1000, "bases": 20000}, "kernel": fk[0], "write_bytes": fk[1] * 1024.0, "fetch_bytes_raw": fk[2] * 1024.0,
           "fetch_bytes_corrected": fk[2] * 2048.0,
           "note": "WRITE_SIZE/FETCH_SIZE from rocprofv3 --pmc (separate passes); fetch doubled per the gfx950 FETCH_SIZE correction"},
          open("profiles/%s_pmc_k2_fill.json" % tag, "w"), indent=1)
with open("profiles/%s_pmc_instruction_mix.csv" % tag, "w") as o:
    o.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE, bench.py --steps 2 --warmup 1 --inflight 1; per launch\n")
    o.write("kernel,VALU,SALU,LDS,SMEM,waves,gui_active_cycles_per_xcd\n")
    for k in sorted(ins):
        if k.startswith("__amd") or "selftest" in k: continue
        c = ins[k]; m = ni[k]
        o.write("%s,%.4g,%.4g,%.4g,%.4g,%.0f,%.4g\n" % (k, c["SQ_INSTS_VALU"] / m, c["SQ_INSTS_SALU"] / m, c["SQ_INSTS_LDS"] / m, c["SQ_INSTS_SMEM"] / m,
                                                      c["SQ_WAVES"] / m, c["GRBM_GUI_ACTIVE"] / m / 8))
print(open("profiles/%s_pmc_hbm_traffic.csv" % tag).read()); print(open("profiles/%s_pmc_instruction_mix.csv" % tag).read())
