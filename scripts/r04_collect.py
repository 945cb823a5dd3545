"""Copies the summaries of scripts/r04_profiles.sh (gpurun_out/r04p/) into profiles/ (tracked): the bench line and the rocprofv3 kernel stats of
the same command, the decode-kernel rows of the four PMC passes of every workload, and the traffic_*.json summaries (with the kernel-source
hash bench.py checks).   usage: python scripts/r04_collect.py <tag, e.g. r04_v2>"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1]
O, P = os.path.join(ROOT, "gpurun_out", "r04p"), os.path.join(ROOT, "profiles")
h = bench.kernel_source_hash()
for w in ("gf256", "gf16", "n96", "gf256_v2"):
    j = json.load(open("%s/traffic_%s.json" % (O, w)))
    print(w, j["kernel"].split("::")[-1][:44], "hash ok" if j["kernel_source_sha16"] == h else "HASH DIFFERS", "%.1f MB per frame" % (j["hbm_bytes_per_frame"] / 1e6),
          {k: round(v, 1) for k, v in j["sq_per_visit"].items() if k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")})
    shutil.copy("%s/traffic_%s.json" % (O, w), "%s/traffic_%s.json" % (P, w))
shutil.copy(O + "/r04_v1_bench.json", "%s/%s_bench.json" % (P, tag))
shutil.copy(glob.glob(O + "/r04_v1_stats/**/*kernel_stats.csv", recursive=True)[0], "%s/%s_bench_kernel_stats.csv" % (P, tag))
for d, name in (("gf256", "gf256"), ("gf16", "gf16"), ("n96", "n96"), ("v2", "v2flavour")):
    for p_ in ("fetch", "write", "sq1", "sq2"):
        f = glob.glob("%s/%s_%s/**/*counter_collection.csv" % (O, d, p_), recursive=True)
        if f:
            lines = open(f[0]).read().split("\n")
            open("%s/%s_%s_pmc_%s.csv" % (P, tag, name, p_), "w").write("\n".join([lines[0]] + [l for l in lines[1:] if "k_decode" in l]) + "\n")
j = json.loads(open("%s/%s_bench.json" % (P, tag)).read().strip().splitlines()[-1])
print("headline", j["value"], "frames/s", j["ms_per_step"], "ms per step, golden", j["counters_match_golden"], "frac", j["roofline"]["frac"], "launch", j["roofline"]["avg_launch_ms"], "ms, traffic", j["roofline"]["traffic"])
for k, v in j["config"]["other_workloads"].items():
    print(" ", k, v["value"], "golden", v["counters_match_golden"], v["roofline"]["bound"], v["roofline"]["frac"], v["roofline"]["avg_launch_ms"], "ms, traffic", v["roofline"].get("traffic"))
