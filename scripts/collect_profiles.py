"""Copies the summaries of scripts/profiles.sh (gpurun_out/<tag>/) into profiles/ (tracked): the bench line and the rocprofv3 kernel stats of
the same command, the decode-kernel rows of the four PMC passes of every workload, and the traffic_*.json summaries (with the kernel-source
hash bench.py checks).   usage: python scripts/collect_profiles.py <tag, e.g. r05_v1>"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1]
O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
h = bench.kernel_source_hash()
for tj in sorted(glob.glob(O + "/traffic_*.json")):
    w = os.path.basename(tj)[len("traffic_"):-len(".json")]
    j = json.load(open(tj))
    if not j.get("kernel"):
        print(w, "no decode kernel in the passes: skipped"); continue
    print(w, j["kernel"].split("::")[-1][:44], "hash ok" if j["kernel_source_sha16"] == h else "HASH DIFFERS", "%.1f MB per frame" % (j.get("hbm_bytes_per_frame", 0) / 1e6),
          {k: round(v, 1) for k, v in j["sq_per_visit"].items() if k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")})
    shutil.copy(tj, "%s/traffic_%s.json" % (P, w))
    for p_ in ("fetch", "write", "sq1", "sq2"):
        f = glob.glob("%s/%s_%s/**/*counter_collection.csv" % (O, w, p_), recursive=True)
        if f:
            lines = open(f[0]).read().split("\n")
            open("%s/%s_%s_pmc_%s.csv" % (P, tag, w, p_), "w").write("\n".join([lines[0]] + [l for l in lines[1:] if "k_decode" in l]) + "\n")
if os.path.exists(O + "/bench.json"):
    shutil.copy(O + "/bench.json", "%s/%s_bench.json" % (P, tag))
    st = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)
    if st: shutil.copy(st[0], "%s/%s_bench_kernel_stats.csv" % (P, tag))
    j = json.loads(open("%s/%s_bench.json" % (P, tag)).read().strip().splitlines()[-1])
    rf = j["roofline"]
    print("headline", j["value"], "frames/s", j["ms_per_step"], "ms per step, golden", j["counters_match_golden"], "hbm frac", rf["frac"], "valu", rf.get("valu", {}).get("frac"), "launch", rf["avg_launch_ms"], "ms, traffic", rf["traffic"])
    for k, v in j["config"].get("other_workloads", {}).items():
        r = v["roofline"]
        print(" ", k, v["value"], "golden", v["counters_match_golden"], r["bound"], "hbm", r.get("hbm", {}).get("frac"), "valu", r.get("valu", {}).get("frac"), r["avg_launch_ms"], "ms, traffic", r.get("traffic"))
