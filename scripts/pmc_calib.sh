#!/bin/bash
# Developer probe (GPU box): FETCH_SIZE / WRITE_SIZE of every mode of build/fetch_calib (csrc/probes/fetch_calib.hip) next to the bytes
# the mode knows it moved.  usage: scripts/pmc_calib.sh [log2 lanes]   -> gpurun_out/calib/summary.txt
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/calib
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/build/fetch_calib ${1:-24} > $O/plain.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o run -- $R/build/fetch_calib ${1:-24} > $O/$c.txt 2>&1
done
python3 - $O <<'PY'
import csv, glob, re, sys
O = sys.argv[1]
known = {int(m.group(1)): (float(m.group(2)), m.group(3)) for m in re.finditer(r"mode (\d+)\s+known_bytes (\d+)\s+[\d.]+ ms\s+[\d.]+ GB/s\s+(.*)", open(O + "/plain.txt").read())}
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (O, c), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_calib<(\d+)>", r["Kernel_Name"])
            if m and r["Counter_Name"] == c:
                val.setdefault(int(m.group(1)), {})[c] = float(r["Counter_Value"]) * 1024.0
with open(O + "/summary.txt", "w") as out:
    for mode in sorted(known):
        kb, what = known[mode]
        v = val.get(mode, {})
        line = "mode %d  known %.3f GB  FETCH_SIZE %.3f GB (%.3f x)  WRITE_SIZE %.3f GB (%.3f x)  %s" % (
            mode, kb / 1e9, v.get("FETCH_SIZE", 0) / 1e9, v.get("FETCH_SIZE", 0) / kb, v.get("WRITE_SIZE", 0) / 1e9, v.get("WRITE_SIZE", 0) / kb, what)
        print(line); out.write(line + "\n")
PY
