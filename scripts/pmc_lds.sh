#!/bin/bash
# Developer probe (GPU box): LDS counters of the decode kernel for several library builds.
# usage: scripts/pmc_lds.sh <workload> <visits per frame> <frames> lib...   -> gpurun_out/pmclds/<lib>.txt
set -e
wl=$1; visits=$2; frames=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmclds
for lib in "$@"; do
  n=$(basename $lib .so)
  export EMS_LIB=$R/$lib
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmclds/$n -o run -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-operational --no-other-workloads > $R/gpurun_out/pmclds/$n.log 2>&1
  python3 - $R/gpurun_out/pmclds/$n $visits $frames $n <<'PY'
import csv, glob, sys
d, visits, frames, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(n, {k: round(sum(v) / len(v) / (visits * frames), 1) for k, v in sorted(acc.items())})
PY
done
