#!/bin/bash
# Developer probe (GPU box): the first SQ counter set of one bench step -> per-visit instruction counts of the decode kernel.
# usage: scripts/pmc_sq.sh <tag> <visits per frame> <frames> [env VAR=val ...] -- [bench args...]
#        PMC_SET="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT ..." replaces the counter list (one pass: at most 8 SQ counters)
tag=$1; visits=$2; frames=$3; shift 3
envs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$(dirname $tag)
cd /tmp && export TMPDIR=/tmp
for e in "${envs[@]}"; do export "$e"; done
rocprofv3 --pmc ${PMC_SET:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY} --kernel-trace --output-format csv -d $R/gpurun_out/$tag -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operational --no-other-workloads "$@" > $R/gpurun_out/$tag.log 2>&1
python3 - $R/gpurun_out/$tag $visits $frames <<'PY'
import csv, glob, sys
d, visits, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
acc = {}; name = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"]:
            name = r["Kernel_Name"]; acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(name, {k: round(sum(v) / len(v) / (visits * frames), 1) for k, v in sorted(acc.items())})
PY
