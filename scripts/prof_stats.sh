#!/bin/bash
# Developer probe (GPU box): rocprofv3 --kernel-trace --stats of one bench.py command -> gpurun_out/<tag>_stats/ (+ the bench line itself).
# usage: scripts/prof_stats.sh <tag> [bench args...]
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$(dirname $tag)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-operational "$@" > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
