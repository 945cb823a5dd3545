#!/bin/bash
# Evidence run of a round (GPU box, through gpurun): rocprofv3 --kernel-trace --stats of the default bench command, then the four PMC
# passes (FETCH_SIZE | WRITE_SIZE | two SQ sets, each in its own run, as MI355X_MICROARCH.md prescribes) of every workload whose
# `roofline.traffic` / `roofline.valu` the bench line prints; the summaries carry the kernel-source hash bench.py checks.
#   scripts/profiles.sh <tag, e.g. r05_v1> [workload ...]      -> gpurun_out/<tag>/ ; then scripts/collect_profiles.py <tag> copies into profiles/
set -e
tag=$1; shift
wls=${@:-gf256 gf16 n96 gf256_v2 irr345 dc20}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-operational --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "stats done" > $O/progress.log
cd $R
for w in $wls; do
  case $w in
    gf256)    B=1536; V=117450; args="--no-other-workloads" ;;       # visits per frame = M x sweeps
    gf16)     B=256;  V=117450; args="--workload gf16 --no-other-workloads" ;;
    n96)      B=8192; V=232;    args="--workload n96 --no-other-workloads" ;;
    gf256_v2) B=1536; V=117450; args="--workload gf256_v2" ;;
    irr345)   B=256;  V=117450; args="--workload irr345 --no-other-workloads" ;;
    dc20)     B=256;  V=23490;  args="--workload dc20 --no-other-workloads" ;;
  esac
  bash scripts/pmc_passes.sh $tag/$w $args
  python scripts/pmc_summary.py $O/$w $B $V --json $O/traffic_$w.json > $O/traffic_$w.txt
  echo "$w done" >> $O/progress.log
done
