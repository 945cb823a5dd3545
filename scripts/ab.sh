#!/bin/bash
# Developer probe (GPU box): A/B of library builds or environment settings on one bench workload, interleaved on the same box.
#   scripts/ab.sh <rounds> <workload> <libA.so> <libB.so> [...]          libraries (EMS_LIB)
#   scripts/ab.sh <rounds> <workload> "VAR=val ..." "VAR=val ..." [...]   environments (an argument with '=' is an environment)
# workload: gf256 | gf16 | n96 | gf256_v2 | irr345 | dc20; prints decode-kernel ms per launch, frames/s and the golden check of every run.
# EMS_AB_ARGS adds bench arguments (default steps / warmup: 4 / 1; n96: 10 / 3; --batch N for other shapes).
rounds=$1; wl=$2; shift 2
steps="--steps 4 --warmup 1"; [ "$wl" = n96 ] && steps="--steps 10 --warmup 3"; [ "$wl" = gf256_v2 ] && steps="--steps 3 --warmup 1"
for r in $(seq $rounds); do
  for x in "$@"; do
    if [[ "$x" == *=* ]]; then pre="env $x"; else pre="env EMS_LIB=$x"; fi
    $pre timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-operational --no-other-workloads $steps $EMS_AB_ARGS 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$x]', d['roofline']['avg_launch_ms'], d['value'], d.get('counters_match_golden'))"
  done
done
