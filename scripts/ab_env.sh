#!/bin/bash
# Developer probe: A/B (interleaved, same box) of environment settings on a bench workload.
# usage: scripts/ab_env.sh <rounds> <workload> "<VAR=val ...>" "<VAR=val ...>" ...
rounds=$1; wl=$2; shift 2
for r in $(seq $rounds); do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-operational --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', d['roofline']['avg_launch_ms'], d['value'])"
  done
done
