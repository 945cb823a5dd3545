#!/bin/bash
# Developer probe: A/B (interleaved, same box) of library builds on the headline bench.
# usage: scripts/ab.sh <rounds> <libA> <libB> [...]; prints decode-kernel ms per launch and frames/s of each run
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    EMS_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-operational --no-other-workloads --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['avg_launch_ms'], d['value'])"
  done
done
