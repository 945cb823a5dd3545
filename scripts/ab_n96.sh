#!/bin/bash
# Developer probe: A/B (interleaved, same box) of library builds on the N96 workload.
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    EMS_LIB=$lib timeout -k 10 200 python bench.py --workload n96 --no-cpu-baseline --no-operational --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['avg_launch_ms'], d['value'], d['counters_match_golden'])"
  done
done
