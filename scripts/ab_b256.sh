#!/bin/bash
# Developer probe: A/B (interleaved, same box) of library builds on the headline code at batch 256 (one frame per CU).
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    EMS_LIB=$lib timeout -k 10 200 python bench.py --batch 256 --no-cpu-baseline --no-operational --no-other-workloads --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['avg_launch_ms'], d['value'], d['roofline']['kernel'][:40])"
  done
done
