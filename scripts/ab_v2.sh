#!/bin/bash
# Developer probe: A/B (interleaved, same box) of library builds on the v2 flavour's bench.
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    EMS_LIB=$lib timeout -k 10 300 python bench.py --workload gf256_v2 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['avg_launch_ms'], d['value'])"
  done
done
