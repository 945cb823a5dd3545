#!/bin/bash
# Developer probe (GPU box): the GF(16) bench step on library builds, interleaved.  usage: scripts/ab_q16t.sh <rounds> <lib> [<lib> ...] ("" = product)
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    EMS_LIB=$lib timeout -k 10 200 python bench.py --workload gf16 --no-cpu-baseline --no-operational --no-other-workloads --synthetic-zero-codeword --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib ${lib:-product}', d['roofline']['kernel'][:40], d['roofline']['avg_launch_ms'], d['value'])"
  done
done
