#!/bin/bash
# Developer probe (GPU box): the headline step under pair orders 1 / 2 (EMS_PAIR_ORDER) x non-temporal / plain row accesses (build/libems_plain.so),
# interleaved on one box.  usage: scripts/ab_order.sh <rounds>
for r in $(seq $1); do
  for ord in 1 2; do
    for lib in "" build/libems_plain.so; do
      EMS_PAIR_ORDER=$ord EMS_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-operational --no-other-workloads --synthetic-zero-codeword --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('order $ord lib ${lib:-product}', d['roofline']['avg_launch_ms'], d['value'])"
    done
  done
done
