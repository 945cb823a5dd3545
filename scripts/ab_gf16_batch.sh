#!/bin/bash
# Developer probe: the GF(16) workload at other batch sizes / workgroup shapes: ab_gf16_batch.sh "BATCH ENV=.." ...
for cfg in "$@"; do
  set -- $cfg; b=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload gf16 --batch $b --no-cpu-baseline --no-operational --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['roofline']['avg_launch_ms'], d['value'], d['roofline']['kernel'][:90])"
done
