#!/bin/bash
# Round-4 evidence run (GPU box): rocprofv3 kernel stats of the default bench command, then the four PMC passes of every workload whose
# `roofline.traffic` the bench line prints (the summaries carry the kernel-source hash bench.py checks).  Outputs under gpurun_out/r04p/.
set -e
O=gpurun_out/r04p
mkdir -p $O
bash scripts/prof_stats.sh r04p/r04_v1 --steps 5 --warmup 1
echo "stats done" > $O/progress.log
bash scripts/pmc_passes.sh r04p/gf256 --no-other-workloads
python scripts/pmc_summary.py $O/gf256 1536 117450 --json $O/traffic_gf256.json > $O/traffic_gf256.txt
echo "gf256 done" >> $O/progress.log
bash scripts/pmc_passes.sh r04p/gf16 --workload gf16 --no-other-workloads
python scripts/pmc_summary.py $O/gf16 256 117450 --json $O/traffic_gf16.json > $O/traffic_gf16.txt
echo "gf16 done" >> $O/progress.log
bash scripts/pmc_passes.sh r04p/n96 --workload n96 --no-other-workloads
python scripts/pmc_summary.py $O/n96 8192 232 --json $O/traffic_n96.json > $O/traffic_n96.txt
echo "n96 done" >> $O/progress.log
bash scripts/pmc_passes.sh r04p/v2 --workload gf256_v2
python scripts/pmc_summary.py $O/v2 1536 117450 --json $O/traffic_gf256_v2.json > $O/traffic_gf256_v2.txt
echo "v2 done" >> $O/progress.log
