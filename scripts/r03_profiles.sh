set -e
mkdir -p gpurun_out/r03v5
bash scripts/prof_stats.sh r03v5/r03_v5 --steps 5 --warmup 1
bash scripts/pmc_passes.sh r03v5/gf16 --workload gf16 --no-other-workloads
python scripts/pmc_summary.py gpurun_out/r03v5/gf16 256 117450 --json gpurun_out/r03v5/traffic_gf16.json > gpurun_out/r03v5/traffic_gf16.txt
