#!/bin/bash
# Developer probe (run on the GPU box through gpurun): rocprofv3 PMC passes over one bench step.
# usage: scripts/pmc_passes.sh <tag> [bench args...]   -> gpurun_out/<tag>_{fetch,write,sq1,sq2}/
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$1 -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operational "${@:3}" > $R/gpurun_out/${tag}_$1.log 2>&1
}
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "$@"
run sq2 "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" "$@"
