#!/bin/bash
# SQ counters of the ping-pong conv kernel vs the ring kernel (one pass each)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pp in 0 3; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_pp$pp -o out --output-format csv -- python3 $R/experiments/pp_one.py $pp > $R/gpurun_out/pmc_pp$pp.log 2>&1 || exit 1
done
