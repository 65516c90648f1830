#!/bin/bash
# SQ counter passes over one attention shape (experiments/attn_one.py); usage: pmc_attn.sh <d> <S> <tag> [debug word] [Skv]
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_$3a -- python3 $R/experiments/attn_one.py $1 $2 $4 $5 > $R/gpurun_out/pmc_$3a.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES -d $R/gpurun_out/pmc_$3b -- python3 $R/experiments/attn_one.py $1 $2 $4 $5 > $R/gpurun_out/pmc_$3b.log 2>&1
cd $R
python3 experiments/pmc_summary.py gpurun_out/pmc_$3a attn
python3 experiments/pmc_summary.py gpurun_out/pmc_$3b attn
