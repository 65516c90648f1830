#!/bin/bash
# rocprofv3 kernel trace of 20 GroupNorm calls per shape: stats vs apply kernel time
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "16 4096 320 f32" "16 4096 320 bf16" "16 1024 640 f32" "16 256 1280 f32"; do
  rm -rf $R/gpurun_out/gn_trace
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gn_trace -o t -- python3 $R/experiments/gn_one.py $cfg > /dev/null 2>&1
  echo "== $cfg"
  python3 $R/experiments/gn_trace_print.py $R/gpurun_out/gn_trace
done
