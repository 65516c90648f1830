#!/bin/bash
# Kernel trace of a short bench run: how much of the denoising loop's span is kernel time, how much is gaps
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gap_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gap_trace -o t -- python3 $R/bench.py --steps 1 --warmup 1 --skip-cpu-baseline --skip-roofline --skip-fp32-latency > $R/gpurun_out/gap_bench.json 2> $R/gpurun_out/gap_bench.err
python3 $R/experiments/gap_trace_print.py $R/gpurun_out/gap_trace | tee $R/gpurun_out/gap_trace.txt
rm -rf $R/gpurun_out/gap_trace
