#!/bin/bash
# Round artefacts in one GPU call: PMC traffic passes (eager UNet steps), rocprofv3 kernel stats of bench.py,
# then the un-profiled bench line.  Each step has its own time limit; the chain stops at the first failure.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-v3}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/experiments/unet_steps.py 3 > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout -s KILL 240 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/experiments/unet_steps.py 3 > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
python3 experiments/pmc_traffic.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_traffic_$TAG.json > gpurun_out/pmc_traffic_$TAG.txt
cp gpurun_out/pmc_traffic_$TAG.json profiles/r01_pmc_traffic.json
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --skip-cpu-baseline --skip-fp32-latency --skip-roofline > $R/gpurun_out/stats_$TAG.log 2>&1
cd $R
timeout -k 10 500 python3 -u bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 3000 gpurun_out/bench_$TAG.json
