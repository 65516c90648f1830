#!/bin/bash
# Round-3 artefacts in one GPU call, all on the same tree and box.  Every step has its own time limit and the chain
# stops at the first failure.  Outputs land in gpurun_out/r03_<tag>/ (merged back by gpurun); copy what is to be
# judged into profiles/ afterwards (experiments/README.md).
#   1-3  rocprofv3 --pmc passes (one counter block per run, csv) over a SHORT bench run: 2 generate() calls of 6
#        denoising steps + CLIP + VAE decode.  The full 50-step run makes the profiler's own thread fault once it holds
#        more than ~20k dispatch records (profiles/r03_pmc_crash_*.log); 6 steps hold every kernel of the path.
#   4    rocprofv3 --kernel-trace --stats of the bench command (per-kernel average durations)
#   5    per-shape HIP-event tables of one eager UNet step (configs 3, bf16 batch 2, config 2)
#   6    the un-profiled bench line
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
O=$R/gpurun_out/r03_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# every profiled process leaves its module map (bench.py, SDAMD_DUMP_MAPS): should a profiler-side fault ever show again,
# its logged PCs can be attributed to a module instead of asserted
export SDAMD_DUMP_MAPS=$O/maps
mkdir -p $O/maps
SHORT="$R/bench.py --steps 1 --warmup 1 --denoise-steps 6 --skip-cpu-baseline --skip-fp32-latency --skip-roofline --skip-predrawn"
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $SHORT > $O/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- python3 $SHORT > $O/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -- python3 $SHORT > $O/pmc_mfma.log 2>&1
cd $R
python3 experiments/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.txt
python3 experiments/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma_util.json > $O/pmc_mfma_util.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma            # raw counter csv: tens of MB
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json     # bench.py reads roofline.traffic from here (this box only)
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 1 --skip-cpu-baseline --skip-fp32-latency --skip-roofline --skip-predrawn > $O/stats_bench.json 2> $O/stats_bench.err
cd $R
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
rm -rf $O/stats
timeout -k 10 200 python3 -u experiments/unet_profile.py 8 bf16 > $O/unet_profile_b16_bf16.txt 2>&1
timeout -k 10 200 python3 -u experiments/unet_profile.py 1 bf16 > $O/unet_profile_b2_bf16.txt 2>&1
timeout -k 10 200 python3 -u experiments/unet_profile.py 1 f32 > $O/unet_profile_b2_f32.txt 2>&1
timeout -k 10 500 python3 -u bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
