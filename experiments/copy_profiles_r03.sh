#!/bin/bash
# After `gpurun -- 'bash experiments/collect_profiles_r03.sh <tag>'`: copy the judged artefacts into profiles/.
T=${1:-a}; O=gpurun_out/r03_$T
cp $O/bench.json profiles/r03_bench_c3.json
cp $O/kernel_stats.csv profiles/r03_bench_c3_kernel_stats.csv
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
cp $O/pmc_mfma_util.json profiles/r03_pmc_mfma_util.json
for f in b16_bf16 b2_bf16 b2_f32; do grep -v amdgpu.ids $O/unet_profile_$f.txt > profiles/r03_unet_profile_$f.txt; done
