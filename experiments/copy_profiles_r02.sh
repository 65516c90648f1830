#!/bin/bash
# After `gpurun -- 'bash experiments/collect_profiles_r02.sh <tag>'`: copy the judged artefacts into profiles/.
T=${1:-a}; O=gpurun_out/r02_$T
cp $O/bench.json profiles/r02_bench_c3.json
cp $O/kernel_stats.csv profiles/r02_bench_c3_kernel_stats.csv
cp $O/pmc_traffic.json profiles/r02_pmc_traffic.json
cp $O/pmc_mfma_util.json profiles/r02_pmc_mfma_util.json
for f in b16_bf16 b2_bf16 b2_f32; do grep -v amdgpu.ids $O/unet_profile_$f.txt > profiles/r02_unet_profile_$f.txt; done
