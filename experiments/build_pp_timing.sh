#!/bin/bash
# Experiment build of the library with -DSDAMD_PP_TIMING (phase timestamps in the ping-pong GEMM kernels, gemm2_kernel.h):
# objects under /tmp, the library at experiments/micro/libsdamd_pptiming.so (git-ignored; travels with gpurun).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/stable-diffusion_amd/csrc
B=/tmp/pp_timing_build; mkdir -p $B
F="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm -DSDAMD_PP_TIMING -I$S -I$R/include"
for f in api elementwise norm gemm_conv gemm2 gemm3 gemm4 gemm5 attention attention2 attention3; do
  X=""; case $f in attention|attention2|attention3|gemm2|gemm3|gemm4) X="-mllvm -amdgpu-mfma-vgpr-form";; esac
  /opt/rocm/bin/hipcc $F $X -c $S/$f.hip -o $B/$f.o &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o $R/experiments/micro/libsdamd_pptiming.so
ls -la $R/experiments/micro/libsdamd_pptiming.so
