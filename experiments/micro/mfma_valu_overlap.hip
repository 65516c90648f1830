// Does the vector stream of one wave (or of the other waves of a SIMD) overlap with in-flight MFMAs on gfx950?
// Per loop iteration: 8 MFMAs and / or 32 vector instructions, interleaved 1 : 4 in program order; cycles per iteration
// from s_memtime, at 1, 2 and 3 waves per SIMD.   build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// MODE bits: 1 = 32x32x16 MFMAs, 2 = v_exp_f32, 4 = v_fma_f32, 8 = 16x16x32 MFMAs (twice as many), 16 = v_cvt_pk_bf16_f32
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  const int t = threadIdx.x;
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = -0.001f * (t + i);
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (t & 7)); b[i] = (__bf16)(0.02f * (t & 3)); }
  f32x16_t acc[2];
  f32x4_t acc4[4];
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc4[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (MODE & 1) acc[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 1], 0, 0, 0);
      if (MODE & 8) {
        acc4[(2 * g) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[(2 * g) & 3], 0, 0, 0);
        acc4[(2 * g + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[(2 * g + 1) & 3], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE & 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[4 * g + i]));
        if (MODE & 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[4 * g + i]));
        if (MODE & 16) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[4 * g + i]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
  for (int e = 0; e < 16; ++e) s += acc[0][e] + acc[1][e];
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc4[j][0] + acc4[j][1] + acc4[j][2] + acc4[j][3];
  out[blockIdx.x * 256 + t] = s;
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  for (int wps = 1; wps <= 3; ++wps) {          // waves per SIMD = workgroups per CU (256 threads = one wave per SIMD)
    const int grid = 256 * wps;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[768]; hipMemcpy(h, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < grid; ++i) m += h[i]; m /= grid;
    printf("%-44s waves/SIMD %d: %7.1f us  = %7.1f ns per iteration (all waves of a SIMD: %6.1f ns per wave-iteration); s_memtime ticks/iter %.1f\n",
           name, wps, ms * 1e3, ms * 1e6 / iters, ms * 1e6 / iters / wps, m / iters);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 768 * 256 * 4); hipMalloc(&cyc, 768 * 8);
  run<1>("8 mfma 32x32x16", out, cyc);
  run<8>("16 mfma 16x16x32", out, cyc);
  run<2>("32 v_exp", out, cyc);
  run<4>("32 v_fma", out, cyc);
  run<16>("32 v_cvt_pk_bf16", out, cyc);
  run<3>("8 mfma 32x32x16 + 32 v_exp", out, cyc);
  run<5>("8 mfma 32x32x16 + 32 v_fma", out, cyc);
  run<17>("8 mfma 32x32x16 + 32 v_cvt_pk", out, cyc);
  run<10>("16 mfma 16x16x32 + 32 v_exp", out, cyc);
  run<12>("16 mfma 16x16x32 + 32 v_fma", out, cyc);
  run<7>("8 mfma 32x32x16 + 32 v_exp + 32 v_fma", out, cyc);
  return 0;
}
