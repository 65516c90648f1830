// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libsdamd.
// Wave = 64 lanes; bf16 is carried as raw 16-bit words, all arithmetic in f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define SDAMD_F32 0
#define SDAMD_BF16 1

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA bf16 operand (4 VGPR)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ float bf2f(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t_;
  const f32x2_t_ v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t_));  // one v_cvt_pk_bf16_f32
}

// Exact three-term bf16 split of two f32 values, a = hi + mid + lo (8 + 8 + 8 significant bits; each residual is exact
// in f32): packed pairs for the bf16 MFMA operands of the "bf16x3" f32 paths (gemm_common.h split3_bf16, attention.hip).
struct Split3 { unsigned h, m, l; };
__device__ __forceinline__ Split3 split3_pair(float a0, float a1) {
  Split3 o;
  o.h = pack_bf2(a0, a1);
  const float r0 = a0 - __builtin_bit_cast(float, o.h << 16), r1 = a1 - __builtin_bit_cast(float, o.h & 0xffff0000u);
  o.m = pack_bf2(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, o.m << 16), s1 = r1 - __builtin_bit_cast(float, o.m & 0xffff0000u);
  o.l = pack_bf2(s0, s1);
  return o;
}
// eight values (two 16-byte f32 chunks) -> the three bf16x8 MFMA operands
__device__ __forceinline__ void split3_bf16(const f32x4_t a, const f32x4_t b, bf16x8_t& hi, bf16x8_t& mid, bf16x8_t& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  u32x4_t h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const Split3 t = split3_pair(v[2 * q], v[2 * q + 1]); h[q] = t.h; m[q] = t.m; l[q] = t.l; }
  hi = __builtin_bit_cast(bf16x8_t, h); mid = __builtin_bit_cast(bf16x8_t, m); lo = __builtin_bit_cast(bf16x8_t, l);
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16 { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float (&v)[16 / sizeof(T)]);
template <> __device__ __forceinline__ void load_vec<float>(const float* p, float (&v)[4]) {
  f32x4_t t = *reinterpret_cast<const f32x4_t*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* p, float (&v)[8]) {
  u32x4_t t = *reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __builtin_bit_cast(float, t[i] << 16);
    v[2 * i + 1] = __builtin_bit_cast(float, t[i] & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float (&v)[16 / sizeof(T)]);
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float (&v)[4]) {
  f32x4_t t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4_t*>(p) = t;
}
template <> __device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float (&v)[8]) {
  u32x4_t t;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4_t*>(p) = t;
}

// store V consecutive values as T (V = 4 or 8): one 8- or 16-byte vector store (two for 8 floats)
template <typename T, int V> __device__ __forceinline__ void store_n(T* p, const float (&v)[V]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int k = 0; k < V / 4; ++k)
      *reinterpret_cast<f32x4_t*>(p + 4 * k) = (f32x4_t){v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
  } else if constexpr (V == 8) {
    u32x4_t t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<u32x4_t*>(p) = t;
  } else {
    u32x2_t t = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *reinterpret_cast<u32x2_t*>(p) = t;
  }
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// x * Phi(x) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7; relative L2 of the GELU vs
// float64 5e-8, the same as the libm erff form): branch-free, 1 v_rcp + 1 v_exp + 11 plain VALU.  The
// libm erff costs ~50 VALU per value -- fused into the GeGLU GEMM epilogue that was 25k of the 40k
// cycles a 256x256 tile took at K = 320.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
  const float y = __builtin_fmaf(-(poly * t), e, 1.0f);   // erf(|x| / sqrt 2)
  const float hx = 0.5f * x;
  return __builtin_fmaf(__builtin_fabsf(hx), y, hx);
}
// The same GELU where the result is rounded to bf16 right away (the fused GeGLU epilogue of the bf16 GEMM): erf(x / sqrt 2)
// = x Q(x^2) on |x| <= 4.4 (clamped: 1 - erf(4.4 / sqrt 2) = 1.1e-5), Q of degree 8 fitted to a maximum error of 3.4e-5 in
// erf = 7.4e-5 in the GELU at worst (x ~ 4), 1/50 of a bf16 rounding step there.  No v_rcp / v_exp (7.7 cycles each) and the
// Horner chain packs two values per v_pk_fma_f32: the epilogue's math is what a GeGLU tile waits for after its K loop.
// Two values at a time, so that every step is a packed instruction (v_pk_mul_f32 / v_pk_fma_f32).
typedef __attribute__((ext_vector_type(2))) float gelu_f32x2_t;
__device__ __forceinline__ gelu_f32x2_t gelu_erf_bf16out_f2(gelu_f32x2_t x) {
  typedef gelu_f32x2_t v2;
  const v2 xc = {__builtin_fminf(__builtin_fmaxf(x[0], -4.4f), 4.4f), __builtin_fminf(__builtin_fmaxf(x[1], -4.4f), 4.4f)};
  const v2 u = xc * xc;
  auto k = [](float c) { return (v2){c, c}; };
  v2 q = __builtin_elementwise_fma(k(8.874779384e-11f), u, k(-8.991754896e-09f));
  q = __builtin_elementwise_fma(q, u, k(4.008112171e-07f));
  q = __builtin_elementwise_fma(q, u, k(-1.045191675e-05f));
  q = __builtin_elementwise_fma(q, u, k(1.796257420e-04f));
  q = __builtin_elementwise_fma(q, u, k(-2.180628711e-03f));
  q = __builtin_elementwise_fma(q, u, k(1.953514107e-02f));
  q = __builtin_elementwise_fma(q, u, k(-1.325753778e-01f));
  q = __builtin_elementwise_fma(q, u, k(7.977659106e-01f));
  const v2 yq = xc * q;
  const v2 y = {__builtin_fminf(__builtin_fmaxf(yq[0], -1.0f), 1.0f), __builtin_fminf(__builtin_fmaxf(yq[1], -1.0f), 1.0f)};
  const v2 hx = x * k(0.5f);
  return __builtin_elementwise_fma(hx, y, hx);
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + expf(-1.702f * x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// host-side error plumbing (defined in api.hip)
extern "C" const char* sdamd_last_error();
int sdamd_set_error(int code, const char* fmt, ...);
#define SD_CHECK(cond, ...) do { if (!(cond)) return sdamd_set_error(1, __VA_ARGS__); } while (0)
#define SD_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); \
  if (e_ != hipSuccess) return sdamd_set_error(2, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

// Opt a kernel in to `bytes` of dynamic LDS, once per (kernel instantiation, device): the attribute applies to the
// CURRENT device only, so the done-flag is a per-device bit mask (one static mask per launcher instantiation).
// Setting it twice from two host threads is harmless.
static inline hipError_t sd_lds_opt_in(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}
