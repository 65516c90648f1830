// GroupNorm(+SiLU) over NHWC and LayerNorm: HBM-bound kernels with wavefront reductions.
// Replaces the eager chains diffusion.py:26-28,38-40,93,292-294; decoder.py:25-30,49,118-120;
// clip.py:54,63,95 of the reference.  Statistics are accumulated in f64 (deterministic order,
// no atomics) so fp32 results stay well inside the 1e-4 budget of the whole UNet.
#include "common.h"
#include <mutex>

// ---------------------------------------------------------------------------------------------
// GroupNorm stats: grid (nblk, B), 256 threads.  A block walks a contiguous range of pixels of
// image n; thread t owns channel-vector cv = t % tpp (+ i*tpp for i < nchunk) and pixel lane
// t / tpp, so the channel (hence the group) of every register accumulator is fixed.
// Per-block partial sums go to ws[n][blk][32][2] (double); every gn_apply block finishes the reduction.
// ---------------------------------------------------------------------------------------------
constexpr int GN_THREADS = 256;
constexpr int GN_MAXCHUNK = 4;

// NCHUNK is a template parameter and the pixel loop is unrolled so that every thread keeps >= 4 independent
// 16-byte loads in flight: with one load per iteration the pass was latency-bound (the bf16-input pass took as
// long as the f32 one although it moves half the bytes).
template <typename T, int NCHUNK>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1,
                                                              double* __restrict__ ws, int HW, int C, int tpp,
                                                              int pix_per_blk) {
  constexpr int nchunk = NCHUNK;
  constexpr int UNR = NCHUNK >= 4 ? 1 : NCHUNK >= 2 ? 2 : 4;
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* part = reinterpret_cast<double*>(smem);  // [pif][C][2]
  const int n = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int t = threadIdx.x;
  const int pif = GN_THREADS / tpp;  // pixels in flight
  const int pl = t / tpp, cvt = t % tpp;
  const int p0 = blk * pix_per_blk;
  const int p1 = min(HW, p0 + pix_per_blk);
  double s[GN_MAXCHUNK][V], q[GN_MAXCHUNK][V];
#pragma unroll
  for (int i = 0; i < GN_MAXCHUNK; ++i)
#pragma unroll
    for (int j = 0; j < V; ++j) { s[i][j] = 0.0; q[i][j] = 0.0; }
  if (pl < pif) {
    // x2 != null: the tensor is the channel concat [x (C1 channels) | x2 (C - C1)] of two NHWC tensors that is
    // never materialised (UNet skip connections); a thread's channel vectors are fixed, so is their source
    const T* src[GN_MAXCHUNK];
    int rs[GN_MAXCHUNK];
#pragma unroll
    for (int i = 0; i < GN_MAXCHUNK; ++i) {
      const int c0 = (cvt + i * tpp) * V;
      if (x2 && c0 >= C1) { rs[i] = C - C1; src[i] = x2 + (size_t)n * HW * rs[i] + (c0 - C1); }
      else { rs[i] = x2 ? C1 : C; src[i] = x + (size_t)n * HW * rs[i] + c0; }
    }
    int p = p0 + pl;
    for (; p + (UNR - 1) * pif < p1; p += UNR * pif) {
      float v[UNR][NCHUNK][V];
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int i = 0; i < NCHUNK; ++i) load_vec<T>(src[i] + (size_t)(p + u * pif) * rs[i], v[u][i]);
#pragma unroll
      for (int u = 0; u < UNR; ++u)      // same accumulation order as the rolled loop: pixel-major
#pragma unroll
        for (int i = 0; i < NCHUNK; ++i)
#pragma unroll
          for (int j = 0; j < V; ++j) { double d = (double)v[u][i][j]; s[i][j] += d; q[i][j] += d * d; }
    }
    for (; p < p1; p += pif) {
#pragma unroll
      for (int i = 0; i < NCHUNK; ++i) {
        float v[V];
        load_vec<T>(src[i] + (size_t)p * rs[i], v);
#pragma unroll
        for (int j = 0; j < V; ++j) { double d = (double)v[j]; s[i][j] += d; q[i][j] += d * d; }
      }
    }
#pragma unroll
    for (int i = 0; i < GN_MAXCHUNK; ++i) {
      if (i < nchunk) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          int c = (cvt + i * tpp) * V + j;
          part[((size_t)pl * C + c) * 2 + 0] = s[i][j];
          part[((size_t)pl * C + c) * 2 + 1] = q[i][j];
        }
      }
    }
  }
  __syncthreads();
  // 8 threads per group, fixed summation order -> deterministic
  const int g = t >> 3, sub = t & 7;
  const int cpg = C / 32;
  const int nent = pif * cpg;
  double ss = 0.0, qq = 0.0;
  for (int e = sub; e < nent; e += 8) {
    int ppl = e / cpg, c = g * cpg + e % cpg;
    ss += part[((size_t)ppl * C + c) * 2 + 0];
    qq += part[((size_t)ppl * C + c) * 2 + 1];
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); qq += __shfl_xor(qq, o, 64); }
  if (sub == 0) {
    double* dst = ws + (((size_t)n * nblk + blk) * 32 + g) * 2;
    dst[0] = ss; dst[1] = qq;
  }
}

// GroupNorm apply (+ optional SiLU): grid (nblk2, B); streaming pass, 16-byte vectors.  Every block first
// finishes the statistics of its image itself: 256 threads = 32 groups x 8 lanes, each lane sums every 8th
// per-block partial (independent loads from L2), then a fixed-order shuffle tree -- the same order in every
// block and every run, so results are deterministic.  (As a kernel of its own this step was 4.7 us per
// GroupNorm, 1.2 % of the bench's GPU time.)
template <typename T, typename TO>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1,
                                                       TO* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const double* __restrict__ ws, int nblk_stats, float eps, int HW,
                                                       int C, int silu, int pix_per_blk) {
  constexpr int V = 16 / sizeof(T);
  __shared__ float s_mean[32], s_rstd[32];
  // per-channel {mean, rstd*gamma, beta} so that the streaming loop is  y = silu((x - m) * a + b)  with three
  // 16-byte LDS reads per vector and no group arithmetic (the per-element group lookup + libm expf + IEEE divide
  // made this pass VALU-bound: 34 us for the 64x64x320 map at batch 16 whether the input was f32 or bf16)
  extern __shared__ __attribute__((aligned(16))) char gn_smem[];
  float* t_mean = reinterpret_cast<float*>(gn_smem);
  float* t_a = t_mean + C;
  float* t_b = t_a + C;
  const int n = blockIdx.y;
  const int t = threadIdx.x;
  const int cpg = C / 32;
  {
    const int g = t >> 3, sub = t & 7;
    double ss = 0.0, qq = 0.0;
    for (int b = sub; b < nblk_stats; b += 8) {
      const double* src = ws + (((size_t)n * nblk_stats + b) * 32 + g) * 2;
      ss += src[0]; qq += src[1];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); qq += __shfl_xor(qq, o, 64); }
    if (sub == 0) {
      const double cnt = (double)HW * cpg;
      const double mean = ss / cnt;
      double var = qq / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    const int g = (int)(((float)c + 0.5f) / (float)cpg);   // exact: (c+0.5)/cpg is never within rounding of an integer
    t_mean[c] = s_mean[g];
    t_a[c] = s_rstd[g] * gamma[c];
    t_b[c] = beta[c];
  }
  __syncthreads();
  const int vpp = C / V;  // vectors per pixel
  const int p0 = blockIdx.x * pix_per_blk;
  const int p1 = min(HW, p0 + pix_per_blk);
  const size_t img = (size_t)n * HW * C;
  const int nvec = (p1 - p0) * vpp;
  const int dp = 256 / vpp, dcv = 256 % vpp;
  int p = p0 + t / vpp, cv = t % vpp;
  auto src_of = [&](int pp, int c0) -> const T* {
    if (!x2) return x + img + (size_t)pp * C + c0;
    return c0 < C1 ? x + ((size_t)n * HW + pp) * C1 + c0 : x2 + ((size_t)n * HW + pp) * (C - C1) + (c0 - C1);
  };
  auto finish = [&](int pp, int c0, float (&v)[V]) {
#pragma unroll
    for (int k = 0; k < V / 4; ++k) {
      const f32x4_t mn = *reinterpret_cast<const f32x4_t*>(t_mean + c0 + 4 * k);
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(t_a + c0 + 4 * k);
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(t_b + c0 + 4 * k);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r = __builtin_fmaf(v[4 * k + j] - mn[j], a[j], b[j]);
        // silu(r) = r / (1 + 2^(-r log2 e)): v_exp + v_rcp (1 ulp each) instead of libm expf + IEEE divide
        v[4 * k + j] = silu ? r * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * r)) : r;
      }
    }
    store_n<TO, V>(y + img + (size_t)pp * C + c0, v);
  };
  auto advance = [&]() { p += dp; cv += dcv; if (cv >= vpp) { cv -= vpp; ++p; } };
  // 4 independent 16-byte loads in flight per thread (one per iteration left the pass latency-bound)
  constexpr int UNR = 4;
  int i = t;
  for (; i + (UNR - 1) * 256 < nvec; i += UNR * 256) {
    int pp[UNR], cc[UNR];
    float v[UNR][V];
#pragma unroll
    for (int u = 0; u < UNR; ++u) { pp[u] = p; cc[u] = cv * V; load_vec<T>(src_of(p, cv * V), v[u]); advance(); }
#pragma unroll
    for (int u = 0; u < UNR; ++u) finish(pp[u], cc[u], v[u]);
  }
  for (; i < nvec; i += 256) {
    float v[V];
    load_vec<T>(src_of(p, cv * V), v);
    finish(p, cv * V, v);
    advance();
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, the row lives in registers (C <= 64*5*V).
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAXIT = 5;
// LPR lanes share one row (64 / LPR rows per wave): C = 320 f32 is 80 vectors = 16 lanes x 5, so a wave takes
// 4 rows with every lane busy (one wave per row left 48 of 64 lanes idle on the second sweep: 3.7 -> 5 TB/s).
template <typename T, typename TO, int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, TO* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, float eps) {
  constexpr int V = 16 / sizeof(T);
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < rows;
  const int nvec = C / V;
  const T* xr = x + (size_t)(live ? row : 0) * C;
  // All of the row's vectors are requested before any is used, from clamped addresses (a load under `if (cv < nvec)` next
  // to its use is emitted as load, s_waitcnt vmcnt(0), use: up to five HBM round trips in series per row), and gamma /
  // beta with them: they arrive while the statistics are reduced.
  float v[LN_MAXIT][V];
  f32x4_t gm[LN_MAXIT][V / 4], bt[LN_MAXIT][V / 4];
#pragma unroll
  for (int i = 0; i < LN_MAXIT; ++i) load_vec<T>(xr + (size_t)min(sub + i * LPR, nvec - 1) * V, v[i]);
#pragma unroll
  for (int i = 0; i < LN_MAXIT; ++i)
#pragma unroll
    for (int k = 0; k < V / 4; ++k) {
      const int c0 = min(sub + i * LPR, nvec - 1) * V + 4 * k;
      gm[i][k] = *reinterpret_cast<const f32x4_t*>(gamma + c0);
      bt[i][k] = *reinterpret_cast<const f32x4_t*>(beta + c0);
    }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXIT; ++i) {
    int cv = sub + i * LPR;
    if (cv < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) sum += v[i][j];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXIT; ++i) {
    int cv = sub + i * LPR;
    if (cv < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) { float d = v[i][j] - mean; sq += d * d; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  if (!live) return;
  TO* yr = y + (size_t)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXIT; ++i) {
    int cv = sub + i * LPR;
    if (cv < nvec) {
      float o[V];
#pragma unroll
      for (int k = 0; k < V / 4; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * k + j] = (v[i][4 * k + j] - mean) * rstd * gm[i][k][j] + bt[i][k][j];
      }
      store_n<TO, V>(yr + (size_t)cv * V, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
static int gn_geometry(int C, int V, int* tpp, int* nchunk) {
  int cvs = C / V;
  int nc = (cvs + GN_THREADS - 1) / GN_THREADS;
  while (nc <= GN_MAXCHUNK && cvs % nc != 0) ++nc;
  if (nc > GN_MAXCHUNK) return -1;
  *nchunk = nc;
  *tpp = cvs / nc;
  return 0;
}

// workspace: per-block partials [B][64][32][2] f64 (sized as before: the tail once held the finalized statistics)
extern "C" size_t sdamd_groupnorm_ws_bytes(int B) { return (size_t)B * (64 * 32 * 2 * sizeof(double) + 32 * 2 * sizeof(float)); }

// GroupNorm of a small map in ONE launch: one workgroup per (sample, group) keeps the group's HW x C/32 elements in
// registers (pairs of channels, <= GNG_MAXP pairs per thread), so statistics and normalisation need no second pass over
// memory and no second kernel.  Only for small maps (a group of <= 10240 elements: the 8x8 level of the UNet with 256
// threads, the un-concatenated 16x16 level with 512), where it saves 5-8 us per call; with more work per workgroup
// (40 pairs per thread, or 1024 threads on the 32x32 level) it measured 10-30 % SLOWER than the two-launch form
// (512 workgroups, each a serial load -> reduce -> reduce -> store chain), so larger groups keep that form.
// Two-pass variance in f32 (mean first, then the centred squares).
constexpr int GNG_MAXP = 10;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <typename T, typename TO, int NT>
__global__ __launch_bounds__(NT) void gn_group_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1,
                                                       TO* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int HW, int C, int silu) {
  __shared__ float s_red[NT / 64];
  const int n = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  const int cpg = C / 32, ppp = cpg >> 1;           // channels per group, channel pairs per pixel
  const int items = HW * ppp, c_base = g * cpg;
  const int dp = NT / ppp, dc = NT % ppp;
  auto block_sum = [&](float a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    __syncthreads();                                  // (s_red may still be read from the previous call)
    if ((t & 63) == 0) s_red[t >> 6] = a;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tot += s_red[w];
    return tot;
  };
  // All of the thread's items are requested before any is used (a load under `if (t + NT * k < items)` next to its use is
  // emitted as load, s_waitcnt vmcnt(0), use: up to GNG_MAXP HBM round trips in series); items past the end re-read the
  // thread's first one (or item 0) and are not accumulated.
  float v[GNG_MAXP][2];
  float sum = 0.f;
  {
    int px = t / ppp, pc = t - px * ppp;
    const int px0 = t < items ? px : 0, pc0 = t < items ? pc : 0;
    typedef typename std::conditional<sizeof(T) == 4, f32x2_t, unsigned>::type raw_t;
    raw_t raw[GNG_MAXP];
#pragma unroll
    for (int k = 0; k < GNG_MAXP; ++k) {
      const bool ok = t + NT * k < items;
      const int pxk = ok ? px : px0, c = c_base + 2 * (ok ? pc : pc0);
      const T* src = !x2 ? x + ((size_t)n * HW + pxk) * C + c
                         : c < C1 ? x + ((size_t)n * HW + pxk) * C1 + c : x2 + ((size_t)n * HW + pxk) * (C - C1) + (c - C1);
      raw[k] = *reinterpret_cast<const raw_t*>(src);
      px += dp; pc += dc;
      if (pc >= ppp) { pc -= ppp; ++px; }
    }
#pragma unroll
    for (int k = 0; k < GNG_MAXP; ++k) {
      v[k][0] = v[k][1] = 0.f;
      if (t + NT * k < items) {
        if constexpr (sizeof(T) == 4) {
          v[k][0] = raw[k][0]; v[k][1] = raw[k][1];
        } else {
          v[k][0] = __builtin_bit_cast(float, raw[k] << 16); v[k][1] = __builtin_bit_cast(float, raw[k] & 0xffff0000u);
        }
        sum += v[k][0] + v[k][1];
      }
    }
  }
  const float cnt = (float)HW * (float)cpg;
  const float mean = block_sum(sum) / cnt;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < GNG_MAXP; ++k)
    if (t + NT * k < items) { const float a = v[k][0] - mean, b = v[k][1] - mean; sq += a * a + b * b; }
  const float rstd = 1.0f / sqrtf(block_sum(sq) / cnt + eps);
  {
    int px = t / ppp, pc = t - px * ppp;
    const int pc0 = t < items ? pc : 0;
    f32x2_t gav[GNG_MAXP], bev[GNG_MAXP];               // gamma / beta of every item, one batch (as above)
    {
      int qc = pc;
#pragma unroll
      for (int k = 0; k < GNG_MAXP; ++k) {
        const int c = c_base + 2 * (t + NT * k < items ? qc : pc0);
        gav[k] = *reinterpret_cast<const f32x2_t*>(gamma + c);
        bev[k] = *reinterpret_cast<const f32x2_t*>(beta + c);
        qc += dc;
        if (qc >= ppp) qc -= ppp;
      }
    }
#pragma unroll
    for (int k = 0; k < GNG_MAXP; ++k) {
      if (t + NT * k < items) {
        const int c = c_base + 2 * pc;
        const f32x2_t ga = gav[k], be = bev[k];
        float r[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          r[j] = __builtin_fmaf(v[k][j] - mean, rstd * ga[j], be[j]);
          if (silu) r[j] = r[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * r[j]));
        }
        TO* dst = y + ((size_t)n * HW + px) * C + c;
        if constexpr (sizeof(TO) == 4) *reinterpret_cast<f32x2_t*>(dst) = (f32x2_t){r[0], r[1]};
        else *reinterpret_cast<unsigned*>(dst) = pack_bf2(r[0], r[1]);
      }
      px += dp; pc += dc;
      if (pc >= ppp) { pc -= ppp; ++px; }
    }
  }
}

// apply pass: ~4K vectors (64 KB) per block, but keep >= ~1024 blocks in flight; every block first finishes its sample's
// statistics from the nblk_stats per-block partials in ws
static int gn_launch_apply(const void* x, const void* x2, int C1, void* y, const float* gamma, const float* beta, const double* ws,
                           int nblk_stats, int B, int HW, int C, float eps, int silu, int dtype, int out_dtype, hipStream_t st) {
  const int V = dtype == SDAMD_BF16 ? 8 : 4;
  int vpp = C / V;
  long long total_vec = (long long)B * HW * vpp;
  int vec_per_blk = 4096;
  while (vec_per_blk > 512 && total_vec / vec_per_blk < 1024) vec_per_blk >>= 1;
  int ppb2 = (vec_per_blk + vpp - 1) / vpp;
  if (ppb2 < 1) ppb2 = 1;
  int nblk2 = (HW + ppb2 - 1) / ppb2;
  dim3 grid2(nblk2, B);
  const size_t lds2 = (size_t)3 * C * sizeof(float);   // <= 30 KB at C = 2560
  if (dtype == SDAMD_BF16)
    gn_apply_kernel<bf16_t, bf16_t><<<grid2, 256, lds2, st>>>((const bf16_t*)x, (const bf16_t*)x2, C1, (bf16_t*)y, gamma, beta, ws, nblk_stats, eps, HW, C, silu, ppb2);
  else if (out_dtype == SDAMD_BF16)
    gn_apply_kernel<float, bf16_t><<<grid2, 256, lds2, st>>>((const float*)x, (const float*)x2, C1, (bf16_t*)y, gamma, beta, ws, nblk_stats, eps, HW, C, silu, ppb2);
  else
    gn_apply_kernel<float, float><<<grid2, 256, lds2, st>>>((const float*)x, (const float*)x2, C1, (float*)y, gamma, beta, ws, nblk_stats, eps, HW, C, silu, ppb2);
  SD_LAUNCH_CHECK("gn_apply");
  return 0;
}

static int groupnorm_impl(const void* x, const void* x2, int C1, void* y, const float* gamma, const float* beta, void* ws, int B,
                          int HW, int C, float eps, int silu, int dtype, int out_dtype, void* stream) {
  const int V = dtype == SDAMD_BF16 ? 8 : 4;
  SD_CHECK(!x2 || (C1 > 0 && C1 < C && C1 % V == 0), "groupnorm_cat: bad split C1=%d of C=%d", C1, C);
  SD_CHECK((dtype == SDAMD_BF16 || dtype == SDAMD_F32) && (out_dtype == SDAMD_BF16 || out_dtype == SDAMD_F32) &&
           !(dtype == SDAMD_BF16 && out_dtype == SDAMD_F32), "groupnorm: unsupported dtypes in=%d out=%d", dtype, out_dtype);
  SD_CHECK(C % 32 == 0 && C % V == 0, "groupnorm: C=%d must be a multiple of 32 and of the 16-byte vector", C);
  SD_CHECK(B > 0 && HW > 0, "groupnorm: empty input B=%d HW=%d", B, HW);
  hipStream_t st = (hipStream_t)stream;
  {   // small maps: one launch, one workgroup per (sample, group)
    const int cpg = C / 32;
    const long long items = (long long)HW * (cpg / 2);
    if (cpg % 2 == 0 && items <= 512LL * GNG_MAXP && (!x2 || C1 % 2 == 0)) {
      dim3 gg(32, B);
#define GNG(TT, TTO, NT) gn_group_kernel<TT, TTO, NT><<<gg, NT, 0, st>>>((const TT*)x, (const TT*)x2, C1, (TTO*)y, gamma, beta, eps, HW, C, silu)
#define GNG_T(TT, TTO) { if (items <= 256LL * GNG_MAXP) GNG(TT, TTO, 256); else GNG(TT, TTO, 512); }
      if (dtype == SDAMD_BF16) GNG_T(bf16_t, bf16_t)
      else if (out_dtype == SDAMD_BF16) GNG_T(float, bf16_t)
      else GNG_T(float, float)
#undef GNG_T
#undef GNG
      SD_LAUNCH_CHECK("gn_group");
      return 0;
    }
  }
  int tpp, nchunk;
  SD_CHECK(gn_geometry(C, V, &tpp, &nchunk) == 0, "groupnorm: unsupported C=%d", C);
  const int pif = GN_THREADS / tpp;
  size_t lds = (size_t)pif * C * 2 * sizeof(double);
  SD_CHECK(lds <= 160 * 1024, "groupnorm: C=%d needs %zu B of LDS", C, lds);
  // stats blocks per image: enough to fill the chip, at most 64 (ws size), >= 1 pixel sweep each
  extern int g_plan_samples;     // deterministic plans (gemm_common.h): partial blocks per sample as if the batch were one sample
  const int Bp = (g_plan_samples > 1 && B % g_plan_samples == 0) ? B / g_plan_samples : B;
  int want = (2048 + Bp - 1) / Bp;
  int nblk = HW / (pif * 4);
  if (nblk > want) nblk = want;
  if (nblk > 64) nblk = 64;
  if (nblk < 1) nblk = 1;
  int ppb = (HW + nblk - 1) / nblk;
  nblk = (HW + ppb - 1) / ppb;
  dim3 grid(nblk, B);
#define GN_STATS(TT, NC)                                                                                               \
  {                                                                                                                    \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)gn_stats_kernel<TT, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    gn_stats_kernel<TT, NC><<<grid, GN_THREADS, lds, st>>>((const TT*)x, (const TT*)x2, C1, (double*)ws, HW, C, tpp, ppb);  \
  }
#define GN_STATS_T(TT) switch (nchunk) { case 1: GN_STATS(TT, 1) break; case 2: GN_STATS(TT, 2) break; case 3: GN_STATS(TT, 3) break; default: GN_STATS(TT, 4) }
  if (dtype == SDAMD_BF16) GN_STATS_T(bf16_t) else GN_STATS_T(float)
#undef GN_STATS_T
#undef GN_STATS
  SD_LAUNCH_CHECK("gn_stats");
  return gn_launch_apply(x, x2, C1, y, gamma, beta, (const double*)ws, nblk, B, HW, C, eps, silu, dtype, out_dtype, st);
}

// GroupNorm statistics from the per-column partial sums a producing GEMM / conv epilogue left behind
// (GemmArgs::colstats, gemm2.hip): s1 [B*HW/64][C1][2] (and s2 [..][C2][2] for the second source of a virtual concat)
// hold sum and sum of squares of every channel over each 64-pixel block; sample n owns blocks [n*HW/64, (n+1)*HW/64).
// One workgroup per sample folds them (f64, fixed order) into the [n][1 block][32 groups][2] layout gn_apply_kernel
// finishes, so the statistics pass over the tensor itself (one full read of the f32 trunk) is not run at all.
__global__ __launch_bounds__(256) void gn_finalize_pre_kernel(const float* __restrict__ s1, const float* __restrict__ s2, int C1,
                                                              int C2, double* __restrict__ ws, int HW) {
  // one workgroup per (group, sample): HW/64 blocks x C/32 channels partials, a few independent loads per thread
  __shared__ double red[2][4];
  const int g = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const int C = C1 + C2, cpg = C / 32, nrb = HW / 64;
  double ss = 0.0, qq = 0.0;
  // four entries per thread requested at a time (clamped addresses; one per iteration is load, wait, add in series), added
  // in the same order as a plain loop
  const int tot = nrb * cpg;
  for (int e0 = t; e0 < tot; e0 += 4 * 256) {
    f32x2_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + 256 * u, tot - 1);
      const int rb = e / cpg, c = g * cpg + (e - rb * cpg);
      const float* src = c < C1 ? s1 + ((size_t)(n * nrb + rb) * C1 + c) * 2 : s2 + ((size_t)(n * nrb + rb) * C2 + (c - C1)) * 2;
      v[u] = *reinterpret_cast<const f32x2_t*>(src);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + 256 * u < tot) { ss += (double)v[u][0]; qq += (double)v[u][1]; }
  }
  ss = wave_sum_d(ss); qq = wave_sum_d(qq);
  if ((t & 63) == 0) { red[0][t >> 6] = ss; red[1][t >> 6] = qq; }
  __syncthreads();
  if (t == 0) {
    double* dst = ws + ((size_t)n * 32 + g) * 2;
    dst[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    dst[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

extern "C" int sdamd_groupnorm_nhwc_pre(const void* x1, const void* x2, int C1, int C2, void* y, const float* gamma,
                                        const float* beta, const float* stats1, const float* stats2, void* ws, int B, int HW,
                                        float eps, int silu, int dtype, int out_dtype, void* stream) {
  const int C = C1 + C2;
  const int V = dtype == SDAMD_BF16 ? 8 : 4;
  SD_CHECK(x1 && stats1 && C1 > 0 && (C2 == 0) == (x2 == nullptr) && (C2 == 0 || stats2), "groupnorm_pre: missing source or statistics");
  SD_CHECK(HW % 64 == 0, "groupnorm_pre: HW=%d must be a multiple of the 64-row statistics block", HW);
  SD_CHECK(B > 0 && C % 32 == 0 && C % V == 0 && (C2 == 0 || C1 % V == 0), "groupnorm_pre: unsupported shape B=%d C=%d+%d", B, C1, C2);
  SD_CHECK((dtype == SDAMD_BF16 || dtype == SDAMD_F32) && (out_dtype == SDAMD_BF16 || out_dtype == SDAMD_F32) &&
           !(dtype == SDAMD_BF16 && out_dtype == SDAMD_F32), "groupnorm_pre: unsupported dtypes in=%d out=%d", dtype, out_dtype);
  {   // small maps: the one-launch kernel holds the whole group in registers, nothing to save
    const int cpg = C / 32;
    const long long items = (long long)HW * (cpg / 2);
    if (cpg % 2 == 0 && items <= 512LL * GNG_MAXP && (!x2 || C1 % 2 == 0))
      return groupnorm_impl(x1, x2, C1, y, gamma, beta, ws, B, HW, C, eps, silu, dtype, out_dtype, stream);
  }
  hipStream_t st = (hipStream_t)stream;
  gn_finalize_pre_kernel<<<dim3(32, B), 256, 0, st>>>(stats1, stats2, C1, C2, (double*)ws, HW);
  SD_LAUNCH_CHECK("gn_finalize_pre");
  return gn_launch_apply(x1, x2, C1, y, gamma, beta, (const double*)ws, 1, B, HW, C, eps, silu, dtype, out_dtype, st);
}

extern "C" int sdamd_groupnorm_nhwc(const void* x, void* y, const float* gamma, const float* beta, void* ws, int B,
                                    int HW, int C, float eps, int silu, int dtype, int out_dtype, void* stream) {
  return groupnorm_impl(x, nullptr, 0, y, gamma, beta, ws, B, HW, C, eps, silu, dtype, out_dtype, stream);
}

extern "C" int sdamd_groupnorm_nhwc_cat(const void* x1, const void* x2, int C1, int C2, void* y, const float* gamma,
                                        const float* beta, void* ws, int B, int HW, float eps, int silu, int dtype,
                                        int out_dtype, void* stream) {
  SD_CHECK(x1 && x2 && C1 > 0 && C2 > 0, "groupnorm_cat: null source or empty half");
  return groupnorm_impl(x1, x2, C1, y, gamma, beta, ws, B, HW, C1 + C2, eps, silu, dtype, out_dtype, stream);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm for callers that cannot allocate (code under hipGraph capture, tight serving loops, harnesses that
// forbid hipMalloc): same kernels as sdamd_groupnorm_nhwc, but the per-block partial statistics live in a
// file-scope device array instead of a caller workspace.  Every GPU the library runs on holds its own copy of the
// array (2 MiB) at its own address until the process ends; the address is looked up once per device.
// Nothing has to start at zero: every slot the apply pass reads was written by the stats pass of the same call.
// Calls that overlap on one GPU (two streams) share the array and race -- serialise them.
// ---------------------------------------------------------------------------------------------
constexpr int GN_NOALLOC_MAXB = 64;                                   // samples per call
constexpr size_t GN_NOALLOC_DOUBLES = (size_t)GN_NOALLOC_MAXB * 64 * 32 * 2;   // [B][64 blocks][32 groups][sum, sumsq]
static __device__ __align__(256) double scratch[GN_NOALLOC_DOUBLES];

constexpr int GN_MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[GN_MAX_DEVICES] = {nullptr};

// device address of `scratch` on the current device (looked up on the first call per device); null + error text on failure
static void* gn_scratch_for_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GN_MAX_DEVICES) {
    sdamd_set_error(2, "groupnorm_noalloc: no current HIP device (or device index >= %d)", GN_MAX_DEVICES);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      sdamd_set_error(2, "groupnorm_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return nullptr;
    }
    g_scratch_ptr[dev] = ptr;
  }
  return g_scratch_ptr[dev];
}

extern "C" int sdamd_noalloc_init(void) { return gn_scratch_for_current_device() ? 0 : 2; }
extern "C" int sdamd_groupnorm_noalloc_max_batch(void) { return GN_NOALLOC_MAXB; }

extern "C" int sdamd_groupnorm_nhwc_noalloc(const void* x, void* y, const float* gamma, const float* beta, int B, int HW,
                                            int C, float eps, int silu, int dtype, int out_dtype, void* stream) {
  SD_CHECK(B <= GN_NOALLOC_MAXB, "groupnorm_noalloc: B=%d exceeds the built-in scratch (max %d samples per call); "
           "split the batch or use sdamd_groupnorm_nhwc with a workspace", B, GN_NOALLOC_MAXB);
  void* ws = gn_scratch_for_current_device();
  if (!ws) return 2;
  return groupnorm_impl(x, nullptr, 0, y, gamma, beta, ws, B, HW, C, eps, silu, dtype, out_dtype, stream);
}

extern "C" int sdamd_layernorm(const void* x, void* y, const float* gamma, const float* beta, int rows, int C, float eps,
                               int dtype, int out_dtype, void* stream) {
  const int V = dtype == SDAMD_BF16 ? 8 : 4;
  SD_CHECK((dtype == SDAMD_BF16 || dtype == SDAMD_F32) && (out_dtype == SDAMD_BF16 || out_dtype == SDAMD_F32) &&
           !(dtype == SDAMD_BF16 && out_dtype == SDAMD_F32), "layernorm: unsupported dtypes in=%d out=%d", dtype, out_dtype);
  SD_CHECK(C % V == 0 && C / V <= 64 * LN_MAXIT, "layernorm: unsupported C=%d", C);
  SD_CHECK(rows > 0, "layernorm: empty input");
  hipStream_t st = (hipStream_t)stream;
  const int nvec = C / V;
  const int lpr = nvec <= 16 * LN_MAXIT ? 16 : nvec <= 32 * LN_MAXIT ? 32 : 64;   // lanes per row
#define LN_LAUNCH(TT, TTO, L)                                                                       \
  layernorm_kernel<TT, TTO, L><<<(rows + 4 * (64 / L) - 1) / (4 * (64 / L)), 256, 0, st>>>((const TT*)x, (TTO*)y, gamma, beta, rows, C, eps)
#define LN_DISPATCH(TT, TTO) \
  { if (lpr == 16) LN_LAUNCH(TT, TTO, 16); else if (lpr == 32) LN_LAUNCH(TT, TTO, 32); else LN_LAUNCH(TT, TTO, 64); }
  if (dtype == SDAMD_BF16) LN_DISPATCH(bf16_t, bf16_t)
  else if (out_dtype == SDAMD_BF16) LN_DISPATCH(float, bf16_t)
  else LN_DISPATCH(float, float)
#undef LN_DISPATCH
#undef LN_LAUNCH
  SD_LAUNCH_CHECK("layernorm");
  return 0;
}

