// MFMA GEMM / implicit-GEMM conv3x3 for gfx950 (bf16 and exact-f32 matrix cores).
//
//   out[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] + residual[m][n] )
//
// One kernel serves nn.Linear, conv1x1 (dense A rows) and conv3x3 over NHWC activations
// (A rows gathered per output pixel and filter tap: stride 1/2, symmetric or bottom/right-only
// padding, optional fused nearest-2x upsample of the input).  It replaces nn.Conv2d / nn.Linear
// in diffusion.py:9-18,54,82-85,146,196-221,287, decoder.py:10-18,78-123, encoder.py:10-57,
// clip.py:45-46 and the projections of attention.py:19-20,81-85.
//
// Tiling: 256 threads = 4 waves (2x2), block tile BM x BN, K-tile = 128 bytes per row
// (64 bf16 / 32 f32).  Both operands are K-contiguous, staged through registers into LDS with a
// 16-byte-chunk XOR swizzle (chunk ^= (row>>1)&7: conflict-free ds_read_b128 for the MFMA lane
// groups), double-buffered with one barrier per K-tile.  MFMA "A" = weight rows, "B" =
// activation rows, so each lane ends up with 4 consecutive output channels of one row: bias,
// residual and the store are 8/16-byte vector accesses.
//   bf16: v_mfma_f32_16x16x32_bf16, lane (r,q) feeds k = 8q..8q+7 of each 32-wide k-group.
//   f32 : v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain): lane (r,q) loads 4 consecutive k and
//         issues 4 MFMAs; the k order is a permutation applied to both operands alike.
#include "common.h"

#include "gemm_common.h"

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8_t type; };
template <> struct Frag<float> { typedef f32x4_t type; };
template <> struct Frag<f32e_t> { typedef f32x4_t type; };   // f32 operands, bf16x3 products (gemm_common.h)

template <typename T>
__device__ __forceinline__ void mma16(f32x4_t& acc, const typename Frag<T>::type& w, const typename Frag<T>::type& x);
template <>
__device__ __forceinline__ void mma16<bf16_t>(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& x) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<float>(f32x4_t& acc, const f32x4_t& w, const f32x4_t& x) {
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], x[e], acc, 0, 0, 0);
}

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
  constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
  constexpr int BKE = 8 * EPC;         // K elements per tile
  constexpr int PA = BM / 32, PW = BN / 32;
  constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 tiles per wave along m / n
  constexpr int STAGE = (BM + BN) * 128;
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD one
  // contiguous run of tiles, n fastest, so a run re-uses its A rows and the whole W panel.
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int batch = blockIdx.y;
  const T* __restrict__ A = (const T*)p.A + (size_t)batch * p.sA;
  const T* __restrict__ W = (const T*)p.W + (size_t)batch * p.sW;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ld_row = t >> 3, ld_chunk = t & 7;

  // per-thread source rows
  const T* wptr[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    int n = min(n0 + ld_row + 32 * i, p.N - 1);
    wptr[i] = W + (size_t)n * p.ldw + ld_chunk * EPC;
  }
  const T* aptr[PA];       // dense: row pointer; conv: image base pointer
  int ay[PA], ax[PA];      // conv: top-left input coordinate of the 3x3 window (upsampled grid)
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int m = min(m0 + ld_row + 32 * i, p.M - 1);
    if (CONV) {
      int hw = p.Ho * p.Wo;
      int n_img = m / hw, rem = m - n_img * hw;
      int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      ay[i] = oy * p.stride - p.pad;
      ax[i] = ox * p.stride - p.pad;
      aptr[i] = A + (size_t)n_img * p.H * p.Wd * p.Cin + ld_chunk * EPC;
    } else {
      aptr[i] = A + (size_t)m * p.lda + ld_chunk * EPC;
      ay[i] = ax[i] = 0;
    }
  }
  const int Hin = p.H << p.ups, Win = p.Wd << p.ups;

  // split-K (grid.z): this workgroup's K-tiles are [kt_begin, kt_begin + nk)
  const int nk_total = p.K / BKE;
  const int split = blockIdx.z;
  const int kt_begin = (int)((long long)nk_total * split / p.splits);
  const int nk = (int)((long long)nk_total * (split + 1) / p.splits) - kt_begin;
  u32x4_t ra[PA], rw[PW];
  int tap = 0, cc = 0;  // conv K position: filter tap and channel offset
  if (CONV) { const int kpt = p.Cin / BKE; tap = kt_begin / kpt; cc = (kt_begin - tap * kpt) * BKE; }
  auto issue_loads = [&](int k0) {
#pragma unroll
    for (int i = 0; i < PW; ++i) rw[i] = *reinterpret_cast<const u32x4_t*>(wptr[i] + k0);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if (CONV) {
        int ky = tap / 3, kx = tap - ky * 3;
        int iy = ay[i] + ky, ix = ax[i] + kx;
        bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        u32x4_t z = {0u, 0u, 0u, 0u};
        if (ok) z = *reinterpret_cast<const u32x4_t*>(aptr[i] + ((size_t)(iy >> p.ups) * p.Wd + (ix >> p.ups)) * p.Cin + cc);
        ra[i] = z;
      } else {
        ra[i] = *reinterpret_cast<const u32x4_t*>(aptr[i] + k0);
      }
    }
    if (CONV) { cc += BKE; if (cc >= p.Cin) { cc = 0; ++tap; } }
  };
  auto write_lds = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sw = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<u32x4_t*>(sa + swz_off(ld_row + 32 * i, ld_chunk)) = ra[i];
#pragma unroll
    for (int i = 0; i < PW; ++i) *reinterpret_cast<u32x4_t*>(sw + swz_off(ld_row + 32 * i, ld_chunk)) = rw[i];
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  issue_loads(kt_begin * BKE);
  write_lds(0);
  __syncthreads();
  const int r16 = lane & 15, q4 = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue_loads((kt_begin + kt + 1) * BKE);
    const char* sa = smem + cur * STAGE;
    const char* sw = sa + BM * 128;
    if constexpr (std::is_same<T, f32e_t>::value) {
      bf16x8_t xh[TM], xm[TM], xl[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / 2) + i * 16 + r16;
        split3_bf16(*reinterpret_cast<const f32x4_t*>(sa + swz_off(row, q4)), *reinterpret_cast<const f32x4_t*>(sa + swz_off(row, 4 + q4)), xh[i], xm[i], xl[i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / 2) + j * 16 + r16;
        bf16x8_t wh, wmid, wl;
        split3_bf16(*reinterpret_cast<const f32x4_t*>(sw + swz_off(row, q4)), *reinterpret_cast<const f32x4_t*>(sw + swz_off(row, 4 + q4)), wh, wmid, wl);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[i], acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[i], acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmid, xm[i], acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmid, xh[i], acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm[i], acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[i], acc[j][i], 0, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      frag_t xf[TM], wf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        xf[i] = *reinterpret_cast<const frag_t*>(sa + swz_off(wm * (BM / 2) + i * 16 + r16, 4 * g + q4));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        wf[j] = *reinterpret_cast<const frag_t*>(sw + swz_off(wn * (BN / 2) + j * 16 + r16, 4 * g + q4));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) mma16<T>(acc[j][i], wf[j], xf[i]);
    }
    if (kt + 1 < nk) write_lds(cur ^ 1);
    __syncthreads();
  }

  if (p.splits > 1) {  // raw f32 partial sums; splitk_reduce_kernel (gemm2.hip) applies the epilogue
    float* __restrict__ ws = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * (BM / 2) + i * 16 + r16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + q4 * 4;
        if (n < p.N) *reinterpret_cast<f32x4_t*>(ws + (size_t)m * p.N + n) = acc[j][i];
      }
    }
    return;
  }

  // epilogue: lane holds out[m][n..n+3]
  const float* __restrict__ bias = p.bias;
  const bool of32 = p.out_f32 || sizeof(T) == 4;
  const T* __restrict__ R = (p.R && !of32) ? (const T*)p.R + (size_t)batch * p.sR : nullptr;
  const float* __restrict__ Rf = (p.R && of32) ? (const float*)p.R + (size_t)batch * p.sR : nullptr;
  // Bias and f32 residual of every (row tile, column tile) of the lane are requested first, as one batch, from clamped
  // addresses (loaded under `if (m < M && n < N)` next to their use they cost one memory round trip each, in series:
  // hipcc emits load, s_waitcnt vmcnt(0), use per iteration); out-of-range tiles are not stored.
  const int m_last = p.M - 1, n_last = (p.N - 1) & ~3;
  f32x4_t bv[TN], rvv[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bv[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (bias) bv[j] = *reinterpret_cast<const f32x4_t*>(bias + min(n0 + wn * (BN / 2) + j * 16 + q4 * 4, n_last));
  }
  if (Rf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        rvv[i][j] = *reinterpret_cast<const f32x4_t*>(Rf + (size_t)min(m0 + wm * (BM / 2) + i * 16 + r16, m_last) * p.ldr +
                                                      min(n0 + wn * (BN / 2) + j * 16 + q4 * 4, n_last));
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 16 + r16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 16 + q4 * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
      }
      if (Rf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rvv[i][j][r];
      } else if (R) {
        const T* rp = R + (size_t)m * p.ldr + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += Cvt<T>::ld(rp + r);
      }
      if (p.act == ACT_QUICK_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
      } else if (p.act == ACT_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
      }
      if (of32) {
        float* cp = (float*)p.C + (size_t)batch * p.sC + (size_t)m * p.ldc + n;
        *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
      } else {
        bf16_t* cp = (bf16_t*)p.C + (size_t)batch * p.sC + (size_t)m * p.ldc + n;
        u32x2_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(cp) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// small-channel convolutions at the model boundaries (FLOPs negligible, VALU kernels)
// ---------------------------------------------------------------------------------------------
// 3x3, pad 1, stride 1, tiny Cin (3, 4): input NCHW f32, output NHWC T.  Weights [ky][kx][ci][Cout] f32.
// One thread = one output pixel x 4 output channels.
template <typename T, int CIN>
__global__ __launch_bounds__(256) void conv3x3_cin_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, T* __restrict__ y, int B,
                                                                int H, int Wd, int Cout, int copies) {
  const int q = Cout / 4;
  const size_t total = (size_t)B * H * Wd * q;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    int cq = (int)(idx % q);
    size_t pix = idx / q;
    int xw = (int)(pix % Wd);
    int yh = (int)((pix / Wd) % H);
    int n = (int)(pix / ((size_t)Wd * H));
    f32x4_t acc = *reinterpret_cast<const f32x4_t*>(bias + cq * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      int iy = yh + ky - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int ix = xw + kx - 1;
        if ((unsigned)ix >= (unsigned)Wd) continue;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          float v = x[(((size_t)n * CIN + ci) * H + iy) * Wd + ix];
          f32x4_t ww = *reinterpret_cast<const f32x4_t*>(w + ((size_t)((ky * 3 + kx) * CIN + ci)) * Cout + cq * 4);
          acc += v * ww;
        }
      }
    }
    float o[4] = {acc[0], acc[1], acc[2], acc[3]};
    // one 16-byte (f32) / 8-byte (bf16) store per copy: copy c of sample n is output sample n + c*B (the UNet input
    // under classifier-free guidance is the same latents twice, pipeline.py:147)
    for (int c = 0; c < copies; ++c) store_n<T, 4>(y + ((size_t)c * B * H * Wd + pix) * Cout + cq * 4, o);
  }
}

// 3x3, pad 1, stride 1, tiny Cout (3, 4, 8): input NHWC T, output NCHW f32.
// Weights [tap][Cin/V][COUT][V] in T.  8 lanes share one output pixel (each takes every 8th
// 16-byte channel chunk: 128 contiguous bytes per pixel per instruction), then a shuffle reduce.
// The whole filter (9*Cin*COUT elements, 46 KB for the UNet's f32 320->4 output conv) is staged in LDS once per
// block: COUT weight vectors per input vector straight from global memory made the kernel L1-bandwidth-bound
// (224 us for 16 x 64x64 pixels).
template <typename T, int COUT>
__global__ __launch_bounds__(256) void conv3x3_cout_small_kernel(const T* __restrict__ x, const T* __restrict__ wg,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 int B, int H, int Wd, int Cin) {
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char cs_smem[];
  T* w = reinterpret_cast<T*>(cs_smem);
  {
    const int nv = 9 * Cin * COUT / V;
    for (int i = threadIdx.x; i < nv; i += 256)
      reinterpret_cast<u32x4_t*>(w)[i] = reinterpret_cast<const u32x4_t*>(wg)[i];
  }
  __syncthreads();
  const int sub = threadIdx.x & 7;
  const int cvs = Cin / V;
  const size_t npix = (size_t)B * H * Wd;
  const size_t pix0 = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const size_t pstride = ((size_t)gridDim.x * 256) >> 3;
  for (size_t pixi = pix0; pixi < ((npix + 31) / 32) * 32; pixi += pstride) {  // keep 8-lane groups converged
    const bool live = pixi < npix;
    const size_t pix = live ? pixi : npix - 1;
    int xw = (int)(pix % Wd);
    int yh = (int)((pix / Wd) % H);
    int n = (int)(pix / ((size_t)Wd * H));
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      int iy = yh + ky - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        int ix = xw + kx - 1;
        if ((unsigned)ix >= (unsigned)Wd) continue;
        const T* src = x + (((size_t)n * H + iy) * Wd + ix) * Cin;
        const T* wt = w + (size_t)(ky * 3 + kx) * cvs * COUT * V;
        for (int cv = sub; cv < cvs; cv += 8) {
          float v[V];
          load_vec<T>(src + (size_t)cv * V, v);
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            float ww[V];
            load_vec<T>(wt + ((size_t)cv * COUT + o) * V, ww);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[o] += v[j] * ww[j];
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      float a = acc[o];
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 4, 64);
      acc[o] = a;
    }
    if (live && sub == 0) {
#pragma unroll
      for (int o = 0; o < COUT; ++o) y[(((size_t)n * COUT + o) * H + yh) * Wd + xw] = acc[o] + bias[o];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, bool CONV>
static int launch_gemm(GemmArgs& a, int batch, hipStream_t st) {
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const int lds = 2 * (BM + BN) * 128;
  static std::atomic<unsigned long long> lds_ok{0};
  if (lds > 48 * 1024 && sd_lds_opt_in(lds_ok, (const void*)gemm_kernel<T, BM, BN, CONV>, lds) != hipSuccess)
    return sdamd_set_error(2, "gemm: cannot reserve %d bytes of LDS", lds);
  // split-K for the parallelism-starved shapes (fp32 UNet at batch 2: M = 128 .. 2048 with K up to
  // 23040 leaves most CUs idle): aim at >= 512 workgroups with up to 16 splits (8 -> 16 on the 8x8-level convs of the
  // fp32 UNet at batch 2: -7...-10 % in situ), keep >= 8 K-tiles per split
  a.splits = 1;
  const long long blocks = (long long)((plan_rows(a.M) + BM - 1) / BM) * a.tiles_n * batch;   // (deterministic plans: one sample's rows)
  const int nk = a.K / (int)(128 / sizeof(T));
  const size_t slab = (size_t)a.M * a.N * sizeof(float);
  if (batch == 1 && a.ws && blocks < 256 && nk >= 16 && a.ldr % 4 == 0 && a.ldc % 4 == 0) {
    long long s = (512 + blocks - 1) / blocks;
    if (s > 16) s = 16;
    if (s > nk / 8) s = nk / 8;
    if (s * slab > a.ws_bytes) s = (long long)(a.ws_bytes / slab);
    if (s >= 2) a.splits = (int)s;
  }
  dim3 grid(a.tiles_m * a.tiles_n, batch, a.splits);
  gemm_kernel<T, BM, BN, CONV><<<grid, 256, lds, st>>>(a);
  SD_LAUNCH_CHECK("gemm");
  if (a.splits > 1) {
    GemmArgs r = a;
    if (sizeof(T) == 4) r.out_f32 = 1;   // f32 operands: C and the residual are f32
    return launch_splitk_reduce(r, st);
  }
  return 0;
}

// Tile choice (pure function of the problem size, also exported for the bench's bookkeeping):
// prefer 128x128; drop to smaller tiles when the grid would leave most of the 256 CUs idle or
// when N wastes a large part of a 128-wide tile.  Returns BM*1000 + BN.
static int choose_tile(int M, int N, int batch) {
  M = plan_rows(M);
  auto blocks = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
  const bool n64 = (N % 128) != 0 && (N % 128) <= 64;  // e.g. N = 320: 5 x 64 exact, 128 wastes 17 %
  if (!n64 && blocks(128, 128) >= 512) return 128128;
  if (blocks(128, 64) >= 512) return 128064;
  if (M > 64 && blocks(128, 64) >= 256) return 128064;
  return 64064;
}

// 0 = auto (8-wave LDS-DMA kernels of gemm2.hip where a tile of that family fits, else the 4-wave
// kernels of this file), 1 = this file only, 2 = gemm2.hip whenever one of its tiles divides N, however
// few workgroups that makes, 3 = as 2 with a 3-way K split (parity tests at small sizes).
int g_plan_samples = 1;
extern "C" int sdamd_set_plan_samples(int n) { int old = g_plan_samples; if (n >= 1) g_plan_samples = n; return old; }
static int g_gemm_impl = 0;
int g_gemm_debug = 0;
// f32 operands: 1 = products on the bf16 matrix pipe from three-term operand splits (six MFMAs, gemm_common.h
// split3_bf16: f32-rounding-level accuracy at ~2.7x the matrix rate of v_mfma_f32_16x16x4_f32), 0 = exact f32 MFMA.
int g_f32_emu = 1;
extern "C" int sdamd_set_f32_mode(int mode) { int old = g_f32_emu; if (mode == 0 || mode == 1) g_f32_emu = mode; return old; }
// Ping-pong kernels (gemm3.hip) for the 256x256 launches (GeGLU GEMMs, VAE / CLIP widths): bit 0 = on (default),
// bit 2 = the variant that re-reads the B0 fragments; bit 1 = the big-wave halo conv (gemm5.hip) OFF, bit 3 = every
// halo-staged conv (gemm4.hip, gemm5.hip) OFF (A/B runs and tests).  sdamd_set_gemm_impl carries `g_gemm_pp ^ 1` in bits 20-23, so an
// impl word without those bits (every existing caller) keeps the default and bit 20 switches the kernels OFF (A/B runs).
int g_gemm_pp = 1;
extern "C" int sdamd_set_gemm_impl(int impl) {
  int old = g_gemm_impl | (g_gemm_debug << 8) | ((g_gemm_pp ^ 1) << 20);
  g_gemm_impl = impl & 255; g_gemm_debug = (impl >> 8) & 0xfff; g_gemm_pp = ((impl >> 20) & 15) ^ 1;
  return old;
}

// Weights of an f32 GEMM / conv3x3, split once into the three bf16 terms of the bf16x3 mode (common.h split3_pair):
// planes[row][k/32][hi|mid|lo][32], the 32 k of a K-tile stored in the order the MFMA fragments of the f32 activation
// operand take them (16-byte chunk q holds k = 4q..4q+3 and 16+4q..16+4q+3), so a kernel reads ready operands.
__global__ __launch_bounds__(256) void split3_weights_kernel(const float* __restrict__ w, unsigned* __restrict__ planes, long long rows,
                                                             int K, int ldw) {
  const long long pairs = rows * (K >> 1);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < pairs; idx += (long long)gridDim.x * 256) {
    const long long row = idx / (K >> 1);
    const int k = (int)(idx - row * (K >> 1)) * 2;
    const float* src = w + row * ldw + k;
    const Split3 t = split3_pair(src[0], src[1]);
    const int kt = k >> 5, kk = k & 31;
    const int slot = ((kk & 15) >> 2) * 8 + (kk >> 4) * 4 + (kk & 3);      // even, with slot + 1 = the pair's second value
    unsigned* dst = planes + (row * 3 * K + (long long)kt * 96 + slot) / 2;
    dst[0] = t.h; dst[16] = t.m; dst[32] = t.l;
  }
}
extern "C" int sdamd_split3_weights(const float* w, void* planes, long long rows, int K, int ldw, void* stream) {
  SD_CHECK(w && planes && rows > 0 && K > 0 && K % 32 == 0 && ldw >= K && ldw % 2 == 0, "split3_weights: rows=%lld K=%d ldw=%d", rows, K, ldw);
  SD_CHECK(((uintptr_t)w % 8) == 0 && ((uintptr_t)planes % 16) == 0, "split3_weights: alignment");
  const long long pairs = rows * (K >> 1);
  const int g = (int)((pairs + 255) / 256 > 4096 ? 4096 : (pairs + 255) / 256);
  split3_weights_kernel<<<g, 256, 0, (hipStream_t)stream>>>(w, (unsigned*)planes, rows, K, ldw);
  SD_LAUNCH_CHECK("split3_weights");
  return 0;
}

// 1 when sdamd_gemm can run `act` fused for this dtype / N with the current kernel family selection
extern "C" int sdamd_gemm_supports_act(int act, int dtype, int N) {
  if (act != ACT_GEGLU) return act >= ACT_NONE && act <= ACT_SILU;
  return dtype == SDAMD_BF16 && g_gemm_impl != 1 && N % 256 == 0;
}

extern "C" int sdamd_gemm_tile(int M, int N, int K, int batch, int dtype, size_t ws_bytes, int act) {
  if (g_gemm_impl != 1) {
    int splits = 1;
    if (act == ACT_GEGLU && dtype == SDAMD_BF16) return 256256;
    if (int t2 = gemm2_choose_tile(M, N, K, batch, g_gemm_impl, ws_bytes, &splits, dtype == SDAMD_BF16 ? 64 : 32))
      return t2 + 1000000 * (splits - 1);
  }
  return choose_tile(M, N, batch);
}

// Host-only, as sdamd_gemm_tile: the image width (64 / 32 / 16) when sdamd_conv3x3 runs this problem on a halo-staged
// kernel (gemm4.hip; + 256 when it is the 256-row big-wave kernel of gemm5.hip), else 0.
extern "C" int sdamd_conv3x3_halo(int B, int H, int Wd, int Cin, int Cout, int stride, int pad_mode, int upsample, int dtype,
                                  size_t ws_bytes) {
  if (g_gemm_impl == 1 || dtype != SDAMD_BF16 || upsample == 1 || pad_mode != 0 || stride != 1 || B <= 0 || H <= 0 || Wd <= 0 || Cin % 64 != 0) return 0;
  if (upsample == 2) {   // phase form: one batched launch of the four phases (the K-split per-phase launches stay on the gathered kernel)
    GemmArgs u = {};
    u.M = B * H * Wd; u.N = Cout; u.K = 4 * Cin; u.ldw = u.K;
    u.H = u.Ho = H; u.Wd = u.Wo = Wd; u.Cin = Cin; u.stride = 1; u.pad = 1; u.phase_up = 1;
    if (ws_bytes && 4LL * ((plan_rows(u.M) + 127) / 128) * ((u.N + 159) / 160) < 256) {
      int sp = 1;
      if (gemm2_choose_tile(u.M, u.N, u.K, 1, g_gemm_impl, ws_bytes, &sp, 64) && sp > 1) return 0;
    }
    int stages = 2;
    if (gemm2_choose_tile(u.M, u.N, u.K, 4, g_gemm_impl, ws_bytes, &u.splits, 64, &stages) != 128160 || stages != 2) return 0;
    return gemm2hbw_takes(u, 4) ? (Wd | 0x300) : 0;   // bits 8 + 9: the big-wave kernel's phase form (gemm2hbwup_kernel)
  }
  GemmArgs a = {};
  a.M = B * H * Wd; a.N = Cout; a.K = 9 * Cin; a.ldw = a.K;
  a.H = a.Ho = H; a.Wd = a.Wo = Wd; a.Cin = Cin; a.stride = 1; a.pad = 1;
  int stages = 2;
  if (gemm2_choose_tile(a.M, a.N, a.K, 1, g_gemm_impl, ws_bytes, &a.splits, 64, &stages) != 128160 || stages != 2) return 0;
  if (Wd == 8) return gemm2hbw_takes(a, 1) ? (8 | 0x100) : 0;   // 8 x 8 images: the big-wave kernel only
  const int w = gemm2halo_width(a, 1);
  return w && gemm2hbw_takes(a, 1) ? w | 0x100 : w;   // bit 8: the 256-row big-wave kernel (gemm5.hip)
}

template <typename T, bool CONV>
static int pick_tile(GemmArgs& a, int batch, hipStream_t st) {
  switch (choose_tile(a.M, a.N, batch)) {
    case 128128: return launch_gemm<T, 128, 128, CONV>(a, batch, st);
    case 128064: return launch_gemm<T, 128, 64, CONV>(a, batch, st);
    default: return launch_gemm<T, 64, 64, CONV>(a, batch, st);
  }
}

static int check_common(const GemmArgs& a, int dtype, const char* what) {
  const int esz = dtype == SDAMD_BF16 ? 2 : 4;
  const int bke = 128 / esz;
  SD_CHECK(dtype == SDAMD_BF16 || dtype == SDAMD_F32, "%s: bad dtype %d", what, dtype);
  SD_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "%s: empty problem M=%d N=%d K=%d", what, a.M, a.N, a.K);
  SD_CHECK(a.K % bke == 0, "%s: K=%d must be a multiple of %d", what, a.K, bke);
  SD_CHECK(a.N % 16 == 0, "%s: N=%d must be a multiple of 16", what, a.N);
  SD_CHECK(a.A && a.W && a.C, "%s: null pointer", what);
  SD_CHECK(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0 && ((uintptr_t)a.C % 16) == 0, "%s: pointers must be 16-byte aligned", what);
  SD_CHECK((a.ldw * esz) % 16 == 0 && a.ldc % 4 == 0, "%s: leading dimensions must keep 16-byte alignment", what);
  return 0;
}

extern "C" int sdamd_gemm(const void* A, const void* W, void* C, const float* bias, const void* residual, int M, int N, int K,
                          int lda, int ldw, int ldc, int ldr, int batch, long long strideA, long long strideW,
                          long long strideC, long long strideR, int act, int dtype, int out_f32, void* ws,
                          size_t ws_bytes, float* colstats, int* colstats_written, const void* w_planes, void* stream) {
  if (colstats_written) *colstats_written = 0;
  GemmArgs a = {};
  a.colstats = colstats;
  a.splits = 1; a.ws = (float*)ws; a.ws_bytes = ws_bytes; a.debug = g_gemm_debug;
  a.A = A; a.W = W; a.W3 = dtype == SDAMD_F32 ? w_planes : nullptr; a.C = C; a.bias = bias; a.R = residual;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
  a.sA = strideA; a.sW = strideW; a.sC = strideC; a.sR = strideR;
  a.act = act; a.out_f32 = out_f32;
  if (int e = check_common(a, dtype, "gemm")) return e;
  const int esz = dtype == SDAMD_BF16 ? 2 : 4;
  SD_CHECK((lda * esz) % 16 == 0 && lda >= K && ldw >= K && ldc >= (act == ACT_GEGLU ? N / 2 : out_f32 == 2 ? 2 * N : N), "gemm: bad leading dimensions lda=%d ldw=%d ldc=%d", lda, ldw, ldc);
  SD_CHECK(out_f32 >= 0 && out_f32 <= 2, "gemm: out_f32=%d (0 operand type, 1 f32, 2 bf16 hi|lo)", out_f32);
  SD_CHECK(batch >= 1, "gemm: batch=%d", batch);
  SD_CHECK(!residual || (ldr >= N && ldr % 4 == 0 && ((uintptr_t)residual % 16) == 0), "gemm: residual needs ldr=%d >= N, ldr %% 4 == 0 and 16-byte alignment", ldr);
  hipStream_t st = (hipStream_t)stream;
  if (g_gemm_impl != 1) {
    const int r = launch_gemm2(a, dtype, batch, false, g_gemm_impl, st);
    if (r >= 0) { if (colstats_written && r == 0) *colstats_written = a.colstats_done; return r; }
  }
  SD_CHECK(act != ACT_GEGLU, "gemm: the fused GeGLU epilogue needs bf16, N %% 256 == 0, no residual, bf16 output and the 8-wave kernels");
  SD_CHECK(out_f32 != 2, "gemm: the hi|lo output needs bf16 operands and a shape the LDS-DMA kernels take (sdamd_gemm_tile != 0)");
  if (dtype == SDAMD_BF16) return pick_tile<bf16_t, false>(a, batch, st);
  if (g_f32_emu) return pick_tile<f32e_t, false>(a, batch, st);
  return pick_tile<float, false>(a, batch, st);
}

// out [M][N] f32 = A [M][K] W[N][K]^T + bias (+ residual f32), and ln_out [M][N] bf16 = LayerNorm(out row) * gamma + beta
// from the same launch (gemm2_kernel.h EPI 3).  bf16 operands, N == 320 (the 128x320 tile holds whole rows).
extern "C" int sdamd_gemm_ln(const void* A, const void* W, float* out, const float* bias, const float* residual, void* ln_out,
                             const float* ln_gamma, const float* ln_beta, float ln_eps, int M, int N, int K, int lda,
                             void* stream) {
  SD_CHECK(N == 320, "gemm_ln: N=%d (the fused LayerNorm needs whole rows in one 320-wide tile)", N);
  SD_CHECK(ln_out && ln_gamma && ln_beta && out, "gemm_ln: null pointer");
  SD_CHECK(((uintptr_t)ln_out % 16) == 0 && ((uintptr_t)ln_gamma % 16) == 0 && ((uintptr_t)ln_beta % 16) == 0, "gemm_ln: alignment");
  GemmArgs a = {};
  a.splits = 1; a.debug = g_gemm_debug;
  a.A = A; a.W = W; a.C = out; a.bias = bias; a.R = residual;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = K; a.ldc = N; a.ldr = N;
  a.act = ACT_NONE; a.out_f32 = 1;
  a.ln_out = ln_out; a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_eps = ln_eps;
  if (int e = check_common(a, SDAMD_BF16, "gemm_ln")) return e;
  SD_CHECK((lda * 2) % 16 == 0 && lda >= K, "gemm_ln: lda=%d", lda);
  SD_CHECK(!residual || ((uintptr_t)residual % 16) == 0, "gemm_ln: residual alignment");
  SD_CHECK(g_gemm_impl != 1, "gemm_ln: needs the LDS-DMA kernels");
  const int r = launch_gemm2(a, SDAMD_BF16, 1, false, g_gemm_impl, (hipStream_t)stream);
  SD_CHECK(r >= 0, "gemm_ln: no kernel takes M=%d N=%d K=%d", M, N, K);
  return r;
}

// conv3x3 over NHWC.  pad_mode 0: symmetric pad 1 (diffusion.py / decoder.py convs);
// pad_mode 1: pad bottom/right only (encoder.py:65-66 F.pad(0,1,0,1) + stride-2 pad-0 conv).
// upsample=1 reads the input through a nearest 2x upsample (diffusion.py:148-150, decoder.py:90-111).
extern "C" int sdamd_conv3x3(const void* x, const void* w, void* y, const float* bias, const void* residual, int B, int H,
                             int Wd, int Cin, int Cout, int stride, int pad_mode, int upsample, int act, int dtype,
                             int out_f32, void* ws, size_t ws_bytes, float* colstats, int* colstats_written, const void* w_planes,
                             void* stream) {
  if (colstats_written) *colstats_written = 0;
  if (dtype != SDAMD_F32) w_planes = nullptr;
  SD_CHECK(stride == 1 || stride == 2, "conv3x3: stride=%d", stride);
  SD_CHECK(pad_mode == 0 || pad_mode == 1, "conv3x3: pad_mode=%d", pad_mode);
  SD_CHECK(upsample >= 0 && upsample <= 2, "conv3x3: upsample=%d", upsample);
  SD_CHECK(B > 0 && H > 0 && Wd > 0, "conv3x3: empty input");
  hipStream_t st = (hipStream_t)stream;
  if (upsample == 2) {
    // nearest-2x upsample + conv3x3 (pad 1, stride 1) as four 2x2 convs, one per output phase:
    // w is [4 phases (py,px)][Cout][2][2][Cin] with the coinciding 3x3 taps pre-summed (engine.py)
    SD_CHECK(g_gemm_impl != 1 && stride == 1 && pad_mode == 0 && !residual,
             "conv3x3: the phase-decomposed upsample form needs the 8-wave kernels, stride 1, pad 1, no residual");
    SD_CHECK(Cin % (dtype == SDAMD_BF16 ? 64 : 32) == 0, "conv3x3: Cin=%d must be a multiple of the K-tile", Cin);
    GemmArgs a = {};
    a.A = x; a.W = w; a.W3 = w_planes; a.C = y; a.bias = bias; a.R = nullptr;
    a.M = B * H * Wd; a.N = Cout; a.K = 4 * Cin;
    a.lda = 0; a.ldw = 4 * Cin; a.ldc = Cout; a.ldr = Cout;
    a.sA = 0; a.sW = (long long)Cout * 4 * Cin; a.sC = 0; a.sR = 0;
    a.act = act; a.out_f32 = out_f32;
    a.splits = 1; a.ws = nullptr; a.ws_bytes = 0; a.debug = g_gemm_debug;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = H; a.Wo = Wd; a.stride = 1; a.pad = 1; a.ups = 0; a.phase_up = 1;
    if (int e = check_common(a, dtype, "conv3x3")) return e;
    // Few low-resolution rows (UNet batch 2: 8x8 .. 32x32 maps): four phases of a handful of tiles leave the chip idle
    // and a batched launch cannot split K.  Then each phase is a launch of its own with the planner's K split and the
    // reduce kernel placing the rows (fp32 UNet batch 2: the three upsample convs 1.62 -> ~0.6 ms per step).
    if (g_gemm_impl != 1 && ws && 4LL * ((plan_rows(a.M) + 127) / 128) * ((a.N + 159) / 160) < 256) {   // (256 workgroups fill the chip once: measured, 256 keep the batched launch)
      int splits = 1;
      const int tile = gemm2_choose_tile(a.M, a.N, a.K, 1, g_gemm_impl, ws_bytes, &splits, dtype == SDAMD_BF16 ? 64 : 32);
      if (tile && splits > 1) {
        const size_t esz = dtype == SDAMD_BF16 ? 2 : 4;
        for (int ph = 0; ph < 4; ++ph) {
          GemmArgs b = a;
          b.W = (const char*)w + (size_t)ph * a.sW * esz;
          if (w_planes) b.W3 = (const char*)w_planes + (size_t)ph * a.sW * 3 * 2;
          b.phase_base = ph;
          b.ws = (float*)ws; b.ws_bytes = ws_bytes;
          const int r = launch_gemm2(b, dtype, 1, true, g_gemm_impl, st);
          SD_CHECK(r >= 0, "conv3x3: phase launch failed for Cout=%d", Cout);
          if (r) return r;
        }
        return 0;
      }
    }
    int r = launch_gemm2(a, dtype, 4, true, g_gemm_impl, st);
    if (r < 0) r = launch_gemm2(a, dtype, 4, true, 2, st);   // few workgroups: any tile that divides Cout
    SD_CHECK(r >= 0, "conv3x3: no 8-wave tile divides Cout=%d", Cout);
    return r;
  }
  const int Hin = H << upsample, Win = Wd << upsample;
  const int pad = pad_mode == 0 ? 1 : 0;
  const int Ho = pad_mode == 0 ? (Hin + 2 - 3) / stride + 1 : (Hin + 1 - 3) / stride + 1;
  const int Wo = pad_mode == 0 ? (Win + 2 - 3) / stride + 1 : (Win + 1 - 3) / stride + 1;
  GemmArgs a = {};
  a.A = x; a.W = w; a.W3 = w_planes; a.C = y; a.bias = bias; a.R = residual;
  a.M = B * Ho * Wo; a.N = Cout; a.K = 9 * Cin;
  a.lda = 0; a.ldw = 9 * Cin; a.ldc = Cout; a.ldr = Cout;
  a.act = act; a.out_f32 = out_f32;
  a.splits = 1; a.ws = (float*)ws; a.ws_bytes = ws_bytes; a.debug = g_gemm_debug;
  a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad = pad; a.ups = upsample;
  a.colstats = colstats;
  if (int e = check_common(a, dtype, "conv3x3")) return e;
  const int bke = dtype == SDAMD_BF16 ? 64 : 32;
  SD_CHECK(Cin % bke == 0, "conv3x3: Cin=%d must be a multiple of %d (use the small-channel kernels)", Cin, bke);
  if (g_gemm_impl != 1) {
    const int r = launch_gemm2(a, dtype, 1, true, g_gemm_impl, st);
    if (r >= 0) { if (colstats_written && r == 0) *colstats_written = a.colstats_done; return r; }
  }
  if (dtype == SDAMD_BF16) return pick_tile<bf16_t, true>(a, 1, st);
  if (g_f32_emu) return pick_tile<f32e_t, true>(a, 1, st);
  return pick_tile<float, true>(a, 1, st);
}

extern "C" int sdamd_conv3x3_cin_small(const float* x_nchw, const float* w_tap_ci_co, const float* bias, void* y_nhwc, int B,
                                       int H, int Wd, int Cin, int Cout, int copies, int dtype, void* stream) {
  SD_CHECK(Cin == 3 || Cin == 4, "conv3x3_cin_small: Cin=%d (3 or 4)", Cin);
  SD_CHECK(Cout % 4 == 0 && B > 0 && H > 0 && Wd > 0 && copies >= 1, "conv3x3_cin_small: bad shape");
  hipStream_t st = (hipStream_t)stream;
  size_t total = (size_t)B * H * Wd * (Cout / 4);
  int g = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
#define CS(TT, CI) conv3x3_cin_small_kernel<TT, CI><<<g, 256, 0, st>>>(x_nchw, w_tap_ci_co, bias, (TT*)y_nhwc, B, H, Wd, Cout, copies)
  if (dtype == SDAMD_BF16) { if (Cin == 3) CS(bf16_t, 3); else CS(bf16_t, 4); }
  else { if (Cin == 3) CS(float, 3); else CS(float, 4); }
#undef CS
  SD_LAUNCH_CHECK("conv3x3_cin_small");
  return 0;
}

extern "C" int sdamd_conv3x3_cout_small(const void* x_nhwc, const void* w_packed, const float* bias, float* y_nchw, int B,
                                        int H, int Wd, int Cin, int Cout, int dtype, void* stream) {
  const int V = dtype == SDAMD_BF16 ? 8 : 4;
  SD_CHECK(Cout == 3 || Cout == 4 || Cout == 8, "conv3x3_cout_small: Cout=%d (3, 4 or 8)", Cout);
  SD_CHECK(Cin % V == 0 && B > 0 && H > 0 && Wd > 0, "conv3x3_cout_small: bad shape");
  hipStream_t st = (hipStream_t)stream;
  size_t npix = (size_t)B * H * Wd;
  size_t blocks = (npix * 8 + 255) / 256;
  int g = (int)(blocks > 2048 ? 2048 : blocks);   // every block stages the filter once, then strides over pixels
  const size_t lds = (size_t)9 * Cin * Cout * (dtype == SDAMD_BF16 ? 2 : 4);
  SD_CHECK(lds <= 160 * 1024, "conv3x3_cout_small: filter of %zu bytes does not fit the LDS", lds);
#define CS(TT, CO)                                                                                                     \
  do {                                                                                                                 \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv3x3_cout_small_kernel<TT, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    conv3x3_cout_small_kernel<TT, CO><<<g, 256, lds, st>>>((const TT*)x_nhwc, (const TT*)w_packed, bias, y_nchw, B, H, Wd, Cin); \
  } while (0)
  if (dtype == SDAMD_BF16) { if (Cout == 3) CS(bf16_t, 3); else if (Cout == 4) CS(bf16_t, 4); else CS(bf16_t, 8); }
  else { if (Cout == 3) CS(float, 3); else if (Cout == 4) CS(float, 4); else CS(float, 8); }
#undef CS
  SD_LAUNCH_CHECK("conv3x3_cout_small");
  return 0;
}
