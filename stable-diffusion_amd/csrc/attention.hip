// Launchers and the remaining attention kernels; attn_body (the generic flash kernel) lives in attention_body.h,
// shared with attention2.hip.
#include "attention_body.h"

template <typename T, int D, bool TRACK = false>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs p) { attn_body<T, D, TRACK>(p); }

int launch_attn40p(const AttnArgs& a, int B, hipStream_t st);   // attention2.hip
int launch_xattn(const AttnArgs& a, int B, int d, hipStream_t st);   // attention3.hip

template <typename T, int D, bool TRACK = false>
static int launch_attn(const AttnArgs& a, int B, hipStream_t st) {
  dim3 grid((a.Sq + 127) / 128, B * a.H);
  constexpr int lds = 2 * AttnCfg<T, D>::STAGE + 16;   // + the fast path's redo flag
  static std::atomic<unsigned long long> lds_ok{0};
  if (lds > 48 * 1024 && sd_lds_opt_in(lds_ok, (const void*)attn_kernel<T, D, TRACK>, lds) != hipSuccess)
    return sdamd_set_error(2, "attention: cannot reserve %d bytes of LDS", lds);
  attn_kernel<T, D, TRACK><<<grid, 256, lds, st>>>(a);
  SD_LAUNCH_CHECK("attention");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// f32 attention on the bf16 matrix pipe ("bf16x3", the attention side of sdamd_set_f32_mode, include/sdamd.h).
// Same algorithm and workgroup shape as attn_kernel (128 queries per workgroup, 32 per wave, S^T = K Q^T, online softmax
// with a lazy reference, ones column in V for the denominator), f32 in / f32 out, but every f32 operand is split exactly
// into three bf16 terms and each product is accumulated in f32 from its six partial products >= 2^-16 (hi*hi, hi*mid,
// mid*hi, hi*lo, lo*hi, mid*mid; the dropped ones are below one f32 rounding of the product):
//   Q   scaled by log2(e)/sqrt(d) in f32 and split once per query, in registers;
//   K,V split ONCE PER WORKGROUP when a prefetched tile is committed to LDS, as three bf16 planes each (so the four
//       waves do not repeat the split, and V^T fragments still come from ds_read_b64_tr_b16);
//   P   = exp2(s - m), split in registers per 32x32 tile (the only per-wave split work: 88 VALU per tile).
// Per 32-key x 32-query tile at d = 40: 42 v_mfma_f32_32x32x16_bf16 (1344 cycles) instead of 52 v_mfma_f32_32x32x2_f32
// (3328 cycles).  32-key tiles, two LDS stages.
// ---------------------------------------------------------------------------------------------
template <int D> struct AttnX3Cfg {
  static constexpr int DK = (D + 15) / 16 * 16;
  static constexpr int DV = (D + 1 + 31) / 32 * 32;                   // always one spare column: V[:, D] = 1
  static constexpr int KS = DK * 2 + 16;                              // bf16 plane row strides (bytes), as AttnCfg<bf16_t>
  static constexpr int VS = (DV * 2) % 256 == 64 || (DV * 2) % 256 == 192 ? DV * 2 : DV * 2 + 64;
  static constexpr int BKV = 32;
  static constexpr int KP = BKV * KS, VP = BKV * VS;                  // one plane of a staged tile
  static constexpr int STAGE = 3 * (KP + VP);                         // [K hi | K mid | K lo | V hi | V mid | V lo]
};

template <int D>
__global__ __launch_bounds__(256) void attn_f32x3_kernel(const AttnArgs p) {
  typedef AttnX3Cfg<D> Cfg;
  constexpr int DK = Cfg::DK, DV = Cfg::DV, KS = Cfg::KS, VS = Cfg::VS, BKV = Cfg::BKV, KP = Cfg::KP, VP = Cfg::VP;
  constexpr int STAGE = Cfg::STAGE;
  constexpr int NQ = DK / 16, NB = DV / 32;
  constexpr int DC = D / 4;                                  // 16-byte f32 chunks per global row
  constexpr int L_BLK = D / 32, L_REG = ((D % 32) & 3) + 4 * ((D % 32) >> 3), L_HALF = ((D % 32) >> 2) & 1;
  static_assert(D % 8 == 0 && D < DV, "head dim");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y / p.H, head = blockIdx.y % p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = q0 + r;
  const float* __restrict__ Qp = (const float*)p.Q + (size_t)b * p.q_bs + (size_t)head * D;
  const float* __restrict__ Kp = (const float*)p.K + (size_t)b * p.k_bs + (size_t)head * D;
  const float* __restrict__ Vp = (const float*)p.V + (size_t)b * p.v_bs + (size_t)head * D;

  // Q fragments (B operand of S^T = K Q^T): lane (r, h) holds Q[qrow][16s + 8h .. +7], pre-scaled, three planes
  bf16x8_t qh[NQ], qm[NQ], ql[NQ];
  {
    const float* qr = Qp + (size_t)min(qrow, p.Sq - 1) * p.q_rs;
    const float sc = p.scale_log2e;
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
      const int d0 = 16 * s + 8 * h;
      f32x4_t a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
      if (d0 < D) { a = *reinterpret_cast<const f32x4_t*>(qr + d0); c = *reinterpret_cast<const f32x4_t*>(qr + d0 + 4); }
      split3_bf16(a * sc, c * sc, qh[s], qm[s], ql[s]);
    }
  }

  // K/V tile: f32 chunks prefetched into registers, split into planes when committed
  constexpr int NCH = BKV * DC;
  constexpr int PT = (NCH + 255) / 256;
  f32x4_t rk[PT], rv[PT];
  auto issue = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = t + 256 * i;
      if (NCH % 256 == 0 || c < NCH) {
        const int row = c / DC, ch = c - row * DC;
        const size_t grow = (size_t)min(kv0 + row, p.Skv - 1);
        rk[i] = *reinterpret_cast<const f32x4_t*>(Kp + grow * p.k_rs + ch * 4);
        rv[i] = *reinterpret_cast<const f32x4_t*>(Vp + grow * p.v_rs + ch * 4);
      }
    }
  };
  auto commit = [&](int stage) {
    char* ks = smem + stage * STAGE;
    char* vs = ks + 3 * KP;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = t + 256 * i;
      if (NCH % 256 == 0 || c < NCH) {
        const int row = c / DC, ch = c - row * DC;
        const Split3 k0 = split3_pair(rk[i][0], rk[i][1]), k1 = split3_pair(rk[i][2], rk[i][3]);
        const Split3 v0 = split3_pair(rv[i][0], rv[i][1]), v1 = split3_pair(rv[i][2], rv[i][3]);
        const u32x2_t kh = {k0.h, k1.h}, km = {k0.m, k1.m}, kl = {k0.l, k1.l}, vh = {v0.h, v1.h}, vm = {v0.m, v1.m}, vl = {v0.l, v1.l};
        char* kd = ks + row * KS + ch * 8;
        char* vd = vs + row * VS + ch * 8;
        *reinterpret_cast<u32x2_t*>(kd) = kh; *reinterpret_cast<u32x2_t*>(kd + KP) = km; *reinterpret_cast<u32x2_t*>(kd + 2 * KP) = kl;
        *reinterpret_cast<u32x2_t*>(vd) = vh; *reinterpret_cast<u32x2_t*>(vd + VP) = vm; *reinterpret_cast<u32x2_t*>(vd + 2 * VP) = vl;
      }
    }
  };
  {  // constant pad chunks of both stages and all planes: zeros, and V_hi[:, D] = 1
    constexpr int KPADC = (DK - D) / 8, VPADC = (DV - D) / 8;   // 16-byte chunks per row
    constexpr int NPAD = KPADC + VPADC;
    for (int c = t; c < 2 * 3 * BKV * NPAD; c += 256) {
      const int pc = c % NPAD, rest = c / NPAD;
      const int row = rest % BKV, pl = (rest / BKV) % 3, stage = rest / (3 * BKV);
      char* base = smem + stage * STAGE;
      u32x4_t z = {0u, 0u, 0u, 0u};
      if (pc < KPADC) {
        *reinterpret_cast<u32x4_t*>(base + pl * KP + row * KS + D * 2 + pc * 16) = z;
      } else {
        const int vc = pc - KPADC;
        if (vc == 0 && pl == 0) z[0] = 0x00003F80u;
        *reinterpret_cast<u32x4_t*>(base + 3 * KP + pl * VP + row * VS + D * 2 + vc * 16) = z;
      }
    }
  }

  f32x16_t o[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -1.0e30f;

  int kv_end = p.Skv;
  if (p.causal) kv_end = min(p.Skv, (int)blockIdx.x * 128 + 128);
  const int ntiles = (kv_end + BKV - 1) / BKV;

  issue(0);
  commit(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int kv0 = kt * BKV;
    const char* ks = smem + (kt & 1) * STAGE;
    const char* vs = ks + 3 * KP;
    if (kt + 1 < ntiles) issue(kv0 + BKV);

    // S^T = K Q^T (log2 units: Q carries the scale), small terms first
    f32x16_t s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < NQ; ++st) {
      const char* ka = ks + r * KS + (16 * st + 8 * h) * 2;
      const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(ka);
      const bf16x8_t km = *reinterpret_cast<const bf16x8_t*>(ka + KP);
      const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(ka + 2 * KP);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qm[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qh[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qm[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[st], s, 0, 0, 0);
    }

    // online softmax, lazy reference (moves only when a row outgrows it by 2^RESCALE_THR)
    if (kv0 + BKV > p.Skv || p.causal) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= p.Skv || (p.causal && key > qrow)) s[e] = -1.0e30f;
      }
    }
    float mx = -1.0e30f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = mx > m_run + RESCALE_THR ? mx : m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    }
    // P^T fragments (B operand): element pairs (e, e+1), three planes
    u32x4_t ph[2], pm[2], pl[2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p0 = __builtin_amdgcn_exp2f(s[8 * st + 2 * j] - m_new), p1 = __builtin_amdgcn_exp2f(s[8 * st + 2 * j + 1] - m_new);
        const Split3 sp = split3_pair(p0, p1);
        ph[st][j] = sp.h; pm[st][j] = sp.m; pl[st][j] = sp.l;
      }

    // O^T += V^T P^T; V^T fragments by transpose reads from the bf16 planes (see attn_kernel)
    {
      const int g16 = lane >> 4, li = lane & 15;
      const char* vbase = vs + (4 * (g16 >> 1) + (li >> 2)) * VS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      auto vfrag = [&](const char* a) {
        const s16x4_t lo = lds_read_tr16(a), hi = lds_read_tr16(a + 8 * VS);
        const s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, vv);
      };
#pragma unroll
      for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const char* va = vbase + 16 * st * VS + i * 64;
          const bf16x8_t vh = vfrag(va), vm = vfrag(va + VP), vl = vfrag(va + 2 * VP);
          const bf16x8_t Ph = __builtin_bit_cast(bf16x8_t, ph[st]), Pm = __builtin_bit_cast(bf16x8_t, pm[st]), Pl = __builtin_bit_cast(bf16x8_t, pl[st]);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, Ph, o[i], 0, 0, 0);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, Pl, o[i], 0, 0, 0);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, Pm, o[i], 0, 0, 0);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, Ph, o[i], 0, 0, 0);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, Pm, o[i], 0, 0, 0);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, Ph, o[i], 0, 0, 0);
        }
      }
    }

    if (kt + 1 < ntiles) commit((kt + 1) & 1);
    __syncthreads();
  }

  // epilogue: O[q][d] = O^T[d][q] / l, l = O^T[D][q] (the ones column)
  if (qrow < p.Sq) {
    const float l = __shfl(o[L_BLK][L_REG], r + 32 * L_HALF, 64);
    const float inv = 1.0f / l;
    float* op = (float*)p.O + (size_t)b * p.o_bs + (size_t)qrow * p.o_rs + (size_t)head * D;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * i + 8 * g + 4 * h;
        if (d0 < D) *reinterpret_cast<f32x4_t*>(op + d0) = (f32x4_t){o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
      }
  }
}

template <int D>
static int launch_attn_f32x3(const AttnArgs& a, int B, hipStream_t st) {
  dim3 grid((a.Sq + 127) / 128, B * a.H);
  constexpr int lds = 2 * AttnX3Cfg<D>::STAGE;
  static_assert(lds <= 160 * 1024, "attention f32x3: LDS budget");
  static std::atomic<unsigned long long> lds_ok{0};
  if (lds > 48 * 1024 && sd_lds_opt_in(lds_ok, (const void*)attn_f32x3_kernel<D>, lds) != hipSuccess)
    return sdamd_set_error(2, "attention: cannot reserve %d bytes of LDS", lds);
  attn_f32x3_kernel<D><<<grid, 256, lds, st>>>(a);
  SD_LAUNCH_CHECK("attention");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Flash attention for the VAE's single 512-wide head (decoder.py:39-70 -> attention.py:46-68; S = 4096 at 512^2,
// 9216 at 768^2).  An O^T accumulator of 512 rows x 32 queries does not fit one wave's registers, so the head dim is
// split over the workgroup's 4 waves ("split-d accumulators"):
//   workgroup = 32 queries of one (batch, head); wave w owns head-dim slice [128w, 128w+128)
//   per 32-key tile:  S^T_w = K[:, slice] Q[:, slice]^T      (partial scores over the wave's slice of the contraction)
//                     partials exchanged through LDS, every wave sums the four -> full S^T (same order in every wave,
//                     so all four waves hold bit-identical scores and soft-max state)
//                     online softmax (redundant per wave: 16 exp per lane)
//                     O^T[slice] += V[:, slice]^T P^T         (the wave's 128 output columns)
// No [S, S] tensor exists anywhere.  bf16: 32x32x16 MFMA, K/V tiles double-buffered (register prefetch), V^T via
// ds_read_b64_tr_b16.  f32: exact 32x32x2 MFMA, one stage (a 32-key f32 tile of K and V is 128 KB).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn512_kernel(const AttnArgs p) {
  constexpr int D = 512, SL = 128;                       // head dim, per-wave slice
  constexpr bool BF = sizeof(T) == 2;
  constexpr int EPC = 16 / sizeof(T);
  constexpr int KS = D * (int)sizeof(T) + 16;            // K row stride (bytes): odd multiple of 16
  constexpr int VS = BF ? D * 2 + 64 : D * 4;            // V row stride (bf16: 4 rows x 64 B tile a bank row for the transposed read)
  constexpr int BKV = 32;
  constexpr int STAGES = BF ? 2 : 1;
  constexpr int STAGE = BKV * (KS + VS);
  constexpr int DC = D / EPC;                            // 16-byte chunks per row
  constexpr int NCH = BKV * DC;                          // chunks per K (or V) tile
  constexpr int PT = NCH / 256;                          // chunks per thread
  constexpr int NQ = BF ? SL / 16 : SL / 8;              // Q fragments of the wave's slice (4 VGPR each)
  constexpr int NB = SL / 32;                            // 32-row output blocks of the wave's slice
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xs = smem + STAGES * STAGE;                // partial-score exchange: [4 waves][4][64 lanes][16 B]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y / p.H, head = blockIdx.y % p.H;
  const int q0 = blockIdx.x * 32;
  const int qrow = q0 + r;
  const T* __restrict__ Qp = (const T*)p.Q + (size_t)b * p.q_bs + (size_t)head * D;
  const T* __restrict__ Kp = (const T*)p.K + (size_t)b * p.k_bs + (size_t)head * D;
  const T* __restrict__ Vp = (const T*)p.V + (size_t)b * p.v_bs + (size_t)head * D;
  const int d_w = wave * SL;

  u32x4_t qf[NQ];
  {
    const T* qr = Qp + (size_t)min(qrow, p.Sq - 1) * p.q_rs + d_w;
#pragma unroll
    for (int s = 0; s < NQ; ++s) qf[s] = *reinterpret_cast<const u32x4_t*>(qr + (BF ? 16 * s + 8 * h : 8 * s + 4 * h));
  }

  u32x4_t rk[BF ? PT : 1], rv[BF ? PT : 1];              // bf16: register prefetch of the next tile
  auto issue = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = t + 256 * i, row = c / DC, ch = c - row * DC;
      const size_t grow = (size_t)min(kv0 + row, p.Skv - 1);
      rk[i] = *reinterpret_cast<const u32x4_t*>(Kp + grow * p.k_rs + ch * EPC);
      rv[i] = *reinterpret_cast<const u32x4_t*>(Vp + grow * p.v_rs + ch * EPC);
    }
  };
  auto commit = [&](int stage) {
    char* ks = smem + stage * STAGE;
    char* vs = ks + BKV * KS;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = t + 256 * i, row = c / DC, ch = c - row * DC;
      *reinterpret_cast<u32x4_t*>(ks + row * KS + ch * 16) = rk[i];
      *reinterpret_cast<u32x4_t*>(vs + row * VS + ch * 16) = rv[i];
    }
  };
  auto load_sync = [&](int kv0) {                        // f32: straight through registers, 8 chunks at a time
    char* ks = smem;
    char* vs = ks + BKV * KS;
    for (int i0 = 0; i0 < PT; i0 += 8) {
      u32x4_t a[8], c8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = t + 256 * (i0 + i), row = c / DC, ch = c - row * DC;
        const size_t grow = (size_t)min(kv0 + row, p.Skv - 1);
        a[i] = *reinterpret_cast<const u32x4_t*>(Kp + grow * p.k_rs + ch * EPC);
        c8[i] = *reinterpret_cast<const u32x4_t*>(Vp + grow * p.v_rs + ch * EPC);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = t + 256 * (i0 + i), row = c / DC, ch = c - row * DC;
        *reinterpret_cast<u32x4_t*>(ks + row * KS + ch * 16) = a[i];
        *reinterpret_cast<u32x4_t*>(vs + row * VS + ch * 16) = c8[i];
      }
    }
  };

  f32x16_t o[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;
  const int ntiles = (p.Skv + BKV - 1) / BKV;
  const float sc = p.scale_log2e;

  if constexpr (BF) { issue(0); commit(0); } else { load_sync(0); }
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int kv0 = kt * BKV;
    const char* ks = smem + (BF ? (kt & 1) : 0) * STAGE;
    const char* vs = ks + BKV * KS;
    if constexpr (BF) { if (kt + 1 < ntiles) issue(kv0 + BKV); }

    // partial scores over this wave's slice of the head dim
    f32x16_t s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    if constexpr (BF) {
#pragma unroll
      for (int st = 0; st < NQ; ++st) {
        bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + r * KS + (d_w + 16 * st + 8 * h) * 2);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8_t, qf[st]), s, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < NQ; ++g) {
        f32x4_t kf = *reinterpret_cast<const f32x4_t*>(ks + r * KS + (d_w + 8 * g + 4 * h) * 4);
        f32x4_t qv = __builtin_bit_cast(f32x4_t, qf[g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qv[e], s, 0, 0, 0);
      }
    }
    // exchange: every wave publishes its partial, then sums the four in wave order 0..3
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4_t*>(xs + wave * 4096 + g * 1024 + lane * 16) = (f32x4_t){s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4_t a = *reinterpret_cast<const f32x4_t*>(xs + 0 * 4096 + g * 1024 + lane * 16);
#pragma unroll
      for (int w = 1; w < 4; ++w) a += *reinterpret_cast<const f32x4_t*>(xs + w * 4096 + g * 1024 + lane * 16);
      s[4 * g] = a[0]; s[4 * g + 1] = a[1]; s[4 * g + 2] = a[2]; s[4 * g + 3] = a[3];
    }
    // online softmax in the log2 domain (register e of lane (r, h) = key (e&3) + 8*(e>>2) + 4h, query r)
    if (kv0 + BKV > p.Skv) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (kv0 + (e & 3) + 8 * (e >> 2) + 4 * h >= p.Skv) s[e] = -1.0e30f;
    }
    float mx = -1.0e30f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], sc, -m_new)); rs += s[e]; }
    rs += __shfl_xor(rs, 32, 64);
    l_run = __builtin_fmaf(l_run, alpha, rs);
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    }
    // O^T[slice] += V[:, slice]^T P^T
    if constexpr (BF) {
      const int g16 = lane >> 4, li = lane & 15;
      const char* vbase = vs + (4 * (g16 >> 1) + (li >> 2)) * VS + (d_w + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
      bf16x8_t pf[2];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        u32x4_t w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack_bf2(s[8 * st + 2 * j], s[8 * st + 2 * j + 1]);
        pf[st] = __builtin_bit_cast(bf16x8_t, w);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          s16x4_t lo = lds_read_tr16(vbase + (16 * st) * VS + i * 64);
          s16x4_t hi = lds_read_tr16(vbase + (16 * st + 8) * VS + i * 64);
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vv), pf[st], o[i], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = (e & 3) + 8 * (e >> 2) + 4 * h;
          float a = *reinterpret_cast<const float*>(vs + key * VS + (d_w + i * 32 + r) * 4);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[e], o[i], 0, 0, 0);
        }
      }
    }
    if constexpr (BF) {
      if (kt + 1 < ntiles) commit((kt + 1) & 1);
      __syncthreads();                                   // next tile visible; everyone is done with xs and this tile
    } else {
      __syncthreads();                                   // everyone is done with the (single) tile and xs
      if (kt + 1 < ntiles) { load_sync(kv0 + BKV); __syncthreads(); }
    }
  }

  if (qrow < p.Sq) {
    const float inv = 1.0f / l_run;
    T* op = (T*)p.O + (size_t)b * p.o_bs + (size_t)qrow * p.o_rs + (size_t)head * D + d_w;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * i + 8 * g + 4 * h;
        float v0 = o[i][4 * g] * inv, v1 = o[i][4 * g + 1] * inv, v2 = o[i][4 * g + 2] * inv, v3 = o[i][4 * g + 3] * inv;
        if constexpr (BF) {
          u32x2_t w = {pack_bf2(v0, v1), pack_bf2(v2, v3)};
          *reinterpret_cast<u32x2_t*>(op + d0) = w;
        } else {
          *reinterpret_cast<f32x4_t*>(op + d0) = (f32x4_t){v0, v1, v2, v3};
        }
      }
    }
  }
}

template <typename T>
static int launch_attn512(const AttnArgs& a, int B, hipStream_t st) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int KS = 512 * (int)sizeof(T) + 16, VS = BF ? 512 * 2 + 64 : 512 * 4;
  constexpr int lds = (BF ? 2 : 1) * 32 * (KS + VS) + 4 * 4096;
  static_assert(lds <= 160 * 1024, "attn512: LDS budget");
  static std::atomic<unsigned long long> lds_ok{0};
  if (sd_lds_opt_in(lds_ok, (const void*)attn512_kernel<T>, lds) != hipSuccess)
    return sdamd_set_error(2, "attention(d=512): cannot reserve %d bytes of LDS", lds);
  dim3 grid((a.Sq + 31) / 32, B * a.H);
  attn512_kernel<T><<<grid, 256, lds, st>>>(a);
  SD_LAUNCH_CHECK("attention512");
  return 0;
}

// q/k/v/out: [B][S][row stride] with head h at columns [h*d, (h+1)*d); strides in elements.
extern "C" int sdamd_attention(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq, int Skv, int d,
                               long long q_bs, int q_rs, long long k_bs, int k_rs, long long v_bs, int v_rs,
                               long long o_bs, int o_rs, int causal, int dtype, void* stream) {
  SD_CHECK(B > 0 && H > 0 && Sq > 0 && Skv > 0, "attention: empty problem B=%d H=%d Sq=%d Skv=%d", B, H, Sq, Skv);
  SD_CHECK(d == 40 || d == 64 || d == 80 || d == 160 || d == 512, "attention: head dim %d not built (40, 64, 80, 160, 512)", d);
  SD_CHECK(d != 512 || !causal, "attention: the 512-wide head (VAE) has no causal form");
  SD_CHECK(dtype == SDAMD_BF16 || dtype == SDAMD_F32, "attention: bad dtype");
  const int esz = dtype == SDAMD_BF16 ? 2 : 4;
  SD_CHECK(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0,
           "attention: pointers must be 16-byte aligned");
  SD_CHECK((q_rs * esz) % 16 == 0 && (k_rs * esz) % 16 == 0 && (v_rs * esz) % 16 == 0 && (o_rs * esz) % 16 == 0 &&
               (q_bs * esz) % 16 == 0 && (k_bs * esz) % 16 == 0 && (v_bs * esz) % 16 == 0 && (o_bs * esz) % 16 == 0,
           "attention: strides must keep 16-byte alignment");
  SD_CHECK(!causal || Sq == Skv, "attention: causal needs Sq == Skv");
  AttnArgs a;
  a.Q = q; a.K = k; a.V = v; a.O = out; a.H = H; a.Sq = Sq; a.Skv = Skv;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.q_rs = q_rs; a.k_rs = k_rs; a.v_rs = v_rs; a.o_rs = o_rs;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)d);
  a.causal = causal;
  { extern int g_gemm_debug; a.debug = g_gemm_debug; }
  hipStream_t st = (hipStream_t)stream;
  if (d == 512) return dtype == SDAMD_BF16 ? launch_attn512<bf16_t>(a, B, st) : launch_attn512<float>(a, B, st);
  if (d == 40 && dtype == SDAMD_BF16 && a.debug == 17) return launch_attn<bf16_t, 40, true>(a, B, st);
  // Experiment, off by default (debug word 19): the software-pipelined one-wave-per-SIMD d = 40 kernel of attention2.hip --
  // bit-equal to the kernel below and SLOWER (B16 S4096: 589 vs 478 us; DESIGN.md section 4 has its ablations).
  if (d == 40 && dtype == SDAMD_BF16 && !causal && Skv % 64 == 0 && Skv >= 8 * 64 && a.debug >= 19 && a.debug <= 24) return launch_attn40p(a, B, st);
  // Short key sequences (the 77-token cross attention): K / V staged once per workgroup, attention3.hip.  Debug word 26
  // keeps the generic kernel (A/B, and the tests compare the two).
  if (dtype == SDAMD_BF16 && !causal && Skv <= 96 && (d == 40 || d == 80 || d == 160) && a.debug != 26) return launch_xattn(a, B, d, st);
  { extern int g_f32_emu;
    if (dtype == SDAMD_F32 && g_f32_emu) {   // sdamd_set_f32_mode: f32 attention on the bf16 matrix pipe (the 512-wide VAE head keeps the f32 MFMA)
      if (d == 40) return launch_attn_f32x3<40>(a, B, st);
      if (d == 64) return launch_attn_f32x3<64>(a, B, st);
      if (d == 80) return launch_attn_f32x3<80>(a, B, st);
      if (d == 160) return launch_attn_f32x3<160>(a, B, st);   // (256 VGPRs + 89 AGPRs, one wave per SIMD: its grids are small anyway)
    } }
#define ATT(TT) switch (d) { case 40: return launch_attn<TT, 40>(a, B, st); case 64: return launch_attn<TT, 64>(a, B, st); \
                             case 80: return launch_attn<TT, 80>(a, B, st); default: return launch_attn<TT, 160>(a, B, st); }
  if (dtype == SDAMD_BF16) ATT(bf16_t) else ATT(float)
#undef ATT
}
