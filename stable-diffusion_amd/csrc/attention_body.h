#pragma once
// MFMA flash attention for gfx950 -- replaces the eager softmax(QK^T)V of attention.py:47-68,
// 112-127 and the Triton path (customtriton.py, flash_attention.py) of the reference.
// No S x S matrix is materialised: one workgroup = 4 waves = 128 query rows of one (batch, head);
// each wave owns 32 query rows and sweeps the keys in LDS-staged, double-buffered tiles (64 keys
// bf16 / 32 keys f32, next tile prefetched through registers) with an online softmax.  Scores are computed transposed, S^T = K Q^T (32x32 MFMA, key on the accumulator row,
// query on the lane), so a query's running max / sum are lane-local (one cross-half exchange)
// and the exponentiated tile is already the B operand of the O^T += V^T P^T MFMAs.
//   bf16: v_mfma_f32_32x32x16_bf16; V^T fragments come from row-major V in LDS through
//         ds_read_b64_tr_b16 (hardware transpose read).
//   f32 : v_mfma_f32_32x32x2_f32 (exact f32); one element per lane per MFMA, so V is read
//         row-major with ds_read_b32 and no transpose is needed.
// Head dims of SD-v1.5: 40 / 80 / 160 (UNet), 64 (CLIP).  The contraction dim of QK^T is padded
// with zeros to DK, the output dim of PV to DV (multiples of the MFMA tile).  KV lengths that are
// not a multiple of 32 (77-token cross attention / CLIP) are masked; `causal` masks key > query
// (mask before scale, attention.py:55-61 -- identical for -inf).
#include "common.h"
#include <type_traits>

struct AttnArgs {
  const void* Q; const void* K; const void* V; void* O;
  int H, Sq, Skv;
  long long q_bs, k_bs, v_bs, o_bs;  // batch strides, elements
  int q_rs, k_rs, v_rs, o_rs;        // token strides, elements; head h starts at column h*D
  float scale_log2e;  // log2(e) / sqrt(D)
  int debug;          // timing ablations (only in builds with -DSDAMD_ATTN_ABLATE): 1 no softmax, 2 no PV, 3 no QK, 4 no staging, 5 no barrier
  int causal;
};

__device__ __forceinline__ s16x4_t lds_read_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}

template <typename T, int D> struct AttnCfg;
template <int D> struct AttnCfg<bf16_t, D> {
  static constexpr int DK = (D + 15) / 16 * 16;
  static constexpr int DV = (D + 31) / 32 * 32;
  static constexpr int KS = DK * 2 + 16;                             // K row stride (bytes): odd multiple of 16
  static constexpr int VS = (DV * 2) % 256 == 64 || (DV * 2) % 256 == 192 ? DV * 2 : DV * 2 + 64;  // 4 rows x 64 B tile a bank row
  static constexpr int BKV = 64;                                     // keys per staged tile
  static constexpr int STAGE = BKV * (KS + VS);
};
template <int D> struct AttnCfg<float, D> {
  static constexpr int DK = D;  // multiple of 8
  static constexpr int DV = (D + 31) / 32 * 32;
  static constexpr int KS = DK * 4 + 16;
  static constexpr int VS = DV * 4;
  static constexpr int BKV = 32;   // f32 tiles are twice as wide in bytes: 2 stages of 64 keys would not fit at d=160
  static constexpr int STAGE = BKV * (KS + VS);
};

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
constexpr float RESCALE_THR = 8.0f;

// (forcing 4 waves/SIMD through __launch_bounds__ spills 8-27 registers and measured slower)
// Runtime ablation switches split the K/V loop into a dozen basic blocks (no scheduling across them: every
// fragment read was followed by its own wait): they only exist in -DSDAMD_ATTN_ABLATE builds.
#ifdef SDAMD_ATTN_ABLATE
#define ATTN_DBG(x) (p.debug == (x))
#else
#define ATTN_DBG(x) false
#endif

// OPT (d = 40 only, where the reference rides in the QK product): after the first staged tile has set each query's
// reference, the later tiles do not look for a row maximum at all -- scores come out of the MFMA relative to that
// reference and go straight to exp2 (no 16 v_max, cross-half exchange, vote and branch per sub-tile).  That is exact
// as long as no row outgrows its first-tile reference by ~2^100; the denominator (accumulated by the P V product in
// the ones column) tells: if any query of the workgroup ends with l >= 2^100, inf or NaN, the whole workgroup runs its
// loop again with the reference tracked per tile (the path above).  Softmax is invariant to the reference, so both
// paths compute the same function; they differ in the rounding of P to bf16 only.
// TRACK = true compiles the fast path out (experiments/attn_ab.py and the tests use it as the comparison).
template <typename T, int D, bool TRACK = false>
__device__ __forceinline__ void attn_body(const AttnArgs& p) {
  typedef AttnCfg<T, D> Cfg;
  constexpr int DK = Cfg::DK, DV = Cfg::DV, KS = Cfg::KS, VS = Cfg::VS, BKV = Cfg::BKV;
  constexpr int EPC = 16 / sizeof(T);
  constexpr int KC = DK / EPC, VC = DV / EPC, DC = D / EPC;  // 16-byte chunks per row
  constexpr int NB = DV / 32;                                // 32-wide output blocks
  constexpr int NSUB = BKV / 32;                             // 32-key sub-tiles per staged tile
  constexpr bool BF = sizeof(T) == 2;
  // When the head dim leaves a spare padded V column (d = 40, 80), that column is set to one, so the
  // P V product itself accumulates the softmax denominator (from the same rounded P as the
  // numerator) and the online rescale of O covers it: no per-score adds, no cross-half exchange.
  constexpr bool ONES = D < DV;
  // When the head dim also leaves a spare padded K column (d = 40: DK = 48), the softmax reference rides in the QK^T
  // product: Q is pre-scaled by scale*log2(e) once, K[:, D] = 1 and Q[:, D] = -m (the query's current reference, kept
  // bf16-exact), so the MFMA returns scale*s - m and exp2 applies to it directly -- no per-score FMA (16 v_pk_fma_f32
  // of the 131 VALU instructions per 64-key tile in a VALU-issue-bound loop).  The reference still only moves when a
  // row outgrows it by 2^RESCALE_THR; tiles where it moves subtract the difference explicitly (rare path).
  constexpr bool AUG = BF && DK > D && ONES;
  constexpr bool OPT = AUG && !TRACK;
  constexpr int L_BLK = D / 32, L_REG = ((D % 32) & 3) + 4 * ((D % 32) >> 3), L_HALF = ((D % 32) >> 2) & 1;
  constexpr int NQ = BF ? DK / 16 : D / 8;                   // Q fragments (4 VGPR each)
  constexpr int KCH = BKV * KC, VCH = BKV * VC;              // chunks per staged tile
  constexpr int KPT = (KCH + 255) / 256, VPT = (VCH + 255) / 256;
  constexpr int STAGE = Cfg::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages of [K tile | V tile]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y / p.H, head = blockIdx.y % p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = q0 + r;
  const T* __restrict__ Qp = (const T*)p.Q + (size_t)b * p.q_bs + (size_t)head * D;
  const T* __restrict__ Kp = (const T*)p.K + (size_t)b * p.k_bs + (size_t)head * D;
  const T* __restrict__ Vp = (const T*)p.V + (size_t)b * p.v_bs + (size_t)head * D;

  // Q fragments (B operand): lane (r, h) holds Q[qrow][...] for its k slots
  u32x4_t qf[NQ];
  {
    const T* qr = Qp + (size_t)min(qrow, p.Sq - 1) * p.q_rs;
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
      const int d0 = BF ? 16 * s + 8 * h : 8 * s + 4 * h;
      u32x4_t z = {0u, 0u, 0u, 0u};
      if (d0 < D) z = *reinterpret_cast<const u32x4_t*>(qr + d0);
      qf[s] = z;
    }
    // Launder the fragments through a VALU move: otherwise hipcc's waitcnt pass carries "Q load may
    // still be outstanding" around the loop back edge and puts s_waitcnt vmcnt(0) in front of the
    // first MFMAs of every K/V tile -- which also waits for the K/V prefetch issued just before.
#pragma unroll
    for (int s = 0; s < NQ; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(qf[s][j]) : "v"(qf[s][j]));
    if constexpr (AUG) {   // Q *= scale*log2(e), rounded to bf16 once per query element
#pragma unroll
      for (int s = 0; s < NQ; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = __builtin_bit_cast(float, qf[s][j] << 16) * p.scale_log2e;
          const float hi = __builtin_bit_cast(float, qf[s][j] & 0xffff0000u) * p.scale_log2e;
          qf[s][j] = pack_bf2(lo, hi);
        }
    }
  }

  // Register-staged K/V tile: the loads of tile t+1 are issued before tile t is computed and written
  // to the other LDS stage afterwards.  Only the D real columns travel; the pad columns of both
  // stages (zeros, and the ones column of V) are constant and written once, below.  Rows past Skv
  // are clamped to the last valid row: their scores are masked, so they only have to be finite.
  constexpr int KCH2 = BKV * DC;                       // real 16-byte chunks per K (or V) tile
  constexpr int PT = (KCH2 + 255) / 256;
  u32x4_t rk[PT], rv[PT];
  auto issue = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = t + 256 * i;
      if (KCH2 % 256 == 0 || c < KCH2) {
        const int row = c / DC, ch = c - row * DC;
        const size_t grow = (size_t)min(kv0 + row, p.Skv - 1);
        rk[i] = *reinterpret_cast<const u32x4_t*>(Kp + grow * p.k_rs + ch * EPC);
        rv[i] = *reinterpret_cast<const u32x4_t*>(Vp + grow * p.v_rs + ch * EPC);
      }
    }
  };
  auto commit = [&](int stage) {
    char* ks = smem + stage * STAGE;
    char* vs = ks + BKV * KS;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = t + 256 * i;
      if (KCH2 % 256 == 0 || c < KCH2) {
        const int row = c / DC, ch = c - row * DC;
        *reinterpret_cast<u32x4_t*>(ks + row * KS + ch * 16) = rk[i];
        *reinterpret_cast<u32x4_t*>(vs + row * VS + ch * 16) = rv[i];
      }
    }
  };
  for (int c = t; c < 2 * BKV * (KC - DC + VC - DC); c += 256) {   // constant pad chunks of both stages
    constexpr int NPAD = KC - DC + VC - DC > 0 ? KC - DC + VC - DC : 1;   // (no pads at all: the loop does not run)
    const int srow = c / NPAD, pc = c - srow * NPAD;               // srow = stage * BKV + row
    char* base = smem + (srow / BKV) * STAGE;
    const int row = srow % BKV;
    u32x4_t z = {0u, 0u, 0u, 0u};
    if (pc < KC - DC) {
      if (AUG && pc == 0) z[0] = 0x00003F80u;                        // K[:, D] = 1: carries Q[:, D] = -m into every score
      *reinterpret_cast<u32x4_t*>(base + row * KS + (DC + pc) * 16) = z;
    } else {
      const int ch = DC + pc - (KC - DC);
      if (ONES && ch == DC) z[0] = BF ? 0x00003F80u : 0x3F800000u;   // V[:, D] = 1: O^T row D accumulates the row sums
      *reinterpret_cast<u32x4_t*>(base + BKV * KS + row * VS + ch * 16) = z;
    }
  }

  // causal: this block needs keys 0 .. q0_block+127 only
  int kv_end = p.Skv;
  if (p.causal) kv_end = min(p.Skv, (int)blockIdx.x * 128 + 128);
  const int ntiles = (kv_end + BKV - 1) / BKV;
  const float sc = p.scale_log2e;
  int* redo = reinterpret_cast<int*>(smem + 2 * STAGE);   // OPT: "some row overflowed" flag (the launch adds 16 bytes)
  if (OPT && t == 0) *redo = 0;

  f32x16_t o[NB];
  float m_run, l_run, m_qk;
  bool safe = !OPT;                                        // track the reference in every tile (wave-uniform)
  for (;;) {
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  m_run = AUG ? 0.f : -1.0e30f, l_run = 0.f;  // running max (scaled, log2 domain) and denominator
  m_qk = 0.f;                                   // AUG: the reference currently baked into Q[:, D]

  issue(0);
  commit(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int kv0 = kt * BKV;
    const char* ks = smem + (kt & 1) * STAGE;
    const char* vs = ks + BKV * KS;
    if (kt + 1 < ntiles && !ATTN_DBG(4) && !ATTN_DBG(5)) issue(kv0 + BKV);

    // Per 32-key sub-tile: S^T = K Q^T (rows = keys in the accumulator registers, cols = queries on
    // the lanes) -> online softmax (log2 domain) -> O^T += V^T P^T.  Program order is
    //   QK(0) QK(1) softmax(0) PV(0) softmax(1) PV(1)
    // so that within one wave the VALU work of a sub-tile is issued while the MFMAs of its neighbour
    // are still executing (a wave issues in order; MFMA runs asynchronously to VALU).
    f32x16_t s[NSUB];
    auto qk = [&](int sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
      if constexpr (BF) {
#pragma unroll
        for (int st = 0; st < NQ; ++st) {
          bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + (sub * 32 + r) * KS + (16 * st + 8 * h) * 2);
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8_t, qf[st]), s[sub], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
          f32x4_t kf = *reinterpret_cast<const f32x4_t*>(ks + (sub * 32 + r) * KS + (8 * g + 4 * h) * 4);
          f32x4_t qv = __builtin_bit_cast(f32x4_t, qf[g]);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qv[e], s[sub], 0, 0, 0);
        }
      }
    };
    const bool need_mask = kv0 + BKV > p.Skv || p.causal;   // boundary / causal tiles only (wave-uniform)
    auto softmax = [&](int sub) {
      if (need_mask) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kv0 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= p.Skv || (p.causal && key > qrow)) s[sub][e] = -1.0e30f;
        }
      }
      if constexpr (OPT) {
        if (!safe && kt > 0) {   // reference fixed by tile 0 and baked into Q: the MFMA result is the exponent
#pragma unroll
          for (int e = 0; e < 16; ++e) s[sub][e] = __builtin_amdgcn_exp2f(s[sub][e]);
          return;
        }
      }
      float mx = -1.0e30f;
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if constexpr (AUG) {
        // s = scale*score - m_qk already.  The first sub-tile of the row sets the reference outright (it starts at 0,
        // which may be far above every score); afterwards it moves only when outgrown by RESCALE_THR.
        const float off = m_run - m_qk;                              // what the baked-in reference is short of the current one
        const float rel = mx - off;                                  // row maximum relative to the current reference
        const bool first = kt == 0 && sub == 0;
        if (first || !__all(off == 0.f && rel <= RESCALE_THR)) {     // rare: some row of this wave moves (or has moved) its reference
          const float m_new = (first || rel > RESCALE_THR) ? bf2f(f2bf(m_run + rel)) : m_run;
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          const float shift = m_new - m_qk;
#pragma unroll
          for (int e = 0; e < 16; ++e) s[sub][e] -= shift;
          m_run = m_new;
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = __builtin_amdgcn_exp2f(s[sub][e]);
        return;
      }
      // Lazy running max: the reference point m only moves when a row's maximum outgrows it by more
      // than RESCALE_THR (log2 units), so P stays <= 2^THR and the O rescale below is rare.  Exact in
      // exact arithmetic (numerator and denominator carry the same factor); P is a float either way.
      const float mxs = mx * sc;
      const float m_new = mxs > m_run + RESCALE_THR ? mxs : m_run;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rs = 0.f;
      const f32x2_t sc2 = {sc, sc}, nm2 = {-m_new, -m_new};
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        // v_pk_fma_f32 (measured: scalar v_fma_f32 + -fno-slp-vectorize is 2.5 % slower here)
        const f32x2_t a = (f32x2_t){s[sub][e], s[sub][e + 1]} * sc2 + nm2;
        const float p0 = __builtin_amdgcn_exp2f(a[0]), p1 = __builtin_amdgcn_exp2f(a[1]);
        s[sub][e] = p0;
        s[sub][e + 1] = p1;
        if (!ONES) rs += p0 + p1;
      }
      if (!ONES) {
        rs += __shfl_xor(rs, 32, 64);
        l_run = fmaf(l_run, alpha, rs);
      }
      m_run = m_new;
      if (!__all(alpha == 1.0f)) {  // exact: skip the O rescale when no row of this wave moved its reference
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
      }
    };
    auto pv = [&](int sub) {   // O^T += V^T P^T
      if constexpr (BF) {
        // transpose read: lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of a
        // 4-key x 16-column block; element j of the A fragment is key 16*st + 8*(j>>2) + 4h + (j&3)
        const int g16 = lane >> 4, li = lane & 15;
        const char* vbase = vs + (4 * (g16 >> 1) + (li >> 2)) * VS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;
        bf16x8_t pf[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          u32x4_t w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pack_bf2(s[sub][8 * st + 2 * j], s[sub][8 * st + 2 * j + 1]);
          pf[st] = __builtin_bit_cast(bf16x8_t, w);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            s16x4_t lo = lds_read_tr16(vbase + (sub * 32 + 16 * st) * VS + i * 64);
            s16x4_t hi = lds_read_tr16(vbase + (sub * 32 + 16 * st + 8) * VS + i * 64);
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
            const int key = sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            float a = *reinterpret_cast<const float*>(vs + key * VS + (i * 32 + r) * 4);
            o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[sub][e], o[i], 0, 0, 0);
          }
        }
      }
    };
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) if (!ATTN_DBG(3)) qk(sub);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (!ATTN_DBG(1)) softmax(sub);
      if (!ATTN_DBG(2)) pv(sub);
    }

    if constexpr (AUG) {
      if (!__all(m_run == m_qk)) {   // the reference moved in this tile: Q[:, D] = -m for the next one (lanes of half 1 hold column D)
        const unsigned nm = __builtin_bit_cast(unsigned, -m_run) >> 16;      // m_run is bf16-exact
        if (h == 1) qf[NQ - 1][0] = (qf[NQ - 1][0] & 0xffff0000u) | nm;
        m_qk = m_run;
      }
    }
    if (kt + 1 < ntiles && !ATTN_DBG(4) && !ATTN_DBG(5)) commit((kt + 1) & 1);
    if (!ATTN_DBG(5)) __syncthreads();
  }
  if constexpr (!OPT) {
    break;
  } else {
    if (safe) break;
    const float l = __shfl(o[L_BLK][L_REG], r + 32 * L_HALF, 64);
    if (__any(!(l < 0x1p100f)) && lane == 0) *redo = 1;
    __syncthreads();                 // also: every wave is done with the staged tiles before tile 0 is staged again
    if (*redo == 0) break;
    safe = true;
    if (h == 1) qf[NQ - 1][0] &= 0xffff0000u;   // Q[:, D] = 0 again
  }
  }

  // epilogue: O[q][d] = O^T[d][q] / l ; lane holds d = 32i + 8*(e>>2) + 4h + (e&3)
  if (qrow < p.Sq) {
    float l = l_run;
    if (ONES) l = __shfl(o[L_BLK][L_REG], r + 32 * L_HALF, 64);   // O^T[D][q] sits in the lane of half L_HALF
    const float inv = 1.0f / l;
    T* op = (T*)p.O + (size_t)b * p.o_bs + (size_t)qrow * p.o_rs + (size_t)head * D;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * i + 8 * g + 4 * h;
        if (d0 < D) {
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
}

