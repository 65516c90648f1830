// The pipelined one-wave-per-SIMD d = 40 attention kernel: a translation unit of its own so that its MFMA accumulators
// can live in AGPRs (attention.hip is built with -amdgpu-mfma-vgpr-form).
#include "attention_body.h"

// ---------------------------------------------------------------------------------------------
// d = 40 self-attention, ONE wave per SIMD, hand-scheduled (round 3).  Same algorithm, data layout and arithmetic per
// element as attn_kernel<bf16, 40>'s fast path (S^T = K Q^T with the softmax reference riding in the product, exp2 of
// the MFMA result, ones column in V for the denominator; reference fixed by the first 64 keys, whole-workgroup re-run
// through attn_body<.., TRACK = true> when a denominator overflows) -- but the 64-key tile loop is software-pipelined
// inside the wave instead of left to three co-resident waves: the loop is vector-issue-bound at d = 40 (per 32 x 32
// score tile 16 v_exp + 8 v_cvt_pk + 11 LDS reads against 7 MFMAs), and three waves per SIMD measured 461 cycles per
// tile where the issue costs add up to ~265.  Iteration t of the loop:
//   QK^T of tile t+1 (6 MFMAs, K fragments read an iteration ago)  ||  exp2 / pack of sub-tile 0 of tile t, its V reads
//   P V of sub-tile 0 (4 MFMAs)                                     ||  exp2 / pack of sub-tile 1, its V reads
//   P V of sub-tile 1 (4 MFMAs)                                     ||  K fragment reads of tile t+2, commit of tile t+3
//                                                                       to LDS, global loads of tile t+4
// as 14 slots of one MFMA + its fillers, fenced by sched_barriers; one workgroup barrier per tile; 4 LDS stages.
// Takes: bf16, d = 40, no mask (Skv a multiple of 64, >= 8 tiles, not causal).
// ---------------------------------------------------------------------------------------------
// ABL: timing ablations (wrong results; experiments/attn_ab.py): 1 no barrier, 2 no LDS fragment reads, 3 no exp2 / pack, 4 no MFMA
template <int ABL>
__global__ __launch_bounds__(256) void attn40p_kernel(const AttnArgs p) {
  typedef AttnCfg<bf16_t, 40> Cfg;
  constexpr int D = 40, KS = Cfg::KS, VS = Cfg::VS, BKV = 64, STAGE = Cfg::STAGE, NST = 4;
  constexpr int DC = 5, KC = 6, VC = 8, NQ = 3, NB = 2;
  constexpr int L_BLK = D / 32, L_REG = ((D % 32) & 3) + 4 * ((D % 32) >> 3), L_HALF = ((D % 32) >> 2) & 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NST stages of [K tile | V tile] + the redo flag

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y / p.H, head = blockIdx.y % p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = q0 + r;
  const bf16_t* __restrict__ Qp = (const bf16_t*)p.Q + (size_t)b * p.q_bs + (size_t)head * D;
  const bf16_t* __restrict__ Kp = (const bf16_t*)p.K + (size_t)b * p.k_bs + (size_t)head * D;
  const bf16_t* __restrict__ Vp = (const bf16_t*)p.V + (size_t)b * p.v_bs + (size_t)head * D;

  u32x4_t qf[NQ];
  {
    const bf16_t* qr = Qp + (size_t)min(qrow, p.Sq - 1) * p.q_rs;
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
      const int d0 = 16 * s + 8 * h;
      u32x4_t z = {0u, 0u, 0u, 0u};
      if (d0 < D) z = *reinterpret_cast<const u32x4_t*>(qr + d0);
      qf[s] = z;
    }
#pragma unroll
    for (int s = 0; s < NQ; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(qf[s][j]) : "v"(qf[s][j]));   // (see attn_body)
#pragma unroll
    for (int s = 0; s < NQ; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = __builtin_bit_cast(float, qf[s][j] << 16) * p.scale_log2e;
        const float hi = __builtin_bit_cast(float, qf[s][j] & 0xffff0000u) * p.scale_log2e;
        qf[s][j] = pack_bf2(lo, hi);
      }
  }

  // K / V tile through registers, as attn_body (only the D real columns travel; pads are constant, written once)
  constexpr int KCH2 = BKV * DC, PT = (KCH2 + 255) / 256;
  // Four register sets: a tile is requested four iterations (~1.5 us) before it is committed to LDS -- one workgroup per
  // CU has no other wave to hide an L2 / Infinity-Cache round trip behind.  Thread t moves chunk t & 3 of row t >> 2 and,
  // when t & 3 == 0, the row's fifth chunk: the same work for every wave.
  constexpr int NPF = 4;
  u32x4_t rk[NPF][PT], rv[NPF][PT];
  // (per-thread row pointers once; a tile is a wave-uniform offset: Skv is a multiple of the tile, nothing to clamp)
  const bf16_t* kp0 = Kp + (size_t)(t >> 2) * p.k_rs + (t & 3) * 8;
  const bf16_t* vp0 = Vp + (size_t)(t >> 2) * p.v_rs + (t & 3) * 8;
  const bool fifth = (t & 3) == 0;
  auto issue = [&](int set, int kv0) {
    const long long ko = (long long)kv0 * p.k_rs, vo = (long long)kv0 * p.v_rs;
    rk[set][0] = *reinterpret_cast<const u32x4_t*>(kp0 + ko);
    rv[set][0] = *reinterpret_cast<const u32x4_t*>(vp0 + vo);
    if (fifth) {
      rk[set][1] = *reinterpret_cast<const u32x4_t*>(kp0 + ko + 32);
      rv[set][1] = *reinterpret_cast<const u32x4_t*>(vp0 + vo + 32);
    }
  };
  const int kst = (t >> 2) * KS + (t & 3) * 16, vst = BKV * KS + (t >> 2) * VS + (t & 3) * 16;
  auto commit = [&](int set, int stage) {
    char* st = smem + stage * STAGE;
    *reinterpret_cast<u32x4_t*>(st + kst) = rk[set][0];
    *reinterpret_cast<u32x4_t*>(st + vst) = rv[set][0];
    if (fifth) {
      *reinterpret_cast<u32x4_t*>(st + kst + 64) = rk[set][1];
      *reinterpret_cast<u32x4_t*>(st + vst + 64) = rv[set][1];
    }
  };
  for (int c = t; c < NST * BKV * (KC - DC + VC - DC); c += 256) {   // constant pad chunks of every stage
    constexpr int NPAD = KC - DC + VC - DC;
    const int srow = c / NPAD, pc = c - srow * NPAD;
    char* base = smem + (srow / BKV) * STAGE;
    const int row = srow % BKV;
    u32x4_t z = {0u, 0u, 0u, 0u};
    if (pc < KC - DC) {
      if (pc == 0) z[0] = 0x00003F80u;                                // K[:, D] = 1
      *reinterpret_cast<u32x4_t*>(base + row * KS + (DC + pc) * 16) = z;
    } else {
      const int ch = DC + pc - (KC - DC);
      if (ch == DC) z[0] = 0x00003F80u;                               // V[:, D] = 1
      *reinterpret_cast<u32x4_t*>(base + BKV * KS + row * VS + ch * 16) = z;
    }
  }
  const int ntiles = p.Skv / BKV;
  int* redo = reinterpret_cast<int*>(smem + NST * STAGE);
  if (t == 0) *redo = 0;

  f32x16_t o[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = 0.f, m_qk = 0.f;

  // fragment addressing (attn_body: qk / pv)
  const int koff = r * KS + 8 * h * 2;                                  // + sub*32*KS + 32*st
  const int g16 = lane >> 4, li = lane & 15;
  const int voff = BKV * KS + (4 * (g16 >> 1) + (li >> 2)) * VS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;   // + (sub*32 + 16*st [+8]) * VS + i*64
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;

  f32x16_t S[2][2];                                   // [parity][sub-tile]: scores of the current and of the next tile
  bf16x8_t kf[2][NQ];                                 // K fragments of the next QK^T
  auto rd_k = [&](const char* st, int sub, int q) { kf[sub][q] = *reinterpret_cast<const bf16x8_t*>(st + koff + sub * 32 * KS + 32 * q); };
  auto rd_v = [&](const char* st, int sub, int i, int q) {
    const s16x4_t lo = lds_read_tr16(st + voff + (sub * 32 + 16 * q) * VS + i * 64);
    const s16x4_t hi = lds_read_tr16(st + voff + (sub * 32 + 16 * q + 8) * VS + i * 64);
    const s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, vv);
  };

  // ---- prologue: tiles 0..3 staged, loads of tile 4 in flight; tile 0 sets every query's reference (the tracked softmax of
  // attn_body), then S(1) and the K fragments of tile 2
  issue(0, 0); issue(1, BKV); issue(2, 2 * BKV); issue(3, 3 * BKV);
  commit(0, 0); commit(1, 1); commit(2, 2); commit(3, 3);
  issue(1, 4 * BKV); issue(2, 5 * BKV); issue(3, 6 * BKV); issue(0, 7 * BKV);   // iteration t commits tile t+3 out of set t & 3 and refills it with tile t+7
  __syncthreads();
  {
    const char* st0 = smem;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) rd_k(st0, sub, q);
#pragma unroll
      for (int e = 0; e < 16; ++e) S[0][sub][e] = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) S[0][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][q], __builtin_bit_cast(bf16x8_t, qf[q]), S[0][sub], 0, 0, 0);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      float mx = -1.0e30f;
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[0][sub][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float off = m_run - m_qk, rel = mx - off;
      const bool first = sub == 0;
      if (first || !__all(off == 0.f && rel <= RESCALE_THR)) {
        const float m_new = (first || rel > RESCALE_THR) ? bf2f(f2bf(m_run + rel)) : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float shift = m_new - m_qk;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[0][sub][e] -= shift;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) S[0][sub][e] = __builtin_amdgcn_exp2f(S[0][sub][e]);
      bf16x8_t pf[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        u32x4_t w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack_bf2(S[0][sub][8 * q + 2 * j], S[0][sub][8 * q + 2 * j + 1]);
        pf[q] = __builtin_bit_cast(bf16x8_t, w);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_v(st0, sub, i, q), pf[q], o[i], 0, 0, 0);
    }
    if (!__all(m_run == m_qk)) {   // bake the reference into Q[:, D] (lanes of half 1 hold column D)
      const unsigned nm = __builtin_bit_cast(unsigned, -m_run) >> 16;
      if (h == 1) qf[NQ - 1][0] = (qf[NQ - 1][0] & 0xffff0000u) | nm;
      m_qk = m_run;
    }
    const char* st1 = smem + STAGE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) rd_k(st1, sub, q);
#pragma unroll
      for (int e = 0; e < 16; ++e) S[1][sub][e] = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) S[1][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][q], __builtin_bit_cast(bf16x8_t, qf[q]), S[1][sub], 0, 0, 0);
    }
    const char* st2 = smem + 2 * STAGE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int q = 0; q < NQ; ++q) rd_k(st2, sub, q);
  }
  __syncthreads();   // every wave is done with tile 0: its stage takes tile 4

  // ---- the pipelined loop.  PAR = parity of the tile whose scores are exponentiated (S[PAR]); S[PAR ^ 1] receives tile t+1.
  // FULL = every look-ahead exists (t + 4 < ntiles is not needed: the loads clamp their rows); the last three tiles run
  // the guarded form.
  bf16x8_t vf[2][NB][2];
  u32x4_t pw[2][2];
  auto iter = [&](auto SETC, auto FULLC, int tt) {
    constexpr int SET = decltype(SETC)::value, P = SET & 1, N = P ^ 1;
    constexpr bool FULL = decltype(FULLC)::value;
    const char* sv = smem + (tt & 3) * STAGE;              // V of tile t
    const char* sk = smem + ((tt + 2) & 3) * STAGE;        // K of tile t+2
    const bool has1 = FULL || tt + 1 < ntiles, has2 = FULL || tt + 2 < ntiles, has3 = FULL || tt + 3 < ntiles;
    auto E = [&](int sub, int e0, int n) {
      if (ABL == 3) return;
#pragma unroll
      for (int e = e0; e < e0 + n; ++e) S[P][sub][e] = __builtin_amdgcn_exp2f(S[P][sub][e]);
    };
    auto C = [&](int sub, int q, int j0, int n) {
      if (ABL == 3) return;
#pragma unroll
      for (int j = j0; j < j0 + n; ++j) pw[sub][q][j] = pack_bf2(S[P][sub][8 * q + 2 * j], S[P][sub][8 * q + 2 * j + 1]);
    };
    auto MQ = [&](int sub, int q) {
      if (!has1 || ABL == 4) return;
      if (q == 0) {
        const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        S[N][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][q], __builtin_bit_cast(bf16x8_t, qf[q]), z, 0, 0, 0);
      } else {
        S[N][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][q], __builtin_bit_cast(bf16x8_t, qf[q]), S[N][sub], 0, 0, 0);
      }
    };
    auto MP = [&](int sub, int i, int q) { if (ABL == 4) return; o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[sub][i][q], __builtin_bit_cast(bf16x8_t, pw[sub][q]), o[i], 0, 0, 0); };
    auto RV = [&](int sub, int i, int q) { if (ABL != 2) vf[sub][i][q] = rd_v(sv, sub, i, q); };
    auto RK = [&](int sub, int q) { if (has2 && ABL != 2) rd_k(sk, sub, q); };
#define SLOT __builtin_amdgcn_sched_barrier(0)
    // 14 slots = one MFMA + its fillers each.  The 32 exp2 are spread 3,3,3,3,2,2 | 3,3,3,3,2,2 (the loop is bound by
    // vector issue: an even spread keeps every slot above the MFMA's 32 cycles); a K fragment register is re-read right
    // after the MFMA that used it; V fragments are read >= 4 slots ahead of their MFMA; the LDS commit and the global
    // loads sit in slots 11 / 12 so that the closing barrier finds nothing of this wave in flight.
    MQ(0, 0); E(0, 0, 3); RV(0, 0, 0); SLOT;
    MQ(1, 0); E(0, 3, 3); RV(0, 1, 0); RK(0, 0); SLOT;
    MQ(0, 1); E(0, 6, 3); RV(0, 0, 1); RK(1, 0); SLOT;
    MQ(1, 1); E(0, 9, 3); RV(0, 1, 1); RK(0, 1); C(0, 0, 0, 2); SLOT;
    MQ(0, 2); E(0, 12, 2); C(0, 0, 2, 2); RK(1, 1); SLOT;
    MQ(1, 2); E(0, 14, 2); C(0, 1, 0, 2); RK(0, 2); SLOT;
    C(0, 1, 2, 2); MP(0, 0, 0); E(1, 0, 3); RV(1, 0, 0); RK(1, 2); SLOT;
    MP(0, 1, 0); E(1, 3, 3); RV(1, 1, 0); SLOT;
    MP(0, 0, 1); E(1, 6, 3); RV(1, 0, 1); SLOT;
    MP(0, 1, 1); E(1, 9, 3); RV(1, 1, 1); C(1, 0, 0, 2); SLOT;
    C(1, 0, 2, 2); MP(1, 0, 0); E(1, 12, 2); if (has3 && ABL != 5) commit(SET, (tt + 3) & 3); SLOT;
    MP(1, 1, 0); E(1, 14, 2); if (ABL != 5) issue(SET, min(tt + 7, ntiles - 1) * BKV); SLOT;   // (past the end: the last tile again, never committed)
    C(1, 1, 0, 4); MP(1, 0, 1); SLOT;
    MP(1, 1, 1); SLOT;
#undef SLOT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes (tile t+3) and reads are done ...
    if (ABL != 1) __builtin_amdgcn_s_barrier();           // ... before any wave re-stages V(t)'s buffer or reads tile t+3
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>;
  int tt = 1;
  for (; tt + 6 < ntiles; tt += 4) {      // tiles tt .. tt + 3 (tt & 3 == 1), every commit in range
    iter(T1{}, std::true_type{}, tt);
    iter(T2{}, std::true_type{}, tt + 1);
    iter(T3{}, std::true_type{}, tt + 2);
    iter(T0{}, std::true_type{}, tt + 3);
  }
  for (; tt < ntiles; ++tt) {
    switch (tt & 3) {
      case 0: iter(T0{}, std::false_type{}, tt); break;
      case 1: iter(T1{}, std::false_type{}, tt); break;
      case 2: iter(T2{}, std::false_type{}, tt); break;
      default: iter(T3{}, std::false_type{}, tt); break;
    }
  }

  // overflow check of the fixed-reference path (attn_body): re-run the workgroup with the reference tracked per tile
  {
    const float l = __shfl(o[L_BLK][L_REG], r + 32 * L_HALF, 64);
    if (ABL == 0 && __any(!(l < 0x1p100f)) && lane == 0) *redo = 1;
    __syncthreads();
    if (*redo) {
      __syncthreads();
      attn_body<bf16_t, 40, true>(p);
      return;
    }
  }
  if (qrow < p.Sq) {
    const float l = __shfl(o[L_BLK][L_REG], r + 32 * L_HALF, 64);
    const float inv = 1.0f / l;
    bf16_t* op = (bf16_t*)p.O + (size_t)b * p.o_bs + (size_t)qrow * p.o_rs + (size_t)head * D;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * i + 8 * g + 4 * h;
        if (d0 < D) {
          const u32x2_t w = {pack_bf2(o[i][4 * g] * inv, o[i][4 * g + 1] * inv), pack_bf2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv)};
          *reinterpret_cast<u32x2_t*>(op + d0) = w;
        }
      }
  }
}

template <int ABL>
static int launch_attn40p_t(const AttnArgs& a, int B, hipStream_t st) {
  dim3 grid((a.Sq + 127) / 128, B * a.H);
  constexpr int lds = 4 * AttnCfg<bf16_t, 40>::STAGE + 16;
  static std::atomic<unsigned long long> lds_ok{0};
  if (sd_lds_opt_in(lds_ok, (const void*)attn40p_kernel<ABL>, lds) != hipSuccess)
    return sdamd_set_error(2, "attention: cannot reserve %d bytes of LDS", lds);
  attn40p_kernel<ABL><<<grid, 256, lds, st>>>(a);
  SD_LAUNCH_CHECK("attention40p");
  return 0;
}

int launch_attn40p(const AttnArgs& a, int B, hipStream_t st) {
#ifdef SDAMD_ATTN_ABLATE
  switch (a.debug) {
    case 20: return launch_attn40p_t<5>(a, B, st);   // no global loads / LDS commits
    case 21: return launch_attn40p_t<1>(a, B, st);
    case 22: return launch_attn40p_t<2>(a, B, st);
    case 23: return launch_attn40p_t<3>(a, B, st);
    case 24: return launch_attn40p_t<4>(a, B, st);
  }
#endif
  return launch_attn40p_t<0>(a, B, st);
}
