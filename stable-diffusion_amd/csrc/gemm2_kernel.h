// 8-wave bf16 MFMA GEMM / implicit-GEMM conv3x3 for gfx950 with LDS-DMA staging.
//
//   out[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] + residual[m][n] )       (same contract as
//   gemm_conv.hip; this file is the fast path for the bf16 problem sizes of the SD UNet / VAE)
//
// Workgroup = 512 threads = 8 waves (2 per SIMD) arranged WGM x WGN over a BM x BN output tile,
// one workgroup per CU.  K-tile = 64 bf16 = 128 bytes per row.  Both operands are K-contiguous
// and go global -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip, no ds_write):
// one wave-instruction lands 8 rows x 128 B = 1 KiB, lane l -> row l>>3, 16-byte chunk l&7.
// The LDS image is linear per instruction, so the bank-conflict swizzle (chunk ^= (row>>1)&7,
// conflict-free for the ds_read_b128 lane groups of the 16x16x32 MFMA operands) is applied to
// the SOURCE chunk each lane fetches and again on the fragment reads.
// Conv: the A row of output pixel m at K-tile (tap, c0) is 128 contiguous bytes of the NHWC
// input (or of a zero page when the tap falls into the padding), so the gather is just a
// per-lane source address; stride 1/2, symmetric or bottom/right padding and the fused nearest-2x
// upsample are address arithmetic.
// Pipeline: a ring of STAGES LDS buffers, one workgroup barrier per K-tile:
//     wait until tile t has landed (counted vmcnt: tile t+1 .. t+STAGES-2 stay in flight)
//     s_barrier          (tile t visible to every wave; every wave is done with tile t-1)
//     issue the DMA of tile t+STAGES-1 into the buffer tile t-1 just vacated
//     ds_read + MFMA on tile t
// The barrier is a raw s_barrier with hand-counted s_waitcnt (a __syncthreads() would drain vmcnt to
// 0 and serialise DMA and MFMA: measured, each alone runs at ~1.5 PF-equivalent, two stages with a
// full drain per tile only reach 1.0-1.2).  The K loop is unrolled STAGES times so that buffer
// addresses are compile-time.  MFMA "A" = weight rows, "B" = activation rows: a lane ends up with 4 consecutive
// output channels of one row (vector bias / residual / store), as in gemm_conv.hip.
#pragma once
#include "gemm_common.h"
#include <type_traits>
#include <utility>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// source for conv taps that fall into the zero padding: 16 bytes per lane for the gathered form; the halo-staged form
// adds its channel offset to every source alike, so the page spans the largest Cin it takes (HALO_MAX_CIN) + 64 bytes
#define HALO_MAX_CIN 4096
static __device__ __attribute__((aligned(256))) const unsigned int g_zero_page[HALO_MAX_CIN / 2 + 64] = {0};

__device__ __forceinline__ int swz2(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds, lds + 1024).
// ASM = true issues it from inline assembly.  hipcc models the builtin as a FLAT access that may touch LDS: from the first
// one on, its waitcnt pass regards the LGKM counter as out of order and every wait it inserts for an LDS read becomes
// lgkmcnt(0) -- "ds_read, wait for everything, MFMA" even when the fragment was requested long ago and younger reads are
// in flight.  Hidden in assembly, the DMA leaves the compiler's LDS bookkeeping alone (counted lgkmcnt waits); its
// completion is the kernel's business either way (hand-counted vmcnt + s_barrier before the data is read).
template <bool ASM> __device__ __forceinline__ void lds_dma16(const void* g, void* lds) {
  if constexpr (ASM) {
#ifdef HBW_NO_DMA
    return;
#endif
    const unsigned l = (unsigned)(unsigned long)(lds_void_t*)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory", "m0");
  } else {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)lds, 16, 0, 0);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Timing ablations: p.debug 1 = no DMA, 2 = no MFMA (experiments only; sdamd_set_gemm_impl(debug << 8)).
// For the dense GEMM instantiations they are compiled out unless -DSDAMD_GEMM_ABLATE: as runtime branches they
// split the K-loop body and cost the 8-wave 128x320 tile 6-12 %.  The CONV and 4-wave instantiations keep the
// (always false) runtime test on purpose: with it hipcc unswitches the K loop and keeps each tile's gather-address
// arithmetic next to its DMA issue; without it the address arithmetic of two tiles is scheduled between the
// MFMAs of one and the 4-wave conv tile drops from 800-930 to 560-620 TFLOP/s (same box,
// experiments/ab_conv.py).
#ifdef SDAMD_GEMM_ABLATE
#define GEMM2_DBG(x) (p.debug == (x))
#else
#define GEMM2_DBG(x) ((CONV || WGM * WGN == 4) && p.debug == (x))
#endif

// -DSDAMD_PP_TIMING (experiments/build_pp_timing.sh, never in the product build): every un-split launch of gemm2_body
// records s_memtime per workgroup -- at entry / first MFMA (ping-pong loop only) / end of the K loop / end of the epilogue
// (slots 0-3; 4-6: after the epilogue's barrier / first pass staged in LDS / first pass stored) -- into the split-K
// workspace it was handed and does not use (GemmArgs::ws; experiments/pp_phases.py reads it back).
#ifdef SDAMD_PP_TIMING
#define PP_STAMP(i) do { if (p.splits == 1 && p.ws && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) \
    reinterpret_cast<long long*>(p.ws)[8 * blockIdx.x + (i)] = __builtin_readcyclecounter(); } while (0)
struct PPStampEnd {
  const GemmArgs& p;
  __device__ ~PPStampEnd() { PP_STAMP(3); }
};
#else
#define PP_STAMP(i) do {} while (0)
#endif

// EPI 0: every epilogue (split-K partials, GeGLU, f32 / hi|lo, operand type), chosen at run time.
// EPI 1: the f32 / hi|lo epilogue with an f32 residual only, the residual prefetched ahead of the K loop.
// EPI 2: the GeGLU epilogue only (bf16).
// EPI 3: EPI 1 (the residual optional) + LayerNorm of the finished rows as a second, bf16 output (GemmArgs::ln_out): for
//        tiles that hold whole rows (BN == N).  diffusion.py:110-117: out_proj + residue, then layernorm_2 of the sum.
// (EPI 1 is a kernel of its own because a first build with the prefetch and every other epilogue in one kernel made
// the launches WITHOUT a residual 6-13 us slower inside a UNet step, where consecutive launches are different
// kernels, while a back-to-back microbenchmark saw nothing: code executed once per launch is not free.)
// PP: 0 = the one-barrier-per-K-tile ring of the header; 1, 2 = the "ping-pong" main loop (see GEMM2 PING-PONG below);
//     3, 4, 5 = the halo-staged conv3x3 loop for images 64, 32, 16 pixels wide (see GEMM2 HALO below);
//     6, 7, 8 = the same on a 256-row tile with one wave per SIMD and a software-pipelined step (GEMM2 HALO BIG-WAVE);
//     9, 10 = the big-wave loop on the phase-decomposed upsample conv (p.phase_up; low-resolution images 32, 16 wide);
//     11 = the big-wave loop on 8 x 8 images (a 256-row tile = four whole images; split-K launches only).
template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, bool CONV, int EPI, int PP>
__device__ __forceinline__ void gemm2_body(const GemmArgs& p) {
#ifdef SDAMD_PP_TIMING
  PP_STAMP(0);
  PPStampEnd pp_stamp_end{p};
#endif
  constexpr bool PPL = PP == 1 || PP == 2;
  constexpr int HALO_W = (PP == 3 || PP == 6) ? 64 : (PP == 4 || PP == 7 || PP == 9) ? 32 : (PP == 5 || PP == 8 || PP == 10) ? 16 : PP == 11 ? 8 : 0;
  constexpr bool HUP = PP == 9 || PP == 10;            // big-wave halo loop on the phase form of upsample + conv3x3: 4 taps (2 x 2) per phase
  constexpr int H_NT = HUP ? 4 : 9;        // filter taps = items per 32-channel chunk = steps per cycle of two chunks
  constexpr bool HALO = HALO_W != 0;
  constexpr bool ASM_DMA = PP >= 6;        // LDS-DMA from inline assembly (lds_dma16); measured on every other kernel: bit-equal results, no gain (+0.0 ... +0.2 % per UNet step)
  constexpr bool HBW = PP >= 6;            // halo loop, one wave per SIMD with a 128x80 wave tile (see GEMM2 HALO BIG-WAVE)
  constexpr bool F32 = sizeof(T) == 4;     // exact-f32 MFMA (v_mfma_f32_16x16x4_f32) on f32 operands
  // f32 operands split into 3 bf16 terms, 6 bf16 MFMAs (split3_bf16).  WPL: the weight operand arrives split already
  // (GemmArgs::W3, three bf16 planes of 32 k = 64 bytes per row and K-tile), so only the activation fragments are split
  // in registers: TM instead of TM + TN fragment splits per K-tile (4 of 9 on the 128x160 tile, 2 of 7 on 64x160).
  constexpr bool WPL = std::is_same<T, f32w_t>::value;
  constexpr bool EMU = std::is_same<T, f32e_t>::value || WPL;
  typedef typename std::conditional<WPL, bf16_t, T>::type TW;   // element type behind the W pointers
  constexpr int EPC = 16 / sizeof(T);      // elements per 16-byte chunk
  constexpr int BKE = 8 * EPC;             // K elements per 128-byte tile row (64 bf16 / 32 f32)
  constexpr int NW = WGM * WGN;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;  // per-wave output tile
  constexpr int TM = WTM / 16, TN = WTN / 16;    // 16x16 MFMA tiles per wave
  // 1-KiB DMA pieces (8 rows) per K-tile: piece g of A (or B) belongs to wave g % NW; when the
  // piece count does not divide, the last round is predicated on the (scalar) wave id
  // (WPL: a 1-KiB piece of W is 16 rows x 64 bytes of one plane; plane p of 16-row block b is piece p*BN/16 + b)
  constexpr int A_PIECES = BM / 8, B_PIECES = WPL ? 3 * (BN / 16) : BN / 8;
  constexpr int A_INS = (A_PIECES + NW - 1) / NW, B_INS = (B_PIECES + NW - 1) / NW;
  constexpr int MIN_PIECES = A_PIECES / NW + B_PIECES / NW;   // fewest pieces any wave issues per tile
  constexpr int NPIECE = A_INS + B_INS;
  constexpr int STAGE = BM * 128 + B_PIECES * 1024;
  static_assert(BM % 8 == 0 && BN % 8 == 0 && WTM % 16 == 0 && WTN % 16 == 0 && STAGES >= 2 && STAGES <= 4, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware (bijective) tile order, n fastest: an XCD's run of tiles shares A rows and the W panel
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int batch = blockIdx.y;   // (split-K launches have batch == 1 and use blockIdx.z = split)
  const T* __restrict__ A = (const T*)p.A + (size_t)batch * p.sA;
  const TW* __restrict__ W = (const TW*)p.W + (size_t)batch * p.sW * (WPL ? 3 : 1);
  // Phase-decomposed upsample conv (CONV, p.phase_up): "batch" is the output phase (py, px).  Output
  // pixel (2y+py, 2x+px) of conv3x3(nearest2x(in)) only sees the 2x2 input window starting at
  // (y-(1-py), x-(1-px)), with the 3x3 taps that land on the same input pixel pre-summed on the host:
  // 4/9 of the MACs.  Rows m index the LOW-resolution grid; out_row() maps them to the output.
  const bool phase_up = CONV && EPI == 0 && p.phase_up;   // (the phase form never carries a residual)
  const int phase = batch + p.phase_base;
  const int py = phase_up ? (phase >> 1) : 0, px = phase_up ? (phase & 1) : 0;
  const int ntx = phase_up ? 2 : 3;                       // taps per filter row
  auto out_row = [&](int m) -> size_t {
    if (!phase_up) return (size_t)m;
    const int hw = p.Ho * p.Wo;
    const int n_img = m / hw, rem = m - n_img * hw;
    const int y = rem / p.Wo, x = rem - y * p.Wo;
    return ((size_t)n_img * 2 * p.Ho + 2 * y + py) * (2 * p.Wo) + 2 * x + px;
  };

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int rr = lane >> 3, cl = lane & 7;

  // per-lane DMA sources: piece i of this wave covers tile rows (wave + NW*i)*8 .. +7
  const TW* wsrc[B_INS];
#pragma unroll
  for (int i = 0; i < B_INS; ++i) {
    if constexpr (WPL) {
      // lane l of a piece: row l>>2 of the 16-row block, 16-byte chunk l&3 of the 64-byte plane row; the chunk index is
      // swizzled with (row>>2)&3 on the source so that the 16 rows a ds_read_b128 lane group touches cover all banks
      const int g = min(wave + NW * i, B_PIECES - 1);
      const int plane = g / (BN / 16), blk = g - plane * (BN / 16);
      const int row = blk * 16 + (lane >> 2);
      const int n = min(n0 + row, p.N - 1);
      wsrc[i] = W + (size_t)n * 3 * p.K + plane * 32 + (((lane & 3) ^ ((row >> 2) & 3)) * 8);
    } else {
      const int row = min(wave + NW * i, B_PIECES - 1) * 8 + rr;
      const int n = min(n0 + row, p.N - 1);
      wsrc[i] = W + (size_t)n * p.ldw + (cl ^ ((row >> 1) & 7)) * EPC;
    }
  }
  // A sources.  Dense: row pointer (+ swizzled chunk).  Conv: pointer to the top-left tap of the pixel's window
  // (may lie outside the image) + a bit mask of the taps that are inside: per K-tile the gather address is then
  // that pointer plus a wave-uniform (scalar) offset (ky*W + kx)*Cin + c0, and padding taps select the zero page.
  // (The gathered nearest-2x form is not built here: launch_gemm2 sends it to gemm_conv.hip; the models use the
  // phase form.)
  const T* asrc[A_INS];
  unsigned amask[A_INS];
#pragma unroll
  for (int i = 0; i < A_INS; ++i) {
    const int row = min(wave + NW * i, A_PIECES - 1) * 8 + rr;
    const int m = min(m0 + row, p.M - 1);
    const int coff = (cl ^ ((row >> 1) & 7)) * EPC;
    amask[i] = 0;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int n_img = m / hw, rem = m - n_img * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int ay = oy * p.stride - (phase_up ? 1 - py : p.pad);
      const int ax = ox * p.stride - (phase_up ? 1 - px : p.pad);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          if (ky < ntx && kx < ntx && (unsigned)(ay + ky) < (unsigned)p.H && (unsigned)(ax + kx) < (unsigned)p.Wd)
            amask[i] |= 1u << (ky * ntx + kx);
      asrc[i] = A + ((long long)n_img * p.H * p.Wd + (long long)ay * p.Wd + ax) * p.Cin + coff;
    } else {
      asrc[i] = A + (size_t)m * p.lda + coff;
    }
  }
  const T* zsrc = reinterpret_cast<const T*>(g_zero_page) + cl * EPC;

  // this workgroup's K-tiles: [kt_begin, kt_begin + nk)
  const int nk_total = p.K / BKE;
  const int split = blockIdx.z;
  const int kt_begin = (int)((long long)nk_total * split / p.splits);
  const int nk = (int)((long long)nk_total * (split + 1) / p.splits) - kt_begin;
  int tap = 0, cc = 0;  // conv K position of the next tile to stage: filter tap, channel offset
  if (CONV) { const int kpt = p.Cin / BKE; tap = kt_begin / kpt; cc = (kt_begin - tap * kpt) * BKE; }
  // DMA piece PI (compile-time) of the tile at K offset k0 / conv position (ky, kx, cc)
  auto stage_piece = [&](auto PIC, char* stage, int k0, int ky, int kx) {
    constexpr int PI = decltype(PIC)::value;
    if constexpr (PI < A_INS) {
      constexpr int i = PI;
      if (A_PIECES % NW != 0 && i == A_INS - 1 && wave + NW * i >= A_PIECES) return;
#ifdef SDAMD_FILL_ABLATE   // timing experiment (wrong results): the activation tile is only fetched for tap (0, 0) = A fill / 9
      if (CONV && p.debug == 3 && (ky | kx)) return;
#endif
      const T* g;
      if (CONV) {
        const int delta = __builtin_amdgcn_readfirstlane((ky * p.Wd + kx) * p.Cin + cc);   // wave-uniform
        g = ((amask[i] >> __builtin_amdgcn_readfirstlane(ky * ntx + kx)) & 1u) ? asrc[i] + delta : zsrc;
      } else {
        g = asrc[i] + k0;
      }
      lds_dma16<ASM_DMA>(g, stage + (wave + NW * i) * 1024);
    } else {
      constexpr int i = PI - A_INS;
      if (B_PIECES % NW != 0 && i == B_INS - 1 && wave + NW * i >= B_PIECES) return;
      lds_dma16<ASM_DMA>(wsrc[i] + k0 * (WPL ? 3 : 1), stage + BM * 128 + (wave + NW * i) * 1024);
    }
  };
  auto advance_k = [&]() { if (CONV) { cc += BKE; if (cc >= p.Cin) { cc = 0; ++tap; } } };
  auto stage_tile = [&](char* stage, int k0) {   // all pieces back to back (prologue)
    if (GEMM2_DBG(1)) return;
    int ky = 0, kx = 0;
    if (CONV) { ky = tap / ntx; kx = tap - ky * ntx; }
    [&]<int... PI>(std::integer_sequence<int, PI...>) { (stage_piece(std::integral_constant<int, PI>{}, stage, k0, ky, kx), ...); }
    (std::make_integer_sequence<int, NPIECE>{});
    advance_k();
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, q4 = lane >> 4;
  // MFMAs of one K-tile; optionally (INTERLEAVE_DMA) with the DMA pieces of a later tile spread
  // between the MFMA groups instead of issued back to back ahead of them.  Measured: no gain on the
  // 8-wave tiles and -25 % on the 4-wave 128x160 tile, so it is off; the limiter of the K loop is the
  // L2 -> LDS fill rate (~60 GB/s per CU), i.e. bytes per FLOP, i.e. the tile size.
  constexpr bool INTERLEAVE_DMA = false;
  auto compute_tile = [&](const char* stage, char* next_stage, bool do_stage, int k0) {
    const char* sa = stage;
    const char* sw = stage + BM * 128;
    int ky = 0, kx = 0;
    if (CONV) { ky = tap / ntx; kx = tap - ky * ntx; }
    if (GEMM2_DBG(1)) do_stage = false;
    constexpr int NGROUP = 2 * TN;
    // one 16-byte fragment per lane and tile: 8 bf16 (one 16x16x32 MFMA) or 4 f32 (four 16x16x4
    // MFMAs; the k order inside the group is permuted alike for both operands)
    typedef typename std::conditional<F32, f32x4_t, bf16x8_t>::type frag_t;
    // The two K halves (g) of the tile use separate fragment registers and the reads of half 1 are
    // issued before the MFMAs of half 0 (sched_barriers pin that order): left to itself hipcc
    // re-uses one register quad and emits "ds_read; s_waitcnt lgkmcnt(0); 4 MFMAs" ten times per
    // tile, i.e. an exposed LDS latency every 64 MFMA cycles.
    // Measured (same box, experiments/gemm_bench.py): +3-9 % on the 8-wave 128x320 tile, -35 % on the 4-wave
    // 128x160 tile (two workgroups per CU: its waves interleave best at the compiler's fine grain), so 8 waves only.
    constexpr bool PREFETCH = !INTERLEAVE_DMA && !EMU && NW == 8 && (TM + TN) <= 9;
    frag_t xf[2][TM], wf[2][TN];
    auto load_frags = [&](int g) {
#pragma unroll
      for (int i = 0; i < TM; ++i) xf[g][i] = *reinterpret_cast<const frag_t*>(sa + swz2(wm * WTM + i * 16 + r16, 4 * g + q4));
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[g][j] = *reinterpret_cast<const frag_t*>(sw + swz2(wn * WTN + j * 16 + r16, 4 * g + q4));
    };
    if constexpr (PREFETCH) {
      load_frags(0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(1);
      // one scheduling region = [reads of half 1 | MFMAs of half 0], pinned to "1 ds_read, then a few MFMAs":
      // the reads ride under the matrix pipe instead of bursting (a burst of all 18 reads with every wave of
      // the workgroup behind the same barrier measured 40 % slower on the 4-wave tile)
      constexpr int NRD = TM + TN, NMF = TM * TN * (F32 ? 4 : 1), PER = NMF / NRD;
#pragma unroll
      for (int r = 0; r < NRD; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMF - PER * NRD, 0);
    }
    if constexpr (EMU) {
      if constexpr (WPL) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int i = 0; i < TM; ++i) xf[g][i] = *reinterpret_cast<const frag_t*>(sa + swz2(wm * WTM + i * 16 + r16, 4 * g + q4));
      } else {
        load_frags(0);
        load_frags(1);
      }
      bf16x8_t xh[TM], xm[TM], xl[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) split3_bf16(xf[0][i], xf[1][i], xh[i], xm[i], xl[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bf16x8_t wh, wmid, wl;
        if constexpr (WPL) {   // (sdamd_split3_weights stores the 32 k of a tile in the fragment order of the f32 reads)
          const char* wb = sw + ((wn * WTN + j * 16) >> 4) * 1024 + r16 * 64 + ((q4 ^ ((r16 >> 2) & 3)) << 4);
          wh = *reinterpret_cast<const bf16x8_t*>(wb);
          wmid = *reinterpret_cast<const bf16x8_t*>(wb + (BN / 16) * 1024);
          wl = *reinterpret_cast<const bf16x8_t*>(wb + 2 * (BN / 16) * 1024);
        } else {
          split3_bf16(wf[0][j], wf[1][j], wh, wmid, wl);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {   // small terms first
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
      if constexpr (!PREFETCH) load_frags(g);
      [&]<int... J>(std::integer_sequence<int, J...>) {
        ([&] {
          constexpr int j = J;
          if (!GEMM2_DBG(2)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              if constexpr (F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][j][e], xf[g][i][e], acc[j][i], 0, 0, 0);
              } else {
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][j], xf[g][i], acc[j][i], 0, 0, 0);
              }
            }
          }
          // pieces assigned to MFMA group q = g*TN + j: those with PI*NGROUP/NPIECE == q
          [&]<int... PI>(std::integer_sequence<int, PI...>) {
            ([&] {
              if constexpr (PI * NGROUP / NPIECE == 0 * TN + j) { if (g == 0 && do_stage) stage_piece(std::integral_constant<int, PI>{}, next_stage, k0, ky, kx); }
              if constexpr (PI * NGROUP / NPIECE == 1 * TN + j) { if (g == 1 && do_stage) stage_piece(std::integral_constant<int, PI>{}, next_stage, k0, ky, kx); }
            }(), ...);
          }(std::make_integer_sequence<int, NPIECE>{});
        }(), ...);
      }(std::make_integer_sequence<int, TN>{});
      if constexpr (PREFETCH) __builtin_amdgcn_sched_barrier(0);
    }
    if (do_stage) advance_k();
  };

  // main loop (see the header): tile t lives in ring buffer t % STAGES
#define GEMM2_STEP(BUF, NEXTBUF)                                                         \
  {                                                                                      \
    if (kt + STAGES - 1 <= nk) wait_vmcnt<(STAGES - 2) * MIN_PIECES>(); else wait_vmcnt<0>(); \
    __builtin_amdgcn_s_barrier();                                                        \
    if (INTERLEAVE_DMA) {                                                                \
      compute_tile(smem + (BUF) * STAGE, smem + (NEXTBUF) * STAGE, kt + STAGES - 1 < nk, (kt_begin + kt + STAGES - 1) * BKE); \
    } else {                                                                             \
      if (kt + STAGES - 1 < nk) stage_tile(smem + (NEXTBUF) * STAGE, (kt_begin + kt + STAGES - 1) * BKE); \
      compute_tile(smem + (BUF) * STAGE, nullptr, false, 0);                             \
    }                                                                                    \
    if (++kt >= nk) break;                                                               \
  }
  int kt = 0;
  // ---- GEMM2 PING-PONG (PP): 8 waves = two groups of 4 (waves 0-3 / 4-7: the two waves of every SIMD are in different
  // groups), 2M x 4N wave grid, 2-stage ring.  A K-tile is four phases, one per quadrant of the wave's sub-tile
  //   p0: read A0, B0 fragments -> MFMA C[B0][A0]      p1: read B1 -> C[B1][A0]
  //   p2: read A1               -> MFMA C[B1][A1]      p3: re-read B0 -> C[B0][A1]
  // and a phase is [LDS reads + the DMA issue of a quarter of the NEXT tile] s_barrier [MFMA cluster] s_barrier.  Group 1
  // runs one barrier behind group 0 (one extra s_barrier before the loop), so on every SIMD one wave is in its MFMA
  // cluster while its partner is in its read / DMA-issue segment: the matrix pipe sees a dense MFMA stream and the LDS
  // reads never stand in front of it (MI355X_MICROARCH.md "Two waves per SIMD").  DMA of tile t+1 (into the other ring
  // buffer) by phase of tile t, per wave: p0 the A pieces with even index (rows of every wave's A0 half), p1 / p2 the W
  // pieces, p3 the odd A pieces (A1 halves).  Counted waits, never zero in the loop: at the end of p3's read segment
  // vmcnt(A1 pieces) retires A0 + W of tile t+1 (first read two barriers later, in p0 of t+1, also for the lagging
  // group); at the end of p1's read segment vmcnt(everything issued after A1(t)) retires A1 of the CURRENT tile (read in
  // p2).  WAR: a ring buffer's rows are re-staged no earlier than two phases after their last read by either group.
  constexpr int PP_TMH = TM / 2, PP_TN0 = (TN + 1) / 2, PP_TN1 = TN - PP_TN0;
  constexpr int PP_WA = B_INS / 2;                       // W pieces issued in p1 (the rest in p2)
  typedef typename std::conditional<F32, f32x4_t, bf16x8_t>::type ppfrag_t;
  [[maybe_unused]] auto pp_issue = [&](auto LO, auto HI, auto STEP, char* stage, int k0, int ky, int kx) {   // pieces LO, LO+STEP, ... < HI
    constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value, st = decltype(STEP)::value;
    if (GEMM2_DBG(1)) return;
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
      (stage_piece(std::integral_constant<int, lo + Q * st>{}, stage, k0, ky, kx), ...);
    }(std::make_integer_sequence<int, (hi - lo + st - 1) / st>{});
  };
  // ---- GEMM2 HALO (PP 3 / 4 / 5): conv3x3, stride 1, pad 1, on images HALO_W = 64 / 32 / 16 pixels wide, 4-wave 128-row
  // tile = BM / HALO_W whole image rows (never across images).  The gathered form stages the tile's 128 input rows once
  // per filter tap -- 9 x 16 KB per 64 input channels, 45 % of everything the tile stages.  Here the K loop runs
  // channel-chunk-major: the (rows + 2) x (HALO_W + 2) input halo of a 32-channel chunk is staged ONCE (64 bytes per
  // pixel, <= 20 KB) and the nine taps are nine address shifts of the activation fragment reads.  A step (one barrier)
  // is two consecutive (chunk, tap) items, K = 2 x 32: the W tile is 160 rows x [32 channels of item 0 | 32 of item 1],
  // laid out and read exactly like the 64-deep tile of the gathered form.  Nine steps = 18 items = two chunks = one
  // "cycle"; the even chunk lives in halo buffer 0, the odd one in buffer 1, so the (buffer, ky, kx) of every fragment
  // read is a compile-time immediate on top of one per-lane offset per (kx, 16-row block): the halo row pitch is a
  // multiple of 16 pixels, which keeps the bank swizzle (chunk ^= (pixel >> 2) & 3) invariant under ky.  DMA per step:
  // the next step's W tile (5 pieces per wave) + the halo pieces of the chunk after next, spread over the four steps
  // in which its buffer is idle (buffer 1: steps 0-3, buffer 0: steps 5-8; 5 pieces per wave and chunk) -- 6.25 pieces
  // per wave and step instead of 9.  LDS: W ring 2 x 20 KB + halo 2 x <= 20 KB = 80 KB, two workgroups per CU.
  // The K order (chunk-major) differs from the gathered form's (tap-major): same products, another f32 summation order.
  constexpr int H_WD = HALO ? HALO_W : BM;
  // 8 x 8 images (H8): a tile is BM / 64 whole images, each with its own 10 x 10 halo at a row pitch of 16 pixels
  constexpr bool H8 = HALO_W == 8;
  constexpr int H_HP = H8 ? 16 : H_WD + 16, H_HR = H8 ? 10 * (BM / 64) : BM / H_WD + 2;   // halo row pitch (pixels) and rows
  constexpr int H_BYTES = H_HR * H_HP * 64, H_PIECES = H_BYTES / 1024, H_INS = (H_PIECES + NW - 1) / NW;
  constexpr int HP3 = HUP ? 9 : H8 ? 4 : 3;                              // big-wave loop: halo pieces per wave and DMA window
  constexpr int H_WST = BN * 128;                                        // one W ring stage
  constexpr int H_S = HBW ? 4 : 2;                                       // W ring depth
  constexpr int H_BASE = HBW ? 0 : 2 * H_WST, H_WBASE = HBW ? 2 * H_BYTES : 0;   // halo buffers / W ring inside smem
  [[maybe_unused]] const T* hsrc[H_INS];       // per-lane source of halo piece wave + NW*i at channel 0
  [[maybe_unused]] const TW* hwsrc[B_INS];     // per-lane source of W piece wave + NW*i at k = 0
  [[maybe_unused]] int aoff[3][TM];            // activation fragment offset inside a halo buffer, per kx and 16-row block
  [[maybe_unused]] bool wsel = false;          // this lane's 16-byte chunk of a W row belongs to item 1 of the step
  [[maybe_unused]] int wpar = 0, hq = 0, hq1 = 0;   // W ring buffer of the current step; current / end cycle
  [[maybe_unused]] auto h_issue_w = [&](int par, int ka, int kb) {
    if (GEMM2_DBG(1)) return;
    const int delta = wsel ? kb : ka;
#pragma unroll
    for (int i = 0; i < B_INS; ++i)
      lds_dma16<ASM_DMA>(hwsrc[i] + delta, smem + H_WBASE + par * H_WST + (wave + NW * i) * 1024);
  };
  [[maybe_unused]] auto h_issue_h = [&](auto IC, int hb, int chunk) {
    constexpr int i = decltype(IC)::value;
    if (GEMM2_DBG(1)) return;
    if (H_PIECES % NW != 0 && i == H_INS - 1 && wave + NW * i >= H_PIECES) return;
    lds_dma16<ASM_DMA>(hsrc[i] + chunk * 32, smem + H_BASE + hb * H_BYTES + (wave + NW * i) * 1024);
  };
  if constexpr (HALO) {
    static_assert(CONV && !F32 && !EMU && NW == 4 && WGM == 2 && (BM == 128 || HBW) && STAGES == 2 && B_PIECES % NW == 0 && BM % H_WD == 0 && (H_WD % 16 == 0 || (H8 && HBW)),
                  "halo loop: bf16 conv on the 4-wave 128-row tile");
    static_assert(H_BYTES % 1024 == 0 && (HBW || (H_INS == 5 && H_BASE + 2 * H_BYTES <= 80 * 1024)), "halo loop: piece schedule / LDS budget (two workgroups per CU)");
    static_assert(!HBW || EPI == 0, "big-wave halo loop: the counted waits assume no other vector-memory loads in flight (no residual prefetch)");
    static_assert(!HUP || (EPI == 0 && H_NT >= H_S), "phase form: EPI 0 only (out_row), and a launch has >= H_S steps");
    static_assert(!HBW || (BM == 256 && H_INS <= 3 * HP3 && 2 * H_BYTES + H_S * H_WST <= 160 * 1024 && B_INS == 5), "big-wave halo loop: piece schedule / LDS budget");
    const int hw = p.Ho * p.Wo;
    const int n_img = m0 / hw, y0 = (m0 - n_img * hw) / HALO_W;
#pragma unroll
    for (int i = 0; i < H_INS; ++i) {
      const int g = min(wave + NW * i, H_PIECES - 1);
      const int slot = g * 16 + (lane >> 2);                            // pixel slot of the halo buffer
      const int hy = slot / H_HP, hx = slot - hy * H_HP;
      const int sc = (lane & 3) ^ ((slot >> 2) & 3);                    // source chunk landing in this lane's position
      const int img = H8 ? hy / 10 : 0;                                 // (H8: image of the tile, row inside its 10-row halo)
      const int y = H8 ? hy - 10 * img - 1 : y0 - 1 + hy, x = hx - 1;
      const bool in = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)HALO_W;
      hsrc[i] = (in ? A + ((long long)((n_img + img) * p.H + y) * HALO_W + x) * p.Cin : reinterpret_cast<const T*>(g_zero_page)) + sc * 8;
    }
#pragma unroll
    for (int i = 0; i < B_INS; ++i) {
      const int row = (wave + NW * i) * 8 + rr;
      const int n = min(n0 + row, p.N - 1);
      hwsrc[i] = W + (size_t)n * p.ldw + ((cl ^ ((row >> 1) & 7)) & 3) * 8;
    }
    wsel = ((cl ^ (((wave * 8 + rr) >> 1) & 7)) >> 2) != 0;             // (32 rows further the swizzle repeats: the same for every piece)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ml = wm * WTM + i * 16 + (lane & 15);
      const int yl = H8 ? 10 * (ml >> 6) + ((ml >> 3) & 7) : ml / HALO_W, xl = ml - (ml / HALO_W) * HALO_W;   // halo row of the pixel
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        // (phase form: tap (dy, dx) of phase (py, px) reads input pixel (y - 1 + py + dy, x - 1 + px + dx): the 3x3 form's
        //  (ky, kx) = (py + dy, px + dx), so the phase only shifts the lane's base slot)
        const int hp = (yl + (HUP ? py : 0)) * H_HP + xl + (HUP ? px : 0) + kx;
        aoff[kx][i] = hp * 64 + ((((lane >> 4) ^ (hp >> 2)) & 3) << 4);
      }
    }
    const int ncyc = p.Cin / 64;
    hq = (int)((long long)ncyc * split / p.splits);
    hq1 = (int)((long long)ncyc * (split + 1) / p.splits);
    [&]<int... I>(std::integer_sequence<int, I...>) { (h_issue_h(std::integral_constant<int, I>{}, 0, 2 * hq), ...); }(std::make_integer_sequence<int, H_INS>{});
    if constexpr (HBW) {   // both chunks of the first cycle, then the first H_S W tiles (a launch has >= 9 steps)
      [&]<int... I>(std::integer_sequence<int, I...>) { (h_issue_h(std::integral_constant<int, I>{}, 1, 2 * hq + 1), ...); }(std::make_integer_sequence<int, H_INS>{});
#pragma unroll
      for (int t = 0; t < H_S; ++t)
        h_issue_w(t, ((2 * t) % H_NT) * p.Cin + hq * 64 + (2 * t >= H_NT ? 32 : 0), ((2 * t + 1) % H_NT) * p.Cin + hq * 64 + (2 * t + 1 >= H_NT ? 32 : 0));
    } else {
      h_issue_w(0, hq * 64, p.Cin + hq * 64);
    }
  } else if constexpr (PPL) {
    static_assert(NW == 8 && WGM == 2 && STAGES == 2 && TM % 2 == 0 && !F32 && !EMU, "ping-pong: 8 waves as 2M x 4N, bf16, 2-stage ring");
    static_assert(A_PIECES % NW == 0 && B_PIECES % NW == 0 && A_INS % 2 == 0 && B_INS >= 2, "ping-pong: uniform DMA piece counts per wave");
    int ky = 0, kx = 0;
    if (CONV) { ky = tap / ntx; kx = tap - ky * ntx; }
    const int k0 = kt_begin * BKE;
    pp_issue(std::integral_constant<int, 0>{}, std::integral_constant<int, A_INS>{}, std::integral_constant<int, 2>{}, smem, k0, ky, kx);           // A0 halves
    pp_issue(std::integral_constant<int, A_INS>{}, std::integral_constant<int, NPIECE>{}, std::integral_constant<int, 1>{}, smem, k0, ky, kx);     // W
    pp_issue(std::integral_constant<int, 1>{}, std::integral_constant<int, A_INS>{}, std::integral_constant<int, 2>{}, smem, k0, ky, kx);           // A1 halves
    advance_k();
  } else {
    stage_tile(smem, kt_begin * BKE);
#pragma unroll
    for (int s = 1; s < STAGES - 1; ++s)
      if (nk > s) stage_tile(smem + s * STAGE, (kt_begin + s) * BKE);
  }
  // f32 residual prefetch: the epilogue's residual chunks (same (m, n) per lane as its row-segment stores) are
  // requested here, behind the first K-tile's DMA, so the fetch overlaps the K loop instead of standing between
  // the last MFMA and the first store -- the short-K GEMMs on the f32 trunk are bound by exactly those bytes.
  // Only where the 80 extra registers fit beside two waves per SIMD (launch2 picks EPI accordingly).
  constexpr bool RPRE = EPI == 1 || EPI == 3;
  constexpr bool LNF = EPI == 3;
  constexpr int E_CPR = WTN / 4;                          // 16-byte chunks per sub-tile row (f32 epilogue)
  constexpr int E_RS = WTN * 4 + 16;
  constexpr int E_IPMAX = (STAGES * STAGE / NW) / (16 * E_RS);
  constexpr int E_IP = E_IPMAX >= TM ? TM : (E_IPMAX >= TM / 2 && TM % 2 == 0) ? TM / 2 : (E_IPMAX >= TM / 4 && TM % 4 == 0) ? TM / 4 : 1;
  constexpr int E_IT = E_IP * 16 * E_CPR / 64;            // chunks per lane and pass
  static_assert(!RPRE || (E_IP * 16 * E_CPR) % 64 == 0, "residual prefetch expects whole lane rounds");
  f32x4_t rpre[RPRE ? (TM / E_IP) * E_IT : 1];
  if constexpr (RPRE) {
    // (the always-true run-time test keeps the block a region of its own: scheduled together with the prologue
    // the 20 address computations stay live across it and the kernel spills)
    if (p.R) {
      const float* __restrict__ Rf = (const float*)p.R + (size_t)batch * p.sR;
      const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
#pragma unroll
      for (int ps = 0; ps < TM / E_IP; ++ps)
#pragma unroll
        for (int it = 0; it < E_IT; ++it) {
          const int idx = lane + it * 64;
          const int row = idx / E_CPR, ch = idx - row * E_CPR;
          const int m = mw + ps * E_IP * 16 + row, n = nw + ch * 4;
          rpre[ps * E_IT + it] = (m < p.M && n < p.N) ? *reinterpret_cast<const f32x4_t*>(Rf + out_row(m) * p.ldr + n) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
    } else if constexpr (LNF) {   // (EPI 1 is only launched with a residual)
#pragma unroll
      for (int k = 0; k < (TM / E_IP) * E_IT; ++k) rpre[k] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  }
  if constexpr (HBW) {
    // ---- GEMM2 HALO BIG-WAVE (PP 6 / 7 / 8): the halo loop on a 256 x 160 tile, 4 waves = ONE wave per SIMD, wave tile
    // 128 x 80 (13 fragment reads per 40 MFMAs instead of 9 per 20; W staged once per 256 rows instead of 128).  With
    // no second wave to hide behind, the step is software-pipelined inside the wave and scheduled by hand: an item
    // (K = 32: 40 MFMAs) is 20 slots of two MFMAs, fenced by sched_barriers; slot k of an item also issues fragment read
    // k of the NEXT item (two fragment register sets) and, in a step's second item, one LDS-DMA piece, so reads and DMA
    // issue ride in the shadow of the matrix pipe.  The LDS-DMA is issued from inline assembly (lds_dma16<true>): with
    // the builtin every LDS wait the compiler inserts is lgkmcnt(0).  The barrier of a step sits in its middle -- after
    // item 0, before the first read of the NEXT step's W tile -- and the W ring is 4 deep: after barrier B(s) the slot
    // of tile s is free (its second half sits in registers) and takes tile s + 4, so a tile has three steps to land.
    // Counted wait at B(s): everything but the W pieces of the two younger tiles (10) and the halo pieces of the last
    // window (3, issued AHEAD of that window's W pieces) has landed: tile s + 1, and every halo piece two windows old.
    // Halo chunks: buffer 0 (even chunk) is last read through fragments loaded before B(4) and refilled in the windows
    // after B(4), B(5), B(6) -> landed at B(8); buffer 1 (odd) is refilled after B(8), B(0), B(1) of the next cycle
    // and first read after B(3).  A window issues exactly three halo pieces per wave (missing ones repeat the wave's
    // piece 0: same bytes, same place).  s_waitcnt lgkmcnt(0) precedes every barrier: no LDS read of a wave is in
    // flight when another wave's DMA into that slot can be issued.
    const int r16h = lane & 15, q4h = lane >> 4;
    int aoff1[3][TM];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < TM; ++i) aoff1[kx][i] = aoff[kx][i] + H_BYTES;
    const int wfb0 = H_WBASE + swz2(wn * WTN + r16h, q4h), wfb1 = H_WBASE + swz2(wn * WTN + r16h, 4 + q4h);
    bf16x8_t xf[2][TM], wf[2][TN];
    const int hq0 = hq;
    constexpr int SC = H_NT;                              // steps per cycle (two chunks x H_NT items, two items per step)
    int wslot = 0, srem = (hq1 - hq) * SC - 1;            // ring slot of the current step; steps after the current one
    constexpr int NRD = TM + TN, NMF = TM * TN, NSLOT = NMF / 2;
    static_assert(NMF % 2 == 0 && NRD <= NSLOT && B_INS + HP3 <= NSLOT && H_INS <= 3 * HP3, "big-wave halo loop: slot schedule");
    // fragment read R of item IT (0..17 of the cycle) into register set G, W tile in ring slot `slot`; order: W 0, A 0..TM-1, W 1..
    auto rd = [&](auto GC, auto ITC, auto RC, int slot) {
      constexpr int g = decltype(GC)::value, it = decltype(ITC)::value, r = decltype(RC)::value;
      constexpr int hb = it >= H_NT ? 1 : 0, tap = it % H_NT, ky = HUP ? tap / 2 : tap / 3, kx = HUP ? tap % 2 : tap % 3;
      if constexpr (r == 0 || r > TM) {
        constexpr int j = r == 0 ? 0 : r - TM;
        wf[g][j] = *reinterpret_cast<const bf16x8_t*>(smem + slot * H_WST + ((it & 1) ? wfb1 : wfb0) + j * 2048);
      } else {
        constexpr int i = r - 1;
        xf[g][i] = *reinterpret_cast<const bf16x8_t*>(smem + H_BASE + ky * H_HP * 64 + (hb ? aoff1[kx][i] : aoff[kx][i]));
      }
    };
    auto mf = [&](auto GC, auto NC) {
      constexpr int g = decltype(GC)::value, n = decltype(NC)::value, j = n / TM, i = n % TM;
      acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][j], xf[g][i], acc[j][i], 0, 0, 0);
    };
    // one item: MFMAs on set G, fragment reads of item ITN into the other set, `extra(k)` in slot k (DMA issue)
    auto item = [&](auto GC, auto ITN, int slot, auto&& extra) {
      constexpr int g = decltype(GC)::value;
      [&]<int... K>(std::integer_sequence<int, K...>) {
        ([&] {
          if constexpr (K < NRD) rd(std::integral_constant<int, 1 - g>{}, ITN, std::integral_constant<int, K>{}, slot);
          extra(std::integral_constant<int, K>{});
          mf(GC, std::integral_constant<int, 2 * K>{});
          mf(GC, std::integral_constant<int, 2 * K + 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }(), ...);
      }(std::make_integer_sequence<int, NSLOT>{});
    };
    auto halo3 = [&](auto KC, auto FIRST, int hb, int chunk) {   // slot KC (0..HP3-1) of a halo window whose first piece slot is FIRST
      constexpr int k = decltype(KC)::value, i = decltype(FIRST)::value + k;
      if constexpr (i < H_INS) {
        if (H_PIECES % NW == 0 || i < H_INS - 1 || wave + NW * i < H_PIECES) { h_issue_h(std::integral_constant<int, i>{}, hb, chunk); return; }
      }
      h_issue_h(std::integral_constant<int, 0>{}, hb, chunk);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    wait_vmcnt<(H_S - 1) * B_INS>();     // halo chunks + W tile 0 landed
    __builtin_amdgcn_s_barrier();
    [&]<int... R>(std::integer_sequence<int, R...>) { (rd(C0{}, C0{}, std::integral_constant<int, R>{}, 0), ...); }(std::make_integer_sequence<int, NRD>{});
    auto hb_step = [&](auto JC) {
      constexpr int J = decltype(JC)::value;
      // item 0: MFMAs of item 2J (set 0), reads of item 2J + 1 (set 1, same W tile)
      item(C0{}, std::integral_constant<int, 2 * J + 1>{}, wslot, [](auto) {});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // B(s): tile s + 1 and every halo piece two windows old have landed
      // (3x3: halo windows after B(4), B(5), B(6) -> buffer 0, after B(8), B(0), B(1) -> buffer 1, three pieces each;
      //  phase form, 4 steps per cycle: after B(1) -> buffer 0, after B(3) -> buffer 1, the whole chunk each)
      const bool hprev = HUP ? (J == 2 ? hq + 1 < hq1 : J == 0 ? hq > hq0 : false)
                             : ((J == 0 || J == 1 || J == 2) ? hq > hq0 : (J >= 5 && J <= 7) ? hq + 1 < hq1 : false);
      if (srem >= 3) { if (hprev) wait_vmcnt<2 * B_INS + HP3>(); else wait_vmcnt<2 * B_INS>(); }
      else if (srem == 2) wait_vmcnt<B_INS>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      // item 1: MFMAs of item 2J + 1 (set 1), reads of the next step's first item (set 0, next W tile), this window's DMA
      const int nslot = (wslot + 1) & (H_S - 1);
      const bool wmore = srem >= H_S;
      constexpr int nj = (J + H_S) % SC, ia = 2 * nj, ib = ia + 1;
      const int kq = (hq + (J + H_S) / SC) * 64;
      const int wdelta = wsel ? (ib % H_NT) * p.Cin + kq + (ib >= H_NT ? 32 : 0) : (ia % H_NT) * p.Cin + kq + (ia >= H_NT ? 32 : 0);
      constexpr bool HWIN = HUP ? (J == 1 || J == 3) : (J <= 1 || (J >= 4 && J <= 6) || J == 8);   // a halo window
      const bool hact = (!HUP && J <= 1) ? hq > hq0 : hq + 1 < hq1;                                   // ... that has a chunk to fetch
      constexpr int hfirst = HUP ? 0 : J <= 1 ? HP3 * (J + 1) : J == 8 ? 0 : HP3 * (J - 4);
      constexpr int hbuf = HUP ? (J == 1 ? 0 : 1) : (J >= 4 && J <= 6) ? 0 : 1;
      const int hchunk = HUP ? (J == 1 ? 2 * hq + 2 : 2 * hq + 3) : J <= 1 ? 2 * hq + 1 : J == 8 ? 2 * hq + 3 : 2 * hq + 2;
      item(C1{}, std::integral_constant<int, 2 * ((J + 1) % SC)>{}, nslot, [&](auto KC) {
        constexpr int k = decltype(KC)::value;
        if constexpr (HWIN && k < HP3) { if (hact) halo3(KC, std::integral_constant<int, hfirst>{}, hbuf, hchunk); }
        if constexpr (k >= HP3 && k < HP3 + B_INS) {
          if (wmore && !GEMM2_DBG(1)) lds_dma16<true>(hwsrc[k - HP3] + wdelta, smem + H_WBASE + wslot * H_WST + (wave + NW * (k - HP3)) * 1024);
        }
      });
      wslot = nslot;
      --srem;
      if constexpr (J == SC - 1) ++hq;
    };
    for (;;) {
      [&]<int... J>(std::integer_sequence<int, J...>) { (hb_step(std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, SC>{});
      if (hq >= hq1) break;
    }
  } else if constexpr (HALO) {
    const int r16h = lane & 15, q4h = lane >> 4;
    auto h_step = [&](auto JC) {
      constexpr int J = decltype(JC)::value;
      const bool more = !(J == 8 && hq + 1 >= hq1);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (more) {   // W tile of the next step
        constexpr int nj = (J + 1) % 9, ia = 2 * nj, ib = ia + 1;
        const int kq = (hq + (J == 8 ? 1 : 0)) * 64;
        h_issue_w(wpar ^ 1, (ia % 9) * p.Cin + kq + (ia >= 9 ? 32 : 0), (ib % 9) * p.Cin + kq + (ib >= 9 ? 32 : 0));
      }
      // halo: the odd chunk of this cycle into buffer 1 while steps 0-3 only read buffer 0; the next cycle's even chunk
      // into buffer 0 once step 4 (its last reader, one barrier ago) is done
      if constexpr (J == 0) { h_issue_h(std::integral_constant<int, 0>{}, 1, 2 * hq + 1); h_issue_h(std::integral_constant<int, 1>{}, 1, 2 * hq + 1); }
      if constexpr (J >= 1 && J <= 3) h_issue_h(std::integral_constant<int, J + 1>{}, 1, 2 * hq + 1);
      if constexpr (J == 5) { if (hq + 1 < hq1) { h_issue_h(std::integral_constant<int, 0>{}, 0, 2 * hq + 2); h_issue_h(std::integral_constant<int, 1>{}, 0, 2 * hq + 2); } }
      if constexpr (J >= 6) { if (hq + 1 < hq1) h_issue_h(std::integral_constant<int, J - 4>{}, 0, 2 * hq + 2); }
      const char* sw = smem + wpar * H_WST;
      bf16x8_t xf[2][TM], wf[2][TN];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int it = 2 * J + g, hb = it >= 9 ? 1 : 0, tap = it % 9, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[g][i] = *reinterpret_cast<const bf16x8_t*>(smem + H_BASE + hb * H_BYTES + ky * H_HP * 64 + aoff[kx][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[g][j] = *reinterpret_cast<const bf16x8_t*>(sw + swz2(wn * WTN + j * 16 + r16h, 4 * g + q4h));
        if (!GEMM2_DBG(2)) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][j], xf[g][i], acc[j][i], 0, 0, 0);
        }
      }
      wpar ^= 1;
      if constexpr (J == 8) ++hq;
    };
    for (;;) {
      [&]<int... J>(std::integer_sequence<int, J...>) { (h_step(std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, 9>{});
      if (hq >= hq1) break;
    }
  } else if constexpr (PPL) {
    ppfrag_t xa[2][PP_TMH], wb0[2][PP_TN0], wb1[2][PP_TN1 > 0 ? PP_TN1 : 1];
    auto ld_a = [&](const char* sa, int half) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < PP_TMH; ++i)
          xa[g][i] = *reinterpret_cast<const ppfrag_t*>(sa + swz2(wm * WTM + (half * PP_TMH + i) * 16 + r16, 4 * g + q4));
    };
    auto ld_b0 = [&](const char* sw) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < PP_TN0; ++j)
          wb0[g][j] = *reinterpret_cast<const ppfrag_t*>(sw + swz2(wn * WTN + j * 16 + r16, 4 * g + q4));
    };
    auto ld_b1 = [&](const char* sw) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < PP_TN1; ++j)
          wb1[g][j] = *reinterpret_cast<const ppfrag_t*>(sw + swz2(wn * WTN + (PP_TN0 + j) * 16 + r16, 4 * g + q4));
    };
    // MFMA cluster of one quadrant: B half `bh` (0: wb0, 1: wb1) x A half `ah`
    auto mm = [&](auto BH, int ah) {
      constexpr int bh = decltype(BH)::value;
      constexpr int nj = bh ? PP_TN1 : PP_TN0;
      if (GEMM2_DBG(2)) return;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < nj; ++j)
#pragma unroll
          for (int i = 0; i < PP_TMH; ++i) {
            if constexpr (bh) acc[PP_TN0 + j][ah * PP_TMH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb1[g][j], xa[g][i], acc[PP_TN0 + j][ah * PP_TMH + i], 0, 0, 0);
            else acc[j][ah * PP_TMH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb0[g][j], xa[g][i], acc[j][ah * PP_TMH + i], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // prologue wait: A0 + W of tile 0 landed (its A1 pieces stay in flight), visible to every wave
    wait_vmcnt<A_INS / 2>();
    __builtin_amdgcn_s_barrier();
    PP_STAMP(1);
    if (wm == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0
    // PP = 1: the B0 fragments stay in registers from p0 to p3; 2: they are re-read in p3 (16 VGPRs fewer on the 256x256
    // tile; measured 0-3 % slower, experiments/pp_bench.py).  Issuing the DMA ahead of the reads, or dropping the
    // s_setprio pair, measured the same within noise.
    constexpr bool DMA_FIRST = false, KEEP_B0 = PP == 1;
#define GEMM2_PP_TILE(BUF, NEXTBUF)                                                                        \
    {                                                                                                      \
      const char* sa = smem + (BUF) * STAGE;                                                               \
      const char* sw = sa + BM * 128;                                                                      \
      char* nx = smem + (NEXTBUF) * STAGE;                                                                 \
      const bool hn = kt + 1 < nk;                                                                         \
      const int k0n = (kt_begin + kt + 1) * BKE;                                                           \
      int ky = 0, kx = 0;                                                                                  \
      if (CONV) { ky = tap / ntx; kx = tap - ky * ntx; }                                                   \
      /* p0 */                                                                                             \
      if (!DMA_FIRST) { ld_a(sa, 0); ld_b0(sw); }                                                          \
      if (hn) pp_issue(C0{}, std::integral_constant<int, A_INS>{}, std::integral_constant<int, 2>{}, nx, k0n, ky, kx); \
      if (DMA_FIRST) { ld_a(sa, 0); ld_b0(sw); }                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      mm(C0{}, 0);                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      /* p1 */                                                                                             \
      if (!DMA_FIRST) ld_b1(sw);                                                                           \
      if (hn) pp_issue(std::integral_constant<int, A_INS>{}, std::integral_constant<int, A_INS + PP_WA>{}, C1{}, nx, k0n, ky, kx); \
      if (DMA_FIRST) ld_b1(sw);                                                                            \
      if (hn) wait_vmcnt<A_INS / 2 + PP_WA>(); else wait_vmcnt<0>();                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      mm(C1{}, 0);                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      /* p2 */                                                                                             \
      if (!DMA_FIRST) ld_a(sa, 1);                                                                         \
      if (hn) pp_issue(std::integral_constant<int, A_INS + PP_WA>{}, std::integral_constant<int, NPIECE>{}, C1{}, nx, k0n, ky, kx); \
      if (DMA_FIRST) ld_a(sa, 1);                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      mm(C1{}, 1);                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      /* p3 */                                                                                             \
      if (!DMA_FIRST && !KEEP_B0) ld_b0(sw);                                                               \
      if (hn) {                                                                                            \
        pp_issue(C1{}, std::integral_constant<int, A_INS>{}, std::integral_constant<int, 2>{}, nx, k0n, ky, kx); \
        advance_k();                                                                                       \
      }                                                                                                    \
      if (DMA_FIRST && !KEEP_B0) ld_b0(sw);                                                                \
      if (hn) wait_vmcnt<A_INS / 2>();                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      mm(C0{}, 1);                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      __builtin_amdgcn_s_barrier();                                                                        \
      if (++kt >= nk) break;                                                                               \
    }
    for (;;) {
      GEMM2_PP_TILE(0, 1)
      GEMM2_PP_TILE(1, 0)
    }
#undef GEMM2_PP_TILE
    if (wm == 0) __builtin_amdgcn_s_barrier();        // group 0 catches up: every wave has passed the same number of barriers
  } else if (STAGES == 2) {
    for (;;) {
      GEMM2_STEP(0, 1)
      GEMM2_STEP(1, 0)
    }
  } else if (STAGES == 3) {
    for (;;) {
      GEMM2_STEP(0, 2)
      GEMM2_STEP(1, 0)
      GEMM2_STEP(2, 1)
    }
  } else {
    for (;;) {
      GEMM2_STEP(0, 3)
      GEMM2_STEP(1, 0)
      GEMM2_STEP(2, 1)
      GEMM2_STEP(3, 2)
    }
  }
#undef GEMM2_STEP

  if (EPI == 0 && p.splits > 1) {  // raw f32 partial sums; bias / residual / activation happen in splitk_reduce_kernel
    float* __restrict__ ws = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WTM + i * 16 + r16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 16 + q4 * 4;
        if (n >= p.N) continue;
        *reinterpret_cast<f32x4_t*>(ws + (size_t)m * p.N + n) = acc[j][i];
      }
    }
    return;
  }

  // Epilogue.  A lane holds out[m][n..n+3] for 4 x TM x TN scattered (m, n): stored directly that is
  // 8-16 bytes per lane at a row stride, which is store-issue bound (measured: the K=320 GEMMs spent
  // more time here than in the K loop).  Instead every wave transposes its WTM x WTN sub-tile
  // through its own slice of the (now idle) staging LDS and writes whole row segments, 16 bytes per
  // lane, consecutive lanes on consecutive addresses.  No workgroup barrier is needed after the one
  // below: a wave only reads back what it wrote itself, and LDS operations of one wave complete in
  // order.
  PP_STAMP(2);
  const float* __restrict__ bias = p.bias;
  const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;  // this wave's sub-tile origin
  // The bias of the wave's TN column tiles, loaded ONCE, as one batch, ahead of the barrier.  (Loaded where it is added --
  // `if (bias && n < N) v += bias[n]` inside the row-tile x column-tile loops -- hipcc emits load, s_waitcnt vmcnt(0), use
  // for every one of the TM x TN iterations: 16 L2 round trips in series in the GeGLU epilogue of a 256x256 tile, ~6 us of
  // the 8.6 us that tile cost beyond its K loop; experiments/pp_phases.py.)  Columns past N read the last valid group:
  // they are never stored.
  const bool has_bias = bias != nullptr;
  f32x4_t bv[TN];
  if (has_bias) {
    const int n_last = (p.N - 1) & ~3;
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const f32x4_t*>(bias + min(nw + j * 16 + q4 * 4, n_last));
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();  // every wave is done reading the last K-tile (and no DMA is outstanding)
  PP_STAMP(4);   // (slots 4..6: after the barrier / first pass staged in LDS / first pass stored)
  if (EPI == 2 || (EPI == 0 && !F32 && p.act == ACT_GEGLU)) {
    // GeGLU epilogue (diffusion.py:128-130): the weight rows were packed so that 16-column tile 2c
    // holds the value channels [16c, 16c+16) and tile 2c+1 their gates; value * gelu_erf(gate) is
    // lane-local on the f32 accumulators and only the N/2 products are rounded and stored (bf16).
    if constexpr (TN % 2 == 0) {
      constexpr int WTO = WTN / 2;                         // output columns of this wave
      constexpr int RS = WTO * 2 + 16;
      constexpr int IPMAX = (STAGES * STAGE / NW) / (16 * RS);
      constexpr int IP = IPMAX >= TM ? TM : (IPMAX >= TM / 2 && TM % 2 == 0) ? TM / 2 : (IPMAX >= TM / 4 && TM % 4 == 0) ? TM / 4 : 1;
      static_assert(IPMAX >= 1, "epilogue staging does not fit");
      constexpr int CPR = WTO / 8;
      char* const slice = smem + wave * (IP * 16 * RS);
      bf16_t* __restrict__ C = (bf16_t*)p.C + (size_t)batch * p.sC;
      const int nwo = (n0 >> 1) + wn * WTO;                // first output column of this wave
      // FAST = the usual launch (a bias, every column pair inside N, the polynomial GELU): no test inside the 16 sub-tile
      // iterations -- with the three of the general form each iteration was basic blocks of its own.
      const bool geglu_fast = has_bias && n0 + BN <= p.N && p.debug != 29;
#pragma unroll
      for (int ps = 0; ps < TM / IP; ++ps) {
        auto stage = [&](auto FAST) {
          constexpr bool fast = decltype(FAST)::value;
#pragma unroll
          for (int ii = 0; ii < IP; ++ii) {
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
              f32x4_t v = acc[j][ps * IP + ii], g = acc[j + 1][ps * IP + ii];
              const int n = nw + j * 16 + q4 * 4;
              if (fast || (has_bias && n + 16 < p.N)) {
                v += bv[j];
                g += bv[j + 1];
              }
              if (!fast && p.debug == 29) {   // (library debug word 29: the f32-accurate erf form, for A/B runs and the accuracy test)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_f(g[r]);
              } else {              // the product is stored as bf16: polynomial erf, no transcendentals (common.h)
                const gelu_f32x2_t g01 = gelu_erf_bf16out_f2((gelu_f32x2_t){g[0], g[1]}), g23 = gelu_erf_bf16out_f2((gelu_f32x2_t){g[2], g[3]});
                v[0] *= g01[0]; v[1] *= g01[1]; v[2] *= g23[0]; v[3] *= g23[1];
              }
              const u32x2_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
              *reinterpret_cast<u32x2_t*>(slice + (ii * 16 + r16) * RS + ((j >> 1) * 16 + q4 * 4) * 2) = o;
            }
          }
        };
        if (geglu_fast) stage(std::true_type{}); else stage(std::false_type{});
        __builtin_amdgcn_wave_barrier();
        if (ps == 0) PP_STAMP(5);
        for (int idx = lane; idx < IP * 16 * CPR; idx += 64) {
          const int row = idx / CPR, ch = idx - row * CPR;
          const int m = mw + ps * IP * 16 + row, n = nwo + ch * 8;
          const u32x4_t v = *reinterpret_cast<const u32x4_t*>(slice + row * RS + ch * 16);
          if (m < p.M && n < (p.N >> 1)) *reinterpret_cast<u32x4_t*>(C + out_row(m) * p.ldc + n) = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (ps == 0) PP_STAMP(6);
      }
    }
  } else if (EPI != 2 && (RPRE || F32 || p.out_f32)) {
    constexpr int RS = WTN * 4 + 16;                     // padded LDS row stride (bytes)
    constexpr int IPMAX = (STAGES * STAGE / NW) / (16 * RS);  // 16-row tiles per pass that fit the wave's slice
    constexpr int IP = IPMAX >= TM ? TM : (IPMAX >= TM / 2 && TM % 2 == 0) ? TM / 2 : (IPMAX >= TM / 4 && TM % 4 == 0) ? TM / 4 : 1;
    static_assert(IPMAX >= 1, "epilogue staging does not fit");
    constexpr int CPR = WTN / 4;  // 16-byte chunks per row
    static_assert(WTM % 64 != 0 || 64 % (IP * 16) == 0, "column statistics: an epilogue pass must not span two 64-row blocks");
    char* const slice = smem + wave * (IP * 16 * RS);
    const float* __restrict__ Rf = p.R ? (const float*)p.R + (size_t)batch * p.sR : nullptr;
    float* __restrict__ C = (float*)p.C + (size_t)batch * p.sC;
    // column statistics of the stored values for a following GroupNorm (GemmArgs::colstats): the finished chunks go
    // back into the wave's LDS slice and every lane then sums one (or two) columns over the pass's rows
    const bool do_stats = p.colstats != nullptr;
    // (1 plain, 2 hi|lo output only, 3 values kept for statistics / LayerNorm only, 0 anything else: tests at run time)
    const int mode32 = p.act != ACT_NONE ? 0 : (p.out_f32 == 2 ? ((do_stats || LNF) ? 0 : 2) : ((do_stats || LNF) ? 3 : 1));
    constexpr int NCR = (WTN + 63) / 64;
    float cs[NCR], cq[NCR];
#pragma unroll
    for (int k = 0; k < NCR; ++k) { cs[k] = 0.f; cq[k] = 0.f; }
#pragma unroll
    for (int ps = 0; ps < TM / IP; ++ps) {
#pragma unroll
      for (int ii = 0; ii < IP; ++ii) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x4_t v = acc[j][ps * IP + ii];
          const int n = nw + j * 16 + q4 * 4;
          if (has_bias) v += bv[j];
          *reinterpret_cast<f32x4_t*>(slice + (ii * 16 + r16) * RS + (j * 16 + q4 * 4) * 4) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (ps == 0) PP_STAMP(5);
      // (the launch-uniform tests -- activation, hi|lo output, values kept for statistics / LayerNorm -- are made once per
      //  pass for the three common combinations instead of four times per chunk, where each ended a basic block)
      auto finish = [&](auto PLAIN, int idx, f32x4_t r, bool have_r) {   // row-segment chunk idx of this pass: residual, activation, store
        constexpr int mode = decltype(PLAIN)::value;          // 0 run-time tests, 1 plain, 2 hi|lo output only, 3 values kept only
        constexpr bool plain = mode != 0;
        const int row = idx / CPR, ch = idx - row * CPR;
        const int m = mw + ps * IP * 16 + row, n = nw + ch * 4;
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(slice + row * RS + ch * 16);
        if (m < p.M && n < p.N) {
          if (have_r) v += r;
          else if (Rf) v += *reinterpret_cast<const f32x4_t*>(Rf + out_row(m) * p.ldr + n);
          if (!plain && p.act == ACT_QUICK_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
          } else if (!plain && p.act == ACT_SILU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
          }
          if (mode == 2 || (mode == 0 && p.out_f32 == 2)) {   // hi|lo: [bf16(v) | bf16(v - bf16(v))] at columns n and N + n of a bf16 [M][2N] matrix
            bf16_t* Ch = (bf16_t*)p.C + (size_t)batch * p.sC + out_row(m) * p.ldc + n;
            float hi[4], lo[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { hi[r] = bf2f(f2bf(v[r])); lo[r] = v[r] - hi[r]; }
            store_n<bf16_t, 4>(Ch, hi);
            store_n<bf16_t, 4>(Ch + p.N, lo);
          } else {
            *reinterpret_cast<f32x4_t*>(C + out_row(m) * p.ldc + n) = v;
          }
          if (mode == 3 || (mode == 0 && (do_stats || LNF))) *reinterpret_cast<f32x4_t*>(slice + row * RS + ch * 16) = v;
        }
      };
      if constexpr (RPRE) {
        auto all = [&](auto MODE) {
#pragma unroll
          for (int it = 0; it < E_IT; ++it) finish(MODE, lane + it * 64, rpre[ps * E_IT + it], true);
        };
        if constexpr (LNF) {   // (one copy: a second one of this pass spilled 11 registers)
          all(std::integral_constant<int, 0>{});
        } else if constexpr (F32 || EMU) {   // (f32 operands: two copies, the other two cost these kernels their second wave per SIMD)
          if (mode32 == 1) all(std::integral_constant<int, 1>{});
          else all(std::integral_constant<int, 0>{});
        } else {
          if (mode32 == 1) all(std::integral_constant<int, 1>{});
          else if (mode32 == 2) all(std::integral_constant<int, 2>{});
          else if (mode32 == 3) all(std::integral_constant<int, 3>{});
          else all(std::integral_constant<int, 0>{});
        }
      } else {
        auto all = [&](auto MODE) {
          for (int idx = lane; idx < IP * 16 * CPR; idx += 64) finish(MODE, idx, (f32x4_t){0.f, 0.f, 0.f, 0.f}, false);
        };
        if constexpr (F32 || EMU) {
          all(std::integral_constant<int, 0>{});
        } else {
          if (mode32 == 1) all(std::integral_constant<int, 1>{});
          else if (mode32 == 2) all(std::integral_constant<int, 2>{});
          else if (mode32 == 3) all(std::integral_constant<int, 3>{});
          else all(std::integral_constant<int, 0>{});
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (ps == 0) PP_STAMP(6);
      if (do_stats) {
        const int row0 = mw + ps * IP * 16;
        const int nrow = min(IP * 16, p.M - row0);          // rows of this pass inside the matrix
#pragma unroll
        for (int k = 0; k < NCR; ++k) {
          const int c = lane + 64 * k;
          if (c < WTN && nw + c < p.N) {
            for (int row = 0; row < nrow; ++row) {
              const float x = *reinterpret_cast<const float*>(slice + row * RS + c * 4);
              cs[k] += x; cq[k] = __builtin_fmaf(x, x, cq[k]);
            }
          }
        }
        if (((ps + 1) * IP * 16) % 64 == 0 || ps == TM / IP - 1) {   // a 64-row block is complete
          const int rb = row0 / 64;
#pragma unroll
          for (int k = 0; k < NCR; ++k) {
            const int c = lane + 64 * k;
            if (c < WTN && nw + c < p.N && rb * 64 < p.M) {
              float* dst = p.colstats + ((size_t)rb * p.N + nw + c) * 2;
              dst[0] = cs[k]; dst[1] = cq[k];
            }
            cs[k] = 0.f; cq[k] = 0.f;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if constexpr (LNF) {
        // LayerNorm of the pass's finished rows (their f32 values sit in the slices of the WGN waves that share a row band).
        // Same lane map, summation order and arithmetic as layernorm_kernel<float, bf16, 16> (norm.hip): 16 lanes per row,
        // lane `sub` owns the 4-column vectors sub, sub + 16, ... of the row.
        static_assert(BN % 64 == 0 && WTN % 4 == 0, "fused LayerNorm: 16 lanes x 4 columns per sweep");
        __syncthreads();
        constexpr int SLICE = IP * 16 * RS, NV = BN / 64;          // NV vectors of 4 columns per lane
        constexpr int PROWS = WGM * IP * 16;                       // rows of this pass in the workgroup
        const int sub = lane & 15;
        bf16_t* __restrict__ Y = (bf16_t*)p.ln_out;
        for (int rr = wave * 4 + (lane >> 4); rr < PROWS; rr += NW * 4) {
          const int band = rr / (IP * 16), rl = rr - band * (IP * 16);
          const int m = m0 + band * WTM + ps * IP * 16 + rl;
          f32x4_t xv[NV];
          float sum = 0.f;
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const int col = (sub + 16 * i) * 4, wq = col / WTN, c = col - wq * WTN;
            xv[i] = *reinterpret_cast<const f32x4_t*>(smem + (band * WGN + wq) * SLICE + rl * RS + c * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += xv[i][j];
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
          const float mean = sum / (float)BN;
          float sq = 0.f;
#pragma unroll
          for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = xv[i][j] - mean; sq += d * d; }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
          const float rstd = 1.0f / sqrtf(sq / (float)BN + p.ln_eps);
          if (m < p.M) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              const int col = (sub + 16 * i) * 4;
              const f32x4_t gm = *reinterpret_cast<const f32x4_t*>(p.ln_gamma + n0 + col);
              const f32x4_t bt = *reinterpret_cast<const f32x4_t*>(p.ln_beta + n0 + col);
              float o[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] = (xv[i][j] - mean) * rstd * gm[j] + bt[j];
              store_n<bf16_t, 4>(Y + (size_t)m * p.N + n0 + col, o);
            }
          }
        }
        __syncthreads();
      }
    }
  } else if constexpr (EPI == 0) {
    constexpr int RS = WTN * 2 + 16;
    constexpr int IPMAX = (STAGES * STAGE / NW) / (16 * RS);
    constexpr int IP = (IPMAX >= TM && TM <= 4) ? TM : (IPMAX >= TM / 2 && TM % 2 == 0) ? TM / 2 : (IPMAX >= TM / 4 && TM % 4 == 0) ? TM / 4 : 1;   // (at most one 64-row block per pass)
    static_assert(IPMAX >= 1, "epilogue staging does not fit");
    constexpr int CPR = WTN / 8;
    static_assert(WTM % 64 != 0 || 64 % (IP * 16) == 0, "column statistics: an epilogue pass must not span two 64-row blocks");
    char* const slice = smem + wave * (IP * 16 * RS);
    const bf16_t* __restrict__ R = p.R ? (const bf16_t*)p.R + (size_t)batch * p.sR : nullptr;
    const bool plain = !R && p.act == ACT_NONE;
    bf16_t* __restrict__ C = (bf16_t*)p.C + (size_t)batch * p.sC;
    // column statistics (GemmArgs::colstats) of the ROUNDED values, as a GroupNorm reading the bf16 tensor would see them:
    // a lane sums one pair of adjacent columns over the pass's rows
    const bool do_stats = p.colstats != nullptr;
    constexpr int NCP = (WTN / 2 + 63) / 64;               // column-pair rounds per lane
    float cs[NCP][2], cq[NCP][2];
#pragma unroll
    for (int k = 0; k < NCP; ++k) { cs[k][0] = cs[k][1] = 0.f; cq[k][0] = cq[k][1] = 0.f; }
#pragma unroll
    for (int ps = 0; ps < TM / IP; ++ps) {
#pragma unroll
      for (int ii = 0; ii < IP; ++ii) {
        const int m = mw + (ps * IP + ii) * 16 + r16;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x4_t v = acc[j][ps * IP + ii];
          const int n = nw + j * 16 + q4 * 4;
          if (has_bias) v += bv[j];
          // (`plain` -- no bf16 residual, no activation: nearly every launch -- is tested once per tile instead of three
          //  launch-uniform tests per 16x16 sub-tile, each of which ended a basic block)
          if (!plain) {
            if (R && m < p.M && n < p.N) {   // (bf16 residual trunks are the rare case: added before rounding)
              const u32x2_t rv = *reinterpret_cast<const u32x2_t*>(R + out_row(m) * p.ldr + n);
              v[0] += __builtin_bit_cast(float, rv[0] << 16); v[1] += __builtin_bit_cast(float, rv[0] & 0xffff0000u);
              v[2] += __builtin_bit_cast(float, rv[1] << 16); v[3] += __builtin_bit_cast(float, rv[1] & 0xffff0000u);
            }
            if (p.act == ACT_QUICK_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
            } else if (p.act == ACT_SILU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
            }
          }
          const u32x2_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
          *reinterpret_cast<u32x2_t*>(slice + (ii * 16 + r16) * RS + (j * 16 + q4 * 4) * 2) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (ps == 0) PP_STAMP(5);
      for (int idx = lane; idx < IP * 16 * CPR; idx += 64) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const int m = mw + ps * IP * 16 + row, n = nw + ch * 8;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(slice + row * RS + ch * 16);
        if (m < p.M && n < p.N) *reinterpret_cast<u32x4_t*>(C + out_row(m) * p.ldc + n) = v;
      }
      if (do_stats) {
        const int row0 = mw + ps * IP * 16;
        const int nrow = min(IP * 16, p.M - row0);
#pragma unroll
        for (int k = 0; k < NCP; ++k) {
          const int c = 2 * (lane + 64 * k);                 // this lane's column pair
          if (c < WTN && nw + c < p.N) {
            for (int row = 0; row < nrow; ++row) {
              const unsigned u = *reinterpret_cast<const unsigned*>(slice + row * RS + c * 2);
              const float x0 = __builtin_bit_cast(float, u << 16), x1 = __builtin_bit_cast(float, u & 0xffff0000u);
              cs[k][0] += x0; cq[k][0] = __builtin_fmaf(x0, x0, cq[k][0]);
              cs[k][1] += x1; cq[k][1] = __builtin_fmaf(x1, x1, cq[k][1]);
            }
          }
        }
        if (((ps + 1) * IP * 16) % 64 == 0 || ps == TM / IP - 1) {
          const int rb = row0 / 64;
#pragma unroll
          for (int k = 0; k < NCP; ++k) {
            const int c = 2 * (lane + 64 * k);
            if (c < WTN && nw + c < p.N && rb * 64 < p.M)
              *reinterpret_cast<f32x4_t*>(p.colstats + ((size_t)rb * p.N + nw + c) * 2) = (f32x4_t){cs[k][0], cq[k][0], cs[k][1], cq[k][1]};
            cs[k][0] = cs[k][1] = 0.f; cq[k][0] = cq[k][1] = 0.f;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (ps == 0) PP_STAMP(6);
    }
  }
}

