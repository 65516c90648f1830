// Ping-pong instantiations of the LDS-DMA GEMM / implicit-GEMM conv3x3 (gemm2_kernel.h, "GEMM2 PING-PONG"): 256-row
// tiles, 8 waves as two groups of four that alternate MFMA clusters and LDS-read / DMA-issue segments on every SIMD.
// bf16 operands only.  Same arguments, epilogues and results as gemm2_kernel (the K loop's summation order per output
// element is the same: K-tiles in order, two 32-deep MFMAs per tile), so the results are bit-equal to the ring kernels'.
#include "gemm2_kernel.h"

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, bool CONV, int EPI, int PPV>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm2pp_kernel(const GemmArgs p) {
  gemm2_body<T, BM, BN, WGM, WGN, STAGES, CONV, EPI, PPV>(p);
}

template <int BM, int BN, bool CONV, int EPI, int PPV>
static int launch_pp_epi(GemmArgs& a, int batch, hipStream_t st) {
  constexpr int lds = 2 * (BM * 128 + BN * 128);
  static_assert(lds <= 160 * 1024, "gemm2pp: LDS budget");
  static std::atomic<unsigned long long> lds_ok{0};
  if (sd_lds_opt_in(lds_ok, (const void*)gemm2pp_kernel<bf16_t, BM, BN, 2, 4, 2, CONV, EPI, PPV>, lds) != hipSuccess)
    return sdamd_set_error(2, "gemm2pp: cannot reserve %d bytes of LDS", lds);
  dim3 grid(a.tiles_m * a.tiles_n, batch, 1);
  gemm2pp_kernel<bf16_t, BM, BN, 2, 4, 2, CONV, EPI, PPV><<<grid, 512, lds, st>>>(a);
  SD_LAUNCH_CHECK("gemm2pp");
  return 0;
}

template <int BM, int BN, bool CONV, int PPV>
static int launch_pp(GemmArgs& a, int batch, hipStream_t st) {
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.splits = 1;
  // column statistics: as gemm2.hip launch2 (un-batched, plain f32 / bf16 epilogue; the wave sub-tile is 128 rows)
  if (a.colstats && (batch != 1 || a.phase_up || a.out_f32 == 2 || a.act == ACT_GEGLU || (a.N % 2) != 0)) a.colstats = nullptr;
  a.colstats_done = a.colstats != nullptr;
  if constexpr (!CONV && (BN / 4 / 16) % 2 == 0) {
    if (a.act == ACT_GEGLU) return launch_pp_epi<BM, BN, CONV, 2, PPV>(a, batch, st);
  }
  if (a.act == ACT_GEGLU) return -1;
  return launch_pp_epi<BM, BN, CONV, 0, PPV>(a, batch, st);
}

// tile = 256256; returns -1 when this family does not take the launch
int launch_gemm2pp(GemmArgs& a, int tile, int batch, bool conv, hipStream_t st) {
  extern int g_gemm_pp;
  if (a.K < 128 || tile != 256256) return -1;
  if (g_gemm_pp & 4) return conv ? launch_pp<256, 256, true, 2>(a, batch, st) : launch_pp<256, 256, false, 2>(a, batch, st);
  return conv ? launch_pp<256, 256, true, 1>(a, batch, st) : launch_pp<256, 256, false, 1>(a, batch, st);
}
