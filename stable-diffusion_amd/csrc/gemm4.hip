// Halo-staged instantiations of the implicit-GEMM conv3x3 (gemm2_kernel.h, "GEMM2 HALO"): stride 1, pad 1, bf16, the
// 4-wave 128x160 tile on images 64 / 32 / 16 pixels wide.  Same arguments and epilogues as gemm2_kernel; the K loop sums
// channel-chunk-major instead of tap-major, so results differ from the gathered kernel's by f32 summation order only.
// diffusion.py:9,13 (UNET_ResidualBlock.conv_feature / conv_merged) are the launches it takes.
#include "gemm2_kernel.h"

// One kernel per epilogue; the image width picks one of three loop bodies (the width fixes the halo geometry, i.e. the
// immediate offsets of every fragment read) with a workgroup-uniform branch.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm2halo_kernel(const GemmArgs p) {
  if (p.Wo == 64) gemm2_body<bf16_t, 128, 160, 2, 2, 2, true, EPI, 3>(p);
  else if (p.Wo == 32) gemm2_body<bf16_t, 128, 160, 2, 2, 2, true, EPI, 4>(p);
  else gemm2_body<bf16_t, 128, 160, 2, 2, 2, true, EPI, 5>(p);
}

static constexpr int halo_lds(int w) { return 2 * 160 * 128 + 2 * (128 / w + 2) * (w + 16) * 64; }

template <int EPI>
static int launch_halo(const GemmArgs& a, hipStream_t st) {
  constexpr int lds = halo_lds(64);
  static_assert(halo_lds(64) <= 80 * 1024 && halo_lds(32) <= lds && halo_lds(16) <= lds, "gemm2halo: LDS budget (two workgroups per CU)");
  static std::atomic<unsigned long long> lds_ok{0};
  if (sd_lds_opt_in(lds_ok, (const void*)gemm2halo_kernel<EPI>, lds) != hipSuccess)
    return sdamd_set_error(2, "gemm2halo: cannot reserve %d bytes of LDS", lds);
  dim3 grid(a.tiles_m * a.tiles_n, 1, a.splits);
  gemm2halo_kernel<EPI><<<grid, 256, halo_lds(a.Wo), st>>>(a);
  SD_LAUNCH_CHECK("gemm2halo");
  return 0;
}

// Image width (64 / 32 / 16) when the halo form takes a conv planned on the 128x160 bf16 tile with `splits` K splits,
// else 0.  Whole image rows per 128-pixel tile, tiles never across images, 64-channel cycles that divide evenly over the
// splits (an odd cycle out of 2-3 per split costs more than the fill saves: 8x8-level plans stay on the gathered form).
int gemm2halo_width(const GemmArgs& a, int batch) {
  extern int g_gemm_pp;
  if (g_gemm_pp & 8) return 0;
  if (batch != 1 || a.phase_up || a.ups || a.stride != 1 || a.pad != 1 || a.H != a.Ho || a.Wd != a.Wo) return 0;
  if (a.Wo != 64 && a.Wo != 32 && a.Wo != 16) return 0;
  if ((a.Ho * a.Wo) % 128 != 0 || a.M % 128 != 0 || a.Cin % 64 != 0 || a.Cin > HALO_MAX_CIN || a.K != 9 * a.Cin || a.ldw != a.K) return 0;
  if (a.splits < 1 || (a.Cin / 64) % a.splits != 0) return 0;
  return a.Wo;
}

// The launch `a` as planned for gemm2_kernel<bf16, 128, 160, 2, 2, 2, true, epi> (tiles_m / tiles_n / splits set);
// returns -1 when the halo form does not take it.
int launch_gemm2halo(const GemmArgs& a, int epi, int batch, hipStream_t st) {
  if ((epi != 0 && epi != 1) || !gemm2halo_width(a, batch)) return -1;
  return epi ? launch_halo<1>(a, st) : launch_halo<0>(a, st);
}
