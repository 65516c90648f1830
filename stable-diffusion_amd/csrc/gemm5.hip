// Big-wave instantiation of the halo-staged conv3x3 (gemm2_kernel.h, "GEMM2 HALO BIG-WAVE"): 256 x 160 tile, 4 waves =
// one per SIMD, wave tile 128 x 80 with the accumulators in AGPRs (this file is built WITHOUT -amdgpu-mfma-vgpr-form),
// software-pipelined step, 4-deep W ring.  Same results as gemm4.hip's kernels (same K order per output element).
#include "gemm2_kernel.h"

__global__ __launch_bounds__(256) void gemm2hbw_kernel(const GemmArgs p) {
  if (p.Wo == 64) gemm2_body<bf16_t, 256, 160, 2, 2, 2, true, 0, 6>(p);
  else if (p.Wo == 32) gemm2_body<bf16_t, 256, 160, 2, 2, 2, true, 0, 7>(p);
  else if (p.Wo == 16) gemm2_body<bf16_t, 256, 160, 2, 2, 2, true, 0, 8>(p);
  else gemm2_body<bf16_t, 256, 160, 2, 2, 2, true, 0, 11>(p);   // 8 x 8 images, four per tile
}

// the phase form of nearest-2x upsample + conv3x3 (sdamd_conv3x3 upsample = 2; diffusion.py:148-150): grid.y = output phase
__global__ __launch_bounds__(256) void gemm2hbwup_kernel(const GemmArgs p) {
  if (p.Wo == 32) gemm2_body<bf16_t, 256, 160, 2, 2, 2, true, 0, 9>(p);
  else gemm2_body<bf16_t, 256, 160, 2, 2, 2, true, 0, 10>(p);
}

static constexpr int hbw_lds(int w) { return 4 * 160 * 128 + 2 * (w == 8 ? 40 * 16 : (256 / w + 2) * (w + 16)) * 64; }

// `a` planned for the 128x160 tile (tiles_n, splits); takes it when the halo form would and the image rows tile 256 pixels
// phase form: all four phases in one un-split launch, low-resolution images 32 or 16 wide
static bool hbw_up_takes(const GemmArgs& a, int batch) {
  extern int g_gemm_pp;
  if ((g_gemm_pp & (2 | 8)) || !a.phase_up || batch != 4 || a.phase_base != 0 || a.ups || a.stride != 1 || a.R) return false;
  if (a.H != a.Ho || a.Wd != a.Wo || (a.Wo != 32 && a.Wo != 16) || (a.Ho * a.Wo) % 256 != 0 || a.M % 256 != 0) return false;
  // (one workgroup per CU: with fewer than 256 of them the 128-row kernel's 2x finer grid wins -- UNet batch 2, 32x32: 81 vs 100 us)
  // (deterministic plans size that grid from one sample's rows, like every other planner: this form sums K chunk-major, the
  //  gathered phase kernel tap-major, so the choice must not depend on how many samples share the batch)
  if ((long long)(plan_rows(a.M) / 256) * ((a.N + 159) / 160) * 4 < 256) return false;
  return a.Cin % 64 == 0 && a.Cin <= HALO_MAX_CIN && a.K == 4 * a.Cin && a.ldw == a.K && a.splits == 1;
}

// 8 x 8 images: a 256-row tile = four whole images.  The LDS is full (2 x 40 KB halo + 4 x 20 KB W ring), so only launches whose
// epilogue needs none: K-split partial sums -- which every UNet launch of that level is (32 tiles of 256 x 160).  The
// splits need not divide the 64-channel cycles evenly here (20 cycles over 8 splits: 2 or 3 each).
static bool hbw8_takes(const GemmArgs& a, int batch) {
  extern int g_gemm_pp;
  if ((g_gemm_pp & (2 | 8)) || batch != 1 || a.phase_up || a.ups || a.stride != 1 || a.pad != 1) return false;
  // (plan_rows: a deterministic plan decides from one sample's rows -- two 8x8 maps, never four -- so it keeps the gathered kernel)
  if (a.H != 8 || a.Wd != 8 || a.Ho != 8 || a.Wo != 8 || a.M % 256 != 0 || plan_rows(a.M) % 256 != 0) return false;
  if (a.Cin % 64 != 0 || a.Cin > HALO_MAX_CIN || a.K != 9 * a.Cin || a.ldw != a.K) return false;
  // (measured per launch against the gathered kernel with its 8-way split: Cin 1280 56.2 -> 55.4 us, Cin 2560 90.0 -> 84.7 us --
  //  prologue, partial-sum epilogue and the reduce launch dominate these 18 ... 45-step loops; debug word 25 switches it off)
  extern int g_gemm_debug;
  return g_gemm_debug != 25 && a.splits > 1 && a.splits <= a.Cin / 128;
}

bool gemm2hbw_takes(const GemmArgs& a, int batch) {
  extern int g_gemm_pp, g_gemm_debug;
  if (a.phase_up) return hbw_up_takes(a, batch);
  if (a.Wo == 8) return hbw8_takes(a, batch);
  if ((g_gemm_pp & 2) || !gemm2halo_width(a, batch)) return false;
  if ((a.Ho * a.Wo) % 256 != 0 || a.M % 256 != 0) return false;
  // One workgroup per CU: prologue (140 KB of DMA before the first MFMA) and epilogue are exposed, nothing else runs on the
  // CU meanwhile.  In situ (captured UNet step, batch 16, per launch against the 128-row halo kernel): K = 9 * 1280 and
  // up -5 ... -14 %, every 16x16-level launch (2-way K split, partial-sum epilogue) -9 ... -14 %, K = 9 * 960 +-0,
  // K = 9 * 640 +3.5 ... +7 %: so from Cin 1280 on, and the whole 16x16 level.
  return a.Cin >= 1280 || (a.Wo == 16 && a.Cin >= 640) || g_gemm_debug == 18;
}

int launch_gemm2hbw(GemmArgs& a, int batch, hipStream_t st) {
  if (!gemm2hbw_takes(a, batch)) return -1;
  constexpr int lds = hbw_lds(8);
  static_assert(hbw_lds(8) <= 160 * 1024 && hbw_lds(64) <= lds && hbw_lds(32) <= lds && hbw_lds(16) <= lds, "gemm2hbw: LDS budget");
  static std::atomic<unsigned long long> lds_ok{0};
  if (sd_lds_opt_in(lds_ok, (const void*)gemm2hbw_kernel, lds) != hipSuccess)
    return sdamd_set_error(2, "gemm2hbw: cannot reserve %d bytes of LDS", lds);
  a.tiles_m = a.M / 256;
  if (a.phase_up) {
    static std::atomic<unsigned long long> lds_ok_up{0};
    if (sd_lds_opt_in(lds_ok_up, (const void*)gemm2hbwup_kernel, lds) != hipSuccess)
      return sdamd_set_error(2, "gemm2hbw: cannot reserve %d bytes of LDS", lds);
    gemm2hbwup_kernel<<<dim3(a.tiles_m * a.tiles_n, 4, 1), 256, hbw_lds(a.Wo), st>>>(a);
    SD_LAUNCH_CHECK("gemm2hbwup");
    return 0;
  }
  dim3 grid(a.tiles_m * a.tiles_n, 1, a.splits);
  gemm2hbw_kernel<<<grid, 256, hbw_lds(a.Wo), st>>>(a);
  SD_LAUNCH_CHECK("gemm2hbw");
  return 0;
}
