// Launchers, tile / split-K planner and the split-K reduce kernel of the LDS-DMA GEMM / conv family; the kernel body is
// gemm2_kernel.h (shared with gemm3.hip, which instantiates its ping-pong main loop).
#include "gemm2_kernel.h"

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, bool CONV, int EPI>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm2_kernel(const GemmArgs p) {
  gemm2_body<T, BM, BN, WGM, WGN, STAGES, CONV, EPI, 0>(p);
}

// out = act(sum_s ws[s] + bias + residual): the epilogue of a split-K launch; one thread = 4 columns
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
  const int nq = p.N >> 2;
  const size_t total = (size_t)p.M * nq;
  const size_t slab = (size_t)p.M * p.N;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t m_in = idx / nq;
    const int n = (int)(idx - m_in * nq) << 2;
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(p.ws + m_in * p.N + n);
    size_t m = m_in;                       // output row: the low-resolution row itself, or its place in phase (py, px)
    if (p.phase_up) {
      const int hw = p.Ho * p.Wo;
      const int n_img = (int)(m_in / hw), rem = (int)(m_in - (size_t)n_img * hw);
      const int y = rem / p.Wo, x = rem - y * p.Wo;
      m = ((size_t)n_img * 2 * p.Ho + 2 * y + (p.phase_base >> 1)) * (2 * p.Wo) + 2 * x + (p.phase_base & 1);
    }
    // The other splits' partial sums, the bias and an f32 residual are requested as ONE batch and added in the same order
    // as before (split 1, 2, ..., bias, residual): in a loop of unknown length, or under `if (p.bias)` next to its use,
    // every load is waited for before the next one is issued -- splits + 2 memory round trips in series per thread.
    const float* wsp = p.ws + m_in * p.N + n;
    const bool r32 = p.R && p.out_f32;
    f32x4_t bb = {0.f, 0.f, 0.f, 0.f}, rr = {0.f, 0.f, 0.f, 0.f};
    auto batch = [&](auto SC) {
      constexpr int S = decltype(SC)::value;
      f32x4_t part[S > 1 ? S - 1 : 1];
#pragma unroll
      for (int s = 1; s < S; ++s) part[s - 1] = *reinterpret_cast<const f32x4_t*>(wsp + s * slab);
      if (p.bias) bb = *reinterpret_cast<const f32x4_t*>(p.bias + n);
      if (r32) rr = *reinterpret_cast<const f32x4_t*>((const float*)p.R + m * p.ldr + n);
#pragma unroll
      for (int s = 1; s < S; ++s) v += part[s - 1];
    };
    switch (p.splits) {   // (uniform)
      case 2: batch(std::integral_constant<int, 2>{}); break;
      case 3: batch(std::integral_constant<int, 3>{}); break;
      case 4: batch(std::integral_constant<int, 4>{}); break;
      case 8: batch(std::integral_constant<int, 8>{}); break;
      default:
        for (int s = 1; s < p.splits; ++s) v += *reinterpret_cast<const f32x4_t*>(wsp + s * slab);
        batch(std::integral_constant<int, 1>{});
    }
    if (p.bias) v += bb;
    if (p.R) {
      if (p.out_f32) {
        v += rr;
      } else {
        const u32x2_t rv = *reinterpret_cast<const u32x2_t*>((const bf16_t*)p.R + m * p.ldr + n);
        v[0] += __builtin_bit_cast(float, rv[0] << 16); v[1] += __builtin_bit_cast(float, rv[0] & 0xffff0000u);
        v[2] += __builtin_bit_cast(float, rv[1] << 16); v[3] += __builtin_bit_cast(float, rv[1] & 0xffff0000u);
      }
    }
    if (p.act == ACT_QUICK_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
    } else if (p.act == ACT_SILU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
    }
    if (p.out_f32 == 2) {
      bf16_t* Ch = (bf16_t*)p.C + m * p.ldc + n;
      float hi[4], lo[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { hi[r] = bf2f(f2bf(v[r])); lo[r] = v[r] - hi[r]; }
      store_n<bf16_t, 4>(Ch, hi);
      store_n<bf16_t, 4>(Ch + p.N, lo);
    } else if (p.out_f32) {
      *reinterpret_cast<f32x4_t*>((float*)p.C + m * p.ldc + n) = v;
    } else {
      const u32x2_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>((bf16_t*)p.C + m * p.ldc + n) = o;
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, bool CONV, int EPI>
static int launch2_epi(GemmArgs& a, int batch, hipStream_t st) {
  constexpr int lds = STAGES * (BM * 128 + (std::is_same<T, f32w_t>::value ? BN * 192 : BN * 128));
  static_assert(lds <= 160 * 1024, "gemm2: LDS budget");
  if constexpr (CONV && std::is_same<T, bf16_t>::value && BM == 128 && BN == 160 && STAGES == 2 && (EPI == 0 || EPI == 1)) {
    const int r = launch_gemm2halo(a, EPI, batch, st);   // halo-staged form (gemm4.hip) where the geometry allows it
    if (r >= 0) return r;
  }
  static std::atomic<unsigned long long> lds_ok{0};
  if (sd_lds_opt_in(lds_ok, (const void*)gemm2_kernel<T, BM, BN, WGM, WGN, STAGES, CONV, EPI>, lds) != hipSuccess)
    return sdamd_set_error(2, "gemm2: cannot reserve %d bytes of LDS", lds);
  dim3 grid(a.tiles_m * a.tiles_n, batch, a.splits);
  gemm2_kernel<T, BM, BN, WGM, WGN, STAGES, CONV, EPI><<<grid, WGM * WGN * 64, lds, st>>>(a);
  SD_LAUNCH_CHECK("gemm2");
  return 0;
}

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, bool CONV>
static int launch2(GemmArgs& a, int batch, hipStream_t st) {
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  // column statistics ride the f32 / bf16 epilogues of an un-split, un-batched launch whose 64-row blocks are whole
  // wave sub-tiles; everything else clears the request and the consumer computes its own statistics
  // f32-operand kernels (the fp32 mode, 1e-4 budget) never produce them: the partials are f32 sums of x and x^2 over
  // 64 rows and the consumer forms E[x^2] - E[x]^2, whose error grows as 2^-21 * (mean^2 + var) / var -- fine inside the
  // bf16 mode's 2e-2, not for data with |mean| >> std at 1e-4; an fp32 GroupNorm keeps its own f64 statistics pass.
  if (a.colstats && (sizeof(T) == 4 || a.splits != 1 || batch != 1 || a.phase_up || a.out_f32 == 2 || a.act == ACT_GEGLU ||
                     (BM / WGM) % 64 != 0 || (a.N % 2) != 0))
    a.colstats = nullptr;
  a.colstats_done = a.colstats != nullptr;
  if (a.ln_out) {   // fused LayerNorm (EPI 3): whole rows in one tile
    if constexpr (!CONV && BM == 128 && BN == 320 && sizeof(T) == 2) {
      if (a.N != BN || a.out_f32 != 1 || a.act != ACT_NONE || a.splits != 1 || batch != 1) return -1;
      return launch2_epi<T, BM, BN, WGM, WGN, STAGES, CONV, 3>(a, batch, st);
    } else {
      return -1;
    }
  }
  // big-wave halo conv (gemm5.hip): every epilogue of EPI 0 (the residual, if any, is read in the epilogue)
  if constexpr (CONV && std::is_same<T, bf16_t>::value && BM == 128 && BN == 160 && STAGES == 2) {
    const int r = launch_gemm2hbw(a, batch, st);
    if (r > 0) return r;
    if (r == 0) {
      if (a.splits > 1) return launch_splitk_reduce(a, st);
      return 0;
    }
  }
  // residual-prefetch kernel (EPI 1): f32 residual + f32 / hi|lo output, no split, on the tiles with the registers for it
  constexpr bool HAS_RPRE = WGM * WGN == 4 || (!CONV && BM == 128 && BN == 320);
  if constexpr (HAS_RPRE) {
    if (a.R && (sizeof(T) == 4 || a.out_f32) && a.splits == 1 && a.act != ACT_GEGLU)
      return launch2_epi<T, BM, BN, WGM, WGN, STAGES, CONV, 1>(a, batch, st);
  }
  if constexpr (sizeof(T) == 2 && !CONV && (BN / WGN / 16) % 2 == 0) {
    if (a.act == ACT_GEGLU) return launch2_epi<T, BM, BN, WGM, WGN, STAGES, CONV, 2>(a, batch, st);
  }
  if (int e = launch2_epi<T, BM, BN, WGM, WGN, STAGES, CONV, 0>(a, batch, st)) return e;
  if (a.splits > 1) {
    GemmArgs r = a;
    if (sizeof(T) == 4) r.out_f32 = 1;   // f32 operands: C and the residual are f32
    return launch_splitk_reduce(r, st);
  }
  return 0;
}

int launch_splitk_reduce(const GemmArgs& a, hipStream_t st) {
  size_t items = (size_t)a.M * (a.N >> 2);
  int g = (int)((items + 255) / 256 > 2048 ? 2048 : (items + 255) / 256);
  splitk_reduce_kernel<<<g, 256, 0, st>>>(a);
  SD_LAUNCH_CHECK("splitk_reduce");
  return 0;
}

// Tile and split-K plan.  Tile family: 128x320 / 128x160 (every UNet channel count is a multiple of
// 320; the 160-wide tile runs as two 4-wave workgroups per CU and doubles the tile count of the
// low-resolution levels) and 256x256 (VAE / CLIP).  When even the small tile leaves most CUs idle
// (8x8 / 16x16 feature maps with K = 9*Cin up to 23040), K is split across grid.z and a reduce
// kernel finishes the epilogue; that needs the caller's workspace.
// mode: 0 = auto, 2 = any fitting tile however few workgroups, 3 = as 2 plus a 3-way K split.
// Returns BM*1000+BN (0: use the 4-wave kernels of gemm_conv.hip) and *splits.
int gemm2_choose_tile(int M, int N, int K, int batch, int mode, size_t ws_bytes, int* splits, int bke, int* stages) {
  extern int g_gemm_debug;
  if (stages) *stages = 2;
  const size_t slab = (size_t)M * N * sizeof(float);     // split-K partials of the real problem
  M = plan_rows(M);                                      // deterministic plans: size the grid as one sample's rows
  auto blocks = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
  const int nk = K / bke;
  auto max_splits = [&]() { return (batch == 1 && slab > 0) ? (int)(ws_bytes / slab) : 1; };
  *splits = 1;
  if (mode >= 2) {
    int tile = N % 320 == 0 ? 128320 : N % 160 == 0 ? 128160 : N % 256 == 0 ? 256256 : 0;
    if (tile && mode == 3 && nk >= 3 && max_splits() >= 3) *splits = 3;
    return tile;
  }
  if (g_gemm_debug >= 32) {   // experiment: forced tile / split (debug = 32 + 8*tile + splits)
    static const int tiles[8] = {128160, 128320, 256256, 128128, 256320, 128160, 64160, 64160};   // index 5: 128x160 with a 3-stage ring, 7: 64x160 with a 4-stage ring
    const int s = (g_gemm_debug - 32) & 7;
    if (s >= 1 && s <= max_splits() && s <= nk) *splits = s;
    const int ti = ((g_gemm_debug - 32) >> 3) & 7;
    if (stages && ti == 5) *stages = 3;
    if (stages && ti == 7) *stages = 4;
    return tiles[ti];
  }
  // bf16: the 4-wave 128x160 tile (two co-resident workgroups per CU overlap each other's prologue and
  // epilogue) beats the 8-wave 128x320 one by 3-9 % once it fills two rounds (>= 512 workgroups), except on
  // short-K GEMMs with many 128x320 tiles (measured at UNet batch 16, experiments/gemm_bench.py 0,8448)
  const long long b160 = N % 160 == 0 ? blocks(128, 160) : 0, b320 = N % 320 == 0 ? blocks(128, 320) : 0;
  if (bke == 64 && b160 >= 512 && (K >= 2560 || b320 < 512)) return 128160;
  // 64x160 tile (4 waves, two workgroups per CU).  In forced-tile microbenchmarks (experiments/ab_tile64_b16.py,
  // ab_tile64.py) it beat the 128-row plans on the bf16 16x16-level GEMMs (M 4096 N 1280: K 2560 49.1 -> 39.2 us) and the
  // 8x8-level convs (56.7 -> 51.2 us), but the same rules measured IN SITU (experiments/ab_step.py lib:16, captured UNet
  // step, same process) were +0.3 % at UNet batch 16 and +3.3 % at batch 2, so bf16 keeps the 128-row plans.  Kept for
  // f32: M 8192 N 320 short-K GEMMs (64x64 level at UNet batch 2), where un-split 64x160 fills the chip once
  // (32.3 -> 24.1 us) and the alternative is the register-staged 128x64 kernel; g_gemm_debug == 16 switches it off.
  if (g_gemm_debug != 16 && N % 160 == 0 && bke == 32) {
    const long long b64 = blocks(64, 160);
    if (b160 < 256 && b64 >= 256 && b320 < 256 && nk <= 40) return 64160;
  }
  // (Round 3 re-measured the 64x160 tile for just the bf16 16x16-level GEMMs -- M 4096, N 1280, K <= 2560, exactly one
  //  128x160 workgroup per CU: -19...-31 % per launch with rotated operands (experiments/smallgrid.py), -0.03 % per
  //  captured UNet step: again not kept.)
  if (b320 >= 256) return 128320;
  if (bke == 32) {
    // f32 operands (exact 16x16x4 MFMA, ~4x the issue cycles per byte staged): the 8-wave layout only
    // wins over the register-staged 64-wide tiles once >= 256 workgroups exist (measured at UNet
    // batch 2: 105-118 TF vs 65-95 TF above that line, 15-57 TF vs 55-85 TF below it)
    if (N % 160 != 0) return 0;
    const long long b = blocks(128, 160);
    int s = (int)((256 + b - 1) / b);
    if (s > 8 || (s > 1 && s > nk / 16) || s > max_splits()) return 0;
    *splits = s;
    return 128160;
  }
  if (N % 160 == 0) {
    // 4-wave tile, two workgroups per CU: aim at >= 512 workgroups, splitting K up to 4 ways while
    // every split keeps >= 16 K-tiles (measured: 16x16 maps S=2 607->832 TF, 8x8 maps S=4 174->484 TF), up to 8
    // ways on the long-K convs (K >= 8192; in situ the 8x8-level convs S=4 -> 8: 483 -> 539 TF at K=11520,
    // 565 -> 658 TF at K=23040, while the K=5120 GEMMs lose with S=5)
    const long long b = blocks(128, 160);
    int s = (int)((512 + b - 1) / b);
    if (s > (nk >= 128 ? 8 : 4)) s = nk >= 128 ? 8 : 4;
    if (s > nk / 16) s = nk / 16;
    if (s > max_splits()) s = max_splits();
    if (s >= 2) { *splits = s; return 128160; }
    if (b >= 128) return 128160;
    return 0;
  }
  if (N % 256 == 0 && blocks(256, 256) >= 192) return 256256;
  // 128-wide outputs (the VAE's 512 x 512 level: decoder.py:117-122, 2 M rows): an 8-wave 256x128 tile instead of the
  // register-staged 128x128 kernel (551 TF on those convs)
  if (bke == 64 && N % 128 == 0 && blocks(256, 128) >= 256) return 256128;
  return 0;
}

template <typename T>
static int launch_gemm2_t(GemmArgs& a, int batch, bool conv, int mode, hipStream_t st) {
  if (conv && a.ups) return -1;   // gathered nearest-2x conv: gemm_conv.hip (the models use the phase form)
  if (a.out_f32 == 2 && (sizeof(T) != 2 || conv || a.act != ACT_NONE || a.ldc % 4 != 0)) return -1;   // hi|lo output: bf16 dense GEMM only
  if ((a.ldr % 4) != 0 || (a.R && ((uintptr_t)a.R % 16) != 0)) return -1;
  if (sizeof(T) == 2 && !a.out_f32 && ((a.ldc % 8) != 0 || (a.sC % 8) != 0)) return -1;   // 16-byte row-segment stores
  if (sizeof(T) == 4 && ((a.ldc % 4) != 0 || (a.sC % 4) != 0)) return -1;
  int stages = 2;
  int tile = gemm2_choose_tile(a.M, a.N, a.K, batch, mode, a.ws ? a.ws_bytes : 0, &a.splits, 128 / (int)sizeof(T), &stages);
  if (a.ln_out) {   // fused LayerNorm: the one tile whose N extent is the whole row
    if (sizeof(T) != 2 || conv || a.N != 320) return -1;
    tile = 128320; a.splits = 1; stages = 2;
  }
  if (a.act == ACT_GEGLU) {   // needs an even number of 16-column tiles per wave: 256x256 or 128x128
    if (sizeof(T) == 4 || conv || a.out_f32 || a.R || a.N % 256 != 0) return -1;
    extern int g_gemm_debug;
    tile = g_gemm_debug == 6 ? 128128 : 256256;
    a.splits = 1;
  }
  if constexpr (std::is_same<T, f32w_t>::value) {   // 2 x (BM*128 + BN*192) bytes of LDS: 64x160 two workgroups per CU, the others one
    switch (tile) {
      case 128160:
        // (92 KB of LDS: one workgroup per CU.  Only where the launch has no second workgroup per CU to lose:
        //  M 8192 N 960 K 320, 384 workgroups, measured 44.9 us with in-register weight splits, 49.9 us with planes)
        if ((long long)((plan_rows(a.M) + 127) / 128) * ((a.N + 159) / 160) * batch * a.splits > 256) { a.splits = 1; return -1; }
        return conv ? launch2<T, 128, 160, 2, 2, 2, true>(a, batch, st) : launch2<T, 128, 160, 2, 2, 2, false>(a, batch, st);
      case 64160: return conv ? launch2<T, 64, 160, 2, 2, 2, true>(a, batch, st) : launch2<T, 64, 160, 2, 2, 2, false>(a, batch, st);
      case 128320: return conv ? launch2<T, 128, 320, 2, 4, 2, true>(a, batch, st) : launch2<T, 128, 320, 2, 4, 2, false>(a, batch, st);
      default: a.splits = 1; return -1;
    }
  } else {
  if constexpr (sizeof(T) == 2) {   // ping-pong instantiations (gemm3.hip)
    extern int g_gemm_pp;
    if ((g_gemm_pp & 1) && tile == 256256) {
      const int r = launch_gemm2pp(a, 256256, batch, conv, st);
      if (r >= 0) return r;
    }
  }
  switch (tile) {
    case 128128: return conv ? -1 : launch2<T, 128, 128, 2, 2, 2, false>(a, batch, st);
    case 128160:
      if (stages == 3) return conv ? launch2<T, 128, 160, 2, 2, 3, true>(a, batch, st) : launch2<T, 128, 160, 2, 2, 3, false>(a, batch, st);
      return conv ? launch2<T, 128, 160, 2, 2, 2, true>(a, batch, st) : launch2<T, 128, 160, 2, 2, 2, false>(a, batch, st);
    case 64160:
      if (stages == 4) return conv ? launch2<T, 64, 160, 2, 2, 4, true>(a, batch, st) : launch2<T, 64, 160, 2, 2, 4, false>(a, batch, st);
      return conv ? launch2<T, 64, 160, 2, 2, 2, true>(a, batch, st) : launch2<T, 64, 160, 2, 2, 2, false>(a, batch, st);
    case 128320: return conv ? launch2<T, 128, 320, 2, 4, 2, true>(a, batch, st) : launch2<T, 128, 320, 2, 4, 2, false>(a, batch, st);
    case 256256: return conv ? launch2<T, 256, 256, 2, 4, 2, true>(a, batch, st) : launch2<T, 256, 256, 2, 4, 2, false>(a, batch, st);
    case 256128:
      if constexpr (sizeof(T) == 2) return conv ? launch2<T, 256, 128, 4, 2, 2, true>(a, batch, st) : launch2<T, 256, 128, 4, 2, 2, false>(a, batch, st);
      else return -1;
    case 256320:
      if constexpr (sizeof(T) == 2) return conv ? launch2<T, 256, 320, 4, 2, 2, true>(a, batch, st) : launch2<T, 256, 320, 4, 2, 2, false>(a, batch, st);
      else return -1;
    default: a.splits = 1; return -1;
  }
  }
}

int launch_gemm2(GemmArgs& a, int dtype, int batch, bool conv, int mode, hipStream_t st) {
  extern int g_f32_emu;
  if (dtype != SDAMD_BF16 && g_f32_emu) {
    if (a.W3 && (batch == 1 || a.sW == (long long)a.N * a.K)) {   // pre-split weight planes (packed per batch entry)
      GemmArgs b = a;
      b.W = a.W3;
      const int r = launch_gemm2_t<f32w_t>(b, batch, conv, mode, st);
      if (r >= 0) { a.splits = b.splits; a.colstats_done = b.colstats_done; return r; }
    }
    return launch_gemm2_t<f32e_t>(a, batch, conv, mode, st);
  }
  return dtype == SDAMD_BF16 ? launch_gemm2_t<bf16_t>(a, batch, conv, mode, st) : launch_gemm2_t<float>(a, batch, conv, mode, st);
}
