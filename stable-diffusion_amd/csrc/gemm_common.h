// Argument block shared by the MFMA GEMM / implicit-GEMM conv kernels (gemm_conv.hip, gemm2.hip).
#pragma once
#include "common.h"

#define ACT_NONE 0
#define ACT_QUICK_GELU 1
#define ACT_SILU 2
#define ACT_GEGLU 3   // column tiles alternate value | gate (16 wide): out[m][c] = value * gelu_erf(gate), N/2 columns

struct GemmArgs {
  const void* A;     // dense: [M][lda]; conv: NHWC input [B][H][W][Cin]
  const void* W;     // [N][ldw], K-contiguous (conv: [Cout][ky][kx][Cin])
  const void* W3;    // f32 operands only, optional: W pre-split into bf16 planes [N][K/32][hi|mid|lo][32] (sdamd_split3_weights), packed
  void* C;           // [M][ldc]
  const float* bias; // [N] or null
  const void* R;     // residual [M][ldr] (same dtype as C: f32 when out_f32, else the operand dtype) or null
  int M, N, K;
  int lda, ldw, ldc, ldr;
  long long sA, sW, sC, sR;  // batch strides (elements), grid.y = batch
  int act, out_f32;
  // conv geometry (CONV only)
  int H, Wd, Cin, Ho, Wo, stride, pad, ups;
  int phase_up;  // 1: nearest-2x upsample + conv3x3 as four 2x2 convs, grid.y = output phase (py, px); see sdamd_conv3x3
  int phase_base;  // phase_up: added to grid.y to give the phase (one launch per phase when the phases are K-split)
  int tiles_m, tiles_n;
  // split-K (gemm2.hip): grid.z = splits, each writes raw f32 partials to ws[split][M][N]
  int splits;
  float* ws;
  size_t ws_bytes;
  // Column statistics for a GroupNorm that consumes C (gemm2.hip epilogues; null = off): per 64-row block rb and output
  // column n, colstats[(rb * N + n) * 2 + {0, 1}] = sum and sum of squares of the stored C[rb*64 .. rb*64+63][n].
  // The launcher clears the pointer for launches that cannot produce them (split-K, phase form, hi|lo / GeGLU output,
  // batched, 4-wave register-staged kernels) and sets colstats_done = 1 when the kernel writes them.
  float* colstats;
  int colstats_done;
  // LayerNorm of the stored rows, fused into the epilogue (gemm2_kernel.h EPI 3; null = off): ln_out [M][N] bf16 =
  // LayerNorm(C row) * ln_gamma + ln_beta.  Only for launches whose N-tile holds whole rows (N == 320 on the 128x320 tile).
  void* ln_out;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int debug;   // timing experiments only (sdamd_set_gemm_impl(impl | debug << 8)): 1 = skip the DMA, 2 = skip the MFMAs
};


#ifdef __HIPCC__
// f32 operands on the bf16 matrix pipe ("bf16x3"): every f32 value is split exactly into three bf16 terms
// a = hi + mid + lo (8 + 8 + 8 significant bits; each residual is exact in f32), and a product a*w is accumulated in
// f32 from the six terms of weight >= 2^-16: hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid.  The dropped terms
// (mid*lo, lo*mid, lo*lo) are <= 3 * 2^-24 |a w|, the size of one f32 rounding of the product, so the result is an
// f32 GEMM to within f32 rounding -- at 6 x v_mfma_f32_16x16x32_bf16 (96 cycles per 16x16x32 block) instead of
// 8 x v_mfma_f32_16x16x4_f32 (256 cycles).  Operands are staged as f32 exactly as in the exact-f32 kernel; the split
// is done on the fragments in registers (11 VALU per element pair).  Non-finite inputs give NaN (inf - inf in the
// residual) where the exact kernel would give inf.
struct f32e_t { float v; };
// as f32e_t, with the WEIGHT operand split ahead of time (sdamd_split3_weights): W points to bf16 planes, see gemm2.hip
struct f32w_t { float v; };
template <> struct Cvt<f32e_t> {   // (only reached for f32 outputs: the bf16-output branches are dead for 4-byte operands)
  static __device__ __forceinline__ float ld(const f32e_t* p) { return p->v; }
  static __device__ __forceinline__ void st(f32e_t* p, float v) { p->v = v; }
};
#endif

// Deterministic launch plans (sdamd_set_plan_samples, include/sdamd.h): when a launch's rows are n independent samples the
// planners (tile, split-K, kernel family, GroupNorm partial blocks) size the problem as ONE sample's rows, so that a
// sample's result does not depend on how many others share its batch.  plan_rows(M) = the row count the planners see.
extern int g_plan_samples;
static inline int plan_rows(int M) { return (g_plan_samples > 1 && M % g_plan_samples == 0) ? M / g_plan_samples : M; }

// gemm2.hip: 8-wave, LDS-DMA-staged kernels (bf16 and exact-f32 MFMA).  Returns -1 when no tile configuration of that
// family fits the problem (the caller then uses the 4-wave register-staged kernels), 0 on launch.
int launch_gemm2(GemmArgs& a, int dtype, int batch, bool conv, int mode, hipStream_t st);   // mode: g_gemm_impl
int gemm2_choose_tile(int M, int N, int K, int batch, int mode, size_t ws_bytes, int* splits, int bke, int* stages = nullptr);

// gemm3.hip: the ping-pong instantiations (bf16, 256-row tiles); tile = 256256 | 256320; -1 = not taken
int launch_gemm2pp(GemmArgs& a, int tile, int batch, bool conv, hipStream_t st);

// gemm4.hip: the halo-staged conv3x3 on the 4-wave 128x160 tile (bf16, stride 1, pad 1, images 64 / 32 / 16 wide);
// `a` planned as for gemm2_kernel<bf16, 128, 160, 2, 2, 2, true, epi>; -1 = not taken
int launch_gemm2halo(const GemmArgs& a, int epi, int batch, hipStream_t st);
int launch_gemm2hbw(GemmArgs& a, int batch, hipStream_t st);
bool gemm2hbw_takes(const GemmArgs& a, int batch);   // gemm5.hip: 256-row one-wave-per-SIMD form (sets tiles_m)
int gemm2halo_width(const GemmArgs& a, int batch);   // 64 / 32 / 16 when the halo form takes the (planned) launch, else 0

// out = act(sum_s ws[s] + bias + residual) for a split-K launch (gemm2.hip)
int launch_splitk_reduce(const GemmArgs& a, hipStream_t st);
