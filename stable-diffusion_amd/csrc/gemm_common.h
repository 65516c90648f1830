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
  int debug;   // timing experiments only (sdamd_set_gemm_impl(impl | debug << 8)): 1 = skip the DMA, 2 = skip the MFMAs
};


// gemm2.hip: 8-wave, LDS-DMA-staged kernels (bf16 and exact-f32 MFMA).  Returns -1 when no tile configuration of that
// family fits the problem (the caller then uses the 4-wave register-staged kernels), 0 on launch.
int launch_gemm2(GemmArgs& a, int dtype, int batch, bool conv, int mode, hipStream_t st);   // mode: g_gemm_impl
int gemm2_choose_tile(int M, int N, int K, int batch, int mode, size_t ws_bytes, int* splits, int bke, int* stages = nullptr);

// out = act(sum_s ws[s] + bias + residual) for a split-K launch (gemm2.hip)
int launch_splitk_reduce(const GemmArgs& a, hipStream_t st);
