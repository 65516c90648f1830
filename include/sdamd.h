/* libsdamd -- C ABI of the MI355X-native (gfx950) kernels behind the SD-v1.5 denoising hot path.
 *
 * The reference (KhoaDuong2005/Stable-Diffusion) has no FFI: its "operators" are torch.nn
 * calls inside Python modules.  Each entry point below names the reference lines it replaces;
 * the Python host code in stable-diffusion_amd/ binds them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; nothing here allocates, frees or
 *     synchronises; `stream` is a hipStream_t (0 = default stream), so every call can be
 *     captured into a hipGraph;
 *   - `dtype`: SDAMD_F32 (0) or SDAMD_BF16 (1) = storage type of activations/weights;
 *     accumulation is always f32, biases / norm gains are always f32;
 *   - activations are NHWC ([B][H][W][C], i.e. [tokens][C]); caller tensors at module
 *     boundaries are NCHW f32 and are converted by the boundary kernels;
 *   - return 0 on success; non-zero = argument/launch error, text in sdamd_last_error()
 *     (thread-local).  Nothing throws across the boundary.
 */
#ifndef SDAMD_H
#define SDAMD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SDAMD_F32 0
#define SDAMD_BF16 1
#define SDAMD_ACT_NONE 0
#define SDAMD_ACT_QUICK_GELU 1 /* x*sigmoid(1.702x), clip.py:67 */
#define SDAMD_ACT_SILU 2       /* diffusion.py:163 */
#define SDAMD_ACT_GEGLU 3      /* diffusion.py:128-130 fused into linear_geglu_1: W rows packed so that
                                * 16-row tile 2c = value channels [16c,16c+16), tile 2c+1 = their gates;
                                * C is [M][N/2] = value * gelu_erf(gate); bf16, N % 256 == 0, no residual */

const char* sdamd_last_error(void);
int sdamd_abi_version(void);
int sdamd_device_count(void); /* >=0, or -hipError when no driver/GPU */

/* ---- sampler: pipeline.py:147-153 (CFG combine) + sampler/ddpm.py:55-87 (step), one launch.
 * latents [B][n] f32 in/out; eps [2B][n] f32 = UNet output, [pos_0..pos_B-1, neg_0..neg_B-1]
 * (do_cfg=0: eps [B][n]); noise [steps][B][n] pre-drawn host noise (may be null when no row
 * has has_noise); coef [steps][8] = {sqrt(abar_t), sqrt(1-abar_t), c0, cx, sigma, has_noise,
 * cx_on_eps, cfg} computed on the host in the reference's op order:
 *   x0 = (x - sqrt(1-abar_t)*e) / sqrt(abar_t);  x' = x0*c0 + cx*(cx_on_eps ? e : x) [+ sigma*noise]
 * cx_on_eps = 0 is the DDPM step (ddpm.py:59-85), 1 the deterministic DDIM step
 * (sampler/ddim.py:50-72: c0 = sqrt(abar_prev), cx = sqrt(1-abar_prev), no noise).
 * The schedule index is *step_ptr (device int, for graph replay) or `step` when step_ptr is
 * null; coef (and noise) have n_steps rows: a host `step` outside [0, n_steps) is an error, a
 * device index outside it makes the launch a no-op (latents untouched) instead of reading past
 * the tables.  A null `noise` is read as "no noise" whatever has_noise says: the caller checks its
 * table (pipeline.DenoiseLoop does).  A NaN `cfg_scale` means "read the guidance scale from column 7
 * of this step's coef row": a captured hipGraph then follows a new scale written into the table.
 * Bit-exact vs torch CPU. */
int sdamd_cfg_ddpm_step(float* latents, const float* eps, const float* noise, const float* coef,
                        const int* step_ptr, int step, int n_steps, float cfg_scale, int do_cfg,
                        int B, int n, void* stream);
int sdamd_counter_add(int* counter, int delta, void* stream);
/* sampler/ddpm.py:91-111: out = sqrt_abar*x + sqrt_one_minus_abar*noise */
int sdamd_add_noise(float* out, const float* x, const float* noise, float sqrt_abar,
                    float sqrt_one_minus_abar, long long total, void* stream);
/* dst[0:width] = table[idx][0:width], idx = *step_ptr or step (per-step time-bias rows); table has
 * `rows` rows, an index outside them is an error (host step) or leaves dst untouched (device index) */
int sdamd_gather_row(float* dst, const float* table, const int* step_ptr, int step, int rows,
                     int width, void* stream);

/* ---- normalisation: nn.GroupNorm(32,C)(+F.silu) diffusion.py:26-28,38-40,93,292-294,
 * decoder.py:25-30,49,118-120; x,y NHWC [B][HW][C]; ws >= sdamd_groupnorm_ws_bytes(B).
 * x is `dtype`, y is `out_dtype` (f32->f32, bf16->bf16, or f32->bf16: an f32 residual trunk
 * normalised straight into a bf16 MFMA operand). */
size_t sdamd_groupnorm_ws_bytes(int B);
int sdamd_groupnorm_nhwc(const void* x, void* y, const float* gamma, const float* beta, void* ws,
                         int B, int HW, int C, float eps, int silu, int dtype, int out_dtype,
                         void* stream);
/* The same GroupNorm for callers that cannot allocate or carry a workspace (code under hipGraph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * The per-block partial statistics live in a device array built into the library (2 MiB per GPU,
 * held until the process ends) instead of `ws`; B <= sdamd_groupnorm_noalloc_max_batch() (64),
 * larger batches are rejected.  Calls no hipMalloc / hipFree / allocating routine.  All calls on one
 * GPU share the array: calls that overlap in time on one GPU (two streams) race.  The array's
 * address is looked up on the first call per device (hipGetSymbolAddress, not a stream operation);
 * sdamd_noalloc_init() does that lookup ahead of time for the current device. */
int sdamd_noalloc_init(void);
int sdamd_groupnorm_noalloc_max_batch(void);
int sdamd_groupnorm_nhwc_noalloc(const void* x, void* y, const float* gamma, const float* beta, int B,
                                 int HW, int C, float eps, int silu, int dtype, int out_dtype,
                                 void* stream);
/* GroupNorm(32, C1+C2) of torch.cat((x1, x2), dim=channel) without materialising the concat: the
 * skip connections of the UNet's up path, diffusion.py:279 followed by :26-28 of the next
 * UNET_ResidualBlock.  x1 [B][HW][C1], x2 [B][HW][C2], y [B][HW][C1+C2]. */
int sdamd_groupnorm_nhwc_cat(const void* x1, const void* x2, int C1, int C2, void* y,
                             const float* gamma, const float* beta, void* ws, int B, int HW,
                             float eps, int silu, int dtype, int out_dtype, void* stream);
/* GroupNorm whose statistics come from the GEMM / conv that PRODUCED x (sdamd_gemm / sdamd_conv3x3 `colstats`): the
 * pass that reads the whole tensor for its statistics is not run (diffusion.py:26-28,38-40,93: the normalised tensor is
 * the previous conv's / residual add's output).  stats1 [B*HW/64][C1][2] f32, stats2 likewise for the second source of
 * a virtual concat (x2, C2; null / 0 without one); HW % 64 == 0; ws as for sdamd_groupnorm_nhwc.  Small maps (one-launch
 * kernel) ignore the statistics. */
int sdamd_groupnorm_nhwc_pre(const void* x1, const void* x2, int C1, int C2, void* y, const float* gamma,
                             const float* beta, const float* stats1, const float* stats2, void* ws, int B,
                             int HW, float eps, int silu, int dtype, int out_dtype, void* stream);
/* nn.LayerNorm(C): diffusion.py:109,117,126; clip.py:54,63,95; dtypes as for groupnorm */
int sdamd_layernorm(const void* x, void* y, const float* gamma, const float* beta, int rows, int C,
                    float eps, int dtype, int out_dtype, void* stream);

/* ---- elementwise */
/* diffusion.py:128-130: in [rows][2F] (value|gate) -> out [rows][F] = value*gelu_erf(gate) */
int sdamd_geglu(const void* in, void* out, long long rows, int F, int dtype, void* stream);
/* clip.py:13-19: out[r] = table[ids[r]] + pos[r % S]; ids int64 device */
int sdamd_clip_embed(const long long* ids, const void* table, const void* pos, void* out, int rows,
                     int S, int D, int vocab, int dtype, void* stream);
/* diffusion.py:277 torch.cat(dim=1) in NHWC */
int sdamd_concat_nhwc(const void* a, const void* b, void* out, long long pixels, int C1, int C2,
                      int dtype, void* stream);
int sdamd_add(const void* a, const void* b, void* out, long long numel, int dtype, void* stream);
int sdamd_nchw_to_nhwc(const float* src, void* dst, int B, int C, int HW, int dtype, void* stream);
int sdamd_nhwc_to_nchw(const void* src, float* dst, int B, int C, int HW, int dtype, void* stream);
int sdamd_cast(const void* src, void* dst, long long n, int src_dtype, int dst_dtype, void* stream);
/* dst [rows][2K] bf16 = [bf16(x) | bf16(x - bf16(x))]: two-term bf16 split of an f32 GEMM operand.
 * Used with weights laid out [W | W] where the un-normalised f32 residual trunk itself is the
 * operand (1x1 skip convs diffusion.py:18,42; conv_output diffusion.py:85,140), so that bf16
 * MFMA does not round the trunk (no reference counterpart: the reference computes these in f32) */
int sdamd_split_hilo(const float* src, void* dst, long long rows, int K, void* stream);
/* same over the column concat [src1 (K1) | src2 (K2)] (skip concat diffusion.py:279 feeding the
 * 1x1 residual_layer diffusion.py:42): dst [rows][2*(K1+K2)] */
int sdamd_split_hilo_cat(const float* src1, const float* src2, void* dst, long long rows, int K1,
                         int K2, void* stream);
/* F.silu on an f32 vector (diffusion.py:32) */
int sdamd_silu(const float* src, float* dst, long long n, void* stream);
/* pipeline.py:158-160: rescale (-1,1)->(0,255), clamp, truncating cast; NCHW f32 -> NHWC u8 */
int sdamd_image_to_uint8(const float* src, unsigned char* dst, int B, int HW, void* stream);
/* pipeline.py:99-104: uint8 HWC -> NCHW f32 in (-1,1) */
int sdamd_uint8_to_image(const unsigned char* src, float* dst, int B, int HW, void* stream);
/* decoder.py:76,128: y = conv1x1_{4->4}(x/0.18215), NCHW f32; w [4][4], b [4] f32 */
int sdamd_vae_post_quant(const float* x, float* y, const float* w, const float* b, int B, int HW,
                         void* stream);
/* encoder.py:57,70-84: conv1x1_{8->8} + clamp/exp/sqrt/sample/scale; x NCHW f32 [B][8][HW] */
int sdamd_vae_sample(const float* x, const float* noise, float* out, const float* w, const float* b,
                     int B, int HW, void* stream);

/* Deterministic launch plans (no counterpart in the reference, whose pipeline is batch 1: pipeline.py:93; this is the
 * shard-invariance extension of SURVEY.md §4 item 5 / §8e).  n >= 1 = the number of independent samples the rows of the
 * following launches span (UNet batch, VAE micro-batch ...): tile, split-K factor, kernel family and GroupNorm partial
 * blocks are then planned from ONE sample's rows, so a sample's result is bit-identical whatever the batch it runs in
 * (the split-K workspace has to grow with n: pass n times the bytes a single sample's plan needs).  1 = plans follow the
 * batched problem (default, faster).  Process-wide; returns the previous value. */
int sdamd_set_plan_samples(int n);

/* ---- MFMA GEMM: nn.Linear / conv1x1 (attention.py:19-20,81-85; diffusion.py:18,54,82-85,
 * 156-157; clip.py:45-46).  C[m][n] = act(sum_k A[m][k]*W[n][k] + bias[n] + residual[m][n]);
 * K % 64 == 0, N % 16 == 0; `batch` independent problems with element strides.  C and the
 * residual share one type: f32 when out_f32 (or dtype f32), else bf16.  out_f32 == 2 (bf16
 * operands, act none, a shape with sdamd_gemm_tile != 0): f32 residual, and C is bf16 [M][2N]
 * (ldc >= 2N) = [bf16(v) | bf16(v - bf16(v))], the two-term form sdamd_split_hilo makes -- used
 * where the result is only ever read as the hi|lo operand of the next GEMM (linear_geglu_2 ->
 * conv_output, diffusion.py:132-140).  `ws` (may be null) is
 * caller-owned scratch of ws_bytes for split-K partial sums: small-M / large-K problems then
 * spread K over more workgroups and a reduce kernel applies bias/residual/act.
 * `colstats` (may be null): the epilogue also leaves, per 64-row block rb and output column n,
 * colstats[(rb*N + n)*2 + {0,1}] = sum and sum of squares of the stored C[rb*64..rb*64+63][n] (f32,
 * ceil(M/64)*N*2 floats) for a GroupNorm that normalises C (sdamd_groupnorm_nhwc_pre).  Only un-split, un-batched
 * launches on the LDS-DMA kernels with f32 or operand-type output produce them: *colstats_written (host int, may
 * be null) says whether this call did.
 * `w_planes` (may be null; f32 operands in the bf16x3 mode only): W split ahead of time by sdamd_split3_weights
 * (packed per batch entry); the kernels that take it then split only the activation fragments in registers. */
int sdamd_gemm(const void* A, const void* W, void* C, const float* bias, const void* residual, int M,
               int N, int K, int lda, int ldw, int ldc, int ldr, int batch, long long strideA,
               long long strideW, long long strideC, long long strideR, int act, int dtype,
               int out_f32, void* ws, size_t ws_bytes, float* colstats, int* colstats_written,
               const void* w_planes, void* stream);
/* diffusion.py:99-100, 110-117, 119-126 (conv_input -> layernorm_1; attention out_proj + residue -> layernorm_2 / _3):
 * out [M][N] f32 = A [M][K] W[N][K]^T + bias (+ residual [M][N] f32, may be null), and from the same launch
 * ln_out [M][N] bf16 = LayerNorm over each out row (biased variance, eps inside the sqrt) * ln_gamma + ln_beta, the
 * bf16 operand of the next projection.  bf16 A / W, dense (ldw = K, ldc = ldr = N), N == 320: the 128x320 tile holds
 * whole rows, so the f32 trunk is not read again by a LayerNorm launch.  Same arithmetic as sdamd_gemm followed by
 * sdamd_layernorm (f32 in, bf16 out). */
int sdamd_gemm_ln(const void* A, const void* W, float* out, const float* bias, const float* residual,
                  void* ln_out, const float* ln_gamma, const float* ln_beta, float ln_eps, int M, int N,
                  int K, int lda, void* stream);
/* The weights of an f32 Linear / Conv2d (rows x K, K % 32 == 0, row stride ldw), split once into the three bf16
 * terms of the bf16x3 mode (sdamd_set_f32_mode): planes [rows][K/32][hi | mid | lo][32] bf16 = 6 bytes per weight,
 * the 32 k of a K-tile in the order the kernels' MFMA fragments take them.  No reference counterpart (a weight
 * preparation step like the reference's .half() of its parameters, pipeline.py:44-49). */
int sdamd_split3_weights(const float* w, void* planes, long long rows, int K, int ldw, void* stream);
/* block tile the GEMM/conv launcher picks for a problem (BM*1000+BN + 1000000*(splits-1)); host-only helper
 * used by bench.py to attribute launches to kernel instantiations (no reference counterpart) */
int sdamd_gemm_tile(int M, int N, int K, int batch, int dtype, size_t ws_bytes, int act);
/* host-only, as sdamd_gemm_tile: the image width (64 / 32 / 16) when sdamd_conv3x3 runs this problem on a halo-staged
 * kernel (stride 1, pad 1, bf16, whole image rows per 128-pixel tile: the resblock convs of diffusion.py:9,13), plus 256
 * when that is the 256-row one-wave-per-SIMD kernel (Cin >= 1280, or >= 640 on 16-wide images; + 512 more for its phase
 * form, upsample == 2), else 0 */
int sdamd_conv3x3_halo(int B, int H, int Wd, int Cin, int Cout, int stride, int pad_mode, int upsample, int dtype,
                       size_t ws_bytes);
/* 1 when sdamd_gemm can apply `act` in its epilogue for this dtype / N (GEGLU: see above) */
int sdamd_gemm_supports_act(int act, int dtype, int N);
/* kernel family switch for tests/benchmarks: 0 = auto (8-wave LDS-DMA bf16 kernels where they fit,
 * else the 4-wave register-staged ones), 1 = 4-wave kernels only, 2 = 8-wave kernels whenever
 * one of their tiles divides N, 3 = as 2 with a 3-way K split (small-size parity tests);
 * returns the previous value */
int sdamd_set_gemm_impl(int impl);
/* How f32 GEMM / conv3x3 operands (dtype SDAMD_F32) reach the matrix cores; returns the previous mode, any other
 * argument only queries.  No reference counterpart: torch's fp32 nn.Linear / nn.Conv2d (diffusion.py, decoder.py) is
 * the function both modes compute.
 *   1 (default) "bf16x3": every f32 operand value is split exactly into three bf16 terms in registers and a product is
 *      accumulated in f32 from the six partial products >= 2^-16 of it on v_mfma_f32_16x16x32_bf16 (GEMM, conv3x3, attention with d <= 160); the dropped terms
 *      are below one f32 rounding of the product.  Measured against float64 the result is as close as mode 0's
 *      (tests/test_hip_ops.py::test_f32_modes_against_float64), at 1.3-1.5x its speed.  inf inputs give NaN.
 *   0 "exact": v_mfma_f32_16x16x4_f32, bitwise an fmaf chain over k.
 * Process-wide and read at launch time: set it before launching from several host threads, not while they launch
 * (like sdamd_set_gemm_impl it is an unsynchronised global). */
int sdamd_set_f32_mode(int mode);
/* ---- implicit-GEMM conv3x3 over NHWC: nn.Conv2d(k=3) of diffusion.py:9,13,146,204-221,
 * decoder.py:10-13,92-111, encoder.py:13-45.  w [Cout][ky][kx][Cin]; pad_mode 0 = pad 1,
 * 1 = bottom/right pad only (encoder.py:65-66); upsample=1 fuses F.interpolate(nearest,2x)
 * (diffusion.py:148-150, decoder.py:90) into the gather; upsample=2 computes the same result as
 * four 2x2 convs, one per output phase (py,px), at 4/9 of the MACs: w is then
 * [4][Cout][2][2][Cin] with the 3x3 taps that read the same input pixel summed (bf16 only);
 * residual [B][Ho][Wo][Cout] or null.  colstats / colstats_written: as for sdamd_gemm, with
 * M = B*Ho*Wo rows and N = Cout columns (never for the phase form). */
int sdamd_conv3x3(const void* x, const void* w, void* y, const float* bias, const void* residual,
                  int B, int H, int W, int Cin, int Cout, int stride, int pad_mode, int upsample,
                  int act, int dtype, int out_f32, void* ws, size_t ws_bytes, float* colstats,
                  int* colstats_written, const void* w_planes, void* stream);
/* first convs (diffusion.py:196, decoder.py:78, encoder.py:10): x NCHW f32 [B], Cin 3|4,
 * w [ky][kx][Cin][Cout] f32 -> y NHWC dtype [copies*B]: sample n is written `copies` times, at
 * n, n+B, ... (copies = 2: the UNet input under classifier-free guidance is the same latents
 * twice, pipeline.py:147 `latents.repeat(2,1,1,1)`, so the conv runs once) */
int sdamd_conv3x3_cin_small(const float* x_nchw, const float* w_tap_ci_co, const float* bias,
                            void* y_nhwc, int B, int H, int W, int Cin, int Cout, int copies,
                            int dtype, void* stream);
/* last convs (diffusion.py:287, decoder.py:123, encoder.py:54): x NHWC dtype, Cout 3|4|8,
 * w [tap][Cin/V][Cout][V] dtype (V = 16 bytes of dtype) -> y NCHW f32 */
int sdamd_conv3x3_cout_small(const void* x_nhwc, const void* w_packed, const float* bias,
                             float* y_nchw, int B, int H, int W, int Cin, int Cout, int dtype,
                             void* stream);

/* ---- MFMA flash attention: attention.py:47-68 (self, optional causal) and :112-127 (cross);
 * replaces customtriton.py / flash_attention.py outright.  q/k/v/out are [B][S][row stride]
 * with head h at columns [h*d,(h+1)*d); strides in elements; d in {40,64,80,160} (UNet / CLIP
 * heads) or 512 (the VAE's single head, decoder.py:39-70: the head dim is split over the
 * workgroup's waves, non-causal only).  No [S,S] tensor is materialised for any d. */
int sdamd_attention(const void* q, const void* k, const void* v, void* out, int B, int H, int Sq,
                    int Skv, int d, long long q_bs, int q_rs, long long k_bs, int k_rs,
                    long long v_bs, int v_rs, long long o_bs, int o_rs, int causal, int dtype,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif
