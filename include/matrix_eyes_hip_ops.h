/*
 * matrix_eyes_hip_ops.h — kernel-level entry points of libmatrixeyes_hip.so.
 *
 * Not part of the drop-in boundary (matrix_eyes_hip.h is): these expose the individual HIP
 * kernels so that parity tests can check each one against the CPU oracle and bench.py can time
 * the dominant kernel against its roofline.  All pointers are DEVICE pointers; 16-bit operands
 * are in the context's dtype (ME_DTYPE_F16 / ME_DTYPE_BF16); work is enqueued on the context's
 * stream and not synchronised.
 */
#ifndef MATRIX_EYES_HIP_OPS_H
#define MATRIX_EYES_HIP_OPS_H

#include "matrix_eyes_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ME_ACT_NONE = 0, ME_ACT_GELU = 1, ME_ACT_RELU = 2 };

/* Linear (vit.rs:60-62,74,120,122): out = act(A[M][K] . W[N][K]^T + bias); out16 and/or out32
   [M][N].  tile_cfg -1 = automatic. */
int32_t me_op_linear(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, const void* W16,
                     const float* bias, void* out16, float* out32, int32_t act, int32_t tile_cfg);
/* Block residual update (vit.rs:165-169): x[M][N] += gamma[n] * (A . W^T + bias)  in place. */
int32_t me_op_linear_residual(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16,
                              const void* W16, const float* bias, const float* gamma, float* x32,
                              int32_t tile_cfg);
/* Attention (vit.rs:58-75): qkv16 [windows*tokens][3*heads*64] -> out16 [windows*tokens][heads*64] */
int32_t me_op_attention(me_ctx* ctx, const void* qkv16, void* out16, int32_t windows, int32_t tokens,
                        int32_t heads);
/* The form the forward pass runs (pipeline.hip): the qkv linear writes its first `qcols` output columns (Q) multiplied
   by `qscale` = 1/sqrt(64) * log2(e) -- out16 = round16((A . W^T + bias) * qscale) there, one rounding -- and the
   attention kernel takes that Q as it is (its softmax runs on exp2 with the reference point inside the MFMA
   accumulator).  me_op_attention scales a plain Q itself, with a second rounding. */
int32_t me_op_linear_scaled_cols(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, const void* W16,
                                 const float* bias, void* out16, int32_t qcols, float qscale, int32_t tile_cfg);
int32_t me_op_attention_prescaled(me_ctx* ctx, const void* qkv16, void* out16, int32_t windows, int32_t tokens,
                                  int32_t heads);
/* LayerNorm (vit.rs:165,168,343): x32 [rows][dim] -> y16 and/or y32 */
int32_t me_op_layernorm(me_ctx* ctx, const float* x32, const float* weight, const float* bias,
                        void* y16, float* y32, int64_t rows, int32_t dim, float eps);
/* Conv2d k x k (k in {1,3}, padding (k-1)/2, stride in {1,2}) as implicit GEMM.
   in16b: zero-bordered NHWC [B][H+2][W+2][Cin]; w16: packed [Cout][k*k][Cin];
   out32 [B*Ho*Wo][Cout] and/or out16 (zero-bordered [B][Ho+2][Wo+2][Cout] when border16);
   res32/res32b optional f32 residuals [B*Ho*Wo][Cout]; act applies to out16 (and to out32 when
   act_both). */
int32_t me_op_conv2d(me_ctx* ctx, const void* in16b, int32_t B, int32_t H, int32_t W, int32_t Cin,
                     const void* w16, int32_t Cout, int32_t k, int32_t stride, const float* bias,
                     const float* res32, const float* res32b, float* out32, void* out16,
                     int32_t border16, int32_t act, int32_t act_both, int32_t tile_cfg);
/* The depth head's last layers (mod.rs:83-94,329-362): relu(conv3x3(in, w16 [Cmid][9][Cin]) + bias) . w2 + b2, ReLU,
   / f_norm[b] (null: not divided), clamp -> out32 [B][H][W].  in16b: zero-bordered NHWC.  tile_cfg -1: the halo kernel
   (csrc/head_conv.hip) where the shape is the model's (Cin 128, Cmid 32, H % 12 == 0, W % 16 == 0), the implicit-GEMM
   tile elsewhere; >= 0: the implicit-GEMM tile. */
int32_t me_op_head_final(me_ctx* ctx, const void* in16b, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w16,
                         int32_t Cmid, const float* bias, const float* w2, const float* b2, const float* f_norm,
                         float clamp_lo, float clamp_hi, float* out32, int32_t tile_cfg);
/* ConvTranspose2d(2,2,stride 2): in16 NHWC [B*H*W][Cin]; w16 packed [(dy*2+dx)*Cout + co][Cin];
   out32 [B][2H][2W][Cout] and/or out16 (zero-bordered when border16). */
int32_t me_op_conv_transpose2x2(me_ctx* ctx, const void* in16, int32_t B, int32_t H, int32_t W,
                                int32_t Cin, const void* w16, int32_t Cout, const float* bias,
                                float* out32, void* out16, int32_t border16, int32_t tile_cfg);
/* MX block-scaled fp8 (ME_DTYPE_FP8, BASELINE configs[3]; csrc/gemm_fp8.hip, mx_fp8.h).
   me_op_quantize_fp8: f16 [rows][K] -> e4m3 bytes dst8 [rows][K] + one e8m0 scale per 32 K elements into
   `scales`, in the weight operand's layout (weight_layout = 1: rows a multiple of 64, rows*K/32 bytes) or the
   activation operand's (0: ceil(rows/128)*128*K/32 bytes).  me_op_scale_index gives the byte position of
   (row, K block) in either layout, so that a test can read the scales back.
   me_op_layernorm_fp8: LayerNorm with the result quantised as an activation operand.
   me_op_linear_fp8: out = act(A8 . W8^T + bias) with M, N multiples of 256 and K a multiple of 128 (>= 256), as
   ONE of: out16 (f16 [M][N]); out8 + out8_scale (GELU'd, quantised as the next GEMM's activation operand);
   x32 (+ gamma): the residual update x += gamma * (A . W^T + bias). */
int32_t me_op_quantize_fp8(me_ctx* ctx, const void* src16, int64_t rows, int32_t K, int32_t weight_layout,
                           uint8_t* dst8, uint8_t* scales);
int64_t me_op_scale_index(int64_t row, int32_t kblock, int64_t rows, int32_t weight_layout);
int32_t me_op_layernorm_fp8(me_ctx* ctx, const float* x32, const float* weight, const float* bias, uint8_t* y8,
                            uint8_t* yscale, int64_t rows, int32_t dim, float eps);
int32_t me_op_linear_fp8(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const uint8_t* A8, const uint8_t* a_scale,
                         const uint8_t* W8, const uint8_t* w_scale, const float* bias, void* out16, uint8_t* out8,
                         uint8_t* out8_scale, const float* gamma, float* x32);
/* me_op_linear / me_op_linear_residual over up to three row segments with their own weights, as the encoder's merged
   ViT launches run them (pipeline.hip MergedVit): rows [0, seg1) use W16[0] / bias[0] (/ gamma[0]), [seg1, seg2) the [1]
   set, [seg2, M) the [2] set; seg2 == 0: two segments, seg1 == 0: one.  x32 given: the residual form (gamma taken);
   otherwise out16 = act(A . W^T + bias).  Segment boundaries must be multiples of the tile height of tile_cfg, except
   for the 352-row tile (tile_cfg 10), which lays its row tiles out per segment. */
int32_t me_op_linear_segments(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, int32_t seg1, int32_t seg2,
                              const void* const W16[3], const float* const bias[3], const float* const gamma[3],
                              void* out16, float* x32, int32_t act, int32_t tile_cfg);
/* The residual update of me_op_linear_segments with the LayerNorm of the next sublayer in the same launch (the form
   the forward pass runs, vit.rs:165-169; csrc/gemm_core.h resid_ln_epilogue): x32[M][N] += gamma * (A W^T + bias) in
   place and xn16[m][:] = LayerNorm(x32[m][:], eps) * ln_w + ln_b with the weights of the row's segment.  N in {256,
   512, 1024}; the 352-row tile, whose N / 256 column tiles exchange their partial statistics through memory. */
int32_t me_op_linear_residual_layernorm(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, int32_t seg1,
                                        int32_t seg2, const void* const W16[3], const float* const bias[3],
                                        const float* const gamma[3], const float* const ln_w[3], const float* const ln_b[3],
                                        float eps, float* x32, void* xn16);
/* The same launch with the normalised rows written as the next GEMM's MX fp8 operand instead of 16-bit: xn8 [M][N] e4m3
   bytes and xn_scale, one e8m0 byte per 32 columns in the activation layout of me_op_layernorm_fp8 (ceil(M / 128) tiles
   of 128 rows) -- what an ME_DTYPE_FP8 context's 16-bit projection hands to fc1. */
int32_t me_op_linear_residual_layernorm_fp8(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const void* A16, int32_t seg1,
                                            int32_t seg2, const void* const W16[3], const float* const bias[3],
                                            const float* const gamma[3], const float* const ln_w[3], const float* const ln_b[3],
                                            float eps, float* x32, uint8_t* xn8, uint8_t* xn_scale);
/* me_op_linear_fp8 over up to three row segments with their own weights, as the encoder's merged ViT launches run
   it (pipeline.hip MergedVit): rows [0, seg1) use W8[0] / w_scale[0] / bias[0] (/ gamma[0]), [seg1, seg2) the [1]
   set, [seg2, M) the [2] set; seg1, seg2 multiples of 256, seg2 == 0: two segments, seg1 == 0: one. */
int32_t me_op_linear_fp8_segments(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const uint8_t* A8, const uint8_t* a_scale,
                                  int32_t seg1, int32_t seg2, const uint8_t* const W8[3], const uint8_t* const w_scale[3],
                                  const float* const bias[3], const float* const gamma[3], void* out16, uint8_t* out8,
                                  uint8_t* out8_scale, float* x32);
/* The fp8 GEMM's residual form with the LayerNorm of the rows it updates (csrc/gemm_fp8.hip gemm_pp8t_kernel, the 352-row
   tile): x32[M][N] += gamma * (A8 W8^T + bias) in place and LayerNorm(x32[m][:], eps) * ln_w + ln_b of the row's segment as
   the next GEMM's MX fp8 operand (xn8 / xn_scale as in me_op_linear_residual_layernorm_fp8) -- what an ME_DTYPE_FP8
   context's fc2 hands to the next block's qkv.  Operands as me_op_linear_fp8_segments; N in {256, 512, 1024}. */
int32_t me_op_linear_fp8_residual_layernorm(me_ctx* ctx, int32_t M, int32_t N, int32_t K, const uint8_t* A8, const uint8_t* a_scale,
                                            int32_t seg1, int32_t seg2, const uint8_t* const W8[3], const uint8_t* const w_scale[3],
                                            const float* const bias[3], const float* const gamma[3], const float* const ln_w[3],
                                            const float* const ln_b[3], float eps, float* x32, uint8_t* xn8, uint8_t* xn_scale);
/* me_op_attention with the output written as an MX fp8 activation operand (out8 [windows*tokens][heads*64] bytes +
   block scales; heads even): the bytes me_op_quantize_fp8 gives for me_op_attention's 16-bit output. */
int32_t me_op_attention_fp8(me_ctx* ctx, const void* qkv16, uint8_t* out8, uint8_t* out8_scale, int32_t windows,
                            int32_t tokens, int32_t heads);
/* The number formatter of the device OBJ writer (csrc/ryu_f64.h, obj_format.hip) on its own: values[i] (DEVICE f64)
   printed as Rust's `{}` prints an f64 -- shortest round-trip digits, positional notation -- into the `stride`-byte
   slot i of `text` (stride >= 344), its length into lengths[i]. */
int32_t me_op_format_f64(me_ctx* ctx, const double* values, int64_t count, char* text, int32_t stride, int32_t* lengths);
/* Box calibration (csrc/calibrate.hip; bench.py's `calibration` object): two FIXED loops on the context's stream, about
   50 ms, synchronous.  out[0] = TFLOP/s of an MFMA-only loop (v_mfma_f32_16x16x32_f16, operands in registers, two waves per
   SIMD on 256 workgroups), out[1] = the shader clock the part held inside it (GHz, s_memtime / s_memrealtime),
   out[2] = GB/s (read + written) of a 512 MiB device copy, out[3] = the loop's ms, out[4] = ms per copy, out[5] = CUs. */
int32_t me_calibrate(me_ctx* ctx, double* out6);
/* f32 <-> context 16-bit type */
int32_t me_op_cast_to16(me_ctx* ctx, const float* src, void* dst16, int64_t count);
int32_t me_op_cast_to32(me_ctx* ctx, const void* src16, float* dst, int64_t count);
/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg): enable(1)
   clears and starts recording, report() synchronises and writes a JSON array of
   {"kernel", "launches", "total_ms", "flops", "bytes"} (algorithmic work, summed over launches). */
int32_t me_profile_enable(me_ctx* ctx, int32_t on);
int32_t me_profile_report(me_ctx* ctx, char* json, int64_t capacity);
/* Names of the GEMM tile configurations (for reports). */
int32_t me_op_gemm_config_count(void);
const char* me_op_gemm_config_name(int32_t cfg);

#ifdef __cplusplus
}
#endif
#endif /* MATRIX_EYES_HIP_OPS_H */
