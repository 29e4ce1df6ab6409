// MX block-scaled fp8 (OCP e4m3 elements, one e8m0 scale per 32 K elements): scale layouts and conversions
// shared by the fp8 GEMM (gemm_fp8.hip) and the epilogues that produce its operands (gemm_core.h).
#pragma once
#include "common.h"

namespace me {

// ---- scale layouts (bytes) ------------------------------------------------------------------------------
// activation operand, rows m (padded to 128-row wave tiles, `mt` of them), K block kb:
__host__ __device__ __forceinline__ int64_t a_scale_index(int64_t m, int kb, int64_t mt) {
    return (((int64_t)(kb >> 2) * mt + (m >> 7)) * 64 + (kb & 3) * 16 + (m & 15)) * 8 + ((m & 127) >> 4);
}
// weight operand, rows n (N a multiple of 64, `nt` = N / 64 wave tiles), K block kb:
__host__ __device__ __forceinline__ int64_t w_scale_index(int64_t n, int kb, int64_t nt) {
    return (((int64_t)(kb >> 2) * nt + (n >> 6)) * 64 + (kb & 3) * 16 + (n & 15)) * 4 + ((n & 63) >> 4);
}

// e8m0 scale of a block with largest magnitude amax: the element grid is e4m3 (largest normal 448 = 1.75 * 2^8),
// so the block exponent is floor(log2 amax) - 8, one more when amax's significand is >= 1.75 -- then no scaled
// element exceeds 448 and the conversion needs no saturation.  amax = 0 gives byte 0 (2^-127): elements are 0.
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
    const unsigned e = ((__float_as_uint(amax) & 0x7fffffffu) + 0x200000u) >> 23;
    return e > 8u ? (e - 8u > 254u ? 254u : e - 8u) : 0u;
}
// 1 / 2^(byte - 127) as a float (byte in [0, 254]); 2^127 needs the two-step form
__device__ __forceinline__ float mx_inv_scale(unsigned byte) {
    return byte == 0 ? 1.7014118e38f : __uint_as_float((254u - byte) << 23);
}
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

// the fp8 + scales form of an epilogue granule (fc1: bias + GELU, then quantised as the next GEMM's activation
// operand): 8 consecutive columns of one row per lane, 4 adjacent lanes share a 32-column MX block.  Returns the 8
// bytes and stores the block's scale.
// row_ok: row m exists (the tall tile's last row tile of a segment computes rows beyond it: nothing of them is stored)
__device__ __forceinline__ uint2 quantise_granule_fp8(const GemmParams& p, int m, int n, const float (&a)[8], bool row_ok = true) {
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(a[e]));
    // lanes l ^ 1 and l ^ 2 by quad-permute DPP (a register move) instead of two ds_bpermute round trips
    amax = fmaxf(amax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(amax), 0xB1, 0xF, 0xF, true)));
    amax = fmaxf(amax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(amax), 0x4E, 0xF, 0xF, true)));
    const unsigned sb = mx_scale_byte(amax);
    const float inv = mx_inv_scale(sb);
    uint2 v;
    v.x = pack_fp8x4(a[0] * inv, a[1] * inv, a[2] * inv, a[3] * inv);
    v.y = pack_fp8x4(a[4] * inv, a[5] * inv, a[6] * inv, a[7] * inv);
    if ((n & 31) == 0 && row_ok) p.out8_scale[a_scale_index(m, n >> 5, p.out8_mt)] = (uint8_t)sb;
    return v;
}
// Two rows' granules leave as ONE 16-byte store per lane (8-byte stores run at less than half the rate, gemm_core.h):
// lanes l and l ^ 1 hold adjacent granules (columns n and n +- 8) of the same rows m0 and m1; the even lane stores
// both granules of row m0, the odd lane both of row m1.
__device__ __forceinline__ void store_granule_pair_fp8(const GemmParams& p, int m0, int m1, int n, uint2 q0, uint2 q1, int m_end) {
    const bool odd = (n & 8) != 0;
    const uint2 send = odd ? q0 : q1;
    uint2 recv;
    recv.x = __builtin_amdgcn_update_dpp(0, send.x, 0xB1, 0xF, 0xF, true);
    recv.y = __builtin_amdgcn_update_dpp(0, send.y, 0xB1, 0xF, 0xF, true);
    const uint4 o = odd ? make_uint4(recv.x, recv.y, q1.x, q1.y) : make_uint4(q0.x, q0.y, recv.x, recv.y);
    const int m = odd ? m1 : m0;
    if (m < m_end) *reinterpret_cast<uint4*>(p.out8 + (int64_t)m * p.ldc + (n & ~15)) = o;
}

}  // namespace me
