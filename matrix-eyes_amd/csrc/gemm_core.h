// MFMA GEMM / implicit-GEMM convolution core for gfx950.
//
//   D[m][n] = sum_k A(m,k) * W[n][k]      A, W: f16 or bf16;  accumulate f32
//
// Structure (cdna_hip_programming.md §5): BK = 64 K-slab per step, operands staged
// global -> LDS with 16-byte LDS-DMA (`global_load_lds_dwordx4`), two LDS buffers so the load
// of slab t+1 overlaps the MFMAs of slab t, `v_mfma_f32_16x16x32_{f16,bf16}`.
//
// LDS image: each operand tile is [rows][64] 16-bit = 128-byte rows; the 16-byte chunk c of
// row r is kept at slot c ^ ((r >> 1) & 7), which makes every ds_read_b128 fragment read
// conflict-free (16 distinct 16-byte slots per lane group).  LDS-DMA writes linearly
// (base + lane*16), so the permutation is applied to the per-lane *source* address and again
// on the fragment read (rule 21 of the guide).
//
// The weight tile is the MFMA "A" operand (rows -> n) and the activation tile the "B" operand
// (cols -> m): every lane then owns 4 consecutive n of one output row, so epilogues store
// 16 B (f32) / 8 B (16-bit) per lane.
#pragma once
#include "common.h"

namespace me {

template <typename T>
struct MfmaOp;
template <>
struct MfmaOp<f16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct MfmaOp<bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Phi(x) * x with the exact-erf GELU the reference uses (burn activation::gelu, vit.rs:121).
// erfc via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), evaluated on |x| so the negative
// tail keeps its relative accuracy.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    poly *= t;
    const float e = poly * __expf(-z * z);  // erfc(z)
    const float phi = x < 0.0f ? 0.5f * e : 1.0f - 0.5f * e;
    return x * phi;
}

template <typename T>
__device__ __forceinline__ void store4_16(void* dst, float a, float b, float c, float d) {
    typedef T v4 __attribute__((ext_vector_type(4)));
    v4 v;
    v[0] = (T)a;
    v[1] = (T)b;
    v[2] = (T)c;
    v[3] = (T)d;
    *reinterpret_cast<v4*>(dst) = v;
}

template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_ITERS = (BM / 8) / NW, B_ITERS = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split over waves");
    static_assert(A_ITERS >= 1 && B_ITERS >= 1, "tile too small for the wave count");
    typedef typename MfmaOp<T>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- block -> tile, XCD-aware (blocks b and b+8 share an XCD's L2; give each XCD a
    // contiguous run of tiles, n fastest, so the A panel of a tile row is reused from L2) ----
    const int nbn = (p.N + BN - 1) / BN;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (wgid / nbn) * BM;
    const int n0 = (wgid % nbn) * BN;

    // ---- per-lane source pointers for the staging loads ----
    const int srow = lane >> 3;  // row within an 8-row LDS-DMA piece
    const int sslot = lane & 7;  // 16-byte slot written by this lane
    const char* a_src[A_ITERS];
    const char* w_src[B_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int chunk = sslot ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        if constexpr (AMODE == A_PLAIN) {
            a_src[i] = (const char*)p.A + ((int64_t)gm * p.lda) * 2 + chunk * 16;
        } else {
            const int ppi = p.out_H * p.out_W;
            const int b = gm / ppi;
            const int rem = gm - b * ppi;
            const int y = rem / p.out_W;
            const int x = rem - y * p.out_W;
            const int64_t pix = ((int64_t)b * p.in_Hp + y * p.stride) * p.in_Wp + x * p.stride;
            a_src[i] = (const char*)p.A + pix * p.Cin * 2 + chunk * 16;
        }
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int chunk = sslot ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        w_src[i] = (const char*)p.W + ((int64_t)gn * p.K) * 2 + chunk * 16;
    }

    const int nk = p.K / 64;
    // A_CONV: K index = tap * Cin + cin; a 64-wide slab never straddles a tap (Cin % 64 == 0)
    const int cin_steps = (AMODE == A_CONV) ? p.Cin / 64 : 1;
    const int pad = (AMODE == A_CONV) ? (p.KH - 1) / 2 : 0;
    int tap_kc = 0, tap_ky = 0, tap_kx = 0;  // scalar state for the slab being staged

    auto stage = [&](int kt, int buf) {
        char* la = smem + buf * STAGE_BYTES;
        char* lb = la + A_BYTES;
        int64_t a_koff;
        if constexpr (AMODE == A_PLAIN) {
            a_koff = (int64_t)kt * 128;
        } else {
            a_koff = ((int64_t)(tap_ky + 1 - pad) * p.in_Wp + (tap_kx + 1 - pad)) * p.Cin * 2 +
                     tap_kc * 128;
            if (++tap_kc == cin_steps) {
                tap_kc = 0;
                if (++tap_kx == p.KW) {
                    tap_kx = 0;
                    ++tap_ky;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) glds16(a_src[i] + a_koff, la + (i * NW + wave) * 1024);
        const int64_t w_koff = (int64_t)kt * 128;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) glds16(w_src[i] + w_koff, lb + (i * NW + wave) * 1024);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read addressing (byte offsets inside a stage)
    const int frow = lane & 15;
    const int fswz = frow >> 1;
    const int fslot0 = ((lane >> 4) ^ fswz) * 16;      // k-substep 0
    const int fslot1 = (((lane >> 4) + 4) ^ fswz) * 16;  // k-substep 1
    const int a_rd = (wm * TM + frow) * 128;
    const int b_rd = A_BYTES + (wn * TN + frow) * 128;

    stage(0, 0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = kk == 0 ? fslot0 : fslot1;
            frag af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const frag*>(sb + a_rd + i * 2048 + slot);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                wf[j] = *reinterpret_cast<const frag*>(sb + b_rd + j * 2048 + slot);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = MfmaOp<T>::run(wf[j], af[i], acc[i][j]);
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    const int ncol = (lane >> 4) * 4;  // first of this lane's 4 consecutive n within a 16-tile
    if constexpr (EPI == EPI_HEAD_FINAL) {
        // N <= 32 = the whole tile width (WN == 1): relu(acc + bias) . w2, reduced over n.
        float w2v[NI][4], bv[NI][4];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = j * 16 + ncol + r;
                const bool ok = n < p.N;
                w2v[j][r] = ok ? p.w2[n] : 0.f;
                bv[j][r] = ok ? p.bias[n] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += fmaxf(acc[i][j][r] + bv[j][r], 0.f) * w2v[j][r];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const int m = m0 + wm * TM + i * 16 + frow;
            if (lane < 16 && m < p.M) {
                float v = fmaxf(s + p.b2[0], 0.f);
                if (p.f_norm) v = v / p.f_norm[m / p.pixels_per_image];
                v = fminf(fmaxf(v, p.clamp_lo), p.clamp_hi);
                p.out32[m] = v;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * TM + i * 16 + frow;
            if (m >= p.M) continue;
            int64_t row32 = 0, row16 = 0;  // element offsets of this output row
            [[maybe_unused]] int pe_patch = 0;
            [[maybe_unused]] int ct_b = 0, ct_y = 0, ct_x = 0;
            if constexpr (EPI == EPI_RESID_SCALE) {
                row32 = (int64_t)m * p.ldc;
                row16 = row32;
            } else if constexpr (EPI == EPI_STORE) {
                row32 = (int64_t)m * p.ldc;
                if (p.out16_border) {  // rows are pixels of [B][out_H][out_W]
                    const int ppi = p.out_H * p.out_W;
                    const int b = m / ppi;
                    const int rem = m - b * ppi;
                    const int y = rem / p.out_W;
                    const int x = rem - y * p.out_W;
                    row16 = (((int64_t)b * (p.out_H + 2) + y + 1) * (p.out_W + 2) + x + 1) * p.ldc;
                } else {
                    row16 = row32;
                }
            } else if constexpr (EPI == EPI_PATCH_EMBED) {
                const int w = m / p.tokens_per_window;
                pe_patch = m - w * p.tokens_per_window;
                row32 = ((int64_t)w * (p.tokens_per_window + 1) + 1 + pe_patch) * p.ldc;
            } else if constexpr (EPI == EPI_CONVT) {
                const int ppi = p.out_H * p.out_W;  // input pixels per image
                ct_b = m / ppi;
                const int rem = m - ct_b * ppi;
                ct_y = rem / p.out_W;
                ct_x = rem - ct_y * p.out_W;
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn * TN + j * 16 + ncol;
                if (n >= p.N) continue;
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if constexpr (EPI == EPI_STORE) {
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
                        v0 += b4.x, v1 += b4.y, v2 += b4.z, v3 += b4.w;
                    }
                    if (p.res32) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.res32 + row32 + n);
                        v0 += r4.x, v1 += r4.y, v2 += r4.z, v3 += r4.w;
                    }
                    if (p.res32b) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.res32b + row32 + n);
                        v0 += r4.x, v1 += r4.y, v2 += r4.z, v3 += r4.w;
                    }
                    float a0 = v0, a1 = v1, a2 = v2, a3 = v3;
                    if (p.act == ACT_GELU) {
                        a0 = gelu_erf(v0), a1 = gelu_erf(v1), a2 = gelu_erf(v2), a3 = gelu_erf(v3);
                    } else if (p.act == ACT_RELU) {
                        a0 = fmaxf(v0, 0.f), a1 = fmaxf(v1, 0.f), a2 = fmaxf(v2, 0.f),
                        a3 = fmaxf(v3, 0.f);
                    }
                    if (p.out32) {
                        if (p.act16_only)
                            *reinterpret_cast<float4*>(p.out32 + row32 + n) =
                                make_float4(v0, v1, v2, v3);
                        else
                            *reinterpret_cast<float4*>(p.out32 + row32 + n) =
                                make_float4(a0, a1, a2, a3);
                    }
                    if (p.out16) store4_16<T>((T*)p.out16 + row16 + n, a0, a1, a2, a3);
                } else if constexpr (EPI == EPI_RESID_SCALE) {
                    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
                    const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + n);
                    const float4 r4 = *reinterpret_cast<const float4*>(p.res32 + row32 + n);
                    // same operation order as the reference: (xs * gamma) + residual
                    v0 = (v0 + b4.x) * g4.x + r4.x;
                    v1 = (v1 + b4.y) * g4.y + r4.y;
                    v2 = (v2 + b4.z) * g4.z + r4.z;
                    v3 = (v3 + b4.w) * g4.w + r4.w;
                    *reinterpret_cast<float4*>(p.out32 + row32 + n) = make_float4(v0, v1, v2, v3);
                } else if constexpr (EPI == EPI_PATCH_EMBED) {
                    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
                    const float4 e4 = *reinterpret_cast<const float4*>(
                        p.pos + (int64_t)(1 + pe_patch) * p.N + n);
                    *reinterpret_cast<float4*>(p.out32 + row32 + n) =
                        make_float4(v0 + b4.x + e4.x, v1 + b4.y + e4.y, v2 + b4.z + e4.z,
                                    v3 + b4.w + e4.w);
                } else if constexpr (EPI == EPI_CONVT) {
                    const int q = n / p.Cout;
                    const int co = n - q * p.Cout;
                    const int oy = 2 * ct_y + (q >> 1), ox = 2 * ct_x + (q & 1);
                    const int oH = 2 * p.out_H, oW = 2 * p.out_W;
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + co);
                        v0 += b4.x, v1 += b4.y, v2 += b4.z, v3 += b4.w;
                    }
                    if (p.out32) {
                        const int64_t o = (((int64_t)ct_b * oH + oy) * oW + ox) * p.ldc + co;
                        *reinterpret_cast<float4*>(p.out32 + o) = make_float4(v0, v1, v2, v3);
                    }
                    if (p.out16) {
                        float a0 = v0, a1 = v1, a2 = v2, a3 = v3;
                        if (p.act == ACT_RELU)
                            a0 = fmaxf(v0, 0.f), a1 = fmaxf(v1, 0.f), a2 = fmaxf(v2, 0.f),
                            a3 = fmaxf(v3, 0.f);
                        const int64_t o =
                            p.out16_border
                                ? ((((int64_t)ct_b * (oH + 2) + oy + 1) * (oW + 2) + ox + 1) *
                                       p.ldc + co)
                                : ((((int64_t)ct_b * oH + oy) * oW + ox) * p.ldc + co);
                        store4_16<T>((T*)p.out16 + o, a0, a1, a2, a3);
                    }
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI>
void gemm_launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<T, BM, BN, WM, WN, AMODE, EPI>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        ME_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   smem));
        attr_set = true;
    }
    const int64_t nbm = cdiv(p.M, BM), nbn = cdiv(p.N, BN);
    const int64_t grid = nbm * nbn;
    ME_CHECK(grid > 0 && grid < (1ll << 31), ME_ERR_BAD_SHAPE, "gemm grid %lld out of range",
             (long long)grid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), smem, stream, p);
    ME_HIP(hipGetLastError());
}

// One instantiation set per (dtype, amode, epi); defined in gemm_*.hip.
template <typename T, int AMODE, int EPI>
void gemm_dispatch(const GemmParams& p, int cfg, hipStream_t stream);

#define ME_GEMM_DISPATCH_BODY(T, AMODE, EPI)                                              \
    template <>                                                                           \
    void gemm_dispatch<T, AMODE, EPI>(const GemmParams& p, int cfg, hipStream_t stream) { \
        switch (cfg) {                                                                    \
            case 0: gemm_launch_cfg<T, 256, 256, 2, 4, AMODE, EPI>(p, stream); break;     \
            case 1: gemm_launch_cfg<T, 128, 128, 2, 2, AMODE, EPI>(p, stream); break;     \
            case 2: gemm_launch_cfg<T, 64, 64, 2, 2, AMODE, EPI>(p, stream); break;       \
            case 3: gemm_launch_cfg<T, 256, 128, 4, 2, AMODE, EPI>(p, stream); break;     \
            default: fail(ME_ERR_BAD_ARG, "gemm: bad tile config %d", cfg);               \
        }                                                                                 \
    }

}  // namespace me
