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
#include <type_traits>

#include "common.h"
#include "mx_fp8.h"

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

// x * Phi(x), the exact-erf GELU the reference uses (burn activation::gelu, vit.rs:121), as
// max(x, 0) - a * Phi(-a) with a = |x|: the lower tail Phi(-a) = erfc(a / sqrt 2) / 2 keeps its relative
// accuracy for either sign.  log2 Phi(-a) is smooth and nearly quadratic, so a degree-9 polynomial fitted on
// [0, 5.5] (Chebyshev nodes) followed by ONE v_exp_f32 reproduces Phi(-a) to 7e-7 relative there, and it
// extrapolates to a = 9 within 0.5 in the exponent, where a * Phi(-a) = 1e-18 is below every format in use:
// a is clamped to 9.  |error of the result| <= 2.4e-7 over [-9, 9], which is the f32 rounding of the result
// itself (the Abramowitz-Stegun 7.1.26 form used before: 2.1e-7, with a v_rcp_f32 besides the exponential).
// Four values at a time: the Horner steps are v_pk_fma_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 x) {
    // v_med3_f32 for the clamps: fminf / fmaxf cost a canonicalising v_max_f32 each on top
    auto clamp_abs = [](float v) { return __builtin_amdgcn_fmed3f(fabsf(v), 0.0f, 9.0f); };
    auto positive = [](float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 3.0e38f); };
    const f32x2 a0 = {clamp_abs(x[0]), clamp_abs(x[1])}, a1 = {clamp_abs(x[2]), clamp_abs(x[3])};
    const f32x2 p0 = {positive(x[0]), positive(x[1])}, p1 = {positive(x[2]), positive(x[3])};
    // two independent v_pk_fma_f32 chains, step by step (a dependent v_pk_fma_f32 costs a wait state)
    f32x2 q0 = a0 * 7.329674645e-08f - 1.913058668e-06f, q1 = a1 * 7.329674645e-08f - 1.913058668e-06f;
#define ME_GELU_STEP(c) q0 = q0 * a0 + (c), q1 = q1 * a1 + (c)
    ME_GELU_STEP(1.896382855e-05f);
    ME_GELU_STEP(-6.020677392e-05f);
    ME_GELU_STEP(-5.156729021e-04f);
    ME_GELU_STEP(7.680844516e-03f);
    ME_GELU_STEP(-5.303888768e-02f);
    ME_GELU_STEP(-4.589743018e-01f);
    ME_GELU_STEP(-1.151143670e+00f);
    ME_GELU_STEP(-9.999989867e-01f);
#undef ME_GELU_STEP
    const f32x2 h0 = {__builtin_amdgcn_exp2f(q0.x), __builtin_amdgcn_exp2f(q0.y)};  // Phi(-a)
    const f32x2 h1 = {__builtin_amdgcn_exp2f(q1.x), __builtin_amdgcn_exp2f(q1.y)};
    const f32x2 r0 = p0 - h0 * a0, r1 = p1 - h1 * a1;
    return f32x4{r0.x, r0.y, r1.x, r1.y};
}

// The same for results that are rounded to 16 bits (or to fp8) at once -- fc1's epilogue, the only GELU of the model
// (vit.rs:121): a degree-6 fit of log2 Phi(-a) on [0, 6] (Phi(-a) to 4.5e-5 relative, a * Phi(-a) to 6.5e-6 absolute: a
// tenth of the f16 rounding of the result, 2^-11 relative), the argument clamped at 6 (beyond, a * Phi(-a) < 6e-9).
// Three Horner steps fewer per value: 26 instead of 32 vector instructions per four values in an epilogue that is
// VALU-bound (9.4 us of a 37 us fc1 tile with the degree-9 form).
__device__ __forceinline__ f32x4 gelu_erf4_16bit(f32x4 x) {
    auto clamp_abs = [](float v) { return __builtin_amdgcn_fmed3f(fabsf(v), 0.0f, 6.0f); };
    auto positive = [](float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 3.0e38f); };
    const f32x2 a0 = {clamp_abs(x[0]), clamp_abs(x[1])}, a1 = {clamp_abs(x[2]), clamp_abs(x[3])};
    const f32x2 p0 = {positive(x[0]), positive(x[1])}, p1 = {positive(x[2]), positive(x[3])};
    f32x2 q0 = a0 * 2.299005791e-05f - 6.111001130e-04f, q1 = a1 * 2.299005791e-05f - 6.111001130e-04f;
#define ME_GELU_STEP(c) q0 = q0 * a0 + (c), q1 = q1 * a1 + (c)
    ME_GELU_STEP(7.195567712e-03f);
    ME_GELU_STEP(-5.118535087e-02f);
    ME_GELU_STEP(-4.612718821e-01f);
    ME_GELU_STEP(-1.150174260e+00f);
    ME_GELU_STEP(-1.000064731e+00f);
#undef ME_GELU_STEP
    const f32x2 h0 = {__builtin_amdgcn_exp2f(q0.x), __builtin_amdgcn_exp2f(q0.y)};  // Phi(-a)
    const f32x2 h1 = {__builtin_amdgcn_exp2f(q1.x), __builtin_amdgcn_exp2f(q1.y)};
    const f32x2 r0 = p0 - h0 * a0, r1 = p1 - h1 * a1;
    return f32x4{r0.x, r0.y, r1.x, r1.y};
}

// gelu_erf4_16bit over NP2 register pairs at once, step by step: every Horner step is NP2 INDEPENDENT v_pk_fma_f32.  Called
// four values at a time the two chains of a call depend on themselves at every step, and with two waves per SIMD the
// epilogue of fc1 ran at 19 cycles per vector instruction (11.2 us of a 45.7 us tile); eight chains side by side keep the
// pipe issuing.  Same arithmetic per value, bit for bit.
template <int NP2>
__device__ __forceinline__ void gelu_erf_batch_16bit(f32x2 (&x)[NP2]) {
    f32x2 a[NP2], q[NP2];
#pragma unroll
    for (int i = 0; i < NP2; ++i) {
        a[i] = f32x2{__builtin_amdgcn_fmed3f(fabsf(x[i].x), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(fabsf(x[i].y), 0.0f, 6.0f)};
        x[i] = f32x2{__builtin_amdgcn_fmed3f(x[i].x, 0.0f, 3.0e38f), __builtin_amdgcn_fmed3f(x[i].y, 0.0f, 3.0e38f)};
    }
#pragma unroll
    for (int i = 0; i < NP2; ++i) q[i] = a[i] * 2.299005791e-05f - 6.111001130e-04f;
#define ME_GELU_STEP(c)                \
    _Pragma("unroll") for (int i = 0; i < NP2; ++i) q[i] = q[i] * a[i] + (c)
    ME_GELU_STEP(7.195567712e-03f);
    ME_GELU_STEP(-5.118535087e-02f);
    ME_GELU_STEP(-4.612718821e-01f);
    ME_GELU_STEP(-1.150174260e+00f);
    ME_GELU_STEP(-1.000064731e+00f);
#undef ME_GELU_STEP
#pragma unroll
    for (int i = 0; i < NP2; ++i) {
        const f32x2 hh = {__builtin_amdgcn_exp2f(q[i].x), __builtin_amdgcn_exp2f(q[i].y)};  // Phi(-a)
        x[i] = x[i] - hh * a[i];
    }
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

// Dynamic tile order of the persistent kernels.  With the static order (workgroup b takes tiles b, b+G,
// ...) a workgroup that starts late delays the whole launch by its lateness, and beside the side-stream
// ViTs somebody always starts late: their small kernels take CUs at every kernel boundary (bench.py:
// 2.3 ms of a 26.7 ms step).  Here every workgroup still starts on tile b, but takes its further tiles
// from a ticket counter of its XCD (so each XCD keeps walking its own contiguous run of the tile order):
// a workgroup that lost time simply takes fewer tiles.  A tile must be known one tile ahead (its first
// slabs are staged under the previous tile's last ones), so tickets are drawn two tiles ahead, always
// where their latency is already paid: in the prologue behind the first staging loads, and at the start
// of each epilogue beside the bias loads.  Lane 0 of wave 0 draws; the tile index reaches the other
// waves through one word of the (then free) epilogue scratch and one extra barrier per tile.  The last
// workgroup to exit zeroes the counters for the next launch on the stream.  Results do not depend on
// which workgroup computes a tile.
struct TileQueue {
    unsigned* q;
    int xcd, first_round, ntiles;
    __device__ __forceinline__ void init(unsigned* queue, int ntiles_) {
        q = queue, ntiles = ntiles_;
        xcd = blockIdx.x & 7;
        first_round = gridDim.x >> 3;  // tickets continue after the static first round
    }
    // one lane: next tile of this XCD's run, or -1
    __device__ __forceinline__ int draw() {
        const unsigned ticket =
            __hip_atomic_fetch_add(q + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long vb = xcd + 8ll * ((long long)ticket + first_round);
        return vb < ntiles ? (int)vb : -1;
    }
    __device__ __forceinline__ void leave() {
        if (threadIdx.x == 0) {
            const unsigned gone =
                __hip_atomic_fetch_add(q + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gone == gridDim.x - 1) {  // everybody else has drawn its last ticket before counting out
                for (int i = 0; i < 9; ++i)
                    __hip_atomic_store(q + i * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// Row segment of a tile (GemmParams::seg1 / seg2): tiles never straddle a segment boundary.
__device__ __forceinline__ int row_segment(const GemmParams& p, int m0) {
    return p.seg1 == 0 ? 0 : ((p.seg2 != 0 && m0 >= p.seg2) ? 2 : (m0 >= p.seg1 ? 1 : 0));
}
__device__ __forceinline__ const char* segment_weights(const GemmParams& p, int m0) {
    const int g = row_segment(p, m0);
    return (const char*)(g == 0 ? p.W : (g == 1 ? p.W_s1 : p.W_s2));
}

// Implicit-GEMM addressing shared by the three kernels.  Row m of the A operand is output pixel
// (b, y, x); its K index is (tap, cin) with a 64-wide slab never straddling a tap (Cin % 64 == 0).
// conv_pixel: element index / Cin of the tap-(0,0) source pixel of output row gm in the zero-bordered map.
__device__ __forceinline__ int64_t conv_pixel(const GemmParams& p, int gm) {
    const int ppi = p.out_H * p.out_W;
    const int b = gm / ppi;
    const int rem = gm - b * ppi;
    const int y = rem / p.out_W;
    const int x = rem - y * p.out_W;
    return ((int64_t)b * p.in_Hp + y * p.stride) * p.in_Wp + x * p.stride;
}
// K slab kt of the A operand (byte offset from a row's tap-(0,0) address) and of the weight rows.  A convolution's
// K index is (tap, cin) in the packed weights, [Cout][tap][Cin], and its slabs are WALKED input-channel slab outermost,
// taps inside: slab kt = (c, tap) with c = kt / taps.  Every convolution kernel uses this one order -- the halo tile
// below must (it stages one channel slab's halo for its nine taps), and the other tiles follow so that a convolution's
// f32 sums do not depend on the tile configuration the problem size happens to select (a batch equals a loop of
// batch-one calls bit for bit).  Stateless: two multiplications by reciprocals instead of divisions.
template <int AMODE>
struct SlabWalk {
    int pad, taps, kw, cin;
    __device__ __forceinline__ void init(const GemmParams& p) {
        pad = AMODE == A_CONV ? (p.KH - 1) / 2 : 0;
        taps = AMODE == A_CONV ? p.KH * p.KW : 1;
        kw = AMODE == A_CONV ? p.KW : 1;
        cin = p.Cin;
    }
    // slab -> (c, tap): taps is 1 or 9 (KH = KW in {1, 3}); 7282 = ceil(2^16 / 9) is exact for kt < 32768
    __device__ __forceinline__ void split(int kt, int& c, int& tap) const {
        c = taps == 1 ? kt : (int)(((unsigned)kt * 7282u) >> 16);
        tap = kt - c * taps;
    }
    __device__ __forceinline__ int64_t a_off(const GemmParams& p, int kt) const {
        if constexpr (AMODE == A_PLAIN) {
            return (int64_t)kt * 128;
        } else {
            int c, tap;
            split(kt, c, tap);
            const int ky = kw == 1 ? 0 : (int)(((unsigned)tap * 21846u) >> 16);  // tap / 3
            const int kx = tap - ky * kw;
            return ((int64_t)(ky + 1 - pad) * p.in_Wp + (kx + 1 - pad)) * cin * 2 + c * 128;
        }
    }
    __device__ __forceinline__ int64_t w_off(int kt) const {
        if constexpr (AMODE == A_PLAIN) {
            return (int64_t)kt * 128;
        } else {
            int c, tap;
            split(kt, c, tap);
            return ((int64_t)tap * cin + c * 64) * 2;
        }
    }
    __device__ __forceinline__ int64_t next(const GemmParams& p, int kt) const { return a_off(p, kt); }
};

// Row tiles of a launch whose tile height does not divide the row-segment boundaries (the 352-row tile): every segment is
// tiled from its own first row, a tile never straddles a boundary, and the rows a segment's last tile has beyond the
// segment's end are neither computed into nor stored (row limit of the tile = the end of its segment).  For tile
// heights that divide seg1 and seg2 this is the plain m0 = row_tile * BM.
template <int BM>
__device__ __forceinline__ void seg_tile_rows(const GemmParams& p, int rt, int& m0, int& m_lim) {
    if (p.seg1 == 0) {
        m0 = rt * BM, m_lim = p.M;
        return;
    }
    const int e1 = p.seg2 ? p.seg2 : p.M;
    const int t0 = (p.seg1 + BM - 1) / BM, t1 = (e1 - p.seg1 + BM - 1) / BM;
    if (rt < t0)
        m0 = rt * BM, m_lim = p.seg1;
    else if (rt < t0 + t1)
        m0 = p.seg1 + (rt - t0) * BM, m_lim = e1;
    else
        m0 = p.seg2 + (rt - t0 - t1) * BM, m_lim = p.M;
}

// block -> tile.  Blocks b and b+8 share an XCD (and its 4 MiB L2), so each XCD gets a contiguous run
// of the tile order; that order walks "super-rows" of 8 tile rows column by column, so the ~32 tiles
// an XCD has in flight form an 8 x 4 patch: every A k-slab is shared by 4 of them and every W k-slab
// by 8 (with the plain n-fastest order the whole W matrix streams through L2 once per tile row:
// 44 % L2 misses on the fc1 shape, profiles/r01_pmc_gemm.md).
template <int BM, int BN, bool SEG = false>
__device__ __forceinline__ void tile_origin(const GemmParams& p, int bid, int nwg, int& m0, int& n0, int* m_lim = nullptr,
                                            int* row_tile = nullptr) {
    const int GM = SEG && p.patch_rows ? p.patch_rows : 8;
    const int nbm = SEG ? seg_row_tiles<BM>(p.M, p.seg1, p.seg2) : (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);  // bijective
    const int per_sr = GM * nbn;
    const int sr = t / per_sr;
    const int rem = t - sr * per_sr;
    const int h = min(GM, nbm - sr * GM);  // rows of this super-row
    const int n = rem / h;
    const int rr = rem - n * h;
    if (row_tile) *row_tile = sr * GM + rr;
    if constexpr (SEG) {
        int lim;
        seg_tile_rows<BM>(p, sr * GM + rr, m0, lim);
        if (m_lim) *m_lim = lim;
    } else {
        m0 = (sr * GM + rr) * BM;
        if (m_lim) *m_lim = p.M;
    }
    n0 = n * BN;
}

// rows per epilogue pass (in 16-row m-tiles): the largest of {4, 2, 1} dividing MI whose LDS footprint
// (all waves) fits in the main loop's allocation
constexpr int epi_mi_chunk(int MI, int TN, int NW, int main_bytes) {
    for (int c = 4; c >= 1; c >>= 1)
        if (MI % c == 0 && NW * 16 * c * (TN * 4) <= main_bytes) return c;
    return 1;
}

// Epilogue.  Each lane finishes one 8-column granule of one output row at a time: everything that
// depends only on n (bias, gamma) is loaded once per lane, the pixel coordinates of bordered / pixel-
// shuffled outputs walk with the row, f32 results leave as two 16-byte stores and 16-bit results as
// ONE 16-byte store (8-byte stores ran the qkv epilogue at 2.9 TB/s, 16-byte ones at 7 TB/s).
#ifndef ME_EPI_PRE
#define ME_EPI_PRE 1
#endif
// epilogue_granule<EPI_STORE, MODE >= kEpiConst>: which options of the launch are compiled in
constexpr int kEpiConst = 16, kEpiRes = 1, kEpiOut32 = 2, kEpiOut16 = 4, kEpiBorder = 8, kEpiLo = 32, kEpiHi2 = 64, kEpiResB = 128,
              kEpiTapBias = 256;
struct EpiLane {
    float4 bias[2], gamma[2];  // per-lane constants for columns n..n+3 and n+4..n+7
    int q, co;                 // EPI_CONVT: n = q * Cout + co
    int add32, add16;          // EPI_CONVT: element offsets of (sub-pixel q, channel co) from output pixel (2y, 2x)
};
struct EpiRow {
    int m;        // output row
    int b, y, x;  // pixel of that row when rows are pixels of [B][out_H][out_W]
    // EPI_CONVT: element offsets of output pixel (b, 2y, 2x) in out32 and out16 (the bordered layout included), walked with
    // the row -- worked out per granule from (b, y, x) they were two 64-bit products per row: the head's ConvTranspose took
    // 239 us where the same GEMM with a row-major store takes 185
    int64_t o32, o16;
};

// amax16: running largest magnitude this lane has rounded to an f16 operand (the overflow guard of common.h;
// v_max3_f32 with |x| source modifiers: four instructions per granule, nothing for bf16)
template <typename T>
__device__ __forceinline__ void track_amax16(float& amax16, const float (&a)[8], bool hi_ok) {
    if constexpr (std::is_same<T, f16>::value) {
        amax16 = fmaxf(fmaxf(amax16, fabsf(a[0])), fabsf(a[1]));
        amax16 = fmaxf(fmaxf(amax16, fabsf(a[2])), fabsf(a[3]));
        if (hi_ok) {
            amax16 = fmaxf(fmaxf(amax16, fabsf(a[4])), fabsf(a[5]));
            amax16 = fmaxf(fmaxf(amax16, fabsf(a[6])), fabsf(a[7]));
        }
    }
}
template <typename T>
__device__ __forceinline__ void store_16bit(T* dst, const float (&a)[8], bool hi_ok) {
    typedef T v8 __attribute__((ext_vector_type(8)));
    typedef T v4 __attribute__((ext_vector_type(4)));
    if (hi_ok) {
        v8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (T)a[e];
        *reinterpret_cast<v8*>(dst) = v;
    } else {
        v4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (T)a[e];
        *reinterpret_cast<v4*>(dst) = v;
    }
}

// lo part of a split 16-bit output: T(v - T(v)), the operand rounding error of the hi part made an operand
template <typename T>
__device__ __forceinline__ void store_16bit_lo(T* dst, const float (&a)[8], bool hi_ok) {
    float l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float h = (float)(T)a[e];
        l[e] = fabsf(h) == INFINITY ? 0.f : a[e] - h;  // an overflowed hi part stays +-inf, as without the split
    }
    store_16bit<T>(dst, l, hi_ok);
}

// v: accumulators of columns n + 4*h .. n + 4*h + 3 (h = 0, 1); hi_ok: the second half exists (n + 4 < N)
// MODE 3: bias (+ GELU) then quantised to MX fp8 (the fp8 GEMM's fc1 epilogue, mx_fp8.h)
// MODE: 0 = every option checked at run time; 1 / 2 = the ViT fast paths of EPI_STORE (16-bit output
// only, bias, no residual, no border; 1: no activation (qkv), 2: GELU (fc1)) with the branches gone
// pre: the granule's two residual vectors, already loaded (compiled-in residual modes: the pass's loads go out together)
template <typename T, int EPI, int MODE>
__device__ __forceinline__ void epilogue_granule(const GemmParams& p, const EpiRow& r, int n,
                                                 const EpiLane& lc, const f32x4 (&v)[2], bool hi_ok,
                                                 float& amax16, const float4* pre = nullptr) {
    const int m = r.m;
    float a[8];  // values for the 16-bit copy
    if constexpr (EPI == EPI_STORE && MODE != 0 && MODE < kEpiConst) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float x0 = v[h][0] + lc.bias[h].x, x1 = v[h][1] + lc.bias[h].y;
            const float x2 = v[h][2] + lc.bias[h].z, x3 = v[h][3] + lc.bias[h].w;
            if constexpr (MODE == 2 || MODE == 3) {
                const f32x4 g = gelu_erf4_16bit(f32x4{x0, x1, x2, x3});
                a[4 * h] = g[0], a[4 * h + 1] = g[1], a[4 * h + 2] = g[2], a[4 * h + 3] = g[3];
            } else {
                a[4 * h] = x0, a[4 * h + 1] = x1, a[4 * h + 2] = x2, a[4 * h + 3] = x3;
            }
        }
        if constexpr (MODE == 3) {
            *reinterpret_cast<uint2*>(p.out8 + (int64_t)m * p.ldc + n) = quantise_granule_fp8(p, m, n, a);
        } else {
            track_amax16<T>(amax16, a, true);
            store_16bit<T>((T*)p.out16 + (int64_t)m * p.ldc + n, a, hi_ok);
        }
    } else if constexpr (EPI == EPI_STORE) {
        // MODE 0: every option of the launch checked at run time, per granule.  MODE >= 16 (kEpiConst | flag bits): the
        // option combinations the convolutions of the model launch, as compile-time constants, the activation without a
        // branch (a lower bound of 0 or -inf) -- see EPI_CONVT below for what the run-time checks cost a store-heavy launch.
        constexpr bool CF = MODE >= kEpiConst;
        const bool f_res = CF ? (MODE & kEpiRes) != 0 : p.res32 != nullptr;
        const bool f_resb = CF ? (MODE & kEpiResB) != 0 : p.res32b != nullptr;
        const bool f_o32 = CF ? (MODE & kEpiOut32) != 0 : p.out32 != nullptr;
        const bool f_o16 = CF ? (MODE & kEpiOut16) != 0 : p.out16 != nullptr;
        const bool f_border = CF ? (MODE & kEpiBorder) != 0 : p.out16_border != 0;
        const bool f_lo = CF ? (MODE & kEpiLo) != 0 : p.lo_off16 != 0;
        const bool f_hi2 = CF ? (MODE & kEpiHi2) != 0 : p.hi2_off16 != 0;
        // GemmParams::tap_bias (a layer composed into this 3x3 convolution, weights.hip compose_features): at the image border the
        // taps that fall into the zero padding take their share of the bias with them (r.y / r.x: the bordered output's pixel).
        // A compiled-in mode only (gemm_launch admits tap_bias with exactly that mode's options): as a run-time option of MODE 0
        // the plain-GEMM kernels that also carry this body stopped compiling (their LDS-DMA's wave-uniform operands came out in
        // VGPRs).  A wave-uniform branch around branch-free arithmetic.
        const bool f_tap = CF ? (MODE & kEpiTapBias) != 0 : false;
        // (a wave-uniform branch around branch-free arithmetic: a divergent one here let the optimiser thread the kernels' tile
        // loops through it, and the LDS-DMA's wave-uniform operands came out in VGPRs)
        const bool at_edge = f_tap && __builtin_amdgcn_ballot_w64(r.y == 0 || r.y == p.out_H - 1 || r.x == 0 || r.x == p.out_W - 1) != 0;
        // CF: ReLU (or none) as max(x, bound): bound16 for the 16-bit copy, bound32 for the f32 output (act16_only: unclamped)
        const float bound16 = p.act == ACT_RELU ? 0.f : -INFINITY;
        const float bound32 = (p.act == ACT_RELU && !p.act16_only) ? 0.f : -INFINITY;
        const int64_t row32 = (int64_t)m * p.ldc;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !hi_ok) break;
            float x0 = v[h][0] + lc.bias[h].x, x1 = v[h][1] + lc.bias[h].y;
            float x2 = v[h][2] + lc.bias[h].z, x3 = v[h][3] + lc.bias[h].w;
            if (at_edge) {  // rare: the tiles along the image border
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t % 3;
                    const bool out = (ky == 0 && r.y == 0) || (ky == 2 && r.y == p.out_H - 1) || (kx == 0 && r.x == 0) ||
                                     (kx == 2 && r.x == p.out_W - 1);
                    const float4 t4 = *reinterpret_cast<const float4*>(p.tap_bias + t * p.N + n + 4 * h);
                    x0 -= out ? t4.x : 0.f, x1 -= out ? t4.y : 0.f, x2 -= out ? t4.z : 0.f, x3 -= out ? t4.w : 0.f;
                }
            }
            if (f_res) {
                const float4 r4 = (CF && ME_EPI_PRE) ? pre[h] : *reinterpret_cast<const float4*>(p.res32 + row32 + n + 4 * h);
                x0 += r4.x, x1 += r4.y, x2 += r4.z, x3 += r4.w;
            }
            if (f_resb) {
                const float4 r4 = (CF && ME_EPI_PRE) ? pre[2 + h] : *reinterpret_cast<const float4*>(p.res32b + row32 + n + 4 * h);
                x0 += r4.x, x1 += r4.y, x2 += r4.z, x3 += r4.w;
            }
            float a0 = x0, a1 = x1, a2 = x2, a3 = x3;
            if constexpr (CF) {
                a0 = fmaxf(x0, bound16), a1 = fmaxf(x1, bound16), a2 = fmaxf(x2, bound16), a3 = fmaxf(x3, bound16);
                if (f_o32)
                    *reinterpret_cast<float4*>(p.out32 + row32 + n + 4 * h) =
                        make_float4(fmaxf(x0, bound32), fmaxf(x1, bound32), fmaxf(x2, bound32), fmaxf(x3, bound32));
            } else {
                if (p.act == ACT_GELU) {
                    const f32x4 g = gelu_erf4(f32x4{x0, x1, x2, x3});
                    a0 = g[0], a1 = g[1], a2 = g[2], a3 = g[3];
                } else if (p.act == ACT_RELU) {
                    a0 = fmaxf(x0, 0.f), a1 = fmaxf(x1, 0.f), a2 = fmaxf(x2, 0.f), a3 = fmaxf(x3, 0.f);
                }
                if (f_o32)
                    *reinterpret_cast<float4*>(p.out32 + row32 + n + 4 * h) =
                        p.act16_only ? make_float4(x0, x1, x2, x3) : make_float4(a0, a1, a2, a3);
            }
            a[4 * h] = a0, a[4 * h + 1] = a1, a[4 * h + 2] = a2, a[4 * h + 3] = a3;
        }
        if (f_o16) {
            const int64_t ld16 = p.ldc16 ? p.ldc16 : p.ldc;
            const int64_t row16 =
                f_border
                    ? (((int64_t)r.b * (p.out_H + 2) + r.y + 1) * (p.out_W + 2) + r.x + 1) * ld16
                    : (int64_t)m * ld16;
            track_amax16<T>(amax16, a, hi_ok);
            store_16bit<T>((T*)p.out16 + row16 + n, a, hi_ok);
            if (f_lo) store_16bit_lo<T>((T*)p.out16 + row16 + n + p.lo_off16, a, hi_ok);
            if (f_hi2) store_16bit<T>((T*)p.out16 + row16 + n + p.hi2_off16, a, hi_ok);
        }
    } else if constexpr (EPI == EPI_RESID_SCALE) {
        const int64_t row32 = (int64_t)m * p.ldc;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !hi_ok) break;
            const float4 r4 = *reinterpret_cast<const float4*>(p.res32 + row32 + n + 4 * h);
            // same operation order as the reference: (xs * gamma) + residual
            *reinterpret_cast<float4*>(p.out32 + row32 + n + 4 * h) =
                make_float4((v[h][0] + lc.bias[h].x) * lc.gamma[h].x + r4.x,
                            (v[h][1] + lc.bias[h].y) * lc.gamma[h].y + r4.y,
                            (v[h][2] + lc.bias[h].z) * lc.gamma[h].z + r4.z,
                            (v[h][3] + lc.bias[h].w) * lc.gamma[h].w + r4.w);
        }
    } else if constexpr (EPI == EPI_PATCH_EMBED) {
        const int w = m / p.tokens_per_window;
        const int patch = m - w * p.tokens_per_window;
        const int64_t row32 = ((int64_t)w * (p.tokens_per_window + 1) + 1 + patch) * p.ldc;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !hi_ok) break;
            const float4 e4 =
                *reinterpret_cast<const float4*>(p.pos + (int64_t)(1 + patch) * p.N + n + 4 * h);
            *reinterpret_cast<float4*>(p.out32 + row32 + n + 4 * h) =
                make_float4(v[h][0] + lc.bias[h].x + e4.x, v[h][1] + lc.bias[h].y + e4.y,
                            v[h][2] + lc.bias[h].z + e4.z, v[h][3] + lc.bias[h].w + e4.w);
        }
    } else if constexpr (EPI == EPI_CONVT) {
        // MODE 0: every option checked at run time, per granule.  MODE 4 .. 7: the combinations the model launches, as
        // compile-time constants -- 4: 16-bit output, 5: f32 output, 6: both, 7: 16-bit [hi | lo]; no ReLU.  With the
        // checks in the granule (uniform branches, but branches: the granules' LDS reads and stores cannot be moved
        // across them) a store-heavy launch pays for them: the head's ConvTranspose 231 us, the same GEMM with a
        // row-major store through the run-time path 261 us, through the constant path 182 us.
        const bool w32 = MODE == 0 ? p.out32 != nullptr : (MODE == 5 || MODE == 6);
        const bool w16 = MODE == 0 ? p.out16 != nullptr : (MODE == 4 || MODE == 6 || MODE == 7);
        const bool relu = MODE == 0 ? p.act == ACT_RELU : false;
        const bool lo = MODE == 0 ? p.lo_off16 != 0 : MODE == 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !hi_ok) break;
            const float x0 = v[h][0] + lc.bias[h].x, x1 = v[h][1] + lc.bias[h].y;
            const float x2 = v[h][2] + lc.bias[h].z, x3 = v[h][3] + lc.bias[h].w;
            if (w32) {
                const int64_t o = r.o32 + lc.add32 + 4 * h;
                *reinterpret_cast<float4*>(p.out32 + o) = make_float4(x0, x1, x2, x3);
            }
            a[4 * h] = relu ? fmaxf(x0, 0.f) : x0, a[4 * h + 1] = relu ? fmaxf(x1, 0.f) : x1;
            a[4 * h + 2] = relu ? fmaxf(x2, 0.f) : x2, a[4 * h + 3] = relu ? fmaxf(x3, 0.f) : x3;
        }
        if (w16) {
            const int64_t o = r.o16 + lc.add16;
            track_amax16<T>(amax16, a, hi_ok);
            store_16bit<T>((T*)p.out16 + o, a, hi_ok);
            if (lo) store_16bit_lo<T>((T*)p.out16 + o + p.lo_off16, a, hi_ok);
        }
    }
}

// acc[i][j] is the 16x16 tile (m-tile i, n-tile j) of this wave's TM x TN block: lane holds row
// m = ... + (lane & 15), columns n = ... + 4*(lane >> 4) + r.  Each wave transposes its block through a
// private LDS region (the main loop's LDS is free by now) so that 8 consecutive lanes own one whole
// TN-column row segment.
// PIN_CONSTS (two-group kernel): the per-lane constants are loaded under `n_ok` and used under
// `row.m < M && n_ok` inside one of several run() instances, so on the (statically possible) paths where
// a wave stores nothing the waitcnt pass still sees their loads pending and protects the registers with a
// vmcnt(0) at their next write -- inside the K loop, on every slab.  An unconditional empty asm "use"
// after the row set-up settles them for good (cost: what is left of one L2 load latency, once per tile).
// after_loads(): called once the per-lane constant loads have been issued (the tile queue draws its
// ticket there, beside them)
// TILE2D (the halo convolution kernel below): the tile is a 16 x 16 block of output pixels instead of 256
// consecutive rows -- m0 is then the TILE index (image-major, tile rows, tile columns) and row t of the tile is pixel
// (t >> 4, t & 15) of it; everything downstream sees the pixel's linear row index and coordinates as before.
template <typename T, int EPI, int MI, int NI, int TM, int TN, int MI_CH, bool PIN_CONSTS = false,
          typename Hook = NoHook, int OUT8 = 0, bool TILE2D = false, int TILE_H = 16, bool CF_ALLOWED = true>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MI][NI], int m0, int n0,
                                              int wm, int wn, int lane, char* epi_lds,
                                              Hook after_loads = Hook(), int m_lim = -1) {
    // rows >= Mrows are not stored: p.M, or the end of the tile's row segment (seg_tile_rows)
    const int Mrows = m_lim < 0 ? p.M : m_lim;
    // (the epilogue's per-lane constants are derived behind an opaque use of the lane index: see resid_ln_epilogue)
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15;
    const int ncol = (lane >> 4) * 4;  // first of this lane's 4 consecutive n within a 16-tile
    if constexpr (EPI == EPI_HEAD_COMPOSED) {
        // The head behind its composed ConvTranspose o conv3x3 (weights.hip compose_head): this wave's TN = 32 columns are the 32
        // channels of ONE output phase (dy, dx) = its column quarter; row t of the 16 x 16 pixel tile is half-resolution pixel
        // (y, x), whose phase pixel is (2y + dy, 2x + dx) of the full-resolution map.  relu(acc + bias) . w2, reduced over the
        // 32 channels (8 in the lane, 24 in three more lanes), + b2, ReLU, / f_norm, clamp -- EPI_HEAD_FINAL's arithmetic.
        // At the image border the taps of the ORIGINAL 3x3 convolution that fall into its zero padding take their share of the
        // bias with them (tap_bias; the activation part vanishes by itself, the operand being zero-bordered).
        static_assert(TILE2D && TN == 32 && NI == 2, "the composed head runs on the 128-channel halo tile");
        after_loads();
        const int phase = (n0 + wn * TN) >> 5, dy = phase >> 1, dx = phase & 1;
        float w2v[NI][4], bv[NI][4];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = j * 16 + ncol + r;
                w2v[j][r] = p.w2[ch];
                bv[j][r] = p.bias[ch];
            }
        const int tx = p.out_W >> 4, ty = p.out_H / TILE_H;
        const int img = m0 / (tx * ty);
        const int rem = m0 - img * (tx * ty);
        const int y0 = (rem / tx) * TILE_H, x0 = (rem - (rem / tx) * tx) * 16;
        const int H2 = 2 * p.out_H, W2 = 2 * p.out_W;
        const float fn = p.f_norm ? p.f_norm[img] : 1.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int t = wm * TM + i * 16 + frow;
            const int Y = 2 * (y0 + (t >> 4)) + dy, X = 2 * (x0 + (t & 15)) + dx;
            float b[NI][4];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) b[j][r] = bv[j][r];
            if (Y == 0 || Y == H2 - 1 || X == 0 || X == W2 - 1) {  // rare: one pixel in 384
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool out = (ky == 0 && Y == 0) || (ky == 2 && Y == H2 - 1) || (kx == 0 && X == 0) || (kx == 2 && X == W2 - 1);
                        if (!out) continue;
                        const float* tb = p.tap_bias + (ky * 3 + kx) * 32;
#pragma unroll
                        for (int j = 0; j < NI; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) b[j][r] -= tb[j * 16 + ncol + r];
                    }
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += fmaxf(acc[i][j][r] + b[j][r], 0.f) * w2v[j][r];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) {
                float v = fmaxf(s + p.b2[0], 0.f);
                if (p.f_norm) v = v / fn;
                v = fminf(fmaxf(v, p.clamp_lo), p.clamp_hi);
                p.out32[((int64_t)img * H2 + Y) * W2 + X] = v;
            }
        }
    } else if constexpr (EPI == EPI_HEAD_FINAL) {
        // N <= 32 = the whole tile width (WN == 1): relu(acc + bias) . w2, reduced over n.
        after_loads();
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
            if (lane < 16 && m < Mrows) {
                float v = fmaxf(s + p.b2[0], 0.f);
                if (p.f_norm) v = v / p.f_norm[m / p.pixels_per_image];
                v = fminf(fmaxf(v, p.clamp_lo), p.clamp_hi);
                p.out32[m] = v;
            }
        }
    } else {
        // Scratch image: rows of TN f32 (TN*4 bytes, a multiple of 256), the 16-byte chunk c of row r kept
        // at chunk position c ^ (r & 15): conflict-free for the transposing ds_write_b128 (16 rows, one
        // chunk each) and for the row reads, without padding (two 16-row m-tiles of all 8 waves then
        // fit exactly in one 64 KiB stage).
        constexpr int RS = TN * 4;
        constexpr int ROWS = 16 * MI_CH;  // rows per pass
        constexpr int GPR = TN / 8;       // 8-column granules per row
        constexpr int RPI = 64 / GPR;     // rows covered by one wave-instruction
        constexpr int ITERS = ROWS / RPI;
        static_assert(MI % MI_CH == 0 && 64 % GPR == 0 && ROWS % RPI == 0, "epilogue pass shape");
        static_assert(TN % 64 == 0 || TN == 32, "scratch swizzle assumes 16 chunks per row (8 for TN 32)");
        constexpr int CMASK = TN / 4 - 1;  // chunks per row - 1
        const int gc = lane % GPR, r0 = lane / GPR;
        const int n = n0 + wn * TN + gc * 8;
        const bool n_ok = n < p.N, hi_ok = n + 4 < p.N;
        EpiLane lc;
        lc.bias[0] = lc.bias[1] = lc.gamma[0] = lc.gamma[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        lc.q = 0, lc.co = n, lc.add32 = 0, lc.add16 = 0;
        // EPI_CONVT: strides of the output maps (elements) -- out32 [B][2H][2W][ldc], out16 [B][2H (+2)][2W (+2)][ld16]
        [[maybe_unused]] const int ct_ld16 = (int)(p.ldc16 ? p.ldc16 : p.ldc);
        [[maybe_unused]] const int ct_w16 = 2 * p.out_W + (p.out16_border ? 2 : 0);
        if constexpr (EPI == EPI_CONVT) {
            lc.q = n / p.Cout;
            lc.co = n - lc.q * p.Cout;
            lc.add32 = ((lc.q >> 1) * 2 * p.out_W + (lc.q & 1)) * p.ldc + lc.co;
            lc.add16 = ((lc.q >> 1) * ct_w16 + (lc.q & 1)) * ct_ld16 + lc.co;
        }
        if (n_ok) {
            const int seg = row_segment(p, m0);
            const float* bias = seg == 0 ? p.bias : (seg == 1 ? p.bias_s1 : p.bias_s2);
            if (bias) {
                lc.bias[0] = *reinterpret_cast<const float4*>(bias + lc.co);
                if (hi_ok) lc.bias[1] = *reinterpret_cast<const float4*>(bias + lc.co + 4);
            }
            if constexpr (EPI == EPI_STORE) {
                // GemmParams::qcols (the qkv linear): this wave's columns are Q columns -- wave-uniform, qcols being a
                // multiple of 64 >= TN -- and leave scaled: (acc + bias) * qscale as acc * qscale + bias * qscale, the
                // accumulators multiplied in place (no register lives longer for it; other tiles skip the branch)
                const float cs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
                    int, n0 + wn * TN < p.qcols ? p.qscale : 1.0f)));
                if (cs != 1.0f) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j] *= cs;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        lc.bias[h] = make_float4(lc.bias[h].x * cs, lc.bias[h].y * cs, lc.bias[h].z * cs, lc.bias[h].w * cs);
                }
            }
            if constexpr (EPI == EPI_RESID_SCALE) {
                const float* gamma = seg == 0 ? p.gamma : (seg == 1 ? p.gamma_s1 : p.gamma_s2);
                lc.gamma[0] = *reinterpret_cast<const float4*>(gamma + n);
                if (hi_ok) lc.gamma[1] = *reinterpret_cast<const float4*>(gamma + n + 4);
            }
        }
        after_loads();
        EpiRow row;
        row.m = m0 + wm * TM + r0;
        row.b = row.y = row.x = 0;
        row.o32 = row.o16 = 0;
        [[maybe_unused]] int t2_b = 0, t2_y0 = 0, t2_x0 = 0, t2_row = wm * TM + r0;  // TILE2D: tile origin, row in tile
        if constexpr (TILE2D) {
            const int tx = p.out_W >> 4, ty = p.out_H / TILE_H;  // tiles of TILE_H rows x 16 columns of pixels
            t2_b = m0 / (tx * ty);
            const int rem = m0 - t2_b * (tx * ty);
            t2_y0 = (rem / tx) * TILE_H;
            t2_x0 = (rem - (rem / tx) * tx) * 16;
            row.b = t2_b, row.y = t2_y0 + (t2_row >> 4), row.x = t2_x0 + (t2_row & 15);
            row.m = (row.b * p.out_H + row.y) * p.out_W + row.x;
        }
        const bool pix = !TILE2D && (EPI == EPI_CONVT || (EPI == EPI_STORE && p.out16_border));
        if (pix) {  // one pair of divisions per lane; afterwards the pixel walks with the row
            const int ppi = p.out_H * p.out_W;
            const int mm = row.m < Mrows ? row.m : Mrows - 1;
            row.b = mm / ppi;
            const int rem = mm - row.b * ppi;
            row.y = rem / p.out_W;
            row.x = rem - row.y * p.out_W;
            if constexpr (EPI == EPI_CONVT) {
                const int oH = 2 * p.out_H, oW = 2 * p.out_W, bd = p.out16_border ? 1 : 0;
                row.o32 = (((int64_t)row.b * oH + 2 * row.y) * oW + 2 * row.x) * p.ldc;
                row.o16 = (((int64_t)row.b * (oH + 2 * bd) + 2 * row.y + bd) * ct_w16 + 2 * row.x + bd) * ct_ld16;
            }
        }
        float amax16 = 0.f;
        auto run = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            if constexpr (EPI == EPI_RESID_SCALE) {
                // The ViT updates its token stream in place (res32 == out32), and even where it does not the compiler
                // has to assume so: written granule by granule, every residual load stays behind the store before it
                // and the epilogue is a chain of dependent  load -> s_waitcnt vmcnt(0) -> store  round trips, 32 per
                // wave and tile (the ISA showed exactly that; 16.6 us of a 42 us proj tile).  Here the eight residual
                // loads of a pass are issued together and one pass ahead -- between the arithmetic of the pass before
                // and its stores -- so a wave pays the memory latency once per tile and the stores never wait.
                static_assert(!TILE2D, "the residual epilogue walks consecutive rows");
                constexpr int NP = MI / MI_CH;
                // Addresses: buffer instructions on a descriptor of this wave's TM x TN block -- ONE per-lane 32-bit byte
                // offset for all 64 loads and stores of the tile, the row group of an access in an SGPR offset (the
                // kernel has no VGPRs for sixteen 64-bit row pointers beside two passes of data).  Rows >= M and columns
                // >= N are masked by the hardware range check: their offset is pushed beyond num_records (loads return 0,
                // stores are dropped), so the bursts carry no branches.
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                const int mw = m0 + wm * TM;  // first row of this wave's block
                const int64_t wave_el = (int64_t)mw * p.ldc + n0 + wn * TN;
                const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(uniform_ptr((const char*)(p.res32 + wave_el))), 0, 0x7fffffff, 0x00020000);
                const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(uniform_ptr((const char*)(p.out32 + wave_el))), 0, 0x7fffffff, 0x00020000);
                const unsigned row_step = (unsigned)p.ldc * 4u;  // bytes per row
                const unsigned loff = (unsigned)r0 * row_step + gc * 32u;
                const int rows_left = Mrows - mw - r0;  // this lane's rows r0 + rt exist for rt < rows_left
                constexpr unsigned kOut = 0x80000000u;
                auto voff = [&](int rt, bool col_ok) { return rt < rows_left && col_ok ? loff : kOut; };
                f32x4 res[ITERS][2];
                auto load_res = [&](int pass) {
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int rt = pass * ROWS + it * RPI;
                        const unsigned so = (unsigned)rt * row_step;
                        res[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, voff(rt, n_ok), so, 0));
                        res[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, voff(rt, hi_ok) + 16u, so, 0));
                    }
                };
                load_res(0);
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
                    for (int i = 0; i < MI_CH; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            *reinterpret_cast<f32x4*>(epi_lds + (i * 16 + frow) * RS +
                                                      ((((j * 16 + ncol) >> 2) ^ frow) & CMASK) * 16) =
                                acc[pass * MI_CH + i][j];
                    f32x4 o[ITERS][2];
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int rr = it * RPI + r0;
                        const char* src = epi_lds + rr * RS;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((2 * gc + h) ^ rr) & CMASK) * 16);
                            // same operation order as the reference: (xs * gamma) + residual
                            o[it][h] = f32x4{(v[0] + lc.bias[h].x) * lc.gamma[h].x + res[it][h][0],
                                             (v[1] + lc.bias[h].y) * lc.gamma[h].y + res[it][h][1],
                                             (v[2] + lc.bias[h].z) * lc.gamma[h].z + res[it][h][2],
                                             (v[3] + lc.bias[h].w) * lc.gamma[h].w + res[it][h][3]};
                        }
                    }
                    if (pass + 1 < NP) load_res(pass + 1);
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int rt = pass * ROWS + it * RPI;
                        const unsigned so = (unsigned)rt * row_step;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o[it][0]), rout, voff(rt, n_ok), so, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o[it][1]), rout, voff(rt, hi_ok) + 16u, so, 0);
                    }
                }
                return;
            }
#pragma unroll
            for (int pass = 0; pass < MI / MI_CH; ++pass) {
#pragma unroll
                for (int i = 0; i < MI_CH; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        *reinterpret_cast<f32x4*>(epi_lds + (i * 16 + frow) * RS +
                                                  ((((j * 16 + ncol) >> 2) ^ frow) & CMASK) * 16) =
                            acc[pass * MI_CH + i][j];
                // LDS operations of one wave execute in order: the reads below see the writes above
                // (the compiled-in modes read their rows back a group of granules at a time, further down: with every
                // branch gone the scheduler otherwise keeps a whole pass of them in flight beside the residual vectors,
                // and the 256-register kernels spill)
                constexpr bool LATE_V = EPI == EPI_STORE && MODE >= kEpiConst;
                f32x4 v[ITERS][2];
                if constexpr (!LATE_V) {
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int rr = it * RPI + r0;
                        const char* src = epi_lds + rr * RS;
                        v[it][0] = *reinterpret_cast<const f32x4*>(src + (((2 * gc) ^ rr) & CMASK) * 16);
                        v[it][1] = *reinterpret_cast<const f32x4*>(src + (((2 * gc + 1) ^ rr) & CMASK) * 16);
                    }
                }
                if constexpr (MODE == 3) {
                    // fp8 output (N a multiple of 256: every column exists; rows up to Mrows): two rows at a time, so that
                    // each lane's bytes leave as one 16-byte store
                    static_assert(ITERS % 2 == 0, "row pairs");
#pragma unroll
                    for (int it = 0; it < ITERS; it += 2) {
                        uint2 q[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            float a[8];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x4 g = gelu_erf4_16bit(f32x4{v[it + u][h][0] + lc.bias[h].x, v[it + u][h][1] + lc.bias[h].y,
                                                                v[it + u][h][2] + lc.bias[h].z, v[it + u][h][3] + lc.bias[h].w});
                                a[4 * h] = g[0], a[4 * h + 1] = g[1], a[4 * h + 2] = g[2], a[4 * h + 3] = g[3];
                            }
                            q[u] = quantise_granule_fp8(p, row.m + u * RPI, n, a, row.m + u * RPI < Mrows);
                        }
                        store_granule_pair_fp8(p, row.m, row.m + RPI, n, q[0], q[1], Mrows);
                        row.m += 2 * RPI;
                    }
                    continue;
                }
                if (EPI == EPI_STORE && MODE == 2 && !TILE2D && !p.gelu_per_granule) {
                    // fc1: bias + GELU of the pass's ITERS x 8 values in one batch (gelu_erf_batch_16bit), then the stores
                    f32x2 g[ITERS * 4];
#pragma unroll
                    for (int it = 0; it < ITERS; ++it)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float4 b4 = lc.bias[h];
                            g[it * 4 + 2 * h] = f32x2{v[it][h][0] + b4.x, v[it][h][1] + b4.y};
                            g[it * 4 + 2 * h + 1] = f32x2{v[it][h][2] + b4.z, v[it][h][3] + b4.w};
                        }
                    gelu_erf_batch_16bit<ITERS * 4>(g);
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const float a8[8] = {g[it * 4].x,     g[it * 4].y,     g[it * 4 + 1].x, g[it * 4 + 1].y,
                                             g[it * 4 + 2].x, g[it * 4 + 2].y, g[it * 4 + 3].x, g[it * 4 + 3].y};
                        if (row.m < Mrows && n_ok) {
                            track_amax16<T>(amax16, a8, true);
                            store_16bit<T>((T*)p.out16 + (int64_t)row.m * p.ldc + n, a8, hi_ok);
                        }
                        row.m += RPI;
                    }
                    continue;
                }
                // compiled-in residual modes: the residual vectors of PB granules are requested together, before the first of
                // their stores -- loaded where they are used, each granule's load stays behind the store of the granule before
                // it (the output may be the residual, for all the compiler knows) and a pass is ITERS memory latencies long.
                // (A whole pass at once -- 8 registers per granule -- spills in the 256-register kernels.)
                constexpr bool PRE = ME_EPI_PRE && EPI == EPI_STORE && MODE >= kEpiConst && (MODE & kEpiRes) != 0;
                constexpr bool PRE_B = PRE && (MODE & kEpiResB) != 0;  // a second residual: one granule at a time (16 registers)
                constexpr int PB = PRE && !PRE_B ? (ITERS % 2 == 0 ? 2 : 1) : 1;
#pragma unroll
                for (int it0 = 0; it0 < ITERS; it0 += PB) {
                float4 pre[PB][4];
                if constexpr (PRE) {
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        int m_it;
                        if constexpr (TILE2D) {
                            const int tr = t2_row + u * RPI;
                            m_it = (t2_b * p.out_H + t2_y0 + (tr >> 4)) * p.out_W + t2_x0 + (tr & 15);
                        } else {
                            m_it = row.m + u * RPI;
                        }
                        const bool ok = m_it < Mrows && n_ok;
                        const float* src = p.res32 + (int64_t)(ok ? m_it : 0) * p.ldc + n;
                        pre[u][0] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
                        pre[u][1] = ok && hi_ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (PRE_B) {
                            const float* srcb = p.res32b + (int64_t)(ok ? m_it : 0) * p.ldc + n;
                            pre[u][2] = ok ? *reinterpret_cast<const float4*>(srcb) : make_float4(0.f, 0.f, 0.f, 0.f);
                            pre[u][3] = ok && hi_ok ? *reinterpret_cast<const float4*>(srcb + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
                if constexpr (LATE_V) {
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        const int rr = (it0 + u) * RPI + r0;
                        const char* src = epi_lds + rr * RS;
                        v[it0 + u][0] = *reinterpret_cast<const f32x4*>(src + (((2 * gc) ^ rr) & CMASK) * 16);
                        v[it0 + u][1] = *reinterpret_cast<const f32x4*>(src + (((2 * gc + 1) ^ rr) & CMASK) * 16);
                    }
                    asm volatile("" ::: "memory");  // (keeps the next group's reads behind this group's stores)
                }
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int it = it0 + u;
                    if (row.m < Mrows && n_ok) epilogue_granule<T, EPI, MODE>(p, row, n, lc, v[it], hi_ok, amax16, PRE ? pre[u] : nullptr);
                    if constexpr (TILE2D) {
                        t2_row += RPI;
                        row.y = t2_y0 + (t2_row >> 4), row.x = t2_x0 + (t2_row & 15);
                        row.m = (t2_b * p.out_H + row.y) * p.out_W + row.x;
                    } else {
                        row.m += RPI;
                    }
                    if (pix) {
                        row.x += RPI;
                        if constexpr (EPI == EPI_CONVT) row.o32 += 2 * RPI * p.ldc, row.o16 += 2 * RPI * ct_ld16;
                        while (row.x >= p.out_W) {
                            row.x -= p.out_W;
                            // one input row down = two output rows down, 2 W output pixels back
                            if constexpr (EPI == EPI_CONVT)
                                row.o32 += (int64_t)(2 * p.out_W) * p.ldc, row.o16 += (int64_t)(2 * ct_w16 - 2 * p.out_W) * ct_ld16;
                            if (++row.y == p.out_H) {
                                row.y = 0, ++row.b;
                                if constexpr (EPI == EPI_CONVT)  // the next image's first row: over the two border rows
                                    if (p.out16_border) row.o16 += (int64_t)(2 * ct_w16) * ct_ld16;
                            }
                        }
                    }
                }
                }
            }
        };
        if constexpr (PIN_CONSTS) {
            asm volatile(""
                         : "+v"(lc.bias[0].x), "+v"(lc.bias[0].y), "+v"(lc.bias[0].z), "+v"(lc.bias[0].w),
                           "+v"(lc.bias[1].x), "+v"(lc.bias[1].y), "+v"(lc.bias[1].z), "+v"(lc.bias[1].w));
            asm volatile(""
                         : "+v"(lc.gamma[0].x), "+v"(lc.gamma[0].y), "+v"(lc.gamma[0].z), "+v"(lc.gamma[0].w),
                           "+v"(lc.gamma[1].x), "+v"(lc.gamma[1].y), "+v"(lc.gamma[1].z), "+v"(lc.gamma[1].w));
        }
        if constexpr (EPI == EPI_STORE && OUT8 != 0) {
            run(std::integral_constant<int, 3>());  // bias + GELU -> MX fp8 (the only fp8-output user: fc1)
        } else if constexpr (EPI == EPI_STORE) {
            const bool simple = p.out16 && !p.out32 && !p.res32 && !p.res32b && !p.out16_border && p.bias &&
                                !p.lo_off16 && !p.ldc16;
            // the convolutions' combinations (no second residual, no GELU): compiled-in options
            const int mask = (p.res32 ? kEpiRes : 0) | (p.res32b ? kEpiResB : 0) | (p.out32 ? kEpiOut32 : 0) | (p.out16 ? kEpiOut16 : 0) |
                             (p.out16_border ? kEpiBorder : 0) | (p.lo_off16 ? kEpiLo : 0) | (p.hi2_off16 ? kEpiHi2 : 0) |
                             (p.tap_bias ? kEpiTapBias : 0);
            // (not in the 352-row tile's kernel, MI == 11: it serves the ViT's qkv / fc1 through modes 1 and 2, and the other
            // bodies beside them cost it registers -- per-pass scratch reloads inside its store loops)
            constexpr bool CF_MODES = CF_ALLOWED && MI != 11;  // (CF_ALLOWED false: the fp8 GEMM, whose 16-bit stores are mode 1)
            const bool cf_ok = CF_MODES && p.act != ACT_GELU && p.bias;
            if (simple && p.act == ACT_NONE)
                run(std::integral_constant<int, 1>());
            else if (simple && p.act == ACT_GELU)
                run(std::integral_constant<int, 2>());
            else if (cf_ok) {
                if constexpr (CF_MODES) {
                    if (mask == (kEpiOut16 | kEpiBorder))
                        run(std::integral_constant<int, kEpiConst | kEpiOut16 | kEpiBorder>());
                    else if (mask == (kEpiOut32 | kEpiOut16 | kEpiBorder))
                        run(std::integral_constant<int, kEpiConst | kEpiOut32 | kEpiOut16 | kEpiBorder>());
                    else if (mask == (kEpiRes | kEpiOut32 | kEpiOut16 | kEpiBorder))
                        run(std::integral_constant<int, kEpiConst | kEpiRes | kEpiOut32 | kEpiOut16 | kEpiBorder>());
                    else if (mask == (kEpiRes | kEpiResB | kEpiOut32 | kEpiOut16 | kEpiBorder))
                        run(std::integral_constant<int, kEpiConst | kEpiRes | kEpiResB | kEpiOut32 | kEpiOut16 | kEpiBorder>());
                    else if (mask == (kEpiOut16 | kEpiBorder | kEpiTapBias))
                        run(std::integral_constant<int, kEpiConst | kEpiOut16 | kEpiBorder | kEpiTapBias>());
                    else if (mask == (kEpiRes | kEpiOut16 | kEpiBorder))
                        run(std::integral_constant<int, kEpiConst | kEpiRes | kEpiOut16 | kEpiBorder>());
                    else if (mask == (kEpiRes | kEpiOut16 | kEpiLo | kEpiHi2))
                        run(std::integral_constant<int, kEpiConst | kEpiRes | kEpiOut16 | kEpiLo | kEpiHi2>());
                    else if (mask == (kEpiRes | kEpiOut16 | kEpiLo))
                        run(std::integral_constant<int, kEpiConst | kEpiRes | kEpiOut16 | kEpiLo>());
                    else if (mask == kEpiOut16)
                        run(std::integral_constant<int, kEpiConst | kEpiOut16>());
                    else if (mask == (kEpiOut16 | kEpiLo))
                        run(std::integral_constant<int, kEpiConst | kEpiOut16 | kEpiLo>());
                    else
                        run(std::integral_constant<int, 0>());
                }
            }
            else
                run(std::integral_constant<int, 0>());
        } else if constexpr (EPI == EPI_CONVT) {
            const bool plain = p.act == ACT_NONE;
            if (plain && p.out16 && !p.out32 && !p.lo_off16)
                run(std::integral_constant<int, 4>());
            else if (plain && p.out32 && !p.out16)
                run(std::integral_constant<int, 5>());
            else if (plain && p.out32 && p.out16 && !p.lo_off16)
                run(std::integral_constant<int, 6>());
            else if (plain && p.out16 && !p.out32 && p.lo_off16)
                run(std::integral_constant<int, 7>());
            else
                run(std::integral_constant<int, 0>());
        } else {
            run(std::integral_constant<int, 0>());
        }
        if constexpr (EPI == EPI_STORE || EPI == EPI_CONVT) raise_overflow16<T>(p.status, amax16);
    }
}

// ---------------------------------------------------------------------------------------------------
// EPI_RESID_SCALE with the LayerNorm that follows it (GemmParams::ln_out16; vit.rs:165-169: x = x + ls(attn(norm1 x)),
// x = x + ls(mlp(norm2 x)) -- every residual update is followed by the norm of the next sublayer).  The stand-alone
// LayerNorm launch re-read the 89 MB token stream the epilogue had just written (21.5 us x 48 per step); here the
// updated rows are normalised while the workgroup still holds them in registers.
//
// A row's statistics need all N columns and a 352-row tile has 256 of them, so the N / 256 column tiles of a row tile
// -- workgroups that run in the same round of the persistent kernel, on the same XCD by the tile walk, but on other
// CUs -- exchange partial statistics through memory:
//   1. as before: x_new = (acc + bias) * gamma + x, stored pass by pass; the values stay in registers (the accumulators
//      they replace are dead), and each wave reduces (mean, M2) of its 64 columns per row over the 8 lanes that share
//      a row (two shuffle reductions: the sum, then the squares about the wave's own mean) into LDS;
//   2. 352 threads combine the four waves' partials per row (Chan's parallel form, fixed order) and publish the tile's
//      (mean, M2) as ONE 8-byte write-through store per row; every storing wave drains, the workgroup meets, one lane
//      adds 1 to the row tile's arrival counter (cdna_hip_programming.md Guideline 16, recipe R1);
//   3. that lane polls the counter until all N / 256 tiles have arrived (the count grows by N / 256 per launch and is
//      never reset: the target is the next multiple above the value the add returned), the workgroup meets again, and
//      the 352 threads read ALL column tiles' granules with write-through-coherent loads -- their own included, so
//      that every column tile combines the same numbers in the same order and a row's statistics do not depend on
//      which tile computes them -- into (mean, rstd) per row in LDS;
//   4. the retained values leave as (x - mean) * rstd * w + b, 16 bytes per lane and row.
// All workgroups of a round are resident (a persistent grid never exceeds what fits), so the wait cannot deadlock; it is
// bounded all the same and raises ME_STATUS_SYNC_TIMEOUT instead of hanging.  Statistics differ from the stand-alone
// kernel's in the order of their sums only (1e-7 relative); a row's result does not depend on the batch it is part of.
template <typename T, int MI, int NI, int TM, int TN, int BM>
__device__ __forceinline__ void resid_ln_epilogue(const GemmParams& p, f32x4 (&acc)[MI][NI], int m0, int n0, int wm,
                                                  int wn, int lane, int tid, char* epi_lds, char* shared, int m_lim,
                                                  int row_tile) {
    static_assert(TN == 64 && NI == 4 && BM == 2 * TM, "written for the 352 x 256 tile: 2 x 4 waves of 176 x 64");
    // Every per-lane constant of this epilogue (LDS addresses of the transposition, store offsets, row indices) is derived
    // from these two HERE, behind an opaque use: derived from the kernel's own `lane` they are loop-invariant, get
    // hoisted in front of the persistent tile loop, live through the K loop's 256 registers as spills, and come back
    // as a dozen dependent scratch reloads per tile (3 us) -- 30 integer instructions per tile are cheaper.
    asm volatile("" : "+v"(lane), "+v"(tid));
    constexpr int RS = TN * 4, ROWS = 16, GPR = 8, RPI = 8, ITERS = 2, NP = MI, CMASK = 15, WN = 4;
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const int Mrows = m_lim;
    const int frow = lane & 15, ncol = (lane >> 4) * 4;
    const int gc = lane % GPR, r0 = lane / GPR;
    const int n = n0 + wn * TN + gc * 8;  // N is a multiple of 256: every column exists
    const int seg = row_segment(p, m0);
    const float* bias = seg == 0 ? p.bias : (seg == 1 ? p.bias_s1 : p.bias_s2);
    const float* gamma = seg == 0 ? p.gamma : (seg == 1 ? p.gamma_s1 : p.gamma_s2);
    float4 cb[2], cg[2];
    cb[0] = *reinterpret_cast<const float4*>(bias + n), cb[1] = *reinterpret_cast<const float4*>(bias + n + 4);
    cg[0] = *reinterpret_cast<const float4*>(gamma + n), cg[1] = *reinterpret_cast<const float4*>(gamma + n + 4);
    asm volatile("" : "+v"(cb[0].x), "+v"(cb[0].y), "+v"(cb[0].z), "+v"(cb[0].w), "+v"(cb[1].x), "+v"(cb[1].y), "+v"(cb[1].z), "+v"(cb[1].w));
    asm volatile("" : "+v"(cg[0].x), "+v"(cg[0].y), "+v"(cg[0].z), "+v"(cg[0].w), "+v"(cg[1].x), "+v"(cg[1].y), "+v"(cg[1].z), "+v"(cg[1].w));
    const int mw = m0 + wm * TM;
    const int64_t wave_el = (int64_t)mw * p.ldc + n0 + wn * TN;
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(uniform_ptr((const char*)(p.res32 + wave_el))), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(uniform_ptr((const char*)(p.out32 + wave_el))), 0, 0x7fffffff, 0x00020000);
    const unsigned row_step = (unsigned)p.ldc * 4u;
    const unsigned loff = (unsigned)r0 * row_step + gc * 32u;
    const int rows_left = Mrows - mw - r0;
    constexpr unsigned kOut = 0x80000000u;
    auto voff = [&](int rt) { return rt < rows_left ? loff : kOut; };
    float2* stw = reinterpret_cast<float2*>(shared);                  // [BM][WN]: a wave's (mean, M2) of 64 columns
    float2* fin = reinterpret_cast<float2*>(shared + BM * WN * 8);    // [BM]: (mean, rstd) of the whole row
    // the updated values take the place of the accumulators they were computed from: xs(pass, it, h) = acc[pass][2 it + h]
    static_assert(NI == 2 * ITERS, "a pass's four accumulator tiles hold its 2 x 2 result vectors");
#define ME_XS(pass, it, h) acc[pass][2 * (it) + (h)]
    f32x4 res[ITERS][2];
    auto load_res = [&](int pass) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int rt = pass * ROWS + it * RPI;
            const unsigned so = (unsigned)rt * row_step;
            res[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, voff(rt), so, 0));
            res[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, voff(rt) + 16u, so, 0));
        }
    };
    load_res(0);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            *reinterpret_cast<f32x4*>(epi_lds + frow * RS + ((((j * 16 + ncol) >> 2) ^ frow) & CMASK) * 16) = acc[pass][j];
        f32x4 xn_[ITERS][2];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int rr = it * RPI + r0;
            const char* src = epi_lds + rr * RS;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((2 * gc + h) ^ rr) & CMASK) * 16);
                const float4 b4 = cb[h], g4 = cg[h];
                // same operation order as the reference: (xs * gamma) + residual
                const f32x4 xnew = f32x4{(v[0] + b4.x) * g4.x + res[it][h][0], (v[1] + b4.y) * g4.y + res[it][h][1],
                                        (v[2] + b4.z) * g4.z + res[it][h][2], (v[3] + b4.w) * g4.w + res[it][h][3]};
                xn_[it][h] = xnew;
            }
        }
        // (the pass's accumulators have all gone through LDS by now: their registers take the new values)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) ME_XS(pass, it, 0) = xn_[it][0], ME_XS(pass, it, 1) = xn_[it][1];
        if (pass + 1 < NP) load_res(pass + 1);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int rt = pass * ROWS + it * RPI;
            const unsigned so = (unsigned)rt * row_step;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ME_XS(pass, it, 0)), rout, voff(rt), so, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ME_XS(pass, it, 1)), rout, voff(rt) + 16u, so, 0);
        }
    }
    // ---- this wave's 64-column statistics of its 22 rows per lane: eight lanes share a row.  Reductions over those
    // eight lanes by DPP (quad_perm for the lane pairs at distance 1 and 2, row_shr:4 to bring the lower quad's sum to
    // the upper quad, row_shl:4 to hand the total back): vector instructions, no LDS round trip -- as ds_bpermute (what
    // __shfl_xor compiles to) they were 132 dependent LDS operations per lane inside the pass loop.  Outside the loop
    // the 22 rows are independent chains.
    auto dpp_add = [](float v, auto ctrl_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
        return v + __builtin_bit_cast(float, moved);
    };
    auto oct_sum = [&](float v) {  // the sum over the 8 lanes of a row, valid in all of them
        v = dpp_add(v, std::integral_constant<int, 0xB1>());   // quad_perm [1,0,3,2]
        v = dpp_add(v, std::integral_constant<int, 0x4E>());   // quad_perm [2,3,0,1]: every lane of a quad has the quad's sum
        const float up = dpp_add(v, std::integral_constant<int, 0x114>());  // row_shr:4: lanes 4-7 (12-15) add lanes 0-3 (8-11)
        const int down = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, up), 0x104, 0xf, 0xf, true);  // row_shl:4
        return (lane & 4) ? up : __builtin_bit_cast(float, down);
    };
#pragma unroll
    for (int pass = 0; pass < NP; ++pass)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const f32x4 a = ME_XS(pass, it, 0), b = ME_XS(pass, it, 1);
            const float sum = oct_sum(((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3])));
            const float mean = sum * (1.0f / 64.0f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = a[e] - mean, d1 = b[e] - mean;
                q = __builtin_fmaf(d0, d0, q), q = __builtin_fmaf(d1, d1, q);
            }
            q = oct_sum(q);
            if (gc == 0) stw[(wm * TM + pass * ROWS + it * RPI + r0) * WN + wn] = make_float2(mean, q);
        }
    // the LayerNorm weights of this lane's columns: requested now, needed behind the exchange (bias and gamma are done)
    {
        const float* lw = seg == 0 ? p.ln_w : (seg == 1 ? p.ln_w_s1 : p.ln_w_s2);
        const float* lb = seg == 0 ? p.ln_b : (seg == 1 ? p.ln_b_s1 : p.ln_b_s2);
        cb[0] = *reinterpret_cast<const float4*>(lb + n), cb[1] = *reinterpret_cast<const float4*>(lb + n + 4);
        cg[0] = *reinterpret_cast<const float4*>(lw + n), cg[1] = *reinterpret_cast<const float4*>(lw + n + 4);
    }
    // ---- the tile's statistics per row, published
    const int nbn = p.N / 256, ct = n0 / 256;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned gu32;
    gu64* granules = (gu64*)(p.ln_stats + (size_t)row_tile * nbn * BM);
    auto chan = [](float& n_a, float& mean_a, float& m2_a, float n_b, float mean_b, float m2_b) {
        const float nn = n_a + n_b, d = mean_b - mean_a, f = n_b / nn;
        mean_a = __builtin_fmaf(d, f, mean_a);
        m2_a = m2_a + m2_b + d * d * (n_a * f);
        n_a = nn;
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid < BM) {
        const float2 w0 = stw[tid * WN];
        float cn = 64.f, cm = w0.x, c2 = w0.y;
#pragma unroll
        for (int k = 1; k < WN; ++k) {
            const float2 wk = stw[tid * WN + k];
            chan(cn, cm, c2, 64.f, wk.x, wk.y);
        }
        const unsigned long long g = ((unsigned long long)__builtin_bit_cast(unsigned, c2) << 32) | __builtin_bit_cast(unsigned, cm);
        __hip_atomic_store(granules + (size_t)ct * BM + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
    __builtin_amdgcn_s_barrier();                      // ... before ONE lane signals for the workgroup
    asm volatile("" ::: "memory");
    if (nbn > 1) {
        if (tid == 0) {
            gu32* cnt = (gu32*)(p.ln_count + (size_t)row_tile * 16);
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (old / (unsigned)nbn + 1u) * (unsigned)nbn;
            // Bounded: the neighbours are resident and arrive within microseconds; ln_spin_limit polls (2^20 ~ 2 s unless the
            // caller set another bound) without them raise ME_STATUS_SYNC_TIMEOUT.  A workgroup that finds the bit already
            // raised does not wait at all -- the step is lost (the host re-runs it with stand-alone LayerNorm launches,
            // api.hip run_with_ln_fallback), and every further wait of it would only add its own bound to the damage.
            const unsigned limit = p.ln_spin_limit ? (unsigned)p.ln_spin_limit : (1u << 20);
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
                if ((spins & 63u) == 0 && p.status &&
                    (__hip_atomic_load((gu32*)p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2u))
                    break;
                if (spins > limit) {
                    if (p.status) atomicOr(p.status, 2u);
                    break;
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (tid < BM) {
        float cn = 0.f, cm = 0.f, c2 = 0.f;
        for (int k = 0; k < nbn; ++k) {  // every column tile, this one included, in the same order everywhere
            const unsigned long long g = __hip_atomic_load(granules + (size_t)k * BM + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float gm = __builtin_bit_cast(float, (unsigned)g), g2 = __builtin_bit_cast(float, (unsigned)(g >> 32));
            if (k == 0) cn = 256.f, cm = gm, c2 = g2;
            else chan(cn, cm, c2, 256.f, gm, g2);
        }
        const float var = c2 / cn;
        fin[tid] = make_float2(cm, 1.0f / sqrtf(var + p.ln_eps));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- the normalised rows, as the next GEMM's MX fp8 operand (GemmParams::out8 beside ln_out16: ME_DTYPE_FP8 contexts,
    // where the projection runs on this 16-bit tile and fc1 on the scaled fp8 MFMA): a lane's 8 columns are a quarter
    // of a 32-column MX block, the four lanes of a quad share its scale (layernorm_fp8_kernel's arithmetic: block
    // maximum -> e8m0 byte -> e4m3 elements); the two rows a lane holds per pass leave as ONE 16-byte store per lane
    // pair (the even lane stores the first row's 16 columns, the odd lane the second row's), scales as bytes in the
    // activation layout (mx_fp8.h a_scale_index)
    if (p.out8) {
        const __amdgpu_buffer_rsrc_t r8 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(uniform_ptr((const char*)(p.out8 + (int64_t)mw * p.N + n0 + wn * TN))), 0, 0x7fffffff, 0x00020000);
        const bool odd = (gc & 1) != 0;
        unsigned cur8 = (unsigned)(r0 + (odd ? RPI : 0)) * (unsigned)p.N + (unsigned)(gc & ~1) * 8u;
        int left8 = rows_left - (odd ? RPI : 0);
        int row8 = mw + r0;            // global row of this lane's first unit
        int left_s = rows_left;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            asm volatile("" : "+v"(cur8), "+v"(left8), "+v"(row8), "+v"(left_s));
            uint2 q[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int rt = pass * ROWS + it * RPI;
                const float2 ms = fin[wm * TM + rt + r0];
                float o[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 x = ME_XS(pass, it, h);
                    const float4 w4 = cg[h], b4 = cb[h];
                    o[4 * h + 0] = (x[0] - ms.x) * ms.y * w4.x + b4.x, o[4 * h + 1] = (x[1] - ms.x) * ms.y * w4.y + b4.y;
                    o[4 * h + 2] = (x[2] - ms.x) * ms.y * w4.z + b4.z, o[4 * h + 3] = (x[3] - ms.x) * ms.y * w4.w + b4.w;
                }
                float amax = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(o[e]));
                amax = fmaxf(amax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(amax), 0xB1, 0xF, 0xF, true)));
                amax = fmaxf(amax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(amax), 0x4E, 0xF, 0xF, true)));
                const unsigned sb = mx_scale_byte(amax);
                const float inv = mx_inv_scale(sb);
                q[it].x = pack_fp8x4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
                q[it].y = pack_fp8x4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
                if ((gc & 3) == 0 && left_s > it * RPI)
                    p.out8_scale[a_scale_index(row8 + it * RPI, n >> 5, p.out8_mt)] = (uint8_t)sb;
            }
            const uint2 send = odd ? q[0] : q[1];
            uint2 recv;
            recv.x = __builtin_amdgcn_update_dpp(0, send.x, 0xB1, 0xF, 0xF, true);
            recv.y = __builtin_amdgcn_update_dpp(0, send.y, 0xB1, 0xF, 0xF, true);
            const u32x4v ov = odd ? u32x4v{recv.x, recv.y, q[1].x, q[1].y} : u32x4v{q[0].x, q[0].y, recv.x, recv.y};
            __builtin_amdgcn_raw_buffer_store_b128(ov, r8, left8 > 0 ? cur8 : kOut, 0, 0);
            cur8 += (unsigned)ROWS * (unsigned)p.N, left8 -= ROWS, row8 += ROWS, left_s -= ROWS;
        }
        return;
    }
    const int64_t wave_el16 = (int64_t)mw * p.N + n0 + wn * TN;
    const __amdgpu_buffer_rsrc_t r16 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(uniform_ptr((const char*)((T*)p.ln_out16 + wave_el16))), 0, 0x7fffffff, 0x00020000);
    const unsigned row_step16 = (unsigned)p.N * 2u;
    const unsigned loff16 = (unsigned)r0 * row_step16 + gc * 16u;
    float amax16 = 0.f;
    // the store offset and the rows this lane has left walk with the loop (RPI rows per step) as RUNNING values: left
    // to itself the compiler computes all 22 offsets up front, spills them, and every step's reload waits -- vmcnt
    // counts stores -- for every store before it (1.2 us each: the whole 16-bit stream serialised, 26 us per tile)
    unsigned cur16 = loff16;
    int left16 = rows_left;
#pragma unroll
    for (int pass = 0; pass < NP; ++pass)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int rt = pass * ROWS + it * RPI;
            asm volatile("" : "+v"(cur16), "+v"(left16));
            const float2 ms = fin[wm * TM + rt + r0];
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 x = ME_XS(pass, it, h);
                const float4 w4 = cg[h], b4 = cb[h];
                o[4 * h + 0] = (x[0] - ms.x) * ms.y * w4.x + b4.x, o[4 * h + 1] = (x[1] - ms.x) * ms.y * w4.y + b4.y;
                o[4 * h + 2] = (x[2] - ms.x) * ms.y * w4.z + b4.z, o[4 * h + 3] = (x[3] - ms.x) * ms.y * w4.w + b4.w;
            }
            track_amax16<T>(amax16, o, true);
            typedef T v8 __attribute__((ext_vector_type(8)));
            v8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (T)o[e];
            // The row offset goes into the VECTOR offset, not the scalar one: the next iteration's conversions rewrite
            // this store's data registers at once, and with a register in the soffset field the compiler's hazard pass
            // assumes that a wide store has read its data by then -- on gfx950 it has not (the second dword came out
            // overwritten for a quarter of the lanes).  With soffset 0 the pass inserts the wait states itself.
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ov), r16, left16 > 0 ? cur16 : kOut, 0, 0);
            cur16 += (unsigned)RPI * row_step16, left16 -= RPI;
        }
    raise_overflow16<T>(p.status, amax16);
}
#undef ME_XS

// sched_group_barrier wants literal counts: compile-time recursion over the fragment groups
template <int G, int NI, int g>
struct SchedPin {
    // NV > 0 additionally asks for NV VMEM reads spread over the groups (unused: measured slower)
    template <int NV>
    static __device__ __forceinline__ void run() {
        constexpr int reads = (g + 2 < G ? 1 : 0) + (g < NI ? 1 : 0);
        if constexpr (reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, reads, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x8, NI, 0);                                // MFMA
        constexpr int per = (NV + G - 1) / G;  // VMEM per group
        if constexpr (g * per < NV)
            __builtin_amdgcn_sched_group_barrier(0x20, (g + 1) * per <= NV ? per : NV - g * per, 0);
        SchedPin<G, NI, g + 1>::template run<NV>();
    }
};
template <int G, int NI>
struct SchedPin<G, NI, G> {
    template <int NV>
    static __device__ __forceinline__ void run() {}
};

// Persistent kernel: gridDim.x workgroups (as many as are resident at once) each walk the tiles
// bid, bid + gridDim.x, ... (the same XCD every time).  The K loop runs as ONE stream across tile
// boundaries: the last slab iteration of a tile stages the first slab of the next tile, so neither the
// next tile's first-load latency nor the drain of this tile's epilogue stores is exposed.
template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI, bool SPLITK = false>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_ITERS = (BM / 8) / NW, B_ITERS = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split over waves");
    static_assert(A_ITERS >= 1 && B_ITERS >= 1, "tile too small for the wave count");
    // the epilogue's transposition scratch must fit in ONE stage: the other one already holds the
    // next tile's first slab
    constexpr int MI_CH = epi_mi_chunk(MI, TN, NW, STAGE_BYTES);
    static_assert(NW * 16 * MI_CH * (TN * 4) <= STAGE_BYTES, "epilogue scratch exceeds a stage");
    typedef typename MfmaOp<T>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    // SPLITK: a work item is (tile, K range); item vb = tile * S + s, so a tile's S items sit on neighbouring workgroups
    const int S = SPLITK ? p.split_k : 1;
    const int nitems = ntiles * S;

    // ---- per-lane source pointers for the staging loads of one tile ----
    const int srow = lane >> 3;  // row within an 8-row LDS-DMA piece
    const int sslot = lane & 7;  // 16-byte slot written by this lane
    struct TileSrc {
        int m0, n0;
        int tile, kb, ke;  // SPLITK: the tile and the slab range [kb, ke) of this item
        const char* a[A_ITERS];
        const char* w[B_ITERS];
    };
    const int nk = p.K / 64;
    auto setup = [&](TileSrc& t, int vb) {
        if constexpr (SPLITK) {
            t.tile = vb / S;
            const int ks = vb - t.tile * S;
            t.kb = (int)((int64_t)ks * nk / S), t.ke = (int)((int64_t)(ks + 1) * nk / S);
        } else {
            t.tile = vb, t.kb = 0, t.ke = nk;
        }
        tile_origin<BM, BN>(p, t.tile, ntiles, t.m0, t.n0);
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int row = (i * NW + wave) * 8 + srow;
            const int chunk = sslot ^ ((row >> 1) & 7);
            int gm = t.m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            if constexpr (AMODE == A_PLAIN) {
                t.a[i] = (const char*)p.A + ((int64_t)gm * p.lda) * 2 + chunk * 16;
            } else {
                t.a[i] = (const char*)p.A + conv_pixel(p, gm) * p.Cin * 2 + chunk * 16;
            }
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int row = (i * NW + wave) * 8 + srow;
            const int chunk = sslot ^ ((row >> 1) & 7);
            int gn = t.n0 + row;
            gn = gn < p.N ? gn : p.N - 1;
            t.w[i] = segment_weights(p, t.m0) + ((int64_t)gn * p.K) * 2 + chunk * 16;
        }
    };

    // Staging of slab kt of tile t into LDS buffer buf (slabs are staged in order: kt = 0 rewinds the taps)
    SlabWalk<AMODE> walk;
    walk.init(p);
    auto stage_a_offset = [&](int kt) -> int64_t { return walk.next(p, kt); };
    auto stage_piece = [&](const TileSrc& t, int piece, int64_t a_koff, int64_t w_koff, int buf) {
        char* la = smem + buf * STAGE_BYTES;
        if (piece < A_ITERS)
            glds16(t.a[piece] + a_koff, la + (piece * NW + wave) * 1024);
        else
            glds16(t.w[piece - A_ITERS] + w_koff, la + A_BYTES + ((piece - A_ITERS) * NW + wave) * 1024);
    };
    auto stage = [&](const TileSrc& t, int kt, int buf) {
        const int64_t a_koff = stage_a_offset(kt), w_koff = walk.w_off(kt);
#pragma unroll
        for (int i = 0; i < A_ITERS + B_ITERS; ++i) stage_piece(t, i, a_koff, w_koff, buf);
    };

    // fragment read addressing (byte offsets inside a stage)
    const int frow = lane & 15;
    const int fswz = frow >> 1;
    const int fslot0 = ((lane >> 4) ^ fswz) * 16;        // k-substep 0
    const int fslot1 = (((lane >> 4) + 4) ^ fswz) * 16;  // k-substep 1
    const int a_rd = (wm * TM + frow) * 128;
    const int b_rd = A_BYTES + (wn * TN + frow) * 128;

    int vb = blockIdx.x;
    TileSrc cur, nxt;
    setup(cur, vb);
    stage(cur, cur.kb, 0);
    const bool dyn = !SPLITK && p.queue != nullptr && ntiles > (int)gridDim.x;  // dynamic tile order (TileQueue)
    TileQueue tq;
    tq.init(p.queue, ntiles);
    int next_vb = vb + (int)gridDim.x;
    // second tile: drawn behind the first staging loads, handed over through the still idle buffer 1
    if (dyn && tid == 0) *reinterpret_cast<int*>(smem + STAGE_BYTES) = tq.draw();
    __syncthreads();
    if (dyn) {
        next_vb = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem + STAGE_BYTES));
        __syncthreads();
    }
    int buf = 0;
    [[maybe_unused]] int stamp_i = 0;
#ifdef ME_GEMM_STAMPS
#define ME_STAMP()                                                                       \
    do {                                                                                 \
        if (p.stamps && tid == 0 && stamp_i < 16)                                        \
            p.stamps[(size_t)blockIdx.x * 16 + stamp_i++] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    // per-wave phase accounting (shader clocks): [0] LDS-DMA issue, [1] fragment reads + MFMAs,
    // [2] s_waitcnt vmcnt(0), [3] barrier; written behind the tile stamps
    unsigned long long ph[4] = {0, 0, 0, 0}, pt = 0;
#define ME_PHASE(i)                                              \
    do {                                                         \
        __builtin_amdgcn_sched_barrier(0);                       \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                       \
        if ((i) >= 0) ph[(i) < 0 ? 0 : (i)] += t_ - pt;          \
        pt = t_;                                                 \
    } while (0)
#else
#define ME_STAMP() do {} while (0)
#define ME_PHASE(i) do {} while (0)
#endif
    ME_STAMP();

    while (true) {
        const bool has_next = next_vb >= 0 && next_vb < nitems;
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int kt_end = SPLITK ? cur.ke : nk;
        for (int kt = SPLITK ? cur.kb : 0; kt < kt_end; ++kt) {
            ME_PHASE(-1);
            if (kt + 1 < kt_end) {
                stage(cur, kt + 1, buf ^ 1);
            } else if (has_next) {
                setup(nxt, next_vb);
                stage(nxt, SPLITK ? nxt.kb : 0, buf ^ 1);
            }
            ME_PHASE(0);
            const char* sb = smem + buf * STAGE_BYTES;
            {
                // Software-pipelined fragment feed.  The 2*MI "groups" (k-substep kk, m-tile i) each
                // issue NI MFMAs on one A fragment; the A fragment of group g+2 and (early on) the W
                // fragments of the second k-substep are read from LDS while group g's MFMAs run, so
                // after the first two groups no MFMA waits for an LDS round trip.  (Spreading the next
                // slab's LDS-DMA instructions between the groups as well was measured and lost.)
                constexpr int G = 2 * MI;
                frag af[G], wf[2][NI];
                auto rd_a = [&](int g) {
                    return *reinterpret_cast<const frag*>(sb + a_rd + (g % MI) * 2048 +
                                                          (g < MI ? fslot0 : fslot1));
                };
                auto rd_w = [&](int kk, int j) {
                    return *reinterpret_cast<const frag*>(sb + b_rd + j * 2048 +
                                                          (kk == 0 ? fslot0 : fslot1));
                };
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[0][j] = rd_w(0, j);
                af[0] = rd_a(0);
                if (G > 1) af[1] = rd_a(1);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (g + 2 < G) af[g + 2] = rd_a(g + 2);
                    if (g < NI) wf[1][g] = rd_w(1, g);
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[g % MI][j] = MfmaOp<T>::run(wf[g < MI ? 0 : 1][j], af[g], acc[g % MI][j]);
                }
                // pin that order: (reads of the group, then its NI MFMAs) x G
                __builtin_amdgcn_sched_group_barrier(0x100, NI + (G > 1 ? 2 : 1), 0);
                SchedPin<G, NI, 0>::template run<0>();
            }
#ifdef ME_GEMM_STAMPS
            ME_PHASE(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ME_PHASE(2);
#endif
            __syncthreads();  // slab in buf^1 has landed (vmcnt(0)); everyone is done reading buf
            ME_PHASE(3);
            buf ^= 1;
        }
        ME_STAMP();
        // buf now holds the next tile's first slab; buf^1 was consumed last and is the scratch
        char* scratch0 = smem + (buf ^ 1) * STAGE_BYTES;
        int drawn = -1;  // lane 0 of wave 0: the tile after next
        bool finish = true;
        if constexpr (SPLITK) {
            // the partial in accumulator layout: wave-instruction (i, j) of the workgroup writes NW KiB contiguously.
            // Coherence as in resid_ln_epilogue: write-through stores and cache-bypassing loads at agent scope (sc1) instead of
            // fences -- a release / acquire fence pair writes back and invalidates the XCD's whole L2 per workgroup, which made
            // the split launches 2 - 3 x slower than the whole-K ones they replace.
            constexpr int NT = NW * 64;
            f32x4* part = reinterpret_cast<f32x4*>(p.splitk_ws) + (size_t)(cur.tile * S) * (MI * NI * NT) + tid;
            f32x4* mine = part + (size_t)(vb - cur.tile * S) * (MI * NI * NT);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(mine + (i * NI + j) * NT), "v"(acc[i][j]) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its write-through stores ...
            // (... whose data registers must stay untouched until then: dead after the asm for all the compiler knows, it took them
            // for the next store's address one instruction later -- a wide store reads its data after issue)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" : : "v"(acc[i][j]));
            __syncthreads();                                   // ... before ONE lane signals for the workgroup
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(p.splitk_cnt + cur.tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *reinterpret_cast<volatile int*>(scratch0) = old == (unsigned)(S - 1);
            }
            __syncthreads();
            finish = *reinterpret_cast<volatile int*>(scratch0) != 0;
            __syncthreads();  // (the word is epilogue scratch from here on)
            if (finish) {
                if (tid == 0) __hip_atomic_store(p.splitk_cnt + cur.tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int s2 = 0; s2 < S; ++s2) {  // split order, this item's own partial included
                    f32x4 v[MI][NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            asm volatile("global_load_dwordx4 %0, %1, off sc1"
                                         : "=v"(v[i][j])
                                         : "v"(part + (size_t)s2 * (MI * NI * NT) + (i * NI + j) * NT)
                                         : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            asm volatile("" : "+v"(v[i][j]));  // (the loads are invisible to the compiler's waitcnt pass: use behind the wait)
                            acc[i][j] += v[i][j];
                        }
                }
            }
        }
        if (finish)
        gemm_epilogue<T, EPI, MI, NI, TM, TN, MI_CH>(
            p, acc, cur.m0, cur.n0, wm, wn, lane, scratch0 + wave * (16 * MI_CH * (TN * 4)), [&]() {
                if (dyn && has_next && tid == 0) drawn = tq.draw();
            });
        ME_STAMP();
        if (!has_next) {
#ifdef ME_GEMM_STAMPS
            if (p.stamps && lane == 0)
                for (int i = 0; i < 4; ++i)
                    p.stamps[(size_t)gridDim.x * 16 + ((size_t)blockIdx.x * NW + wave) * 4 + i] = ph[i];
#endif
            if (dyn) tq.leave();
            break;
        }
        if (dyn && tid == 0) *reinterpret_cast<int*>(scratch0) = drawn;  // wave 0's own scratch
        // the scratch is restaged by the next tile's first iteration: its LDS reads must be done
        __syncthreads();
        cur = nxt;
        vb = next_vb;
        if (dyn) {
            next_vb = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(scratch0));
            __syncthreads();
        } else {
            next_vb = vb + (int)gridDim.x;
        }
    }
}

#undef ME_STAMP
#undef ME_PHASE

// ---------------------------------------------------------------------------------------------------
// Two-group ("ping-pong") form of the 8-wave kernel.
//
// Phase accounting of the kernel above (tools/gemm_stamps.py) shows where its time goes: the eight
// waves issue their LDS-DMA together at the top of a slab, the texture path moves 64 B/clk per CU, so
// the 64 KiB of a 256x256 slab hold every wave in "issue" for ~0.5 us while the matrix pipe idles; the
// waves of a SIMD then share the pipe for the MFMAs.  DMA time and MFMA time ADD.  Here the two waves of
// a SIMD (w and w + 4) take complementary roles instead:
//   group 0 (waves 0-3) stages the ACTIVATION rows of slab s+1 at the top of slab s, then runs its MFMAs;
//   group 1 (waves 4-7) runs half of its MFMAs first, stages the WEIGHT rows of slab s+2, then the rest.
// While one wave of a SIMD is held by the texture path the other owns the matrix pipe.  The weights are
// requested two slabs ahead, so the later issue point costs no latency at the closing wait; that takes a
// three-slot ring for the weights beside the two-slot ring of the activations: 2*BM*128 + 3*BN*128 B
// (160 KiB for the 256x256 tile: all of a CU's LDS).  One barrier per slab as before; group 1 closes a
// slab with a counted vmcnt (its newest DMA group may stay in flight), group 0 with vmcnt(0).
// Requires K >= 128.
// LNF: the residual epilogue also writes the LayerNorm of the rows it updates (resid_ln_epilogue above)
template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI, int WSLOTS = 3, bool LNF = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
    static_assert(!LNF || (BM > 256 && EPI == EPI_RESID_SCALE), "LayerNorm fusion: the 352-row tile's residual epilogue");
    static_assert(WSLOTS == 3 || WSLOTS == 2, "weight ring of three slots (two slabs ahead) or two (one ahead)");
    static_assert(WM * WN == 8, "two groups of four waves");
    static_assert(BM / WM >= BN / WN, "the first k-substep's MI groups prefetch the NI weight fragments");
    constexpr int HW = 4;  // waves per group
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int W_RING = 2 * A_BYTES;
    constexpr int A_IT = (BM / 8) / HW, B_IT = (BN / 8) / HW;
    constexpr int IT = A_IT > B_IT ? A_IT : B_IT;
    static_assert((BM / 8) % HW == 0 && (BN / 8) % HW == 0, "tile rows must split over a group");
    constexpr int MI_CH = epi_mi_chunk(MI, TN, HW, A_BYTES < B_BYTES ? A_BYTES : B_BYTES);
    constexpr int SCR = 16 * MI_CH * (TN * 4);  // epilogue scratch per wave
    static_assert(HW * SCR <= A_BYTES && HW * SCR <= B_BYTES, "epilogue scratch exceeds a ring slot");
    typedef typename MfmaOp<T>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2, gw = wave & 3;  // waves w and w + 4 share a SIMD
    const int wm = wave / WN, wn = wave % WN;
    // TALL (the 352-row tile): row tiles laid out per row segment (seg_tile_rows), DMA source offsets kept as
    // {first, step, clamp} instead of one register per piece (22 VGPRs of the two tiles in flight become 4)
    constexpr bool TALL = BM > 256;
    static_assert(!TALL || AMODE == A_PLAIN, "the tall tile's affine source offsets need row-major activations");
    // LNF: the column tiles of a row tile must run in the SAME round of the persistent loop (they wait for one another's
    // statistics: a tile that waited for a later round's tile would at best serialise the rounds and at worst wait for
    // a workgroup that waits for it).  The walk above guarantees that only when an XCD's run is a whole number of
    // patches, so the fused launch has its own: virtual tile vb = (x, j) with x = vb & 7 its XCD and j = vb >> 3 its
    // place there; column tile j % nbn of row tile (j / nbn) * 8 + x.  A round gives an XCD 32 consecutive j -- 32 / nbn
    // whole row tiles, each with all its column tiles on neighbouring workgroups; an XCD's 32 tiles of a round are
    // again 8 row panels x 4 weight panels.  Row tiles past the last one (the padding of the last round) do not exist.
    const int lnf_nbn = (p.N + BN - 1) / BN, lnf_nbm = TALL ? seg_row_tiles<BM>(p.M, p.seg1, p.seg2) : 0;
    auto lnf_row_tile = [&](int vb) { return ((vb >> 3) / lnf_nbn) * 8 + (vb & 7); };
    const int ntiles = LNF ? ((lnf_nbm + 7) / 8) * 8 * lnf_nbn
                           : (TALL ? seg_row_tiles<BM>(p.M, p.seg1, p.seg2) : (p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if constexpr (LNF) {
        if (lnf_row_tile(blockIdx.x) >= lnf_nbm) return;  // uniform; this workgroup's later tiles do not exist either
    }

    const int srow = lane >> 3, sslot = lane & 7;
    struct Src {
        int m0, n0, m_lim, row_tile;
        // this wave's DMA sources, activation rows (group 0) or weight rows (group 1): a uniform base (the
        // tile's first row) and per-lane 32-bit byte offsets from it -- the kernel has no VGPR to spare
        // for 64-bit pointers
        const char* base;
        unsigned s[TALL ? 2 : IT];  // TALL: {offset of the first piece, largest offset that stays inside the operand}
    };
    auto pixel_of = [&](int gm) -> int64_t { return conv_pixel(p, gm); };
    auto setup = [&](Src& t, int vb) {
        if constexpr (LNF) {
            t.row_tile = lnf_row_tile(vb);
            t.n0 = ((vb >> 3) % lnf_nbn) * BN;
            seg_tile_rows<BM>(p, t.row_tile, t.m0, t.m_lim);
        } else {
            tile_origin<BM, BN, TALL>(p, vb, ntiles, t.m0, t.n0, &t.m_lim, &t.row_tile);
        }
        if constexpr (TALL) {
            // piece i of a lane: row (i * HW + gw) * 8 + srow of the tile, always the same 16-byte chunk position
            // (the swizzle repeats every 16 rows) -- offsets are affine in i; rows beyond the operand's last row
            // read that last row instead (their products are never stored)
            const int row = gw * 8 + srow;
            const unsigned chunk16 = (unsigned)(sslot ^ ((row >> 1) & 7)) * 16u;
            if (group == 0) {
                t.base = (const char*)p.A + (int64_t)t.m0 * p.lda * 2;
                const int last = p.M - 1 - t.m0;  // >= 0
                t.s[0] = (unsigned)row * (unsigned)(p.lda * 2) + chunk16;
                t.s[1] = last >= BM ? 0xffffffffu : (unsigned)last * (unsigned)(p.lda * 2) + chunk16;
            } else {
                t.base = segment_weights(p, t.m0) + (int64_t)t.n0 * p.K * 2;
                const int last = p.N - 1 - t.n0;
                t.s[0] = (unsigned)row * (unsigned)(p.K * 2) + chunk16;
                t.s[1] = last >= BN ? 0xffffffffu : (unsigned)last * (unsigned)(p.K * 2) + chunk16;
            }
            return;
        }
        if (group == 0) {
            int64_t base_el;  // element offset of the tile's first row (uniform)
            if constexpr (AMODE == A_PLAIN)
                base_el = (int64_t)t.m0 * p.lda;
            else
                base_el = pixel_of(t.m0) * p.Cin;
            t.base = (const char*)p.A + base_el * 2;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int row = (i * HW + gw) * 8 + srow;
                const int chunk = sslot ^ ((row >> 1) & 7);
                int gm = t.m0 + row;
                gm = gm < p.M ? gm : p.M - 1;
                int64_t el;
                if constexpr (AMODE == A_PLAIN)
                    el = (int64_t)gm * p.lda;
                else
                    el = pixel_of(gm) * p.Cin;
                t.s[i] = (unsigned)((el - base_el) * 2) + chunk * 16;
            }
        } else {
            t.base = segment_weights(p, t.m0) + (int64_t)t.n0 * p.K * 2;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int row = (i * HW + gw) * 8 + srow;
                const int chunk = sslot ^ ((row >> 1) & 7);
                int gn = t.n0 + row;
                gn = gn < p.N ? gn : p.N - 1;
                t.s[i] = (unsigned)((int64_t)(gn - t.n0) * p.K * 2) + chunk * 16;
            }
        }
    };

    const int nk = p.K / 64;
    SlabWalk<AMODE> walk;
    walk.init(p);
    auto stage_a_offset = [&](int kt) -> int64_t { return walk.next(p, kt); };
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    auto stage_T = [&](const Src& t, int kt, int slot) {  // group 0
        const char* base = uniform_ptr(t.base + stage_a_offset(kt));
        const unsigned dst = smem_base + slot * A_BYTES + gw * 1024;
        if constexpr (TALL) {
            const unsigned step = (unsigned)(HW * 8) * (unsigned)(p.lda * 2);
            unsigned first = t.s[0];
            asm volatile("" : "+v"(first));  // recomputed per slab (two VALU per piece): hoisted, the A_IT offsets spill
#pragma unroll
            for (int i = 0; i < A_IT; ++i) glds16_raw(base, min(first + i * step, t.s[1]), dst + i * (HW * 1024));
            return;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16_raw(base, t.s[i], dst + i * (HW * 1024));
    };
    auto stage_W = [&](const Src& t, int kt, int slot) {  // group 1
        const char* base = uniform_ptr(t.base + walk.w_off(kt));
        const unsigned dst = smem_base + W_RING + slot * B_BYTES + gw * 1024;
        if constexpr (TALL) {
            const unsigned step = (unsigned)(HW * 8) * (unsigned)(p.K * 2);
            unsigned first = t.s[0];
            asm volatile("" : "+v"(first));
#pragma unroll
            for (int i = 0; i < B_IT; ++i) glds16_raw(base, min(first + i * step, t.s[1]), dst + i * (HW * 1024));
            return;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16_raw(base, t.s[i], dst + i * (HW * 1024));
    };

    const int frow = lane & 15;
    const int fswz = frow >> 1;
    const int fslot0 = ((lane >> 4) ^ fswz) * 16;
    const int fslot1 = (((lane >> 4) + 4) ^ fswz) * 16;
    const int a_rd = (wm * TM + frow) * 128;
    const int b_rd = (wn * TN + frow) * 128;

    // The matrix pipe goes to the higher priority, then to the older wave.  Left alone, group 0 (older)
    // wins it for all 64 of its MFMAs and group 1 reaches its mid-slab DMA only after group 0 has finished
    // -- the DMA then runs beside nothing (phase accounting: group 1's first k-substep 1610 cycles, group 0
    // 1170 cycles at the barrier).  With static priority group 1 computes first, and its DMA issue falls
    // into the time group 0 still has MFMAs to run.
    if (group == 1) __builtin_amdgcn_s_setprio(1);
    int vb = blockIdx.x;
    Src cur, nxt;
    setup(cur, vb);
    const bool dyn = WSLOTS == 3 && p.queue != nullptr && ntiles > (int)gridDim.x;  // dynamic tile order (TileQueue)
    TileQueue tq;
    tq.init(p.queue, ntiles);
    int next_vb = vb + (int)gridDim.x;
    if (group == 0) {
        stage_T(cur, 0, 0);
        // second tile: drawn behind the first staging loads, handed over through the still empty third
        // weight slot (its first DMA comes mid-slab 0, after the extra barrier below)
        if (dyn && tid == 0) *reinterpret_cast<int*>(smem + W_RING + 2 * B_BYTES) = tq.draw();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        stage_W(cur, 0, 0);
        if constexpr (WSLOTS == 3) stage_W(cur, 1, 1);
        if constexpr (WSLOTS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (B_IT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (B_IT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (dyn) {
        next_vb = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem + W_RING + 2 * B_BYTES));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    int ts = 0, ws = 0;  // ring slots of the slab being consumed
    [[maybe_unused]] int stamp_i = 0;
#ifdef ME_GEMM_STAMPS
#define ME_STAMP()                                                                       \
    do {                                                                                 \
        if (p.stamps && tid == 0 && stamp_i < 16)                                        \
            p.stamps[(size_t)blockIdx.x * 16 + stamp_i++] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    // [0] top-of-slab DMA issue (group 0), [1] first k-substep, [2] mid-slab DMA issue (group 1),
    // [3] second k-substep, [4] closing vmcnt wait, [5] barrier
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, pt = 0;
#define ME_PHASE(i)                                                 \
    do {                                                            \
        __builtin_amdgcn_sched_barrier(0);                          \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                          \
        if ((i) >= 0) ph[(i) < 0 ? 0 : (i)] += t_ - pt;             \
        pt = t_;                                                    \
    } while (0)
#else
#define ME_STAMP() do {} while (0)
#define ME_PHASE(i) do {} while (0)
#endif
    ME_STAMP();

    while (true) {
        const bool has_next = next_vb >= 0 && next_vb < ntiles && (!LNF || lnf_row_tile(next_vb) < lnf_nbm);
        if (has_next) setup(nxt, next_vb);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            ME_PHASE(-1);
            if (group == 0) {
                if (kt + 1 < nk)
                    stage_T(cur, kt + 1, ts ^ 1);
                else if (has_next)
                    stage_T(nxt, 0, ts ^ 1);
            }
            ME_PHASE(0);
            const char* sa = smem + ts * A_BYTES + a_rd;
            const char* sw = smem + W_RING + ws * B_BYTES + b_rd;
            constexpr int G = 2 * MI;
            frag af[G], wf[2][NI];
            auto rd_a = [&](int g) {
                return *reinterpret_cast<const frag*>(sa + (g % MI) * 2048 + (g < MI ? fslot0 : fslot1));
            };
            auto rd_w = [&](int kk, int j) {
                return *reinterpret_cast<const frag*>(sw + j * 2048 + (kk == 0 ? fslot0 : fslot1));
            };
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[0][j] = rd_w(0, j);
            af[0] = rd_a(0);
            af[1] = rd_a(1);
            // first k-substep
#pragma unroll
            for (int g = 0; g < MI; ++g) {
                if (g + 2 < G) af[g + 2] = rd_a(g + 2);
                if (g < NI) wf[1][g] = rd_w(1, g);
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[g % MI][j] = MfmaOp<T>::run(wf[0][j], af[g], acc[g % MI][j]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, NI + 2, 0);
            SchedPin<MI, NI, 0>::template run<0>();
            ME_PHASE(1);
            bool newer = false;  // group 1: a DMA group younger than the one the next slab needs
            if (group == 1) {
                if constexpr (WSLOTS == 3) {
                    const int w2 = ws == 0 ? 2 : ws - 1;  // (ws + 2) % 3
                    if (kt + 2 < nk) {
                        stage_W(cur, kt + 2, w2);
                        newer = true;
                    } else if (has_next) {
                        stage_W(nxt, kt + 2 - nk, w2);
                        newer = true;
                    }
                } else {  // two slots: the next slab's weights, into the slot the slab before this one has left
                    if (kt + 1 < nk)
                        stage_W(cur, kt + 1, ws ^ 1);
                    else if (has_next)
                        stage_W(nxt, 0, ws ^ 1);
                }
            }
            ME_PHASE(2);
            // second k-substep
#pragma unroll
            for (int g = MI; g < G; ++g) {
                if (g + 2 < G) af[g + 2] = rd_a(g + 2);
                if (g < NI) wf[1][g] = rd_w(1, g);
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[g % MI][j] = MfmaOp<T>::run(wf[1][j], af[g], acc[g % MI][j]);
            }
            SchedPin<MI, NI, 0>::template run<0>();
            ME_PHASE(3);
            if (newer) {
                if constexpr (B_IT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (B_IT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ME_PHASE(4);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ME_PHASE(5);
            ts ^= 1;
            ws = WSLOTS == 2 ? (ws ^ 1) : (ws == 2 ? 0 : ws + 1);
        }
        ME_STAMP();
        // ts / ws now name the next tile's first slab; the slots consumed last are the scratch
        char* scratch0 = smem + (ts ^ 1) * A_BYTES;  // group 0's region; its first word doubles as hand-off
        int drawn = -1;                              // lane 0 of wave 0: the tile after next
        {
            const int wprev = WSLOTS == 2 ? (ws ^ 1) : (ws == 0 ? 2 : ws - 1);
            char* scr = (group == 0 ? scratch0 : smem + W_RING + wprev * B_BYTES) + gw * SCR;
            if constexpr (LNF) {
                // the statistics of the tile's rows go through the rest of the activation slot that holds group 0's scratch
                static_assert(HW * SCR + BM * (WN * 8 + 8) <= A_BYTES, "row statistics beside the epilogue scratch");
                resid_ln_epilogue<T, MI, NI, TM, TN, BM>(p, acc, cur.m0, cur.n0, wm, wn, lane, tid, scr, scratch0 + HW * SCR,
                                                         cur.m_lim, cur.row_tile);
            } else
            gemm_epilogue<T, EPI, MI, NI, TM, TN, MI_CH, true>(p, acc, cur.m0, cur.n0, wm, wn, lane, scr, [&]() {
                if (dyn && has_next && tid == 0) drawn = tq.draw();
            }, TALL ? cur.m_lim : -1);
        }
        ME_STAMP();
        if (!has_next) {
#ifdef ME_GEMM_STAMPS
            if (p.stamps && lane == 0)
                for (int i = 0; i < 6; ++i)
                    p.stamps[(size_t)gridDim.x * 16 + ((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
            if (dyn) tq.leave();
            break;
        }
        if (dyn && tid == 0) *reinterpret_cast<int*>(scratch0) = drawn;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the scratch slots are restaged by the next tile's first slab
        asm volatile("" ::: "memory");
        cur = nxt;
        vb = next_vb;
        if (dyn) {
            next_vb = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(scratch0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // everyone has read the word before slab 0 restages the slot
            asm volatile("" ::: "memory");
        } else {
            next_vb = vb + (int)gridDim.x;
        }
    }
}
#undef ME_STAMP
#undef ME_PHASE

template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI, int WSLOTS = 3, bool LNF = false>
void gemm_launch_pp(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = (2 * BM + WSLOTS * BN) * 128;
    static_assert(smem <= 160 * 1024, "the tile's rings exceed a CU's LDS");
    ME_CHECK(p.K >= 128, ME_ERR_BAD_SHAPE, "gemm: the two-group kernel needs K >= 128 (K = %d)", p.K);
    auto kern = gemm_pp_kernel<T, BM, BN, WM, WN, AMODE, EPI, WSLOTS, LNF>;
    static PerDeviceOnce once;
    // (workgroups per CU) << 12 | CUs of the device
    const int occ = per_device_once(once, [&](int dev) {
        ME_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int per_cu = 0, cus = 0;
        ME_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, smem));
        ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        return ((per_cu < 1 ? 1 : per_cu) << 12) | (cus < 1 ? 1 : (cus > 4095 ? 4095 : cus));
    });
    // resident workgroups: what fits on the CUs the launch stream may use (all of the device's unless it carries a CU mask)
    int cus = occ & 4095;
    if (p.cu_granted > 0 && p.cu_granted < cus) cus = p.cu_granted;
    int resident = (occ >> 12) * cus;
    resident -= resident % 8;
    if (resident < 8) resident = 8;
    if (p.resident_out) {
        *p.resident_out = resident;
        return;
    }
    int64_t ntiles = (BM > 256 ? (int64_t)seg_row_tiles<BM>(p.M, p.seg1, p.seg2) : cdiv(p.M, BM)) * cdiv(p.N, BN);
    if (LNF) ntiles = cdiv((int64_t)seg_row_tiles<BM>(p.M, p.seg1, p.seg2), 8) * 8 * cdiv(p.N, BN);  // the kernel's padded walk
    ME_CHECK(ntiles > 0 && ntiles < (1ll << 31), ME_ERR_BAD_SHAPE, "gemm grid %lld out of range",
             (long long)ntiles);
    int64_t grid = ntiles < resident ? ntiles : resident;
    if (LNF) {
        // a round is a whole number of row tiles per XCD, and every workgroup of it must be resident (they wait for one another)
        const int64_t unit = 8 * cdiv(p.N, BN);
        // (pipeline.hip ln_fusable asks gemm_lnf_resident first and takes the stand-alone LayerNorm otherwise)
        ME_CHECK(resident >= unit, ME_ERR_HIP, "gemm: the fused LayerNorm needs %lld resident workgroups (%d fit)", (long long)unit, resident);
        grid -= grid % unit;
    }
    // diagnostic (tools/dual_stream_probe.py): cap the persistent grid so that two launches on two streams share
    // the chip instead of the first one taking every CU until it ends.  On the fused LayerNorm launch a cap that is not a
    // whole number of rounds' units puts column tiles of one row tile into different rounds: the waits run into their
    // bound and raise ME_STATUS_SYNC_TIMEOUT -- which is how tests/test_gpu_pipeline.py forces the fallback path.
    static const int grid_limit = getenv("ME_GEMM_GRID_LIMIT") ? atoi(getenv("ME_GEMM_GRID_LIMIT")) : 0;
    if (grid_limit >= 8 && grid > grid_limit) grid = grid_limit - grid_limit % 8;
    if (!LNF && p.grid_cap >= 8 && grid > p.grid_cap) grid = p.grid_cap - p.grid_cap % 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, p);
    ME_HIP(hipGetLastError());
}

template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI, bool SPLITK = false>
void gemm_launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<T, BM, BN, WM, WN, AMODE, EPI, SPLITK>;
    static PerDeviceOnce once;  // workgroups that fit on the chip at once (per instantiation and device)
    const int resident = per_device_once(once, [&](int dev) {
        ME_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int per_cu = 0, cus = 0;
        ME_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, WM * WN * 64, smem));
        ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        int r = (per_cu < 1 ? 1 : per_cu) * cus;
        r -= r % 8;  // whole XCD rounds: workgroup b always lands on XCD b % 8
        return r < 8 ? 8 : r;
    });
    int64_t ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    if constexpr (SPLITK) {
        ME_CHECK(p.split_k > 1 && p.split_k <= p.K / 64 && p.splitk_ws && p.splitk_cnt, ME_ERR_BAD_ARG, "gemm: split-K %d of K = %d", p.split_k, p.K);
        ntiles *= p.split_k;
    }
    ME_CHECK(ntiles > 0 && ntiles < (1ll << 31), ME_ERR_BAD_SHAPE, "gemm grid %lld out of range",
             (long long)ntiles);
    const int64_t grid = ntiles < resident ? ntiles : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), smem, stream, p);
    ME_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// The guide's 256x256 "8-phase" schedule (cdna_hip_programming.md section 5, T3 + T4) as one more tile configuration,
// built for the A/B that VERDICT r2 item 4 asks for: the same tile, operand roles, LDS image and epilogue as the
// two-group kernel above, but a K tile consumed in FOUR phases of 16 MFMAs per wave (one 64 x 32 quadrant of the
// wave's 128 x 64 block over the whole 64-deep slab), each phase = {fragment reads + ONE half-tile of LDS-DMA |
// barrier | MFMAs | barrier}, the two wave groups one barrier apart so that one group's matrix section runs beside
// the other's load section, counted vmcnt once per K tile, raw s_barrier.
//
// Layout.  Two 64 KiB K-tile buffers (A rows at +0, W rows at +32 KiB; the 128-byte-row image with the chunk
// swizzle of the kernels above), 32 KiB of epilogue scratch of its own behind them: 160 KiB.  Wave (wr, wc) owns
// rows wr*64 + [0, 64) of EACH 128-row half of the tile (so that the first two phases read only the A half 0 and
// free it early) and columns wc*64 + [0, 64); its weight fragments are read once per K tile and kept.
//   phase 1: reads W(n-tiles 0,1) + A half 0 (12 x ds_read_b128)   MFMAs (mh 0, nh 0)   stages W half 1 of tile k+1
//   phase 2: reads W(n-tiles 2,3)            (4)                   MFMAs (mh 0, nh 1)   stages A half 1 of tile k+1
//   phase 3: reads A half 1                  (8)                   MFMAs (mh 1, nh 1)   stages A half 0 of tile k+2
//   phase 4: --                                                    MFMAs (mh 1, nh 0)   stages W half 0 of tile k+2
// Every half is restaged two phases or more after its last read (WAR), and read one phase or more after the counted
// wait that retires it, the wait standing in front of phase 4's FIRST barrier so that it also covers the other
// group, which runs one barrier behind (RAW: "read a staged buffer one phase AFTER the wait that retires it").
// The K stream runs on across tile boundaries as in the kernels above.
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_8ph_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256;
    constexpr int BUF = 65536, W_OFF = 32768, SCR_OFF = 2 * BUF, SCR = 4096;
    typedef typename MfmaOp<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;  // waves w and w + 4 share a SIMD: the two groups are wr = 0 / 1
    const int ntiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int nk = p.K / 64;

    const int srow = lane >> 3, sslot = lane & 7;
    struct Src {
        int m0, n0;
        const char *a, *w;        // uniform: the tile's first activation row / weight row
        unsigned ao[2][2], wo[2][2];  // per-lane byte offsets of this wave's two pieces of each 128-row half
    };
    auto setup = [&](Src& t, int vb) {
        tile_origin<BM, BN>(p, vb, ntiles, t.m0, t.n0);
        t.a = (const char*)p.A + (int64_t)t.m0 * p.lda * 2;
        t.w = segment_weights(p, t.m0) + (int64_t)t.n0 * p.K * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int row = h * 128 + (pc * 8 + wave) * 8 + srow;
                const int chunk = sslot ^ ((row >> 1) & 7);
                int gm = t.m0 + row, gn = t.n0 + row;
                gm = gm < p.M ? gm : p.M - 1;
                gn = gn < p.N ? gn : p.N - 1;
                t.ao[h][pc] = (unsigned)((int64_t)(gm - t.m0) * p.lda * 2) + chunk * 16;
                t.wo[h][pc] = (unsigned)((int64_t)(gn - t.n0) * p.K * 2) + chunk * 16;
            }
    };
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    int vb = blockIdx.x;
    Src cur, nxt;
    // one half-tile: this wave's two 1 KiB pieces (8 rows each) of half h of operand `op` (0 = A, 1 = W) of slab kt
    // of the current tile or (of_next) the next one.  The two tiles' fields are chosen by value -- a pointer to one
    // of the two structs would put both into scratch memory, and scratch loads count in vmcnt
    auto stage = [&](bool of_next, int kt, int b, int op, int h) {
        const char* tb = op ? (of_next ? nxt.w : cur.w) : (of_next ? nxt.a : cur.a);
        const char* base = uniform_ptr(tb + (int64_t)kt * 128);
        const unsigned dst = smem_base + b * BUF + (op ? W_OFF : 0) + (h * 128 + wave * 8) * 128;
        const unsigned o0 = op ? (of_next ? nxt.wo[h][0] : cur.wo[h][0]) : (of_next ? nxt.ao[h][0] : cur.ao[h][0]);
        const unsigned o1 = op ? (of_next ? nxt.wo[h][1] : cur.wo[h][1]) : (of_next ? nxt.ao[h][1] : cur.ao[h][1]);
        glds16_raw(base, o0, dst);
        glds16_raw(base, o1, dst + 64 * 128);
    };

    const int frow = lane & 15, fswz = frow >> 1;
    const int fslot[2] = {((lane >> 4) ^ fswz) * 16, (((lane >> 4) + 4) ^ fswz) * 16};
    const int a_rd = (wr * 64 + frow) * 128;            // + mh * 128 rows + i * 16 rows
    const int w_rd = W_OFF + (wc * 64 + frow) * 128;    // + j * 16 rows

    setup(cur, vb);
    int next_vb = vb + (int)gridDim.x;
    bool has_next = next_vb < ntiles;
    nxt = cur;
    if (has_next) setup(nxt, next_vb);
    // stream position d K tiles ahead of (cur, kt): 0 = in the current tile, 1 = in the next one, -1 = past the end;
    // k = its slab
    auto ahead = [&](int kt, int d, int& k) -> int {
        k = kt + d;
        if (k < nk) return 0;
        k -= nk;
        return has_next ? 1 : -1;
    };
    // prologue: K tile 0 whole, then the first two halves of K tile 1 (nk >= 2)
    stage(false, 0, 0, 0, 0), stage(false, 0, 0, 1, 0), stage(false, 0, 0, 1, 1), stage(false, 0, 0, 0, 1);
    stage(false, 1, 1, 0, 0), stage(false, 1, 1, 1, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) {  // group 1 runs one barrier behind group 0 from here on
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    int b = 0;  // buffer of the K tile being consumed

    while (true) {
        f32x4 acc[2][4][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            const char* sb = smem + b * BUF;
            frag af[2][4], wf[2][4];
            int k1, k2;
            const int t1 = ahead(kt, 1, k1), t2 = ahead(kt, 2, k2);
            auto mfma_quadrant = [&](int mh, int nh) {
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[mh][i][nh * 2 + j] = MfmaOp<T>::run(wf[ks][nh * 2 + j], af[ks][i], acc[mh][i][nh * 2 + j]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            };
            auto read_w = [&](int j0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = j0; j < j0 + 2; ++j)
                        wf[ks][j] = *reinterpret_cast<const frag*>(sb + w_rd + j * 2048 + fslot[ks]);
            };
            auto read_a = [&](int mh) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        af[ks][i] = *reinterpret_cast<const frag*>(sb + a_rd + (mh * 128 + i * 16) * 128 + fslot[ks]);
            };
            // ---- phase 1
            read_w(0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(0);
            if (t1 >= 0) stage(t1 == 1, k1, b ^ 1, 1, 1);
            mfma_quadrant(0, 0);
            // ---- phase 2
            read_w(2);
            if (t1 >= 0) stage(t1 == 1, k1, b ^ 1, 0, 1);
            mfma_quadrant(0, 1);
            // ---- phase 3
            read_a(1);
            if (t2 >= 0) stage(t2 == 1, k2, b, 0, 0);
            mfma_quadrant(1, 1);
            // ---- phase 4: K tile k+1 must have landed (its last half was issued in phase 2); the two halves of
            // tile k+2 issued since may stay in flight
            if (t2 >= 0) {
                stage(t2 == 1, k2, b, 1, 0);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            mfma_quadrant(1, 0);
            b ^= 1;
        }
        // epilogue of this tile: the two 128-row halves, each wave's 64 x 64 block of them, through the wave's own
        // scratch (never restaged: the stream of the next tile keeps running underneath)
        char* scr = smem + SCR_OFF + wave * SCR;
        gemm_epilogue<T, EPI, 4, 4, 64, 64, 1, true>(p, acc[0], cur.m0, cur.n0, wr, wc, lane, scr);
        gemm_epilogue<T, EPI, 4, 4, 64, 64, 1, true>(p, acc[1], cur.m0 + 128, cur.n0, wr, wc, lane, scr);
        if (!has_next) break;
        cur = nxt;
        vb = next_vb;
        next_vb = vb + (int)gridDim.x;
        has_next = next_vb < ntiles;
        if (has_next) setup(nxt, next_vb);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // group 1's extra barrier of the prologue, matched
}

template <typename T, int EPI>
void gemm_launch_8ph(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 160 * 1024;
    ME_CHECK(p.K >= 128, ME_ERR_BAD_SHAPE, "gemm: the 8-phase kernel needs K >= 128 (K = %d)", p.K);
    auto kern = gemm_8ph_kernel<T, EPI>;
    static PerDeviceOnce once;
    const int resident = per_device_once(once, [&](int dev) {
        ME_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int per_cu = 0, cus = 0;
        ME_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, smem));
        ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        int r = (per_cu < 1 ? 1 : per_cu) * cus;
        r -= r % 8;
        return r < 8 ? 8 : r;
    });
    const int64_t ntiles = cdiv(p.M, 256) * cdiv(p.N, 256);
    ME_CHECK(ntiles > 0 && ntiles < (1ll << 31), ME_ERR_BAD_SHAPE, "gemm grid %lld out of range", (long long)ntiles);
    const int64_t grid = ntiles < resident ? ntiles : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, p);
    ME_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// 3x3 convolution (stride 1) with the activation tile staged ONCE per 64 input channels: "halo" form of the
// two-group kernel.
//
// What bounds the kernels above is what a CU can stage into LDS per K slab: (BM + BN) x 128 bytes through the
// LDS-DMA path at ~50 GB/s per CU -- 64 KiB per slab for the 256x256 tile, next to 2048 matrix-pipe cycles.  The
// implicit-GEMM convolution stages the same activation pixels nine times, once per tap.  Here the M tile is a 16 x 16
// block of output pixels; for each 64-channel slab of the input its 18 x 18 halo (324 pixels x 128 bytes = 41 KiB)
// is staged once and the nine taps read it at shifted rows: per K slab 32 KiB of weights + 4.6 KiB of activations
// instead of 64 KiB, and nothing but the weights (L2-resident) is fetched more than 1.27 times.
//
// K order: input-channel slab outermost, taps inside (the weights are packed [Cout][tap][Cin], so a slab is the 128
// bytes at (tap * Cin + c * 64) * 2 of a weight row).  LDS: two halo slots of 328 rows + two weight slots of 256 rows
// (148 KiB); the halo image is [pixel hy * 18 + hx][64 channels] with chunk c of a pixel at slot c ^ halo_swizzle(hx), so
// the B-operand fragment of output row ty, tap (ky, kx) is the 16 consecutive rows (ty + ky) * 18 + kx + (0..15).
// Roles: group 0 (waves 0-3) stages the weight rows of slab s+1 at the top of slab s; group 1 (waves 4-7, priority 1)
// computes first and stages, mid-slab, two pieces per wave of the NEXT channel slab's halo during taps 0-5 (48 pieces
// for the 41 needed), retired by its vmcnt(0) at the end of tap 8.  One barrier per slab.
// Requires KH = KW = 3, stride 1, out_H and out_W multiples of 16, Cin a multiple of 64, N a multiple of 4.
// Chunk swizzle of the halo image: chunk c of the pixel in halo column hx (0 .. 17) sits at slot c ^ halo_swizzle(hx).
// The fragment reads of tap column kx touch the 16 consecutive columns kx .. kx + 15, and the usual (hx >> 1) & 7 is
// conflict-free only for kx = 0 (28.6 % of the LDS cycles of the first build were bank conflicts,
// profiles/r03_pmc_kernels.json).  This table -- one 3-bit value per column PAIR, found by search -- gives every
// 16-lane group of ds_read_b128 sixteen distinct (row parity, slot) pairs for kx = 0, 1 AND 2, in both k-substeps.
__device__ __forceinline__ int halo_swizzle(int hx) { return (int)((0xcb5888u >> (3 * (hx >> 1))) & 7u); }

// TH: pixel rows of the tile (16, or 12: a 384 x 384 map is 576 tiles of 16 x 16 -- 2.25 rounds of 256 workgroups, the
// third a quarter full -- and 768 tiles of 12 x 16: three full rounds of three quarters the work)
// BN: output channels per tile (256, or 128 for the head's 256 -> 128 convolution: a wave then owns 32 channels)
template <typename T, int EPI, int TH = 16, int BN = 256>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(const GemmParams p) {
    static_assert(TH == 16 || TH == 12, "tile rows");
    static_assert(BN == 256 || BN == 128, "tile channels");
    constexpr int HW = 4, WN = 4;
    constexpr int MI = TH / 2, TM = 16 * MI, TN = BN / WN, NI = TN / 16;  // a wave: half of the tile's pixel rows x 64 (32) channels
    constexpr int HALO_PX = (TH + 2) * 18;                     // 324 / 252 halo pixels
    constexpr int HALO_ROWS = (HALO_PX + 7) / 8 * 8, HALO_BYTES = HALO_ROWS * 128, HALO_PIECES = HALO_ROWS / 8;  // 41 / 32
    constexpr int W_BYTES = BN * 128, W_BASE = 2 * HALO_BYTES;
    constexpr int B_IT = (BN / 8) / HW;  // 8 weight pieces per wave of group 0 per slab
    constexpr int MI_CH = 2;
    constexpr int SCR = 16 * MI_CH * (TN * 4);  // 8 KiB of epilogue scratch per wave
    static_assert(HW * SCR <= HALO_BYTES && HW * SCR <= W_BYTES, "epilogue scratch exceeds a slot");
    typedef typename MfmaOp<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2, gw = wave & 3;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = p.out_W >> 4, tiles_y = p.out_H / TH;
    const int mtiles = p.M / (TH * 16);                  // TH x 16 pixel tiles (M = B * out_H * out_W)
    const int nbn = (p.N + BN - 1) / BN;
    const int ntiles = mtiles * nbn;
    const int nc = p.Cin / 64;                            // channel slabs; nine taps each
    const int srow = lane >> 3, sslot = lane & 7;

    struct Tile {
        int mt, n0;             // pixel-tile index, first output channel
        const char* a;          // uniform: bordered input pixel (y0, x0) of the tile's image = halo pixel (0, 0)
        const char* w;          // uniform: weight row n0
        unsigned wo[B_IT];      // group 0: per-lane byte offsets of its weight pieces
    };
    auto setup = [&](Tile& t, int vb) {
        // the super-row / XCD walk of tile_origin over (pixel tiles) x (column tiles)
        GemmParams q = p;
        int m0;
        tile_origin<TH * 16, BN>(q, vb, ntiles, m0, t.n0);
        t.mt = m0 / (TH * 16);
        const int b = t.mt / (tiles_x * tiles_y);
        const int rem = t.mt - b * (tiles_x * tiles_y);
        const int y0 = (rem / tiles_x) * TH, x0 = (rem - (rem / tiles_x) * tiles_x) * 16;
        t.a = (const char*)p.A + (((int64_t)b * p.in_Hp + y0) * p.in_Wp + x0) * p.Cin * 2;
        t.w = (const char*)p.W + (int64_t)t.n0 * p.K * 2;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = (i * HW + gw) * 8 + srow;
            const int chunk = sslot ^ ((row >> 1) & 7);
            int gn = t.n0 + row;
            gn = gn < p.N ? gn : p.N - 1;
            t.wo[i] = (unsigned)((int64_t)(gn - t.n0) * p.K * 2) + chunk * 16;
        }
    };
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    // weight rows of slab (c, tap) into weight slot `slot` (group 0)
    auto stage_w = [&](const Tile& t, int c, int tap, int slot) {
        const char* base = uniform_ptr(t.w + ((int64_t)tap * p.Cin + c * 64) * 2);
        const unsigned dst = smem_base + W_BASE + slot * W_BYTES + gw * 1024;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16_raw(base, t.wo[i], dst + i * (HW * 1024));
    };
    // halo piece `pc` (8 pixels) of channel slab c into halo slot `slot` (group 1)
    auto stage_halo_piece = [&](const Tile& t, int c, int pc, int slot) {
        int r = pc * 8 + srow;
        r = r < HALO_PX ? r : HALO_PX - 1;                  // the 4 pad rows repeat the last pixel (never read)
        const int hy = (r * 3641) >> 16;                    // r / 18 for r < 324
        const int hx = r - hy * 18;
        const int chunk = sslot ^ halo_swizzle(hx);         // the halo image's swizzle goes by the pixel's COLUMN
        const unsigned off = (unsigned)(((hy * p.in_Wp + hx) * p.Cin + c * 64) * 2 + chunk * 16);
        glds16_raw(uniform_ptr(t.a), off, smem_base + slot * HALO_BYTES + pc * 1024);
    };

    // fragment addressing
    const int frow = lane & 15, q4 = lane >> 4;
    const int fswz = frow >> 1;
    const int wslot0 = (q4 ^ fswz) * 16, wslot1 = ((q4 + 4) ^ fswz) * 16;
    const int w_rd = (wn * TN + frow) * 128;
    // A fragment of output row ty, tap (ky, kx), k-substep kk: halo row (ty + ky) * 18 + kx + frow, chunk (q4 + 4 kk) ^
    // swizzle -- and the halo image swizzles by the pixel's column hx = kx + frow alone (the row parity that decides
    // the bank half is hx & 1 too, 18 being even), so the lane's byte offset is (ty + ky) * 2304 + a_col[kx] (^ 64 for
    // the second k-substep): three per-lane constants, the rest immediates
    int a_col[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int hx = kx + frow;
        a_col[kx] = hx * 128 + ((q4 ^ halo_swizzle(hx)) << 4);
    }

    if (group == 1) __builtin_amdgcn_s_setprio(1);
    int vb = blockIdx.x;
    Tile cur, nxt;
    setup(cur, vb);
    int next_vb = vb + (int)gridDim.x;
    // prologue: halo of channel slab 0 (group 1: 11 pieces per wave cover 41) and weight slab (0, tap 0) (group 0)
    if (group == 0) {
        stage_w(cur, 0, 0, 0);
    } else {
        for (int k = 0; k < 11; ++k) {
            const int pc = k * 4 + gw;
            if (pc < HALO_PIECES) stage_halo_piece(cur, 0, pc, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int ws = 0, hs = 0;  // weight slot of the slab being consumed, halo slot of its channel slab

    while (true) {
        const bool has_next = next_vb < ntiles;
        if (has_next) setup(nxt, next_vb);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < nc; ++c) {
            // who comes after this channel slab: the next one of this tile, the first of the next tile, or nobody
            const bool next_c_here = c + 1 < nc;
            const bool halo_follows = next_c_here || has_next;
            int ky = 0, kx = 0;
            for (int tap = 0; tap < 9; ++tap) {
                // ---- top of the slab: group 0 stages the weights of the slab after this one
                if (group == 0) {
                    if (tap < 8)
                        stage_w(cur, c, tap + 1, ws ^ 1);
                    else if (next_c_here)
                        stage_w(cur, c + 1, 0, ws ^ 1);
                    else if (has_next)
                        stage_w(nxt, 0, 0, ws ^ 1);
                }
                const char* sw = smem + W_BASE + ws * W_BYTES + w_rd;
                const char* sa = smem + hs * HALO_BYTES;
                const char* sa_tap = sa + (wm * MI + ky) * (18 * 128);
                const int a0 = kx == 0 ? a_col[0] : (kx == 1 ? a_col[1] : a_col[2]);
                const int a1 = a0 ^ 64;
                constexpr int G = 2 * MI;
                frag af[G], wf[2][NI];
                auto rd_a = [&](int g) {
                    return *reinterpret_cast<const frag*>(sa_tap + (g < MI ? a0 : a1) + (g % MI) * (18 * 128));
                };
                auto rd_w = [&](int kk, int j) {
                    return *reinterpret_cast<const frag*>(sw + j * 2048 + (kk == 0 ? wslot0 : wslot1));
                };
                // ---- first k-substep: the fragment feed of the two-group kernel (A fragment two groups ahead, the
                // second substep's weight fragments early on), pinned
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[0][j] = rd_w(0, j);
                af[0] = rd_a(0);
                af[1] = rd_a(1);
#pragma unroll
                for (int g = 0; g < MI; ++g) {
                    if (g + 2 < G) af[g + 2] = rd_a(g + 2);
                    if (g < NI) wf[1][g] = rd_w(1, g);
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[g % MI][j] = MfmaOp<T>::run(wf[0][j], af[g], acc[g % MI][j]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, NI + 2, 0);
                SchedPin<MI, NI, 0>::template run<0>();
                // ---- mid-slab: group 1 stages two pieces of the next channel slab's halo (taps 0 .. 5)
                if (group == 1 && halo_follows && tap < 6) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int pc = tap * 8 + gw * 2 + k;
                        if (pc < HALO_PIECES) {
                            if (next_c_here) stage_halo_piece(cur, c + 1, pc, hs ^ 1);
                            else stage_halo_piece(nxt, 0, pc, hs ^ 1);
                        }
                    }
                }
                // ---- second k-substep
#pragma unroll
                for (int g = MI; g < G; ++g) {
                    if (g + 2 < G) af[g + 2] = rd_a(g + 2);
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[g % MI][j] = MfmaOp<T>::run(wf[1][j], af[g], acc[g % MI][j]);
                }
                SchedPin<MI, NI, 0>::template run<0>();
                // ---- close the slab: group 0's weights of the next slab have landed; group 1's halo pieces must have
                // landed before the first slab that reads them (the one after tap 8)
                if (group == 0 || tap == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                ws ^= 1;
                if (++kx == 3) kx = 0, ++ky;
            }
            hs ^= 1;
        }
        // ws / hs now name the next tile's first weight slab / halo; the slots consumed last are the scratch
        {
            char* scr = (group == 0 ? smem + (hs ^ 1) * HALO_BYTES : smem + W_BASE + (ws ^ 1) * W_BYTES) + gw * SCR;
            gemm_epilogue<T, EPI, MI, NI, TM, TN, MI_CH, true, NoHook, 0, true, TH>(p, acc, cur.mt, cur.n0, wm, wn, lane, scr);
        }
        if (!has_next) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the scratch slots are restaged by the next tile's first slab
        asm volatile("" ::: "memory");
        cur = nxt;
        vb = next_vb;
        next_vb = vb + (int)gridDim.x;
    }
}

template <typename T, int EPI, int TH = 16, int BN = 256>
void conv_halo_launch(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * (((TH + 2) * 18 + 7) / 8 * 8) * 128 + 2 * BN * 128;
    ME_CHECK(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.out_H % TH == 0 && p.out_W % 16 == 0 && p.Cin % 64 == 0 &&
                 p.M % (TH * 16) == 0,
             ME_ERR_BAD_SHAPE, "conv (halo tile): %dx%d stride %d on %dx%d, Cin %d", p.KH, p.KW, p.stride, p.out_H, p.out_W,
             p.Cin);
    auto kern = conv_halo_kernel<T, EPI, TH, BN>;
    static PerDeviceOnce once;
    const int resident = per_device_once(once, [&](int dev) {
        ME_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int per_cu = 0, cus = 0;
        ME_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, smem));
        ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        int r = (per_cu < 1 ? 1 : per_cu) * cus;
        r -= r % 8;
        return r < 8 ? 8 : r;
    });
    const int64_t ntiles = (int64_t)(p.M / (TH * 16)) * cdiv(p.N, BN);
    ME_CHECK(ntiles > 0 && ntiles < (1ll << 31), ME_ERR_BAD_SHAPE, "conv grid %lld out of range", (long long)ntiles);
    const int64_t grid = ntiles < resident ? ntiles : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, p);
    ME_HIP(hipGetLastError());
}

// vmcnt(rem * P) for a run-time rem in [0, R]
template <int R, int P>
struct WaitSlabs {
    static __device__ __forceinline__ void run(int rem) {
        if (rem >= R)
            wait_vmcnt<R * P>();
        else
            WaitSlabs<R - 1, P>::run(rem);
    }
};
template <int P>
struct WaitSlabs<0, P> {
    static __device__ __forceinline__ void run(int) { wait_vmcnt<0>(); }
};

// ---------------------------------------------------------------------------------------------------
// Deep-ring form for problems that cannot fill the chip (the single-window ViTs: M = 577; the
// low-resolution decoder levels).  One tile per workgroup; with a 64x64 tile a slab is only 8 MFMAs per
// wave, so the two-buffer kernel above spends every slab waiting out one full LDS-DMA latency (0.8 us per
// slab: 51 us for the M = 577 fc2).  Here NST ring slots keep NST-1 slabs in flight behind counted
// vmcnt waits; one barrier per slab publishes the landed slab and frees the slot consumed before it.
template <typename T, int BM, int BN, int WM, int WN, int NST, int AMODE, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_ring_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_IT = (BM / 8) / NW, B_IT = (BN / 8) / NW;
    constexpr int P = A_IT + B_IT;  // LDS-DMA instructions per wave per slab
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split over waves");
    static_assert(NST >= 3 && (NST - 2) * P <= 63, "ring depth vs the 6-bit vmcnt");
    constexpr int MI_CH = epi_mi_chunk(MI, TN, NW, NST * STAGE_BYTES);
    typedef typename MfmaOp<T>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    int m0, n0;
    tile_origin<BM, BN>(p, blockIdx.x, ntiles, m0, n0);

    // DMA sources: uniform base of the tile's first row + per-lane 32-bit byte offsets
    const int srow = lane >> 3, sslot = lane & 7;
    auto pixel_of = [&](int gm) -> int64_t { return conv_pixel(p, gm); };
    int64_t a_base_el;
    if constexpr (AMODE == A_PLAIN)
        a_base_el = (int64_t)m0 * p.lda;
    else
        a_base_el = pixel_of(m0) * p.Cin;
    const char* a_base = (const char*)p.A + a_base_el * 2;
    const char* w_base = segment_weights(p, m0) + (int64_t)n0 * p.K * 2;
    unsigned a_off[A_IT], w_off[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int chunk = sslot ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        int64_t el;
        if constexpr (AMODE == A_PLAIN)
            el = (int64_t)gm * p.lda;
        else
            el = pixel_of(gm) * p.Cin;
        a_off[i] = (unsigned)((el - a_base_el) * 2) + chunk * 16;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int chunk = sslot ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        w_off[i] = (unsigned)((int64_t)(gn - n0) * p.K * 2) + chunk * 16;
    }

    const int nk = p.K / 64;
    SlabWalk<AMODE> walk;
    walk.init(p);
    auto stage_a_offset = [&](int kt) -> int64_t { return walk.next(p, kt); };
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    auto stage = [&](int kt, int slot) {
        const char* ab = uniform_ptr(a_base + stage_a_offset(kt));
        const char* wb = uniform_ptr(w_base + walk.w_off(kt));
        const unsigned dst = smem_base + slot * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16_raw(ab, a_off[i], dst + i * (NW * 1024));
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16_raw(wb, w_off[i], dst + A_BYTES + i * (NW * 1024));
    };

    const int frow = lane & 15;
    const int fswz = frow >> 1;
    const int fslot0 = ((lane >> 4) ^ fswz) * 16;
    const int fslot1 = (((lane >> 4) + 4) ^ fswz) * 16;
    const int a_rd = (wm * TM + frow) * 128;
    const int b_rd = A_BYTES + (wn * TN + frow) * 128;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s);

    int slot = 0, fill = NST - 1;  // slot being consumed; slot the next staged slab goes to
    for (int kt = 0; kt < nk; ++kt) {
        // slabs kt+1 .. min(kt+NST-2, nk-1) may stay in flight
        WaitSlabs<NST - 2, P>::run(nk - 1 - kt);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NST - 1 < nk) stage(kt + NST - 1, fill);
        const char* sb = smem + slot * STAGE_BYTES;
        {
            constexpr int G = 2 * MI;
            frag af[G], wf[2][NI];
            auto rd_a = [&](int g) {
                return *reinterpret_cast<const frag*>(sb + a_rd + (g % MI) * 2048 + (g < MI ? fslot0 : fslot1));
            };
            auto rd_w = [&](int kk, int j) {
                return *reinterpret_cast<const frag*>(sb + b_rd + j * 2048 + (kk == 0 ? fslot0 : fslot1));
            };
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[0][j] = rd_w(0, j);
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[1][j] = rd_w(1, j);
#pragma unroll
            for (int g = 0; g < G; ++g) af[g] = rd_a(g);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[g % MI][j] = MfmaOp<T>::run(wf[g < MI ? 0 : 1][j], af[g], acc[g % MI][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers before the next barrier
        slot = slot + 1 == NST ? 0 : slot + 1;
        fill = fill + 1 == NST ? 0 : fill + 1;
    }
    __builtin_amdgcn_s_barrier();  // every wave is done with the ring: it becomes the epilogue scratch
    asm volatile("" ::: "memory");
    gemm_epilogue<T, EPI, MI, NI, TM, TN, MI_CH>(p, acc, m0, n0, wm, wn, lane,
                                                 smem + wave * (16 * MI_CH * (TN * 4)));
}

template <typename T, int BM, int BN, int WM, int WN, int NST, int AMODE, int EPI>
void gemm_launch_ring(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = NST * (BM + BN) * 128;
    auto kern = gemm_ring_kernel<T, BM, BN, WM, WN, NST, AMODE, EPI>;
    static PerDeviceOnce once;
    per_device_once(once, [&](int) {
        ME_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        return 1;
    });
    const int64_t ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    ME_CHECK(ntiles > 0 && ntiles < (1ll << 31), ME_ERR_BAD_SHAPE, "gemm grid %lld out of range",
             (long long)ntiles);
    hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(WM * WN * 64), smem, stream, p);
    ME_HIP(hipGetLastError());
}

// One instantiation set per (dtype, amode, epi); defined in gemm_*.hip.
template <typename T, int AMODE, int EPI>
void gemm_dispatch(const GemmParams& p, int cfg, hipStream_t stream);

#define ME_GEMM_DISPATCH_BODY(T, AMODE, EPI)                                              \
    template <>                                                                           \
    void gemm_dispatch<T, AMODE, EPI>(const GemmParams& p, int cfg, hipStream_t stream) { \
        switch (cfg) {                                                                    \
            case 0: gemm_launch_pp<T, 256, 256, 2, 4, AMODE, EPI>(p, stream); break;       \
            case 1:                                                                       \
                if constexpr (EPI == EPI_STORE) {                                         \
                    if (p.split_k > 1) {                                                  \
                        gemm_launch_cfg<T, 128, 128, 2, 2, AMODE, EPI, true>(p, stream);  \
                        break;                                                            \
                    }                                                                     \
                }                                                                         \
                gemm_launch_cfg<T, 128, 128, 2, 2, AMODE, EPI>(p, stream);                \
                break;                                                                    \
            case 2: gemm_launch_cfg<T, 64, 64, 2, 2, AMODE, EPI>(p, stream); break;       \
            case 3: gemm_launch_cfg<T, 160, 128, 2, 2, AMODE, EPI>(p, stream); break;     \
            case 4: gemm_launch_ring<T, 64, 64, 2, 2, 6, AMODE, EPI>(p, stream); break;    \
            case 5: gemm_launch_pp<T, 192, 256, 2, 4, AMODE, EPI>(p, stream); break;       \
            case 7: /* short tail tile behind whole rounds of config 0 (pipeline.hip resid_all) */ \
                if constexpr (EPI == EPI_STORE || EPI == EPI_RESID_SCALE)                \
                    gemm_launch_pp<T, 96, 256, 1, 8, AMODE, EPI>(p, stream);              \
                else                                                                      \
                    fail(ME_ERR_BAD_ARG, "gemm: the 96-row tile takes store / residual epilogues only"); \
                break;                                                                    \
            case 8: /* the same, three ring slots: two slabs in flight behind a short tile's few MFMAs */ \
                if constexpr (EPI == EPI_STORE || EPI == EPI_RESID_SCALE)                \
                    gemm_launch_ring<T, 128, 256, 2, 4, 3, AMODE, EPI>(p, stream);        \
                else                                                                      \
                    fail(ME_ERR_BAD_ARG, "gemm: the 128-row ring tile takes store / residual epilogues only"); \
                break;                                                                    \
            case 9:                                                                       \
                if constexpr (AMODE == A_CONV && EPI == EPI_STORE)                        \
                    conv_halo_launch<T, EPI>(p, stream);                                  \
                else                                                                      \
                    fail(ME_ERR_BAD_ARG, "gemm: the halo tile is a 3x3 convolution");     \
                break;                                                                    \
            case 12: /* the halo tile with 128 output channels (the head's 256 -> 128 convolution at 768 x 768) */ \
                if constexpr (AMODE == A_CONV && EPI == EPI_STORE)                        \
                    conv_halo_launch<T, EPI, 16, 128>(p, stream);                         \
                else                                                                      \
                    fail(ME_ERR_BAD_ARG, "gemm: the halo tile is a 3x3 convolution");     \
                break;                                                                    \
            case 11: /* the halo tile on 12 x 16 pixels (three full rounds of a 384 x 384 map instead of 2.25) */ \
                if constexpr (AMODE == A_CONV && EPI == EPI_STORE)                        \
                    conv_halo_launch<T, EPI, 12>(p, stream);                              \
                else                                                                      \
                    fail(ME_ERR_BAD_ARG, "gemm: the halo tile is a 3x3 convolution");     \
                break;                                                                    \
            case 10: /* 352-row two-group tile, two weight slots: one exact round where 256-row tiles leave a tail */ \
                if constexpr (AMODE == A_PLAIN && EPI == EPI_RESID_SCALE) {                \
                    if (p.ln_out16)                                                       \
                        gemm_launch_pp<T, 352, 256, 2, 4, AMODE, EPI, 2, true>(p, stream); \
                    else                                                                  \
                        gemm_launch_pp<T, 352, 256, 2, 4, AMODE, EPI, 2>(p, stream);      \
                } else if constexpr (AMODE == A_PLAIN && EPI == EPI_STORE)                \
                    gemm_launch_pp<T, 352, 256, 2, 4, AMODE, EPI, 2>(p, stream);          \
                else                                                                      \
                    fail(ME_ERR_BAD_ARG, "gemm: the 352-row tile takes plain linears with store / residual epilogues only"); \
                break;                                                                    \
            case 6:                                                                       \
                if constexpr (AMODE == A_PLAIN && (EPI == EPI_STORE || EPI == EPI_RESID_SCALE))   \
                    gemm_launch_8ph<T, EPI>(p, stream);                                   \
                else                                                                      \
                    fail(ME_ERR_BAD_ARG, "gemm: the 8-phase configuration takes plain linears only"); \
                break;                                                                    \
            default: fail(ME_ERR_BAD_ARG, "gemm: bad tile config %d", cfg);               \
        }                                                                                 \
    }

}  // namespace me
