// Fused multi-head attention for the DINOv2 blocks (reference src/depth_pro/vit.rs:58-75):
//   attn = softmax((q * scale) k^T) v   per (window, head), head_dim = 64, no mask,
// flash-style: the [tokens x tokens] score matrix (746 MB in the reference at 35 windows,
// SURVEY §3.2) never leaves registers.
//
// One workgroup = 4 waves = 128 query rows of one (window, head); each wave owns 32 queries.
// K/V tiles of 64 keys go global -> registers -> LDS (issue early, write late, double
// buffered).  Per tile and wave:
//   S^T[key][q] = K[key][:] . Q[q][:]          8 x v_mfma_f32_32x32x16 (K tile = A operand)
//   online softmax with the key index in registers and the query on the lane, so row max/sum
//   are in-lane plus one cross-half exchange
//   O^T[d][q] += V^T[d][key] . P^T[key][q]      8 x v_mfma_f32_32x32x16; P^T is the S^T
//   accumulator itself (cdna_hip_programming.md §3 "accumulator tile as the next MFMA's
//   operand"), V^T fragments come from ds_read_b64_tr_b16 on the row-major V tile.
// tokens = 577 is not a multiple of 64: key rows past the end are clamped on load and masked
// to -inf, query rows past the end are clamped on load and not stored.
#include <type_traits>

#include "common.h"

namespace me {

namespace {

template <typename T>
struct Mfma32;
template <>
struct Mfma32<f16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mfma32<bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ s16x4 lds_read_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

constexpr int KT = 64;            // keys per LDS tile
constexpr int TILE_BYTES = KT * 128;

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv,
                                                        T* __restrict__ out, int tokens, int heads,
                                                        float scale_log2e) {
    typedef typename Mfma32<T>::frag frag;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // K0 K1 V0 V1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int C = heads * 64;
    const int ldq = 3 * C;
    const int head = blockIdx.y, win = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int64_t row0 = (int64_t)win * tokens;
    const T* qbase = qkv + head * 64;
    const T* kbase = qkv + C + head * 64;
    const T* vbase = qkv + 2 * C + head * 64;

    // Q fragments: B operand, lane holds Q[q0 + r][16 s + 8 h + 0..7]
    frag qf[4];
    {
        int q = q0 + r;
        q = q < tokens ? q : tokens - 1;
        const T* qp = qbase + (row0 + q) * ldq + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 16 * s);
    }

    // K/V staging: thread owns chunks id = tid and tid + 256 of the 64 x 8 chunk tile
    const int ld_row0 = tid >> 3, ld_c = tid & 7;  // rows ld_row0 and ld_row0 + 32
    // (named registers, not arrays behind a lambda: those end up in scratch)
    uint4 kreg0, kreg1, vreg0, vreg1;
    const int64_t ld_coff = ld_c * 8;
#define ME_ATT_LOAD_TILE(kt_)                                                  \
    do {                                                                       \
        int key0_ = (kt_) * KT + ld_row0, key1_ = key0_ + 32;                  \
        key0_ = key0_ < tokens ? key0_ : tokens - 1;                           \
        key1_ = key1_ < tokens ? key1_ : tokens - 1;                           \
        const int64_t off0_ = (row0 + key0_) * ldq + ld_coff;                  \
        const int64_t off1_ = (row0 + key1_) * ldq + ld_coff;                  \
        kreg0 = *reinterpret_cast<const uint4*>(kbase + off0_);                \
        kreg1 = *reinterpret_cast<const uint4*>(kbase + off1_);                \
        vreg0 = *reinterpret_cast<const uint4*>(vbase + off0_);                \
        vreg1 = *reinterpret_cast<const uint4*>(vbase + off1_);                \
    } while (0)
    const int wr_k0 = ld_row0 * 128 + ((ld_c ^ ((ld_row0 >> 1) & 7)) << 4);
    const int wr_k1 = wr_k0 + 32 * 128;  // (row + 32) has the same swizzle bits
    const int wr_v0 = ld_row0 * 128 + ((ld_c ^ (((ld_row0 >> 1) & 1) << 2)) << 4);
    const int wr_v1 = wr_v0 + 32 * 128;
#define ME_ATT_WRITE_TILE(buf_)                                                \
    do {                                                                       \
        char* kb_ = smem + (buf_) * TILE_BYTES;                                \
        char* vb_ = smem + (2 + (buf_)) * TILE_BYTES;                          \
        *reinterpret_cast<uint4*>(kb_ + wr_k0) = kreg0;                        \
        *reinterpret_cast<uint4*>(kb_ + wr_k1) = kreg1;                        \
        *reinterpret_cast<uint4*>(vb_ + wr_v0) = vreg0;                        \
        *reinterpret_cast<uint4*>(vb_ + wr_v1) = vreg1;                        \
    } while (0)

    // K fragment read: row = ks*32 + r, chunk 2s + h, slot = chunk ^ ((r >> 1) & 7)
    const int k_rd = r * 128;
    const int k_swz = (r >> 1) & 7;
    // V^T fragment via transposed read: group g = lane >> 4, il = lane & 15
    //   row = key0 + (il >> 2), 8-byte piece p = il & 3 of the 16 d-columns d0 = 32 dblk + 16 (g & 1)
    const int il = lane & 15;
    const int v_qrow = il >> 2, v_p = il & 3;
    const int v_dhalf = (lane >> 4) & 1;

    f32x16 o[2];
    o[0] = f32x16{0};
    o[1] = f32x16{0};
    float m_run = -1e30f, l_run = 0.f;

    const int nkt = (tokens + KT - 1) / KT;
    ME_ATT_LOAD_TILE(0);
    ME_ATT_WRITE_TILE(0);
    __syncthreads();

    // One KV tile.  TAIL (last tile only) masks the keys past the end.  Softmax runs on raw scores:
    // p = exp2(s*c - m*c) with c = scale*log2(e) folded into one FMA; O and l are rescaled only in the
    // tiles where some lane's running max actually grows (exact, and rare after the first tiles).
    auto tile = [&](int kt, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int buf = kt & 1;
        if (!TAIL) ME_ATT_LOAD_TILE(kt + 1);
        const char* kb = smem + buf * TILE_BYTES;
        const char* vb = smem + (2 + buf) * TILE_BYTES;

        // ---- S^T = K Q^T for the two 32-key halves of the tile
        f32x16 s[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s[ks] = f32x16{0};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const frag kf = *reinterpret_cast<const frag*>(
                    kb + ks * 32 * 128 + k_rd + (((2 * st + h) ^ k_swz) << 4));
                s[ks] = Mfma32<T>::run(kf, qf[st], s[ks]);
            }
        }
        // ---- running max on the raw scores
        float mloc = -1e30f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (TAIL) {
                    const int key = kt * KT + ks * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (key >= tokens) s[ks][g] = -INFINITY;
                }
                mloc = fmaxf(mloc, s[ks][g]);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        if (__any(mloc > m_run)) {
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 16; ++g) o[d][g] *= alpha;
        }
        const float mc = m_run * scale_log2e;
        float psum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                // raw v_exp_f32: the argument is <= 0, results below 2^-126 may flush to 0
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ks][g], scale_log2e, -mc));
                s[ks][g] = pv;
                psum += pv;
            }
        l_run += psum;

        // ---- O^T += V^T P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                frag pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (T)s[ks][8 * s2 + j];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    s16x4 half0, half1;
                    {
                        const int row = ks * 32 + 16 * s2 + 4 * h + v_qrow;
                        const int chunk = (d * 4 + 2 * v_dhalf + (v_p >> 1)) ^ (((row >> 1) & 1) << 2);
                        half0 = lds_read_tr16(vb + row * 128 + (chunk << 4) + ((v_p & 1) << 3));
                    }
                    {
                        const int row = ks * 32 + 16 * s2 + 8 + 4 * h + v_qrow;
                        const int chunk = (d * 4 + 2 * v_dhalf + (v_p >> 1)) ^ (((row >> 1) & 1) << 2);
                        half1 = lds_read_tr16(vb + row * 128 + (chunk << 4) + ((v_p & 1) << 3));
                    }
                    typedef short s16x8 __attribute__((__vector_size__(16)));
                    const s16x8 both = __builtin_shufflevector(half0, half1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[d] = Mfma32<T>::run(__builtin_bit_cast(frag, both), pf, o[d]);
                }
            }

        if (!TAIL) ME_ATT_WRITE_TILE(buf ^ 1);
        __syncthreads();
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) tile(kt, std::false_type());
    tile(nkt - 1, std::true_type());

    // ---- normalise and store: lane holds q = q0 + r, d = 32 dblk + 8 (g >> 2) + 4 h + (g & 3)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r;
    if (q < tokens) {
        T* op = out + (row0 + q) * C + head * 64 + 4 * h;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef T v4 __attribute__((ext_vector_type(4)));
                v4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (T)(o[d][4 * g4 + e] * inv);
                *reinterpret_cast<v4*>(op + d * 32 + 8 * g4) = v;
            }
    }
}

#undef ME_ATT_LOAD_TILE
#undef ME_ATT_WRITE_TILE

}  // namespace

void attention_launch(const void* qkv, void* out, int32_t windows, int32_t tokens, int32_t heads,
                      int32_t dtype, hipStream_t stream) {
    ME_CHECK(windows > 0 && tokens > 0 && heads > 0, ME_ERR_BAD_SHAPE,
             "attention: windows=%d tokens=%d heads=%d", windows, tokens, heads);
    ME_CHECK(heads <= 65535 && windows <= 65535, ME_ERR_BAD_SHAPE, "attention: grid too large");
    const dim3 grid((tokens + 127) / 128, heads, windows);
    ProfScope prof(stream, "attention_kernel", 4.0 * windows * heads * (double)tokens * tokens * 64, 0.0);
    // scale = 1/sqrt(64) (vit.rs:47), folded with log2(e) so the softmax runs on exp2
    const float scale_log2e = 0.125f * 1.44269504088896340736f;
    if (dtype == ME_DTYPE_F16)
        hipLaunchKernelGGL(attention_kernel<f16>, grid, dim3(256), 0, stream, (const f16*)qkv,
                           (f16*)out, tokens, heads, scale_log2e);
    else if (dtype == ME_DTYPE_BF16)
        hipLaunchKernelGGL(attention_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)qkv,
                           (bf16*)out, tokens, heads, scale_log2e);
    else
        fail(ME_ERR_BAD_ARG, "attention: bad dtype %d", dtype);
    ME_HIP(hipGetLastError());
}

}  // namespace me
