// Pieces shared by the attention kernels (attention.hip: the round-1 / round-4 kernels on 32x32x16 MFMAs; attention3.hip: the
// forward pass's kernel on 16x16x32): LDS fragment reads, the two-element dot product of the operand type, and the
// vector-pipe path for the one query a (window, head) has beyond its whole wave units.
#pragma once
#include <type_traits>

#include "common.h"
#include "mx_fp8.h"

namespace me {
namespace {

template <typename T>
struct Frag16;
template <>
struct Frag16<f16> {
    typedef f16x8 frag;
};
template <>
struct Frag16<bf16> {
    typedef bf16x8 frag;
};

__device__ __forceinline__ s16x4 lds_read_tr16_at(unsigned a) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)a);
}
template <typename F>
__device__ __forceinline__ F lds_read_frag(unsigned a) {
    return *(const __attribute__((address_space(3))) F*)(size_t)a;
}

// ---------------------------------------------------------------------------------------------------
// The query a (window, head) has beyond its whole 48-query wave units -- 577 = 12 x 48 + 1 -- on the VECTOR pipe, one wave per
// query, shared by the four waves of a trailing workgroup of attention3_kernel.  As one more MFMA block it costs a whole wave's lifetime for one valid
// column, and its workgroup idle waves beside it (10 % of round 4's wave slots).  Here:
// lane = key for the scores (the key's 128-byte row against the query held by every lane, 32 two-element dot products),
// an online softmax per 64-key chunk (wave-wide maximum by DPP), lane = (key group, 8-channel chunk) for P V (eight
// 16-byte rows of V per load instruction, the probability of the lane's key by ds_bpermute), the next chunk's K and V
// rows requested while this one is worked on.  Same arithmetic contract as the matrix path: Q arrives scaled by
// scale * log2(e), p = exp2(s - m) is rounded through the operand type before it is summed and multiplied.
template <typename T>
struct Dot2;
template <>
struct Dot2<f16> {
    typedef f16x2 pair;
    static __device__ __forceinline__ float run(pair a, pair b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
};
template <>
struct Dot2<bf16> {
    typedef __bf16 pair __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float run(pair a, pair b, float c) {
        return __builtin_fmaf((float)a[0], (float)b[0], __builtin_fmaf((float)a[1], (float)b[1], c));
    }
};

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL,
                                                                 0xf, 0xf, false));
}
// the maximum over the wave's 64 lanes, in every lane
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x141>(v));  // row_half_mirror
    v = fmaxf(v, dpp_move<0x140>(v));  // row_mirror: every lane of a 16-lane row has the row's maximum
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// NWAVE waves of the workgroup share the query: wave w takes the 64-key chunks w, w + NWAVE, ...; their partial (maximum, sum,
// numerator) meet through `scratch` (NWAVE x 66 floats of LDS) and wave 0 finishes.  Every thread of the workgroup calls this.
template <typename T, int NWAVE>
__device__ __forceinline__ void attention_extra_query(const T* __restrict__ qrow, const char* kwin, const char* vwin,
                                                      unsigned row_bytes, int tokens, int lane, int wave, float* scratch,
                                                      T* __restrict__ out_row, uint8_t* __restrict__ out8_row,
                                                      uint8_t* __restrict__ out8_scale, int64_t m, int head, int64_t out8_mt) {
    typedef typename Frag16<T>::frag frag;  // 8 elements = 16 bytes
    typedef typename Dot2<T>::pair pair;
    frag qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = *reinterpret_cast<const frag*>(qrow + 8 * j);  // the same 128 bytes in every lane
    const int kg = lane >> 3, dc = lane & 7;
    const int nchunk = (tokens + 63) >> 6;
    frag kr[2][8], vr[2][8];
    auto request = [&](int c, int buf) {
        const int key = 64 * c + lane;
        const char* kp = kwin + (size_t)(key < tokens ? key : tokens - 1) * row_bytes;
#pragma unroll
        for (int j = 0; j < 8; ++j) kr[buf][j] = *reinterpret_cast<const frag*>(kp + 16 * j);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kv = 64 * c + 8 * i + kg;
            vr[buf][i] = *reinterpret_cast<const frag*>(vwin + (size_t)(kv < tokens ? kv : tokens - 1) * row_bytes + 16 * dc);
        }
    };
    float m_run = -INFINITY, l_lane = 0.f;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    if (wave < nchunk) request(wave, 0);
    auto chunk = [&](int c, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        if (c + NWAVE < nchunk) request(c + NWAVE, BUF ^ 1);
        float sc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sc = Dot2<T>::run(pair{kr[BUF][j][2 * e], kr[BUF][j][2 * e + 1]}, pair{qv[j][2 * e], qv[j][2 * e + 1]}, sc);
        if (64 * c + lane >= tokens) sc = -INFINITY;
        const float m_new = fmaxf(m_run, wave_max(sc));  // finite: key 64 c of the chunk exists
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 0 on the wave's first chunk (o and l are 0 then)
        m_run = m_new;
        const float p16 = (float)(T)__builtin_amdgcn_exp2f(sc - m_new);  // through the operand type, like the matrix path's P
        l_lane = __builtin_fmaf(l_lane, alpha, p16);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float pi = __shfl(p16, 8 * i + kg);  // the probability of key 64 c + 8 i + kg
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(pi, (float)vr[BUF][i][j], o[j]);
        }
    };
    for (int c = wave; c < nchunk; c += 2 * NWAVE) {
        chunk(c, std::integral_constant<int, 0>());
        if (c + NWAVE < nchunk) chunk(c + NWAVE, std::integral_constant<int, 1>());
    }
    // the sum over the 64 lanes' keys, and o over the eight key groups (lanes of one channel chunk)
    float l_tot = l_lane;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) l_tot += __shfl_xor(l_tot, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        o[j] += __shfl_xor(o[j], 8);
        o[j] += __shfl_xor(o[j], 16);
        o[j] += __shfl_xor(o[j], 32);
    }
    if constexpr (NWAVE > 1) {
        // the waves' partials: [w][0] maximum, [w][1] sum, [w][2 + d] numerator (a wave without a chunk: -inf, 0, 0)
        float* mine = scratch + wave * 66;
        if (lane == 0) mine[0] = m_run, mine[1] = l_tot;
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) mine[2 + 8 * dc + j] = o[j];
        }
        __syncthreads();
        if (wave != 0) return;
        float m_all = scratch[0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) m_all = fmaxf(m_all, scratch[w * 66]);
        l_tot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) {  // fixed order: the result does not depend on which wave finished first
            const float f = __builtin_amdgcn_exp2f(scratch[w * 66] - m_all);  // wave 0 always has chunk 0: m_all is finite
            l_tot = __builtin_fmaf(scratch[w * 66 + 1], f, l_tot);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(scratch[w * 66 + 2 + 8 * dc + j], f, o[j]);
        }
    }
    const float inv = 1.0f / l_tot;
    auto round16 = [](float x) -> T {
        asm volatile("" : "+v"(x));
        return (T)x;
    };
    if (out8_row) {
        // the bytes the matrix path's store stage writes: a 32-channel MX block = the 8 channels of four lanes
        float v[8], amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (float)round16(o[j] * inv);
            amax = fmaxf(amax, fabsf(v[j]));
        }
        amax = fmaxf(amax, dpp_move<0xB1>(amax));
        amax = fmaxf(amax, dpp_move<0x4E>(amax));
        const unsigned sb = mx_scale_byte(amax);
        const float scl = mx_inv_scale(sb);
        if (kg == 0) {
            uint2 w;
            w.x = pack_fp8x4(v[0] * scl, v[1] * scl, v[2] * scl, v[3] * scl);
            w.y = pack_fp8x4(v[4] * scl, v[5] * scl, v[6] * scl, v[7] * scl);
            *reinterpret_cast<uint2*>(out8_row + 8 * dc) = w;
            if ((dc & 3) == 0) out8_scale[a_scale_index(m, head * 2 + (dc >> 2), out8_mt)] = (uint8_t)sb;
        }
    } else if (kg == 0) {
        frag r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = round16(o[j] * inv);
        *reinterpret_cast<frag*>(out_row + 8 * dc) = r;
    }
}

}  // namespace
}  // namespace me
