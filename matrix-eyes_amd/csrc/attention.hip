// Fused multi-head attention for the DINOv2 blocks (reference src/depth_pro/vit.rs:58-75):
//   attn = softmax((q * scale) k^T) v   per (window, head), head_dim = 64, no mask,
// flash-style: the [tokens x tokens] score matrix (746 MB in the reference at 35 windows,
// SURVEY §3.2) never leaves registers.
//
// One workgroup = 4 waves = 128 query rows of one (window, head); each wave owns 32 queries.
// K/V tiles of 64 keys go global -> LDS by LDS-DMA into a ring of NSLOT slots (no staging registers, no
// per-tile address arithmetic).  Measured on one box: two slots with the registers capped for 4 waves per
// SIMD 0.0783 ms at 35 windows, two or three slots at 3 waves per SIMD 0.082 -- the kernel does not wait
// for K/V (a second tile of prefetch distance buys nothing), it is bound by each wave's dependent
// MFMA -> softmax -> MFMA chain, and a fourth wave per SIMD covers more of it.  Per tile and wave:
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
#include "mx_fp8.h"

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
// the same by LDS byte address
__device__ __forceinline__ s16x4 lds_read_tr16_at(unsigned a) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)a);
}
template <typename F>
__device__ __forceinline__ F lds_read_frag(unsigned a) {
    return *(const __attribute__((address_space(3))) F*)(size_t)a;
}

constexpr int KT = 64;            // keys per LDS tile
constexpr int TILE_BYTES = KT * 128;

// NSLOT: LDS ring slots (a K and a V tile each; 2 or 3); MINW: waves per SIMD the register allocation
// must allow (launched as <2, 4>: 32 KiB of LDS and 128 VGPRs, three of them spilled)
#ifdef ME_ATT_STAMPS
// Diagnostic build only (tools/attn_stamps.py): per-wave phase accounting in shader clocks.  Reading the clock
// waits for all LDS operations in flight (lgkmcnt), so the phases are serialised a little more than in the
// product kernel.  [0] DMA wait + barrier, [1] staging issue, [2] S = K Q^T and the running max, [3] rescale +
// exponentials, [4] P V, [5] closing LDS wait, [6] everything (prologue and epilogue included).
__device__ unsigned long long* g_att_stamps = nullptr;
// timing-only ablations of attention2_kernel (results are wrong): 1 = no restaging and no barrier after the first
// tile, 2 = no exponentials, 4 = no P V / row-sum MFMAs, 8 = no S MFMAs
__device__ int g_att_mode = 0;
#define ATT_PH(i)                                                  \
    do {                                                           \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[i] += t_now - t_last;                                   \
        t_last = t_now;                                            \
    } while (0)
#else
#define ATT_PH(i)
#endif

template <typename T, int NSLOT, int MINW>
__global__ __launch_bounds__(256, MINW) void attention_kernel(const T* __restrict__ qkv,
                                                        T* __restrict__ out, int tokens, int heads,
                                                        int ngroups, float scale_log2e, RowSegs segs,
                                                        uint8_t* __restrict__ out8, uint8_t* __restrict__ out8_scale,
                                                        int64_t out8_mt) {
    typedef typename Mfma32<T>::frag frag;
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * 2 * TILE_BYTES];  // slot: K tile, V tile
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef ME_ATT_STAMPS
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_last = t_begin;
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int C = heads * 64;
    const int ldq = 3 * C;
    // Block -> (window, head, query block).  Blocks b and b + 8 share an XCD and its L2: all query blocks of
    // one (window, head) are given to ONE XCD, so its K and V (148 KB at 577 tokens, read by every query
    // block) cross the fabric once (3-5 % over the plain x-fastest grid at 35 to 140 windows).
    const int nqb = (tokens + 127) >> 7;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int group = xcd + 8 * (slot0 / nqb);  // = win * heads + head
    const int qblk = slot0 - (slot0 / nqb) * nqb;
    if (group >= ngroups) return;               // uniform: the whole workgroup leaves before any barrier
    const int win = group / heads, head = group - win * heads;
    const int q0 = qblk * 128 + wave * 32;
    // first row of the window: windows are `tokens` rows apart inside a row segment (RowSegs)
    int64_t row0 = (int64_t)win * tokens;
    if (segs.seg1 && win >= segs.win0)
        row0 = win < segs.win0 + segs.win1 ? segs.seg1 + (int64_t)(win - segs.win0) * tokens
                                           : segs.seg2 + (int64_t)(win - segs.win0 - segs.win1) * tokens;
    const T* qbase = qkv + head * 64;
    const T* kbase = qkv + C + head * 64;
    const T* vbase = qkv + 2 * C + head * 64;

    // K/V staging by LDS-DMA: a wave-instruction moves 8 rows x 128 B (lane -> row lane / 8, 16-byte chunk
    // lane % 8); the LDS image is linear in the lane, so the swizzle goes on the SOURCE chunk: the LDS slot
    // (lane % 8) of row rr holds chunk (lane % 8) ^ swz(rr).  Wave w stages pieces 2w and 2w + 1 of K and V.
    const int st_row = lane >> 3, st_slot = lane & 7;
    unsigned koff[2], voff[2];
    auto stage_offsets = [&](int kt) {  // per tile only because the last tile clamps its rows
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = (2 * wave + i) * 8 + st_row;
            int key = kt * KT + rr;
            key = key < tokens ? key : tokens - 1;
            const unsigned rowb = (unsigned)key * (unsigned)ldq * 2u;  // bytes from the window's first row
            koff[i] = rowb + ((st_slot ^ ((rr >> 1) & 7)) << 4);
            voff[i] = rowb + ((st_slot ^ (((rr >> 1) & 1) << 2)) << 4);
        }
    };
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    const char* kwin = uniform_ptr((const char*)(kbase + row0 * ldq));
    const char* vwin = uniform_ptr((const char*)(vbase + row0 * ldq));
    auto stage = [&](int slot) {  // 4 LDS-DMA instructions per wave and tile
        const unsigned dst = smem_base + slot * (2 * TILE_BYTES) + (2 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16_raw(kwin, koff[i], dst + i * 1024);
            glds16_raw(vwin, voff[i], dst + TILE_BYTES + i * 1024);
        }
    };

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
    stage_offsets(0);
    stage(0);
    if (NSLOT == 3 && nkt > 1) {
        stage_offsets(1);
        stage(1);
    }

    // (after the first tile's DMA has been issued: the two memory latencies of a workgroup's start overlap)
    // Q fragments: B operand, lane holds Q[q0 + r][16 s + 8 h + 0..7]
    frag qf[4];
    {
        int q = q0 + r;
        q = q < tokens ? q : tokens - 1;
        const T* qp = qbase + (row0 + q) * ldq + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 16 * s);
        // settle these loads here: left pending into the loop, the compiler's wait for them sits at the
        // first MFMA of EVERY tile and, returns being in order, waits for the ring's DMA as well
        typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            i32x4 t = __builtin_bit_cast(i32x4, qf[s]);
            asm volatile("" : "+v"(t));
            qf[s] = __builtin_bit_cast(frag, t);
        }
    }


    // One KV tile.  TAIL (last tile only) masks the keys past the end.  Softmax runs on raw scores:
    // p = exp2(s*c - m*c) with c = scale*log2(e) folded into one FMA; O and l are rescaled only in the
    // tiles where some lane's running max actually grows (exact, and rare after the first tiles).
    // TAIL (last tile) masks the keys past the end.  A wave whose 32 queries all lie past the end (the
    // last query block of 577 tokens has two such waves' worth) only helps with the staging.
    // Ring protocol per tile kt: wait for this wave's pieces of tile kt (those of tile kt+1 may stay in
    // flight), barrier (everybody's pieces have landed; everybody is done with tile kt-1), restage the slot
    // of tile kt-1 with tile kt+2, compute.
    const bool active = q0 < tokens;
    int slot = 0, fill = NSLOT - 1;
    auto tile = [&](int kt, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#ifdef ME_ATT_STAMPS
        t_last = __builtin_amdgcn_s_memtime();
#endif
        if (NSLOT == 3 && kt + 1 < nkt)
            wait_vmcnt<4>();  // the next tile's pieces may stay in flight
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ATT_PH(0);
        if (kt + NSLOT - 1 < nkt) {
            stage_offsets(kt + NSLOT - 1);
            stage(fill);
        }
        ATT_PH(1);
        const char* kb = smem + slot * (2 * TILE_BYTES);
        const char* vb = kb + TILE_BYTES;
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
        fill = fill == NSLOT - 1 ? 0 : fill + 1;

        if (active) {
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
        ATT_PH(2);
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
        ATT_PH(3);

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
        ATT_PH(4);
        }  // active
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's LDS reads are done before the
                                                             // barrier that lets its slot be restaged
        ATT_PH(5);
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) tile(kt, std::false_type());
    if ((tokens % KT) == 1 && nkt >= 2) {
        // 577 = 9 x 64 + 1: a whole tile of MFMAs and exponentials for ONE key would be a tenth of the
        // kernel; that key (row 0 of the tile staged last) is folded in as a rank-one update instead
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (active) {
            const char* kb = smem + slot * (2 * TILE_BYTES);
            const char* vb = kb + TILE_BYTES;
            // s = q . k over this lane's 32 of the 64 dimensions (d = 16 st + 8 h + j), then the other half
            float dot = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const frag kf = *reinterpret_cast<const frag*>(kb + ((2 * st + h) << 4));  // row 0: no swizzle
#pragma unroll
                for (int j = 0; j < 8; ++j) dot = __builtin_fmaf((float)kf[j], (float)qf[st][j], dot);
            }
            dot += __shfl_xor(dot, 32);
            const float m_new = fmaxf(m_run, dot);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            const float pv = __builtin_amdgcn_exp2f((dot - m_new) * scale_log2e);
            m_run = m_new;
            l_run = l_run * alpha + (h == 0 ? pv : 0.f);  // l is summed over the two lane halves at the end
            const float p16 = (float)(T)pv;               // P goes through the operand type like every other key's
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    typedef T v4 __attribute__((ext_vector_type(4)));
                    const v4 vv = *reinterpret_cast<const v4*>(vb + ((4 * d + g4) << 4) + 8 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[d][4 * g4 + e] = __builtin_fmaf(p16, (float)vv[e], o[d][4 * g4 + e] * alpha);
                }
        }
    } else {
        tile(nkt - 1, std::true_type());
    }

    // ---- normalise and store: lane holds q = q0 + r, d = 32 dblk + 8 (g >> 2) + 4 h + (g & 3)
    // The product is rounded to f32 and THEN to 16 bit in both output forms: left to itself the compiler fuses
    // multiply and conversion into one v_fma_mixlo (a single rounding) in one form and not in the other, and the
    // fp8 form's bytes must be those of the 16-bit form quantised.
    auto round16 = [](float x) -> T {
        asm volatile("" : "+v"(x));
        return (T)x;
    };
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r;
    if (out8) {
        // ME_DTYPE_FP8: the output leaves as the projection's MX fp8 activation operand (mx_fp8.h).  A 32-column
        // block of a row is this lane's 16 values of one d-block and lane ^ 32's; values are rounded to 16 bit
        // first, so the bytes equal those of quantize_f16_to_fp8_launch over the 16-bit output.
        const int64_t m = row0 + q;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float v[16];
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e] = (float)round16(o[d][e] * inv);
                amax = fmaxf(amax, fabsf(v[e]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const unsigned sb = mx_scale_byte(amax);
            const float sc = mx_inv_scale(sb);
            if (q < tokens) {
                uint8_t* op = out8 + m * C + head * 64 + d * 32 + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<unsigned*>(op + 8 * g4) =
                        pack_fp8x4(v[4 * g4] * sc, v[4 * g4 + 1] * sc, v[4 * g4 + 2] * sc, v[4 * g4 + 3] * sc);
                if (h == 0) out8_scale[a_scale_index(m, head * 2 + d, out8_mt)] = (uint8_t)sb;
            }
        }
    } else if (q < tokens) {
        T* op = out + (row0 + q) * C + head * 64 + 4 * h;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef T v4 __attribute__((ext_vector_type(4)));
                v4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round16(o[d][4 * g4 + e] * inv);
                *reinterpret_cast<v4*>(op + d * 32 + 8 * g4) = v;
            }
    }
#ifdef ME_ATT_STAMPS
    if (g_att_stamps && lane == 0 && blockIdx.x < 4096) {
        ph[6] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int i = 0; i < 7; ++i) g_att_stamps[((size_t)blockIdx.x * 4 + wave) * 8 + i] = ph[i];
        g_att_stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 7] = active ? 1 : 0;
    }
#endif
}



// ---------------------------------------------------------------------------------------------------
// The kernel of the forward pass (Q pre-scaled by the qkv linear's epilogue, GemmParams::qcols): the same decomposition
// as the kernel above (one workgroup = 4 waves = 128 queries of one (window, head), 64-key K/V tiles through a two-slot
// LDS-DMA ring, S^T = K Q^T with the S^T accumulator as the B operand of O^T += V^T P^T) with the VECTOR work per score
// cut from 4 instructions to 2.2 (PMC: 14.7 -> 7.5 VALU instructions per MFMA):
//   * the softmax reference point lives in the matrix pipe.  Q arrives multiplied by scale * log2(e), and the first
//     MFMA of a score block takes C = -m', sixteen registers that all hold the negated reference point of the lane's
//     query in exp2 units: S' = K Qc^T - m' needs no FMA per score, p = exp2(S') directly;
//   * the reference point follows the running maximum only when some score passes it by more than 2^defer_thr (T13 of
//     the guide; the first tile always sets it): with 32 queries to a wave SOME maximum grows in nearly every tile, and
//     the rescale (60 vector instructions: O, l, the score block and C) ran every tile -- as it does in the kernel above;
//   * the row sums come from the matrix pipe too: one more MFMA per 16 keys with an all-ones A operand adds up the
//     P^T operand over its keys (all 32 rows of that product are the same sum; both lane halves included, so there is
//     no cross-half exchange at the end, and l is the sum of the ROUNDED probabilities that the numerator uses);
//   * staging addresses are per-lane constants beside a uniform base that walks with the tile (the ragged last tile
//     computes its clamped rows when it is staged, once).
// Per tile and wave: 32 v_exp_f32, 16 v_max3_f32, 16 v_cvt_pk, 20 MFMAs (8 + 8 + 4).  168 VGPRs: three waves per SIMD.
// Measured (tools/attn_ab.py, 37 windows, one process, interleaved): 82 us against 88 - 91 us for the kernel above
// (614 against 554 - 570 TFLOP/s on the real FLOPs) -- half the vector instructions buy 8 %: at three or four waves per SIMD a
// wave's tile is a serial chain (K reads -> 8 MFMAs -> max -> exp -> cvt -> 12 MFMAs) and its lifetime, 4800 cycles per tile
// beside two or three others, barely moves with the instruction count (phase stamps and timing-only ablations:
// profiles/r04_attention_ablations.txt).  What that points at is two query blocks per wave (the second block's MFMAs
// under the first one's softmax), not fewer instructions.
// HALVES: the two 32-key halves of a tile one after the other through ONE score block, row sums on the vector pipe:
// 32 registers less (128 VGPRs, four waves per SIMD) than with the halves side by side and the sums on the matrix pipe
template <typename T, int MINW, bool HALVES>
__global__ __launch_bounds__(256, MINW) void attention2_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                               int tokens, int heads, int ngroups,
                                                               RowSegs segs, uint8_t* __restrict__ out8,
                                                               uint8_t* __restrict__ out8_scale, int64_t out8_mt,
                                                               float defer_thr) {
    typedef typename Mfma32<T>::frag frag;
    constexpr int NSLOT = 2;
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * 2 * TILE_BYTES];  // slot: K tile, V tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int bid = blockIdx.x;
#ifdef ME_ATT_STAMPS
    // [0] DMA wait + barrier, [1] staging issue, [2] S' = K Qc^T, [3] max + branch, [4] exponentials, [5] P V + row sums,
    // [6] everything; [7] prologue (up to the first tile)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_last = t_begin;
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int C = heads * 64;
    const int ldq = 3 * C;
    const int nqb = (tokens + 127) >> 7;
    const int xcd = bid & 7, slot0 = bid >> 3;
    const int group = xcd + 8 * (slot0 / nqb);  // = win * heads + head
    const int qblk = slot0 - (slot0 / nqb) * nqb;
    if (group >= ngroups) return;  // uniform: the whole workgroup leaves before any barrier
    const int win = group / heads, head = group - win * heads;
    const int q0 = qblk * 128 + wave * 32;
    int64_t row0 = (int64_t)win * tokens;
    if (segs.seg1 && win >= segs.win0)
        row0 = win < segs.win0 + segs.win1 ? segs.seg1 + (int64_t)(win - segs.win0) * tokens
                                           : segs.seg2 + (int64_t)(win - segs.win0 - segs.win1) * tokens;
    const T* qbase = qkv + head * 64;
    const T* kbase = qkv + C + head * 64;
    const T* vbase = qkv + 2 * C + head * 64;

    // K/V staging by LDS-DMA (see the kernel above): wave w stages pieces 2w and 2w + 1 of K and of V
    const int st_row = lane >> 3, st_slot = lane & 7;
    const unsigned row_bytes = (unsigned)ldq * 2u;
    // offsets of this wave's FIRST piece (rows 16 wave + st_row); the second piece, eight rows on, is derived from them
    // where it is issued: K's swizzle (rr >> 1) & 7 changes by 4 (chunk ^ 4 = byte offset ^ 64), V's (rr >> 1) & 1 does not
    unsigned koff0, voff0;
    {
        const int rr = (2 * wave) * 8 + st_row;
        koff0 = (unsigned)rr * row_bytes + ((st_slot ^ ((rr >> 1) & 7)) << 4);
        voff0 = (unsigned)rr * row_bytes + ((st_slot ^ (((rr >> 1) & 1) << 2)) << 4);
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    const char* kwin = uniform_ptr((const char*)(kbase + row0 * ldq));
    const char* vwin = uniform_ptr((const char*)(vbase + row0 * ldq));
    const int nkt = (tokens + KT - 1) / KT;
    auto stage = [&](int kt, int slot) {  // 4 LDS-DMA instructions per wave and tile
        const unsigned dst = smem_base + slot * (2 * TILE_BYTES) + (2 * wave) * 1024;
        const char* kt_k = uniform_ptr(kwin + (size_t)kt * KT * row_bytes);
        const char* kt_v = uniform_ptr(vwin + (size_t)kt * KT * row_bytes);
        if ((kt + 1) * KT <= tokens) {
            unsigned k0 = koff0, v0 = voff0;
            asm volatile("" : "+v"(k0), "+v"(v0));  // (the derived offsets stay temporaries: hoisted, they spill)
            glds16_raw(kt_k, k0, dst);
            glds16_raw(kt_v, v0, dst + TILE_BYTES);
            glds16_raw(kt_k, (k0 ^ 64u) + 8u * row_bytes, dst + 1024);
            glds16_raw(kt_v, v0 + 8u * row_bytes, dst + TILE_BYTES + 1024);
        } else {  // the ragged last tile: rows past the end read the last row (masked or unused below)
            const int last = tokens - 1 - kt * KT;  // >= 0
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rr = (2 * wave + i) * 8 + st_row;
                const unsigned rowb = (unsigned)(rr < last ? rr : last) * row_bytes;
                glds16_raw(kt_k, rowb + ((st_slot ^ ((rr >> 1) & 7)) << 4), dst + i * 1024);
                glds16_raw(kt_v, rowb + ((st_slot ^ (((rr >> 1) & 1) << 2)) << 4), dst + TILE_BYTES + i * 1024);
            }
        }
    };

    // Fragment read addresses as TWO per-lane constants beside immediates and XORs with immediates (written out, the
    // swizzles give the compiler a register per (half, sub-block, d, st) combination, the 128-register variant spills
    // them into the tile loop, and every reload's vmcnt wait there also waits for the next tile's LDS-DMA):
    //   K: row = 32 ks + r, chunk 2 st + h at slot (2 st + h) ^ ((r >> 1) & 7) = 2 (st ^ t) + (h ^ u): byte offset
    //      k_lane ^ (st << 5) + ks * 4096, k_lane being the offset of st = 0
    //   V (transposed read, group g = lane >> 4, il = lane & 15): row = 32 ks + 16 s2 + 8 half + 4 h + (il >> 2), chunk
    //      (4 d + 2 (g & 1) + (il >> 1 & 1)) ^ (row >> 1 & 1) << 2 -- the swizzle bit is (il >> 3) & 1 for every row of the
    //      lane, so it only exchanges the two d blocks: v_lane ^ (d << 6) + (32 ks + 16 s2 + 8 half) * 128
    const int il = lane & 15;
    int k_lane, v_lane;
    {
        const int k_swz = (r >> 1) & 7;
        k_lane = r * 128 + ((h ^ k_swz) << 4);
        const int v_qrow = il >> 2, v_p = il & 3, v_dhalf = (lane >> 4) & 1;
        v_lane = (4 * h + v_qrow) * 128 + ((2 * v_dhalf + (v_p >> 1)) << 4) + ((v_p & 1) << 3) + (((v_qrow >> 1) & 1) << 6);
    }
    f32x16 o[2], lsum, negm;
    o[0] = f32x16{0};
    o[1] = f32x16{0};
    lsum = f32x16{0};
    negm = f32x16{0};
    float m_run = 0.f;  // the running maximum (exp2 units); negm == -m_run in all sixteen registers
    float l_run = 0.f;  // HALVES: this lane's part of the row sum
    frag ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;

    stage(0, 0);

    // Q fragments: B operand, lane holds Qc[q0 + r][16 s + 8 h + 0..7], Qc = Q * scale * log2(e)
    frag qf[4];
    {
        int q = q0 + r;
        q = q < tokens ? q : tokens - 1;
        const T* qp = qbase + (row0 + q) * ldq + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 16 * s);
        // settle these loads here (see the kernel above)
        typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            i32x4 t = __builtin_bit_cast(i32x4, qf[s]);
            asm volatile("" : "+v"(t));
            qf[s] = __builtin_bit_cast(frag, t);
        }
    }

    const bool active = q0 < tokens;
    int slot = 0;
    auto tile = [&](int kt, auto first_tag, auto tail_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool TAIL = decltype(tail_tag)::value;
#ifdef ME_ATT_STAMPS
        if (FIRST) ph[7] = __builtin_amdgcn_s_memtime() - t_begin;
        t_last = __builtin_amdgcn_s_memtime();
#endif
#ifdef ME_ATT_STAMPS
        const int abl = __builtin_amdgcn_readfirstlane(g_att_mode);
        if (!(abl & 1) || FIRST) {
#endif
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ATT_PH(0);
        if (kt + 1 < nkt) stage(kt + 1, slot ^ 1);
#ifdef ME_ATT_STAMPS
        }
#endif
        ATT_PH(1);
        // this tile's fragment addresses (LDS byte addresses; the slots are 16 KiB-aligned, so the XORs below stay inside)
        unsigned ka = smem_base + slot * (2 * TILE_BYTES) + (unsigned)k_lane;
        unsigned va = smem_base + slot * (2 * TILE_BYTES) + TILE_BYTES + (unsigned)v_lane;
        asm volatile("" : "+v"(ka), "+v"(va));  // (derived per tile: hoisted, the variants spill)
        slot ^= 1;

        auto part = [&](auto k0_tag, auto nks_tag, bool first_part) {
            constexpr int K0 = decltype(k0_tag)::value, NKS = decltype(nks_tag)::value;
            // ---- S' = K Qc^T - m' for 32-key halves K0 .. K0 + NKS - 1 of the tile
            f32x16 s[NKS];
#ifdef ME_ATT_STAMPS
            if (abl & 8) {
                for (int ks = 0; ks < NKS; ++ks) s[ks] = negm;
                asm volatile("" : "+v"(s[0]), "+v"(s[NKS - 1]));
            } else
#endif
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const frag kf = lds_read_frag<frag>((ka ^ (unsigned)(st << 5)) + (K0 + ks) * 32 * 128);
                    s[ks] = Mfma32<T>::run(kf, qf[st], st == 0 ? negm : s[ks]);
                }
            }
#ifdef ME_ATT_STAMPS
            asm volatile("" : "+v"(s[0]), "+v"(s[NKS - 1]));
#endif
            ATT_PH(2);
            // ---- does the running maximum grow?
            float mloc = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    if (TAIL) {
                        const int key = kt * KT + (K0 + ks) * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                        if (key >= tokens) s[ks][g] = -INFINITY;
                    }
                    mloc = fmaxf(mloc, s[ks][g]);
                }
            {
                // the other lane half holds the other 32 keys of the tile: after the swap `a` is the low half's
                // maximum in both halves and `b` the high half's.  (As asm: given one value for both operands of
                // __builtin_amdgcn_permlane32_swap the compiler folds the two results into one and the high half's
                // keys never reach the comparison.)
                float a = mloc, b = mloc;
                // (two wait states between the VALU write of an operand and the swap: the hazard pass cannot see into asm)
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
                mloc = fmaxf(a, b);
            }
            // The reference point follows the maximum only when some lane's scores pass it by more than defer_thr
            // (exp2 units; cdna_hip_programming.md T13): until then p = exp2(S') may reach 2^defer_thr instead of 1 --
            // the same relative precision in the 16-bit P operand, and O and l carry the same factor, so the quotient is
            // unchanged.  With 32 queries to a wave SOME maximum grows in nearly every tile (the branch, 60 vector
            // instructions, would run every time); past the threshold almost never after the first tile.
            if ((FIRST && first_part) || __any(mloc > defer_thr)) {
                const float delta = (FIRST && first_part) ? mloc : fmaxf(mloc, 0.f);
                if (!(FIRST && first_part)) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int g = 0; g < 16; ++g) o[0][g] *= alpha, o[1][g] *= alpha;
                    if constexpr (HALVES) {
                        l_run *= alpha;
                    } else {
#pragma unroll
                        for (int g = 0; g < 16; ++g) lsum[g] *= alpha;
                    }
                }
                m_run += delta;
#pragma unroll
                for (int g = 0; g < 16; ++g) negm[g] = -m_run;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int g = 0; g < 16; ++g) s[ks][g] -= delta;
            }
            ATT_PH(3);
            // ---- p = exp2(S')  (raw v_exp_f32: the argument is <= 0, results below 2^-126 may flush to 0)
#ifdef ME_ATT_STAMPS
            if (!(abl & 2))
#endif
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int g = 0; g < 16; ++g) s[ks][g] = __builtin_amdgcn_exp2f(s[ks][g]);

#ifdef ME_ATT_STAMPS
            asm volatile("" : "+v"(s[0]), "+v"(s[NKS - 1]));
#endif
            ATT_PH(4);
            // ---- O^T += V^T P^T,  l += 1^T P^T
#ifdef ME_ATT_STAMPS
            if (abl & 4) {
                asm volatile("" : "+v"(s[0]), "+v"(s[NKS - 1]));
            } else
#endif
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    frag pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (T)s[ks][8 * s2 + j];
                    if constexpr (HALVES) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) l_run += s[ks][8 * s2 + j];
                    } else {
                        lsum = Mfma32<T>::run(ones, pf, lsum);
                    }
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const unsigned vp = (va ^ (unsigned)(d << 6)) + ((K0 + ks) * 32 + 16 * s2) * 128;
                        const s16x4 half0 = lds_read_tr16_at(vp), half1 = lds_read_tr16_at(vp + 8 * 128);
                        typedef short s16x8 __attribute__((__vector_size__(16)));
                        const s16x8 both = __builtin_shufflevector(half0, half1, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[d] = Mfma32<T>::run(__builtin_bit_cast(frag, both), pf, o[d]);
                    }
                }
        };
        if (active) {
            if constexpr (HALVES) {
                part(std::integral_constant<int, 0>(), std::integral_constant<int, 1>(), true);
                part(std::integral_constant<int, 1>(), std::integral_constant<int, 1>(), false);
            } else {
                part(std::integral_constant<int, 0>(), std::integral_constant<int, 2>(), true);
            }
        }  // active
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's LDS reads are done before the
                                                             // barrier that lets its slot be restaged
#ifdef ME_ATT_STAMPS
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(lsum));
#endif
        ATT_PH(5);
    };
    float l_tot;
    const bool tail_key = (tokens % KT) == 1 && nkt >= 2;
    const int nfull = tail_key ? nkt - 1 : nkt;  // tiles that run on the matrix pipe
    if (nfull == 1) {
        tile(0, std::true_type(), std::true_type());
    } else {
        tile(0, std::true_type(), std::false_type());
        for (int kt = 1; kt + 1 < nfull; ++kt) tile(kt, std::false_type(), std::false_type());
        if (tail_key)
            tile(nfull - 1, std::false_type(), std::false_type());  // a whole tile: 64 valid keys
        else
            tile(nfull - 1, std::false_type(), std::true_type());
    }
    if constexpr (HALVES) l_tot = l_run + __shfl_xor(l_run, 32);
    else l_tot = lsum[0];
    if (tail_key) {
        // 577 = 9 x 64 + 1: the single key of the last tile (row 0 of the tile staged last) as a rank-one update
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (active) {
            const char* kb = smem + slot * (2 * TILE_BYTES);
            const char* vb = kb + TILE_BYTES;
            float dot = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const frag kf = *reinterpret_cast<const frag*>(kb + ((2 * st + h) << 4));  // row 0: no swizzle
#pragma unroll
                for (int j = 0; j < 8; ++j) dot = __builtin_fmaf((float)kf[j], (float)qf[st][j], dot);
            }
            dot += __shfl_xor(dot, 32);
            const float rel = dot - m_run;
            const float delta = fmaxf(rel, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            const float p16 = (float)(T)__builtin_amdgcn_exp2f(rel - delta);  // through the operand type like every other key's
            l_tot = l_tot * alpha + p16;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    typedef T v4 __attribute__((ext_vector_type(4)));
                    const v4 vv = *reinterpret_cast<const v4*>(vb + ((4 * d + g4) << 4) + 8 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[d][4 * g4 + e] = __builtin_fmaf(p16, (float)vv[e], o[d][4 * g4 + e] * alpha);
                }
        }
    }

    // ---- normalise and store (see the kernel above)
    auto round16 = [](float x) -> T {
        asm volatile("" : "+v"(x));
        return (T)x;
    };
    const float inv = 1.0f / l_tot;
    const int q = q0 + r;
    if (out8) {
        const int64_t m = row0 + q;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float v[16];
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e] = (float)round16(o[d][e] * inv);
                amax = fmaxf(amax, fabsf(v[e]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const unsigned sb = mx_scale_byte(amax);
            const float sc = mx_inv_scale(sb);
            if (q < tokens) {
                uint8_t* op = out8 + m * C + head * 64 + d * 32 + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<unsigned*>(op + 8 * g4) =
                        pack_fp8x4(v[4 * g4] * sc, v[4 * g4 + 1] * sc, v[4 * g4 + 2] * sc, v[4 * g4 + 3] * sc);
                if (h == 0) out8_scale[a_scale_index(m, head * 2 + d, out8_mt)] = (uint8_t)sb;
            }
        }
    } else if (q < tokens) {
        T* op = out + (row0 + q) * C + head * 64 + 4 * h;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef T v4 __attribute__((ext_vector_type(4)));
                v4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round16(o[d][4 * g4 + e] * inv);
                *reinterpret_cast<v4*>(op + d * 32 + 8 * g4) = v;
            }
    }
#ifdef ME_ATT_STAMPS
    if (g_att_stamps && lane == 0 && bid < 4096) {
        ph[6] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int i = 0; i < 7; ++i) g_att_stamps[((size_t)bid * 4 + wave) * 8 + i] = ph[i];
        // [7]: prologue cycles for active waves, 0 for waves without queries (attn_stamps.py keys on > 0)
        g_att_stamps[((size_t)bid * 4 + wave) * 8 + 7] = active ? (ph[7] ? ph[7] : 1) : 0;
    }
#endif
}


// ---------------------------------------------------------------------------------------------------
// attention2_kernel with the score MFMAs of tile t + 1 issued in front of the softmax of tile t (round 5; ME_ATT_V=4).
// Same decomposition, operands, arithmetic and results as attention2_kernel; what changes is the order inside a wave: the
// round-4 kernel's tile is a serial chain K reads -> 8 MFMAs -> max -> 32 exponentials -> 12 MFMAs, and the phase clocks showed
// co-resident waves interleaving such chains at 40 % of what the matrix and vector pipes could issue.  Here a wave holds TWO
// score blocks: while the vector pipe works through exp / cvt of tile t the matrix pipe already has S'(t + 1) = K(t + 1) Qc^T - m'
// to do, from the same wave's instruction stream (cross-tile software pipelining; the deferred maximum makes it legal: when the
// reference point does move in tile t, the block computed ahead is corrected by the same delta).  Costs 32 registers (200: two
// waves per SIMD) and a third ring slot (K of tile t + 1 is read while V of tile t is, and tile t + 2 is in flight).
template <typename T, int MINW>
__global__ __launch_bounds__(256, MINW) void attention2p_kernel(const T* __restrict__ qkv, T* __restrict__ out, int tokens, int heads,
                                                             int ngroups, RowSegs segs, uint8_t* __restrict__ out8,
                                                             uint8_t* __restrict__ out8_scale, int64_t out8_mt, float defer_thr) {
    typedef typename Mfma32<T>::frag frag;
    constexpr int NSLOT = 3;
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * 2 * TILE_BYTES];  // slot: K tile, V tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int bid = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int C = heads * 64;
    const int ldq = 3 * C;
    const int nqb = (tokens + 127) >> 7;
    const int xcd = bid & 7, slot0 = bid >> 3;
    const int group = xcd + 8 * (slot0 / nqb);  // = win * heads + head
    const int qblk = slot0 - (slot0 / nqb) * nqb;
    if (group >= ngroups) return;  // uniform: the whole workgroup leaves before any barrier
    const int win = group / heads, head = group - win * heads;
    const int q0 = qblk * 128 + wave * 32;
    int64_t row0 = (int64_t)win * tokens;
    if (segs.seg1 && win >= segs.win0)
        row0 = win < segs.win0 + segs.win1 ? segs.seg1 + (int64_t)(win - segs.win0) * tokens
                                           : segs.seg2 + (int64_t)(win - segs.win0 - segs.win1) * tokens;
    const T* qbase = qkv + head * 64;
    const T* kbase = qkv + C + head * 64;
    const T* vbase = qkv + 2 * C + head * 64;
    // K/V staging by LDS-DMA (see attention2_kernel): wave w stages pieces 2w and 2w + 1 of K and of V
    const int st_row = lane >> 3, st_slot = lane & 7;
    const unsigned row_bytes = (unsigned)ldq * 2u;
    unsigned koff0, voff0;
    {
        const int rr = (2 * wave) * 8 + st_row;
        koff0 = (unsigned)rr * row_bytes + ((st_slot ^ ((rr >> 1) & 7)) << 4);
        voff0 = (unsigned)rr * row_bytes + ((st_slot ^ (((rr >> 1) & 1) << 2)) << 4);
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    const char* kwin = uniform_ptr((const char*)(kbase + row0 * ldq));
    const char* vwin = uniform_ptr((const char*)(vbase + row0 * ldq));
    const int nkt = (tokens + KT - 1) / KT;
    auto stage = [&](int kt, int slot) {  // 4 LDS-DMA instructions per wave and tile
        const unsigned dst = smem_base + slot * (2 * TILE_BYTES) + (2 * wave) * 1024;
        const char* kt_k = uniform_ptr(kwin + (size_t)kt * KT * row_bytes);
        const char* kt_v = uniform_ptr(vwin + (size_t)kt * KT * row_bytes);
        if ((kt + 1) * KT <= tokens) {
            unsigned k0 = koff0, v0 = voff0;
            asm volatile("" : "+v"(k0), "+v"(v0));
            glds16_raw(kt_k, k0, dst);
            glds16_raw(kt_v, v0, dst + TILE_BYTES);
            glds16_raw(kt_k, (k0 ^ 64u) + 8u * row_bytes, dst + 1024);
            glds16_raw(kt_v, v0 + 8u * row_bytes, dst + TILE_BYTES + 1024);
        } else {  // the ragged last tile: rows past the end read the last row (masked or unused below)
            const int last = tokens - 1 - kt * KT;  // >= 0
            int sr = st_row, ss = st_slot;
            asm volatile("" : "+v"(sr), "+v"(ss));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rr = (2 * wave + i) * 8 + sr;
                const unsigned rowb = (unsigned)(rr < last ? rr : last) * row_bytes;
                glds16_raw(kt_k, rowb + ((ss ^ ((rr >> 1) & 7)) << 4), dst + i * 1024);
                glds16_raw(kt_v, rowb + ((ss ^ (((rr >> 1) & 1) << 2)) << 4), dst + TILE_BYTES + i * 1024);
            }
        }
    };
    const int il = lane & 15;
    int k_lane, v_lane;
    {
        const int k_swz = (r >> 1) & 7;
        k_lane = r * 128 + ((h ^ k_swz) << 4);
        const int v_qrow = il >> 2, v_p = il & 3, v_dhalf = (lane >> 4) & 1;
        v_lane = (4 * h + v_qrow) * 128 + ((2 * v_dhalf + (v_p >> 1)) << 4) + ((v_p & 1) << 3) + (((v_qrow >> 1) & 1) << 6);
    }
    f32x16 o[2], lsum, negm;
    o[0] = f32x16{0};
    o[1] = f32x16{0};
    lsum = f32x16{0};
    negm = f32x16{0};
    float m_run = 0.f;  // the reference point (exp2 units); negm == -m_run in all sixteen registers
    frag ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;

    const bool tail_key = (tokens % KT) == 1 && nkt >= 2;
    const int nfull = tail_key ? nkt - 1 : nkt;  // tiles that run on the matrix pipe
    stage(0, 0);
    if (nkt > 1) stage(1, 1);

    // Q fragments: B operand, lane holds Qc[q0 + r][16 s + 8 h + 0..7], Qc = Q * scale * log2(e)
    frag qf[4];
    {
        int q = q0 + r;
        q = q < tokens ? q : tokens - 1;
        const T* qp = qbase + (row0 + q) * ldq + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const frag*>(qp + 16 * s);
        typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            i32x4 t = __builtin_bit_cast(i32x4, qf[s]);
            asm volatile("" : "+v"(t));
            qf[s] = __builtin_bit_cast(frag, t);
        }
    }
    const bool active = q0 < tokens;

    // S'(kt) = K(kt) Qc^T - m' from ring slot kt % 3; the ragged last tile's keys past the end become -inf
    auto scores = [&](int kt, f32x16 (&s)[2]) {
        unsigned ka = smem_base + (kt % NSLOT) * (2 * TILE_BYTES) + (unsigned)k_lane;
        asm volatile("" : "+v"(ka));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const frag kf = lds_read_frag<frag>((ka ^ (unsigned)(st << 5)) + ks * 32 * 128);
                s[ks] = Mfma32<T>::run(kf, qf[st], st == 0 ? negm : s[ks]);
            }
        if ((kt + 1) * KT > tokens) {  // uniform: the ragged tile
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int key = kt * KT + ks * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (key >= tokens) s[ks][g] = -INFINITY;
                }
        }
    };

    // tile 0's scores in front of the loop
    f32x16 sc[2], sn[2];
    wait_vmcnt<4>();  // tile 0 has landed (tile 1's four requests may still be in flight)
    if (nkt == 1) wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (active) scores(0, sc);
    for (int kt = 0; kt < nfull; ++kt) {
        // tile kt + 1 has landed in every wave's view; the slot of tile kt - 1 is free for tile kt + 2
        if (kt + 1 < nkt) {
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nkt) stage(kt + 2, (kt + 2) % NSLOT);
        }
        if (active) {
            // ---- the NEXT tile's scores first: the matrix pipe works on them while the vector pipe does this tile's softmax.
            // (Not in the first tile: its maximum SETS the reference point, and the next block then starts its accumulators
            // from it like every later one -- computed ahead it would be corrected by a subtraction instead, the same value
            // to the last rounding but not the same bits as attention2_kernel's.)
            const bool has_next = kt + 1 < nfull;
            if (has_next && kt != 0) scores(kt + 1, sn);
            // ---- does the running maximum grow?
            float mloc = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 16; ++g) mloc = fmaxf(mloc, sc[ks][g]);
            {
                float a = mloc, b = mloc;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
                mloc = fmaxf(a, b);
            }
            if (kt == 0 || __any(mloc > defer_thr)) {
                const float delta = kt == 0 ? mloc : fmaxf(mloc, 0.f);
                if (kt != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int g = 0; g < 16; ++g) o[0][g] *= alpha, o[1][g] *= alpha, lsum[g] *= alpha;
                }
                m_run += delta;
#pragma unroll
                for (int g = 0; g < 16; ++g) negm[g] = -m_run;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int g = 0; g < 16; ++g) sc[ks][g] -= delta;
                if (has_next && kt != 0) {  // the block computed ahead used the old reference point
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int g = 0; g < 16; ++g) sn[ks][g] -= delta;
                }
            }
            if (has_next && kt == 0) scores(1, sn);
            // ---- p = exp2(S')
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 16; ++g) sc[ks][g] = __builtin_amdgcn_exp2f(sc[ks][g]);
            // ---- O^T += V^T P^T,  l += 1^T P^T
            unsigned va = smem_base + (kt % NSLOT) * (2 * TILE_BYTES) + TILE_BYTES + (unsigned)v_lane;
            asm volatile("" : "+v"(va));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    frag pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (T)sc[ks][8 * s2 + j];
                    lsum = Mfma32<T>::run(ones, pf, lsum);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const unsigned vp = (va ^ (unsigned)(d << 6)) + (ks * 32 + 16 * s2) * 128;
                        const s16x4 half0 = lds_read_tr16_at(vp), half1 = lds_read_tr16_at(vp + 8 * 128);
                        typedef short s16x8 __attribute__((__vector_size__(16)));
                        const s16x8 both = __builtin_shufflevector(half0, half1, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[d] = Mfma32<T>::run(__builtin_bit_cast(frag, both), pf, o[d]);
                    }
                }
            sc[0] = sn[0], sc[1] = sn[1];
        }
    }
    float l_tot = lsum[0];
    if (tail_key) {
        // 577 = 9 x 64 + 1: the single key of the last tile (row 0 of slot (nkt - 1) % 3; its DMA was waited for and the
        // workgroup met in the loop's last iteration) as a rank-one update
        if (active) {
            const char* kb = smem + ((nkt - 1) % NSLOT) * (2 * TILE_BYTES);
            const char* vb = kb + TILE_BYTES;
            float dot = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const frag kf = *reinterpret_cast<const frag*>(kb + ((2 * st + h) << 4));  // row 0: no swizzle
#pragma unroll
                for (int j = 0; j < 8; ++j) dot = __builtin_fmaf((float)kf[j], (float)qf[st][j], dot);
            }
            dot += __shfl_xor(dot, 32);
            const float rel = dot - m_run;
            const float delta = fmaxf(rel, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            const float p16 = (float)(T)__builtin_amdgcn_exp2f(rel - delta);
            l_tot = l_tot * alpha + p16;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    typedef T v4 __attribute__((ext_vector_type(4)));
                    const v4 vv = *reinterpret_cast<const v4*>(vb + ((4 * d + g4) << 4) + 8 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[d][4 * g4 + e] = __builtin_fmaf(p16, (float)vv[e], o[d][4 * g4 + e] * alpha);
                }
        }
    }
    // ---- normalise and store (as attention2_kernel)
    auto round16 = [](float x) -> T {
        asm volatile("" : "+v"(x));
        return (T)x;
    };
    const float inv = 1.0f / l_tot;
    const int q = q0 + r;
    if (out8) {
        const int64_t m = row0 + q;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float v[16];
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e] = (float)round16(o[d][e] * inv);
                amax = fmaxf(amax, fabsf(v[e]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const unsigned sb = mx_scale_byte(amax);
            const float scl = mx_inv_scale(sb);
            if (q < tokens) {
                uint8_t* op = out8 + m * C + head * 64 + d * 32 + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<unsigned*>(op + 8 * g4) =
                        pack_fp8x4(v[4 * g4] * scl, v[4 * g4 + 1] * scl, v[4 * g4 + 2] * scl, v[4 * g4 + 3] * scl);
                if (h == 0) out8_scale[a_scale_index(m, head * 2 + d, out8_mt)] = (uint8_t)sb;
            }
        }
    } else if (q < tokens) {
        T* op = out + (row0 + q) * C + head * 64 + 4 * h;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef T v4 __attribute__((ext_vector_type(4)));
                v4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round16(o[d][4 * g4 + e] * inv);
                *reinterpret_cast<v4*>(op + d * 32 + 8 * g4) = v;
            }
    }
}

}  // namespace

#ifdef ME_ATT_STAMPS
extern "C" int32_t me_debug_set_att_stamps(void* dev_ptr) {
    unsigned long long* p = (unsigned long long*)dev_ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_att_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
extern "C" int32_t me_debug_set_att_mode(int32_t mode) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_att_mode), &mode, sizeof(mode)) == hipSuccess ? 0 : 1;
}
#endif

void attention_launch(const void* qkv, void* out, int32_t windows, int32_t tokens, int32_t heads,
                      int32_t dtype, hipStream_t stream, const RowSegs* segs_opt, uint8_t* out8,
                      uint8_t* out8_scale, int64_t out8_mt, bool q_prescaled) {
    const RowSegs segs = segs_opt ? *segs_opt : RowSegs();
    ME_CHECK(windows > 0 && tokens > 0 && heads > 0, ME_ERR_BAD_SHAPE,
             "attention: windows=%d tokens=%d heads=%d", windows, tokens, heads);
    ME_CHECK(!out8 || (out8_scale && out8_mt > 0 && heads % 2 == 0), ME_ERR_BAD_ARG,
             "attention: fp8 output needs its scale buffer and K = 64 heads a multiple of 128");
    ME_CHECK((int64_t)windows * heads * ((tokens + 127) / 128) < (1ll << 30), ME_ERR_BAD_SHAPE,
             "attention: grid too large");
    const int nqb = (tokens + 127) / 128, ngroups = windows * heads;
    const dim3 grid(8 * nqb * ((ngroups + 7) / 8));
    // ME_ATT_V=3 (development, tools/attn_ab.py): round 5's re-cut (attention3.hip: 48 queries per wave on 16x16x32 MFMAs, three
    // workgroups per (window, head), persistent, the 577th query on the vector pipe) instead of attention2_kernel.  Measured
    // equal stand-alone and slower in the step (profiles/r05_attention_recut_ab.txt), so attention2_kernel stays the forward
    // pass's kernel.
    const char* av = getenv("ME_ATT_V");
    const bool use_v2 = !(av && atoi(av) == 3);
    const bool use_pipe = av && atoi(av) == 4;  // attention2p_kernel: the next tile's score MFMAs in front of this tile's softmax
    ProfScope prof(stream, !q_prescaled ? "attention_kernel" : (use_v2 ? "attention2_kernel" : "attention3_kernel"),
                   4.0 * windows * heads * (double)tokens * tokens * 64,
                   (double)windows * tokens * heads * 64 * (out8 ? 7.03 : 8.0));  // q, k, v read once + the output
    // scale = 1/sqrt(64) (vit.rs:47), folded with log2(e) so the softmax runs on exp2
    const float scale_log2e = 0.125f * 1.44269504088896340736f;
    if (q_prescaled) {
        // attention2_kernel.  ME_ATT_THR (development): the deferred-maximum threshold in exp2 units; 0 = the
        // reference point is the exact running maximum (tools/attn_ab.py compares the two)
        const char* th = getenv("ME_ATT_THR");
        const float defer_thr = th ? (float)atof(th) : 8.0f;
        if (!use_v2) {
            attention3_launch(qkv, out, windows, tokens, heads, dtype, stream, segs, out8, out8_scale, out8_mt, defer_thr);
            return;
        }
        if (use_pipe) {
            const int nqb_p = (tokens + 127) / 128;
            const dim3 grid_p(8 * nqb_p * ((windows * heads + 7) / 8));
            const char* mw = getenv("ME_ATT_MINW");  // (development: 2 = 200 registers, no scratch; default 3 = 168 registers, 16 bytes of scratch)
            const bool two = mw && atoi(mw) == 2;
#define ME_ATT2P(T, MW)                                                                                                     \
    hipLaunchKernelGGL((attention2p_kernel<T, MW>), grid_p, dim3(256), 0, stream, (const T*)qkv, (T*)out, tokens, heads, \
                       windows * heads, segs, out8, out8_scale, out8_mt, defer_thr)
            if (dtype == ME_DTYPE_F16) {
                if (two) ME_ATT2P(f16, 2);
                else ME_ATT2P(f16, 3);
            } else if (dtype == ME_DTYPE_BF16) {
                if (two) ME_ATT2P(bf16, 2);
                else ME_ATT2P(bf16, 3);
            } else {
                fail(ME_ERR_BAD_ARG, "attention: bad dtype %d", dtype);
            }
#undef ME_ATT2P
            ME_HIP(hipGetLastError());
            return;
        }
        // Two forms (HALVES above), within 1.5 % of each other from 37 to 296 windows (profiles/r04_attention_ablations.txt:
        // three waves per SIMD with the leaner tile against four with the vector-pipe row sums).  ONE form runs at every
        // size -- their row sums round differently, and a batch must equal a loop of batch-one calls bit for bit;
        // ME_ATT_HALVES=1 (development) selects the other.
        const char* hv = getenv("ME_ATT_HALVES");
        const bool halves = hv && atoi(hv) != 0;
#define ME_ATT2(T)                                                                                                          \
    do {                                                                                                                    \
        if (halves)                                                                                                         \
            hipLaunchKernelGGL((attention2_kernel<T, 4, true>), grid, dim3(256), 0, stream, (const T*)qkv, (T*)out, tokens, \
                               heads, ngroups, segs, out8, out8_scale, out8_mt, defer_thr);                                 \
        else                                                                                                                \
            hipLaunchKernelGGL((attention2_kernel<T, 3, false>), grid, dim3(256), 0, stream, (const T*)qkv, (T*)out, tokens, \
                               heads, ngroups, segs, out8, out8_scale, out8_mt, defer_thr);                                 \
    } while (0)
        if (dtype == ME_DTYPE_F16) ME_ATT2(f16);
        else if (dtype == ME_DTYPE_BF16) ME_ATT2(bf16);
        else fail(ME_ERR_BAD_ARG, "attention: bad dtype %d", dtype);
#undef ME_ATT2
        ME_HIP(hipGetLastError());
        return;
    }
    // a plain Q: the kernel that scales the scores itself (scaling and rounding Q again would cost a second rounding)
    if (dtype == ME_DTYPE_F16)
        hipLaunchKernelGGL((attention_kernel<f16, 2, 4>), grid, dim3(256), 0, stream, (const f16*)qkv,
                           (f16*)out, tokens, heads, ngroups, scale_log2e, segs, out8, out8_scale, out8_mt);
    else if (dtype == ME_DTYPE_BF16)
        hipLaunchKernelGGL((attention_kernel<bf16, 2, 4>), grid, dim3(256), 0, stream, (const bf16*)qkv,
                           (bf16*)out, tokens, heads, ngroups, scale_log2e, segs, out8, out8_scale, out8_mt);
    else
        fail(ME_ERR_BAD_ARG, "attention: bad dtype %d", dtype);
    ME_HIP(hipGetLastError());
}

}  // namespace me
