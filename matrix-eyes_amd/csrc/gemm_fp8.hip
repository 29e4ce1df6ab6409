// MX block-scaled fp8 GEMM for the ViT linears (BASELINE configs[3]) and the quantisers that feed it.
//
//   D[m][n] = sum_kb 2^(sa[m][kb] + sw[n][kb] - 254) * sum_{k in block kb} A8[m][k] * W8[n][k]
//
// A8 / W8: OCP e4m3 bytes, row-major with K contiguous; sa / sw: one e8m0 byte per 32 consecutive K elements
// (the OCP MX format); accumulate f32.  v_mfma_scale_f32_16x16x128_f8f6f4 runs at twice the 16-bit MFMA rate
// (MI355X_MICROARCH.md, Matrix cores): the same cycles per instruction as two 16x16x32 at four times the K.
//
// Kernel structure = gemm_pp_kernel (gemm_core.h): 256x256 tile, two groups of four waves, LDS-DMA staging with
// the chunk swizzle on the source address, a two-slot activation ring and a three-slot weight ring -- a K slab
// is still 128 BYTES per row, so it now holds 128 K elements and one slab is one MFMA deep.  What the operand
// layout of the scaled MFMA asks for (tools/micro/mx_mfma_discover.hip, profiles/r02_mx_mfma_layout.txt):
//   * lane (r = lane & 15, q = lane >> 4) supplies row r's bytes k = 16q .. 16q+15 and 64+16q .. 64+16q+15 of
//     the slab: exactly the two 16-byte chunks (q, q + 4) the 16-bit kernel reads for its two k-substeps, so
//     the LDS image, its swizzle and the conflict-free ds_read_b128 pattern carry over unchanged;
//   * lane (r, q) supplies the scale of (row r, K block q of the slab) in one byte of a VGPR (op_sel picks
//     the byte).  Scales are therefore stored pre-arranged per wave: for the activation operand 8 bytes per
//     lane (the 8 m-tiles of a wave's 128 rows), for the weight operand 4 bytes per lane (4 n-tiles of 64
//     columns) -- one dwordx2 and one dword load per wave per slab, straight to registers.  They are issued
//     as asm (invisible to the compiler's waitcnt pass, like the LDS-DMA) at the top of the slab BEFORE the
//     one they are for, and retired by the kernel's own closing waits, which name the registers as operands.
#include <cstdlib>
#include <string>

#include "gemm_core.h"

namespace me {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int OA, int OB>
__device__ __forceinline__ f32x4 mx_mfma(v8i a, v8i b, f32x4 c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, sa, OB, sb);
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

#define ME_PIN_ACC(g0)                                                                                         \
    asm volatile(""                                                                                            \
                 : "+v"(acc[g0][0]), "+v"(acc[g0][1]), "+v"(acc[g0][2]), "+v"(acc[g0][3]), "+v"(acc[g0 + 1][0]), \
                   "+v"(acc[g0 + 1][1]), "+v"(acc[g0 + 1][2]), "+v"(acc[g0 + 1][3]), "+v"(acc[g0 + 2][0]),     \
                   "+v"(acc[g0 + 2][1]), "+v"(acc[g0 + 2][2]), "+v"(acc[g0 + 2][3]), "+v"(acc[g0 + 3][0]),     \
                   "+v"(acc[g0 + 3][1]), "+v"(acc[g0 + 3][2]), "+v"(acc[g0 + 3][3]))

template <int EPI, int OUT8>
__global__ __launch_bounds__(512, 2) void gemm_pp8_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256, WN = 4, HW = 4;
    constexpr int TM = 128, TN = 64, MI = 8, NI = 4;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, W_RING = 2 * A_BYTES;
    constexpr int IT = 8;  // LDS-DMA instructions per wave per operand slab
    constexpr int MI_CH = epi_mi_chunk(MI, TN, HW, A_BYTES);
    constexpr int SCR = 16 * MI_CH * (TN * 4);
    static_assert(HW * SCR <= A_BYTES, "epilogue scratch exceeds a ring slot");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2, gw = wave & 3;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    const int srow = lane >> 3, sslot = lane & 7;
    const int64_t sa_kstride = (int64_t)p.a_mt * 512, sw_kstride = (int64_t)(p.N / 64) * 256;

    // M is a multiple of 256 (checked at launch): no row clamping, so a wave's per-lane DMA offsets are the same
    // for every tile -- 8 VGPRs for the whole kernel; what changes per tile is uniform (SGPRs)
    unsigned soff[IT];
    {
        const int64_t ld = group == 0 ? p.lda : p.K;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int row = (i * HW + gw) * 8 + srow;
            soff[i] = (unsigned)((int64_t)row * ld) + (sslot ^ ((row >> 1) & 7)) * 16;
        }
    }
    struct Src {
        int m0, n0;
        const char* base;     // this wave's DMA operand: activation rows (group 0) or weight rows (group 1)
        const char *sa, *sw;  // this wave's scale blocks of slab 0
    };
    auto setup = [&](Src& t, int vb) {
        tile_origin<BM, BN>(p, vb, ntiles, t.m0, t.n0);
        const int seg = row_segment(p, t.m0);
        t.base = group == 0 ? (const char*)p.A + (int64_t)t.m0 * p.lda : segment_weights(p, t.m0) + (int64_t)t.n0 * p.K;
        t.sa = (const char*)p.a_scale + (int64_t)(t.m0 / 128 + wm) * 512;
        const uint8_t* ws = seg == 0 ? p.w_scale : (seg == 1 ? p.w_scale_s1 : p.w_scale_s2);
        t.sw = (const char*)ws + (int64_t)(t.n0 / 64 + wn) * 256;
    };

    const int nk = p.K / 128;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    auto stage_T = [&](const Src& t, int kt, int slot) {
        const char* base = uniform_ptr(t.base + (int64_t)kt * 128);
        const unsigned dst = smem_base + slot * A_BYTES + gw * 1024;
#pragma unroll
        for (int i = 0; i < IT; ++i) glds16_raw(base, soff[i], dst + i * (HW * 1024));
    };
    auto stage_W = [&](const Src& t, int kt, int slot) {
        const char* base = uniform_ptr(t.base + (int64_t)kt * 128);
        const unsigned dst = smem_base + W_RING + slot * B_BYTES + gw * 1024;
#pragma unroll
        for (int i = 0; i < IT; ++i) glds16_raw(base, soff[i], dst + i * (HW * 1024));
    };
    // Scale registers.  sa_cur / sw_cur: the slab being consumed.  nsa / nsw: the next slab's, in flight -- two VMEM
    // loads per wave issued as asm, so between that statement and the one that retires them the registers hold
    // garbage the compiler believes to be data.  It must never read them there: the retiring statement takes them
    // as plain inputs and does wait + copy inside ONE asm (a separate wait statement with "+v" operands got its
    // input/output copies placed in front of the wait on one path: the copies read the registers before the data
    // had landed -- wrong scales whenever the loads were slow, i.e. from cold caches).
    unsigned long long sa_cur = 0, nsa = 0;
    unsigned sw_cur = 0, nsw = 0;
    auto load_scales = [&](const Src& t, int kt) {
        const char* a = uniform_ptr(t.sa + (int64_t)kt * sa_kstride);
        const char* w = uniform_ptr(t.sw + (int64_t)kt * sw_kstride);
        // s_nop 4: the bases may come straight from v_readfirstlane (VALU-written SGPR -> VMEM base)
        asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %3\n\tglobal_load_dword %1, %4, %5"
                     : "=&v"(nsa), "=&v"(nsw)
                     : "v"(lane * 8), "s"(a), "v"(lane * 4), "s"(w)
                     : "memory");
    };
#define ME_RETIRE_SCALES(N)                                                                              \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_mov_b64 %0, %2\n\tv_mov_b32 %1, %3\n\ts_nop 1"            \
                 : "=&v"(sa_cur), "=&v"(sw_cur)                                                          \
                 : "v"(nsa), "v"(nsw)                                                                    \
                 : "memory")

    const int frow = lane & 15, fswz = frow >> 1;
    const int fslot0 = ((lane >> 4) ^ fswz) * 16;
    const int fslot1 = (((lane >> 4) + 4) ^ fswz) * 16;
    const int a_rd = (wm * TM + frow) * 128;
    const int b_rd = (wn * TN + frow) * 128;

    if (group == 1) __builtin_amdgcn_s_setprio(1);
    int vb = blockIdx.x;
    Src cur, nxt;
    setup(cur, vb);
    int next_vb = vb + (int)gridDim.x;
    if (group == 0) {
        stage_T(cur, 0, 0);
        load_scales(cur, 0);
        ME_RETIRE_SCALES(0);
    } else {
        load_scales(cur, 0);
        stage_W(cur, 0, 0);
        stage_W(cur, 1, 1);
        ME_RETIRE_SCALES(8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int ts = 0, ws = 0;

    while (true) {
        const bool has_next = next_vb < ntiles;
        if (has_next) setup(nxt, next_vb);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            // scales of THIS slab (retired by the previous slab's closing wait)
            const int sa_lo = (int)(unsigned)sa_cur, sa_hi = (int)(unsigned)(sa_cur >> 32), sw = (int)sw_cur;
            if (group == 0) {
                if (kt + 1 < nk)
                    stage_T(cur, kt + 1, ts ^ 1);
                else if (has_next)
                    stage_T(nxt, 0, ts ^ 1);
            }
            const bool more = kt + 1 < nk || has_next;
            if (more) {
                if (kt + 1 < nk)
                    load_scales(cur, kt + 1);
                else
                    load_scales(nxt, 0);
            }
            const char* sa = smem + ts * A_BYTES + a_rd;
            const char* sw_lds = smem + W_RING + ws * B_BYTES + b_rd;
            auto rd = [&](const char* base, int tile) -> v8i {
                const v4i lo = *reinterpret_cast<const v4i*>(base + tile * 2048 + fslot0);
                const v4i hi = *reinterpret_cast<const v4i*>(base + tile * 2048 + fslot1);
                return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            // Fragment feed: the NI weight fragments and two activation fragments up front, then m-tile g's NI
            // MFMAs with the reads of m-tile g + 2 beside them (pinned: the scheduler otherwise hoists every LDS
            // read to the top and the 8 activation fragments alone take 64 registers)
            v8i wf[NI], af[MI];
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = rd(sw_lds, j);
            af[0] = rd(sa, 0);
            af[1] = rd(sa, 1);
            // first half of the slab: m-tiles 0..3
            static_for<0, MI / 2>([&](auto G) {
                constexpr int g = decltype(G)::value;
                if constexpr (g + 2 < MI) af[g + 2] = rd(sa, g + 2);
                static_for<0, NI>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    acc[g][j] = mx_mfma<j, g & 3>(wf[j], af[g], acc[g][j], sw, sa_lo);
                });
            });
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NI + 4, 0);
#pragma unroll
            for (int g = 0; g < MI / 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NI, 0);
            }
            // An MFMA touches no memory, so the asm statements' "memory" clobbers do not hold it in place, and the
            // optimiser sinks all 32 of them below the closing barrier (next to their only user, the next slab's
            // MFMAs).  Volatile asm statements keep their order among themselves: an empty one that "modifies" the
            // half's accumulators pins their MFMAs above the mid-slab DMA issue / the closing waits.
            ME_PIN_ACC(0);
            bool newer = false;
            if (group == 1) {
                const int w2 = ws == 0 ? 2 : ws - 1;
                if (kt + 2 < nk) {
                    stage_W(cur, kt + 2, w2);
                    newer = true;
                } else if (has_next) {
                    stage_W(nxt, kt + 2 - nk, w2);
                    newer = true;
                }
            }
            static_for<MI / 2, MI>([&](auto G) {
                constexpr int g = decltype(G)::value;
                if constexpr (g + 2 < MI) af[g + 2] = rd(sa, g + 2);
                static_for<0, NI>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    acc[g][j] = mx_mfma<j, g & 3>(wf[j], af[g], acc[g][j], sw, sa_hi);
                });
            });
#pragma unroll
            for (int g = MI / 2; g < MI; ++g) {
                if (g + 2 < MI) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NI, 0);
            }
            ME_PIN_ACC(4);
            if (newer)
                ME_RETIRE_SCALES(8);
            else
                ME_RETIRE_SCALES(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ts ^= 1;
            ws = ws == 2 ? 0 : ws + 1;
        }
        {
            char* scratch0 = smem + (ts ^ 1) * A_BYTES;
            const int wprev = ws == 0 ? 2 : ws - 1;
            char* scr = (group == 0 ? scratch0 : smem + W_RING + wprev * B_BYTES) + gw * SCR;
            gemm_epilogue<f16, EPI, MI, NI, TM, TN, MI_CH, true, NoHook, OUT8, false, 16, false>(p, acc, cur.m0, cur.n0, wm, wn, lane, scr);
        }
        if (!has_next) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur = nxt;
        vb = next_vb;
        next_vb = vb + (int)gridDim.x;
    }
}

template <int EPI, int OUT8>
static void launch_pp8(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = (2 * 256 + 3 * 256) * 128;
    auto kern = gemm_pp8_kernel<EPI, OUT8>;
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
    const int64_t ntiles = cdiv(p.M, 256) * (p.N / 256);
    const int64_t grid = ntiles < resident ? ntiles : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, p);
    ME_HIP(hipGetLastError());
}

// ---- the 352-row tile ------------------------------------------------------------------------------------
// gemm_pp_kernel's tall form (gemm_core.h: 2 x 4 waves of 176 x 64, two activation slots + TWO weight slots = 152 KiB, row tiles
// laid out per row segment, DMA source offsets {first, step, clamp}) with the scaled fp8 MFMA: at one image proj / fc2 are ONE
// round of 256 tiles where the 256-row tile needs 340 (1.33 rounds), fc1 is four rounds instead of 5.3 -- and the residual
// epilogue can carry the LayerNorm of the rows it updates (LNF: resid_ln_epilogue, written for exactly this wave layout), which
// on the 256-row tile it could not.  A slab is 128 bytes per row = one MFMA deep, 44 MFMAs per wave.
// Activation scales: the layout is per 128-row block (8 m-tiles = the 8 bytes of a lane), a wave's 11 m-tiles start at m-tile
// t0 = (m0 + 176 wm) / 16 -- anywhere in a block.  Three loads per slab (dwordx2, dwordx2, dword: blocks t0 / 8 .. + 2, clamped
// to the operand's last block) give 20 bytes per lane, byte (t0 & 7) + i of them belongs to m-tile i; op_sel is an immediate, so
// the string is shifted down by t0 & 7 bytes with v_alignbyte (three per slab) and m-tile i reads byte i & 3 of register i / 4.
#define ME_PIN_ROW(g) asm volatile("" : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2]), "+v"(acc[g][3]))
template <int EPI, int OUT8, bool LNF>
__global__ __launch_bounds__(512, 2) void gemm_pp8t_kernel(const GemmParams p) {
    static_assert(true, "MH = 6 below: ME_PIN_ROW(0 .. 5) close the first half of a slab, (6 .. 10) the second");
    static_assert(!LNF || EPI == EPI_RESID_SCALE, "LayerNorm fusion: the residual epilogue");
    constexpr int BM = 352, BN = 256, WN = 4, HW = 4;
    constexpr int TM = 176, TN = 64, MI = 11, NI = 4, MH = 6;  // MH: m-tiles of the first half of a slab
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, W_RING = 2 * A_BYTES;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;  // LDS-DMA instructions per wave per operand slab
    constexpr int MI_CH = epi_mi_chunk(MI, TN, HW, B_BYTES);
    constexpr int SCR = 16 * MI_CH * (TN * 4);
    static_assert(HW * SCR <= B_BYTES, "epilogue scratch exceeds a ring slot");
    static_assert(HW * SCR + BM * (WN * 8 + 8) <= A_BYTES, "row statistics beside the epilogue scratch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2, gw = wave & 3;
    const int wm = wave / WN, wn = wave % WN;
    // the tile walks of gemm_pp_kernel: super-rows of 8 row tiles, or (LNF) whole row tiles per XCD and round
    const int nbn = p.N / BN, nbm = seg_row_tiles<BM>(p.M, p.seg1, p.seg2);
    auto lnf_row_tile = [&](int vb) { return ((vb >> 3) / nbn) * 8 + (vb & 7); };
    const int ntiles = LNF ? ((nbm + 7) / 8) * 8 * nbn : nbm * nbn;
    if constexpr (LNF) {
        if (lnf_row_tile(blockIdx.x) >= nbm) return;  // uniform; this workgroup's later tiles do not exist either
    }
    const int srow = lane >> 3, sslot = lane & 7;
    const int64_t sa_kstride = (int64_t)p.a_mt * 512, sw_kstride = (int64_t)(p.N / 64) * 256;

    struct Src {
        int m0, n0, m_lim, row_tile;
        const char* base;  // this wave's DMA operand: activation rows (group 0) or weight rows (group 1)
        int last;          // group 0: last row of the operand relative to the tile's first (rows beyond it read that row)
        const char* sw;    // this wave's weight scale block of slab 0
        int sblk[3];       // byte offsets of the three activation scale blocks within a slab's scales
        int shift;         // t0 & 7
    };
    auto setup = [&](Src& t, int vb) {
        if constexpr (LNF) {
            t.row_tile = lnf_row_tile(vb);
            t.n0 = ((vb >> 3) % nbn) * BN;
            seg_tile_rows<BM>(p, t.row_tile, t.m0, t.m_lim);
        } else {
            tile_origin<BM, BN, true>(p, vb, ntiles, t.m0, t.n0, &t.m_lim, &t.row_tile);
        }
        const int seg = row_segment(p, t.m0);
        if (group == 0) {
            t.base = (const char*)p.A + (int64_t)t.m0 * p.lda;
            t.last = p.M - 1 - t.m0;  // >= 0
        } else {
            t.base = segment_weights(p, t.m0) + (int64_t)t.n0 * p.K;  // N is a multiple of 256: every weight row exists
            t.last = BN;
        }
        const int t0 = (t.m0 + wm * TM) >> 4;
        t.shift = t0 & 7;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int b = (t0 >> 3) + q;
            t.sblk[q] = (b < p.a_mt ? b : p.a_mt - 1) * 512;
        }
        const uint8_t* ws = seg == 0 ? p.w_scale : (seg == 1 ? p.w_scale_s1 : p.w_scale_s2);
        t.sw = (const char*)ws + (int64_t)(t.n0 / 64 + wn) * 256;
    };

    const int nk = p.K / 128;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    // piece i of a lane: row (i * HW + gw) * 8 + srow of the tile, always the same 16-byte chunk position (the swizzle repeats
    // every 16 rows): offsets affine in i from ONE per-lane offset that is the same for every tile; the clamp is per tile (uniform)
    const unsigned ld_op = (unsigned)(group == 0 ? p.lda : p.K);
    const unsigned chunk16 = (unsigned)(sslot ^ (((gw * 8 + srow) >> 1) & 7)) * 16u;
    const unsigned off0 = (unsigned)(gw * 8 + srow) * ld_op + chunk16;
    auto stage_T = [&](const Src& t, int kt, int slot) {
        const char* base = uniform_ptr(t.base + (int64_t)kt * 128);
        const unsigned dst = smem_base + slot * A_BYTES + gw * 1024;
        const unsigned step = (unsigned)(HW * 8) * ld_op;
        unsigned first = off0;
        asm volatile("" : "+v"(first));  // recomputed per slab (two VALU per piece): hoisted, the offsets spill
        const unsigned lim = t.last >= BM ? 0xffffffffu : (unsigned)t.last * ld_op + chunk16;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16_raw(base, min(first + i * step, lim), dst + i * (HW * 1024));
    };
    auto stage_W = [&](const Src& t, int kt, int slot) {
        const char* base = uniform_ptr(t.base + (int64_t)kt * 128);
        const unsigned dst = smem_base + W_RING + slot * B_BYTES + gw * 1024;
        const unsigned step = (unsigned)(HW * 8) * ld_op;
        unsigned first = off0;
        asm volatile("" : "+v"(first));
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16_raw(base, first + i * step, dst + i * (HW * 1024));
    };
    // Scale registers as in gemm_pp8_kernel: issued as asm one slab ahead, retired at the slab's close by ONE asm that waits and
    // leaves the slab's scales ready for use: the five activation dwords shifted down to m-tile 0 (sas: bytes of m-tiles 0-3, 4-7,
    // 8-10) and the weight dword.  (Dword loads: the halves of a 64-bit asm operand cannot be named.)
    int sas0 = 0, sas1 = 0, sas2 = 0, sw_cur = 0;
    unsigned nd0 = 0, nd1 = 0, nd2 = 0, nd3 = 0, nd4 = 0, nsw = 0;
    auto load_scales = [&](const Src& t, int kt) {
        const char* a = (const char*)p.a_scale + (int64_t)kt * sa_kstride;
        const char* a0 = uniform_ptr(a + t.sblk[0]);
        const char* a1 = uniform_ptr(a + t.sblk[1]);
        const char* a2 = uniform_ptr(a + t.sblk[2]);
        const char* w = uniform_ptr(t.sw + (int64_t)kt * sw_kstride);
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %6, %7\n\tglobal_load_dword %1, %6, %7 offset:4\n\t"
                     "global_load_dword %2, %6, %8\n\tglobal_load_dword %3, %6, %8 offset:4\n\t"
                     "global_load_dword %4, %6, %9\n\tglobal_load_dword %5, %10, %11"
                     : "=&v"(nd0), "=&v"(nd1), "=&v"(nd2), "=&v"(nd3), "=&v"(nd4), "=&v"(nsw)
                     : "v"(lane * 8), "s"(a0), "s"(a1), "s"(a2), "v"(lane * 4), "s"(w)
                     : "memory");
    };
    // shift: t0 & 7 of the tile the retired scales belong to
    auto retire_scales = [&](int shift) {
        const int by = shift & 3;
        if (shift >= 4)
            asm volatile("s_waitcnt vmcnt(0)\n\tv_alignbyte_b32 %0, %6, %5, %9\n\tv_alignbyte_b32 %1, %7, %6, %9\n\t"
                         "v_alignbyte_b32 %2, %8, %7, %9\n\tv_mov_b32 %3, %10\n\ts_nop 1"
                         : "=&v"(sas0), "=&v"(sas1), "=&v"(sas2), "=&v"(sw_cur)
                         : "v"(nd0), "v"(nd1), "v"(nd2), "v"(nd3), "v"(nd4), "s"(by), "v"(nsw)
                         : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\tv_alignbyte_b32 %0, %5, %4, %9\n\tv_alignbyte_b32 %1, %6, %5, %9\n\t"
                         "v_alignbyte_b32 %2, %7, %6, %9\n\tv_mov_b32 %3, %10\n\ts_nop 1"
                         : "=&v"(sas0), "=&v"(sas1), "=&v"(sas2), "=&v"(sw_cur)
                         : "v"(nd0), "v"(nd1), "v"(nd2), "v"(nd3), "v"(nd4), "s"(by), "v"(nsw)
                         : "memory");
    };

    const int frow = lane & 15, fswz = frow >> 1;
    const int fslot0 = ((lane >> 4) ^ fswz) * 16;
    const int fslot1 = (((lane >> 4) + 4) ^ fswz) * 16;
    const int a_rd = (wm * TM + frow) * 128;
    const int b_rd = (wn * TN + frow) * 128;

    if (group == 1) __builtin_amdgcn_s_setprio(1);
    int vb = blockIdx.x;
    Src cur, nxt;
    setup(cur, vb);
    int next_vb = vb + (int)gridDim.x;
    if (group == 0) {
        stage_T(cur, 0, 0);
        load_scales(cur, 0);
    } else {
        load_scales(cur, 0);
        stage_W(cur, 0, 0);
    }
    retire_scales(cur.shift);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int ts = 0, ws = 0;

    while (true) {
        const bool has_next = next_vb < ntiles && (!LNF || lnf_row_tile(next_vb) < nbm);
        if (has_next) setup(nxt, next_vb);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            // scales of THIS slab (retired by the previous slab's closing wait)
            const int sas[3] = {sas0, sas1, sas2};
            const int sw = sw_cur;
            if (group == 0) {
                if (kt + 1 < nk)
                    stage_T(cur, kt + 1, ts ^ 1);
                else if (has_next)
                    stage_T(nxt, 0, ts ^ 1);
            }
            if (kt + 1 < nk)
                load_scales(cur, kt + 1);
            else if (has_next)
                load_scales(nxt, 0);
            const char* sa = smem + ts * A_BYTES + a_rd;
            const char* sw_lds = smem + W_RING + ws * B_BYTES + b_rd;
            auto rd = [&](const char* base, int tile) -> v8i {
                const v4i lo = *reinterpret_cast<const v4i*>(base + tile * 2048 + fslot0);
                const v4i hi = *reinterpret_cast<const v4i*>(base + tile * 2048 + fslot1);
                return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            v8i wf[NI], af[MI];
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = rd(sw_lds, j);
            // (one m-tile ahead: an m-tile's NI MFMAs are 128 cycles, longer than the LDS latency; two ahead the kernel spills)
            af[0] = rd(sa, 0);
            static_for<0, MH>([&](auto G) {
                constexpr int g = decltype(G)::value;
                if constexpr (g + 1 < MI) af[g + 1] = rd(sa, g + 1);
                static_for<0, NI>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    acc[g][j] = mx_mfma<j, g & 3>(wf[j], af[g], acc[g][j], sw, sas[g >> 2]);
                });
            });
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NI + 2, 0);
#pragma unroll
            for (int g = 0; g < MH; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NI, 0);
            }
            // (volatile asm statements keep their order: the empty ones pin the half's MFMAs above the mid-slab DMA issue)
            ME_PIN_ROW(0); ME_PIN_ROW(1); ME_PIN_ROW(2); ME_PIN_ROW(3); ME_PIN_ROW(4); ME_PIN_ROW(5);
            if (group == 1) {  // two weight slots: the next slab's weights, into the slot the slab before this one has left
                if (kt + 1 < nk)
                    stage_W(cur, kt + 1, ws ^ 1);
                else if (has_next)
                    stage_W(nxt, 0, ws ^ 1);
            }
            static_for<MH, MI>([&](auto G) {
                constexpr int g = decltype(G)::value;
                if constexpr (g + 1 < MI) af[g + 1] = rd(sa, g + 1);
                static_for<0, NI>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    acc[g][j] = mx_mfma<j, g & 3>(wf[j], af[g], acc[g][j], sw, sas[g >> 2]);
                });
            });
#pragma unroll
            for (int g = MH; g < MI; ++g) {
                if (g + 1 < MI) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NI, 0);
            }
            ME_PIN_ROW(6); ME_PIN_ROW(7); ME_PIN_ROW(8); ME_PIN_ROW(9); ME_PIN_ROW(10);
            retire_scales(kt + 1 < nk ? cur.shift : nxt.shift);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ts ^= 1, ws ^= 1;
        }
        {
            char* scratch0 = smem + (ts ^ 1) * A_BYTES;
            char* scr = (group == 0 ? scratch0 : smem + W_RING + (ws ^ 1) * B_BYTES) + gw * SCR;
            if constexpr (LNF)
                resid_ln_epilogue<f16, MI, NI, TM, TN, BM>(p, acc, cur.m0, cur.n0, wm, wn, lane, tid, scr, scratch0 + HW * SCR, cur.m_lim,
                                                           cur.row_tile);
            else
                gemm_epilogue<f16, EPI, MI, NI, TM, TN, MI_CH, true, NoHook, OUT8, false, 16, false>(p, acc, cur.m0, cur.n0, wm, wn, lane,
                                                                                                     scr, NoHook(), cur.m_lim);
        }
        if (!has_next) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur = nxt;
        vb = next_vb;
        next_vb = vb + (int)gridDim.x;
    }
}

// workgroups of the tall fp8 kernel that are resident at once on the CUs `stream` may use (one per CU: 152 KiB of LDS)
template <int EPI, int OUT8, bool LNF>
static void launch_pp8t(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = (2 * 352 + 2 * 256) * 128;
    auto kern = gemm_pp8t_kernel<EPI, OUT8, LNF>;
    static PerDeviceOnce once;
    const int occ = per_device_once(once, [&](int dev) {
        ME_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int per_cu = 0, cus = 0;
        ME_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, smem));
        ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        return ((per_cu < 1 ? 1 : per_cu) << 12) | (cus < 1 ? 1 : (cus > 4095 ? 4095 : cus));
    });
    int cus = occ & 4095;
    if (p.cu_granted > 0 && p.cu_granted < cus) cus = p.cu_granted;
    int resident = (occ >> 12) * cus;
    resident -= resident % 8;
    if (resident < 8) resident = 8;
    const int64_t nbn = p.N / 256, nbm = seg_row_tiles<352>(p.M, p.seg1, p.seg2);
    const int64_t ntiles = LNF ? cdiv(nbm, 8) * 8 * nbn : nbm * nbn;  // LNF: the kernel's padded walk
    int64_t grid = ntiles < resident ? ntiles : resident;
    if (LNF) {
        const int64_t unit = 8 * nbn;  // a round is a whole number of row tiles per XCD, all of it resident (they wait for one another)
        ME_CHECK(resident >= unit, ME_ERR_HIP, "fp8 gemm: the fused LayerNorm needs %lld resident workgroups (%d fit)", (long long)unit, resident);
        grid -= grid % unit;
    }
    // (the test knob of gemm_launch_pp: a cap that splits a round makes the LayerNorm exchange time out)
    static const int grid_limit = getenv("ME_GEMM_GRID_LIMIT") ? atoi(getenv("ME_GEMM_GRID_LIMIT")) : 0;
    if (grid_limit >= 8 && grid > grid_limit) grid = grid_limit - grid_limit % 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, p);
    ME_HIP(hipGetLastError());
}

// The 352-row tile where its rounds x 352 undercut the 256-row tile's rounds x 256 (pipeline.hip tall_tile_wins, without the
// short-tail form the fp8 path does not have); ME_FP8_TALL=0 turns it off, =1 forces it.
static bool fp8_tall_wins(const GemmParams& p) {
    const char* env = getenv("ME_FP8_TALL");  // (read per launch: the tests flip it inside one process)
    const int knob = env ? atoi(env) : -1;
    if (knob == 0) return false;
    if (knob == 1) return true;
    const int64_t nbn = p.N / 256;
    const double cost352 = (double)cdiv((int64_t)seg_row_tiles<352>(p.M, p.seg1, p.seg2) * nbn, 256) * 352.0;
    const double cost256 = (double)cdiv(cdiv(p.M, 256) * nbn, 256) * 256.0;
    // a row of the tall tile costs about a tenth more (weights one slab ahead instead of two): fc1 at one image -- four rounds
    // of 352 against six of 256, 1408 : 1536 -- measured 3.40 against 3.33 ms per step on the 256-row tile
    return cost352 * 1.1 < cost256;
}

// fp8 x fp8 GEMM.  p.A / p.W: e4m3 bytes, p.lda in bytes; p.a_scale / p.w_scale (+ _s1 / _s2) in the layouts
// above with p.a_mt = (rows padded to 128) / 128.  epi: EPI_STORE (16-bit output in f16, or fp8 + scales into
// p.out8 / p.out8_scale when out8) or EPI_RESID_SCALE.
void gemm_fp8_launch(const GemmParams& p_in, EpiKind epi, hipStream_t stream) {
    GemmParams p = p_in;
    p.status = current_status_word();
    // the 256-row tile takes whole tiles only (M and the row segments multiples of 256); the 352-row tile clamps its rows, its
    // row segments start on a 16-row MFMA tile
    const bool whole256 = p.M % 256 == 0 && p.seg1 % 256 == 0 && p.seg2 % 256 == 0;
    ME_CHECK(p.M > 0 && p.N > 0 && p.N % 256 == 0 && p.K % 128 == 0 && p.K >= 256, ME_ERR_BAD_SHAPE,
             "fp8 gemm: M=%d N=%d K=%d (N a multiple of 256; K a multiple of 128 and >= 256)", p.M, p.N, p.K);
    ME_CHECK(p.A && p.W && p.a_scale && p.w_scale && p.a_mt * 128 >= p.M, ME_ERR_BAD_ARG, "fp8 gemm: operands");
    ME_CHECK(p.lda % 16 == 0 && p.lda >= p.K, ME_ERR_BAD_SHAPE, "fp8 gemm: lda=%lld", (long long)p.lda);
    ME_CHECK(p.seg1 % 16 == 0 && p.seg2 % 16 == 0 && p.seg1 >= 0 && p.seg2 >= 0 && p.seg1 <= p.M && p.seg2 <= p.M, ME_ERR_BAD_ARG,
             "fp8 gemm: row segments %d / %d of %d rows", p.seg1, p.seg2, p.M);
    // a caller mistake here would be a device fault, not a status code
    ME_CHECK(p.seg1 == 0 || (p.W_s1 && p.w_scale_s1), ME_ERR_BAD_ARG, "fp8 gemm: row segment 1 without weights / scales");
    ME_CHECK(p.seg2 == 0 || (p.seg1 != 0 && p.seg2 > p.seg1 && p.W_s2 && p.w_scale_s2), ME_ERR_BAD_ARG,
             "fp8 gemm: row segment 2 without weights / scales");
    if (epi == EPI_RESID_SCALE) {
        ME_CHECK(p.gamma && p.res32 && p.out32 && p.bias, ME_ERR_BAD_ARG, "fp8 gemm: the residual form takes bias, gamma, res32 and out32");
        ME_CHECK((p.seg1 == 0 || (p.gamma_s1 && p.bias_s1)) && (p.seg2 == 0 || (p.gamma_s2 && p.bias_s2)), ME_ERR_BAD_ARG,
                 "fp8 gemm: a row segment without bias / gamma");
    } else {
        ME_CHECK((p.seg1 == 0 || p.bias_s1) && (p.seg2 == 0 || p.bias_s2), ME_ERR_BAD_ARG, "fp8 gemm: a row segment without bias");
    }
    // the residual form with the LayerNorm of the updated rows (GemmParams::ln_out16 / out8 as in gemm_launch): the tall tile only
    const bool lnf = epi == EPI_RESID_SCALE && (p.ln_out16 || p.out8);
    if (lnf) {
        ME_CHECK((p.N == 256 || p.N == 512 || p.N == 1024) && p.ldc == p.N && p.ln_w && p.ln_b && p.ln_stats && p.ln_count, ME_ERR_BAD_ARG,
                 "fp8 gemm: the fused LayerNorm takes N in {256, 512, 1024} with its weights, statistics and counters");
        ME_CHECK(p.seg1 == 0 || (p.ln_w_s1 && p.ln_b_s1 && (p.seg2 == 0 || (p.ln_w_s2 && p.ln_b_s2))), ME_ERR_BAD_ARG,
                 "fp8 gemm: a row segment without LayerNorm weights");
        if (p.out8)
            ME_CHECK(p.out8_scale && (int64_t)p.out8_mt * 128 >= p.M, ME_ERR_BAD_ARG,
                     "fp8 gemm: the fused LayerNorm's fp8 output needs its block scales (%d tiles of 128 rows for %d rows)", p.out8_mt, p.M);
    }
    const bool tall = lnf || !whole256 || fp8_tall_wins(p);
    if (tall) {
        p.cu_granted = stream_cu_count(stream);
        static const int spin = getenv("ME_LN_SPIN_LIMIT") ? atoi(getenv("ME_LN_SPIN_LIMIT")) : 0;  // test knob
        p.ln_spin_limit = spin > 0 ? spin : 0;
    }
    const std::string pname = std::string("gemm_kernel<fp8,") + (tall ? "352" : "256") + "x256x128/8w-pp,plain," +
                              (epi == EPI_RESID_SCALE ? "resid_scale>" : "store>");
    // algorithmic bytes: operands + scales once, the output once (the residual form reads and writes the f32 rows)
    const double rows = (double)(p.flop_rows ? p.flop_rows : p.M);
    const double out_bytes = epi == EPI_RESID_SCALE ? rows * p.N * 8 + (lnf ? rows * p.N * (p.out8 ? 1.03 : 2) : 0)
                                                    : rows * p.N * (p.out8 ? 1.03 : 2);
    ProfScope prof(stream, pname.c_str(), 2.0 * rows * p.N * p.K, rows * p.K * 1.03 + (double)p.N * p.K * 1.03 + out_bytes);
    if (epi == EPI_RESID_SCALE) {
        if (lnf) launch_pp8t<EPI_RESID_SCALE, 0, true>(p, stream);
        else if (tall) launch_pp8t<EPI_RESID_SCALE, 0, false>(p, stream);
        else launch_pp8<EPI_RESID_SCALE, 0>(p, stream);
    } else if (epi == EPI_STORE) {
        if (p.out8) {
            ME_CHECK(p.out8_scale && p.out8_mt * 128 >= p.M && p.bias, ME_ERR_BAD_ARG, "fp8 gemm: fp8 output");
            if (tall) launch_pp8t<EPI_STORE, 1, false>(p, stream);
            else launch_pp8<EPI_STORE, 1>(p, stream);
        } else {
            ME_CHECK(p.out16 && !p.out32 && !p.res32 && p.bias && !p.out16_border, ME_ERR_BAD_ARG,
                     "fp8 gemm: the 16-bit output form takes bias and out16 only");
            if (tall) launch_pp8t<EPI_STORE, 0, false>(p, stream);
            else launch_pp8<EPI_STORE, 0>(p, stream);
        }
    } else {
        fail(ME_ERR_BAD_ARG, "fp8 gemm: epilogue %d", (int)epi);
    }
}

// ---- quantisers -----------------------------------------------------------------------------------------
// f16 [rows][K] -> e4m3 [rows][K] + one e8m0 per 32 K elements, in the weight (which = 1) or activation
// (which = 0) scale layout.  One thread per block of 32.
__global__ void quantize_f16_kernel(const f16* __restrict__ src, uint8_t* __restrict__ dst,
                                    uint8_t* __restrict__ scales, int64_t rows, int K, int64_t tiles, int which) {
    const int kbs = K / 32;
    const int64_t total = rows * kbs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / kbs;
        const int kb = (int)(i - r * kbs);
        const f16x8* s = reinterpret_cast<const f16x8*>(src + r * K + kb * 32);
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f16x8 h = s[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c * 8 + e] = (float)h[e], amax = fmaxf(amax, fabsf(v[c * 8 + e]));
        }
        const unsigned sb = mx_scale_byte(amax);
        const float inv = mx_inv_scale(sb);
        uint4 o[2];
        unsigned* w = reinterpret_cast<unsigned*>(o);
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c] = pack_fp8x4(v[4 * c] * inv, v[4 * c + 1] * inv, v[4 * c + 2] * inv, v[4 * c + 3] * inv);
        uint4* d = reinterpret_cast<uint4*>(dst + r * K + kb * 32);
        d[0] = o[0], d[1] = o[1];
        scales[which ? w_scale_index(r, kb, tiles) : a_scale_index(r, kb, tiles)] = (uint8_t)sb;
    }
}

void quantize_f16_to_fp8_launch(const void* src16, uint8_t* dst8, uint8_t* scales, int64_t rows, int32_t K,
                                int32_t weight_layout, hipStream_t stream) {
    ME_CHECK(K % 128 == 0 && rows > 0, ME_ERR_BAD_SHAPE, "fp8 quantise: rows=%lld K=%d", (long long)rows, K);
    if (weight_layout) ME_CHECK(rows % 64 == 0, ME_ERR_BAD_SHAPE, "fp8 quantise: %lld weight rows", (long long)rows);
    const int64_t tiles = weight_layout ? rows / 64 : cdiv(rows, 128);
    const int64_t total = rows * (K / 32);
    ProfScope prof(stream, "quantize_f16_kernel", 0.0, (double)rows * K * 3.03);
    const int64_t g = cdiv(total, 256);
    hipLaunchKernelGGL(quantize_f16_kernel, dim3((unsigned)(g > 65535 ? 65535 : g)), dim3(256), 0, stream,
                       (const f16*)src16, dst8, scales, rows, K, tiles, weight_layout);
    ME_HIP(hipGetLastError());
}

// LayerNorm (vit.rs:165,168) with the result quantised as the next GEMM's MX activation operand: one wave per
// row, lane l holds elements (i * 64 + l) * 4 .. + 3 for i < DIM / 256, so an MX block of 32 is 8 lanes of one i.
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, uint8_t* __restrict__ y8,
                                                            uint8_t* __restrict__ ys, int64_t rows, int64_t mt,
                                                            float eps, RowSegs segs) {
    constexpr int DIM = NPL * 64, NA = NPL / 4;
    static_assert(NPL % 4 == 0, "rows of at least 256 elements");
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (segs.seg1 && row >= segs.seg1) {
        const bool third = segs.seg2 && row >= segs.seg2;
        w = third ? segs.w2 : segs.w1;
        b = third ? segs.b2 : segs.b1;
    }
    const float* xr = x + row * DIM;
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
        v[i * 4 + 0] = t.x, v[i * 4 + 1] = t.y, v[i * 4 + 2] = t.z, v[i * 4 + 3] = t.w;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / DIM);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) q += (v[i] - mean) * (v[i] - mean);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / DIM) + eps);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = (i * 64 + lane) * 4;
        const float4 wv = *reinterpret_cast<const float4*>(w + e);
        const float4 bv = *reinterpret_cast<const float4*>(b + e);
        const float o0 = (v[i * 4 + 0] - mean) * rstd * wv.x + bv.x, o1 = (v[i * 4 + 1] - mean) * rstd * wv.y + bv.y;
        const float o2 = (v[i * 4 + 2] - mean) * rstd * wv.z + bv.z, o3 = (v[i * 4 + 3] - mean) * rstd * wv.w + bv.w;
        float amax = fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3)));
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        const unsigned sb = mx_scale_byte(amax);
        const float inv = mx_inv_scale(sb);
        *reinterpret_cast<unsigned*>(y8 + row * DIM + e) = pack_fp8x4(o0 * inv, o1 * inv, o2 * inv, o3 * inv);
        if ((lane & 7) == 0) ys[a_scale_index(row, e >> 5, mt)] = (uint8_t)sb;
    }
}

void layernorm_fp8_launch(const float* x, const float* w, const float* b, uint8_t* y8, uint8_t* yscale, int64_t rows,
                          int32_t dim, float eps, hipStream_t stream, const RowSegs* segs_opt) {
    const RowSegs segs = segs_opt ? *segs_opt : RowSegs();
    const int64_t mt = cdiv(rows, 128);
    const dim3 grid((unsigned)cdiv(rows, 4)), block(256);
    ProfScope prof(stream, "layernorm_fp8_kernel", 0.0, (double)rows * dim * 5);
    switch (dim / 64) {
        case 4: hipLaunchKernelGGL((layernorm_fp8_kernel<4>), grid, block, 0, stream, x, w, b, y8, yscale, rows, mt, eps, segs); break;
        case 8: hipLaunchKernelGGL((layernorm_fp8_kernel<8>), grid, block, 0, stream, x, w, b, y8, yscale, rows, mt, eps, segs); break;
        case 16: hipLaunchKernelGGL((layernorm_fp8_kernel<16>), grid, block, 0, stream, x, w, b, y8, yscale, rows, mt, eps, segs); break;
        default: fail(ME_ERR_BAD_SHAPE, "fp8 layernorm: dim %d not in {256,512,1024}", dim);
    }
    ME_HIP(hipGetLastError());
}

}  // namespace me
