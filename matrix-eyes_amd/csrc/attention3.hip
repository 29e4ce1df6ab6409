// Fused multi-head attention of the forward pass (reference src/depth_pro/vit.rs:58-75), round 5's re-cut: the decomposition of
// attention.hip's attention2_kernel -- S^T = K Qc^T with the accumulator as the B operand of O^T += V^T P^T, Q pre-scaled by the
// qkv linear's epilogue, the softmax reference point inside the MFMA accumulator, deferred maximum, row sums by MFMA, K/V tiles
// of 64 keys by LDS-DMA into a two-slot ring -- on v_mfma_f32_16x16x32 instead of 32x32x16, so that a wave's queries come in
// blocks of 16:
//   * a wave owns 48 queries (three 16-query blocks), a workgroup of four waves 192: 577 = 3 x 192 + 1, three workgroups per
//     (window, head) with no idle wave and no padded block (five workgroups of 128 left one wave in twenty idle and one with a
//     single query), and the 577th query goes to attention_extra_query on trailing workgroups.  (Six waves of 32 queries, the
//     other way to 192, lose by half: profiles/r05_attention_six_wave_workgroups.txt -- workgroups want a multiple of four waves.)
//   * every K and V fragment read from LDS feeds THREE MFMAs (one per query block) instead of one: a third less LDS traffic per
//     query, a K/V tile staged once per 192 queries instead of 128, one prologue and one store stage per 48 queries of a wave.
//   * the three query blocks are three independent dependency chains in one wave's instruction stream.
// Fragment layouts (v_mfma_f32_16x16x32: A 16 x 32, lane l holds row l & 15, k = 8 (l >> 4) + 0..7; B likewise with its column;
// C / D: lane l holds column l & 15, rows 4 (l >> 4) + 0..3).  c16 = lane & 15, g = lane >> 4:
//   S^T block (16 keys kb x 16 queries qb) = K[16 kb + c16][32 s + 8 g ..] . Qc[q0 + 16 qb + c16][32 s + 8 g ..], s = 0, 1;
//     the lane ends up with the scores of ITS query against keys 16 kb + 4 g + 0..3;
//   P^T as the B operand of a 32-key step t: the lane's eight probabilities of key blocks 2t and 2t + 1, i.e. contraction slot
//     8 g + j <-> key 32 t + 4 g + j (j < 4), 32 t + 16 + 4 g + (j - 4) (j >= 4);
//   V^T as the A operand in the same slot order: two transposed reads (ds_read_b64_tr_b16) of the 4-row x 16-column blocks at
//     rows 32 t + 4 g and 32 t + 16 + 4 g of channel block db -- lane c16 receives column 16 db + c16 of each.
// LDS images: K rows of 128 bytes, 16-byte chunk c at slot c ^ ((row >> 1) & 7) (a fragment read's sixteen rows spread over all
// eight slots); V rows with their 32-byte columns at column ^ (row & 3) (a transposed read's four rows touch every bank once).
#include "attention_shared.h"

namespace me {

namespace {

template <typename T>
struct Mfma16;
template <>
struct Mfma16<f16> {
    static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <>
struct Mfma16<bf16> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

#ifdef ME_ATT_STAMPS
// Diagnostic build only (tools/attn3_stamps.py): per-wave phase accounting in shader clocks.  Reading the clock waits for all
// LDS operations in flight, so the phases are serialised a little more than in the product kernel.  [0] DMA wait + barrier,
// [1] staging issue, [2] S' = K Qc^T, [3] max + branch, [4] exponentials, [5] P V + row sums + closing LDS wait, [6] everything,
// [7] prologue (up to the first tile)
__device__ unsigned long long* g_att3_stamps = nullptr;
#define ATT_PH(i)                                                      \
    do {                                                               \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[i] += t_now - t_last;                                       \
        t_last = t_now;                                                \
    } while (0)
#else
#define ATT_PH(i)
#endif

constexpr int KT = 64;            // keys per LDS tile
constexpr int TILE_BYTES = KT * 128;
constexpr int QW = 48;            // queries per wave: three blocks of 16

// One main item = 192 queries (four waves x 48) of one (window, head).  Items are numbered i = 8 * slot0 + xcd with (window, head)
// group = xcd + 8 * (slot0 / nqb) and query block slot0 % nqb: all query blocks of a group keep one XCD (its K and V cross the
// fabric once), and a persistent workgroup b -- always on XCD b % 8 -- takes items b, b + G, b + 2 G, ... (G = the grid, a multiple
// of 8).  Then the extra items (the one query beyond a group's whole wave units): n_main + e <-> group (e & 7) + 8 (e >> 3).
struct AttItem {
    int id;        // -1: none
    int q0;        // first query of the workgroup's 192
    int head;
    int64_t row0;  // first token row of the window
};

template <typename T>
__global__ __launch_bounds__(256, 2) void attention3_kernel(const T* __restrict__ qkv, T* __restrict__ out, int tokens, int heads,
                                                            int ngroups, RowSegs segs, uint8_t* __restrict__ out8,
                                                            uint8_t* __restrict__ out8_scale, int64_t out8_mt, float defer_thr,
                                                            int nqb, int nunits, int n_main, int n_extra) {
    typedef typename Frag16<T>::frag frag;
    constexpr int NSLOT = 2;
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * 2 * TILE_BYTES];  // slot: K tile, V tile
    __shared__ __attribute__((aligned(16))) char qsm[4 * QW * 128];             // the NEXT item's Q rows, 6 KiB per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int bid = blockIdx.x, G = gridDim.x;
#ifdef ME_ATT_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_last = t_begin;
    int items_done = 0;
#endif
    const int C = heads * 64;
    const int ldq = 3 * C;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    auto window_row0 = [&](int win) {
        int64_t r = (int64_t)win * tokens;
        if (segs.seg1 && win >= segs.win0)
            r = win < segs.win0 + segs.win1 ? segs.seg1 + (int64_t)(win - segs.win0) * tokens
                                            : segs.seg2 + (int64_t)(win - segs.win0 - segs.win1) * tokens;
        return r;
    };
    // the workgroup's next main item at or after `i` (groups beyond ngroups pad the last XCD round: skipped)
    auto next_item = [&](int i) {
        AttItem it;
        it.id = -1, it.q0 = 0, it.head = 0, it.row0 = 0;
        for (; i < n_main; i += G) {
            const int slot0 = i >> 3;
            const int group = (i & 7) + 8 * (slot0 / nqb);
            if (group >= ngroups) continue;
            const int win = group / heads;
            it.id = i, it.q0 = (slot0 - (slot0 / nqb) * nqb) * (4 * QW), it.head = group - win * heads, it.row0 = window_row0(win);
            break;
        }
        return it;
    };

    // K/V staging by LDS-DMA: a wave-instruction moves 8 rows x 128 B (lane -> row lane / 8, 16-byte chunk lane % 8); the LDS
    // image is linear in the lane, so the swizzle goes on the SOURCE chunk.  Wave w stages pieces 2w and 2w + 1 of K and of V.
    const int st_row = lane >> 3, st_slot = lane & 7;
    const unsigned row_bytes = (unsigned)ldq * 2u;
    unsigned koff0, voff0;  // the FIRST piece's; the second, eight rows on: K's slot changes by 4 (byte offset ^ 64), V's does not
    {
        const int rr = (2 * wave) * 8 + st_row;
        koff0 = (unsigned)rr * row_bytes + ((st_slot ^ ((rr >> 1) & 7)) << 4);
        voff0 = (unsigned)rr * row_bytes + ((st_slot ^ ((rr & 3) << 1)) << 4);
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    const int nkt = (tokens + KT - 1) / KT;
    // tile kt of item `it` into ring slot `sl`: 4 LDS-DMA instructions per wave
    auto stage = [&](const AttItem& it, int kt, int sl) {
        const unsigned dst = smem_base + sl * (2 * TILE_BYTES) + (2 * wave) * 1024;
        const char* kt_k = uniform_ptr((const char*)(qkv + (it.row0 + (int64_t)kt * KT) * ldq + C + it.head * 64));
        const char* kt_v = uniform_ptr(kt_k + (size_t)C * 2);
        if ((kt + 1) * KT <= tokens) {
            unsigned k0 = koff0, v0 = voff0;
            asm volatile("" : "+v"(k0), "+v"(v0));  // (the derived offsets stay temporaries: hoisted, they spill)
            glds16_raw(kt_k, k0, dst);
            glds16_raw(kt_v, v0, dst + TILE_BYTES);
            glds16_raw(kt_k, (k0 ^ 64u) + 8u * row_bytes, dst + 1024);
            glds16_raw(kt_v, v0 + 8u * row_bytes, dst + TILE_BYTES + 1024);
        } else {  // the ragged last tile: rows past the end read the last row (masked or unused below)
            const int last = tokens - 1 - kt * KT;  // >= 0
            int sr = st_row, ss = st_slot;
            asm volatile("" : "+v"(sr), "+v"(ss));  // (derived HERE: hoisted out of the item loop the four offsets are spilled, and
                                                    // every reload's vmcnt wait also waits for the LDS-DMA issued before it)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rr = (2 * wave + i) * 8 + sr;
                const unsigned rowb = (unsigned)(rr < last ? rr : last) * row_bytes;
                glds16_raw(kt_k, rowb + ((ss ^ ((rr >> 1) & 7)) << 4), dst + i * 1024);
                glds16_raw(kt_v, rowb + ((ss ^ ((rr & 3) << 1)) << 4), dst + TILE_BYTES + i * 1024);
            }
        }
    };

    // Fragment read addresses as two per-lane constants beside immediates and XORs with immediates:
    //   K: row 16 kb + c16, chunk 4 s + g at slot (4 s + g) ^ ((c16 >> 1) & 7): k_lane ^ (s << 6) + kb * 2048
    //   V (transposed read): row 32 t + 16 half + 4 g + (c16 >> 2), byte column (32 db + 8 (c16 & 3)) ^ ((c16 >> 2) << 5):
    //      (v_lane ^ (db << 5)) + (32 t + 16 half) * 128
    int k_lane, v_lane;
    {
        const int k_swz = (c16 >> 1) & 7;
        k_lane = c16 * 128 + ((g ^ k_swz) << 4);
        v_lane = (4 * g + (c16 >> 2)) * 128 + ((8 * (c16 & 3)) ^ ((c16 >> 2) << 5));
    }
    frag ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
    const bool tail_key = (tokens % KT) == 1 && nkt >= 2;
    const int nfull = tail_key ? nkt - 1 : nkt;  // tiles that run on the matrix pipe
    auto round16 = [](float x) -> T {
        asm volatile("" : "+v"(x));
        return (T)x;
    };

    // ---- the main items: ONE stream of K/V tiles across them.  Consuming a tile = wait for its DMA, meet, request the stream's
    // next tile (this item's, or the first of the workgroup's next item) into the slot the barrier has just freed, compute.  An
    // item's first-tile latency and its store stage thereby run under DMA that is already in flight, and its Q rows have been in
    // LDS for a whole item.
    AttItem cur = next_item(bid);
    int slot = 0;
    if (cur.id >= 0) stage(cur, 0, 0);
    // Q fragments: B operand, lane holds Qc[q0 + 16 qb + c16][32 s + 8 g + 0..7], Qc = Q * scale * log2(e).  An item's Q rows are
    // brought into LDS by LDS-DMA during the first tile of the item BEFORE it (a load from memory where the fragments are
    // used would put a memory latency into every item switch), in K's image: 128-byte rows, 16-byte chunk c at slot
    // c ^ ((row >> 1) & 7) -- the fragment read is K's (k_lane).  A wave stages and reads only its own 48 rows.
    const unsigned q_base = __builtin_amdgcn_readfirstlane(lds_address(qsm)) + wave * (QW * 128);
    auto stage_q = [&](const AttItem& it) {  // 6 LDS-DMA instructions per wave
        const char* qwin = uniform_ptr((const char*)(qkv + it.row0 * ldq + it.head * 64));
        int sr = st_row, ss = st_slot;
        asm volatile("" : "+v"(sr), "+v"(ss));
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int r = 8 * pc + sr;
            int q = it.q0 + wave * QW + r;
            q = q < tokens ? q : tokens - 1;
            glds16_raw(qwin, (unsigned)q * row_bytes + ((ss ^ ((r >> 1) & 7)) << 4), q_base + pc * 1024);
        }
    };
    frag qf[3][2];
    if (cur.id >= 0) stage_q(cur);
    int stores_behind_dma = 0;  // the previous item's row-store instructions: the youngest memory operations in flight at a switch
    while (cur.id >= 0) {
        const AttItem nxt = next_item(cur.id + G);
        const int unit = cur.q0 / QW + wave;
        const int q0 = cur.q0 + wave * QW;
        const bool active = unit < nunits;
        const int head = cur.head;
        const int64_t row0 = cur.row0;
        f32x4 o[4][3], lsum[3], negm[3];
        float m_run[3];
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) {
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            lsum[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run[qb] = 0.f;  // the reference point (exp2 units); negm == -m_run in all four registers
        }
        // the stream's tile after tile kt of this item, into the slot that is not being consumed
        auto stage_next = [&](int kt) {
            if (kt + 1 < nkt) stage(cur, kt + 1, slot ^ 1);
            else if (nxt.id >= 0) stage(nxt, 0, slot ^ 1);
        };
        auto tile = [&](int kt, auto first_tag, auto tail_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr bool TAIL = decltype(tail_tag)::value;
#ifdef ME_ATT_STAMPS
            if (FIRST) ph[7] += __builtin_amdgcn_s_memtime() - t_last;
            t_last = __builtin_amdgcn_s_memtime();
#endif
            // vmcnt counts stores too: behind an item that wrote its 16-bit rows (six row stores per lane, issued AFTER this
            // tile's DMA) only that DMA is waited for -- the stores' acknowledgements would put a memory round trip into every
            // item switch.  (The fp8 store stage issues a data-dependent number of stores: it is waited for.)
            if (FIRST) {
                switch (stores_behind_dma) {  // uniform
                    case 6: wait_vmcnt<6>(); break;
                    case 5: wait_vmcnt<5>(); break;
                    case 4: wait_vmcnt<4>(); break;
                    case 3: wait_vmcnt<3>(); break;
                    case 2: wait_vmcnt<2>(); break;
                    case 1: wait_vmcnt<1>(); break;
                    default: wait_vmcnt<0>(); break;
                }
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ATT_PH(0);
            if (FIRST) {
                // this item's Q fragments (their DMA was issued a whole item ago, or in front of the loop), then the next item's rows
                const unsigned qa = q_base + (unsigned)k_lane;
#pragma unroll
                for (int qb = 0; qb < 3; ++qb)
#pragma unroll
                    for (int st = 0; st < 2; ++st) qf[qb][st] = lds_read_frag<frag>((qa ^ (unsigned)(st << 6)) + qb * 2048);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the next item's rows overwrite them
                if (nxt.id >= 0) stage_q(nxt);
            }
            stage_next(kt);
            ATT_PH(1);
            // this tile's fragment addresses (LDS byte addresses; the slots are 16 KiB-aligned, so the XORs below stay inside)
            unsigned ka = smem_base + slot * (2 * TILE_BYTES) + (unsigned)k_lane;
            unsigned va = smem_base + slot * (2 * TILE_BYTES) + TILE_BYTES + (unsigned)v_lane;
            asm volatile("" : "+v"(ka), "+v"(va));  // (derived per tile: hoisted, they would live through the loop)
            slot ^= 1;
            if (active) {
                // A tile's two 32-key steps one after the other through ONE score block of 2 key blocks x 3 query blocks (24
                // registers instead of 48: with the whole tile's scores side by side the kernel needs 254 registers and two waves
                // per SIMD; the phase clocks show a wave alone on its SIMD only 15 % faster than one of a pair -- the kernel is
                // bound by each wave's own dependency chain, so a third wave per SIMD is worth more than the longer chain costs).
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bool first = FIRST && t == 0;
                    // ---- S' = K Qc^T - m' : each K fragment read feeds three MFMAs
                    f32x4 s[2][3];
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            const frag kf = lds_read_frag<frag>((ka ^ (unsigned)(st << 6)) + (2 * t + k2) * 2048);
#pragma unroll
                            for (int qb = 0; qb < 3; ++qb) s[k2][qb] = Mfma16<T>::run(kf, qf[qb][st], st == 0 ? negm[qb] : s[k2][qb]);
                        }
#ifdef ME_ATT_STAMPS
                    asm volatile("" : "+v"(s[0][0]), "+v"(s[1][2]));
#endif
                    ATT_PH(2);
                    // ---- does a reference point have to follow its maximum?
                    float mloc[3];
#pragma unroll
                    for (int qb = 0; qb < 3; ++qb) {
                        mloc[qb] = -INFINITY;
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (TAIL) {
                                    const int key = kt * KT + 16 * (2 * t + k2) + 4 * g + j;
                                    if (key >= tokens) s[k2][qb][j] = -INFINITY;
                                }
                                mloc[qb] = fmaxf(mloc[qb], s[k2][qb][j]);
                            }
                    }
                    // Deferred maximum (cdna_hip_programming.md T13): the reference point follows only when some score passes it by
                    // more than defer_thr (exp2 units); until then p = exp2(S') may reach 2^defer_thr instead of 1 -- the same
                    // relative precision in the 16-bit P operand, and O and l carry the same factor.  The first step always sets it.
                    if (first || __any(fmaxf(fmaxf(mloc[0], mloc[1]), mloc[2]) > defer_thr)) {
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb) {
                            // the query's maximum over the step: its scores sit in the four lanes c16, c16 + 16, c16 + 32, c16 + 48
                            float m = mloc[qb];
                            m = fmaxf(m, __shfl_xor(m, 16));
                            m = fmaxf(m, __shfl_xor(m, 32));
                            // (a ragged last tile's second step may hold no valid key at all: -inf stays out of the reference point)
                            const float delta = first ? m : fmaxf(m, 0.f);
                            if (!first) {
                                const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                                for (int db = 0; db < 4; ++db)
#pragma unroll
                                    for (int j = 0; j < 4; ++j) o[db][qb][j] *= alpha;
#pragma unroll
                                for (int j = 0; j < 4; ++j) lsum[qb][j] *= alpha;
                            }
                            m_run[qb] += delta;
#pragma unroll
                            for (int j = 0; j < 4; ++j) negm[qb][j] = -m_run[qb];
#pragma unroll
                            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                                for (int j = 0; j < 4; ++j) s[k2][qb][j] -= delta;
                        }
                    }
                    ATT_PH(3);
                    // ---- p = exp2(S')  (raw v_exp_f32: results below 2^-126 may flush to 0)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb)
#pragma unroll
                            for (int j = 0; j < 4; ++j) s[k2][qb][j] = __builtin_amdgcn_exp2f(s[k2][qb][j]);
#ifdef ME_ATT_STAMPS
                    asm volatile("" : "+v"(s[0][0]), "+v"(s[1][2]));
#endif
                    ATT_PH(4);
                    // ---- O^T += V^T P^T,  l += 1^T P^T : every V fragment feeds three MFMAs
                    frag pf[3];
#pragma unroll
                    for (int qb = 0; qb < 3; ++qb) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) pf[qb][j] = (T)s[0][qb][j], pf[qb][4 + j] = (T)s[1][qb][j];
                        lsum[qb] = Mfma16<T>::run(ones, pf[qb], lsum[qb]);
                    }
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        const unsigned vp = (va ^ (unsigned)(db << 5)) + (32 * t) * 128;
                        const s16x4 half0 = lds_read_tr16_at(vp), half1 = lds_read_tr16_at(vp + 16 * 128);
                        typedef short s16x8 __attribute__((__vector_size__(16)));
                        const s16x8 both = __builtin_shufflevector(half0, half1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const frag vf = __builtin_bit_cast(frag, both);
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb) o[db][qb] = Mfma16<T>::run(vf, pf[qb], o[db][qb]);
                    }
#ifdef ME_ATT_STAMPS
                    asm volatile("" : "+v"(o[0][0]), "+v"(o[3][2]), "+v"(lsum[0]));
#endif
                    ATT_PH(5);
                }
            }  // active
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's LDS reads are done before the barrier that lets
                                                                 // its slot be restaged
        };
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
        float l_tot[3];
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) l_tot[qb] = lsum[qb][0];
        if (tail_key) {
            // 577 = 9 x 64 + 1: the single key of the last tile (row 0 of the tile staged last) as a rank-one update
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage_next(nkt - 1);  // (the next item's first tile: this one is the item's last)
            const char* kb0 = smem + slot * (2 * TILE_BYTES);
            slot ^= 1;
            if (active) {
                const char* vb0 = kb0 + TILE_BYTES;
                frag kf[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) kf[st] = *reinterpret_cast<const frag*>(kb0 + ((4 * st + g) << 4));  // row 0: no swizzle
                typedef T v4 __attribute__((ext_vector_type(4)));
                v4 vv[4];
#pragma unroll
                for (int db = 0; db < 4; ++db) vv[db] = *reinterpret_cast<const v4*>(vb0 + 32 * db + 8 * g);
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) {
                    float dot = 0.f;
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int j = 0; j < 8; ++j) dot = __builtin_fmaf((float)kf[st][j], (float)qf[qb][st][j], dot);
                    dot += __shfl_xor(dot, 16);
                    dot += __shfl_xor(dot, 32);
                    const float rel = dot - m_run[qb];
                    const float delta = fmaxf(rel, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    const float p16 = (float)(T)__builtin_amdgcn_exp2f(rel - delta);  // through the operand type like every other key's
                    l_tot[qb] = l_tot[qb] * alpha + p16;
#pragma unroll
                    for (int db = 0; db < 4; ++db)
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[db][qb][j] = __builtin_fmaf(p16, (float)vv[db][j], o[db][qb][j] * alpha);
                }
            }
        }

        // ---- normalise and store: the lane holds query q0 + 16 qb + c16, channels 16 db + 4 g + 0..3
        if (out8) {
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) {
                const float inv = 1.0f / l_tot[qb];
                const int q = q0 + 16 * qb + c16;
                const bool ok = active && q < tokens;
                const int64_t m = row0 + q;
                // MX fp8: a 32-channel block = channel blocks 2 dp and 2 dp + 1 of the query's four lanes
#pragma unroll
                for (int dp = 0; dp < 2; ++dp) {
                    float v[8], amax = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = (float)round16(o[2 * dp + (e >> 2)][qb][e & 3] * inv);
                        amax = fmaxf(amax, fabsf(v[e]));
                    }
                    amax = fmaxf(amax, __shfl_xor(amax, 16));
                    amax = fmaxf(amax, __shfl_xor(amax, 32));
                    const unsigned sb = mx_scale_byte(amax);
                    const float sc = mx_inv_scale(sb);
                    if (ok) {
                        uint8_t* op = out8 + m * C + head * 64 + 32 * dp + 4 * g;
                        *reinterpret_cast<unsigned*>(op) = pack_fp8x4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
                        *reinterpret_cast<unsigned*>(op + 16) = pack_fp8x4(v[4] * sc, v[5] * sc, v[6] * sc, v[7] * sc);
                        if (g == 0) out8_scale[a_scale_index(m, head * 2 + dp, out8_mt)] = (uint8_t)sb;
                    }
                }
            }
        } else {
            // 16-bit output through LDS: a lane's four channels of a (query, channel block) are 8 bytes of a 128-byte row -- stored
            // from here, an instruction touches sixteen rows with 32-byte pieces.  The wave lays one query block (16 rows) at a
            // time out as rows in its own 2 KiB of the ring slot whose tile was consumed LAST (the other slot is receiving the
            // next item's first tile), and reads them back eight lanes per row, 16 bytes each.  Not before every wave has read
            // that tile (or the tail key's row): one more meeting per item.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            typedef T v4 __attribute__((ext_vector_type(4)));
            char* mine = smem + (slot ^ 1) * (2 * TILE_BYTES) + wave * 2048;
            const int rsub = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) {
                const float inv = 1.0f / l_tot[qb];
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    v4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = round16(o[db][qb][j] * inv);
                    // row c16, 8-byte piece 4 db + g; the 16-byte chunk index is XORed with the row's low bits so that the
                    // sixteen rows of a write instruction spread over the banks
                    const int piece = 4 * db + g;
                    *reinterpret_cast<v4*>(mine + c16 * 128 + ((((piece >> 1) ^ (c16 & 7)) << 4) | ((piece & 1) << 3))) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 8 * i + rsub;
                    const int q = q0 + 16 * qb + row;
                    const frag v = *reinterpret_cast<const frag*>(mine + row * 128 + ((ch ^ (row & 7)) << 4));
                    if (active && q < tokens) *reinterpret_cast<frag*>(out + (row0 + q) * C + head * 64 + 8 * ch) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (read back before the next query block overwrites the rows)
            }
        }
        // the store instructions this wave has just issued: (query block, row half) pairs with a row below `tokens` (an
        // instruction whose lanes are all masked is branched around, or counts for nothing)
        stores_behind_dma = 0;
        if (!out8 && active) {
#pragma unroll
            for (int qb = 0; qb < 3; ++qb)
#pragma unroll
                for (int i = 0; i < 2; ++i) stores_behind_dma += (q0 + 16 * qb + 8 * i) < tokens ? 1 : 0;
        }
#ifdef ME_ATT_STAMPS
        ++items_done;
#endif
        cur = nxt;
    }

    // ---- the extra items: the one query beyond the whole wave units of a (window, head), its keys shared by the four waves
    if (n_extra > 0) {
        wait_vmcnt<0>();
        __syncthreads();  // nothing of the ring is in flight or being read any more: it is the four waves' scratch from here
        for (int e = bid; e < n_extra; e += G) {
            const int group = (e & 7) + 8 * (e >> 3);
            if (group >= ngroups) continue;  // uniform
            const int win = group / heads, head = group - win * heads;
            const int64_t row0 = window_row0(win);
            const int64_t m = row0 + tokens - 1;
            attention_extra_query<T, 4>(qkv + m * ldq + head * 64, uniform_ptr((const char*)(qkv + row0 * ldq + C + head * 64)),
                                        uniform_ptr((const char*)(qkv + row0 * ldq + 2 * C + head * 64)), (unsigned)ldq * 2u, tokens,
                                        lane, wave, reinterpret_cast<float*>(smem), out ? out + m * C + head * 64 : nullptr,
                                        out8 ? out8 + m * C + head * 64 : nullptr, out8_scale, m, head, out8_mt);
            __syncthreads();  // wave 0 has read the partials before the next item's overwrite them
        }
    }
#ifdef ME_ATT_STAMPS
    if (g_att3_stamps && lane == 0 && bid < 4096) {
        ph[6] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int i = 0; i < 7; ++i) g_att3_stamps[((size_t)bid * 4 + wave) * 8 + i] = ph[i];
        g_att3_stamps[((size_t)bid * 4 + wave) * 8 + 7] = items_done ? (ph[7] ? ph[7] : 1) : 0;
    }
#endif
}

}  // namespace

#ifdef ME_ATT_STAMPS
extern "C" int32_t me_debug_set_att3_stamps(void* dev_ptr) {
    unsigned long long* p = (unsigned long long*)dev_ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_att3_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

// the forward pass's attention (Q pre-scaled); attention.hip's attention_launch dispatches here
void attention3_launch(const void* qkv, void* out, int32_t windows, int32_t tokens, int32_t heads, int32_t dtype, hipStream_t stream,
                       const RowSegs& segs, uint8_t* out8, uint8_t* out8_scale, int64_t out8_mt, float defer_thr) {
    const int ngroups = windows * heads;
    // wave units of 48 queries; the one query beyond whole units (577 = 12 x 48 + 1) is an extra item on the vector path
    const bool extra = tokens > QW && (tokens - 1) % QW == 0;
    const int nunits = extra ? (tokens - 1) / QW : (tokens + QW - 1) / QW;
    const int nqb = (nunits + 3) / 4;
    const int n_main = 8 * nqb * ((ngroups + 7) / 8);
    const int n_extra = extra ? 8 * ((ngroups + 7) / 8) : 0;
    // persistent grid: the workgroups that are resident at once (a multiple of 8: workgroup b stays on XCD b % 8), no more than
    // there are items.  ME_ATT_GRID (development): another grid, e.g. n_main = one workgroup per item.
    static PerDeviceOnce once_f16, once_bf16;
    auto resident = [&](PerDeviceOnce& once, const void* kern) {
        return per_device_once(once, [&](int dev) {
            int per_cu = 0, cus = 0;
            ME_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0));
            ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            int r = (per_cu < 1 ? 1 : per_cu) * cus;
            r -= r % 8;
            return r < 8 ? 8 : r;
        });
    };
    int grid_n = dtype == ME_DTYPE_F16 ? resident(once_f16, (const void*)attention3_kernel<f16>)
                                       : resident(once_bf16, (const void*)attention3_kernel<bf16>);
    if (const char* gv = getenv("ME_ATT_GRID")) {
        const int v = atoi(gv);
        if (v >= 8) grid_n = v - v % 8;
    }
    if (grid_n > n_main) grid_n = n_main;
    const dim3 grid(grid_n);
    if (dtype == ME_DTYPE_F16)
        hipLaunchKernelGGL((attention3_kernel<f16>), grid, dim3(256), 0, stream, (const f16*)qkv, (f16*)out, tokens, heads, ngroups, segs,
                           out8, out8_scale, out8_mt, defer_thr, nqb, nunits, n_main, n_extra);
    else if (dtype == ME_DTYPE_BF16)
        hipLaunchKernelGGL((attention3_kernel<bf16>), grid, dim3(256), 0, stream, (const bf16*)qkv, (bf16*)out, tokens, heads, ngroups,
                           segs, out8, out8_scale, out8_mt, defer_thr, nqb, nunits, n_main, n_extra);
    else
        fail(ME_ERR_BAD_ARG, "attention: bad dtype %d", dtype);
    ME_HIP(hipGetLastError());
}

}  // namespace me
