// Tile-configuration choice and (dtype, A-mode, epilogue) dispatch for the GEMM core.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "gemm_core.h"

namespace me {

void fail(int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error{code, buf};
}

// explicit specialisations live in gemm_{f16,bf16}_{plain,conv}.hip
#define ME_DECL(T, A, E) \
    template <>          \
    void gemm_dispatch<T, A, E>(const GemmParams&, int, hipStream_t);
ME_DECL(f16, A_PLAIN, EPI_STORE)
ME_DECL(f16, A_PLAIN, EPI_RESID_SCALE)
ME_DECL(f16, A_PLAIN, EPI_PATCH_EMBED)
ME_DECL(f16, A_PLAIN, EPI_CONVT)
ME_DECL(f16, A_CONV, EPI_STORE)
ME_DECL(f16, A_CONV, EPI_HEAD_FINAL)
ME_DECL(bf16, A_PLAIN, EPI_STORE)
ME_DECL(bf16, A_PLAIN, EPI_RESID_SCALE)
ME_DECL(bf16, A_PLAIN, EPI_PATCH_EMBED)
ME_DECL(bf16, A_PLAIN, EPI_CONVT)
ME_DECL(bf16, A_CONV, EPI_STORE)
ME_DECL(bf16, A_CONV, EPI_HEAD_FINAL)
#undef ME_DECL

static thread_local unsigned* g_status_word = nullptr;
unsigned* current_status_word() { return g_status_word; }
void set_current_status_word(unsigned* w) { g_status_word = w; }

Profiler& profiler() {
    static thread_local Profiler p;
    return p;
}
ProfScope::ProfScope(hipStream_t s, const std::string& name, double flops, double bytes) : stream(s) {
    Profiler& p = profiler();
    if (!p.enabled) return;
    ProfEntry e;
    e.name = name, e.flops = flops, e.bytes = bytes;
    if (hipEventCreate(&e.e0) != hipSuccess || hipEventCreate(&e.e1) != hipSuccess) return;
    (void)hipEventRecord(e.e0, stream);
    index = (int)p.entries.size();
    p.entries.push_back(e);
}
ProfScope::~ProfScope() {
    if (index >= 0) (void)hipEventRecord(profiler().entries[index].e1, stream);
}

// Tile configurations (the ids are what me_op_* take as tile_cfg)
static const char* kCfgNames[] = {"256x256x64/8w-pp", "128x128x64/4w", "64x64x64/4w", "160x128x64/4w",
                                  "64x64x64/4w-ring6", "192x256x64/8w-pp", "256x256x64/8w-8ph", "96x256x64/8w-pp",
                                  "128x256x64/8w-ring3", "16x16px-x256x64/8w-halo", "352x256x64/8w-pp", "12x16px-x256x64/8w-halo", "16x16px-x128x64/8w-halo"};
enum { CFG_PP256 = 0, CFG_128 = 1, CFG_64 = 2, CFG_160 = 3, CFG_RING64 = 4, CFG_PP192 = 5, CFG_8PH = 6, CFG_PP96 = 7,
       CFG_RING128 = 8, CFG_HALO = 9, CFG_PP352 = 10, CFG_HALO12 = 11, CFG_HALO_N128 = 12, CFG_COUNT = 13 };
int gemm_num_configs() { return CFG_COUNT; }
const char* gemm_config_name(int cfg) { return cfg >= 0 && cfg < CFG_COUNT ? kCfgNames[cfg] : "?"; }

// Tile choice.  The persistent kernels run ceil(tiles / resident workgroups) rounds, so the cost of a
// configuration is rounds x (tile area) x (workgroups sharing a CU) / (its main-loop efficiency relative
// to the two-group 256x256 kernel, measured at M = 80780 where rounds do not matter: profiles/
// r01_kernel_microbench_f16.json).  Examples at M = 20195: qkv / fc1 (N = 3072 / 4096) -> 256x256 (4 and
// 5 rounds); proj / fc2 (N = 1024) -> 160x128: 1016 tiles = 2 rounds of 512, where 128x128 needs 3 and
// 256x256 leaves 3/4 of the second round idle; the 256-channel convolutions at 768^2 and 384^2 -> 256x256.
// The 64x64 tile is for the single-window ViTs (M = 577) and the low-resolution decoder levels.
static bool dynamic_tile_order() {
    static const bool on = getenv("ME_GEMM_DYNAMIC_TILES") != nullptr;
    return on;
}

static int pick_config(int64_t M, int64_t N, int64_t K, int64_t seg1, int64_t seg2, bool resid) {
    const int64_t t1 = cdiv(M, 128) * cdiv(N, 128);
    if (N < 128 || t1 < 256) {
        // few 64x64 tiles and a long K (the M = 577 fc2: 160 tiles x 64 slabs): the six-slot ring keeps
        // five slabs in flight (33 -> 19 us); with many tiles its 96 KiB of LDS per workgroup costs more
        // in occupancy than the latency it hides
        return cdiv(M, 64) * cdiv(N, 64) <= 512 && K >= 2048 ? CFG_RING64 : CFG_64;
    }
    struct Cand {
        int cfg, bm, bn, per_cu;
        double eff;
    };
    // 192x256 (residual epilogue only): at M = 21760, N = 1024 it turns 340 tiles (1.33 rounds of 256) into 432 (1.69
    // rounds).  Measured standalone: proj (K = 1024) 94.7 -> 80.3 us, fc2 (K = 4096) 229 -> 252 us -- the shorter
    // tile moves 17 % more operand bytes per MFMA through L2 -> LDS and a K = 4096 main loop pays for that, a
    // K = 1024 one is half epilogue and gains from the rounds.
    const Cand cands[] = {{CFG_PP256, 256, 256, 1, 1.0},
                          {CFG_PP192, 192, 256, 1, K <= 1024 ? 0.88 : 0.68},
                          {CFG_160, 160, 128, 2, 0.80},
                          {CFG_128, 128, 128, 2, 0.76}};
    int best = CFG_128;
    double best_cost = 0.0;
    for (const Cand& c : cands) {
        if (c.cfg == CFG_PP192 && !resid) continue;
        if (N < c.bn || ((c.cfg == CFG_PP256 || c.cfg == CFG_PP192) && K < 128)) continue;
        if (seg1 % c.bm || seg2 % c.bm) continue;  // row segments must start on tile boundaries
        const int64_t tiles = cdiv(M, c.bm) * cdiv(N, c.bn);
        // static order: whole rounds; dynamic order: workgroups draw tiles until none are left, so the
        // launch takes the average share plus about half a tile of tail
        const double rounds = dynamic_tile_order() && tiles > 256 * c.per_cu
                                  ? (double)tiles / (256.0 * c.per_cu) + 0.5
                                  : (double)cdiv(tiles, (int64_t)256 * c.per_cu);
        const double cost = rounds * c.bm * c.bn * c.per_cu / c.eff;
        if (best_cost == 0.0 || cost < best_cost) best = c.cfg, best_cost = cost;
    }
    return best;
}

template <typename T>
static void launch_typed(const GemmParams& p, AMode amode, EpiKind epi, int cfg, hipStream_t s) {
    if (amode == A_PLAIN) {
        switch (epi) {
            case EPI_STORE: gemm_dispatch<T, A_PLAIN, EPI_STORE>(p, cfg, s); return;
            case EPI_RESID_SCALE: gemm_dispatch<T, A_PLAIN, EPI_RESID_SCALE>(p, cfg, s); return;
            case EPI_PATCH_EMBED: gemm_dispatch<T, A_PLAIN, EPI_PATCH_EMBED>(p, cfg, s); return;
            case EPI_CONVT: gemm_dispatch<T, A_PLAIN, EPI_CONVT>(p, cfg, s); return;
            default: break;
        }
    } else {
        switch (epi) {
            case EPI_STORE: gemm_dispatch<T, A_CONV, EPI_STORE>(p, cfg, s); return;
            case EPI_HEAD_FINAL: gemm_dispatch<T, A_CONV, EPI_HEAD_FINAL>(p, cfg, s); return;
            default: break;
        }
    }
    fail(ME_ERR_BAD_ARG, "gemm: unsupported (A-mode %d, epilogue %d)", (int)amode, (int)epi);
}

// One tile-queue block per launch stream (kernels of a stream run one after the other; the last
// workgroup of a launch leaves the counters zeroed for the next).
static unsigned* queue_for_stream(hipStream_t stream) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, unsigned*> blocks;
    int dev = 0;
    ME_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    unsigned*& q = blocks[{dev, stream}];
    if (!q) {
        ME_HIP(hipMalloc((void**)&q, kGemmQueueWords * sizeof(unsigned)));
        ME_HIP(hipMemset(q, 0, kGemmQueueWords * sizeof(unsigned)));
    }
    return q;
}

// Compute units a stream may use.  A stream created with hipExtStreamCreateWithCUMask (or a process under a global CU
// mask) gets fewer than the device reports; a persistent grid sized from the device's count would then queue workgroups
// behind one another -- harmless for the plain kernels, a wait without partner for the fused LayerNorm launch.
int stream_cu_count(hipStream_t stream) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int> cache;
    int dev = 0;
    ME_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({dev, stream});
    if (it != cache.end()) return it->second;
    int cus = 0;
    ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int granted = cus;
    uint32_t mask[32] = {0};
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (!capturing && hipExtStreamGetCUMask(stream, 32, mask) == hipSuccess) {
        int bits = 0;
        for (uint32_t w : mask) bits += __builtin_popcount(w);
        if (bits > 0 && bits < cus) granted = bits;
    }
    (void)hipGetLastError();
    if (!capturing) cache[{dev, stream}] = granted;
    return granted;
}

void head_composed_launch_f16(const GemmParams& p, hipStream_t stream);
void head_composed_launch_bf16(const GemmParams& p, hipStream_t stream);

bool head_composed_fits(const GemmParams& p) {
    return p.N == 128 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.Cin % 64 == 0 && p.K == 9 * p.Cin && p.out_H % 16 == 0 &&
           p.out_W % 16 == 0 && p.M % 256 == 0 && p.M == (p.M / (p.out_H * p.out_W)) * p.out_H * p.out_W;
}

void head_composed_launch(const GemmParams& p_in, int32_t dtype, hipStream_t stream) {
    GemmParams p = p_in;
    p.resident_out = nullptr, p.queue = nullptr, p.status = current_status_word();
    p.cu_granted = stream_cu_count(stream);
    ME_CHECK(head_composed_fits(p), ME_ERR_BAD_SHAPE, "composed head: %dx%d map, Cin %d, N %d", p.out_H, p.out_W, p.Cin, p.N);
    ME_CHECK(p.A && p.W && p.bias && p.tap_bias && p.w2 && p.b2 && p.out32 && p.pixels_per_image == 4 * p.out_H * p.out_W,
             ME_ERR_BAD_ARG, "composed head: missing operand");
    static const bool log_launches = getenv("ME_LOG_LAUNCH") != nullptr;
    if (log_launches) fprintf(stderr, "gemm_launch conv head_composed M=%d N=%d K=%d\n", p.M, p.N, p.K);
    // algorithmic FLOPs: every output phase reads a 2 x 2 window of the 3 x 3 one (4 of 9 taps carry weights): the dense launch
    // executes 9 / 4 of them; counted here are the FLOPs of the two layers it replaces at their own shapes (SURVEY App. B:
    // ConvTranspose 2 * H * W * Cin * Cin * 4 + conv3x3 2 * 4 H W * Cin * 32 * 9)
    const double px = (double)p.M;
    ProfScope prof(stream, std::string("gemm_kernel<") + (dtype == ME_DTYPE_F16 ? "f16" : "bf16") + ",16x16px-x128x64/8w-halo,conv,head_composed>",
                   2.0 * px * p.Cin * p.Cin * 4 + 2.0 * 4 * px * p.Cin * 32 * 9,
                   px * p.Cin * 2 + (double)p.N * p.K * 2 + 4 * px * 4);
    if (dtype == ME_DTYPE_F16) head_composed_launch_f16(p, stream);
    else if (dtype == ME_DTYPE_BF16) head_composed_launch_bf16(p, stream);
    else fail(ME_ERR_BAD_ARG, "composed head: bad dtype %d", dtype);
}

int gemm_lnf_resident(int32_t dtype, hipStream_t stream) {
    GemmParams p = GemmParams();
    int32_t resident = 0;
    p.resident_out = &resident;
    p.cu_granted = stream_cu_count(stream);
    p.ln_out16 = &resident;  // (selects the LNF instantiation; nothing is launched)
    p.K = 128;
    if (dtype == ME_DTYPE_F16)
        gemm_dispatch<f16, A_PLAIN, EPI_RESID_SCALE>(p, CFG_PP352, stream);
    else if (dtype == ME_DTYPE_BF16)
        gemm_dispatch<bf16, A_PLAIN, EPI_RESID_SCALE>(p, CFG_PP352, stream);
    else
        fail(ME_ERR_BAD_ARG, "gemm: bad dtype %d", dtype);
    return resident;
}

void gemm_launch(const GemmParams& p_in, AMode amode, EpiKind epi, int32_t dtype, hipStream_t stream,
                 int32_t force_cfg) {
    GemmParams p = p_in;
    p.resident_out = nullptr;
    p.cu_granted = stream_cu_count(stream);
    {
        static const int spin = getenv("ME_LN_SPIN_LIMIT") ? atoi(getenv("ME_LN_SPIN_LIMIT")) : 0;  // test knob
        p.ln_spin_limit = spin > 0 ? spin : 0;
    }
    // Dynamic tile order (TileQueue in gemm_core.h) is an opt-in: measured on the full step it gains 0.4 %
    // (26.57 vs 26.68 ms) -- the launches the side streams disturb most (proj / fc2) have two tiles per
    // workgroup, too coarse for a late workgroup to hand work to its neighbours.
    p.queue = dynamic_tile_order() ? queue_for_stream(stream) : nullptr;
    p.status = current_status_word();
    {
        const char* gb = getenv("ME_GELU_BATCH");
        p.gelu_per_granule = gb && atoi(gb) == 0;
    }
    static const int patch_rows = getenv("ME_GEMM_PATCH_ROWS") ? atoi(getenv("ME_GEMM_PATCH_ROWS")) : 0;  // A/B knob
    p.patch_rows = patch_rows;
    ME_CHECK(p.M > 0 && p.N > 0 && p.K > 0, ME_ERR_BAD_SHAPE, "gemm: empty problem %dx%dx%d", p.M,
             p.N, p.K);
    ME_CHECK(p.K % 64 == 0, ME_ERR_BAD_SHAPE, "gemm: K=%d is not a multiple of 64", p.K);
    ME_CHECK(p.M < (1 << 30) && p.N < (1 << 30), ME_ERR_BAD_SHAPE, "gemm: %dx%d too large", p.M, p.N);
    ME_CHECK(p.N % 4 == 0, ME_ERR_BAD_SHAPE, "gemm: N=%d is not a multiple of 4", p.N);
    ME_CHECK(p.A && p.W, ME_ERR_BAD_ARG, "gemm: null operand");
    if (amode == A_CONV) {
        ME_CHECK(p.Cin % 64 == 0 && p.K == p.KH * p.KW * p.Cin, ME_ERR_BAD_SHAPE,
                 "conv: Cin=%d K=%d KHxKW=%dx%d", p.Cin, p.K, p.KH, p.KW);
        ME_CHECK(p.out_H > 0 && p.out_W > 0 && p.M % (p.out_H * p.out_W) == 0, ME_ERR_BAD_SHAPE,
                 "conv: M=%d is not a multiple of %dx%d", p.M, p.out_H, p.out_W);
    } else {
        ME_CHECK(p.lda % 8 == 0 && p.lda >= p.K, ME_ERR_BAD_SHAPE, "gemm: lda=%lld K=%d",
                 (long long)p.lda, p.K);
    }
    if (epi == EPI_HEAD_FINAL) ME_CHECK(p.N <= 32, ME_ERR_BAD_SHAPE, "head: N=%d > 32", p.N);
    if (p.tap_bias)  // the epilogue needs the output pixel's coordinates: the bordered 16-bit output has them
        ME_CHECK(amode == A_CONV && epi == EPI_STORE && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.out16 && p.out16_border && p.bias &&
                     p.N % 8 == 0 && !p.out32 && !p.res32 && !p.res32b && !p.lo_off16 && !p.hi2_off16 && p.act != ACT_GELU,
                 ME_ERR_BAD_ARG, "conv: per-tap bias shares take a 3x3 stride-1 convolution with a bordered 16-bit output and nothing else");
    if (p.ln_out16) {
        ME_CHECK(force_cfg == CFG_PP352 && amode == A_PLAIN && epi == EPI_RESID_SCALE && (p.N == 256 || p.N == 512 || p.N == 1024) &&
                     p.ldc == p.N && p.ln_w && p.ln_b && p.ln_stats && p.ln_count && p.bias && p.gamma && p.res32 && p.out32 &&
                     p.K >= 128,
                 ME_ERR_BAD_ARG, "gemm: the fused LayerNorm takes the 352-row tile's residual epilogue with N in {256, 512, 1024}");
        ME_CHECK(p.seg1 == 0 || (p.ln_w_s1 && p.ln_b_s1 && (p.seg2 == 0 || (p.ln_w_s2 && p.ln_b_s2))), ME_ERR_BAD_ARG,
                 "gemm: a row segment without LayerNorm weights");
        // out8 beside ln_out16: the normalised rows leave as MX fp8 + activation-layout scales instead of 16-bit
        if (p.out8)
            ME_CHECK(p.out8_scale && (int64_t)p.out8_mt * 128 >= p.M, ME_ERR_BAD_ARG,
                     "gemm: the fused LayerNorm's fp8 output needs its block scales (%d tiles of 128 rows for %d rows)", p.out8_mt, p.M);
    }
    if (p.qcols)
        ME_CHECK(epi == EPI_STORE && p.qcols % 64 == 0 && p.qcols <= p.N && p.out16 && !p.out32 && p.act == ACT_NONE &&
                     !p.lo_off16 && !p.hi2_off16,
                 ME_ERR_BAD_ARG, "gemm: scaled leading columns (qcols = %d) take a plain 16-bit output", p.qcols);
    int cfg = force_cfg >= 0 ? force_cfg : pick_config(p.M, p.N, p.K, p.seg1, p.seg2, amode == A_PLAIN && epi == EPI_RESID_SCALE);
    // 3x3 convolutions big enough for the 256x256 tile take its halo form (gemm_core.h conv_halo_kernel): the 18 x 18
    // halo of a 16 x 16 pixel tile staged once per 64 input channels instead of the pixels once per tap
    static const bool halo_on = !(getenv("ME_CONV_HALO") && atoi(getenv("ME_CONV_HALO")) == 0);
    const bool halo_fits = amode == A_CONV && epi == EPI_STORE && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.out_H % 16 == 0 &&
                           p.out_W % 16 == 0 && p.N % 256 == 0;
    if (force_cfg < 0 && halo_on && halo_fits && cfg == CFG_PP256) {
        cfg = CFG_HALO;
        // 12-row tiles where they fill the rounds better: a tile of 12 x 16 pixels costs about 0.78 of a 16 x 16 one
        // (three quarters of the MFMAs, the same weight staging per slab)
        static const bool halo12_on = !(getenv("ME_CONV_HALO12") && atoi(getenv("ME_CONV_HALO12")) == 0);
        if (halo12_on && p.out_H % 12 == 0) {
            const int64_t nbn = p.N / 256, t16 = (int64_t)(p.M / 256) * nbn, t12 = (int64_t)(p.M / 192) * nbn;
            if ((double)cdiv(t12, 256) * 0.78 < (double)cdiv(t16, 256) * 0.97) cfg = CFG_HALO12;
        }
    }
    if (cfg == CFG_HALO) ME_CHECK(halo_fits, ME_ERR_BAD_SHAPE, "conv: the halo tile takes 3x3 stride-1 convolutions on maps of 16-pixel multiples, N a multiple of 256");
    // N = 128 (the head's first convolution, 256 -> 128 at 768 x 768): the halo tile with 128 channels where the
    // 128x128 implicit-GEMM tile would have been chosen on a map of at least a round of pixel tiles
    const bool halo128_fits = amode == A_CONV && epi == EPI_STORE && p.KH == 3 && p.KW == 3 && p.stride == 1 &&
                              p.out_H % 16 == 0 && p.out_W % 16 == 0 && p.N % 128 == 0;
    static const bool halo128_on = !(getenv("ME_CONV_HALO128") && atoi(getenv("ME_CONV_HALO128")) == 0);
    if (force_cfg < 0 && halo_on && halo128_on && halo128_fits && p.N % 256 != 0 && cfg == CFG_128 &&
        (int64_t)(p.M / 256) * (p.N / 128) >= 256)
        cfg = CFG_HALO_N128;
    if (cfg == CFG_HALO_N128)
        ME_CHECK(halo128_fits, ME_ERR_BAD_SHAPE, "conv: the 128-channel halo tile takes 3x3 stride-1 convolutions on maps of 16-pixel multiples, N a multiple of 128");
    if (cfg == CFG_HALO12)
        ME_CHECK(amode == A_CONV && epi == EPI_STORE && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.out_H % 12 == 0 &&
                     p.out_W % 16 == 0 && p.N % 256 == 0,
                 ME_ERR_BAD_SHAPE, "conv: the 12-row halo tile takes 3x3 stride-1 convolutions on maps of 12 x 16 pixel multiples, N a multiple of 256");
    if ((cfg == CFG_PP256 || cfg == CFG_PP192 || cfg == CFG_8PH || cfg == CFG_PP96 || cfg == CFG_PP352) && p.K < 128) cfg = CFG_128;  // they prefetch two slabs ahead
    {
        // (the 352-row tile lays its row tiles out per segment, gemm_core.h seg_tile_rows: any boundary will do)
        static const int kTileRows[CFG_COUNT] = {256, 128, 64, 160, 64, 192, 256, 96, 128, 256, 1, 1, 256};
        const int bm = epi == EPI_HEAD_FINAL ? 256 : kTileRows[cfg];
        ME_CHECK(p.seg1 % bm == 0 && p.seg2 % bm == 0 && (p.seg2 == 0 || p.seg2 > p.seg1), ME_ERR_BAD_ARG,
                 "gemm: row segments %d / %d do not start on %d-row tile boundaries", p.seg1, p.seg2, bm);
    }
    static const char* kEpi[] = {"store", "resid_scale", "patch_embed", "?", "convt", "head_final"};
    // the head's final layers on a halo tile (head_conv.hip) where the shape is the model's own; force_cfg >= 0 or
    // ME_HEAD_HALO=0: the implicit-GEMM tile below
    static const bool head_halo_on = !(getenv("ME_HEAD_HALO") && atoi(getenv("ME_HEAD_HALO")) == 0);
    const bool head_halo = epi == EPI_HEAD_FINAL && amode == A_CONV && force_cfg < 0 && head_halo_on && head_final_halo_fits(p);
    static const bool log_launches = getenv("ME_LOG_LAUNCH") != nullptr;  // one line per launch: which shape a trace row is
    if (log_launches)
        fprintf(stderr, "gemm_launch %s %s M=%d N=%d K=%d k=%dx%d/s%d cfg=%s res32=%d out32=%d out16=%d border=%d lo=%d hi2=%d ln=%d\n",
                amode == A_PLAIN ? "plain" : "conv", kEpi[epi], p.M, p.N, p.K, p.KH, p.KW, p.stride, gemm_config_name(cfg),
                p.res32 != nullptr, p.out32 != nullptr, p.out16 != nullptr, p.out16_border, (int)p.lo_off16, (int)p.hi2_off16,
                p.ln_out16 != nullptr);
    // algorithmic bytes of the launch (every operand once): what an HBM-bound launch is priced against in bench.py's
    // kernels[] (the MFMA fraction misstates the residual launches and the ConvTransposes)
    double launch_bytes = 0.0;
    {
        const double MN = (double)p.M * p.N;
        launch_bytes += amode == A_PLAIN ? (double)p.M * p.K * 2                                              // activation rows
                                         : (double)p.M * (p.stride > 0 ? p.stride * p.stride : 1) * p.Cin * 2;  // input pixels, once
        launch_bytes += (double)p.N * p.K * 2;
        if (epi == EPI_HEAD_FINAL) launch_bytes += (double)p.M * 4;
        else {
            if (p.out16) launch_bytes += MN * 2 * (1 + (p.lo_off16 ? 1 : 0) + (p.hi2_off16 ? 1 : 0));
            if (p.out32) launch_bytes += MN * 4;
            if (p.res32) launch_bytes += MN * 4;
            if (p.res32b) launch_bytes += MN * 4;
            if (p.ln_out16) launch_bytes += p.out8 ? MN : MN * 2;
            else if (p.out8) launch_bytes += MN;
        }
    }
    ProfScope prof(stream,
                   std::string("gemm_kernel<") + (dtype == ME_DTYPE_F16 ? "f16" : "bf16") + "," +
                       (head_halo ? "12x16px-x32/8w-halo" : (epi == EPI_HEAD_FINAL ? "256x32x64/4w" : gemm_config_name(cfg))) + "," +
                       (amode == A_PLAIN ? "plain" : "conv") + "," + kEpi[epi] + ">",
                   2.0 * (p.flop_rows ? p.flop_rows : p.M) * p.N * (p.flop_k ? p.flop_k : p.K), launch_bytes);
    if (head_halo) {
        head_final_halo_launch(p, dtype, stream);
        return;
    }
    if (dtype == ME_DTYPE_F16)
        launch_typed<f16>(p, amode, epi, cfg, stream);
    else if (dtype == ME_DTYPE_BF16)
        launch_typed<bf16>(p, amode, epi, cfg, stream);
    else
        fail(ME_ERR_BAD_ARG, "gemm: bad dtype %d", dtype);
}

}  // namespace me
