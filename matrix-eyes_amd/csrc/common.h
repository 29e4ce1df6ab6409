// Internal declarations shared by the HIP translation units of libmatrixeyes_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/matrix_eyes_hip.h"

namespace me {

typedef _Float16 f16;
typedef __bf16 bf16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((__vector_size__(8)));

struct Status {
    int32_t code = ME_OK;
    std::string msg;
};

// Thrown inside the library, caught at the C ABI; nothing escapes to the caller.
struct Error {
    int32_t code;
    std::string msg;
};

[[noreturn]] void fail(int32_t code, const char* fmt, ...);

#define ME_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            ::me::fail(e__ == hipErrorOutOfMemory ? ME_ERR_OOM : ME_ERR_HIP, "%s: %s (%s:%d)", \
                       #expr, hipGetErrorString(e__), __FILE__, __LINE__);                   \
    } while (0)

#define ME_CHECK(cond, code, ...)                   \
    do {                                            \
        if (!(cond)) ::me::fail((code), __VA_ARGS__); \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Per-kernel launch configuration that must be set up once PER DEVICE (dynamic-LDS attribute, resident
// workgroup count): contexts may live on several devices, and on several host threads, of one process.
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
    std::atomic<int> value[kMaxDevices];
    PerDeviceOnce() {
        for (auto& v : value) v.store(0, std::memory_order_relaxed);
    }
};
// value cached for the current device (0 = not configured yet); `configure` returns the value to cache (> 0).
// Two threads racing on the same device both configure (idempotent HIP calls) and store the same value.
template <typename F>
static inline int per_device_once(PerDeviceOnce& once, F&& configure) {
    int dev = 0;
    ME_HIP(hipGetDevice(&dev));
    ME_CHECK(dev >= 0 && dev < kMaxDevices, ME_ERR_BAD_ARG, "device ordinal %d out of range", dev);
    int v = once.value[dev].load(std::memory_order_acquire);
    if (!v) {
        v = configure(dev);
        once.value[dev].store(v, std::memory_order_release);
    }
    return v;
}

// ---------------------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; when on, every instrumented launch is bracketed by two events.
// ---------------------------------------------------------------------------------------
struct ProfEntry {
    std::string name;
    double flops, bytes;
    hipEvent_t e0, e1;
};
struct Profiler {
    bool enabled = false;
    std::vector<ProfEntry> entries;
};
Profiler& profiler();
struct ProfScope {
    hipStream_t stream;
    int index = -1;
    ProfScope(hipStream_t s, const std::string& name, double flops, double bytes);
    ~ProfScope();
};

// ---------------------------------------------------------------------------------------
// 16-bit operand overflow guard.  An f16 operand holds magnitudes up to 65504; past it the stored operand is
// +-inf, and on a conv stage infinities of both signs sum to NaN which the ReLU behind maps to 0 -- the branch
// drops out silently, where the fp32 reference (decoder.rs:35-44) has no such limit.  Every kernel that rounds f32
// values to f16 operands keeps the largest magnitude it stored and raises bit ME_STATUS_OVERFLOW_16BIT of the
// calling context's status word when it exceeds 65504 (one atomicOr on the rare path).
// The word of the context whose entry point is running on this host thread (set by ME_API_BEGIN).
// ---------------------------------------------------------------------------------------
unsigned* current_status_word();
void set_current_status_word(unsigned* w);
#if defined(__HIPCC__)
template <typename T>
__device__ __forceinline__ void raise_overflow16(unsigned* status, float amax) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16)) {
        if (status && amax > 65504.0f) atomicOr(status, 1u);
    }
}
#endif

// ---------------------------------------------------------------------------------------
// Device helpers shared by the GEMM and attention kernels.
// ---------------------------------------------------------------------------------------
// 16-byte LDS-DMA: lane l's 16 bytes at gsrc land at lds_base + 16 l (M0 carries the wave-uniform base).
// LDS-DMA issued behind the compiler's back (two-group and ring GEMM kernels, attention).  The waitcnt pass orders every
// later LDS access behind a builtin LDS-DMA with vmcnt(0), which would serialise exactly what that kernel
// overlaps; issued as asm the DMA is invisible to it and the kernel places its own counted waits.  The
// compiler's own vmcnt waits stay safe: an unknown extra VMEM operation in flight can only make a counted
// wait longer (returns are in order).  lds_base: wave-uniform LDS byte address; lane l writes base + 16 l.
// Address = uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: no address VALU, half the
// address registers.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // m0 is "reserved": the kernel has no other user of it
__device__ __forceinline__ void glds16_raw(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_base)
                 : "memory", "m0");
}
#pragma clang diagnostic pop
// a pointer the compiler may hold in VGPRs although every lane has the same value -> SGPR pair
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// s_waitcnt vmcnt(N) with a compile-time N (gfx9 encoding: vmcnt in bits 3:0 and 15:14)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// ---------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution: D[m][n] = sum_k A(m,k) * W[n][k]  (+ epilogue).
// W is always row-major [N][K], K contiguous (the PyTorch Linear layout; conv weights are
// repacked to [Cout][tap][Cin] at load time).
// ---------------------------------------------------------------------------------------
enum AMode : int32_t {
    A_PLAIN = 0,  // A row-major [M][lda]
    A_CONV = 1    // A gathered from a zero-bordered NHWC map: row m = (b,y,x), k = (tap, cin)
};

enum EpiKind : int32_t {
    EPI_STORE = 0,       // out32 [M][ldc] / out16 (row-major, or pixels of a zero-bordered NHWC
                         // map when out16_border) = act(acc + bias + res32 + res32b)
    EPI_RESID_SCALE = 1, // out32[m][n] = res32[m][n] + gamma[n]*(acc + bias[n])  (ViT proj / fc2)
    EPI_PATCH_EMBED = 2, // out32[(m/P)*(P+1) + 1 + m%P][n] = acc + bias[n] + pos[1 + m%P][n]
    EPI_CONVT = 4,       // ConvTranspose2d(2,2,s2) pixel shuffle: n = (dy*2+dx)*Cout + co
    EPI_HEAD_FINAL = 5,  // relu(acc+bias) . w2 + b2 -> relu -> / f_norm -> clamp  (N <= 32)
    EPI_HEAD_COMPOSED = 6 // the same behind the composed ConvTranspose o conv3x3 of the head (weights.hip compose_head): N = 4
                          // output phases x 32 channels on the half-resolution map, out32 = the full-resolution depth
};

enum Act : int32_t { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };

struct GemmParams {
    // problem
    int32_t M, N, K;
    // A operand
    const void* A;     // f16/bf16
    int64_t lda;       // elements (A_PLAIN)
    // A_CONV geometry: input map [B][Hp][Wp][Cin] with Hp = Hin + 2, Wp = Win + 2 (zero border)
    int32_t in_Hp, in_Wp, Cin;
    int32_t out_H, out_W;     // output pixels per image: M = B * out_H * out_W
    int32_t KH, KW, stride;   // 1x1 or 3x3 (pad (KH-1)/2); stride 1 or 2
    // W operand
    const void* W;     // [N][K]
    // epilogue
    const float* bias;     // [N] or [Cout] (EPI_CONVT) or null
    const float* gamma;    // EPI_RESID_SCALE
    const float* res32;    // optional f32 residual, same row mapping as out32
    const float* res32b;   // optional second f32 residual
    const float* pos;      // EPI_PATCH_EMBED: pos_embed [P+1][N]
    float* out32;          // optional f32 output
    void* out16;           // optional f16/bf16 output
    int64_t ldc;           // row stride (elements) of out32/out16/res32 for row-major outputs
    int64_t ldc16;         // row / pixel stride of out16 when it differs from ldc (0: same)
    // Split 16-bit output (pipeline.hip "split operands"): besides hi = T(v) at its place, lo = T(v - hi) is
    // stored lo_off16 elements further on, so that a consumer whose K axis reads [hi | lo] against
    // duplicated weights sees v to ~2^-22 instead of 2^-11.  0: plain 16-bit output.
    int32_t lo_off16;
    int32_t hi2_off16;     // with lo_off16: a second copy of hi this many elements on ([hi | lo | hi] operands)
    int32_t act;           // activation applied to the f16 output (and f32 for EPI_STORE)
    int32_t act16_only;    // 1: activation only on the out16 copy (out32 stays raw)
    int32_t out16_border;  // EPI_STORE/EPI_CONVT: out16 is zero-bordered [B][out_H+2][out_W+2][C]
    int32_t tokens_per_window;  // EPI_PATCH_EMBED: P
    int32_t Cout;          // EPI_CONVT: N = 4*Cout
    // EPI_HEAD_FINAL
    // f32 [9][ld]: the share of each 3x3 tap in `bias`, taken out again where the tap falls into the zero padding of a layer that
    // was composed into this convolution at load time (weights.hip compose_head: EPI_HEAD_COMPOSED, ld = 32; compose_features: the
    // halo tile's EPI_STORE with a bordered 16-bit output, ld = N)
    const float* tap_bias;
    const float* w2;       // [N]
    const float* b2;       // device scalar
    const float* f_norm;   // device [B] (one per image) or null (= 1): out32[m] = clamp(v / f_norm[b])
    float clamp_lo, clamp_hi;  // mod.rs:362 clamp(1e-4, 1e4); +-inf for the canonical head output
    int32_t pixels_per_image;
    // diagnostic builds only (tools/gemm_stamps.py): per-workgroup s_memrealtime stamps, or null
    unsigned long long* stamps;
    // Row segments with their own weights (the three ViTs of the encoder in one launch, pipeline.hip):
    // rows [0, seg1) use W / bias / gamma, [seg1, seg2) the *_s1 set, [seg2, M) the *_s2 set.  seg1 == 0:
    // one segment.  seg2 == 0: two.  Both must be multiples of the tile height (checked at launch).
    int32_t seg1, seg2;
    const void *W_s1, *W_s2;
    const float *bias_s1, *bias_s2, *gamma_s1, *gamma_s2;
    // MX fp8 operands (gemm_fp8.hip): e8m0 block scales of A and W in the per-wave layouts of mx_fp8.h, a_mt =
    // 128-row tiles of the A scale array; fp8 output (the next GEMM's A operand) with its scale array
    const uint8_t *a_scale, *w_scale, *w_scale_s1, *w_scale_s2;
    int32_t a_mt;
    uint8_t *out8, *out8_scale;
    int32_t out8_mt;
    // Algorithmic work of the launch for the profiler (bench.py's roofline): rows that belong to real tokens /
    // pixels (0: M; the merged ViT row space pads each segment to 256 rows) and the K of the layer itself (0: K; a
    // split [hi | lo] operand doubles or triples the K the kernel walks without adding algorithmic FLOPs)
    int32_t flop_rows, flop_k;
    // Context status word (me_status_flags): bit 0 is set when an f16 operand store met a magnitude beyond 65504
    // (gemm_launch fills it in from the calling context)
    unsigned* status;
    // tile rows of an XCD's patch in the tile walk (gemm_core.h tile_origin); 0: 8
    int32_t patch_rows;
    // EPI_STORE, 16-bit output without activation (the qkv linear, vit.rs:58-62): columns n < qcols leave multiplied
    // by qscale -- Q of the attention kernel arrives scaled by 1/sqrt(head_dim) * log2(e) with ONE rounding, the
    // product (acc + bias) * qscale rounded to 16 bits, instead of being scaled and rounded again where it is
    // consumed (attention.hip attention2_kernel).  qcols == 0: off.  A multiple of 64 (whole heads).
    int32_t qcols;
    float qscale;
    // LayerNorm of the updated token rows inside EPI_RESID_SCALE (gemm_core.h resid_ln_epilogue; the 352-row tile, N in
    // {256, 512, 1024}): besides x += gamma * (A W^T + b) the launch writes ln_out16[m][:] = LayerNorm(x[m][:]) * w + b
    // with the weights of the row's segment -- the next linear's operand (vit.rs:165-169: norm1 / norm2 of Block::forward).
    // The N / 256 column tiles of a row tile exchange their partial (mean, M2) through ln_stats ([row tile][column tile]
    // [352] 8-byte granules) and count their arrivals in ln_count (one word per row tile, 64 bytes apart; the count only
    // ever grows by N / 256 per launch, so it needs no reset between launches or graph replays).  ln_out16 == null: off.
    void* ln_out16;
    const float *ln_w, *ln_b, *ln_w_s1, *ln_b_s1, *ln_w_s2, *ln_b_s2;
    float ln_eps;
    unsigned long long* ln_stats;
    unsigned* ln_count;
    // polls a workgroup spends waiting for its row tile's other column tiles before it raises ME_STATUS_SYNC_TIMEOUT
    // (0: 2^20, about 2 s; gemm_launch fills it in from ME_LN_SPIN_LIMIT, a test knob)
    int32_t ln_spin_limit;
    // compute units the launch stream may use when that is fewer than the device has (a stream created with a CU mask;
    // gemm_launch fills it in, gemm.hip stream_cu_count): persistent grids are sized from it.  0: all of them
    int32_t cu_granted;
    // query form (gemm.hip gemm_lnf_resident): the launcher writes the number of workgroups it would keep resident
    // here and returns without launching.  Host pointer; null for a launch
    int32_t* resident_out;
    // development A/B (ME_GELU_BATCH=0): fc1's GELU four values at a time instead of a pass's sixteen at once
    int32_t gelu_per_granule;
    // persistent kernels: at most this many workgroups (a multiple of 8), so that a launch on another stream finds
    // free CUs beside this one; 0: as many as are resident
    int32_t grid_cap;
    // Deterministic split-K (gemm_core.h gemm_kernel<..., SPLITK>; the 128x128 tile, EPI_STORE): split_k > 1 work items per tile,
    // each over a K range; every one writes its f32 partial (accumulator layout) to splitk_ws[(tile * split_k + s)][BM * BN] and
    // counts its arrival in splitk_cnt[tile] (zero before the launch; the last arriver resets it); the LAST arriver sums the
    // partials in split order -- its own included, so the result does not depend on who is last -- and runs the epilogue.
    int32_t split_k;
    float* splitk_ws;
    unsigned* splitk_cnt;
    // Tile queue of the launch stream (gemm_launch fills it in), or null for the static tile order.
    // Word 32 x: next-tile ticket of XCD x (x < 8), word 256: exited workgroups -- one 128-byte line each
    // (on one line the 512 prologue draws of a launch serialise in a single L2 channel).
    unsigned* queue;
};
constexpr int kGemmQueueWords = 9 * 32;

// row tiles of a launch tiled per row segment (gemm_core.h seg_tile_rows; the 352-row tile)
template <int BM>
__host__ __device__ __forceinline__ int seg_row_tiles(int M, int seg1, int seg2) {
    if (seg1 == 0) return (M + BM - 1) / BM;
    const int e1 = seg2 ? seg2 : M;
    return (seg1 + BM - 1) / BM + (e1 - seg1 + BM - 1) / BM + (seg2 ? (M - seg2 + BM - 1) / BM : 0);
}


// dtype: ME_DTYPE_F16 / ME_DTYPE_BF16.  Picks a tile configuration from (M, N, K).
void gemm_launch(const GemmParams& p, AMode amode, EpiKind epi, int32_t dtype, hipStream_t stream,
                 int32_t force_cfg = -1);
// The depth head's final 3x3 (128 -> 32) + 1x1 (32 -> 1) on a pixel halo tile with the weights held in registers
// (head_conv.hip); gemm_launch takes it for EPI_HEAD_FINAL where the shape fits (ME_HEAD_HALO=0: the implicit-GEMM tile)
bool head_final_halo_fits(const GemmParams& p);
// the head's ConvTranspose + conv3x3 + ReLU + conv1x1 + ReLU + / f_norm + clamp as ONE launch on the half-resolution map
// (gemm_*_conv.hip: the 128-channel halo tile with the EPI_HEAD_COMPOSED epilogue).  p: A = bordered [B][H+2][W+2][Cin] 16-bit,
// W = composed [128][9][Cin], bias = f32 [32], tap_bias = f32 [9][32], w2 / b2 / f_norm / clamp / pixels_per_image (of the
// FULL-resolution map) as for EPI_HEAD_FINAL, out32 = depth [B][2H][2W]
void head_composed_launch(const GemmParams& p, int32_t dtype, hipStream_t stream);
bool head_composed_fits(const GemmParams& p);
// compute units `stream` may use: the bits of its CU mask (hipExtStreamGetCUMask), the device's count when the stream
// has no mask or the runtime cannot say
int stream_cu_count(hipStream_t stream);
// workgroups of the fused residual + LayerNorm launch (gemm_core.h resid_ln_epilogue) that are resident at once on
// the CUs `stream` may use; a launch with N columns needs 8 * N / 256 of them (one row tile per XCD with all its column
// tiles: they wait for one another)
int gemm_lnf_resident(int32_t dtype, hipStream_t stream);
void head_final_halo_launch(const GemmParams& p, int32_t dtype, hipStream_t stream);

// MX fp8 x fp8 (gemm_fp8.hip): EPI_STORE (f16 out16, or fp8 out8 + scales) / EPI_RESID_SCALE
void gemm_fp8_launch(const GemmParams& p, EpiKind epi, hipStream_t stream);
// f16 [rows][K] -> e4m3 + e8m0 scales in the weight (1) or activation (0) scale layout
void quantize_f16_to_fp8_launch(const void* src16, uint8_t* dst8, uint8_t* scales, int64_t rows, int32_t K,
                                int32_t weight_layout, hipStream_t stream);
int gemm_num_configs();
const char* gemm_config_name(int cfg);

// ---------------------------------------------------------------------------------------
// Attention: qkv [rows][3*C] (q | k | v, each [heads][64]), rows = windows * tokens;
// out [rows][C].  softmax(q*scale . k^T) v per (window, head); head_dim is 64.
// ---------------------------------------------------------------------------------------
// Row segments of the encoder's merged ViT launches (pipeline.hip): segment 0 = rows [0, seg1), 1 =
// [seg1, seg2), 2 = [seg2, ...).  Windows are contiguous (stride `tokens` rows) inside a segment; win0 /
// win1 are the numbers of windows of segments 0 and 1.  seg1 == 0: no segmentation.
struct RowSegs {
    int64_t seg1 = 0, seg2 = 0;
    int32_t win0 = 0, win1 = 0;
    const float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;  // LayerNorm weights of 1 and 2
};
// out8 != nullptr: the output is written as MX fp8 bytes [rows][C] + activation-layout block scales (mx_fp8.h,
// out8_mt = 128-row tiles of the operand) instead of 16-bit `out`
void attention_launch(const void* qkv, void* out, int32_t windows, int32_t tokens, int32_t heads,
                      int32_t dtype, hipStream_t stream, const RowSegs* segs = nullptr, uint8_t* out8 = nullptr,
                      uint8_t* out8_scale = nullptr, int64_t out8_mt = 0, bool q_prescaled = false);
// attention3.hip: the kernel attention_launch runs for a pre-scaled Q (48 queries per wave on 16x16x32 MFMAs, the query beyond
// whole wave units on the vector pipe)
void attention3_launch(const void* qkv, void* out, int32_t windows, int32_t tokens, int32_t heads, int32_t dtype, hipStream_t stream,
                       const RowSegs& segs, uint8_t* out8, uint8_t* out8_scale, int64_t out8_mt, float defer_thr);
// 1/sqrt(head_dim) (vit.rs:47) times log2(e): the factor a pre-scaled Q carries (GemmParams::qscale of the qkv launch)
constexpr float kAttnQScale = 0.125f * 1.44269504088896340736f;

// ---------------------------------------------------------------------------------------
// Row-wise and layout kernels (elementwise.hip)
// ---------------------------------------------------------------------------------------
// y16[r][:] = (x[r][:] - mean) / sqrt(var + eps) * w + b ; optional f32 copy y32.
void layernorm_launch(const float* x, const float* w, const float* b, void* y16, float* y32,
                      int64_t rows, int32_t dim, float eps, int32_t dtype, hipStream_t stream, const RowSegs* segs = nullptr);
// the same with the result quantised to MX fp8 (y8 [rows][dim] + block scales, activation layout)
void layernorm_fp8_launch(const float* x, const float* w, const float* b, uint8_t* y8, uint8_t* yscale, int64_t rows,
                          int32_t dim, float eps, hipStream_t stream, const RowSegs* segs = nullptr);
// u8 HWC -> f32 NCHW, reconstruction.rs:114-124
void preprocess_u8_launch(const uint8_t* rgb, float* img, int32_t batch, int32_t size,
                          hipStream_t stream);
// f32 NCHW -> f16 NCHW (same layout)
void cast_f32_to_16_launch(const float* src, void* dst, int64_t count, int32_t dtype,
                           hipStream_t stream);
void cast_16_to_f32_launch(const void* src, float* dst, int64_t count, int32_t dtype,
                           hipStream_t stream);
// bilinear resample of f32 planes [planes][in][in] -> 16-bit [planes][out][out] (encoder.rs:125-140)
void bilinear_launch(const float* src32, void* dst16, int32_t planes, int32_t in_size,
                     int32_t out_size, int32_t align_corners, int32_t dtype, hipStream_t stream);
// windows of the three pyramid levels -> patch matrix [(B*35)*g*g][3*256]; window order is
// image-major here: row = ((b*35 + win)*g*g + patch)
void patchify_launch(const void* x0, const void* x1, const void* x2, void* patches, int32_t batch,
                     int32_t grid, int32_t dtype, hipStream_t stream);
// one planar image stack [W][3][16g][16g] -> patch matrix [W*g*g][768]
void patchify_windows_launch(const void* xs16, void* patches, int32_t windows, int32_t grid,
                             int32_t dtype, hipStream_t stream);
// cls rows: tokens[w*(P+1)][:] = cls + pos[0]
void cls_rows_launch(float* tokens, const float* cls, const float* pos, int32_t windows,
                     int32_t tokens_per_window, int32_t dim, hipStream_t stream);
// encoder.rs:158-208 reshape_feature + merge: token rows of `steps*steps` windows (first window
// index win0 of each image's 35) -> NHWC f16 map [B][side][side][C], side = merged size.
// src is f32 tokens (src32) or f16 tokens (src16) with rows (b*35+win)*(P+1) + 1 + patch.
// split: dst16 holds [hi | lo] pairs of 2*dim channels per pixel (f32 source only)
void merge_launch(const float* src32, const void* src16, void* dst16, int32_t batch,
                  int32_t windows_per_image, int32_t win0, int32_t steps, int32_t padding,
                  int32_t grid, int32_t dim, int32_t dtype, hipStream_t stream, int32_t split = 0);
// NHWC (f16 or f32, optional zero border) <-> NCHW f32
void nhwc16_to_nchw32_launch(const void* src16, float* dst, int32_t batch, int32_t H, int32_t W,
                             int32_t C, int32_t border, int32_t dtype, hipStream_t stream, int32_t split = 0);
void nhwc32_to_nchw32_launch(const float* src, float* dst, int32_t batch, int32_t H, int32_t W,
                             int32_t C, hipStream_t stream);
// NCHW f32 -> NHWC: optional f32 copy, optional f16 copy (zero border, optional relu)
void nchw32_to_nhwc_launch(const float* src, float* dst32, void* dst16, int32_t batch, int32_t H,
                           int32_t W, int32_t C, int32_t border, int32_t relu16, int32_t dtype,
                           hipStream_t stream, int32_t split = 0);
// NHWC f32 -> 16-bit NHWC with a zero border (optional relu)
void nhwc32_to_16b_launch(const float* src, void* dst16b, int32_t batch, int32_t H, int32_t W,
                          int32_t C, int32_t relu, int32_t dtype, hipStream_t stream);
// channel concat of two NHWC f16 maps
void concat_channels_launch(const void* a, const void* b, void* dst, int64_t pixels, int32_t Ca,
                            int32_t Cb, hipStream_t stream);
// FOV tail (fov.rs:67-87)
void fov_add_relu_launch(const float* lin /*[B][P][C] tokens (cls dropped by caller offset)*/,
                         const float* low /*NHWC f32 [B][g][g][C]*/, void* dst16 /*bordered*/,
                         int32_t batch, int32_t grid, int32_t C, int32_t tokens_per_window,
                         int32_t dtype, hipStream_t stream);
void fov_final_launch(const void* x16 /*NHWC [B][k][k][C]*/, const float* w /*[k][k][C]*/,
                      const float* bias, float* fov_deg, float* f_norm, int32_t batch, int32_t k,
                      int32_t C, int32_t dtype, hipStream_t stream);

// ---------------------------------------------------------------------------------------
// Output back end (output.hip)
// ---------------------------------------------------------------------------------------
void depth_clamp_minmax_launch(float* depth, int64_t count, float* minmax_dev /*[2]*/,
                               hipStream_t stream);
// range_dev: device {min, max} (from depth_clamp_minmax_launch) used instead of the scalars when not null
void stereogram_launch(const float* depth, int32_t rows, int32_t cols, float min_depth,
                       float max_depth, const float* range_dev, int32_t out_w, int32_t out_h, float amplitude,
                       const uint8_t* noise, uint8_t* out, hipStream_t stream);
void depthmap_rgb_launch(const float* depth, int64_t count, float min_depth, float max_depth,
                         const float* range_dev, uint8_t* rgb, hipStream_t stream);
// synchronises the stream (the counts come back to the host); workspace: mesh_workspace_bytes() device bytes
size_t mesh_workspace_bytes(int32_t width, int32_t height);
void mesh_index_run(const float* depth_dev, int32_t width, int32_t height, int32_t* vertex_index_dev,
                    int32_t* faces_dev /*nullable*/, int64_t* nverts, int64_t* nfaces, void* workspace,
                    hipStream_t stream);
void mesh_vertices_launch(const float* depth, int32_t width, int32_t height,
                          const int32_t* vertex_index, float xm, float ym, float* uv, float* xyz,
                          hipStream_t stream);
// OBJ text on the device (obj_format.hip): measure returns the exact size (header included), write fills the buffer
size_t obj_format_workspace_bytes(int64_t nverts, int64_t nfaces);
int64_t obj_format_measure(const float* uv, const float* xyz, const uint8_t* vertex_rgb, const int32_t* faces,
                           int64_t nverts, int64_t nfaces, bool tex, int64_t header_bytes, void* workspace,
                           hipStream_t stream);
void obj_format_write(const float* uv, const float* xyz, const uint8_t* vertex_rgb, const int32_t* faces, int64_t nverts,
                      int64_t nfaces, bool tex, char* text, void* workspace, hipStream_t stream);
void format_f64_launch(const double* v, int64_t n, char* out, int stride, int* lens, hipStream_t stream);
void obj_vertex_colors_launch(const int32_t* vindex, const uint8_t* pixel_rgb, int64_t npix, uint8_t* vertex_rgb,
                              hipStream_t stream);

}  // namespace me
