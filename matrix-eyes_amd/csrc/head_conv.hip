// The depth head's last two layers in one kernel (mod.rs:83-94,329-333): Conv2d 3x3 128 -> 32 at full resolution, ReLU,
// Conv2d 1x1 32 -> 1, ReLU, then canonical / f_norm and the clamp (mod.rs:340-362).
//
// As an implicit GEMM (gemm_core.h gemm_kernel<256, 32>, EPI_HEAD_FINAL) this layer is bound by what a CU can bring into
// LDS: with 32 output channels every staged activation byte feeds 32 MFMA columns, and the nine taps restage the same
// pixels nine times (18 slabs of 32 KiB for 1 MFLOP each; 0.17 of the MFMA roofline).  Here:
//   * a workgroup works on a tile of 12 x 16 output pixels and stages its 14 x 18 pixel HALO once, all 128 input channels
//     (252 pixels of 256 bytes = 63 KiB, LDS-DMA, double buffered: the next tile's halo lands while this one is used);
//   * the weights never pass through LDS: wave (kh, g) -- kh = which 64 input channels, g = which three pixel rows of
//     the tile -- holds the 9 taps x 32 output channels x its 64 input channels as 36 MFMA operand fragments in 144
//     registers for the whole (persistent) kernel;
//   * a fragment of halo pixels (16 pixels of one halo row at a column offset dx, 32 channels) is read from LDS once
//     and used for every output row and tap row it belongs to: 30 fragment reads feed a wave's 108 MFMAs per tile;
//   * the two channel halves of a pixel row meet through LDS, and the kh = 0 wave finishes: bias, ReLU, the 1x1
//     convolution as a dot product over the 32 channels (8 in the lane, the other 24 in three more lanes), ReLU,
//     1 / f_norm, clamp, one f32 per pixel.
// MFMA orientation as in gemm_core.h: A operand = weights (16 output channels x 32 k), B operand = pixels (16 x 32 k);
// a lane's accumulator holds channels 4 (lane >> 4) + r of pixel lane & 15.
#include <cstdlib>

#include "gemm_core.h"

namespace me {

namespace {

constexpr int HEAD_TH = 12;                            // tile rows (12 x 16 output pixels)
constexpr int HEAD_HALO_PX = (HEAD_TH + 2) * 18;       // 252
constexpr int HEAD_HALO_BYTES = HEAD_HALO_PX * 256;    // 64512
constexpr int HEAD_DMA = HEAD_HALO_PX / 4;             // 63 LDS-DMA instructions (4 pixels each) per halo
constexpr int HEAD_SCR = 4 * 6 * 64 * 16;              // the kh = 1 waves' partial sums: 24 KiB
constexpr int HEAD_SMEM = 2 * HEAD_HALO_BYTES + HEAD_SCR;
static_assert(HEAD_HALO_PX % 4 == 0, "whole LDS-DMA instructions");
static_assert(HEAD_SMEM <= 160 * 1024, "LDS");

#ifdef ME_HEAD_STAMPS
// development: per-wave clocks of a tile's phases (tools/head_stamps.py): [0] halo wait + barrier, [1] MFMAs (with the
// halo requests of the next tile), [2] partial sums + barrier, [3] finishing, [4] whole kernel, [5] tiles
__device__ unsigned long long* g_head_stamps;
#define HEAD_PH(i)                                          \
    do {                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ph[i] += now_ - t_last;                             \
        t_last = now_;                                      \
    } while (0)
#else
#define HEAD_PH(i)
#endif

template <typename T>
__global__ __launch_bounds__(512, 2) void head_halo_kernel(const GemmParams p) {
    typedef typename MfmaOp<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, g = wave & 3;
    const int j = lane & 15, kq = lane >> 4;
    const int tiles_x = p.out_W >> 4, tiles_y = p.out_H / HEAD_TH;
    const int ntiles = (p.M / (p.out_H * p.out_W)) * tiles_x * tiles_y;
    const int64_t in_row_bytes = (int64_t)p.in_Wp * 256;

    // ---- this wave's weights: fragment (tap, ks, nh) = W[nh * 16 + j][tap][64 kh + 32 ks + 8 kq .. + 7]
    frag wf[9][2][2];
    {
        const T* w = (const T*)p.W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
                    wf[tap][ks][nh] = *reinterpret_cast<const frag*>(w + ((int64_t)(nh * 16 + j) * 9 + tap) * 128 + 64 * kh + 32 * ks + 8 * kq);
    }

    // ---- halo staging: instruction i of the halo covers halo pixels 4 i .. 4 i + 3; lane l brings the 16-byte chunk that
    // belongs at chunk POSITION l & 15 of pixel 4 i + (l >> 4): channel chunk (l & 15) ^ (halo column & 15) -- the read side
    // then finds chunk c of a pixel at position c ^ (column & 15), and the 16 pixels of a fragment read hit 16 different
    // positions.  Wave w issues instructions w, w + 8, ...  (An LDS-DMA whose lines come from beyond L2 holds its wave in
    // issue for about 150 cycles: the eight per wave cost 1200 cycles of every wave at once.  Measured alternatives:
    // sent one at a time between the MFMAs, the two waves of a SIMD taking turns -- no change; all sixteen of a SIMD's
    // pair by the kh = 1 wave behind its MFMAs, the kh = 0 wave none -- that wave's 2400 cycles become the tile's
    // critical path, 200 against 176 us.)
    unsigned soff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = wave + 8 * k;
        const int hp = (i < HEAD_DMA ? i : HEAD_DMA - 1) * 4 + (lane >> 4);
        const int hr = hp / 18, hc = hp - hr * 18;
        soff[k] = (unsigned)(hr * in_row_bytes + hc * 256 + (((lane & 15) ^ (hc & 15)) << 4));
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_address(smem));
    // Tile walk.  Neighbouring tiles share two halo columns or rows, and each XCD has its own L2 (workgroup w runs on XCD
    // w & 7): where the map allows, XCD x works on the strip of tile columns [x sw, (x + 1) sw), sw = tiles_x / 8, its
    // workgroups side by side in row-major order of the strip, so that what two tiles share is fetched from beyond L2 once.
    // Otherwise (narrow maps, a capped grid that is no multiple of 8) workgroups take tiles in plain row-major order.
    const bool strips = tiles_x % 8 == 0 && gridDim.x % 8 == 0;
    const int sw = strips ? tiles_x >> 3 : tiles_x;
    const int x_base = strips ? (int)(blockIdx.x & 7) * sw : 0;
    const int per_image_local = tiles_y * sw;
    const int n_local = strips ? ntiles >> 3 : ntiles;
    const int t_step = strips ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    auto tile_origin_px = [&](int t, int& b, int& y0, int& x0) {
        b = t / per_image_local;
        const int r = t - b * per_image_local;
        const int ty = r / sw;
        y0 = ty * HEAD_TH, x0 = (x_base + r - ty * sw) << 4;
    };
    auto halo_base = [&](int b, int y0, int x0) {  // bordered input pixel (y0, x0) of the tile's image = halo pixel (0, 0)
        return uniform_ptr((const char*)p.A + ((int64_t)b * p.in_Hp + y0) * in_row_bytes + (int64_t)x0 * 256);
    };
    auto stage_halo = [&](const char* base, int buf) {
        const unsigned dst = smem_base + buf * HEAD_HALO_BYTES;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = wave + 8 * k;
            if (i < HEAD_DMA) glds16_raw(base, soff[k], dst + i * 1024);
        }
    };

    // ---- fragment read addresses: halo pixel (3 g + hr, j + dx), channel chunk 8 kh + 4 ks + kq at position chunk ^ (column & 15)
    unsigned rd[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int hc = j + dx;
        rd[dx] = (unsigned)((3 * g * 18 + hc) * 256 + (((8 * kh + kq) ^ (hc & 15)) << 4));
    }
    // epilogue constants: this lane's 8 output channels
    float bv[2][4], w2v[2][4];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nh * 16 + 4 * kq + r;
            bv[nh][r] = p.bias[n], w2v[nh][r] = p.w2[n];
        }
    const float b2 = p.b2[0];
    char* scr = smem + 2 * HEAD_HALO_BYTES + (g * 6 * 64 + lane) * 16;

    int t = strips ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (t >= n_local) return;
    // (b, y0, x0) and 1 / f_norm's operand of the tile in hand; the next tile's are worked out -- and its f_norm requested --
    // a tile ahead: a load issued where the value is needed costs the finishing wave a trip to memory per tile
    int b, y0, x0;
    tile_origin_px(t, b, y0, x0);
    float fn = p.f_norm ? p.f_norm[b] : 1.0f;
    stage_halo(halo_base(b, y0, x0), 0);
    int buf = 0;
#ifdef ME_HEAD_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_last = t_begin;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the first tile's halo has landed
    asm volatile("" ::: "memory");
    // Per tile TWO barriers, placed so that the finishing wave's vector work runs beside its SIMD partner's MFMAs:
    //   B2: every wave's MFMAs on this halo are done, the kh = 1 waves' sums are in the scratch, and -- each wave having
    //       waited for its own requests first -- the NEXT tile's halo has landed;
    //   B3: the kh = 0 waves have read the scratch.
    // Behind B3 a kh = 1 wave starts the next tile at once (alone on its SIMD's matrix pipe) while the kh = 0 wave of that
    // SIMD finishes this one (1900 cycles of vector instructions); then the kh = 0 wave's MFMAs run while the kh = 1
    // wave waits at B2.  With one barrier pair around the whole epilogue instead the matrix pipe stood still for it: 7400
    // cycles per tile of which 3500 are MFMAs.
    for (;;) {
        HEAD_PH(0);
        const int tn = t + t_step;
        const bool has_next = tn < n_local;
        int nb = b, ny0 = y0, nx0 = x0;
        if (has_next) tile_origin_px(tn, nb, ny0, nx0);
        const float fn_next = p.f_norm ? p.f_norm[nb] : 1.0f;
        // the next tile's halo: requested now, it lands while this one is used; its buffer was read last before the previous B2
        if (has_next) stage_halo(halo_base(nb, ny0, nx0), buf ^ 1);
        const char* halo = smem + buf * HEAD_HALO_BYTES;
        f32x4 acc[3][2];
#pragma unroll
        for (int y = 0; y < 3; ++y) acc[y][0] = acc[y][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag f[5];
#pragma unroll
                for (int hr = 0; hr < 5; ++hr)
                    f[hr] = *reinterpret_cast<const frag*>(halo + (rd[dx] ^ (unsigned)(ks << 6)) + hr * (18 * 256));
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int y = 0; y < 3; ++y)
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh)
                            acc[y][nh] = MfmaOp<T>::run(wf[dy * 3 + dx][ks][nh], f[y + dy], acc[y][nh]);
            }
#ifdef ME_HEAD_STAMPS
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]));
        HEAD_PH(1);
#endif
        if (kh == 1) {
#pragma unroll
            for (int y = 0; y < 3; ++y)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) *reinterpret_cast<f32x4*>(scr + (y * 2 + nh) * 1024) = acc[y][nh];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // B2
        asm volatile("" ::: "memory");
        HEAD_PH(2);
        f32x4 other[3][2];
        if (kh == 0) {
#pragma unroll
            for (int y = 0; y < 3; ++y)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) other[y][nh] = *reinterpret_cast<const f32x4*>(scr + (y * 2 + nh) * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // B3
        asm volatile("" ::: "memory");
        if (kh == 0) {
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                float s = 0.f;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += fmaxf((acc[y][nh][r] + other[y][nh][r]) + bv[nh][r], 0.f) * w2v[nh][r];
                }
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (lane < 16) {
                    float v = fmaxf(s + b2, 0.f);
                    if (p.f_norm) v = v / fn;
                    v = fminf(fmaxf(v, p.clamp_lo), p.clamp_hi);
                    p.out32[((int64_t)b * p.out_H + y0 + 3 * g + y) * p.out_W + x0 + lane] = v;
                }
            }
        }
        buf ^= 1;
#ifdef ME_HEAD_STAMPS
        HEAD_PH(3);
        ph[5] += 1;
#endif
        if (!has_next) break;
        t = tn, b = nb, y0 = ny0, x0 = nx0, fn = fn_next;
    }
#ifdef ME_HEAD_STAMPS
    ph[4] = __builtin_amdgcn_s_memtime() - t_begin;
    if (g_head_stamps && lane == 0)
        for (int i = 0; i < 6; ++i) g_head_stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}

}  // namespace

#ifdef ME_HEAD_STAMPS
}  // namespace me
extern "C" int32_t me_debug_set_head_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(me::g_head_stamps), &buf, sizeof buf) == hipSuccess ? 0 : 1;
}
namespace me {
#endif

bool head_final_halo_fits(const GemmParams& p) {
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.Cin == 128 && p.N == 32 && p.K == 9 * 128 && p.out_H % HEAD_TH == 0 &&
           p.out_W % 16 == 0 && p.in_Hp == p.out_H + 2 && p.in_Wp == p.out_W + 2 && p.out32 && p.w2 && p.b2 && p.bias &&
           p.M % (p.out_H * p.out_W) == 0 && (!p.f_norm || p.pixels_per_image == p.out_H * p.out_W);
}

void head_final_halo_launch(const GemmParams& p, int32_t dtype, hipStream_t stream) {
    ME_CHECK(head_final_halo_fits(p), ME_ERR_BAD_SHAPE, "head: the halo kernel takes 3x3 / 128 -> 32 on maps of 12 x 16 pixel multiples");
    static PerDeviceOnce once;  // per device: the LDS the kernel asks for, and how many workgroups are resident (one per CU)
    const int resident = per_device_once(once, [&](int dev) {
        ME_HIP(hipFuncSetAttribute((const void*)head_halo_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, HEAD_SMEM));
        ME_HIP(hipFuncSetAttribute((const void*)head_halo_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, HEAD_SMEM));
        int cus = 0;
        ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        return cus > 0 ? cus : 256;
    });
    const int ntiles = (p.M / (p.out_H * p.out_W)) * (p.out_H / HEAD_TH) * (p.out_W / 16);
    int grid = ntiles < resident ? ntiles : resident;
    if (p.grid_cap >= 8 && grid > p.grid_cap) grid = p.grid_cap;
    if (dtype == ME_DTYPE_F16)
        hipLaunchKernelGGL(head_halo_kernel<f16>, dim3((unsigned)grid), dim3(512), HEAD_SMEM, stream, p);
    else if (dtype == ME_DTYPE_BF16)
        hipLaunchKernelGGL(head_halo_kernel<bf16>, dim3((unsigned)grid), dim3(512), HEAD_SMEM, stream, p);
    else
        fail(ME_ERR_BAD_ARG, "head: bad dtype %d", dtype);
    ME_HIP(hipGetLastError());
}

}  // namespace me
