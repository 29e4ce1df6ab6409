// Box calibration for bench.py: two FIXED loops whose rates say what THIS device gives -- the boxes
// of the pool differ by +-3 % on the whole step and the guide reports 12 % device to device on MFMA-dense loops, more
// than a round's whole gain, so a driver-timed headline alone cannot tell progress from the box it ran on.
//
//   THESE TWO KERNELS AND THEIR LAUNCH SHAPES MUST NEVER BE EDITED: every `calibration` object that bench.py has ever
//   printed is only comparable with the next one as long as the loops are the same instructions on the same data.
//   (A change needs a new entry point and a new reference calibration in bench.py.)
//
// 1. calib_mfma_kernel: 256 workgroups x 512 threads (two waves per SIMD on 256 CUs), every wave 40 000 x 32
//    v_mfma_f32_16x16x32_f16 on 32 accumulator tiles (the register footprint of the 256 x 256 GEMM tile) with operands
//    held in registers -- pseudo-random bits in [-1, 1), so the data paths toggle as they do on real activations.  No
//    memory traffic inside the loop: what it measures is the matrix pipe at the clock the part holds under a dense
//    16-bit MFMA stream (power-limited; about 25 ms, long enough for the clock to settle).  Lane 0 of every wave reads
//    s_memtime (shader clock) and s_memrealtime (100 MHz) at both ends: their ratio is the in-kernel clock.
// 2. calib_copy_kernel: a 16-byte-per-lane copy of 512 MiB to another 512 MiB (more than the 256 MiB Infinity Cache),
//    10 launches of 2048 x 256 threads striding the buffer: bytes read + written per second.
#include <vector>

#include "common.h"
#include "model.h"

namespace me {
namespace {

typedef _Float16 cal_f16x8 __attribute__((ext_vector_type(8)));
typedef float cal_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCalWorkgroups = 256, kCalThreads = 512, kCalIters = 40000, kCalMfmaPerIter = 32;
constexpr size_t kCalCopyBytes = 512ull << 20;
constexpr int kCalCopyReps = 10;

__device__ __forceinline__ _Float16 cal_value(unsigned i) {  // fixed pseudo-random value in [-1, 1)
    unsigned x = i * 2654435761u + 0x9e3779b9u;
    x ^= x >> 16, x *= 0x85ebca6bu, x ^= x >> 13, x *= 0xc2b2ae35u, x ^= x >> 16;
    return (_Float16)((float)(int)(x & 2047u) * (1.0f / 1024.0f) - 1.0f);
}

__global__ __launch_bounds__(512, 2) void calib_mfma_kernel(unsigned long long* __restrict__ clocks, float* __restrict__ sink,
                                                            int iters) {
    const unsigned lane = threadIdx.x;
    cal_f16x8 a[4], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = cal_value((lane * 4 + i) * 8 + j);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = cal_value(0x40000u + (lane * 8 + i) * 8 + j);
    cal_f32x4 c[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = cal_f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[i], c[i][j], 0, 0, 0);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += c[i][j][0] + c[i][j][3];
    asm volatile("" : "+v"(acc));  // (the clock reads stay behind the accumulators' last use)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((lane & 63u) == 0) {
        const size_t w = (size_t)blockIdx.x * (kCalThreads / 64) + (lane >> 6);
        clocks[2 * w] = t1 - t0;
        clocks[2 * w + 1] = r1 - r0;
    }
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

// out[0] MFMA loop TFLOP/s, [1] in-kernel shader clock (GHz) during it, [2] copy GB/s (read + written), [3] MFMA loop ms,
// [4] ms per copy launch, [5] compute units of the device
void calibrate(me_ctx* ctx, double* out) {
    hipStream_t s = ctx->stream;
    const size_t nwaves = (size_t)kCalWorkgroups * (kCalThreads / 64);
    unsigned long long* clocks = nullptr;
    float* sink = nullptr;
    char* copy = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Free {
        unsigned long long*& a;
        float*& b;
        char*& c;
        hipEvent_t &e0, &e1;
        ~Free() {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            if (c) (void)hipFree(c);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } guard{clocks, sink, copy, e0, e1};
    ME_HIP(hipMalloc((void**)&clocks, nwaves * 16));
    ME_HIP(hipMalloc((void**)&sink, (size_t)kCalWorkgroups * kCalThreads * 4));
    ME_HIP(hipMalloc((void**)&copy, 2 * kCalCopyBytes));
    ME_HIP(hipMemsetAsync(copy, 0x5a, 2 * kCalCopyBytes, s));
    ME_HIP(hipEventCreate(&e0));
    ME_HIP(hipEventCreate(&e1));
    // MFMA loop: a quarter-length launch first (code object, clock ramp), then the timed one
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(kCalWorkgroups), dim3(kCalThreads), 0, s, clocks, sink, kCalIters / 4);
    ME_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(kCalWorkgroups), dim3(kCalThreads), 0, s, clocks, sink, kCalIters);
    ME_HIP(hipEventRecord(e1, s));
    ME_HIP(hipGetLastError());
    ME_HIP(hipEventSynchronize(e1));
    float mfma_ms = 0.f;
    ME_HIP(hipEventElapsedTime(&mfma_ms, e0, e1));
    std::vector<unsigned long long> h(nwaves * 2);
    ME_HIP(hipMemcpy(h.data(), clocks, nwaves * 16, hipMemcpyDeviceToHost));
    double shader = 0.0, real = 0.0;
    for (size_t w = 0; w < nwaves; ++w) shader += (double)h[2 * w], real += (double)h[2 * w + 1];
    const double flop = (double)nwaves * kCalIters * kCalMfmaPerIter * (16.0 * 16 * 32 * 2);
    // copy: one untimed launch, then kCalCopyReps timed
    const size_t n16 = kCalCopyBytes / 16;
    hipLaunchKernelGGL(calib_copy_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)copy, (uint4*)(copy + kCalCopyBytes), n16);
    ME_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < kCalCopyReps; ++i)
        hipLaunchKernelGGL(calib_copy_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)copy, (uint4*)(copy + kCalCopyBytes), n16);
    ME_HIP(hipEventRecord(e1, s));
    ME_HIP(hipGetLastError());
    ME_HIP(hipEventSynchronize(e1));
    float copy_ms = 0.f;
    ME_HIP(hipEventElapsedTime(&copy_ms, e0, e1));
    int cus = 0;
    ME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    out[0] = flop / (mfma_ms * 1e-3) / 1e12;
    out[1] = real > 0.0 ? shader / real * 0.1 : 0.0;  // s_memrealtime counts at 100 MHz
    out[2] = 2.0 * (double)kCalCopyBytes * kCalCopyReps / (copy_ms * 1e-3) / 1e9;
    out[3] = mfma_ms;
    out[4] = copy_ms / kCalCopyReps;
    out[5] = cus;
}

}  // namespace me
