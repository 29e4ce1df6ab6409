// Diagnostic: sustained f16 MFMA rate of the two gfx950 shapes with operands in registers (random bits, so
// the data paths toggle) -- which shape holds the higher clock under the board's power limit?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.hip -o gpurun_out/mfma_shapes && gpurun_out/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    f16x8 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = in[(lane + 64 * i) & 4095];
    for (int i = 0; i < 8; ++i) b[i] = in[(lane * 3 + 64 * i + 17) & 4095];
    float acc_sum = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 c[8][4];  // 128 accumulator registers, as the 256x256 GEMM tile has per wave
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) c[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[i], c[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) acc_sum += c[i][j][0] + c[i][j][3];
    } else {
        f32x16 c[4][2];  // the same 128 registers as 32x32 blocks
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j)
                for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)  // same MACs per iteration: 8 blocks x 2 x (32*32*16) = 32 x (16*16*32)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 2 * k], b[i + 4 * k], c[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) acc_sum += c[i][j][0] + c[i][j][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc_sum;
}

int main() {
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    f16x8* in;
    float* out;
    hipMalloc((void**)&in, h.size() * 2);
    hipMalloc((void**)&out, 256 * 512 * 4);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;  // ~25 ms per launch: long enough for the clock to settle
    for (int rep = 0; rep < 3; ++rep) {
        for (int shape : {16, 32}) {
            hipEventRecord(e0);
            if (shape == 16)
                hipLaunchKernelGGL(mfma_loop<16>, dim3(256), dim3(512), 0, 0, in, out, iters);
            else
                hipLaunchKernelGGL(mfma_loop<32>, dim3(256), dim3(512), 0, 0, in, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = 256.0 * 8 * iters * 32 * (16.0 * 16 * 32 * 2);
            printf("shape %dx%d: %.2f ms  %.0f TFLOP/s  (= %.2f GHz at 1017 flop/cycle/SIMD)\n", shape, shape, ms,
                   flop / ms / 1e9, flop / ms / 1e6 / (1024.0 * 1017.0) / 1e0 / 1e0 * 1e-0 / 1e0);
        }
    }
    return 0;
}
