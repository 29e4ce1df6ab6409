// Diagnostic: cost of the GEMM epilogue's GELU (gemm_core.h gelu_erf4) by itself: shader clocks per 4 values on one
// SIMD with 1, 2 and 4 waves per SIMD, values in registers (128 per lane as in the 256x256 tile).
//   hipcc --offload-arch=gfx950 -O3 -I matrix-eyes_amd/csrc tools/micro/gelu_rate.hip -o build_ab/gelu_rate && build_ab/gelu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 gelu_erf4(f32x4 x) {
    auto clamp_abs = [](float v) { return __builtin_amdgcn_fmed3f(fabsf(v), 0.0f, 9.0f); };
    auto positive = [](float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 3.0e38f); };
    const f32x2 a0 = {clamp_abs(x[0]), clamp_abs(x[1])}, a1 = {clamp_abs(x[2]), clamp_abs(x[3])};
    const f32x2 p0 = {positive(x[0]), positive(x[1])}, p1 = {positive(x[2]), positive(x[3])};
    f32x2 q0 = a0 * 7.329674645e-08f - 1.913058668e-06f, q1 = a1 * 7.329674645e-08f - 1.913058668e-06f;
#define STEP(c) q0 = q0 * a0 + (c), q1 = q1 * a1 + (c)
    STEP(1.896382855e-05f); STEP(-6.020677392e-05f); STEP(-5.156729021e-04f); STEP(7.680844516e-03f);
    STEP(-5.303888768e-02f); STEP(-4.589743018e-01f); STEP(-1.151143670e+00f); STEP(-9.999989867e-01f);
#undef STEP
    const f32x2 h0 = {__builtin_amdgcn_exp2f(q0.x), __builtin_amdgcn_exp2f(q0.y)};
    const f32x2 h1 = {__builtin_amdgcn_exp2f(q1.x), __builtin_amdgcn_exp2f(q1.y)};
    const f32x2 r0 = p0 - h0 * a0, r1 = p1 - h1 * a1;
    return f32x4{r0.x, r0.y, r1.x, r1.y};
}

__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* ticks, int iters) {
    f32x4 v[32];
    for (int i = 0; i < 32; ++i) v[i] = f32x4{threadIdx.x * 0.01f + i, i * 0.1f - 1.f, -0.3f * i, 0.5f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = gelu_erf4(v[i]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 32; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&ticks, 256 * 16 * 8);
    const int iters = 64;
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            k<<<256, 256 * wps>>>(out, ticks, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h[256 * 16];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        double sum = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 4 * wps; ++w) sum += (double)h[b * 16 + w];
        const double per_wave = sum / (256.0 * 4 * wps);
        printf("%d wave(s)/SIMD: %.1f clocks per gelu_erf4 per wave, %.1f per gelu_erf4 on the SIMD (128 values per lane: %.0f clocks for the SIMD's waves)\n",
               wps, per_wave / (iters * 32.0), per_wave / (iters * 32.0) / wps, per_wave / iters);
    }
    return 0;
}
