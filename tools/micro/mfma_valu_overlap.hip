// Diagnostic: does a VALU stream of one wave run beside the MFMA stream of another wave on the same SIMD?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) issue independent 32x32x16 MFMAs, waves 4-7
// independent v_fma_f32.  Each role is timed alone and together (s_memtime ticks per instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o build_ab/mfma_valu_overlap && build_ab/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(1024) void overlap(float* out, unsigned long long* ticks, int iters, int mode, int valu_waves) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = wave < 4;
    float acc_out = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (mfma_role) {
        if (mode & 1) {
            f16x8 a, b;
            for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
            f32x16 c[4];
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int j = 0; j < 4; ++j) acc_out += c[j][0] + c[j][7];
        }
    } else if ((mode & 2) && wave < 4 + 4 * valu_waves) {
        float r[8];
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 0.001f + i * 0.01f;
        const float c = 0.999f, d = 0.001f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) acc_out += r[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc_out;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&ticks, 256 * 16 * 8);
    const int iters = 8192;
    for (int valu_waves = 1; valu_waves <= 3; valu_waves += 2)
        for (int mode = 1; mode <= 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(ticks, 0, 256 * 16 * 8);
                overlap<<<256, 256 + 256 * valu_waves>>>(out, ticks, iters, mode, valu_waves);
                hipDeviceSynchronize();
            }
            unsigned long long h[256 * 16];
            hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0, v = 0;
            int nv = 0;
            for (int b = 0; b < 256; ++b) {
                for (int w = 0; w < 4; ++w) m += (double)h[b * 16 + w];
                for (int w = 4; w < 4 + 4 * valu_waves; ++w) v += (double)h[b * 16 + w], ++nv;
            }
            printf("%d VALU wave(s)/SIMD, %s: MFMA %.2f ticks per 32x32x16, VALU %.2f ticks per v_fma_f32 (per wave)\n", valu_waves,
                   mode == 1 ? "MFMA alone " : (mode == 2 ? "VALU alone " : "both       "), m / (256.0 * 4 * iters * 4), nv ? v / (nv * iters * 8.0) : 0.0);
        }
    return 0;
}
