// Diagnostic: issue cost (shader clocks per wave-instruction on one SIMD) of v_exp_f32, v_rcp_f32, v_fma_f32,
// v_pk_fma_f32 and v_med3_f32, with 1 and with 4 waves per SIMD (s_memtime around an unrolled independent stream).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o build_ab/valu_rates && build_ab/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void rate(float* out, unsigned long long* cycles, int iters) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = (float)threadIdx.x * 0.001f + i * 0.01f;
    float c = 0.999f, d = 0.001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            if (OP == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
            if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
            if (OP == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
        }
        if (OP == 3) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2* p = reinterpret_cast<f2*>(r);
            f2 cc = {c, c}, dd = {d, d};
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(dd));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(dd));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4);
    hipMalloc(&cyc, 4096 * 8);
    const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_pk_fma_f32", "v_med3_f32"};
    const int iters = 4096;
    for (int waves = 1; waves <= 4; waves *= 4)
        for (int op = 0; op < 5; ++op) {
            const int threads = 256 * waves;  // 4 SIMDs x waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                if (op == 0) rate<0><<<256, threads>>>(out, cyc, iters);
                if (op == 1) rate<1><<<256, threads>>>(out, cyc, iters);
                if (op == 2) rate<2><<<256, threads>>>(out, cyc, iters);
                if (op == 3) rate<3><<<256, threads>>>(out, cyc, iters);
                if (op == 4) rate<4><<<256, threads>>>(out, cyc, iters);
                hipDeviceSynchronize();
            }
            unsigned long long h[256];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < 256; ++i) avg += (double)h[i];
            avg /= 256;
            // per SIMD: `waves` waves each issue 8 * iters instructions
            printf("%-13s %d wave(s)/SIMD: %.2f clocks per wave-instruction on the SIMD\n", names[op], waves,
                   avg / (8.0 * iters * waves));
        }
    return 0;
}
