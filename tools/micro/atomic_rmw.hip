// Diagnostic: the residual update x += v of a 256x256 f32 tile per workgroup (what the proj / fc2 epilogue
// does, 256 workgroups at once) as (a) load + add + store of float4, (b) four global_atomic_add_f32 per
// float4 (no return value, same lane -> address map), (c) atomics with consecutive lanes on consecutive words.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_rmw.hip -o gpurun_out/atomic_rmw && gpurun_out/atomic_rmw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// tile t: rows [256 t', +256) x cols [256 c, +256) of a [M][1024] f32 matrix; 512 threads, 8 lanes x 16 B per
// 128-byte row segment, as in the GEMM epilogue (each wave: 64 cols x 128 rows)
template <int MODE>
__global__ __launch_bounds__(512) void rmw(float* __restrict__ x, int ldc, float v) {
    const int tile = blockIdx.x, tr = tile >> 2, tc = tile & 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 2, wn = wave & 3;
    float* base = x + (size_t)(tr * 256 + wm * 128) * ldc + tc * 256 + wn * 64;
    if (MODE == 2) {
        for (int r = 0; r < 128; ++r)  // one row of 64 floats per instruction
            __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(base + (size_t)r * ldc + lane), v);
        return;
    }
    const int r0 = lane >> 3, c0 = (lane & 7) * 4;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {  // 32 granules of 8 rows x (2 x 32 cols)
        const int r = (i >> 1) * 8 + r0, c = (i & 1) * 32 + c0;
        float* p = base + (size_t)r * ldc + c;
        if (MODE == 0) {
            float4 a = *reinterpret_cast<float4*>(p);
            a.x += v, a.y += v, a.z += v, a.w += v;
            *reinterpret_cast<float4*>(p) = a;
        } else {
            for (int e = 0; e < 4; ++e)
                __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(p + e), v);
        }
    }
}

int main() {
    const int M = 16384, N = 1024;  // 256 tiles
    float* x;
    hipMalloc(&x, (size_t)M * N * 4);
    hipMemset(x, 0, (size_t)M * N * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) {
                if (mode == 0) rmw<0><<<256, 512>>>(x, N, 1.0f);
                if (mode == 1) rmw<1><<<256, 512>>>(x, N, 1.0f);
                if (mode == 2) rmw<2><<<256, 512>>>(x, N, 1.0f);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.2f us per launch (%.2f TB/s read+write)\n", mode, ms / 20 * 1e3, 2.0 * M * N * 4 / (ms / 20 * 1e-3) / 1e12);
        }
    }
    float h[4];
    hipMemcpy(h, x, 16, hipMemcpyDeviceToHost);
    printf("x[0] = %g (expect 120)\n", h[0]);
    return 0;
}
