// Diagnostic: what slows the L2 -> LDS staging of a GEMM slab down when the CU also computes?
// One workgroup of 8 waves per CU.  Waves 0-3 stage 64 KiB per slab by LDS-DMA (L2-resident rows, two slabs in flight:
// alone 0.49 us per slab = 64 B/clk, tools/micro/dma_issue_waves.hip).  Waves 4-7 (one per SIMD) do, per slab, the
// LDS fragment reads and / or the MFMAs the 256x256x64 tile needs per SIMD (48 ds_read_b128 and 128 v_mfma 16x16x32).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_beside_compute.hip -o build_ab/dma_beside_compute
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_raw(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base)
                 : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

// DMA: stage or not; READS: ds_read_b128 per slab and compute wave; MFMAS: per slab and compute wave
template <int DMA, int READS, int MFMAS>
__global__ __launch_bounds__(512, 2) void k(const char* A, int K, int ntiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int srow = lane >> 3, sslot = lane & 7;
    const int nk = K / 64;
    const unsigned smem_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
    int slot = 0;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 fa[4], fb[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) fa[i][e] = (_Float16)(0.001f * (lane + i + e)), fb[i][e] = (_Float16)(0.002f * (lane * 3 + i - e));
    const int frow = lane & 15, fsl = ((lane >> 4) ^ (frow >> 1)) * 16;
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        const int m0 = ((vb >> 3) & 1) * 512;
        unsigned s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i * 4 + (wave & 3)) * 8 + srow;
            s[i] = (unsigned)(row * K * 2) + (sslot ^ ((row >> 1) & 7)) * 16;
        }
        const char* base = A + (int64_t)m0 * K * 2;
        for (int kt = 0; kt < nk; ++kt) {
            if (wave < 4) {
                if constexpr (DMA) {
                    const char* b = uniform_ptr(base + kt * 128);
                    const unsigned dst = smem_base + slot * 65536 + wave * 1024;
#pragma unroll
                    for (int i = 0; i < 16; ++i) glds16_raw(b, s[i], dst + i * 4096);
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                }
            } else {
                const char* src = smem + (slot ^ 1) * 65536 + ((wave & 3) * 64 + frow) * 128 + fsl;
                constexpr int STEPS = READS > MFMAS / 8 ? READS : MFMAS / 8;
#pragma unroll
                for (int i = 0; i < STEPS; ++i) {
                    if (i < READS) {
                        const f16x8 v = *reinterpret_cast<const f16x8*>(src + (i & 15) * 2048 + (i >> 4) * 16384);
                        if (MFMAS) fa[i & 3] = v; else asm volatile("" :: "v"(v));
                    }
                    if (i * 8 < MFMAS) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            acc[(i * 8 + j) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[j & 3], fb[(j >> 1) & 3], acc[(i * 8 + j) & 15], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][3];
    if (t == 12345.678f) sink[threadIdx.x] = t;
}

template <int DMA, int READS, int MFMAS>
void run(const char* A, float* sink, hipEvent_t e0, hipEvent_t e1, const char* what) {
    (void)hipFuncSetAttribute((const void*)k<DMA, READS, MFMAS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int ntiles = 2048, K = 1024;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<DMA, READS, MFMAS>), dim3(256), dim3(512), 131072, 0, A, K, ntiles, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 100.0, slabs = 8.0 * (K / 64);
        if (rep == 2) printf("%-58s %.3f us per slab\n", what, us / slabs);
    }
}

int main() {
    char* A;
    float* sink;
    (void)hipMalloc((void**)&A, (size_t)1024 * 1024 * 2);
    (void)hipMemset(A, 0x3c, (size_t)1024 * 1024 * 2);
    (void)hipMalloc((void**)&sink, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    run<1, 0, 0>(A, sink, e0, e1, "DMA 64 KiB alone");
    run<0, 48, 0>(A, sink, e0, e1, "48 ds_read_b128 per SIMD alone");
    run<0, 0, 128>(A, sink, e0, e1, "128 MFMA per SIMD alone");
    run<0, 48, 128>(A, sink, e0, e1, "reads + MFMA");
    run<1, 48, 0>(A, sink, e0, e1, "DMA beside 48 ds_read_b128 per SIMD");
    run<1, 0, 128>(A, sink, e0, e1, "DMA beside 128 MFMA per SIMD");
    run<1, 48, 128>(A, sink, e0, e1, "DMA beside reads + MFMA (the GEMM slab)");
    run<1, 32, 48>(A, sink, e0, e1, "DMA beside 32 reads + 48 MFMA (the 96x256 tile's share)");
    return 0;
}
