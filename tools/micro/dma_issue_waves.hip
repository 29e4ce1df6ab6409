// Diagnostic: how does the L2 -> LDS staging rate of one CU depend on HOW MANY WAVES issue the LDS-DMA instructions?
// One workgroup per CU stages 64 KiB slabs (256 activation + 256 weight rows of 128 B, row stride 2K bytes, the
// gemm_pp_kernel pattern) with two slabs in flight; the same bytes are issued by 2, 4, 8 or 16 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_issue_waves.hip -o build_ab/dma_issue_waves && build_ab/dma_issue_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

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

// NW waves in the workgroup, the first NI of them issue; 64 pieces of 1 KiB per slab
template <int NW, int NI>
__global__ __launch_bounds__(NW * 64) void ingest(const char* A, int K, int ntiles, int rows_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int srow = lane >> 3, sslot = lane & 7;
    const int nk = K / 64;
    constexpr int PER = 64 / NI;  // pieces per issuing wave and slab
    const unsigned smem_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
    int slot = 0;
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        const int m0 = ((vb >> 3) & 1) * 512;  // 1024 rows in all: L2-resident in every XCD
        unsigned s[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int row = (i * NI + wave) * 8 + srow;
            s[i] = (unsigned)(row * K * 2) + (sslot ^ ((row >> 1) & 7)) * 16;
        }
        const char* base = A + (int64_t)m0 * K * 2;
        for (int kt = 0; kt < nk; ++kt) {
            if (wave < NI) {
                const char* b = uniform_ptr(base + kt * 128);
                const unsigned dst = smem_base + slot * 65536 + wave * 1024;
#pragma unroll
                for (int i = 0; i < PER; ++i) glds16_raw(b, s[i], dst + i * (NI * 1024));
                // the previous slab has landed
                if constexpr (PER == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else if constexpr (PER == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if constexpr (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NW, int NI>
void run(const char* A, int K, int rows, hipEvent_t e0, hipEvent_t e1) {
    (void)hipFuncSetAttribute((const void*)ingest<NW, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int ntiles = 1024;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((ingest<NW, NI>), dim3(256), dim3(NW * 64), 131072, 0, A, K, ntiles, rows);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 100.0, slabs = 4.0 * (K / 64);
        if (rep == 2)
            printf("K %d  waves in workgroup %2d, issuing %2d (%2d x 1 KiB each per slab): %.3f us per 64 KiB slab = %5.1f GB/s per CU\n", K, NW,
                   NI, 64 / NI, us / slabs, 65536.0 / (us / slabs) / 1e3);
    }
}

int main() {
    const int rows = 21760;
    char* A;
    (void)hipMalloc((void**)&A, (size_t)rows * 4096 * 2);
    (void)hipMemset(A, 1, (size_t)rows * 4096 * 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int K : {1024}) {
        run<8, 2>(A, K, rows, e0, e1);
        run<8, 4>(A, K, rows, e0, e1);
        run<8, 8>(A, K, rows, e0, e1);
        run<16, 16>(A, K, rows, e0, e1);
        run<4, 4>(A, K, rows, e0, e1);
        run<16, 8>(A, K, rows, e0, e1);
    }
    return 0;
}
