// Diagnostic: what does the memory LAYOUT of a GEMM's operands cost the L2 -> LDS staging of a 256x256x64 slab?
// The staging pattern of gemm_pp_kernel (8 waves, 64 KiB per K slab by global_load_lds_dwordx4, two slabs in flight,
// the tile walk of tile_origin: an 8 x 4 patch of tiles per XCD) with NO MFMAs, on two layouts of the same operands:
//   row-major   [rows][K]           -- a slab of a tile is 256 segments of 128 B at a stride of 2K bytes
//   slab-major  [K/64][rows][64]    -- a slab of a tile is 32 KiB contiguous
// If address interleaving over the L2's channels makes the strided form collide, the slab-major form reads faster.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ingest_layout.hip -o gpurun_out/ingest_layout && gpurun_out/ingest_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

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

struct Prob {
    const char *A, *W;
    int M, N, K;
    int layout;  // 0 row-major, 1 slab-major, 2 slab-major with a 4 KiB-granular XOR of the slab index into the row block
};

template <int BM>
__device__ __forceinline__ void tile_origin(const Prob& p, int bid, int nwg, int& m0, int& n0) {
    constexpr int GM = 8, BN = 256;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_sr = GM * nbn;
    const int sr = t / per_sr;
    const int rem = t - sr * per_sr;
    const int h = min(GM, nbm - sr * GM);
    const int n = rem / h;
    const int rr = rem - n * h;
    m0 = (sr * GM + rr) * BM;
    n0 = n * BN;
}

// 512 threads; wave w stages rows [w*8 + 64*i, +8) of the 256 activation rows and the same of the weight rows per slab
template <int BM>
__global__ __launch_bounds__(512, 2) void ingest(const Prob p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = lane >> 3, sslot = lane & 7;
    const int nk = p.K / 64;
    const int ntiles = ((p.M + BM - 1) / BM) * ((p.N + 255) / 256);
    constexpr int A_IT = BM / 64;  // 8-row pieces per wave
    const unsigned smem_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
    int slot = 0;
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        int m0, n0;
        tile_origin<BM>(p, vb, ntiles, m0, n0);
        unsigned sa[4], sw[4];
        const int64_t rowb = p.layout == 0 ? (int64_t)p.K * 2 : 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 8 + wave) * 8 + srow;
            const int chunk = sslot ^ ((row >> 1) & 7);
            sw[i] = (unsigned)(row * rowb) + chunk * 16;
            sa[i] = (unsigned)((row % BM) * rowb) + chunk * 16;
        }
        const char* abase = p.A + (p.layout == 0 ? (int64_t)m0 * p.K * 2 : (int64_t)m0 * 128);
        const char* wbase = p.W + (p.layout == 0 ? (int64_t)n0 * p.K * 2 : (int64_t)n0 * 128);
        const int64_t a_slab = p.layout == 0 ? 128 : (int64_t)p.M * 128;
        const int64_t w_slab = p.layout == 0 ? 128 : (int64_t)p.N * 128;
        for (int kt = 0; kt < nk; ++kt) {
            const char* ab = uniform_ptr(abase + kt * a_slab);
            const char* wb = uniform_ptr(wbase + kt * w_slab);
            const unsigned dst = smem_base + slot * 65536 + wave * 1024;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) glds16_raw(ab, sa[i], dst + i * 8192);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_raw(wb, sw[i], dst + 32768 + i * 8192);
            if constexpr (A_IT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the previous slab has landed
            else if constexpr (A_IT == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
    const int shapes[4][3] = {{21760, 3072, 1024}, {21760, 4096, 1024}, {21760, 1024, 1024}, {21760, 1024, 4096}};
    const char* names[4] = {"qkv", "fc1", "proj", "fc2"};
    const size_t abytes = (size_t)21760 * 4096 * 2, wbytes = (size_t)4096 * 4096 * 2;
    char *A, *W;
    hipMalloc((void**)&A, abytes);
    hipMalloc((void**)&W, wbytes);
    hipMemset(A, 1, abytes);
    hipMemset(W, 2, wbytes);
    hipFuncSetAttribute((const void*)ingest<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)ingest<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)ingest<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int bms[3] = {256, 128, 64};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int s = 0; s < 4; ++s) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int layout = 0; layout < 2; ++layout) {
                Prob p{A, W, shapes[s][0], shapes[s][1], shapes[s][2], layout};
                const int bm = bms[rep % 3];
                const int ntiles = ((p.M + bm - 1) / bm) * ((p.N + 255) / 256);
                const int grid = ntiles < 256 ? ntiles - ntiles % 8 : 256;
                hipEventRecord(e0);
                for (int it = 0; it < 10; ++it) {
                    if (bm == 256) hipLaunchKernelGGL(ingest<256>, dim3(grid), dim3(512), 131072, 0, p);
                    else if (bm == 128) hipLaunchKernelGGL(ingest<128>, dim3(grid), dim3(512), 131072, 0, p);
                    else hipLaunchKernelGGL(ingest<64>, dim3(grid), dim3(512), 131072, 0, p);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 100.0;
                const double bytes = (double)ntiles * (p.K / 64) * (bm + 256) * 128.0;
                const int rounds = (ntiles + grid - 1) / grid;
                printf("%-4s M %d N %d K %d BM %3d %s: %8.1f us per launch, %6.2f TB/s staged, %5.1f GB/s per CU, %.3f us per slab (%d rounds)\n",
                       names[s], p.M, p.N, p.K, bm, layout == 0 ? "row-major " : "slab-major", us, bytes / us / 1e6,
                       bytes / us / 1e3 / 256, us / (rounds * (p.K / 64)), rounds);
            }
        }
    }
    return 0;
}
