// Probe (diagnostic, GPU box only): operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and
// e8m0 block scales, and whether the f16 MFMA keeps f16 subnormal operands.  Prints PASS / FAIL lines.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mx_mfma_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void mx_kernel(const uint8_t* a, const uint8_t* b, const int* sa, const int* sb, float* out, int opsel) {
    const int l = threadIdx.x;
    v8i av, bv;
    for (int i = 0; i < 8; ++i) {
        av[i] = ((const int*)a)[l * 8 + i];
        bv[i] = ((const int*)b)[l * 8 + i];
    }
    f32x4 c = {0, 0, 0, 0};
    if (opsel == 0)
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 0, sb[l]);
    else if (opsel == 1)
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 1, sa[l], 1, sb[l]);
    else if (opsel == 2)
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 2, sa[l], 2, sb[l]);
    else
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 3, sa[l], 3, sb[l]);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void f16_denorm_kernel(float* out) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)9.5367431640625e-07f;  // 2^-20: subnormal in f16
        b[i] = (_Float16)1024.0f;
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[l] = c[0];
}

static uint8_t e4m3(int v) {  // small integers, exact
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    int m = abs(v), e = 0;
    while ((1 << (e + 1)) <= m) ++e;              // m in [2^e, 2^(e+1))
    const int frac = ((m << 3) >> e) & 7;          // needs m representable: m * 8 / 2^e integer
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)frac;
}

int main() {
    // Layout found by mx_mfma_discover.hip: lane l = (r = l & 15, q = l >> 4), byte i of A <-> A[row r][k = 64*(i>>4) +
    // 16*q + (i&15)]; B the same with col r; the scale byte (selected by op_sel) of lane r + 16*kb scales (row r, K
    // block kb = k / 32); C: col = l&15, row = 4*(l>>4) + reg.
    std::vector<float> A(16 * 128), B(128 * 16);
    std::vector<uint8_t> a(64 * 32), b(64 * 32);
    std::vector<int> sa(64), sb(64);
    srand(7);
    static const int vals[] = {-4, -3, -2, -1, 0, 1, 2, 3, 4, 6, -6, 8};
    for (int m = 0; m < 16; ++m)
        for (int k = 0; k < 128; ++k) A[m * 128 + k] = (float)vals[rand() % 12];
    for (int k = 0; k < 128; ++k)
        for (int n = 0; n < 16; ++n) B[k * 16 + n] = (float)vals[rand() % 12];
    std::vector<int> ea(64), eb(64);
    for (int l = 0; l < 64; ++l) ea[l] = 127 + (rand() % 7) - 3, eb[l] = 127 + (rand() % 7) - 3;
    uint8_t *da, *db;
    int *dsa, *dsb;
    float* dout;
    hipMalloc(&da, 2048), hipMalloc(&db, 2048), hipMalloc(&dsa, 256), hipMalloc(&dsb, 256), hipMalloc(&dout, 1024);
    int fails = 0;
    for (int opsel = 0; opsel < 4; ++opsel) {
        for (int l = 0; l < 64; ++l) {
            for (int i = 0; i < 32; ++i) {
                a[l * 32 + i] = e4m3((int)A[(l & 15) * 128 + 64 * (i >> 4) + 16 * (l >> 4) + (i & 15)]);
                b[l * 32 + i] = e4m3((int)B[(64 * (i >> 4) + 16 * (l >> 4) + (i & 15)) * 16 + (l & 15)]);
            }
            // the selected byte carries the scale; the other bytes carry junk that must be ignored
            unsigned wa = 0x11223344u, wb = 0x55667788u;
            wa = (wa & ~(0xffu << (8 * opsel))) | ((unsigned)ea[l] << (8 * opsel));
            wb = (wb & ~(0xffu << (8 * opsel))) | ((unsigned)eb[l] << (8 * opsel));
            sa[l] = (int)wa, sb[l] = (int)wb;
        }
        hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout, opsel);
        std::vector<float> out(256);
        hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        double worst = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                double ref = 0;
                for (int kb = 0; kb < 4; ++kb) {
                    double s = 0;
                    for (int i = 0; i < 32; ++i) s += (double)A[row * 128 + 32 * kb + i] * B[(32 * kb + i) * 16 + col];
                    // scale of (row, kb) sits in lane row + 16 kb; of (col, kb) in lane col + 16 kb
                    ref += s * std::ldexp(1.0, ea[row + 16 * kb] - 127) * std::ldexp(1.0, eb[col + 16 * kb] - 127);
                }
                const double d = fabs(out[l * 4 + r] - ref);
                if (d > 1e-3 * (fabs(ref) + 1)) ++bad;
                if (d > worst) worst = d;
            }
        printf("mx 16x16x128 e4m3, discovered layout, opsel %d: %s (bad %d / 256, worst abs diff %g)\n", opsel,
               bad ? "FAIL" : "PASS", bad, worst);
        fails += bad != 0;
    }
    if (fails) {
        // discovery: one-hot A byte against all-ones B (row map), one-hot scale (block map)
        printf("discovery dump (lane byte -> rows hit with B = 1, unit scales):\n");
        for (int l = 0; l < 64; l += 1)
            for (int i = 0; i < 32; i += 8) {
                std::fill(a.begin(), a.end(), 0);
                std::fill(b.begin(), b.end(), 0x38);
                a[l * 32 + i] = 0x38;
                for (int j = 0; j < 64; ++j) sa[j] = sb[j] = 127;
                hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice);
                hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
                hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
                hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout, 0);
                std::vector<float> out(256);
                hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
                printf("A lane %d byte %d:", l, i);
                for (int j = 0; j < 256; ++j)
                    if (out[j] != 0) { printf(" first hit lane %d reg %d val %g", j / 4, j % 4, out[j]); break; }
                printf("\n");
            }
    }
    float* dd;
    hipMalloc(&dd, 256);
    hipLaunchKernelGGL(f16_denorm_kernel, dim3(1), dim3(64), 0, 0, dd);
    float h[64];
    hipMemcpy(h, dd, 256, hipMemcpyDeviceToHost);
    printf("f16 MFMA with subnormal operand 2^-20 x 1024 x K=32: got %g, exact %g -> %s\n", h[0], 32.0 * 1024 / 1048576.0,
           h[0] == 32.0f * 1024 / 1048576.0f ? "subnormals KEPT" : (h[0] == 0 ? "subnormals FLUSHED" : "other"));
    return 0;
}
