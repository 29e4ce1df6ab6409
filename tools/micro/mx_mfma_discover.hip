// Discovery of the v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3) operand layout on gfx950: which (lane, byte) of A meets
// which (lane, byte) of B (same k), and which lanes' scale bytes apply to which (row, k) elements.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Case {
    uint8_t a[2048], b[2048];
    int sa[64], sb[64];
};
__global__ void run(const Case* cs, float* out) {
    const Case& c = cs[blockIdx.x];
    const int l = threadIdx.x;
    v8i av, bv;
    for (int i = 0; i < 8; ++i) av[i] = ((const int*)c.a)[l * 8 + i], bv[i] = ((const int*)c.b)[l * 8 + i];
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, c.sa[l], 0, c.sb[l]);
    for (int r = 0; r < 4; ++r) out[blockIdx.x * 256 + l * 4 + r] = acc[r];
}
int main() {
    std::vector<Case> cases;
    auto blank = [] {
        Case c;
        memset(&c, 0, sizeof c);
        for (int i = 0; i < 64; ++i) c.sa[i] = c.sb[i] = 127;
        return c;
    };
    // 1) k matching: A one-hot at (lane 16qa, byte ia) [row 0], B = 1 on all bytes of lanes with col 0 except
    //    we want the exact partner: B one-hot at (lane 16qb, byte ib) [col 0].
    for (int qa = 0; qa < 4; ++qa)
        for (int ia = 0; ia < 32; ++ia)
            for (int qb = 0; qb < 4; ++qb)
                for (int ib = 0; ib < 32; ++ib) {
                    Case c = blank();
                    c.a[(16 * qa) * 32 + ia] = 0x38;
                    c.b[(16 * qb) * 32 + ib] = 0x38;
                    cases.push_back(c);
                }
    const size_t n1 = cases.size();
    // 2) scales: A = 1 everywhere, B = 1 only in lane group qb (all its lanes, all bytes); A-scale lane ls doubled
    for (int ls = 0; ls < 64; ++ls)
        for (int qb = 0; qb < 4; ++qb) {
            Case c = blank();
            memset(c.a, 0x38, 2048);
            for (int l = 16 * qb; l < 16 * qb + 16; ++l) memset(c.b + l * 32, 0x38, 32);
            c.sa[ls] = 128;
            cases.push_back(c);
        }
    const size_t n2 = cases.size() - n1;
    // 3) B-scale: B = 1 everywhere, A = 1 only in lane group qa; B-scale lane ls doubled
    for (int ls = 0; ls < 64; ++ls)
        for (int qa = 0; qa < 4; ++qa) {
            Case c = blank();
            memset(c.b, 0x38, 2048);
            for (int l = 16 * qa; l < 16 * qa + 16; ++l) memset(c.a + l * 32, 0x38, 32);
            c.sb[ls] = 128;
            cases.push_back(c);
        }
    Case* dc;
    float* dout;
    hipMalloc(&dc, cases.size() * sizeof(Case));
    hipMalloc(&dout, cases.size() * 1024);
    hipMemcpy(dc, cases.data(), cases.size() * sizeof(Case), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(run, dim3((unsigned)cases.size()), dim3(64), 0, 0, dc, dout);
    std::vector<float> out(cases.size() * 256);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    printf("k partners: A(lane-group qa, byte ia) meets B(qb, ib)\n");
    size_t ci = 0;
    for (int qa = 0; qa < 4; ++qa)
        for (int ia = 0; ia < 32; ++ia) {
            printf("A q%d b%2d ->", qa, ia);
            for (int qb = 0; qb < 4; ++qb)
                for (int ib = 0; ib < 32; ++ib, ++ci)
                    if (out[ci * 256] != 0) printf(" B q%d b%2d (%g)", qb, ib, out[ci * 256]);
            printf("\n");
        }
    printf("A-scale lane ls doubled, B nonzero in lane group qb: rows whose C[.][col 0] changed from 32\n");
    for (int ls = 0; ls < 64; ++ls)
        for (int qb = 0; qb < 4; ++qb, ++ci) {
            printf("sa lane %2d, B group %d:", ls, qb);
            for (int row = 0; row < 16; ++row) {
                const float v = out[ci * 256 + ((row >> 2) * 16) * 4 + (row & 3)];  // lane 16*(row/4), reg row%4 = col 0
                if (v != 32.f) printf(" row %d = %g", row, v);
            }
            printf("\n");
        }
    printf("B-scale lane ls doubled, A nonzero in lane group qa: cols whose C[row 0][.] changed from 32\n");
    for (int ls = 0; ls < 64; ++ls)
        for (int qa = 0; qa < 4; ++qa, ++ci) {
            printf("sb lane %2d, A group %d:", ls, qa);
            for (int col = 0; col < 16; ++col) {
                const float v = out[ci * 256 + col * 4 + 0];  // lane col, reg 0 = row 0
                if (v != 32.f) printf(" col %d = %g", col, v);
            }
            printf("\n");
        }
    (void)n2;
    return 0;
}
