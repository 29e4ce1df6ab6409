// The text of an OBJ file produced on the GPU (reference src/output.rs:484-630 ObjWriter): every "vt", "v" and "f"
// line formatted by one thread -- coordinates as Rust's `{}` prints an f64 (shortest round-trip digits, positional
// notation: ryu_f64.h) -- and the lines packed back to back into ONE byte buffer in the reference's order, so that a
// 1536x1536 textured mesh (up to 450 MB of text) costs one D2H copy and one file write instead of 0.23 s of host
// formatting beside a 26 ms forward pass.
//
// Three launches per file:
//   measure: every line is formatted into an LDS slot and only its length kept; one byte count per workgroup;
//   scan:    exclusive prefix over the workgroups of all sections in file order (one workgroup);
//   write:   the lines are formatted again, a workgroup's lines are laid end to end with an LDS scan of their
//            lengths and copied out as aligned 16-byte chunks (byte stores only at a workgroup's ragged ends).
// Formatting twice costs ~200 integer instructions per number and saves a 200-byte slot per line in HBM.
#include "model.h"
#include "ryu_f64.h"

namespace me {

namespace {

constexpr int kThreads = 128;
// longest line of each section.  A number that came from an f32 (every coordinate; 1 - v is an exact f64 difference
// of one) prints in at most 64 characters: sign, "0.", 44 zeros, 17 digits; a colour c / 255.0 in 20.
constexpr int kSlotVt = 136;   // "vt " + 64 + ' ' + 64 + '\n'
constexpr int kSlotV = 264;    // "v " + 3 x (64 + ' ') + 3 x (20 + ' ') + '\n'
constexpr int kSlotF = 72;     // 'f' + 3 x (' ' + 10 + '/' + 10) + '\n'

struct ObjArgs {
    const float* uv;         // [count][2]  (vt)
    const float* xyz;        // [count][3]  (v)
    const uint8_t* colors;   // per vertex id [count][3], or null
    const int32_t* faces;    // [count][3]  (f)
    int64_t count;           // lines of this section
    int tex;                 // f: "a/a" pairs
};

typedef __attribute__((address_space(3))) char lds_char;

template <int SEC>
__device__ __forceinline__ int format_line(const ObjArgs& a, int64_t i, char* out) {
    int n = 0;
    if constexpr (SEC == 0) {  // output.rs:592-602: vt u (1 - v), both widened to f64 first
        out[0] = 'v', out[1] = 't', out[2] = ' ';
        n = 3;
        n += ryu::format_fixed((double)a.uv[2 * i], out + n);
        out[n++] = ' ';
        n += ryu::format_fixed(1.0 - (double)a.uv[2 * i + 1], out + n);
    } else if constexpr (SEC == 1) {  // output.rs:566-590: v x -y -z [r g b]
        out[0] = 'v', out[1] = ' ';
        n = 2;
        n += ryu::format_fixed((double)a.xyz[3 * i], out + n);
        out[n++] = ' ';
        n += ryu::format_fixed((double)(-a.xyz[3 * i + 1]), out + n);
        out[n++] = ' ';
        n += ryu::format_fixed((double)(-a.xyz[3 * i + 2]), out + n);
        if (a.colors) {
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {
                out[n++] = ' ';
                n += ryu::format_fixed((double)a.colors[3 * i + c] / 255.0, out + n);
            }
        }
    } else {  // output.rs:604-620: f a/a b/b c/c, 1-based
        out[0] = 'f';
        n = 1;
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const uint64_t idx = (uint64_t)a.faces[3 * i + k] + 1;
            out[n++] = ' ';
            n += ryu::format_u64(idx, out + n);
            if (a.tex) {
                out[n++] = '/';
                n += ryu::format_u64(idx, out + n);
            }
        }
    }
    out[n++] = '\n';
    return n;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// WRITE = false: block_bytes[first_block + blockIdx.x] = bytes of this workgroup's lines.
// WRITE = true: the lines, packed, at text + block_off[first_block + blockIdx.x].
template <int SEC, int SLOT, bool WRITE>
__global__ __launch_bounds__(kThreads) void obj_lines_kernel(ObjArgs a, int64_t first_block,
                                                              unsigned* __restrict__ block_bytes,
                                                              const unsigned long long* __restrict__ block_off,
                                                              char* __restrict__ text) {
    __shared__ __attribute__((aligned(16))) char slots[kThreads * SLOT];
    __shared__ int off[kThreads + 1];
    __shared__ int wave_tot[kThreads / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i = (int64_t)blockIdx.x * kThreads + t;
    int len = 0;
    if (i < a.count) len = format_line<SEC>(a, i, slots + t * SLOT);
    const int incl = wave_incl_scan(len, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    if constexpr (!WRITE) {
        if (t == kThreads - 1) block_bytes[first_block + blockIdx.x] = (unsigned)(base + incl);
    } else {
        off[t] = base + incl - len;
        if (t == kThreads - 1) off[kThreads] = base + incl;
        __syncthreads();
        const int total = off[kThreads];
        char* dst = text + block_off[first_block + blockIdx.x];
        const int mis = (int)((uintptr_t)dst & 15);
        const int nchunks = (mis + total + 15) >> 4;
        for (int c = t; c < nchunks; c += kThreads) {
            const int start = 16 * c - mis;  // first byte of the chunk, relative to this workgroup's text
            const int p0 = start < 0 ? 0 : start;
            // the line that holds byte p0: last index with off[] <= p0
            int lo = 0, hi = kThreads;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off[mid] <= p0) lo = mid;
                else hi = mid;
            }
            int line = lo;
            unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int p = start + k;
                if (p >= 0 && p < total) {
                    while (p >= off[line + 1]) ++line;
                    w[k >> 2] |= (unsigned)(unsigned char)slots[line * SLOT + (p - off[line])] << (8 * (k & 3));
                }
            }
            char* g = dst + start;
            if (start >= 0 && start + 16 <= total) {
                *reinterpret_cast<uint4*>(g) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {  // a ragged end: the bytes outside belong to a neighbouring workgroup (or to nobody)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int p = start + k;
                    if (p >= 0 && p < total) g[k] = (char)(w[k >> 2] >> (8 * (k & 3)));
                }
            }
        }
    }
}

// exclusive prefix of `n` byte counts -> 64-bit offsets from `base`; out_off[n] = the end
__global__ __launch_bounds__(1024) void obj_scan_kernel(const unsigned* __restrict__ bytes, int64_t n,
                                                         unsigned long long base,
                                                         unsigned long long* __restrict__ out_off) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = (int64_t)t * per, e = b + per < n ? b + per : n;
    unsigned long long s = 0;
    for (int64_t i = b; i < e; ++i) s += bytes[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
        const unsigned long long v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = base + part[t] - s;
    for (int64_t i = b; i < e; ++i) {
        out_off[i] = run;
        run += bytes[i];
    }
    if (t == 1023) out_off[n] = base + part[1023];
}

// colours by vertex id: the pixel that first used a vertex gives it its colour (output.rs:206-218, 578-587)
__global__ void obj_vertex_colors_kernel(const int32_t* __restrict__ vindex, const uint8_t* __restrict__ pixel_rgb,
                                         int64_t npix, uint8_t* __restrict__ vertex_rgb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = vindex[i];
        if (v >= 0) {
            vertex_rgb[3 * (int64_t)v] = pixel_rgb[3 * i];
            vertex_rgb[3 * (int64_t)v + 1] = pixel_rgb[3 * i + 1];
            vertex_rgb[3 * (int64_t)v + 2] = pixel_rgb[3 * i + 2];
        }
    }
}

// test surface (me_op_format_f64): value i printed into slot i of `stride` bytes, its length into lens[i]
__global__ void format_f64_kernel(const double* __restrict__ v, int64_t n, char* __restrict__ out, int stride,
                                  int* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lens[i] = ryu::format_fixed(v[i], out + i * stride);
}

template <int SEC, int SLOT, bool WRITE>
void launch_lines(const ObjArgs& a, int64_t first_block, unsigned* block_bytes, const unsigned long long* block_off,
                  char* text, hipStream_t s) {
    const int64_t blocks = cdiv(a.count, kThreads);
    if (blocks == 0) return;
    hipLaunchKernelGGL((obj_lines_kernel<SEC, SLOT, WRITE>), dim3((unsigned)blocks), dim3(kThreads), 0, s, a,
                       first_block, block_bytes, block_off, text);
    ME_HIP(hipGetLastError());
}

}  // namespace

void obj_vertex_colors_launch(const int32_t* vindex, const uint8_t* pixel_rgb, int64_t npix, uint8_t* vertex_rgb,
                              hipStream_t s) {
    const int64_t blocks = cdiv(npix, 256);
    hipLaunchKernelGGL(obj_vertex_colors_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, vindex,
                       pixel_rgb, npix, vertex_rgb);
    ME_HIP(hipGetLastError());
}

void format_f64_launch(const double* v, int64_t n, char* out, int stride, int* lens, hipStream_t s) {
    ME_CHECK(stride >= ryu::kMaxFixedChars, ME_ERR_BAD_ARG, "format_f64: slots of %d bytes (need %d)", stride,
             ryu::kMaxFixedChars);
    if (n <= 0) return;
    hipLaunchKernelGGL(format_f64_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, v, n, out, stride, lens);
    ME_HIP(hipGetLastError());
}

// Scratch for the byte counts and offsets of every workgroup of the three sections.
size_t obj_format_workspace_bytes(int64_t nverts, int64_t nfaces) {
    const int64_t blocks = 2 * cdiv(nverts, kThreads) + cdiv(nfaces, kThreads);
    return (size_t)(blocks * 4 + 256 + (blocks + 1) * 8 + 256);
}
// Two steps, so that the text buffer can be sized exactly.  obj_format_measure: the byte count of every workgroup
// and their offsets behind `header_bytes` bytes the caller fills in itself (mtllib / usemtl lines); returns the size
// of the whole text including the header (one stream synchronisation: the size comes back to the host).
// obj_format_write: the lines, into `text` (at least that many bytes; 16-byte aligned).
int64_t obj_format_measure(const float* uv, const float* xyz, const uint8_t* vertex_rgb, const int32_t* faces,
                           int64_t nverts, int64_t nfaces, bool tex, int64_t header_bytes, void* workspace,
                           hipStream_t s) {
    const int64_t bv = cdiv(nverts, kThreads), bf = cdiv(nfaces, kThreads);
    const int64_t first_v = tex ? bv : 0, first_f = first_v + bv, blocks = first_f + bf;
    unsigned* bytes = (unsigned*)workspace;
    unsigned long long* off = (unsigned long long*)((char*)workspace + ((size_t)blocks * 4 + 255) / 256 * 256);
    ObjArgs vt = {uv, nullptr, nullptr, nullptr, nverts, 0};
    ObjArgs v = {nullptr, xyz, vertex_rgb, nullptr, nverts, 0};
    ObjArgs f = {nullptr, nullptr, nullptr, faces, nfaces, tex ? 1 : 0};
    if (tex) launch_lines<0, kSlotVt, false>(vt, 0, bytes, nullptr, nullptr, s);
    launch_lines<1, kSlotV, false>(v, first_v, bytes, nullptr, nullptr, s);
    launch_lines<2, kSlotF, false>(f, first_f, bytes, nullptr, nullptr, s);
    hipLaunchKernelGGL(obj_scan_kernel, dim3(1), dim3(1024), 0, s, bytes, blocks, (unsigned long long)header_bytes, off);
    ME_HIP(hipGetLastError());
    unsigned long long total = 0;
    ME_HIP(hipMemcpyAsync(&total, off + blocks, 8, hipMemcpyDeviceToHost, s));
    ME_HIP(hipStreamSynchronize(s));
    return (int64_t)total;
}

void obj_format_write(const float* uv, const float* xyz, const uint8_t* vertex_rgb, const int32_t* faces, int64_t nverts,
                      int64_t nfaces, bool tex, char* text, void* workspace, hipStream_t s) {
    const int64_t bv = cdiv(nverts, kThreads), bf = cdiv(nfaces, kThreads);
    const int64_t first_v = tex ? bv : 0, first_f = first_v + bv, blocks = first_f + bf;
    const unsigned long long* off = (const unsigned long long*)((char*)workspace + ((size_t)blocks * 4 + 255) / 256 * 256);
    ObjArgs vt = {uv, nullptr, nullptr, nullptr, nverts, 0};
    ObjArgs v = {nullptr, xyz, vertex_rgb, nullptr, nverts, 0};
    ObjArgs f = {nullptr, nullptr, nullptr, faces, nfaces, tex ? 1 : 0};
    if (tex) launch_lines<0, kSlotVt, true>(vt, 0, nullptr, off, text, s);
    launch_lines<1, kSlotV, true>(v, first_v, nullptr, off, text, s);
    launch_lines<2, kSlotF, true>(f, first_f, nullptr, off, text, s);
}

}  // namespace me
