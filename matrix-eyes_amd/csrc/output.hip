// Output back end (reference src/output.rs): clamp + range, autostereogram, colour map, mesh
// indexing.  Integer/byte work and f32 index arithmetic — HBM-bound, no MFMA.
//
// Bit-exactness: the reference is safe Rust, which never contracts a*b+c into an FMA and uses
// IEEE division, round-half-away `f32::round` and saturating float->usize casts.
// Every f32 operation below that feeds an index or a byte is therefore written with the
// explicitly rounded intrinsics (__fmul_rn, __fadd_rn, __fsub_rn, __fdiv_rn) and the whole file is
// compiled with floating-point contraction off.
#include "common.h"
#include "../../include/me_viridis_lut.h"

// The rounded intrinsics are plain operators in the HIP headers, so contraction has to be switched
// off for this translation unit as well (the Makefile also passes -ffp-contract=off).
#pragma clang fp contract(off)

namespace me {

namespace {

__device__ __forceinline__ float rs_clamp(float v, float lo, float hi) {
    // f32::clamp: NaN stays NaN
    return v < lo ? lo : (v > hi ? hi : v);
}

// `x as usize` for f32: saturating, NaN -> 0 (here capped to int64 range, then to `cap`)
__device__ __forceinline__ int64_t rs_as_usize(float v) {
    if (!(v > 0.0f)) return 0;  // NaN, negatives, -0
    if (v >= 9.0e18f) return INT64_MAX;
    return (int64_t)v;
}

__global__ void clamp_minmax_kernel(float* __restrict__ depth, int64_t count,
                                    unsigned* __restrict__ minmax) {
    // output.rs:51-57 (clamp) and 69-75 (fold with f32::min / f32::max, which skip NaN)
    const float lo = 1.0f / 250.0f, hi = 1.0f / 0.1f;
    float mn = INFINITY, mx = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float v = rs_clamp(depth[i], lo, hi);
        depth[i] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    // clamped values are positive, so the u32 order of the bit patterns is the float order
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&minmax[0], __float_as_uint(mn));
        atomicMax(&minmax[1], __float_as_uint(mx));
    }
}

// output.rs:83-98 interpolate_point; data_width = dims[0] (rows), data_height = dims[1] (cols),
// depth_value(x, y) = data[data_height * y + x]  (replicated as written)
__device__ __forceinline__ float interpolate_point(const float* __restrict__ data, int data_width,
                                                   int data_height, float x, float y) {
    x = fmaxf(__fmul_rn(x, (float)data_width), 0.0f);
    y = fmaxf(__fmul_rn(y, (float)data_height), 0.0f);
    int64_t x0 = rs_as_usize(floorf(x)), y0 = rs_as_usize(floorf(y));
    x0 = x0 < data_width - 1 ? x0 : data_width - 1;
    y0 = y0 < data_height - 1 ? y0 : data_height - 1;
    const int64_t x1 = x0 + 1 < data_width - 1 ? x0 + 1 : data_width - 1;
    const int64_t y1 = y0 + 1 < data_height - 1 ? y0 + 1 : data_height - 1;
    const float fx = __fsub_rn(x, truncf(x)), fy = __fsub_rn(y, truncf(y));
    const float ox = __fsub_rn(1.0f, fx), oy = __fsub_rn(1.0f, fy);
    const float v00 = data[data_height * y0 + x0], v10 = data[data_height * y0 + x1];
    const float v01 = data[data_height * y1 + x0], v11 = data[data_height * y1 + x1];
    float acc = __fmul_rn(__fmul_rn(ox, oy), v00);
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(fx, oy), v10));
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ox, fy), v01));
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(fx, fy), v11));
    return acc;
}

// output.rs:141-193.  One workgroup per output row.  The row recurrence
//     out[x] = x >= P ? out[x + shift(x) - P] : noise[x % P]
// only ever points backwards, so it is resolved by pointer jumping in LDS (<= log2(W) rounds)
// followed by one gather from the noise row.  A source index >= x (possible only for degenerate
// amplitudes) reads the not-yet-overwritten noise, exactly like the sequential loop.
// range_dev: the depth range left on the device by clamp_minmax_kernel ({min, max}), read instead of the two
// scalars when not null -- DepthMap::new -> output_stereogram chained without a host round trip
__global__ __launch_bounds__(256) void stereogram_kernel(const float* __restrict__ depth, int rows,
                                                         int cols, float min_depth, float max_depth,
                                                         const float* __restrict__ range_dev,
                                                         int out_w, int out_h, float amplitude,
                                                         const uint8_t* __restrict__ noise,
                                                         uint8_t* __restrict__ out, int rounds) {
    extern __shared__ int lds[];
    if (range_dev) min_depth = range_dev[0], max_depth = range_dev[1];
    int* nxt = lds;            // [out_w] current ancestor
    int* term = lds + out_w;   // [out_w] noise index of a terminal pixel
    const int y = blockIdx.x;
    const float depth_multiplier = __fmul_rn((float)out_w, amplitude);
    const int64_t pattern_width =
        rs_as_usize(roundf(__fadd_rn(__fmul_rn(depth_multiplier, 2.0f), amplitude)));
    const float range = __fsub_rn(max_depth, min_depth);
    const float yn = __fdiv_rn((float)y, (float)out_h);
    for (int x = threadIdx.x; x < out_w; x += 256) {
        int nx = x, tv = x;
        if (x >= pattern_width) {
            float d = interpolate_point(depth, rows, cols, __fdiv_rn((float)x, (float)out_w), yn);
            d = __fdiv_rn(__fsub_rn(d, min_depth), range);
            const int64_t shift = rs_as_usize(roundf(__fmul_rn(d, depth_multiplier)));
            int64_t src = (int64_t)x + shift - pattern_width;
            if (src >= x) {
                // forward reference: the sequential loop reads the initial noise clone
                tv = (int)(src < out_w ? src : out_w - 1);
            } else {
                nx = (int)src;
            }
        } else {
            tv = (int)(pattern_width > 0 ? x % pattern_width : x);
        }
        nxt[x] = nx;
        term[x] = tv;
    }
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        // in-place jumping: a concurrently updated entry is still an ancestor
        for (int x = threadIdx.x; x < out_w; x += 256) nxt[x] = nxt[nxt[x]];
        __syncthreads();
    }
    const uint8_t* nrow = noise + (int64_t)y * out_w * 3;
    uint8_t* orow = out + (int64_t)y * out_w * 3;
    for (int x = threadIdx.x; x < out_w; x += 256) {
        const int s = term[nxt[x]];
        orow[3 * x + 0] = nrow[3 * s + 0];
        orow[3 * x + 1] = nrow[3 * s + 1];
        orow[3 * x + 2] = nrow[3 * s + 2];
    }
}

// output.rs:704-714 map_color
__device__ __forceinline__ uint8_t map_color(int channel, float value) {
    if (value >= 1.0f) return ME_VIRIDIS_REV[255][channel];
    const float step = 1.0f / 255.0f;
    int64_t box = rs_as_usize(floorf(__fdiv_rn(value, step)));
    box = box < 254 ? box : 254;
    const float ratio = __fdiv_rn(__fsub_rn(value, __fmul_rn(step, (float)box)), step);
    const float c1 = (float)ME_VIRIDIS_REV[box][channel];
    const float c2 = (float)ME_VIRIDIS_REV[box + 1][channel];
    const float v = roundf(__fadd_rn(__fmul_rn(c2, ratio), __fmul_rn(c1, __fsub_rn(1.0f, ratio))));
    // `as u8`: saturating, NaN -> 0
    return v > 255.0f ? 255 : (v > 0.0f ? (uint8_t)v : 0);
}

__global__ void depthmap_rgb_kernel(const float* __restrict__ depth, int64_t count, float min_depth,
                                    float max_depth, const float* __restrict__ range_dev,
                                    uint8_t* __restrict__ rgb) {
    // output.rs:128-131
    if (range_dev) min_depth = range_dev[0], max_depth = range_dev[1];
    const float range = __fsub_rn(max_depth, min_depth);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float t = __fdiv_rn(__fsub_rn(max_depth, depth[i]), range);
        rgb[3 * i + 0] = map_color(0, t);
        rgb[3 * i + 1] = map_color(1, t);
        rgb[3 * i + 2] = map_color(2, t);
    }
}

// ---------------------------------------------------------------------------------------
// Mesh indexing, output.rs:264-363.
//   triangle t = 2*quad + {0: [i00,i01,i10], 1: [i10,i01,i11]}, kept iff max/min <= 1.025
//   use key of (kept triangle t, slot s) = 3*t + s; a vertex id is the rank of the vertex's
//   smallest key ("first use while visiting kept triangles in raster order").
// Per vertex the smallest key comes from <= 6 incident triangles; per quad the number of
// vertices it introduces and its kept-triangle count are prefix-summed in raster order.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool tri_keep(float a, float b, float c) {
    // min_by / max_by with total_cmp, then `max / min <= POLYGON_DEPTH_THRESHOLD`
    // (total order == numeric order for the positive finite values DepthMap holds; for NaN the
    // comparison is false either way)
    const float mn = fminf(a, fminf(b, c)), mx = fmaxf(a, fmaxf(b, c));
    if (a != a || b != b || c != c) return false;
    return __fdiv_rn(mx, mn) <= 1.025f;
}

__global__ void mesh_keep_kernel(const float* __restrict__ depth, int width, int height,
                                 uint8_t* __restrict__ keep) {
    const int qw = width - 1;
    const int64_t nq = (int64_t)qw * (height - 1);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
         q += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(q / qw), x = (int)(q - (int64_t)y * qw);
        const float v00 = depth[(int64_t)y * width + x], v10 = depth[(int64_t)y * width + x + 1];
        const float v01 = depth[(int64_t)(y + 1) * width + x];
        const float v11 = depth[(int64_t)(y + 1) * width + x + 1];
        keep[q] = (tri_keep(v00, v01, v10) ? 1 : 0) | (tri_keep(v10, v01, v11) ? 2 : 0);
    }
}

constexpr uint32_t NO_KEY = 0xffffffffu;

// smallest use key of vertex (y, x)
__device__ __forceinline__ uint32_t vertex_first_key(const uint8_t* __restrict__ keep, int width,
                                                     int height, int y, int x) {
    const int qw = width - 1, qh = height - 1;
    uint32_t best = NO_KEY;
    auto consider = [&](int qy, int qx, int tri, int slot) {
        if (qy < 0 || qx < 0 || qy >= qh || qx >= qw) return;
        const int64_t q = (int64_t)qy * qw + qx;
        if (keep[q] & (1 << tri)) {
            const uint32_t key = (uint32_t)(3 * (2 * q + tri) + slot);
            best = key < best ? key : best;
        }
    };
    consider(y - 1, x - 1, 1, 2);  // i11 of the lower-right triangle
    consider(y - 1, x, 0, 1);      // i01 of the upper-left triangle
    consider(y - 1, x, 1, 1);      // i01 of the lower-right triangle
    consider(y, x - 1, 0, 2);      // i10 of the upper-left triangle
    consider(y, x - 1, 1, 0);      // i10 of the lower-right triangle
    consider(y, x, 0, 0);          // i00 of the upper-left triangle
    return best;
}

__global__ void mesh_first_key_kernel(const uint8_t* __restrict__ keep, int width, int height,
                                      uint32_t* __restrict__ first_key) {
    const int64_t nv = (int64_t)width * height;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv;
         v += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(v / width), x = (int)(v - (int64_t)y * width);
        first_key[v] = vertex_first_key(keep, width, height, y, x);
    }
}

// counts[q] = (kept triangles of q) << 32 | (vertices first used by q)
__global__ void mesh_count_kernel(const uint8_t* __restrict__ keep,
                                  const uint32_t* __restrict__ first_key, int width, int height,
                                  unsigned long long* __restrict__ counts) {
    const int qw = width - 1;
    const int64_t nq = (int64_t)qw * (height - 1);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
         q += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(q / qw), x = (int)(q - (int64_t)y * qw);
        const uint32_t lo = (uint32_t)(6 * q), hi = lo + 6;
        int nv = 0;
        const int64_t c[4] = {(int64_t)y * width + x, (int64_t)y * width + x + 1,
                              (int64_t)(y + 1) * width + x, (int64_t)(y + 1) * width + x + 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t fk = first_key[c[k]];
            nv += (fk >= lo && fk < hi) ? 1 : 0;
        }
        const int nf = (keep[q] & 1) + ((keep[q] >> 1) & 1);
        counts[q] = ((unsigned long long)nf << 32) | (unsigned)nv;
    }
}

// ---- exclusive scan of u64 (two packed u32 counters), 1024 elements per block ----
__global__ __launch_bounds__(256) void scan_block_kernel(unsigned long long* __restrict__ data,
                                                         int64_t n,
                                                         unsigned long long* __restrict__ block_sums) {
    __shared__ unsigned long long wsum[4];
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    unsigned long long v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = base + k < n ? data[base + k] : 0ull;
    const unsigned long long tsum = v[0] + v[1] + v[2] + v[3];
    unsigned long long inc = tsum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    unsigned long long run = woff + inc - tsum;  // exclusive prefix of this thread
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(256) void scan_sums_kernel(unsigned long long* __restrict__ sums,
                                                        int64_t nblocks,
                                                        unsigned long long* __restrict__ total) {
    // one workgroup, sequential over chunks of 256
    __shared__ unsigned long long carry_s;
    __shared__ unsigned long long wsum[4];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c = 0; c < nblocks; c += 256) {
        const int64_t i = c + threadIdx.x;
        const unsigned long long v = i < nblocks ? sums[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long woff = carry_s;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (i < nblocks) sums[i] = woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__global__ void mesh_assign_kernel(const uint8_t* __restrict__ keep,
                                   const uint32_t* __restrict__ first_key,
                                   const unsigned long long* __restrict__ scanned,
                                   const unsigned long long* __restrict__ block_sums, int width,
                                   int height, int32_t* __restrict__ vertex_index) {
    const int qw = width - 1;
    const int64_t nq = (int64_t)qw * (height - 1);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
         q += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(q / qw), x = (int)(q - (int64_t)y * qw);
        const uint32_t lo = (uint32_t)(6 * q), hi = lo + 6;
        const uint32_t vbase = (uint32_t)(scanned[q] + block_sums[q >> 10]);
        const int64_t c[4] = {(int64_t)y * width + x, (int64_t)y * width + x + 1,
                              (int64_t)(y + 1) * width + x, (int64_t)(y + 1) * width + x + 1};
        uint32_t fk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) fk[k] = first_key[c[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (fk[k] >= lo && fk[k] < hi) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) rank += (fk[j] >= lo && fk[j] < fk[k]) ? 1 : 0;
                vertex_index[c[k]] = (int32_t)(vbase + rank);
            }
        }
    }
}

__global__ void mesh_unused_kernel(const uint32_t* __restrict__ first_key, int64_t nv,
                                   int32_t* __restrict__ vertex_index) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv;
         v += (int64_t)gridDim.x * blockDim.x)
        if (first_key[v] == NO_KEY) vertex_index[v] = -1;
}

__global__ void mesh_faces_kernel(const uint8_t* __restrict__ keep,
                                  const unsigned long long* __restrict__ scanned,
                                  const unsigned long long* __restrict__ block_sums,
                                  const int32_t* __restrict__ vertex_index, int width, int height,
                                  int32_t* __restrict__ faces) {
    const int qw = width - 1;
    const int64_t nq = (int64_t)qw * (height - 1);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
         q += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t k = keep[q];
        if (!k) continue;
        const int y = (int)(q / qw), x = (int)(q - (int64_t)y * qw);
        int64_t f = (int64_t)((scanned[q] + block_sums[q >> 10]) >> 32);
        const int32_t i00 = vertex_index[(int64_t)y * width + x];
        const int32_t i10 = vertex_index[(int64_t)y * width + x + 1];
        const int32_t i01 = vertex_index[(int64_t)(y + 1) * width + x];
        const int32_t i11 = vertex_index[(int64_t)(y + 1) * width + x + 1];
        if (k & 1) {
            faces[3 * f + 0] = i00, faces[3 * f + 1] = i01, faces[3 * f + 2] = i10;
            ++f;
        }
        if (k & 2) faces[3 * f + 0] = i10, faces[3 * f + 1] = i01, faces[3 * f + 2] = i11;
    }
}

// output.rs:228-249
__global__ void mesh_vertices_kernel(const float* __restrict__ depth, int width, int height,
                                     const int32_t* __restrict__ vertex_index, float xm, float ym,
                                     float* __restrict__ uv, float* __restrict__ xyz) {
    const int64_t nv = (int64_t)width * height;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = vertex_index[i];
        if (id < 0) continue;
        const int64_t x_image = i % width, y_image = i / width;
        const float xn = __fdiv_rn((float)x_image, (float)width);
        const float yn = __fdiv_rn((float)y_image, (float)height);
        if (uv) uv[2 * (int64_t)id] = xn, uv[2 * (int64_t)id + 1] = yn;
        if (xyz) {
            const float z = __fdiv_rn(1.0f, depth[i]);
            xyz[3 * (int64_t)id + 0] = __fmul_rn(__fmul_rn(xm, __fsub_rn(xn, 0.5f)), z);
            xyz[3 * (int64_t)id + 1] = __fmul_rn(__fmul_rn(ym, __fsub_rn(yn, 0.5f)), z);
            xyz[3 * (int64_t)id + 2] = z;
        }
    }
}

inline unsigned grid_for(int64_t total) {
    const int64_t g = cdiv(total, 256);
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

void depth_clamp_minmax_launch(float* depth, int64_t count, float* minmax_dev, hipStream_t stream) {
    // {+inf, 0} as bit patterns, set on the stream without host memory (graph-capturable, no staging copy)
    ME_HIP(hipMemsetD32Async((hipDeviceptr_t)minmax_dev, 0x7f800000, 1, stream));
    ME_HIP(hipMemsetD32Async((hipDeviceptr_t)(minmax_dev + 1), 0, 1, stream));
    unsigned g = grid_for(count);
    g = g > 2048 ? 2048 : g;
    hipLaunchKernelGGL(clamp_minmax_kernel, dim3(g), dim3(256), 0, stream, depth, count,
                       (unsigned*)minmax_dev);
    ME_HIP(hipGetLastError());
}

void stereogram_launch(const float* depth, int32_t rows, int32_t cols, float min_depth,
                       float max_depth, const float* range_dev, int32_t out_w, int32_t out_h, float amplitude,
                       const uint8_t* noise, uint8_t* out, hipStream_t stream) {
    ME_CHECK(out_w > 0 && out_h > 0 && rows > 0 && cols > 0, ME_ERR_BAD_SHAPE,
             "stereogram: %dx%d from %dx%d", out_w, out_h, rows, cols);
    ME_CHECK(out_w <= 16384, ME_ERR_BAD_SHAPE, "stereogram: width %d > 16384", out_w);
    int rounds = 1;
    while ((1 << rounds) < out_w) ++rounds;
    const size_t lds = (size_t)out_w * 2 * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        ME_HIP(hipFuncSetAttribute((const void*)stereogram_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 2 * 4));
        attr_set = true;
    }
    hipLaunchKernelGGL(stereogram_kernel, dim3(out_h), dim3(256), lds, stream, depth, rows, cols,
                       min_depth, max_depth, range_dev, out_w, out_h, amplitude, noise, out, rounds);
    ME_HIP(hipGetLastError());
}

void depthmap_rgb_launch(const float* depth, int64_t count, float min_depth, float max_depth,
                         const float* range_dev, uint8_t* rgb, hipStream_t stream) {
    hipLaunchKernelGGL(depthmap_rgb_kernel, dim3(grid_for(count)), dim3(256), 0, stream, depth,
                       count, min_depth, max_depth, range_dev, rgb);
    ME_HIP(hipGetLastError());
}

void mesh_index_run(const float* depth, int32_t width, int32_t height, int32_t* vertex_index,
                    int32_t* faces, int64_t* nverts, int64_t* nfaces, hipStream_t stream) {
    ME_CHECK(width >= 2 && height >= 2, ME_ERR_BAD_SHAPE, "mesh: %dx%d", width, height);
    const int64_t nv = (int64_t)width * height;
    const int64_t nq = (int64_t)(width - 1) * (height - 1);
    ME_CHECK(6 * nq < 0xffffffffll, ME_ERR_BAD_SHAPE, "mesh: %dx%d too large", width, height);
    const int64_t nblocks = cdiv(nq, 1024);
    uint8_t* keep = nullptr;
    uint32_t* first_key = nullptr;
    unsigned long long *counts = nullptr, *sums = nullptr;
    // one-shot workspace: this is a per-image call outside the inference loop
    ME_HIP(hipMalloc(&keep, nq));
    ME_HIP(hipMalloc(&first_key, nv * 4));
    ME_HIP(hipMalloc(&counts, nq * 8));
    ME_HIP(hipMalloc(&sums, (nblocks + 1) * 8));
    hipLaunchKernelGGL(mesh_keep_kernel, dim3(grid_for(nq)), dim3(256), 0, stream, depth, width,
                       height, keep);
    hipLaunchKernelGGL(mesh_first_key_kernel, dim3(grid_for(nv)), dim3(256), 0, stream, keep, width,
                       height, first_key);
    hipLaunchKernelGGL(mesh_count_kernel, dim3(grid_for(nq)), dim3(256), 0, stream, keep, first_key,
                       width, height, counts);
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, counts, nq,
                       sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, stream, sums, nblocks,
                       sums + nblocks);
    hipLaunchKernelGGL(mesh_unused_kernel, dim3(grid_for(nv)), dim3(256), 0, stream, first_key, nv,
                       vertex_index);
    hipLaunchKernelGGL(mesh_assign_kernel, dim3(grid_for(nq)), dim3(256), 0, stream, keep, first_key,
                       counts, sums, width, height, vertex_index);
    if (faces)
        hipLaunchKernelGGL(mesh_faces_kernel, dim3(grid_for(nq)), dim3(256), 0, stream, keep, counts,
                           sums, vertex_index, width, height, faces);
    hipError_t e = hipGetLastError();
    unsigned long long total = 0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(&total, sums + nblocks, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(keep);
    (void)hipFree(first_key);
    (void)hipFree(counts);
    (void)hipFree(sums);
    ME_HIP(e);
    *nverts = (int64_t)(total & 0xffffffffull);
    *nfaces = (int64_t)(total >> 32);
}

void mesh_vertices_launch(const float* depth, int32_t width, int32_t height,
                          const int32_t* vertex_index, float xm, float ym, float* uv, float* xyz,
                          hipStream_t stream) {
    hipLaunchKernelGGL(mesh_vertices_kernel, dim3(grid_for((int64_t)width * height)), dim3(256), 0,
                       stream, depth, width, height, vertex_index, xm, ym, uv, xyz);
    ME_HIP(hipGetLastError());
}

}  // namespace me
