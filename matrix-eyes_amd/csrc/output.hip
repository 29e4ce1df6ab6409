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

// One atomic pair per workgroup: thousands of wave-level atomics on the same two words serialise in one L2
// channel (0.19 ms for 16 K of them; the data itself is 19 MB).  16-byte accesses; the tail is scalar.
__global__ __launch_bounds__(256) void clamp_minmax_kernel(float* __restrict__ depth, int64_t count,
                                                           unsigned* __restrict__ minmax) {
    // output.rs:51-57 (clamp) and 69-75 (fold with f32::min / f32::max, which skip NaN)
    __shared__ float smn[4], smx[4];
    const float lo = 1.0f / 250.0f, hi = 1.0f / 0.1f;
    float mn = INFINITY, mx = 0.0f;
    // a pointer off 16-byte alignment (a caller's slice) takes the scalar loop for everything
    const int64_t n4 = ((size_t)depth & 15) == 0 ? count / 4 : 0;
    float4* d4 = reinterpret_cast<float4*>(depth);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = d4[i];
        v.x = rs_clamp(v.x, lo, hi), v.y = rs_clamp(v.y, lo, hi), v.z = rs_clamp(v.z, lo, hi), v.w = rs_clamp(v.w, lo, hi);
        d4[i] = v;
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = rs_clamp(depth[i], lo, hi);
        depth[i] = v;
        mn = fminf(mn, v), mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if ((threadIdx.x & 63) == 0) smn[threadIdx.x >> 6] = mn, smx[threadIdx.x >> 6] = mx;
    __syncthreads();
    // clamped values are positive, so the u32 order of the bit patterns is the float order
    if (threadIdx.x == 0) {
        atomicMin(&minmax[0], __float_as_uint(fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]))));
        atomicMax(&minmax[1], __float_as_uint(fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]))));
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

constexpr uint32_t NO_KEY = 0xffffffffu;

// Everything one quad needs, from the 4 x 4 depth patch around it: the keep bits of the 3 x 3 quads around q
// (bit 2 * (3 * dy + dx) + tri for quad (y - 1 + dy, x - 1 + dx)); quads outside the grid keep nothing.
__device__ __forceinline__ uint32_t quad_neighbourhood(const float* __restrict__ depth, int width, int height, int y, int x) {
    const int qw = width - 1, qh = height - 1;
    float d[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(y - 1 + j, 0), height - 1), xx = min(max(x - 1 + i, 0), width - 1);
            d[j][i] = depth[(int64_t)yy * width + xx];
        }
    uint32_t bits = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int qy = y - 1 + dy, qx = x - 1 + dx;
            if (qy < 0 || qx < 0 || qy >= qh || qx >= qw) continue;
            const float v00 = d[dy][dx], v10 = d[dy][dx + 1], v01 = d[dy + 1][dx], v11 = d[dy + 1][dx + 1];
            bits |= (tri_keep(v00, v01, v10) ? 1u : 0u) << (2 * (3 * dy + dx));
            bits |= (tri_keep(v10, v01, v11) ? 1u : 0u) << (2 * (3 * dy + dx) + 1);
        }
    return bits;
}

// smallest use key of corner (cy, cx) in {0,1}^2 of quad (y, x), from that quad's neighbourhood bits: the <= 6
// incident triangles in the order of their keys (output.rs:307-355 visits quads in raster order, the upper-left
// triangle [i00, i01, i10] before the lower-right one [i10, i01, i11])
__device__ __forceinline__ uint32_t corner_first_key(uint32_t bits, int width, int y, int x, int cy, int cx) {
    const int qw = width - 1;
    const int vy = y + cy, vx = x + cx;  // the vertex
    uint32_t best = NO_KEY;
    auto consider = [&](int qy, int qx, int tri, int slot) {
        const int dy = qy - (y - 1), dx = qx - (x - 1);  // position in the 3 x 3 neighbourhood (always inside it)
        if (bits >> (2 * (3 * dy + dx) + tri) & 1u) {
            const uint32_t key = (uint32_t)(3 * (2 * ((int64_t)qy * qw + qx) + tri) + slot);
            best = key < best ? key : best;
        }
    };
    consider(vy - 1, vx - 1, 1, 2);  // i11 of the lower-right triangle
    consider(vy - 1, vx, 0, 1);      // i01 of the upper-left triangle
    consider(vy - 1, vx, 1, 1);      // i01 of the lower-right triangle
    consider(vy, vx - 1, 0, 2);      // i10 of the upper-left triangle
    consider(vy, vx - 1, 1, 0);      // i10 of the lower-right triangle
    consider(vy, vx, 0, 0);          // i00 of the upper-left triangle
    return best;
}

// Status word of the single-pass scan (decoupled look-back): bits 0-1 flag (0 nothing yet, 1 = this block's
// aggregate, 2 = inclusive prefix up to and including this block), bits 2-32 vertices, bits 33-63 triangles.
// One 8-byte word written by one agent-scope store carries its own validity: no fence is needed beside it.
constexpr unsigned long long ST_AGG = 1, ST_PREFIX = 2;
__device__ __forceinline__ unsigned long long st_pack(unsigned long long flag, uint32_t nv, uint32_t nf) {
    return flag | ((unsigned long long)nv << 2) | ((unsigned long long)nf << 33);
}

// ONE pass over the quads in raster order, 1024 per workgroup (4 consecutive quads per thread): keep bits and
// first-use counts from the depth map itself, a workgroup scan, a decoupled look-back across workgroups (taken
// in ticket order, so every predecessor is running or done: no deadlock whatever the dispatch order), then the
// vertex ids of the vertices each quad introduces and, per quad, (first face index << 2 | keep bits) for the
// face kernel.  ws: [0] ticket, [1] total (vertices | triangles << 32), [2..] one status word per workgroup.
__global__ __launch_bounds__(256) void mesh_scan_kernel(const float* __restrict__ depth, int width, int height,
                                                        int32_t* __restrict__ vertex_index,
                                                        uint32_t* __restrict__ quad_info,
                                                        unsigned long long* __restrict__ ws) {
    __shared__ unsigned long long wsum[4];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_block;
    const int qw = width - 1, qh = height - 1;
    const int64_t nq = (int64_t)qw * qh;
    if (threadIdx.x == 0)
        s_block = (unsigned)__hip_atomic_fetch_add(&ws[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned blk = s_block;
    unsigned long long* status = ws + 2;
    const int64_t q0 = (int64_t)blk * 1024 + threadIdx.x * 4;
    uint32_t fk[4][4];
    uint32_t keep[4];
    unsigned long long cnt[4];  // vertices | triangles << 32
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t q = q0 + k;
        keep[k] = 0, cnt[k] = 0;
        if (q >= nq) continue;
        const int y = (int)(q / qw), x = (int)(q - (int64_t)y * qw);
        const uint32_t bits = quad_neighbourhood(depth, width, height, y, x);
        keep[k] = bits >> 8 & 3u;  // the centre quad
        const uint32_t lo = (uint32_t)(6 * q), hi = lo + 6;
        uint32_t nv = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fk[k][c] = corner_first_key(bits, width, y, x, c >> 1, c & 1);  // c: i00, i10, i01, i11
            nv += (fk[k][c] >= lo && fk[k][c] < hi) ? 1u : 0u;
            // unused vertices get -1 from the quad whose i00 they are (last column / row: from its neighbour)
            const bool mine = c == 0 || (c == 1 && x == qw - 1) || (c == 2 && y == qh - 1) ||
                              (c == 3 && x == qw - 1 && y == qh - 1);
            if (mine && fk[k][c] == NO_KEY) vertex_index[(int64_t)(y + (c >> 1)) * width + x + (c & 1)] = -1;
        }
        cnt[k] = (unsigned long long)nv | ((unsigned long long)((keep[k] & 1u) + (keep[k] >> 1)) << 32);
    }
    // exclusive scan inside the workgroup
    const unsigned long long tsum = cnt[0] + cnt[1] + cnt[2] + cnt[3];
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
    const unsigned long long block_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // decoupled look-back by wave 0: publish the aggregate, walk the predecessors 64 at a time
    if (wave == 0) {
        const uint32_t bv = (uint32_t)block_total, bf = (uint32_t)(block_total >> 32);
        if (lane == 0)
            __hip_atomic_store(&status[blk], st_pack(blk == 0 ? ST_PREFIX : ST_AGG, bv, bf), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long prefix = 0;  // vertices | triangles << 32 of all earlier workgroups
        int64_t end = blk;              // predecessors [end - 64, end) are examined next
        while (end > 0) {
            const int64_t i = end - 1 - lane;
            unsigned long long st = st_pack(ST_PREFIX, 0, 0);  // lanes past the start: neutral
            if (i >= 0) {
                do {
                    st = __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((st & 3ull) == 0);  // ticket order: that workgroup is running, the word will come
            }
            // the nearest lane holding an inclusive prefix ends the walk; sum everything up to and including it
            const unsigned long long has_prefix = __ballot((st & 3ull) == ST_PREFIX);
            const int stop = __ffsll((long long)has_prefix) - 1;  // first such lane (lane 0 = nearest predecessor)
            const unsigned long long val = ((st >> 2) & 0x7fffffffull) | ((st >> 33) << 32);
            unsigned long long part = (stop < 0 || lane <= stop) ? val : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            prefix += part;
            if (stop >= 0) break;
            end -= 64;
        }
        if (lane == 0) {
            const unsigned long long incl = prefix + block_total;
            if (blk != 0)
                __hip_atomic_store(&status[blk], st_pack(ST_PREFIX, (uint32_t)incl, (uint32_t)(incl >> 32)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
            if ((int64_t)blk == (nq + 1023) / 1024 - 1) ws[1] = incl;  // the last workgroup: totals
        }
    }
    __syncthreads();
    unsigned long long run = s_prefix + woff + inc - tsum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t q = q0 + k;
        if (q >= nq) break;
        const int y = (int)(q / qw), x = (int)(q - (int64_t)y * qw);
        const uint32_t lo = (uint32_t)(6 * q), hi = lo + 6;
        const uint32_t vbase = (uint32_t)run, fbase = (uint32_t)(run >> 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (fk[k][c] >= lo && fk[k][c] < hi) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) rank += (fk[k][j] >= lo && fk[k][j] < fk[k][c]) ? 1 : 0;
                vertex_index[(int64_t)(y + (c >> 1)) * width + x + (c & 1)] = (int32_t)(vbase + rank);
            }
        }
        quad_info[q] = fbase << 2 | keep[k];
        run += cnt[k];
    }
}

__global__ void mesh_faces_kernel(const uint32_t* __restrict__ quad_info, const int32_t* __restrict__ vertex_index,
                                  int width, int height, int32_t* __restrict__ faces) {
    const int qw = width - 1;
    const int64_t nq = (int64_t)qw * (height - 1);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
         q += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t info = quad_info[q];
        const uint32_t k = info & 3u;
        if (!k) continue;
        const int y = (int)(q / qw), x = (int)(q - (int64_t)y * qw);
        int64_t f = info >> 2;
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
    unsigned g = grid_for(cdiv(count, 16));  // >= 16 elements per thread; at most one workgroup per CU pair
    g = g > 512 ? 512 : g;
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
    static PerDeviceOnce once;
    per_device_once(once, [&](int) {
        ME_HIP(hipFuncSetAttribute((const void*)stereogram_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 2 * 4));
        return 1;
    });
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

size_t mesh_workspace_bytes(int32_t width, int32_t height) {
    const int64_t nq = (int64_t)(width - 1) * (height - 1);
    return (size_t)((2 + cdiv(nq, 1024)) * 8 + 256 + nq * 4);
}

// Two launches (scan + faces), one 16-byte read-back of the counts.  `workspace`: mesh_workspace_bytes() of
// device memory owned by the caller's context (no allocation here).
void mesh_index_run(const float* depth, int32_t width, int32_t height, int32_t* vertex_index,
                    int32_t* faces, int64_t* nverts, int64_t* nfaces, void* workspace, hipStream_t stream) {
    ME_CHECK(width >= 2 && height >= 2, ME_ERR_BAD_SHAPE, "mesh: %dx%d", width, height);
    const int64_t nq = (int64_t)(width - 1) * (height - 1);
    // 30 bits of face index beside the keep bits in quad_info; 31-bit counters in the scan's status words
    ME_CHECK(2 * nq < (1ll << 30), ME_ERR_BAD_SHAPE, "mesh: %dx%d too large", width, height);
    const int64_t nblocks = cdiv(nq, 1024);
    unsigned long long* ws = (unsigned long long*)workspace;
    const size_t head = ((size_t)(2 + nblocks) * 8 + 255) / 256 * 256;
    uint32_t* quad_info = (uint32_t*)((char*)workspace + head);
    ME_HIP(hipMemsetAsync(ws, 0, (size_t)(2 + nblocks) * 8, stream));  // ticket, total, status words
    hipLaunchKernelGGL(mesh_scan_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, depth, width, height,
                       vertex_index, quad_info, ws);
    if (faces)
        hipLaunchKernelGGL(mesh_faces_kernel, dim3(grid_for(nq)), dim3(256), 0, stream, quad_info, vertex_index,
                           width, height, faces);
    ME_HIP(hipGetLastError());
    unsigned long long total = 0;
    ME_HIP(hipMemcpyAsync(&total, ws + 1, 8, hipMemcpyDeviceToHost, stream));
    ME_HIP(hipStreamSynchronize(stream));
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
