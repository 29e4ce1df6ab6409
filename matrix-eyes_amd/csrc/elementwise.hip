// Row-wise and layout kernels around the GEMMs: all HBM-bound byte movers, 16-byte accesses,
// no LDS reuse except the two transposes.
#include "common.h"

namespace me {

namespace {

template <typename T>
struct Vec16 {
    typedef T v8 __attribute__((ext_vector_type(8)));
    typedef T v4 __attribute__((ext_vector_type(4)));
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------
// LayerNorm (burn nn::LayerNorm: biased variance, (x - mean) / sqrt(var + eps) * gamma + beta;
// reference call sites vit.rs:165,168,343).  One wave per row; f32 statistics.
// ---------------------------------------------------------------------------------------
template <typename T, int NPL>  // NPL = dim / 64 elements per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        T* __restrict__ y16, float* __restrict__ y32,
                                                        int64_t rows, float eps, RowSegs segs,
                                                        unsigned* __restrict__ status) {
    constexpr int DIM = NPL * 64;
    // elements per lane per pass: 8 where the row allows it, so that the 16-bit result leaves as ONE 16-byte
    // store per lane (8-byte stores run at less than half the rate: the GEMM epilogue's finding, gemm_core.h)
    constexpr int V = (NPL % 8 == 0) ? 8 : ((NPL % 4 == 0) ? 4 : 1);
    constexpr int NA = NPL / V;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (segs.seg1 && row >= segs.seg1) {  // wave-uniform: a wave owns one row
        const bool third = segs.seg2 && row >= segs.seg2;
        w = third ? segs.w2 : segs.w1;
        b = third ? segs.b2 : segs.b1;
    }
    const float* xr = x + row * DIM;
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = (i * 64 + lane) * V;
        if constexpr (V >= 4) {
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const float4 t = *reinterpret_cast<const float4*>(xr + e + 4 * h);
                v[i * V + 4 * h + 0] = t.x, v[i * V + 4 * h + 1] = t.y, v[i * V + 4 * h + 2] = t.z,
                                  v[i * V + 4 * h + 3] = t.w;
            }
        } else {
            v[i] = xr[e];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / DIM);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    const float var = wave_sum(q) * (1.0f / DIM);
    const float rstd = 1.0f / sqrtf(var + eps);
    float amax16 = 0.f;  // overflow guard of the 16-bit copy (common.h)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = (i * 64 + lane) * V;
        if constexpr (V >= 4) {
            float o[V];
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const float4 wv = *reinterpret_cast<const float4*>(w + e + 4 * h);
                const float4 bv = *reinterpret_cast<const float4*>(b + e + 4 * h);
                o[4 * h + 0] = (v[i * V + 4 * h + 0] - mean) * rstd * wv.x + bv.x;
                o[4 * h + 1] = (v[i * V + 4 * h + 1] - mean) * rstd * wv.y + bv.y;
                o[4 * h + 2] = (v[i * V + 4 * h + 2] - mean) * rstd * wv.z + bv.z;
                o[4 * h + 3] = (v[i * V + 4 * h + 3] - mean) * rstd * wv.w + bv.w;
            }
            if (y16) {
                typedef T vec __attribute__((ext_vector_type(V)));
                vec ov;
#pragma unroll
                for (int j = 0; j < V; ++j) ov[j] = (T)o[j], amax16 = fmaxf(amax16, fabsf(o[j]));
                *reinterpret_cast<vec*>(y16 + row * DIM + e) = ov;
            }
            if (y32) {
#pragma unroll
                for (int h = 0; h < V / 4; ++h)
                    *reinterpret_cast<float4*>(y32 + row * DIM + e + 4 * h) =
                        make_float4(o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]);
            }
        } else {
            const float o0 = (v[i] - mean) * rstd * w[e] + b[e];
            if (y16) y16[row * DIM + e] = (T)o0, amax16 = fmaxf(amax16, fabsf(o0));
            if (y32) y32[row * DIM + e] = o0;
        }
    }
    raise_overflow16<T>(status, amax16);
}

template <typename T>
void layernorm_typed(const float* x, const float* w, const float* b, void* y16, float* y32,
                     int64_t rows, int32_t dim, float eps, hipStream_t stream, const RowSegs& segs) {
    const dim3 grid((unsigned)cdiv(rows, 4)), block(256);
#define ME_LN(NPL)                                                                          \
    hipLaunchKernelGGL((layernorm_kernel<T, NPL>), grid, block, 0, stream, x, w, b, (T*)y16, y32, \
                       rows, eps, segs, current_status_word())
    switch (dim / 64) {
        case 1: ME_LN(1); break;
        case 2: ME_LN(2); break;
        case 4: ME_LN(4); break;
        case 8: ME_LN(8); break;
        case 16: ME_LN(16); break;
        default: fail(ME_ERR_BAD_SHAPE, "layernorm: dim %d not in {64,128,256,512,1024}", dim);
    }
#undef ME_LN
}

// ---------------------------------------------------------------------------------------
#pragma clang fp contract(off)   // reconstruction.rs:116-124 reproduced operation for operation
__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ img,
                                     int64_t pixels_per_image, int64_t total_pixels) {
    // reconstruction.rs:116-124: (x / 255 - 0.5) / 0.5 in f32, HWC -> CHW
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total_pixels;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / pixels_per_image, pp = p - b * pixels_per_image;
        const uint8_t* s = rgb + p * 3;
        float* d = img + b * 3 * pixels_per_image + pp;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = __fdiv_rn((float)s[c], 255.0f);
            d[c * pixels_per_image] = __fdiv_rn(v - 0.5f, 0.5f);
        }
    }
}

template <typename T>
__global__ void cast_to16_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n4,
                                 unsigned* __restrict__ status) {
    float amax16 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        typename Vec16<T>::v4 o;
        o[0] = (T)v.x, o[1] = (T)v.y, o[2] = (T)v.z, o[3] = (T)v.w;
        amax16 = fmaxf(fmaxf(fmaxf(amax16, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
        reinterpret_cast<typename Vec16<T>::v4*>(dst)[i] = o;
    }
    raise_overflow16<T>(status, amax16);
}
template <typename T>
__global__ void cast_to32_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const typename Vec16<T>::v4 v = reinterpret_cast<const typename Vec16<T>::v4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] =
            make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
}

// ---------------------------------------------------------------------------------------
// encoder.rs:125-140 / fov.rs:53: bilinear resample of f32 planes, rounded once to 16 bit.
// align_corners = 1: src = dst * (in - 1) / (out - 1)   (Burn's historical bilinear)
// align_corners = 0: src = max((dst + 0.5) * in / out - 0.5, 0)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void bilinear_kernel(const float* __restrict__ src, T* __restrict__ dst, int planes,
                                int in_size, int out_size, int align_corners) {
    const int64_t total = (int64_t)planes * out_size * out_size;
    const float ratio = align_corners
                            ? (float)((double)(in_size - 1) / (double)(out_size > 1 ? out_size - 1 : 1))
                            : (float)((double)in_size / (double)out_size);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % out_size);
        const int64_t t = i / out_size;
        const int y = (int)(t % out_size);
        const int64_t pl = t / out_size;
        float fy, fx;
        if (align_corners) {
            fy = ratio * (float)y;
            fx = ratio * (float)x;
        } else {
            fy = fmaxf(((float)y + 0.5f) * ratio - 0.5f, 0.f);
            fx = fmaxf(((float)x + 0.5f) * ratio - 0.5f, 0.f);
        }
        int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        y0 = y0 < in_size - 1 ? y0 : in_size - 1;
        x0 = x0 < in_size - 1 ? x0 : in_size - 1;
        const int y1 = y0 + 1 < in_size ? y0 + 1 : in_size - 1;
        const int x1 = x0 + 1 < in_size ? x0 + 1 : in_size - 1;
        const float wy = fy - (float)y0, wx = fx - (float)x0;
        const float* s = src + pl * in_size * in_size;
        const float a = s[(int64_t)y0 * in_size + x0], b = s[(int64_t)y0 * in_size + x1];
        const float c = s[(int64_t)y1 * in_size + x0], d = s[(int64_t)y1 * in_size + x1];
        const float v = a * (1.f - wx) * (1.f - wy) + b * wx * (1.f - wy) + c * (1.f - wx) * wy +
                        d * wx * wy;
        dst[i] = (T)v;
    }
}

// ---------------------------------------------------------------------------------------
// encoder.rs:142-156 split + vit.rs:210-223 patch-embed im2col, no window copy materialised:
// patches[((b*35 + win)*g*g + py*g + px)][c*256 + iy*16 + ix]
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void patchify_kernel(const T* __restrict__ x0, const T* __restrict__ x1,
                                const T* __restrict__ x2, T* __restrict__ patches, int batch,
                                int grid) {
    const int win_px = grid * 16;
    const int P = grid * grid;
    const int64_t total = (int64_t)batch * 35 * P * 96;  // 16-byte chunks
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 96);
        const int64_t row = i / 96;
        const int patch = (int)(row % P);
        const int64_t wi = row / P;
        const int win = (int)(wi % 35), b = (int)(wi / 35);
        const int c = ch >> 5, iy = (ch >> 1) & 15, ixh = ch & 1;
        const int py = patch / grid, px = patch - py * grid;
        const T* img;
        int S, wy, wx;
        if (win < 25) {
            img = x0, S = 4 * win_px;
            wy = (win / 5) * (win_px - win_px / 4), wx = (win % 5) * (win_px - win_px / 4);
        } else if (win < 34) {
            const int k = win - 25;
            img = x1, S = 2 * win_px;
            wy = (k / 3) * (win_px / 2), wx = (k % 3) * (win_px / 2);
        } else {
            img = x2, S = win_px, wy = 0, wx = 0;
        }
        const int64_t so = (((int64_t)b * 3 + c) * S + wy + py * 16 + iy) * S + wx + px * 16 + ixh * 8;
        reinterpret_cast<uint4*>(patches)[i] = *reinterpret_cast<const uint4*>(img + so);
    }
}

template <typename T>
__global__ void patchify_windows_kernel(const T* __restrict__ xs, T* __restrict__ patches,
                                        int windows, int grid) {
    const int S = grid * 16;
    const int P = grid * grid;
    const int64_t total = (int64_t)windows * P * 96;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 96);
        const int64_t row = i / 96;
        const int patch = (int)(row % P);
        const int64_t w = row / P;
        const int c = ch >> 5, iy = (ch >> 1) & 15, ixh = ch & 1;
        const int py = patch / grid, px = patch - py * grid;
        const int64_t so = ((w * 3 + c) * S + py * 16 + iy) * S + px * 16 + ixh * 8;
        reinterpret_cast<uint4*>(patches)[i] = *reinterpret_cast<const uint4*>(xs + so);
    }
}

__global__ void cls_rows_kernel(float* __restrict__ tokens, const float* __restrict__ cls,
                                const float* __restrict__ pos, int windows, int tpw, int dim) {
    // vit.rs:290-294: cls token row = cls + pos[0]
    const int64_t total = (int64_t)windows * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % dim);
        const int64_t w = i / dim;
        tokens[w * tpw * dim + n] = cls[n] + pos[n];
    }
}

// ---------------------------------------------------------------------------------------
// encoder.rs:191-208 reshape_feature + encoder.rs:158-189 merge, as one row gather:
// dst[b][Y][X][:] = tokens[(b*wpi + win0 + j*steps + i)*(P+1) + 1 + ty*g + tx][:]
// ---------------------------------------------------------------------------------------
// split != 0 (f32 source only): the destination has 2*dim channels per pixel, hi = T(v) in [0, dim) and
// lo = T(v - hi) in [dim, 2*dim) (split operands, pipeline.hip)
template <typename T>
__global__ void merge_kernel(const float* __restrict__ src32, const T* __restrict__ src16,
                             T* __restrict__ dst, int batch, int wpi, int win0, int steps,
                             int padding, int grid, int dim, int split, unsigned* __restrict__ status) {
    float amax16 = 0.f;
    const int side = steps == 1 ? grid : 2 * (grid - padding) + (steps - 2) * (grid - 2 * padding);
    const int chunks = dim / 8;
    const int64_t total = (int64_t)batch * side * side * chunks;
    const int P1 = grid * grid + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        int64_t t = i / chunks;
        const int X = (int)(t % side);
        t /= side;
        const int Y = (int)(t % side);
        const int b = (int)(t / side);
        int j = 0, ty = Y, iw = 0, tx = X;
        if (steps > 1) {
            const int first = grid - padding, mid = grid - 2 * padding;
            if (Y >= first) {
                j = 1 + (Y - first) / mid;
                j = j < steps - 1 ? j : steps - 1;
                ty = padding + Y - first - (j - 1) * mid;
            }
            if (X >= first) {
                iw = 1 + (X - first) / mid;
                iw = iw < steps - 1 ? iw : steps - 1;
                tx = padding + X - first - (iw - 1) * mid;
            }
        }
        const int64_t srow = ((int64_t)b * wpi + win0 + j * steps + iw) * P1 + 1 + ty * grid + tx;
        if (src32) {
            const float4 a = *reinterpret_cast<const float4*>(src32 + srow * dim + ch * 8);
            const float4 c = *reinterpret_cast<const float4*>(src32 + srow * dim + ch * 8 + 4);
            typename Vec16<T>::v8 o;
            o[0] = (T)a.x, o[1] = (T)a.y, o[2] = (T)a.z, o[3] = (T)a.w;
            o[4] = (T)c.x, o[5] = (T)c.y, o[6] = (T)c.z, o[7] = (T)c.w;
            amax16 = fmaxf(fmaxf(fmaxf(amax16, fabsf(a.x)), fmaxf(fabsf(a.y), fabsf(a.z))), fabsf(a.w));
            amax16 = fmaxf(fmaxf(fmaxf(amax16, fabsf(c.x)), fmaxf(fabsf(c.y), fabsf(c.z))), fabsf(c.w));
            if (split) {
                const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
                typename Vec16<T>::v8 l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float h = (float)o[e];
                    l[e] = (T)(fabsf(h) == INFINITY ? 0.f : v[e] - h);
                }
                const int64_t pix = i / chunks;
                reinterpret_cast<typename Vec16<T>::v8*>(dst)[pix * 2 * chunks + ch] = o;
                reinterpret_cast<typename Vec16<T>::v8*>(dst)[pix * 2 * chunks + chunks + ch] = l;
            } else {
                reinterpret_cast<typename Vec16<T>::v8*>(dst)[i] = o;
            }
        } else {
            reinterpret_cast<uint4*>(dst)[i] =
                *reinterpret_cast<const uint4*>(src16 + srow * dim + ch * 8);
        }
    }
    raise_overflow16<T>(status, amax16);
}

// ---------------------------------------------------------------------------------------
// Layout changes for the module-level entry points (NCHW f32 at the ABI, NHWC inside).
// 32x32 LDS transposes of the [pixels][channels] matrix.
// ---------------------------------------------------------------------------------------
// split != 0 (16-bit source): pixels hold 2*C channels, value = hi[c] + lo[C + c]
template <typename T, bool SRC32>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ src,
                                                           float* __restrict__ dst, int H, int W,
                                                           int C, int border, int split) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int HW = H * W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (p < HW && c < C) {
            int64_t pix;
            if (border) {
                const int y = p / W, x = p - y * W;
                pix = ((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1;
            } else {
                pix = (int64_t)b * HW + p;
            }
            if (SRC32)
                v = ((const float*)src)[pix * C + c];
            else if (split)
                v = (float)((const T*)src)[pix * 2 * C + c] + (float)((const T*)src)[pix * 2 * C + C + c];
            else
                v = (float)((const T*)src)[pix * C + c];
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, p = p0 + tx;
        if (p < HW && c < C) dst[((int64_t)b * C + c) * HW + p] = tile[tx][ty + 8 * k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst32,
                                                           T* __restrict__ dst16, int H, int W, int C,
                                                           int border, int relu16, int split,
                                                           unsigned* __restrict__ status) {
    __shared__ float tile[32][33];
    float amax16 = 0.f;
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int HW = H * W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, p = p0 + tx;
        tile[ty + 8 * k][tx] = (p < HW && c < C) ? src[((int64_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + ty + 8 * k, c = c0 + tx;
        if (p < HW && c < C) {
            const float v = tile[tx][ty + 8 * k];
            if (dst32) dst32[((int64_t)b * HW + p) * C + c] = v;
            if (dst16) {
                int64_t pix;
                if (border) {
                    const int y = p / W, x = p - y * W;
                    pix = ((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1;
                } else {
                    pix = (int64_t)b * HW + p;
                }
                const float a = relu16 ? fmaxf(v, 0.f) : v;
                amax16 = fmaxf(amax16, fabsf(a));
                if (split) {
                    const T h = (T)a;
                    dst16[pix * 2 * C + c] = h;
                    dst16[pix * 2 * C + C + c] = (T)(fabsf((float)h) == INFINITY ? 0.f : a - (float)h);
                } else {
                    dst16[pix * C + c] = (T)a;
                }
            }
        }
    }
    raise_overflow16<T>(status, amax16);
}

template <typename T>
__global__ void nhwc32_to_16b_kernel(const float* __restrict__ src, T* __restrict__ dst, int batch,
                                     int H, int W, int C, int relu, unsigned* __restrict__ status) {
    float amax16 = 0.f;
    const int c4 = C / 4;
    const int64_t total = (int64_t)batch * H * W * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        int64_t t = i / c4;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        float4 v = reinterpret_cast<const float4*>(src)[i];
        if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
        typename Vec16<T>::v4 o;
        o[0] = (T)v.x, o[1] = (T)v.y, o[2] = (T)v.z, o[3] = (T)v.w;
        amax16 = fmaxf(fmaxf(fmaxf(amax16, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
        const int64_t pix = ((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1;
        *reinterpret_cast<typename Vec16<T>::v4*>(dst + pix * C + c * 4) = o;
    }
    raise_overflow16<T>(status, amax16);
}

__global__ void concat_channels_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                       uint4* __restrict__ dst, int64_t pixels, int ca8, int cb8) {
    const int64_t total = pixels * (ca8 + cb8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (ca8 + cb8));
        const int64_t p = i / (ca8 + cb8);
        dst[i] = c < ca8 ? a[p * ca8 + c] : b[p * cb8 + (c - ca8)];
    }
}

// fov.rs:66-74: x.narrow(1,1,P).permute([0,2,1]).reshape([B,C,g,g]) + relu(downsample(lowres));
// `low` already holds the relu'd conv output, NHWC f32 [B][g*g][C].
template <typename T>
__global__ void fov_add_kernel(const float* __restrict__ lin, const float* __restrict__ low,
                               T* __restrict__ dst16, int batch, int grid, int C, int tpw) {
    const int P = grid * grid;
    const int64_t total = (int64_t)batch * P * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int p = (int)(t % P), b = (int)(t / P);
        const float v = lin[((int64_t)b * tpw + 1 + p) * C + c] + low[i];
        const int y = p / grid, x = p - y * grid;
        dst16[(((int64_t)b * (grid + 2) + y + 1) * (grid + 2) + x + 1) * C + c] = (T)v;
    }
}

// fov.rs:85 head[2] (k x k valid conv to one value) + mod.rs:358 f_norm.  One workgroup per image.
template <typename T>
__global__ __launch_bounds__(256) void fov_final_kernel(const T* __restrict__ x16,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ fov_deg,
                                                        float* __restrict__ f_norm, int n) {
    __shared__ float part[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += (float)x16[(int64_t)b * n + i] * w[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float deg = part[0] + part[1] + part[2] + part[3] + bias[0];
        if (fov_deg) fov_deg[b] = deg;
        // (0.5 * (fov_deg * PI / 180.0)).tan() / 0.5   — mod.rs:358, evaluated in f32
        f_norm[b] = tanf(0.5f * (deg * 3.14159265358979323846f / 180.0f)) / 0.5f;
    }
}

inline unsigned grid_for(int64_t total, int block = 256) {
    const int64_t g = cdiv(total, block);
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

#define ME_BY_DTYPE(dtype, CALL_F16, CALL_BF16)                    \
    do {                                                           \
        if ((dtype) == ME_DTYPE_F16) {                             \
            CALL_F16;                                              \
        } else if ((dtype) == ME_DTYPE_BF16) {                     \
            CALL_BF16;                                             \
        } else {                                                   \
            fail(ME_ERR_BAD_ARG, "bad dtype %d", (int)(dtype));    \
        }                                                          \
        ME_HIP(hipGetLastError());                                 \
    } while (0)

void layernorm_launch(const float* x, const float* w, const float* b, void* y16, float* y32,
                      int64_t rows, int32_t dim, float eps, int32_t dtype, hipStream_t stream,
                      const RowSegs* segs_opt) {
    const RowSegs segs = segs_opt ? *segs_opt : RowSegs();
    ME_CHECK(dim % 64 == 0 && rows > 0, ME_ERR_BAD_SHAPE, "layernorm: rows=%lld dim=%d",
             (long long)rows, dim);
    ProfScope prof(stream, "layernorm_kernel", 0.0, (double)rows * dim * (4 + (y16 ? 2 : 0) + (y32 ? 4 : 0)));
    ME_BY_DTYPE(dtype, layernorm_typed<f16>(x, w, b, y16, y32, rows, dim, eps, stream, segs),
                layernorm_typed<bf16>(x, w, b, y16, y32, rows, dim, eps, stream, segs));
}

void preprocess_u8_launch(const uint8_t* rgb, float* img, int32_t batch, int32_t size,
                          hipStream_t stream) {
    const int64_t ppi = (int64_t)size * size;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for(ppi * batch)), dim3(256), 0, stream, rgb,
                       img, ppi, ppi * batch);
    ME_HIP(hipGetLastError());
}

void cast_f32_to_16_launch(const float* src, void* dst, int64_t count, int32_t dtype,
                           hipStream_t stream) {
    ME_CHECK(count % 4 == 0, ME_ERR_BAD_SHAPE, "cast: count %lld not a multiple of 4",
             (long long)count);
    const int64_t n4 = count / 4;
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(cast_to16_kernel<f16>, dim3(grid_for(n4)), dim3(256), 0, stream,
                                   src, (f16*)dst, n4, current_status_word()),
                hipLaunchKernelGGL(cast_to16_kernel<bf16>, dim3(grid_for(n4)), dim3(256), 0, stream,
                                   src, (bf16*)dst, n4, current_status_word()));
}

void cast_16_to_f32_launch(const void* src, float* dst, int64_t count, int32_t dtype,
                           hipStream_t stream) {
    ME_CHECK(count % 4 == 0, ME_ERR_BAD_SHAPE, "cast: count %lld not a multiple of 4",
             (long long)count);
    const int64_t n4 = count / 4;
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(cast_to32_kernel<f16>, dim3(grid_for(n4)), dim3(256), 0, stream,
                                   (const f16*)src, dst, n4),
                hipLaunchKernelGGL(cast_to32_kernel<bf16>, dim3(grid_for(n4)), dim3(256), 0, stream,
                                   (const bf16*)src, dst, n4));
}

void bilinear_launch(const float* src32, void* dst16, int32_t planes, int32_t in_size,
                     int32_t out_size, int32_t align_corners, int32_t dtype, hipStream_t stream) {
    const int64_t total = (int64_t)planes * out_size * out_size;
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(bilinear_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream,
                                   src32, (f16*)dst16, planes, in_size, out_size, align_corners),
                hipLaunchKernelGGL(bilinear_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0,
                                   stream, src32, (bf16*)dst16, planes, in_size, out_size,
                                   align_corners));
}

void patchify_launch(const void* x0, const void* x1, const void* x2, void* patches, int32_t batch,
                     int32_t grid, int32_t dtype, hipStream_t stream) {
    const int64_t total = (int64_t)batch * 35 * grid * grid * 96;
    // 16-bit payload only: one instantiation serves both dtypes
    (void)dtype;
    hipLaunchKernelGGL(patchify_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const f16*)x0, (const f16*)x1, (const f16*)x2, (f16*)patches, batch, grid);
    ME_HIP(hipGetLastError());
}

void patchify_windows_launch(const void* xs16, void* patches, int32_t windows, int32_t grid,
                             int32_t dtype, hipStream_t stream) {
    (void)dtype;
    const int64_t total = (int64_t)windows * grid * grid * 96;
    hipLaunchKernelGGL(patchify_windows_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const f16*)xs16, (f16*)patches, windows, grid);
    ME_HIP(hipGetLastError());
}

void cls_rows_launch(float* tokens, const float* cls, const float* pos, int32_t windows,
                     int32_t tokens_per_window, int32_t dim, hipStream_t stream) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3(grid_for((int64_t)windows * dim)), dim3(256), 0, stream,
                       tokens, cls, pos, windows, tokens_per_window, dim);
    ME_HIP(hipGetLastError());
}

void merge_launch(const float* src32, const void* src16, void* dst16, int32_t batch,
                  int32_t windows_per_image, int32_t win0, int32_t steps, int32_t padding,
                  int32_t grid, int32_t dim, int32_t dtype, hipStream_t stream, int32_t split) {
    ME_CHECK(dim % 8 == 0, ME_ERR_BAD_SHAPE, "merge: dim %d", dim);
    ME_CHECK((src32 != nullptr) != (src16 != nullptr), ME_ERR_BAD_ARG, "merge: one source");
    ME_CHECK(!split || src32, ME_ERR_BAD_ARG, "merge: a split output needs the f32 source");
    const int side = steps == 1 ? grid : 2 * (grid - padding) + (steps - 2) * (grid - 2 * padding);
    const int64_t total = (int64_t)batch * side * side * (dim / 8);
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(merge_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream,
                                   src32, (const f16*)src16, (f16*)dst16, batch, windows_per_image,
                                   win0, steps, padding, grid, dim, split, current_status_word()),
                hipLaunchKernelGGL(merge_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, stream,
                                   src32, (const bf16*)src16, (bf16*)dst16, batch,
                                   windows_per_image, win0, steps, padding, grid, dim, split, current_status_word()));
}

void nhwc16_to_nchw32_launch(const void* src16, float* dst, int32_t batch, int32_t H, int32_t W,
                             int32_t C, int32_t border, int32_t dtype, hipStream_t stream, int32_t split) {
    const dim3 grid((unsigned)cdiv((int64_t)H * W, 32), (unsigned)cdiv(C, 32), batch);
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL((nhwc_to_nchw_kernel<f16, false>), grid, dim3(256), 0, stream,
                                   src16, dst, H, W, C, border, split),
                hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16, false>), grid, dim3(256), 0, stream,
                                   src16, dst, H, W, C, border, split));
}

void nhwc32_to_nchw32_launch(const float* src, float* dst, int32_t batch, int32_t H, int32_t W,
                             int32_t C, hipStream_t stream) {
    const dim3 grid((unsigned)cdiv((int64_t)H * W, 32), (unsigned)cdiv(C, 32), batch);
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<f16, true>), grid, dim3(256), 0, stream,
                       (const void*)src, dst, H, W, C, 0, 0);
    ME_HIP(hipGetLastError());
}

void nchw32_to_nhwc_launch(const float* src, float* dst32, void* dst16, int32_t batch, int32_t H,
                           int32_t W, int32_t C, int32_t border, int32_t relu16, int32_t dtype,
                           hipStream_t stream, int32_t split) {
    const dim3 grid((unsigned)cdiv((int64_t)H * W, 32), (unsigned)cdiv(C, 32), batch);
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, grid, dim3(256), 0, stream, src, dst32,
                                   (f16*)dst16, H, W, C, border, relu16, split, current_status_word()),
                hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, grid, dim3(256), 0, stream, src,
                                   dst32, (bf16*)dst16, H, W, C, border, relu16, split, current_status_word()));
}

void nhwc32_to_16b_launch(const float* src, void* dst16b, int32_t batch, int32_t H, int32_t W,
                          int32_t C, int32_t relu, int32_t dtype, hipStream_t stream) {
    ME_CHECK(C % 4 == 0, ME_ERR_BAD_SHAPE, "nhwc32_to_16b: C=%d", C);
    const int64_t total = (int64_t)batch * H * W * (C / 4);
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(nhwc32_to_16b_kernel<f16>, dim3(grid_for(total)), dim3(256), 0,
                                   stream, src, (f16*)dst16b, batch, H, W, C, relu, current_status_word()),
                hipLaunchKernelGGL(nhwc32_to_16b_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0,
                                   stream, src, (bf16*)dst16b, batch, H, W, C, relu, current_status_word()));
}

void concat_channels_launch(const void* a, const void* b, void* dst, int64_t pixels, int32_t Ca,
                            int32_t Cb, hipStream_t stream) {
    ME_CHECK(Ca % 8 == 0 && Cb % 8 == 0, ME_ERR_BAD_SHAPE, "concat: %d + %d channels", Ca, Cb);
    hipLaunchKernelGGL(concat_channels_kernel, dim3(grid_for(pixels * (Ca + Cb) / 8)), dim3(256), 0,
                       stream, (const uint4*)a, (const uint4*)b, (uint4*)dst, pixels, Ca / 8, Cb / 8);
    ME_HIP(hipGetLastError());
}

void fov_add_relu_launch(const float* lin, const float* low, void* dst16, int32_t batch,
                         int32_t grid, int32_t C, int32_t tokens_per_window, int32_t dtype,
                         hipStream_t stream) {
    const int64_t total = (int64_t)batch * grid * grid * C;
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(fov_add_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream,
                                   lin, low, (f16*)dst16, batch, grid, C, tokens_per_window),
                hipLaunchKernelGGL(fov_add_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, stream,
                                   lin, low, (bf16*)dst16, batch, grid, C, tokens_per_window));
}

void fov_final_launch(const void* x16, const float* w, const float* bias, float* fov_deg,
                      float* f_norm, int32_t batch, int32_t k, int32_t C, int32_t dtype,
                      hipStream_t stream) {
    const int n = k * k * C;
    ME_BY_DTYPE(dtype,
                hipLaunchKernelGGL(fov_final_kernel<f16>, dim3(batch), dim3(256), 0, stream,
                                   (const f16*)x16, w, bias, fov_deg, f_norm, n),
                hipLaunchKernelGGL(fov_final_kernel<bf16>, dim3(batch), dim3(256), 0, stream,
                                   (const bf16*)x16, w, bias, fov_deg, f_norm, n));
}

}  // namespace me
