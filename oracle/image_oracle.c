/* CPU oracle of the image front end's resampler -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * reconstruction.rs:107-113 calls DynamicImage::resize_exact(IMG_SIZE, IMG_SIZE, FilterType::Lanczos3); output.rs:133-137
 * resizes the depth map back the same way.  The algorithm lives in the third-party crate `image`, pinned at 0.25.10
 * (Cargo.toml:13, Cargo.lock:3037-3040), which is not vendored under /root/reference: this file restates its published
 * algorithm, src/imageops/sample.rs of that release, function by function:
 *
 *   sinc(t)            = t == 0 ? 1 : sin(t * PI) / (t * PI)                                 (f32)
 *   lanczos3_kernel(x) = |x| < 3 ? sinc(x) * sinc(x / 3) : 0
 *   resize()           : same size -> copy; else  tmp: Rgba32F = vertical_sample(image, nheight);
 *                        horizontal_sample(&tmp, nwidth)           -- rows first, the intermediate stays f32, unclamped
 *   *_sample()         : ratio = in / out, sratio = max(ratio, 1), src_support = support * sratio;
 *                        for each output index o: center = (o + 0.5) * ratio;
 *                          left  = clamp(floor(center - src_support), 0, in - 1)
 *                          right = clamp(ceil(center + src_support), left + 1, in)
 *                          w_i   = kernel((i - (center - 0.5)) / sratio), i in [left, right);  w_i /= sum(w)
 *                          t     = sum_i pixel_i * w_i   accumulated in f32 in index order, one channel at a time
 *                        horizontal_sample stores  round(clamp(t, 0, 255))  -- FloatNearest: f32::round, half away from 0
 *
 * Parity unpinned: the reference holds no image fixtures and cannot be built here (no Rust toolchain).  sin() is the C
 * library's sinf, as Rust's f32::sin is on this platform.  Compiled -ffp-contract=off: Rust never fuses a*b+c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float sincf_(float t) {
    if (t == 0.0f) return 1.0f;
    const float a = t * 3.14159265358979323846f;
    return sinf(a) / a;
}
static float lanczos3_kernel(float x) { return fabsf(x) < 3.0f ? sincf_(x) * sincf_(x / 3.0f) : 0.0f; }

/* one pass along an axis of `len_in` samples: `lines` independent lines of `ch` interleaved channels */
static void sample_axis(const float* in, float* out, int64_t len_in, int64_t len_out, int64_t lines, int ch, int64_t in_step,
                        int64_t in_line, int64_t out_step, int64_t out_line) {
    const float ratio = (float)len_in / (float)len_out;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 3.0f * sratio;
    float* ws = (float*)malloc(sizeof(float) * (size_t)(len_in + 2));
    for (int64_t o = 0; o < len_out; ++o) {
        float center = ((float)o + 0.5f) * ratio;
        int64_t left = (int64_t)floorf(center - src_support);
        if (left < 0) left = 0;
        if (left > len_in - 1) left = len_in - 1;
        int64_t right = (int64_t)ceilf(center + src_support);
        if (right < left + 1) right = left + 1;
        if (right > len_in) right = len_in;
        center = center - 0.5f;
        float sum = 0.0f;
        for (int64_t i = left; i < right; ++i) {
            const float w = lanczos3_kernel(((float)i - center) / sratio);
            ws[i - left] = w;
            sum += w;
        }
        for (int64_t i = 0; i < right - left; ++i) ws[i] /= sum;
        for (int64_t l = 0; l < lines; ++l)
            for (int c = 0; c < ch; ++c) {
                float t = 0.0f;
                for (int64_t i = left; i < right; ++i) t += in[l * in_line + i * in_step + c] * ws[i - left];
                out[l * out_line + o * out_step + c] = t;
            }
    }
    free(ws);
}

/* imageops::resize(&ImageBuffer<Rgb<u8>>, nwidth, nheight, Lanczos3): src [h][w][3] u8 -> dst [nh][nw][3] u8 */
int oracle_resize_lanczos3_rgb8(const uint8_t* src, int64_t w, int64_t h, uint8_t* dst, int64_t nw, int64_t nh) {
    if (w <= 0 || h <= 0 || nw <= 0 || nh <= 0) return 1;
    if (w == nw && h == nh) {
        for (int64_t i = 0; i < w * h * 3; ++i) dst[i] = src[i];
        return 0;
    }
    float* a = (float*)malloc(sizeof(float) * (size_t)(w * h * 3));
    float* mid = (float*)malloc(sizeof(float) * (size_t)(w * nh * 3));
    float* fin = (float*)malloc(sizeof(float) * (size_t)(nw * nh * 3));
    if (!a || !mid || !fin) return 2;
    for (int64_t i = 0; i < w * h * 3; ++i) a[i] = (float)src[i];
    sample_axis(a, mid, h, nh, w, 3, w * 3, 3, w * 3, 3);        /* vertical_sample: lines = columns */
    sample_axis(mid, fin, w, nw, nh, 3, 3, w * 3, 3, nw * 3);    /* horizontal_sample: lines = rows */
    for (int64_t i = 0; i < nw * nh * 3; ++i) {
        float t = fin[i];
        t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
        dst[i] = (uint8_t)roundf(t);
    }
    free(a), free(mid), free(fin);
    return 0;
}

/* <u8 as FromPrimitive<u16>>::from_primitive (image 0.25 color.rs): 16-bit sample -> 8 bit, rounded */
uint8_t oracle_u16_to_u8(uint16_t v) { return (uint8_t)(((uint32_t)v + 128u) / 257u); }
