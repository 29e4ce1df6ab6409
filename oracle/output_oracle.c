/*
 * CPU oracle for the output back end — TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * A sequential restatement in plain C of reference src/output.rs (zlogic/matrix-eyes v0.1.7),
 * loop for loop, each function citing the lines it follows.  Built with
 *     gcc -O2 -ffp-contract=off -fno-fast-math
 * so that every f32 operation rounds once like the safe Rust of the reference (no FMA
 * contraction; x86-64 SSE arithmetic is IEEE single).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Pinning: the reference has no tests or golden vectors; this file is pinned by the
 * hand-computed known-answer cases in tests/golden/output_known_answers.json
 * (tests/test_oracle_output.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/me_viridis_lut.h"

/* Rust `x as usize` for f32: saturating, NaN -> 0 */
static uint64_t as_usize(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 18446744073709551615.0f) return UINT64_MAX;
    return (uint64_t)v;
}

static float f32_clamp(float v, float lo, float hi) { /* f32::clamp keeps NaN */
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* output.rs:51-57 DepthMap::new clamp + :69-75 inverse_depth_range */
void oracle_clamp_minmax(float* data, int64_t n, float* min_out, float* max_out) {
    const float lo = 1.0f / 250.0f, hi = 1.0f / 0.1f; /* CLAMP_RANGE, :51 */
    for (int64_t i = 0; i < n; ++i) data[i] = f32_clamp(data[i], lo, hi);
    float mn = data[0], mx = data[0];
    for (int64_t i = 0; i < n; ++i) {
        mn = fminf(mn, data[i]); /* f32::min / f32::max ignore a NaN operand */
        mx = fmaxf(mx, data[i]);
    }
    *min_out = mn;
    *max_out = mx;
}

typedef struct {
    const float* data;
    uint64_t data_width, data_height; /* = dims()[0], dims()[1]  (:52) */
} DepthMap;

static float depth_value(const DepthMap* m, uint64_t x, uint64_t y) { /* :78-80 */
    return m->data[m->data_height * y + x];
}

static uint64_t clamp_u(uint64_t v, uint64_t lo, uint64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

static float interpolate_point(const DepthMap* m, float x, float y) { /* :83-98 */
    x = fmaxf(x * (float)m->data_width, 0.0f);
    y = fmaxf(y * (float)m->data_height, 0.0f);
    const uint64_t x0 = clamp_u(as_usize(floorf(x)), 0, m->data_width - 1);
    const uint64_t y0 = clamp_u(as_usize(floorf(y)), 0, m->data_height - 1);
    const uint64_t x1 = clamp_u(x0 + 1, 0, m->data_width - 1);
    const uint64_t y1 = clamp_u(y0 + 1, 0, m->data_height - 1);
    x = x - truncf(x); /* f32::fract */
    y = y - truncf(y);
    return (1.0f - x) * (1.0f - y) * depth_value(m, x0, y0) + x * (1.0f - y) * depth_value(m, x1, y0) +
           (1.0f - x) * y * depth_value(m, x0, y1) + x * y * depth_value(m, x1, y1);
}

/* output.rs:141-193 output_stereogram; `noise` replaces rand::rng() (:163-171): one RGB triple per
   pixel in raster order.  Returns -1 where the reference would panic on an out-of-range index. */
int oracle_stereogram(const float* depth, int32_t rows, int32_t cols, float min_depth, float max_depth,
                      int32_t output_width, int32_t output_height, float amplitude,
                      const uint8_t* noise, uint8_t* out) {
    const DepthMap m = {depth, (uint64_t)rows, (uint64_t)cols};
    const float depth_multiplier = (float)output_width * amplitude;               /* :160 */
    const uint64_t pattern_width = as_usize(roundf(depth_multiplier * 2.0f + amplitude)); /* :161 */
    uint8_t* output_row = (uint8_t*)malloc((size_t)output_width * 3);
    for (int32_t y = 0; y < output_height; ++y) {
        const uint8_t* noise_row = noise + (size_t)y * output_width * 3;
        memcpy(output_row, noise_row, (size_t)output_width * 3);                  /* :172 */
        for (int32_t xi = 0; xi < output_width; ++xi) {
            float d = interpolate_point(&m, (float)xi / (float)output_width,
                                        (float)y / (float)output_height);         /* :174-177 */
            d = (d - min_depth) / (max_depth - min_depth);                        /* :178 */
            const uint64_t x = (uint64_t)xi;
            const uint8_t* src;
            if (x >= pattern_width) {                                             /* :180-182 */
                const uint64_t shift = as_usize(roundf(d * depth_multiplier));
                const uint64_t idx = x + shift - pattern_width;
                if (idx >= (uint64_t)output_width) {
                    free(output_row);
                    return -1;
                }
                src = output_row + idx * 3;
            } else {
                src = noise_row + (x % pattern_width) * 3;                        /* :184 */
            }
            uint8_t px[3] = {src[0], src[1], src[2]};
            memcpy(output_row + x * 3, px, 3);
        }
        memcpy(out + (size_t)y * output_width * 3, output_row, (size_t)output_width * 3); /* :187-189 */
    }
    free(output_row);
    return 0;
}

static uint8_t map_color(int channel, float value) { /* :704-714 */
    if (value >= 1.0f) return ME_VIRIDIS_REV[255][channel];
    const float step = 1.0f / (float)(256 - 1);
    const uint64_t box_index = clamp_u(as_usize(floorf(value / step)), 0, 256 - 2);
    const float ratio = (value - step * (float)box_index) / step;
    const float c1 = (float)ME_VIRIDIS_REV[box_index][channel];
    const float c2 = (float)ME_VIRIDIS_REV[box_index + 1][channel];
    const float v = roundf(c2 * ratio + c1 * (1.0f - ratio));
    if (!(v > 0.0f)) return 0; /* `as u8` saturates, NaN -> 0 */
    return v >= 255.0f ? 255 : (uint8_t)v;
}

/* output.rs:123-131 (the per-pixel part of output_depth_map) */
void oracle_depthmap_rgb(const float* depth, int64_t n, float min_depth, float max_depth, uint8_t* rgb) {
    for (int64_t i = 0; i < n; ++i) {
        const float d = (max_depth - depth[i]) / (max_depth - min_depth);
        for (int c = 0; c < 3; ++c) rgb[3 * i + c] = map_color(c, d);
    }
}

/* f32::total_cmp ordering key */
static int32_t total_key(float f) {
    int32_t b;
    memcpy(&b, &f, 4);
    return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}
static float min3_total(const float v[3]) { /* Iterator::min_by keeps the first of equal elements */
    float m = v[0];
    for (int i = 1; i < 3; ++i)
        if (total_key(v[i]) < total_key(m)) m = v[i];
    return m;
}
static float max3_total(const float v[3]) { /* Iterator::max_by keeps the last of equal elements */
    float m = v[0];
    for (int i = 1; i < 3; ++i)
        if (total_key(v[i]) >= total_key(m)) m = v[i];
    return m;
}

typedef void (*face_fn)(void* user, const int64_t idx[3]);

/* output.rs:307-355 IndexedMesh::for_each_face */
static void for_each_face(const float* vertices, int64_t width, int64_t height, face_fn fn, void* user) {
    const float threshold = 1.025f; /* POLYGON_DEPTH_THRESHOLD, :40 */
    for (int64_t y = 0; y < height - 1; ++y)
        for (int64_t x = 0; x < width - 1; ++x) {
            const int64_t i00 = y * width + x, i10 = y * width + x + 1;
            const int64_t i01 = (y + 1) * width + x, i11 = (y + 1) * width + x + 1;
            const float v00 = vertices[i00], v10 = vertices[i10], v01 = vertices[i01], v11 = vertices[i11];
            const int64_t i_ul[3] = {i00, i01, i10}, i_lr[3] = {i10, i01, i11};
            const float v_ul[3] = {v00, v01, v10}, v_lr[3] = {v10, v01, v11};
            if (max3_total(v_ul) / min3_total(v_ul) <= threshold) fn(user, i_ul);
            if (max3_total(v_lr) / min3_total(v_lr) <= threshold) fn(user, i_lr);
        }
}

typedef struct {
    int32_t* index; /* -1 = None */
    int64_t next_vertex, faces_count;
    int32_t* faces; /* nullable */
} IndexState;

static void index_face(void* user, const int64_t idx[3]) { /* :276-286 */
    IndexState* s = (IndexState*)user;
    for (int k = 0; k < 3; ++k)
        if (s->index[idx[k]] < 0) s->index[idx[k]] = (int32_t)s->next_vertex++;
    s->faces_count++;
}
static void emit_face(void* user, const int64_t idx[3]) { /* :251-256 with remap_face :357-362 */
    IndexState* s = (IndexState*)user;
    for (int k = 0; k < 3; ++k) s->faces[3 * s->faces_count + k] = s->index[idx[k]];
    s->faces_count++;
}

/* output.rs:272-294 IndexedMesh::new, then the face pass of output_mesh */
void oracle_mesh_index(const float* depth, int32_t width, int32_t height, int32_t* vertex_index,
                       int64_t* nvertices, int64_t* nfaces, int32_t* faces) {
    const int64_t n = (int64_t)width * height;
    for (int64_t i = 0; i < n; ++i) vertex_index[i] = -1;
    IndexState s = {vertex_index, 0, 0, faces};
    for_each_face(depth, width, height, index_face, &s);
    *nvertices = s.next_vertex;
    *nfaces = s.faces_count;
    if (faces) {
        s.faces_count = 0;
        for_each_face(depth, width, height, emit_face, &s);
    }
}

/* output.rs:222-249: uv and xyz per vertex id (sorted_vertices order = id order) */
void oracle_mesh_vertices(const float* depth, int32_t data_width, int32_t data_height,
                          const int32_t* vertex_index, uint32_t original_width, uint32_t original_height,
                          float* uv, float* xyz) {
    const uint32_t mx = original_width > original_height ? original_width : original_height;
    const float x_multiplier = (float)original_width / (float)mx;  /* :222-223 */
    const float y_multiplier = (float)original_height / (float)mx; /* :224-225 */
    const int64_t n = (int64_t)data_width * data_height;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t id = vertex_index[i];
        if (id < 0) continue;
        const int64_t x_image = i % data_width, y_image = i / data_width; /* :301 */
        const float x_norm = (float)x_image / (float)data_width;           /* :229-232 */
        const float y_norm = (float)y_image / (float)data_height;
        uv[2 * (int64_t)id] = x_norm;
        uv[2 * (int64_t)id + 1] = y_norm;
        const float z_norm = 1.0f / depth[i];                               /* :245 */
        xyz[3 * (int64_t)id] = x_multiplier * (x_norm - 0.5f) * z_norm;     /* :246 */
        xyz[3 * (int64_t)id + 1] = y_multiplier * (y_norm - 0.5f) * z_norm; /* :247 */
        xyz[3 * (int64_t)id + 2] = z_norm;
    }
}

/* ---------------------------------------------------------------------------------------------
 * output.rs:484-630 ObjWriter as a file writer, for OBJ files too large for the Python formatter
 * (a 1536^2 textured mesh is ~450 MB).  Rust's `{}` for f64 prints the shortest decimal that
 * round-trips, never in scientific notation.  Restated independently of the product's
 * std::to_chars: the correctly rounded 15-, 16- or 17-significant-digit decimal (printf %.*e) --
 * the first that strtod reads back to the same double is the shortest one, and being correctly
 * rounded it is also the closest of that length, which is the digit string Rust's Grisu / Ryu
 * prints -- then laid out positionally.
 * ------------------------------------------------------------------------------------------- */
#include <stdio.h>

static int rust_display_f64(double v, char* out) { /* returns the length; out needs 400 bytes */
    if (v != v) return sprintf(out, "NaN");
    if (isinf(v)) return sprintf(out, v > 0 ? "inf" : "-inf");
    if (v == 0.0) return sprintf(out, signbit(v) ? "-0" : "0");
    char e[40];
    int prec;
    /* a normal double is identified by any decimal of <= 15 significant digits (DBL_DIG), so 15 digits either
     * round-trip -- then trailing zeros are all that separates them from the shortest form -- or 16 or 17 are
     * needed; subnormals have fewer bits and are searched from one digit up */
    for (prec = fabs(v) < 2.2250738585072014e-308 ? 0 : 14; prec <= 16; ++prec) { /* %.14e = 15 digits */
        snprintf(e, sizeof e, "%.*e", prec, v);
        if (strtod(e, NULL) == v) break;
    }
    /* e = [-]d.ddddde[+-]xx : digits and decimal exponent */
    char digits[24];
    int nd = 0, neg = e[0] == '-';
    const char* p = e + neg;
    for (; *p && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    const int exp10 = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') --nd; /* shortest: trailing zeros carry nothing */
    int n = 0;
    if (neg) out[n++] = '-';
    if (exp10 >= 0) {
        for (int i = 0; i <= exp10; ++i) out[n++] = i < nd ? digits[i] : '0';
        if (nd > exp10 + 1) {
            out[n++] = '.';
            for (int i = exp10 + 1; i < nd; ++i) out[n++] = digits[i];
        }
    } else {
        out[n++] = '0';
        out[n++] = '.';
        for (int i = 0; i < -exp10 - 1; ++i) out[n++] = '0';
        for (int i = 0; i < nd; ++i) out[n++] = digits[i];
    }
    out[n] = 0;
    return n;
}

/* test hook for the formatter above */
int oracle_rust_display_f64(double v, char* out) { return rust_display_f64(v, out); }

/* vertex_mode: 0 plain, 1 vertex colours, 2 texture coordinates (output.rs:34-38).  uv [nverts][2], xyz
 * [nverts][3] as oracle_mesh_vertices returns them, faces [nfaces][3] 0-based, colors [nverts][3] or NULL.
 * Returns 0, or -1 when the file cannot be written. */
int oracle_write_obj(const char* path, const char* stem, int32_t vertex_mode, const float* uv, const float* xyz,
                     int64_t nverts, const int32_t* faces, int64_t nfaces, const uint8_t* colors) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    static char buf[1 << 16];
    setvbuf(f, buf, _IOFBF, sizeof buf);
    char a[400], b[400], c[400];
    if (vertex_mode == 2) { /* :551-564 header, :592-602 write_vertex_texture */
        fprintf(f, "mtllib %s.mtl\nusemtl Textured\n", stem);
        for (int64_t i = 0; i < nverts; ++i) {
            rust_display_f64((double)uv[2 * i], a);
            rust_display_f64(1.0 - (double)uv[2 * i + 1], b);
            fprintf(f, "vt %s %s\n", a, b);
        }
    }
    for (int64_t i = 0; i < nverts; ++i) { /* :566-590 write_vertex: x, -y, -z (negated in f32, :235-249) */
        rust_display_f64((double)xyz[3 * i], a);
        rust_display_f64((double)(-xyz[3 * i + 1]), b);
        rust_display_f64((double)(-xyz[3 * i + 2]), c);
        fprintf(f, "v %s %s %s", a, b, c);
        if (vertex_mode == 1 && colors) {
            for (int k = 0; k < 3; ++k) {
                rust_display_f64((double)colors[3 * i + k] / 255.0, a);
                fprintf(f, " %s", a);
            }
        }
        fputc('\n', f);
    }
    for (int64_t i = 0; i < nfaces; ++i) { /* :604-620 write_face, 1-based */
        const long long x = faces[3 * i] + 1ll, y = faces[3 * i + 1] + 1ll, z = faces[3 * i + 2] + 1ll;
        if (vertex_mode == 2)
            fprintf(f, "f %lld/%lld %lld/%lld %lld/%lld\n", x, x, y, y, z, z);
        else
            fprintf(f, "f %lld %lld %lld\n", x, y, z);
    }
    return fclose(f) == 0 ? 0 : -1;
}
