// JPEG decoder of the C++ host layer: baseline / extended sequential and progressive Huffman JPEG (SOF0, SOF1,
// SOF2), 8-bit, grey or YCbCr with 1x / 2x chroma subsampling, restart intervals, interleaved and
// non-interleaved scans.  The reference decodes through the `image` crate (reconstruction.rs:96-106); like
// every JPEG decoder pair, two conforming decoders agree to within a unit or two of rounding, not bit for bit,
// so the tests compare against Pillow (libjpeg-turbo) with that tolerance.  Follows ITU-T T.81: Annex F
// (sequential), Annex G (progressive); chroma upsampling is libjpeg's triangle filter ("fancy upsampling"),
// the colour conversion libjpeg's 16-bit fixed-point JFIF matrix.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "image_io.hpp"

namespace matrix_eyes {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t values[256];
    int mincode[17], maxcode[18], valptr[17];
    void build(const uint8_t counts[16], const uint8_t* vals, int nvals) {
        std::memcpy(values, vals, (size_t)nvals);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;            // Huffman table selectors of the current scan
    int width = 0, height = 0;     // samples: ceil(W * h / hmax)
    int blocks_w = 0, blocks_h = 0;  // allocated blocks (whole MCUs)
    int pred = 0;
    std::vector<int16_t> coef;     // blocks_w * blocks_h * 64, natural (de-zigzagged) order
    std::vector<uint8_t> plane;    // blocks_w * 8 x blocks_h * 8 samples after the IDCT
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
    void fill() {
        while (bits <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && p < end) {
                if (*p == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) {
                        byte = 0xff, p += 2;
                    } else {
                        hit_marker = true;  // zeros from here on (T.81 F.2.2.5)
                    }
                } else {
                    byte = *p++;
                }
            }
            acc |= byte << (24 - bits);
            bits += 8;
        }
    }
    int get(int n) {
        if (n == 0) return 0;
        if (bits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n, bits -= n;
        return v;
    }
    int bit() { return get(1); }
    // to the byte boundary, then past an RSTn marker if one follows
    void restart() {
        acc = 0, bits = 0, hit_marker = false;
        while (p + 1 < end && !(p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7)) {
            if (p[0] == 0xff && p[1] != 0x00 && p[1] != 0xff) return;  // some other marker: let the caller see it
            ++p;
        }
        if (p + 1 < end) p += 2;
    }
};

int decode_symbol(BitReader& br, const HuffTable& t, const std::string& path) {
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.bit();
        if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len])
            return t.values[t.valptr[len] + code - t.mincode[len]];
    }
    throw ImageError(path + ": bad Huffman code in JPEG data");
}

int receive_extend(BitReader& br, int s) {
    if (s == 0) return 0;
    const int v = br.get(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

struct Decoder {
    const std::vector<uint8_t>& file;
    const std::string& path;
    int width = 0, height = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;
    int adobe_transform = -1;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    std::vector<Component> comps;
    std::vector<uint8_t> exif;

    Decoder(const std::vector<uint8_t>& f, const std::string& p) : file(f), path(p) {}

    [[noreturn]] void fail(const std::string& what) const { throw ImageError(path + ": " + what); }

    uint16_t be16(size_t pos) const {
        if (pos + 2 > file.size()) fail("truncated JPEG");
        return (uint16_t)((file[pos] << 8) | file[pos + 1]);
    }

    void parse_dqt(size_t pos, size_t end) {
        while (pos < end) {
            const int pq = file[pos] >> 4, tq = file[pos] & 15;
            ++pos;
            if (tq > 3 || pos + (pq ? 128 : 64) > end) fail("bad DQT");
            for (int i = 0; i < 64; ++i) {
                qt[tq][kZigzag[i]] = pq ? be16(pos) : file[pos];
                pos += pq ? 2 : 1;
            }
            qt_present[tq] = true;
        }
    }

    void parse_dht(size_t pos, size_t end) {
        while (pos < end) {
            if (pos + 17 > end) fail("bad DHT");
            const int tc = file[pos] >> 4, th = file[pos] & 15;
            if (tc > 1 || th > 3) fail("bad DHT");
            const uint8_t* counts = &file[pos + 1];
            int n = 0;
            for (int i = 0; i < 16; ++i) n += counts[i];
            if (n > 256 || pos + 17 + (size_t)n > end) fail("bad DHT");
            (tc ? ac[th] : dc[th]).build(counts, &file[pos + 17], n);
            pos += 17 + (size_t)n;
        }
    }

    void parse_sof(size_t pos, size_t end, bool prog) {
        if (have_frame) fail("more than one frame in JPEG");
        if (pos + 6 > end) fail("bad SOF");
        if (file[pos] != 8) fail("only 8-bit JPEG is supported");
        height = be16(pos + 1), width = be16(pos + 3);
        const int n = file[pos + 5];
        if (!width || !height) fail("JPEG with zero size");
        if ((int64_t)width * height > kMaxPixels) fail("JPEG larger than 2^28 pixels");
        if (n != 1 && n != 3) fail("JPEG with " + std::to_string(n) + " components is not supported");
        if (pos + 6 + 3 * (size_t)n > end) fail("bad SOF");
        comps.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            Component& c = comps[(size_t)i];
            c.id = file[pos + 6 + 3 * i];
            c.h = file[pos + 7 + 3 * i] >> 4, c.v = file[pos + 7 + 3 * i] & 15;
            c.tq = file[pos + 8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) fail("bad SOF component");
            hmax = std::max(hmax, c.h), vmax = std::max(vmax, c.v);
        }
        if (n == 1) comps[0].h = comps[0].v = hmax = vmax = 1;  // a single component is never interleaved
        const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (Component& c : comps) {
            c.width = (width * c.h + hmax - 1) / hmax, c.height = (height * c.v + vmax - 1) / vmax;
            c.blocks_w = mcus_x * c.h, c.blocks_h = mcus_y * c.v;
            c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        progressive = prog, have_frame = true;
    }

    // ---- entropy-coded segments ---------------------------------------------------------------------
    struct Scan {
        std::vector<int> comp;  // indices into comps
        int ss = 0, se = 63, ah = 0, al = 0;
    };
    int eobrun = 0;

    void block_sequential(BitReader& br, Component& c, int16_t* b) {
        const int t = decode_symbol(br, dc[c.td], path);
        c.pred += receive_extend(br, t);
        b[0] = (int16_t)c.pred;
        const HuffTable& h = ac[c.ta];
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(br, h, path), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) fail("bad AC run in JPEG data");
            b[kZigzag[k]] = (int16_t)receive_extend(br, s);
            ++k;
        }
    }
    void block_dc_first(BitReader& br, Component& c, int16_t* b, int al) {
        const int t = decode_symbol(br, dc[c.td], path);
        c.pred += receive_extend(br, t);
        b[0] = (int16_t)(c.pred * (1 << al));
    }
    void block_dc_refine(BitReader& br, int16_t* b, int al) {
        if (br.bit()) b[0] = (int16_t)(b[0] | (1 << al));
    }
    void block_ac_first(BitReader& br, Component& c, int16_t* b, const Scan& sc) {
        if (eobrun > 0) {
            --eobrun;
            return;
        }
        const HuffTable& h = ac[c.ta];
        for (int k = sc.ss; k <= sc.se;) {
            const int rs = decode_symbol(br, h, path), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1 + (r ? br.get(r) : 0);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) fail("bad AC run in JPEG data");
            b[kZigzag[k]] = (int16_t)(receive_extend(br, s) * (1 << sc.al));
            ++k;
        }
    }
    void block_ac_refine(BitReader& br, Component& c, int16_t* b, const Scan& sc) {
        const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
        const HuffTable& h = ac[c.ta];
        int k = sc.ss;
        auto refine = [&](int16_t& v) {
            if (br.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
        };
        if (eobrun == 0) {
            while (k <= sc.se) {
                const int rs = decode_symbol(br, h, path);
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s) {
                    if (s != 1) fail("bad AC refinement in JPEG data");
                    value = br.bit() ? p1 : m1;
                } else if (r < 15) {
                    eobrun = (1 << r) + (r ? br.get(r) : 0);
                    break;
                }
                while (k <= sc.se) {  // skip r zero-history coefficients, refining the others on the way
                    int16_t& v = b[kZigzag[k]];
                    if (v != 0) {
                        refine(v);
                    } else if (--r < 0) {
                        break;
                    }
                    ++k;
                }
                if (s && k <= sc.se) b[kZigzag[k]] = (int16_t)value;
                ++k;
            }
        }
        if (eobrun > 0) {
            for (; k <= sc.se; ++k) {
                int16_t& v = b[kZigzag[k]];
                if (v != 0) refine(v);
            }
            --eobrun;
        }
    }

    void decode_block(BitReader& br, Component& c, int bx, int by, const Scan& sc) {
        int16_t* b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
        if (!progressive)
            block_sequential(br, c, b);
        else if (sc.ss == 0)
            sc.ah == 0 ? block_dc_first(br, c, b, sc.al) : block_dc_refine(br, b, sc.al);
        else
            sc.ah == 0 ? block_ac_first(br, c, b, sc) : block_ac_refine(br, c, b, sc);
    }

    // returns the position just past the entropy-coded data
    size_t decode_scan(size_t pos, const Scan& sc) {
        BitReader br(&file[pos], file.data() + file.size());
        for (int ci : sc.comp) comps[(size_t)ci].pred = 0;
        eobrun = 0;
        int units_x, units_y;
        const bool interleaved = sc.comp.size() > 1;
        if (interleaved) {
            units_x = (width + 8 * hmax - 1) / (8 * hmax), units_y = (height + 8 * vmax - 1) / (8 * vmax);
        } else {
            const Component& c = comps[(size_t)sc.comp[0]];
            units_x = (c.width + 7) / 8, units_y = (c.height + 7) / 8;
        }
        int until_restart = restart_interval;
        for (int uy = 0; uy < units_y; ++uy)
            for (int ux = 0; ux < units_x; ++ux) {
                if (restart_interval && until_restart == 0) {
                    br.restart();
                    for (int ci : sc.comp) comps[(size_t)ci].pred = 0;
                    eobrun = 0;
                    until_restart = restart_interval;
                }
                if (interleaved) {
                    for (int ci : sc.comp) {
                        Component& c = comps[(size_t)ci];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) decode_block(br, c, ux * c.h + x, uy * c.v + y, sc);
                    }
                } else {
                    decode_block(br, comps[(size_t)sc.comp[0]], ux, uy, sc);
                }
                --until_restart;
            }
        // the next marker: the reader stops consuming at it (stuffed bytes and RSTn are data)
        const uint8_t* p = br.p;
        const uint8_t* end = file.data() + file.size();
        while (p + 1 < end && !(p[0] == 0xff && p[1] != 0x00 && p[1] != 0xff && !(p[1] >= 0xd0 && p[1] <= 0xd7))) ++p;
        return (size_t)(p - file.data());
    }

    size_t parse_sos(size_t pos, size_t end) {
        if (!have_frame) fail("SOS before SOF");
        const int n = file[pos];
        if (n < 1 || n > (int)comps.size() || pos + 1 + 2 * (size_t)n + 3 > end) fail("bad SOS");
        Scan sc;
        for (int i = 0; i < n; ++i) {
            const int id = file[pos + 1 + 2 * i], tables = file[pos + 2 + 2 * i];
            int ci = -1;
            for (size_t k = 0; k < comps.size(); ++k)
                if (comps[k].id == id) ci = (int)k;
            if (ci < 0) fail("SOS names an unknown component");
            comps[(size_t)ci].td = tables >> 4, comps[(size_t)ci].ta = tables & 15;
            if ((tables >> 4) > 3 || (tables & 15) > 3) fail("bad SOS table selector");
            sc.comp.push_back(ci);
        }
        const size_t q = pos + 1 + 2 * (size_t)n;
        sc.ss = file[q], sc.se = file[q + 1], sc.ah = file[q + 2] >> 4, sc.al = file[q + 2] & 15;
        if (!progressive) {
            sc.ss = 0, sc.se = 63, sc.ah = sc.al = 0;
        } else {
            if (sc.ss > sc.se || sc.se > 63 || (sc.ss == 0 && sc.se != 0) || (sc.ss > 0 && n != 1) || sc.al > 13)
                fail("bad progressive scan parameters");
        }
        for (int ci : sc.comp) {
            const Component& c = comps[(size_t)ci];
            const bool need_dc = !progressive || (sc.ss == 0 && sc.ah == 0), need_ac = !progressive || sc.ss > 0;
            if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present)) fail("scan uses an undefined Huffman table");
        }
        return decode_scan(end, sc);
    }

    // ---- reconstruction -------------------------------------------------------------------------------
    void idct_all() {
        // exact separable IDCT in double precision (T.81 A.3.3), level shift, clamp
        double basis[8][8];
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u)
                basis[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        for (Component& c : comps) {
            if (!qt_present[c.tq]) fail("component uses an undefined quantisation table");
            const uint16_t* q = qt[c.tq];
            const int pw = c.blocks_w * 8;
            c.plane.assign((size_t)pw * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx) {
                    const int16_t* b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
                    double in[64], tmp[64];
                    bool ac_zero = true;
                    for (int i = 0; i < 64; ++i) {
                        in[i] = (double)b[i] * q[i];
                        if (i && b[i]) ac_zero = false;
                    }
                    uint8_t* out = &c.plane[(size_t)by * 8 * pw + (size_t)bx * 8];
                    if (ac_zero) {
                        const int v = (int)std::lround(in[0] / 8.0) + 128;
                        const uint8_t px = (uint8_t)std::min(255, std::max(0, v));
                        for (int y = 0; y < 8; ++y) std::memset(out + (size_t)y * pw, px, 8);
                        continue;
                    }
                    for (int v = 0; v < 8; ++v)  // rows: over u
                        for (int x = 0; x < 8; ++x) {
                            double s = 0;
                            for (int u = 0; u < 8; ++u) s += basis[x][u] * in[v * 8 + u];
                            tmp[v * 8 + x] = s;
                        }
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x) {
                            double s = 0;
                            for (int v = 0; v < 8; ++v) s += basis[y][v] * tmp[v * 8 + x];
                            const int px = (int)std::lround(s) + 128;
                            out[(size_t)y * pw + x] = (uint8_t)std::min(255, std::max(0, px));
                        }
                }
            c.coef.clear();
            c.coef.shrink_to_fit();
        }
    }

    // component plane -> full-resolution plane [height][width]
    std::vector<uint8_t> upsample(const Component& c) const {
        const int pw = c.blocks_w * 8;
        std::vector<uint8_t> out((size_t)width * height);
        const int fx = hmax / c.h, fy = vmax / c.v;
        if (hmax % c.h || vmax % c.v) fail("fractional JPEG sampling ratios are not supported");
        auto row = [&](int y) { return &c.plane[(size_t)std::min(std::max(y, 0), c.height - 1) * pw]; };
        const int n = c.width;
        if (fx == 1 && fy == 1) {
            for (int y = 0; y < height; ++y) std::memcpy(&out[(size_t)y * width], row(y), (size_t)width);
        } else if (fx == 2 && fy == 1 && n > 2) {  // libjpeg h2v1_fancy_upsample (plain replication up to 2 columns)
            std::vector<uint8_t> line((size_t)2 * n);
            for (int y = 0; y < height; ++y) {
                const uint8_t* in = row(y);
                line[0] = in[0], line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < n - 1; ++i) {
                    line[2 * i] = (uint8_t)((in[i] * 3 + in[i - 1] + 1) >> 2);
                    line[2 * i + 1] = (uint8_t)((in[i] * 3 + in[i + 1] + 2) >> 2);
                }
                line[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2), line[2 * n - 1] = in[n - 1];
                std::memcpy(&out[(size_t)y * width], line.data(), (size_t)width);
            }
        } else if (fx == 2 && fy == 2 && n > 2) {  // libjpeg h2v2_fancy_upsample
            std::vector<uint8_t> line((size_t)2 * n);
            for (int y = 0; y < height; ++y) {
                const int r = y >> 1;
                const uint8_t* in0 = row(r);
                const uint8_t* in1 = row((y & 1) ? r + 1 : r - 1);
                auto colsum = [&](int i) { return in0[i] * 3 + in1[i]; };
                int last = colsum(0), cur = colsum(0), next = colsum(1);
                line[0] = (uint8_t)((cur * 4 + 8) >> 4), line[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                for (int i = 1; i < n - 1; ++i) {
                    last = cur, cur = next, next = colsum(i + 1);
                    line[2 * i] = (uint8_t)((cur * 3 + last + 8) >> 4);
                    line[2 * i + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                }
                last = cur, cur = next;
                line[2 * n - 2] = (uint8_t)((cur * 3 + last + 8) >> 4), line[2 * n - 1] = (uint8_t)((cur * 4 + 7) >> 4);
                std::memcpy(&out[(size_t)y * width], line.data(), (size_t)width);
            }
        } else if (fx == 1 && fy == 2) {  // libjpeg-turbo h1v2_fancy_upsample
            for (int y = 0; y < height; ++y) {
                const int r = y >> 1;
                const uint8_t* in0 = row(r);
                const uint8_t* in1 = row((y & 1) ? r + 1 : r - 1);
                const int bias = (y & 1) ? 2 : 1;
                for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            }
        } else {  // other integer ratios: replication
            for (int y = 0; y < height; ++y) {
                const uint8_t* in = row(y / fy);
                for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = in[std::min(x / fx, n - 1)];
            }
        }
        return out;
    }

    RgbImage finish() {
        idct_all();
        RgbImage img((uint32_t)width, (uint32_t)height);
        const size_t npx = (size_t)width * height;
        if (comps.size() == 1) {
            const std::vector<uint8_t> y = upsample(comps[0]);
            for (size_t i = 0; i < npx; ++i) img.data[3 * i] = img.data[3 * i + 1] = img.data[3 * i + 2] = y[i];
            return img;
        }
        const std::vector<uint8_t> p0 = upsample(comps[0]), p1 = upsample(comps[1]), p2 = upsample(comps[2]);
        // Adobe transform 0, or component ids 'R','G','B' without a JFIF / Adobe marker: the data is RGB already
        const bool rgb = adobe_transform == 0 || (adobe_transform < 0 && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
        if (rgb) {
            for (size_t i = 0; i < npx; ++i) img.data[3 * i] = p0[i], img.data[3 * i + 1] = p1[i], img.data[3 * i + 2] = p2[i];
            return img;
        }
        // libjpeg's ycc_rgb_convert tables (16-bit fixed point)
        int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
        for (int i = 0; i < 256; ++i) {
            const int x = i - 128;
            cr_r[i] = (int)((91881L * x + 32768) >> 16);     // 1.40200
            cb_b[i] = (int)((116130L * x + 32768) >> 16);    // 1.77200
            cr_g[i] = (int)(-46802L * x);                    // 0.71414
            cb_g[i] = (int)(-22554L * x + 32768);            // 0.34414, with the rounding term
        }
        auto clamp = [](int v) { return (uint8_t)std::min(255, std::max(0, v)); };
        for (size_t i = 0; i < npx; ++i) {
            const int y = p0[i], cb = p1[i], cr = p2[i];
            img.data[3 * i] = clamp(y + cr_r[cr]);
            img.data[3 * i + 1] = clamp(y + ((cb_g[cb] + cr_g[cr]) >> 16));
            img.data[3 * i + 2] = clamp(y + cb_b[cb]);
        }
        return img;
    }

    RgbImage run() {
        if (file.size() < 4 || file[0] != 0xff || file[1] != 0xd8) fail("not a JPEG file");
        size_t pos = 2;
        bool done = false;
        while (!done) {
            while (pos < file.size() && file[pos] != 0xff) ++pos;  // garbage between segments
            while (pos < file.size() && file[pos] == 0xff) ++pos;  // fill bytes
            if (pos >= file.size()) break;
            const int marker = file[pos++];
            if (marker == 0xd9) break;                                                       // EOI
            if (marker == 0x01 || (marker >= 0xd0 && marker <= 0xd7) || marker == 0xd8) continue;  // no payload
            const size_t len = be16(pos);
            if (len < 2 || pos + len > file.size()) fail("truncated JPEG segment");
            const size_t body = pos + 2, end = pos + len;
            switch (marker) {
                case 0xc0:
                case 0xc1: parse_sof(body, end, false); break;
                case 0xc2: parse_sof(body, end, true); break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    fail("lossless, hierarchical and arithmetic-coded JPEG are not supported");
                case 0xc4: parse_dht(body, end); break;
                case 0xdb: parse_dqt(body, end); break;
                case 0xdd:
                    if (len < 4) fail("bad DRI");
                    restart_interval = be16(body);
                    break;
                case 0xe1:
                    if (exif.empty() && len >= 8 && !std::memcmp(&file[body], "Exif\0\0", 6)) exif.assign(file.data() + body + 6, file.data() + end);
                    break;
                case 0xee:
                    if (len >= 14 && !std::memcmp(&file[body], "Adobe", 5)) adobe_transform = file[body + 11];
                    break;
                case 0xda:
                    pos = parse_sos(body, end);
                    continue;
                default: break;  // APPn, COM, ...
            }
            pos = end;
        }
        if (!have_frame) fail("JPEG without a frame header");
        return finish();
    }
};

}  // namespace

RgbImage decode_jpeg(const std::vector<uint8_t>& file, const std::string& path, std::vector<uint8_t>* exif) {
    Decoder d(file, path);
    RgbImage img = d.run();
    if (exif) *exif = std::move(d.exif);
    return img;
}

}  // namespace matrix_eyes
