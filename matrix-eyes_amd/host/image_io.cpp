#include "image_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace matrix_eyes {

namespace {

std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ImageError("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

bool ends_with_ci(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
        if (std::tolower((unsigned char)s[s.size() - n + i]) != suffix[i]) return false;
    return true;
}

// ---- PNG -------------------------------------------------------------------------------------------
const uint8_t kPngSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

RgbImage decode_png(const std::vector<uint8_t>& file, const std::string& path, std::vector<uint8_t>* exif) {
    size_t pos = 8;
    uint32_t width = 0, height = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        if (pos + 12 + len > file.size()) throw ImageError(path + ": truncated PNG chunk");
        const uint8_t* body = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) throw ImageError(path + ": bad IHDR");
            width = be32(body), height = be32(body + 4);
            depth = body[8], ctype = body[9], interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "eXIf", 4)) {
            if (exif) exif->assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (!width || !height || ctype < 0) throw ImageError(path + ": no IHDR");
    if ((int64_t)width * height > kMaxPixels) throw ImageError(path + ": PNG larger than 2^28 pixels");
    if (interlace > 1) throw ImageError(path + ": bad PNG interlace method");
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)
        throw ImageError(path + ": bad PNG bit depth " + std::to_string(depth));
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: throw ImageError(path + ": bad PNG colour type");
    }
    // PNG spec table 11.1: sub-byte depths for grey and palette only, 16 bit not for palette
    if ((depth < 8 && ctype != 0 && ctype != 3) || (depth == 16 && ctype == 3))
        throw ImageError(path + ": PNG bit depth " + std::to_string(depth) + " does not go with colour type " +
                         std::to_string(ctype));
    const size_t bits_pp = (size_t)channels * depth;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;  // the filters' "corresponding byte" distance
    auto row_bytes = [&](uint32_t w) { return ((size_t)w * bits_pp + 7) / 8; };
    // Adam7 passes (x0, y0, dx, dy); a non-interlaced image is one pass over every pixel
    static const int kAdam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4},
                                     {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int kWhole[1][4] = {{0, 0, 1, 1}};
    const int (*passes)[4] = interlace ? kAdam7 : kWhole;
    const int npass = interlace ? 7 : 1;
    size_t total = 0;
    for (int ps = 0; ps < npass; ++ps) {
        const uint32_t pw = (width + passes[ps][2] - 1 - passes[ps][0]) / passes[ps][2];
        const uint32_t ph = (height + passes[ps][3] - 1 - passes[ps][1]) / passes[ps][3];
        if (pw && ph) total += (row_bytes(pw) + 1) * ph;
    }
    std::vector<uint8_t> raw(total);
    uLongf raw_len = raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), idat.size()) != Z_OK || raw_len != raw.size())
        throw ImageError(path + ": PNG data does not inflate to the image size");
    RgbImage img(width, height);
    // samples -> 8 bit the way the reference's decoder stack does: grey of 1 / 2 / 4 bits scaled to the full
    // range (x255, x85, x17), 16-bit samples rounded ((v + 128) / 257), alpha dropped (into_rgb8)
    const int grey_scale = depth < 8 ? 255 / ((1 << depth) - 1) : 1;
    auto sample8 = [&](const uint8_t* row, uint32_t x, int c) -> int {  // channel c of pixel x, depth >= 8
        if (depth == 8) return row[(size_t)x * channels + c];
        const uint8_t* p = row + ((size_t)x * channels + c) * 2;
        return (((int)p[0] << 8 | p[1]) + 128) / 257;
    };
    auto packed = [&](const uint8_t* row, uint32_t x) -> int {  // depth < 8: one channel, big-endian bit order
        const size_t bit = (size_t)x * depth;
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
    };
    size_t off = 0;
    for (int ps = 0; ps < npass; ++ps) {
        const int x0 = passes[ps][0], y0 = passes[ps][1], dx = passes[ps][2], dy = passes[ps][3];
        const uint32_t pw = (width + dx - 1 - x0) / dx, ph = (height + dy - 1 - y0) / dy;
        if (!pw || !ph) continue;
        const size_t stride = row_bytes(pw);
        std::vector<uint8_t> prev(stride, 0);
        for (uint32_t py = 0; py < ph; ++py) {
            uint8_t* row = &raw[off + py * (stride + 1) + 1];
            const int filter = row[-1];
            // undo the scanline filter in place
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= bpp ? row[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int v = row[i];
                switch (filter) {
                    case 0: break;
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) / 2; break;
                    case 4: v += paeth(a, b, c); break;
                    default: throw ImageError(path + ": bad PNG filter");
                }
                row[i] = (uint8_t)v;
            }
            std::memcpy(prev.data(), row, stride);
            const uint32_t y = y0 + py * dy;
            for (uint32_t px = 0; px < pw; ++px) {
                uint8_t* out = &img.data[((size_t)y * width + x0 + (size_t)px * dx) * 3];
                switch (ctype) {
                    case 0: out[0] = out[1] = out[2] = (uint8_t)(depth < 8 ? packed(row, px) * grey_scale : sample8(row, px, 0)); break;
                    case 4: out[0] = out[1] = out[2] = (uint8_t)sample8(row, px, 0); break;
                    case 2:
                    case 6:
                        out[0] = (uint8_t)sample8(row, px, 0), out[1] = (uint8_t)sample8(row, px, 1),
                        out[2] = (uint8_t)sample8(row, px, 2);
                        break;
                    case 3: {
                        const size_t k = (size_t)(depth < 8 ? packed(row, px) : row[px]) * 3;
                        if (k + 2 >= plte.size()) throw ImageError(path + ": palette index out of range");
                        out[0] = plte[k], out[1] = plte[k + 1], out[2] = plte[k + 2];
                        break;
                    }
                }
            }
        }
        off += (stride + 1) * ph;
    }
    return img;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(x >> 24), v.push_back(x >> 16), v.push_back(x >> 8), v.push_back(x);
}

void put_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body) {
    put_be32(out, (uint32_t)body.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put_be32(out, (uint32_t)crc32(0, &out[start], (uInt)(out.size() - start)));
}

void encode_png(const RgbImage& img, const std::string& path) {
    std::vector<uint8_t> out(kPngSig, kPngSig + 8), ihdr;
    put_be32(ihdr, img.width), put_be32(ihdr, img.height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8-bit RGB, no interlace
    put_chunk(out, "IHDR", ihdr);
    const size_t stride = (size_t)img.width * 3;
    std::vector<uint8_t> raw((stride + 1) * img.height);
    for (uint32_t y = 0; y < img.height; ++y) {  // filter 0: the payload is noise or a smooth colour ramp
        raw[y * (stride + 1)] = 0;
        std::memcpy(&raw[y * (stride + 1) + 1], &img.data[y * stride], stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw ImageError("zlib failed");
    comp.resize(clen);
    put_chunk(out, "IDAT", comp);
    put_chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f.write((const char*)out.data(), (std::streamsize)out.size())) throw ImageError("cannot write " + path);
}

// ---- PPM (P6) --------------------------------------------------------------------------------------
RgbImage decode_ppm(const std::vector<uint8_t>& file, const std::string& path) {
    size_t pos = 2;
    auto next_int = [&]() -> long {
        for (;;) {
            while (pos < file.size() && std::isspace(file[pos])) ++pos;
            if (pos < file.size() && file[pos] == '#') {
                while (pos < file.size() && file[pos] != '\n') ++pos;
                continue;
            }
            break;
        }
        long v = 0;
        bool any = false;
        while (pos < file.size() && std::isdigit(file[pos])) v = v * 10 + (file[pos++] - '0'), any = true;
        if (!any) throw ImageError(path + ": bad PPM header");
        return v;
    };
    const long w = next_int(), h = next_int(), maxv = next_int();
    ++pos;  // the single whitespace byte after maxval
    if (w <= 0 || h <= 0 || maxv != 255) throw ImageError(path + ": only 8-bit PPM is supported");
    if ((int64_t)w * h > kMaxPixels) throw ImageError(path + ": PPM larger than 2^28 pixels");
    RgbImage img((uint32_t)w, (uint32_t)h);
    if (pos + img.data.size() > file.size()) throw ImageError(path + ": truncated PPM");
    std::memcpy(img.data.data(), &file[pos], img.data.size());
    return img;
}

// ---- Lanczos3 --------------------------------------------------------------------------------------
float sinc(float t) {
    if (t == 0.0f) return 1.0f;
    const float a = t * 3.14159265358979323846f;
    return std::sin(a) / a;
}
float lanczos3(float x) { return std::fabs(x) < 3.0f ? sinc(x) * sinc(x / 3.0f) : 0.0f; }

// One separable pass over `len_in` samples per line; layout given by (pixel stride, line stride).
// Follows the image crate's sampler: kernel scaled by max(ratio, 1), support 3, weights renormalised,
// f32 accumulation, clamp and round on output.
void resample(const std::vector<float>& in, std::vector<float>& out, uint32_t len_in, uint32_t len_out,
              uint32_t lines, size_t in_px, size_t in_line, size_t out_px, size_t out_line) {
    const float ratio = (float)len_in / (float)len_out, sratio = ratio < 1.0f ? 1.0f : ratio;
    const float support = 3.0f * sratio;
    std::vector<float> w;
    for (uint32_t o = 0; o < len_out; ++o) {
        const float center = ((float)o + 0.5f) * ratio;
        long left = (long)std::floor(center - support), right = (long)std::ceil(center + support);
        left = std::max(0l, std::min<long>(left, (long)len_in - 1));
        right = std::max(left + 1, std::min<long>(right, (long)len_in));
        w.assign((size_t)(right - left), 0.0f);
        float sum = 0.0f;
        for (long i = left; i < right; ++i) sum += (w[(size_t)(i - left)] = lanczos3(((float)i - (center - 0.5f)) / sratio));
        for (float& v : w) v /= sum;
        for (uint32_t l = 0; l < lines; ++l)
            for (int c = 0; c < 3; ++c) {
                float acc = 0.0f;
                for (long i = left; i < right; ++i) acc += in[l * in_line + (size_t)i * in_px + c] * w[(size_t)(i - left)];
                out[l * out_line + o * out_px + c] = acc;
            }
    }
}

}  // namespace

// ---- EXIF ---------------------------------------------------------------------------------------------
// TIFF structure: byte order mark, offset of IFD0; 12-byte entries (tag, type, count, value / offset).
// Orientation (0x0112) lives in IFD0, FocalLengthIn35mmFilm (0xA405) in the Exif sub-IFD (0x8769).
ImageMetadata parse_exif(const std::vector<uint8_t>& e) {
    ImageMetadata m;
    if (e.size() < 8) return m;
    const bool le = e[0] == 'I' && e[1] == 'I';
    if (!le && !(e[0] == 'M' && e[1] == 'M')) return m;
    auto u16 = [&](size_t p) -> uint32_t { return p + 2 <= e.size() ? (le ? e[p] | (e[p + 1] << 8) : (e[p] << 8) | e[p + 1]) : 0u; };
    auto u32 = [&](size_t p) -> uint32_t {
        if (p + 4 > e.size()) return 0u;
        return le ? (uint32_t)e[p] | (e[p + 1] << 8) | (e[p + 2] << 16) | ((uint32_t)e[p + 3] << 24)
                  : ((uint32_t)e[p] << 24) | (e[p + 1] << 16) | (e[p + 2] << 8) | e[p + 3];
    };
    if (u16(2) != 42) return m;
    // value of an entry as an unsigned integer (BYTE, SHORT or LONG, first element)
    auto entry_uint = [&](size_t p, uint32_t* out) {
        const uint32_t type = u16(p + 2), count = u32(p + 4);
        if (count < 1) return false;
        if (type == 1) *out = e[p + 8];
        else if (type == 3) *out = u16(p + 8);
        else if (type == 4) *out = u32(p + 8);
        else return false;
        return true;
    };
    auto walk = [&](size_t ifd, auto&& on_entry) {
        if (ifd == 0 || ifd + 2 > e.size()) return;
        const uint32_t n = u16(ifd);
        for (uint32_t i = 0; i < n; ++i) {
            const size_t p = ifd + 2 + 12 * (size_t)i;
            if (p + 12 > e.size()) return;
            on_entry(u16(p), p);
        }
    };
    size_t exif_ifd = 0;
    walk(u32(4), [&](uint32_t tag, size_t p) {
        uint32_t v;
        if (tag == 0x0112 && entry_uint(p, &v) && v >= 1 && v <= 8) m.orientation = (int)v;
        if (tag == 0x8769 && entry_uint(p, &v)) exif_ifd = v;
        if (tag == 0xa405 && entry_uint(p, &v)) m.focal_length_35mm = v;
    });
    walk(exif_ifd, [&](uint32_t tag, size_t p) {
        uint32_t v;
        if (tag == 0xa405 && entry_uint(p, &v)) m.focal_length_35mm = v;
    });
    return m;
}

RgbImage apply_orientation(const RgbImage& img, int orientation) {
    if (orientation <= 1 || orientation > 8) return img;
    const uint32_t w = img.width, h = img.height;
    const bool swap = orientation >= 5;
    RgbImage out(swap ? h : w, swap ? w : h);
    for (uint32_t y = 0; y < out.height; ++y)
        for (uint32_t x = 0; x < out.width; ++x) {
            uint32_t sx, sy;  // source pixel of output (x, y)
            switch (orientation) {
                case 2: sx = w - 1 - x, sy = y; break;          // flip horizontally
                case 3: sx = w - 1 - x, sy = h - 1 - y; break;  // rotate 180
                case 4: sx = x, sy = h - 1 - y; break;          // flip vertically
                case 5: sx = y, sy = x; break;                  // transpose
                case 6: sx = y, sy = h - 1 - x; break;          // rotate 90 clockwise
                case 7: sx = w - 1 - y, sy = h - 1 - x; break;  // transverse
                default: sx = w - 1 - y, sy = x; break;         // 8: rotate 270 clockwise
            }
            std::memcpy(&out.data[((size_t)y * out.width + x) * 3], &img.data[((size_t)sy * w + sx) * 3], 3);
        }
    return out;
}

RgbImage load_image(const std::string& path, ImageMetadata* metadata) {
    const std::vector<uint8_t> file = read_file(path);
    std::vector<uint8_t> exif;
    RgbImage img;
    if (file.size() >= 8 && !std::memcmp(file.data(), kPngSig, 8))
        img = decode_png(file, path, &exif);
    else if (file.size() >= 3 && file[0] == 0xff && file[1] == 0xd8 && file[2] == 0xff)
        img = decode_jpeg(file, path, &exif);
    else if (file.size() >= 2 && file[0] == 'P' && file[1] == '6')
        img = decode_ppm(file, path);
    else
        throw ImageError(path + ": unsupported image format (this host layer decodes JPEG, PNG and binary PPM)");
    if (metadata) *metadata = parse_exif(exif);
    return img;
}

void save_image(const RgbImage& img, const std::string& path) {
    if (ends_with_ci(path, ".png")) return encode_png(img, path);
    if (ends_with_ci(path, ".ppm")) {
        std::ofstream f(path, std::ios::binary);
        f << "P6\n" << img.width << " " << img.height << "\n255\n";
        if (!f.write((const char*)img.data.data(), (std::streamsize)img.data.size())) throw ImageError("cannot write " + path);
        return;
    }
    throw ImageError(path + ": unsupported output image format (this host layer encodes .png and .ppm)");
}

RgbImage resize_exact_lanczos3(const RgbImage& img, uint32_t width, uint32_t height) {
    if (img.width == width && img.height == height) return img;
    if (!width || !height || !img.width || !img.height) throw ImageError("resize to or from an empty image");
    std::vector<float> src(img.data.begin(), img.data.end());
    // vertical pass first, then horizontal (the order the image crate uses)
    std::vector<float> mid((size_t)img.width * height * 3);
    resample(src, mid, img.height, height, img.width, (size_t)img.width * 3, 3, (size_t)img.width * 3, 3);
    std::vector<float> dst((size_t)width * height * 3);
    resample(mid, dst, img.width, width, height, 3, (size_t)img.width * 3, 3, (size_t)width * 3);
    RgbImage out(width, height);
    for (size_t i = 0; i < dst.size(); ++i) out.data[i] = (uint8_t)std::lround(std::min(255.0f, std::max(0.0f, dst[i])));
    return out;
}

}  // namespace matrix_eyes
