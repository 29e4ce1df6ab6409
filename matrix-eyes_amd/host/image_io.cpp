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

RgbImage decode_png(const std::vector<uint8_t>& file, const std::string& path) {
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
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (!width || !height || ctype < 0) throw ImageError(path + ": no IHDR");
    if (interlace) throw ImageError(path + ": interlaced PNG is not supported");
    if (depth != 8 && depth != 16) throw ImageError(path + ": PNG bit depth " + std::to_string(depth) + " is not supported");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: throw ImageError(path + ": bad PNG colour type");
    }
    if (ctype == 3 && depth != 8) throw ImageError(path + ": palette PNG must be 8-bit here");
    const size_t bpp = (size_t)channels * depth / 8, stride = (size_t)width * bpp;
    std::vector<uint8_t> raw((stride + 1) * height);
    uLongf raw_len = raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), idat.size()) != Z_OK || raw_len != raw.size())
        throw ImageError(path + ": PNG data does not inflate to the image size");
    // undo the scanline filters in place
    std::vector<uint8_t> prev(stride, 0);
    RgbImage img(width, height);
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t* row = &raw[y * (stride + 1) + 1];
        const int filter = row[-1];
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
        uint8_t* out = &img.data[(size_t)y * width * 3];
        const size_t step = depth / 8;  // 16-bit samples: keep the high byte
        for (uint32_t x = 0; x < width; ++x) {
            const uint8_t* px = row + x * bpp;
            switch (ctype) {
                case 0:
                case 4: out[0] = out[1] = out[2] = px[0]; break;
                case 2:
                case 6: out[0] = px[0], out[1] = px[step], out[2] = px[2 * step]; break;
                case 3: {
                    const size_t k = (size_t)px[0] * 3;
                    if (k + 2 >= plte.size()) throw ImageError(path + ": palette index out of range");
                    out[0] = plte[k], out[1] = plte[k + 1], out[2] = plte[k + 2];
                    break;
                }
            }
            out += 3;
        }
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

RgbImage load_image(const std::string& path) {
    const std::vector<uint8_t> file = read_file(path);
    if (file.size() >= 8 && !std::memcmp(file.data(), kPngSig, 8)) return decode_png(file, path);
    if (file.size() >= 2 && file[0] == 'P' && file[1] == '6') return decode_ppm(file, path);
    throw ImageError(path + ": unsupported image format (this host layer decodes PNG and binary PPM)");
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
