// Image file I/O and resampling for the C++ host layer (the reference uses the `image` and `kamadak-exif`
// crates: reconstruction.rs:96-113,133-143, output.rs:133-138,192,206-218).  Decoded here: JPEG (baseline and
// progressive Huffman, jpeg_decoder.cpp), PNG (8/16-bit, non-interlaced, zlib) and binary PPM; encoded: PNG
// and PPM.  Anything else is an ImageError, as an unsupported format is in the reference.
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace matrix_eyes {

struct ImageError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

constexpr int64_t kMaxPixels = 1ll << 28;  // decoders refuse larger pictures instead of allocating for them

struct RgbImage {  // image::RgbImage: row-major, 3 bytes per pixel
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
    RgbImage() = default;
    RgbImage(uint32_t w, uint32_t h) : width(w), height(h), data((size_t)w * h * 3) {}
};

// What the reference reads from the file besides the pixels (reconstruction.rs:97-105): the EXIF orientation
// (1..8, 1 = none) and exif::Tag::FocalLengthIn35mmFilm of the primary image.
struct ImageMetadata {
    int orientation = 1;
    std::optional<uint32_t> focal_length_35mm;
};

// ImageReader::open(..).into_decoder() + DynamicImage::from_decoder(..).into_rgb8(); the orientation is NOT
// applied here (the reference applies it as a separate step)
RgbImage load_image(const std::string& path, ImageMetadata* metadata = nullptr);
// DynamicImage::apply_orientation with the EXIF value: 2 flip horizontally, 3 rotate 180, 4 flip vertically,
// 5 rotate 90 + flip horizontally, 6 rotate 90, 7 rotate 270 + flip horizontally, 8 rotate 270 (clockwise)
RgbImage apply_orientation(const RgbImage& img, int orientation);
// the raw TIFF-structured EXIF block (after "Exif\0\0" in a JPEG APP1 segment, or a PNG eXIf chunk)
ImageMetadata parse_exif(const std::vector<uint8_t>& exif);
RgbImage decode_jpeg(const std::vector<uint8_t>& file, const std::string& path, std::vector<uint8_t>* exif);
void save_image(const RgbImage& img, const std::string& path);    // RgbImage::save: format from the extension
// DynamicImage::resize_exact(w, h, FilterType::Lanczos3); the identity when the size already matches
RgbImage resize_exact_lanczos3(const RgbImage& img, uint32_t width, uint32_t height);

}  // namespace matrix_eyes
