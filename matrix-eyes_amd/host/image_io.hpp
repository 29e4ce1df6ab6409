// Image file I/O and resampling for the C++ host layer (the reference uses the `image` crate:
// reconstruction.rs:96-113, output.rs:133-138,192,206-218).  PNG (8/16-bit, non-interlaced) and binary PPM
// are decoded and encoded here with zlib only; anything else is an ImageError, as an unsupported format is
// in the reference.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace matrix_eyes {

struct ImageError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct RgbImage {  // image::RgbImage: row-major, 3 bytes per pixel
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
    RgbImage() = default;
    RgbImage(uint32_t w, uint32_t h) : width(w), height(h), data((size_t)w * h * 3) {}
};

RgbImage load_image(const std::string& path);                     // ImageReader::open(..).decode().into_rgb8()
void save_image(const RgbImage& img, const std::string& path);    // RgbImage::save: format from the extension
// DynamicImage::resize_exact(w, h, FilterType::Lanczos3); the identity when the size already matches
RgbImage resize_exact_lanczos3(const RgbImage& img, uint32_t width, uint32_t height);

}  // namespace matrix_eyes
