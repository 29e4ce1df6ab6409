#include "matrix_eyes.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "matrix_eyes_hip.h"

namespace matrix_eyes {

namespace {

std::string last_error(const me_ctx* ctx) { return me_last_error(ctx); }

void check_model(const me_ctx* ctx, int32_t rc, const char* what) {
    if (rc != ME_OK) throw ModelError(rc, std::string("Model error: ") + what + ": " + last_error(ctx));
}

void check_output(const me_ctx* ctx, int32_t rc) {
    if (rc != ME_OK) throw output::OutputError(last_error(ctx));
}

bool ends_with_ci(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
        if (std::tolower((unsigned char)s[s.size() - n + i]) != suffix[i]) return false;
    return true;
}

// f32::round (half away from zero) then `as u32` (saturating)
uint32_t round_u32(float x) {
    if (!(x > 0.0f)) return 0;
    const float r = std::round(x);
    return r >= 4294967296.0f ? 4294967295u : (uint32_t)r;
}

void progress_trampoline(void* user, float pos, const char* message) {
    ProgressListener* pl = (ProgressListener*)user;
    pl->report_status(pos);
    if (message) pl->update_message(message);
}

}  // namespace

// ---- Device ----------------------------------------------------------------------------------------
Device::Device() {
    const char* dev = std::getenv("MATRIX_EYES_DEVICE");
    const char* dt = std::getenv("MATRIX_EYES_DTYPE");
    const char* model = std::getenv("MATRIX_EYES_MODEL");  // "tiny": the test geometry of the parity suite
    me_model_config cfg;
    me_default_config(&cfg);
    if (model && std::strcmp(model, "tiny") == 0) {
        cfg.grid = 8, cfg.embed_dim = 128, cfg.num_heads = 2, cfg.depth = 4;
        cfg.tap_blocks[0] = 1, cfg.tap_blocks[1] = 2;
        cfg.enc_dims[0] = 64, cfg.enc_dims[1] = cfg.enc_dims[2] = cfg.enc_dims[3] = 128;
        cfg.dec_dim = 256, cfg.head_dims[0] = 32, cfg.head_dims[1] = 1;
    }
    image_size_ = 64 * cfg.grid;
    const int32_t dtype = dt && std::strcmp(dt, "bf16") == 0 ? ME_DTYPE_BF16 : (dt && std::strcmp(dt, "fp8") == 0 ? ME_DTYPE_FP8 : ME_DTYPE_F16);
    const int32_t rc = me_ctx_create(dev ? std::atoi(dev) : 0, dtype,
                                     &cfg, &ctx_);
    if (rc != ME_OK) throw ModelError(rc, std::string("cannot initialise the HIP device: ") + me_last_error(nullptr));
}

Device::~Device() {
    if (ctx_) me_ctx_destroy(ctx_);
}

void Device::set_write_behind(int files_in_flight) const {
    if (me_ctx_set_write_behind(ctx_, files_in_flight) != ME_OK) throw output::OutputError(me_last_error(ctx_));
}

void Device::flush_outputs() const {
    if (me_output_flush(ctx_) != ME_OK) throw output::OutputError(me_last_error(ctx_));  // OutputError::Io of a file written behind
}

// ---- DepthProModelLoader -----------------------------------------------------------------------------
void DepthProModelLoader::ensure_loaded(const Device& device) const {
    if (device.weights_loaded_) return;
    me_ctx* ctx = device.ctx();
    // mod.rs:229-249: the library reads the PyTorch archive itself; keys the model does not use are skipped
    const int32_t rc = me_load_checkpoint_pt(ctx, checkpoint_path_.c_str());
    if (rc == ME_ERR_IO)  // LoaderError::Pytorch
        throw ModelError(ME_ERR_IO, std::string("Model error: ") + me_last_error(ctx));
    check_model(ctx, rc, "failed to load checkpoint");
    (void)convert_checkpoints_;
    device.weights_loaded_ = true;
}

std::vector<float> DepthProModelLoader::extract_depth(const Device& device, const RgbImage& img,
                                                      std::optional<float> f_norm, ProgressListener* pl) const {
    const int S = device.image_size();
    if ((int)img.width != S || (int)img.height != S)
        throw ModelError(ME_ERR_BAD_SHAPE, "Model error: image is not " + std::to_string(S) + "x" + std::to_string(S));
    ensure_loaded(device);
    me_ctx* ctx = device.ctx();
    me_ctx_set_progress(ctx, pl ? progress_trampoline : nullptr, pl);
    std::vector<float> depth((size_t)S * S);
    const float fn = f_norm.value_or(0.f);
    const int32_t rc = me_extract_depth_u8(ctx, img.data.data(), 1, f_norm ? &fn : nullptr, depth.data(), nullptr);
    me_ctx_set_progress(ctx, nullptr, nullptr);
    check_model(ctx, rc, "failed to run the model");
    return depth;
}

// ---- output::DepthMap --------------------------------------------------------------------------------
namespace output {

DepthMap::DepthMap(const Device& device, std::vector<float> inverse_depth, size_t rows, size_t cols,
                   uint32_t original_width, uint32_t original_height)
    : device_(device), data_(std::move(inverse_depth)), data_width_(rows), data_height_(cols),
      original_width_(original_width), original_height_(original_height) {
    if (data_.size() != rows * cols) throw OutputError("DepthMap: data does not match its dimensions");
    check_output(device_.ctx(), me_depth_clamp_minmax(device_.ctx(), data_.data(), (int64_t)data_.size(), &min_, &max_));
}

void DepthMap::output_image(const std::string& destination_path, const std::string& source_path,
                            ImageOutputFormat image_format, VertexMode vertex_mode) const {
    if (ends_with_ci(destination_path, ".ply") || ends_with_ci(destination_path, ".obj"))
        return output_mesh(destination_path, source_path, vertex_mode);
    if (image_format.kind == ImageOutputFormat::DepthMap) return output_depth_map(destination_path);
    return output_stereogram(destination_path, image_format.resize_scale, image_format.amplitude);
}

void DepthMap::output_depth_map(const std::string& destination_path) const {
    RgbImage out((uint32_t)data_width_, (uint32_t)data_height_);
    check_output(device_.ctx(), me_depthmap_rgb(device_.ctx(), data_.data(), (int64_t)data_.size(), min_, max_, out.data.data()));
    try {
        save_image(resize_exact_lanczos3(out, original_width_, original_height_), destination_path);
    } catch (const ImageError& err) {
        throw OutputError(err.what());
    }
}

void DepthMap::output_stereogram(const std::string& destination_path, std::optional<float> resize_scale,
                                 float amplitude) const {
    uint32_t w = original_width_, h = original_height_;
    if (resize_scale) {  // output.rs:147-151, f32 arithmetic
        w = round_u32((float)original_width_ * *resize_scale);
        h = round_u32((float)original_height_ * *resize_scale);
    }
    RgbImage noise(w, h), out(w, h);
    // output.rs:165-171: rand::rng() fills one [u8; 3] per pixel, row by row.  MATRIX_EYES_SEED makes a
    // run repeatable (the reference is not).
    const char* seed = std::getenv("MATRIX_EYES_SEED");
    std::mt19937 rng(seed ? (unsigned)std::strtoul(seed, nullptr, 10) : std::random_device{}());
    for (size_t i = 0; i < noise.data.size(); i += 4) {
        const uint32_t v = rng();
        for (size_t k = 0; k < 4 && i + k < noise.data.size(); ++k) noise.data[i + k] = (uint8_t)(v >> (8 * k));
    }
    check_output(device_.ctx(), me_stereogram(device_.ctx(), data_.data(), (int32_t)data_width_, (int32_t)data_height_, min_, max_,
                                              (int32_t)w, (int32_t)h, amplitude, noise.data.data(), out.data.data()));
    try {
        save_image(out, destination_path);
    } catch (const ImageError& err) {
        throw OutputError(err.what());
    }
}

void DepthMap::output_mesh(const std::string& destination_path, const std::string& source_path, VertexMode mode) const {
    std::vector<uint8_t> colors;
    if (mode == VertexMode::Color) {  // output.rs:206-218
        try {
            colors = resize_exact_lanczos3(load_image(source_path), (uint32_t)data_width_, (uint32_t)data_height_).data;
        } catch (const ImageError& err) {
            throw OutputError(err.what());
        }
    }
    check_output(device_.ctx(),
                 me_output_mesh(device_.ctx(), data_.data(), (int32_t)data_width_, (int32_t)data_height_, original_width_,
                                original_height_, destination_path.c_str(), source_path.c_str(),
                                mode == VertexMode::Plain ? ME_VERTEX_PLAIN : (mode == VertexMode::Color ? ME_VERTEX_COLOR : ME_VERTEX_TEXTURE),
                                colors.empty() ? nullptr : colors.data()));
}

}  // namespace output

// ---- reconstruction ------------------------------------------------------------------------------------
namespace reconstruction {

namespace {

// indicatif's bar in the reference (reconstruction.rs:207-238); a single rewritten line here
struct ProgressReporter : ProgressListener {
    std::string message;
    void report_status(float pos) override {
        std::fprintf(stderr, "\r[%3d%%]%s\033[K", (int)std::lround(pos * 100.0f), message.c_str());
        std::fflush(stderr);
    }
    void update_message(const std::string& status_message) override { message = ": " + status_message; }
    ~ProgressReporter() override { std::fprintf(stderr, "\r\033[K"); }
};

}  // namespace

SourceImage SourceImage::load(const std::string& path, std::optional<float> focal_length_35mm, int size) {
    SourceImage s;
    RgbImage img;
    ImageMetadata meta;
    try {
        img = load_image(path, &meta);
    } catch (const ImageError& err) {
        throw ReconstructionError(std::string("Image error: ") + err.what());
    }
    // reconstruction.rs:97-103: the caller's focal length wins, else EXIF FocalLengthIn35mmFilm, else none
    s.focal_length_35mm = focal_length_35mm;
    if (!s.focal_length_35mm && meta.focal_length_35mm) s.focal_length_35mm = (float)*meta.focal_length_35mm;
    img = apply_orientation(img, meta.orientation);  // :104-106
    s.original_width = img.width, s.original_height = img.height;
    s.img = resize_exact_lanczos3(img, (uint32_t)size, (uint32_t)size);
    return s;
}

std::optional<double> SourceImage::focal_length_px() const {
    if (!focal_length_35mm) return std::nullopt;
    const double diagonal_35mm = std::sqrt(24.0 * 24.0 + 36.0 * 36.0);
    const double w = original_width, h = original_height;
    return (double)*focal_length_35mm * std::sqrt(w * w + h * h) / diagonal_35mm;
}

void extract_depth(const Device& device, const DepthProModelLoader& model_loader, const std::string& source_path,
                   const std::string& destination_path, std::optional<float> focal_length_35mm,
                   output::ImageOutputFormat image_format, output::VertexMode vertex_mode) {
    SourceImage img;
    try {
        img = SourceImage::load(source_path, focal_length_35mm, device.image_size());
    } catch (const ReconstructionError& err) {
        std::fprintf(stderr, "Failed to load source image: %s\n", err.what());
        throw;
    }
    std::optional<float> f_norm;
    if (const std::optional<double> f_px = img.focal_length_px()) f_norm = (float)(*f_px / (double)img.original_width);  // :174-176
    std::vector<float> inverse_depth;
    try {
        ProgressReporter pl;
        inverse_depth = model_loader.extract_depth(device, img.img, f_norm, &pl);
    } catch (const ModelError& err) {
        std::fprintf(stderr, "Failed to process image: %s\n", err.what());
        throw ReconstructionError(err.what());
    }
    try {
        const size_t S = (size_t)device.image_size();
        const output::DepthMap depth_map(device, std::move(inverse_depth), S, S, img.original_width, img.original_height);
        depth_map.output_image(destination_path, source_path, image_format, vertex_mode);
    } catch (const output::OutputError& err) {
        std::fprintf(stderr, "Failed to output result: %s\n", err.what());
        throw ReconstructionError(std::string("Output error: ") + err.what());
    }
}

}  // namespace reconstruction

}  // namespace matrix_eyes
