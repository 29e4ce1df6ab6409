// C++ host layer above the C ABI of include/matrix_eyes_hip.h: the reference's Rust surface for the hot
// path, type for type, so that main.cpp reads like the reference's main.rs.  (No Rust toolchain exists in
// the build image; the Rust binding itself is sketched in INTEGRATION.md.)
//
//   reference                                             here
//   reconstruction::init_device()           (:42-72)      Device
//   depth_pro::DepthProModelLoader::new     (mod.rs:167)  DepthProModelLoader(path, convert)
//   DepthProModelLoader::extract_depth      (mod.rs:251)  DepthProModelLoader::extract_depth
//   depth_pro::ProgressListener             (mod.rs:366)  ProgressListener
//   depth_pro::ModelError / LoaderError     (mod.rs:430-)  ModelError
//   output::{ImageOutputFormat, VertexMode} (:27-38)      output::ImageOutputFormat, output::VertexMode
//   output::DepthMap::{new, output_image}   (:44,100)     output::DepthMap
//   output::OutputError                     (:716-)       output::OutputError
//   reconstruction::extract_depth           (:155-205)    reconstruction::extract_depth
//   reconstruction::ReconstructionError     (:240-)       reconstruction::ReconstructionError
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "image_io.hpp"

struct me_ctx;

namespace matrix_eyes {

constexpr int IMG_SIZE = 1536;  // depth_pro::IMG_SIZE

// depth_pro::ModelError::Internal(msg, LoaderError): "Model error: {msg}: {err}"
struct ModelError : std::runtime_error {
    int code;  // the C ABI status (ME_ERR_*)
    ModelError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

struct ProgressListener {  // mod.rs:366-372
    virtual ~ProgressListener() = default;
    virtual void report_status(float pos) = 0;
    virtual void update_message(const std::string& status_message) = 0;
};

// reconstruction::init_device(): the MI355X back end has one kind of device; MATRIX_EYES_DEVICE picks the
// GPU ordinal, MATRIX_EYES_DTYPE=bf16 the MFMA operand type (default f16: the fp16 checkpoint bit for bit).
class Device {
public:
    Device();
    ~Device();
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    me_ctx* ctx() const { return ctx_; }
    int image_size() const { return image_size_; }  // IMG_SIZE unless a test model is selected
    // a caller that writes one mesh per image of a batch: .obj files written behind it by up to `files_in_flight` host
    // threads (me_ctx_set_write_behind); flush_outputs() before the files are read -- a failed write surfaces there
    void set_write_behind(int files_in_flight) const;
    void flush_outputs() const;

private:
    me_ctx* ctx_ = nullptr;
    int image_size_ = IMG_SIZE;
    mutable bool weights_loaded_ = false;
    friend class DepthProModelLoader;
};

class DepthProModelLoader {
public:
    // convert_checkpoints: the reference caches a Burn-private .mpk next to the checkpoint (mod.rs:211-227);
    // the packed weight arena lives in HBM for the life of the Device instead, so the flag is accepted and
    // has nothing to do.
    DepthProModelLoader(const std::string& checkpoint_path, bool convert_checkpoints)
        : checkpoint_path_(checkpoint_path), convert_checkpoints_(convert_checkpoints) {}

    // img: RGB8 [size x size] (the normalise / permute of reconstruction.rs:116-124 happens on the GPU);
    // returns the inverse depth [size*size], clamped to [1e-4, 1e4] (mod.rs:340-362)
    std::vector<float> extract_depth(const Device& device, const RgbImage& img, std::optional<float> f_norm,
                                     ProgressListener* pl) const;

private:
    void ensure_loaded(const Device& device) const;
    std::string checkpoint_path_;
    bool convert_checkpoints_;
};

namespace output {

enum class VertexMode { Plain, Color, Texture };  // output.rs:33-38

struct ImageOutputFormat {  // output.rs:27-31
    enum Kind { DepthMap, Stereogram } kind = DepthMap;
    std::optional<float> resize_scale;  // Stereogram(resize_scale, amplitude)
    float amplitude = 1.0f / 16.0f;
    static ImageOutputFormat depth_map() { return ImageOutputFormat{}; }
    static ImageOutputFormat stereogram(std::optional<float> resize_scale, float amplitude) {
        ImageOutputFormat f;
        f.kind = Stereogram, f.resize_scale = resize_scale, f.amplitude = amplitude;
        return f;
    }
};

struct OutputError : std::runtime_error {  // output.rs:716-
    using std::runtime_error::runtime_error;
};

class DepthMap {
public:
    // DepthMap::new (output.rs:44-67): inverse_depth [rows x cols], clamped to [1/250, 1/0.1]
    DepthMap(const Device& device, std::vector<float> inverse_depth, size_t rows, size_t cols,
             uint32_t original_width, uint32_t original_height);
    std::pair<float, float> inverse_depth_range() const { return {min_, max_}; }  // :69-75
    // output.rs:100-121: .ply / .obj by suffix, else the image format
    void output_image(const std::string& destination_path, const std::string& source_path,
                      ImageOutputFormat image_format, VertexMode vertex_mode) const;

private:
    void output_depth_map(const std::string& destination_path) const;                        // :123-139
    void output_stereogram(const std::string& destination_path, std::optional<float> resize_scale,
                           float amplitude) const;                                          // :141-193
    void output_mesh(const std::string& destination_path, const std::string& source_path,
                     VertexMode mode) const;                                                // :195-261
    const Device& device_;
    std::vector<float> data_;
    size_t data_width_, data_height_;  // names as in the reference (rows, cols of the tensor)
    uint32_t original_width_, original_height_;
    float min_ = 0.f, max_ = 0.f;
};

}  // namespace output

namespace reconstruction {

struct ReconstructionError : std::runtime_error {  // reconstruction.rs:240-249
    using std::runtime_error::runtime_error;
};

struct SourceImage {  // reconstruction.rs:74-81
    RgbImage img;                              // size x size
    uint32_t original_width = 0, original_height = 0;
    std::optional<float> focal_length_35mm;
    static SourceImage load(const std::string& path, std::optional<float> focal_length_35mm, int size);  // :87-131
    std::optional<double> focal_length_px() const;                                                       // :145-152
};

// reconstruction.rs:155-205
void extract_depth(const Device& device, const DepthProModelLoader& model_loader, const std::string& source_path,
                   const std::string& destination_path, std::optional<float> focal_length_35mm,
                   output::ImageOutputFormat image_format, output::VertexMode vertex_mode);

}  // namespace reconstruction

}  // namespace matrix_eyes
