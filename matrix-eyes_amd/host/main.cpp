// Command-line twin of the reference's src/main.rs: the same `--name=value` flags, defaults, messages and
// exit codes (0 ok / help, 1 reconstruction failed, 2 usage), driving the MI355X back end.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <optional>
#include <string>

#include "matrix_eyes.hpp"

using matrix_eyes::output::ImageOutputFormat;
using matrix_eyes::output::VertexMode;

namespace {

const char* kUsage =
    "Usage: matrix-eyes [OPTIONS] <IMG_SRC>... <IMG_OUT>\n\n"
    "Arguments:\n"
    "  <IMG_SRC>...  Source image\n"
    "  <IMG_OUT>     Output image\n\n"
    "Options:\n"
    "      --focal-length=<FOCAL_LENGTH>       Focal length in 35mm equivalent\n"
    "      --checkpoint-path=<CHECKPOINT_PATH> Path to checkpoint file [default: ./checkpoints/depth_pro.pt]\n"
    "      --image-output-format=<FORMAT>      Format for output [default: depthmap] [possible values: depthmap, stereogram]\n"
    "      --resize-scale=<SCALE>              Custom scale for stereogram output [default: 1.0]\n"
    "      --stereo-amplitude=<AMPLITUDE>      Custom scale for stereogram output [default: 0.0625]\n"
    "      --mesh=<MESH>                       Mesh options [default: vertex-colors] [possible values: plain, vertex-colors, texture-coordinates]\n"
    "      --convert-checkpoints               Convert checkpoints into a more efficient format [default: disabled]\n"
    "      --help                              Print help";

struct Args {  // main.rs:12-20
    std::optional<float> focal_length;
    std::string checkpoint_path = "./checkpoints/depth_pro.pt";
    bool convert_checkpoints = false;
    ImageOutputFormat output_format = ImageOutputFormat::depth_map();
    VertexMode vertex_mode = VertexMode::Color;  // the code's default (main.rs:43), whatever the help says
    std::string img_src, img_out;
};

[[noreturn]] void usage_exit(const std::string& message) {
    std::fprintf(stderr, "%s\n", message.c_str());
    std::printf("%s\n", kUsage);
    std::exit(2);
}

float parse_float(const std::string& name, const std::string& value) {
    char* end = nullptr;
    const float v = std::strtof(value.c_str(), &end);
    if (value.empty() || *end != '\0') usage_exit("Argument " + name + " has an unsupported value " + value + ": invalid float literal");
    return v;
}

std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}

Args parse(int argc, char** argv) {  // main.rs:37-146
    Args args;
    std::optional<float> resize_scale;
    float stereo_amplitude = 1.0f / 16.0f;
    bool stereogram = false;
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg.rfind("--", 0) == 0 && args.img_src.empty() && args.img_out.empty()) {
            if (arg == "--convert-checkpoints") {
                args.convert_checkpoints = true;
                continue;
            } else if (arg == "--help") {
                std::printf("%s\n", kUsage);
                std::exit(0);
            }
            const size_t eq = arg.find('=');
            if (eq == std::string::npos) usage_exit("Option flag " + arg + " has no value");
            const std::string name = arg.substr(0, eq), value = arg.substr(eq + 1);
            if (name == "--focal-length") {
                args.focal_length = parse_float(name, value);
            } else if (name == "--image-output-format") {
                const std::string v = lower(value);
                if (v == "depthmap")
                    stereogram = false;
                else if (v == "stereogram")
                    stereogram = true;
                else
                    usage_exit("Unsupported output format " + value);
            } else if (name == "--resize-scale") {
                resize_scale = parse_float(name, value);
            } else if (name == "--stereo-amplitude") {
                stereo_amplitude = parse_float(name, value);
            } else if (name == "--mesh") {
                const std::string v = lower(value);
                if (v == "plain")
                    args.vertex_mode = VertexMode::Plain;
                else if (v == "vertex-colors")
                    args.vertex_mode = VertexMode::Color;
                else if (v == "texture-coordinates")
                    args.vertex_mode = VertexMode::Texture;
                else
                    usage_exit("Unsupported mesh vertex output mode " + value);
            } else if (name == "--checkpoint-path") {
                args.checkpoint_path = value;
            } else {
                std::fprintf(stderr, "Unsupported argument %s\n", arg.c_str());  // not fatal in the reference
            }
        } else if (args.img_src.empty()) {
            args.img_src = arg;
        } else if (args.img_out.empty()) {
            args.img_out = arg;
        } else {
            usage_exit("Unexpected argument " + arg);
        }
    }
    if (stereogram) args.output_format = ImageOutputFormat::stereogram(resize_scale, stereo_amplitude);
    if (args.img_src.empty()) usage_exit("No source image provided");
    if (args.img_out.empty()) usage_exit("No output image provided");
    return args;
}

}  // namespace

int main(int argc, char** argv) {
    std::printf("Matrix Eyes version %s\n", "0.1.0-hip");
    const Args args = parse(argc, argv);
    try {
        const matrix_eyes::Device device;  // reconstruction::init_device()
        const matrix_eyes::DepthProModelLoader model_loader(args.checkpoint_path, args.convert_checkpoints);
        matrix_eyes::reconstruction::extract_depth(device, model_loader, args.img_src, args.img_out, args.focal_length,
                                                   args.output_format, args.vertex_mode);
    } catch (const std::exception& err) {
        std::printf("Reconstruction failed: %s\n", err.what());
        return 1;
    }
    return 0;
}
