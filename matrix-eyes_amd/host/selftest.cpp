// Test driver for the pieces of the host layer that need no GPU (tests/test_host_cpp.py):
//   host_selftest pt <checkpoint.pt>            one line per tensor: name dtype dims... fnv1a64(data)
//   host_selftest png <in> <out>                decode, re-encode
//   host_selftest resize <in> <w> <h> <out>     Lanczos3 resize_exact
//   host_selftest decode <in> <out> [oriented]  decode any supported format (optionally apply the EXIF
//                                               orientation), write <out>; prints "orientation focal35 w h"
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "image_io.hpp"
#include "pt_reader.hpp"

int main(int argc, char** argv) {
    try {
        if (argc == 3 && !std::strcmp(argv[1], "pt")) {
            matrix_eyes::PtFile f(argv[2]);
            for (const auto& t : f.tensors()) {
                uint64_t h = 1469598103934665603ull;
                const unsigned char* p = (const unsigned char*)t.data;
                for (size_t i = 0; i < t.nbytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
                std::printf("%s %s", t.name.c_str(), t.dtype.c_str());
                for (int64_t d : t.dims) std::printf(" %" PRId64, d);
                std::printf(" %016" PRIx64 "\n", h);
            }
            return 0;
        }
        if (argc == 4 && !std::strcmp(argv[1], "png")) {
            matrix_eyes::save_image(matrix_eyes::load_image(argv[2]), argv[3]);
            return 0;
        }
        if ((argc == 4 || argc == 5) && !std::strcmp(argv[1], "decode")) {
            matrix_eyes::ImageMetadata meta;
            matrix_eyes::RgbImage img = matrix_eyes::load_image(argv[2], &meta);
            if (argc == 5) img = matrix_eyes::apply_orientation(img, meta.orientation);
            matrix_eyes::save_image(img, argv[3]);
            std::printf("%d %ld %u %u\n", meta.orientation, meta.focal_length_35mm ? (long)*meta.focal_length_35mm : -1l,
                        img.width, img.height);
            return 0;
        }
        if (argc == 6 && !std::strcmp(argv[1], "resize")) {
            matrix_eyes::save_image(matrix_eyes::resize_exact_lanczos3(matrix_eyes::load_image(argv[2]), (uint32_t)std::atoi(argv[3]),
                                                                       (uint32_t)std::atoi(argv[4])),
                                    argv[5]);
            return 0;
        }
    } catch (const std::exception& err) {
        std::fprintf(stderr, "error: %s\n", err.what());
        return 1;
    }
    std::fprintf(stderr, "usage: host_selftest pt|png|resize|decode ...\n");
    return 2;
}
