// Host check of matrix-eyes_amd/csrc/ryu_f64.h (the number formatter of the OBJ writer, host and device): compiled and
// run by tests/test_host.py.  Below 2^53 every double must print exactly as std::to_chars(fixed) does (both are
// "shortest round-trip digits, positional notation", which is what Rust's `{}` prints: output.rs:566-602); beyond,
// to_chars prints the exact integer while Rust pads the shortest digits with zeros, so there the text must read
// back as the same double and carry at most 17 significant digits.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "ryu_f64.h"

static std::string ref(double v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char b[400];
    auto r = std::to_chars(b, b + 400, v, std::chars_format::fixed);
    return std::string(b, r.ptr);
}
static std::string mine(double v) {
    char b[400];
    return std::string(b, (size_t)me::ryu::format_fixed(v, b));
}

int main(int argc, char** argv) {
    const long rounds = argc > 1 ? atol(argv[1]) : 1000000;
    std::mt19937_64 g(1);
    long bad = 0, n = 0;
    auto chk = [&](double v) {
        ++n;
        const std::string m = mine(v);
        if (std::fabs(v) >= 9007199254740992.0 && !std::isinf(v)) {
            const double back = strtod(m.c_str(), nullptr);
            const size_t digits = m.find_last_not_of('0') + 1 - (m[0] == '-');
            if (back != v || digits > 17) {
                if (bad++ < 10) printf("MISMATCH (large) %a %s\n", v, m.c_str());
            }
            return;
        }
        if (ref(v) != m) {
            if (bad++ < 10) printf("MISMATCH %a: %s vs %s\n", v, ref(v).c_str(), m.c_str());
        }
    };
    const double special[] = {0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e21, 1e22, 1e23, 1e-7, 123456789.0, 5e-324,
                              2.2250738585072014e-308, 1.7976931348623157e308, 9007199254740993.0, 0.3, 2.5, 1e15, 1e16,
                              1e17, 4.35, 8.41e21, 2.0e-3, INFINITY, -INFINITY, NAN, 0.10000000149011612, 299792458.0,
                              1.0 / 3.0, 2.0 / 3.0, 9.5367431640625e-7, 8.5e-323};
    for (double v : special) chk(v);
    for (long i = 0; i < rounds; ++i) {  // any bit pattern
        uint64_t b = g();
        double v;
        memcpy(&v, &b, 8);
        if (!std::isnan(v)) chk(v);
    }
    for (long i = 0; i < rounds; ++i) {  // what the writer prints: f32 values widened, and 1 - v
        uint32_t b = (uint32_t)g();
        float f;
        memcpy(&f, &b, 4);
        if (std::isnan(f)) continue;
        chk((double)f);
        chk(1.0 - (double)std::fabs(f));
    }
    std::uniform_real_distribution<double> u(-250, 250);
    for (long i = 0; i < rounds; ++i) {
        const float f = (float)u(g);
        chk((double)f);
        chk((double)(f * 1e-3f));
    }
    for (int c = 0; c < 256; ++c) chk((double)c / 255.0);
    for (int e = -330; e <= 308; ++e) {
        chk(std::pow(10.0, e));
        chk(std::nextafter(std::pow(10.0, e), 0));
        chk(std::nextafter(std::pow(10.0, e), INFINITY));
    }
    for (int e = -1074; e <= 1023; ++e) {
        chk(std::ldexp(1.0, e));
        chk(std::ldexp(1.0, e) * 1.5);
    }
    printf("%ld checked, %ld mismatches\n", n, bad);
    return bad != 0;
}
