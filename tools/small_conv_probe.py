"""Development probe: the low-resolution decoder convolutions (3x3, 48^2 x 1024 ch and 96^2 x 512 / 256 ch) on
every tile configuration.  Usage: python tools/small_conv_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import matrix_eyes_amd as m
from tools.bench_kernels import ptr, timeit


def main():
    ctx = m.Context(0, "f16", m.ModelConfig.tiny())
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lib, h = ctx.lib, ctx.handle
    for (Hh, cin, cout) in [(48, 1024, 1024), (48, 1024, 256), (96, 1024, 256), (96, 512, 512), (96, 256, 256), (192, 256, 256), (192, 512, 256), (384, 256, 256), (24, 1024, 1024)]:
        xb = torch.randn(1, Hh + 2, Hh + 2, cin, device="cuda").half()
        w = (torch.randn(cout, 9 * cin, device="cuda") / (3 * cin ** 0.5)).half()
        bias = torch.randn(cout, device="cuda")
        out16 = torch.zeros(1, Hh + 2, Hh + 2, cout, dtype=torch.float16, device="cuda")
        r32 = torch.randn(Hh * Hh, cout, device="cuda")
        o32 = torch.empty(Hh * Hh, cout, device="cuda")
        line = []
        for cfg in [-1] + list(range(lib.me_op_gemm_config_count())):
            ms = timeit(lambda: lib.me_op_conv2d(h, ptr(xb), 1, Hh, Hh, cin, ptr(w), cout, 3, 1, ptr(bias), ptr(r32), None,
                                                 ptr(o32), ptr(out16), 1, 2, 0, cfg), iters=10)
            line.append("%s %.1f us" % ("auto" if cfg < 0 else lib.me_op_gemm_config_name(cfg).decode(), ms * 1e3))
        print(f"{Hh}^2 {cin}->{cout}: " + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
