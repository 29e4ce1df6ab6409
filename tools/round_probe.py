"""Development probe: how the two-group 256x256 GEMM's time depends on the number of rounds (tiles / 256 CUs),
and what the remainder rows cost on 128x128 tiles.  Usage: python tools/round_probe.py"""
import ctypes as C
import math
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
    shapes = [(1024, 1024, "proj"), (1024, 4096, "fc2"), (4096, 1024, "fc1")]
    cases = None
    if len(sys.argv) > 1 and sys.argv[1] == "full":   # the step's four GEMMs on the merged rows
        shapes = [(3072, 1024, "qkv"), (1024, 1024, "proj"), (1024, 4096, "fc2"), (4096, 1024, "fc1")]
        cases = [(21760, 0), (86016, 0)]
    for (N, K, name) in shapes:
        for (Mx, cfg) in cases or [(16384, 0), (21760, 0), (5376, 1), (5376, 3), (5376, 0), (20480, 0), (1280, 1), (1280, 0)]:
            a = torch.randn(Mx, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
            bias = torch.randn(N, device="cuda")
            out16 = torch.empty(Mx, N, dtype=torch.float16, device="cuda")
            x32 = torch.randn(Mx, N, device="cuda")
            gamma = torch.rand(N, device="cuda")
            if name in ("proj", "fc2"):
                f = lambda: lib.me_op_linear_residual(h, Mx, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(x32), cfg)
            else:
                f = lambda: lib.me_op_linear(h, Mx, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), None, 1 if name == "fc1" else 0, cfg)
            ms = timeit(f, iters=20)
            print(name, "M", Mx, "cfg", lib.me_op_gemm_config_name(cfg).decode(), "ms %.4f" % ms, flush=True)


if __name__ == "__main__":
    main()
