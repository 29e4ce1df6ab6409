"""Residual GEMM (proj, fc2 at M = 21760): one launch of 340 tiles of 256x256 (two rounds, the second a third full)
against whole rounds on 256x256 + the remaining 5376 rows on 224 tiles of 96x256 (config 7) as a second launch.
Interleaved rounds in one process, random data, outputs compared bit for bit; also each part timed on its own."""
import ctypes as C
import json
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import matrix_eyes_amd as m


def ptr(t, off=0):
    return C.c_void_p(t.data_ptr() + off)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    tail_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    ctx = m.Context(0, "f16", m.ModelConfig.tiny())
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lib, h = ctx.lib, ctx.handle
    M, M1 = 21760, 16384
    for name, N, K in (("proj", 1024, 1024), ("fc2", 1024, 4096)):
        g = torch.Generator(device="cuda").manual_seed(K)
        a = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
        w = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / math.sqrt(K / 3)).half()
        bias = torch.randn(N, device="cuda", generator=g)
        gamma = torch.rand(N, device="cuda", generator=g)
        x0 = torch.randn(M, N, device="cuda", generator=g)
        xa, xb = torch.empty_like(x0), torch.empty_like(x0)

        def one():
            assert lib.me_op_linear_residual(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(xa), 0) == 0

        def head():
            assert lib.me_op_linear_residual(h, M1, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(xb), 0) == 0

        def tail():
            assert lib.me_op_linear_residual(h, M - M1, N, K, ptr(a, M1 * K * 2), ptr(w), ptr(bias), ptr(gamma),
                                             ptr(xb, M1 * N * 4), tail_cfg) == 0

        def two():
            head()
            tail()

        times = {"one": [], "two": [], "head": [], "tail": []}
        bad = 0
        for r in range(rounds + 1):
            order = [("one", one, xa), ("two", two, xb)]
            for which, fn, x in (order if r % 2 == 0 else order[::-1]):
                x.copy_(x0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[which].append(e0.elapsed_time(e1))
            bad += 0 if torch.equal(xa, xb) else 1
            for which, fn in (("head", head), ("tail", tail)):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[which].append(e0.elapsed_time(e1))
        row = {"op": name, "M": M, "N": N, "K": K, "tail_cfg": lib.me_op_gemm_config_name(tail_cfg).decode(), "mismatching_rounds": bad}
        for k, v in times.items():
            row[k] = {"median_us": round(statistics.median(v) * 1e3, 1), "min_us": round(min(v) * 1e3, 1)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
