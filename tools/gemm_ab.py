"""A/B of two GEMM tile configurations in ONE process, interleaved rounds, random operands (cdna_hip_programming.md
section 5.4 rules 24 and 25): the two-group 256x256 kernel (config 0) against the guide's 8-phase schedule (config 6)
at the step's ViT shapes and at the guide's own square shapes.  Every round also compares the outputs of the two
configurations bit for bit (both accumulate each output element over K in the same order), which doubles as a race
screen of the new schedule.
usage: gemm_ab.py [dtype] [rounds] [cfgA,cfgB]"""
import ctypes as C
import json
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import matrix_eyes_amd as m


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    cfgs = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "0,6").split(",")]
    t16 = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    ctx = m.Context(0, dtype, m.ModelConfig.tiny())
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lib, h = ctx.lib, ctx.handle
    names = {c: lib.me_op_gemm_config_name(c).decode() for c in cfgs}
    shapes = [("qkv", 21760, 3072, 1024, "store"), ("fc1", 21760, 4096, 1024, "gelu"), ("proj", 21760, 1024, 1024, "resid"),
              ("fc2", 21760, 1024, 4096, "resid"), ("sq4096", 4096, 4096, 4096, "store"), ("sq8192", 8192, 8192, 8192, "store")]
    out = []
    for name, M, N, K, kind in shapes:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        a = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(t16)          # uniform [-1, 1): the guide's data
        w = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / math.sqrt(K / 3)).to(t16)
        bias = torch.randn(N, device="cuda", generator=g)
        gamma = torch.rand(N, device="cuda", generator=g)
        x0 = torch.randn(M, N, device="cuda", generator=g)
        outs = {c: torch.empty(M, N, dtype=t16, device="cuda") for c in cfgs}
        xs = {c: torch.empty(M, N, device="cuda") for c in cfgs}

        def run(c):
            if kind == "resid":
                return lib.me_op_linear_residual(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(xs[c]), c)
            return lib.me_op_linear(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(outs[c]), None, 1 if kind == "gelu" else 0, c)

        times = {c: [] for c in cfgs}
        mismatches = 0
        for c in cfgs:            # warm-up
            xs[c].copy_(x0)
            assert run(c) == 0, ctx.lib.me_last_error(h)
        torch.cuda.synchronize()
        for r in range(rounds):
            for c in (cfgs if r % 2 == 0 else cfgs[::-1]):
                if kind == "resid":
                    xs[c].copy_(x0)
                else:
                    outs[c].fill_(float("nan"))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert run(c) == 0
                e1.record()
                torch.cuda.synchronize()
                times[c].append(e0.elapsed_time(e1))
            ref = xs[cfgs[0]] if kind == "resid" else outs[cfgs[0]]
            for c in cfgs[1:]:
                got = xs[c] if kind == "resid" else outs[c]
                if not torch.equal(got, ref):
                    mismatches += 1
        flop = 2.0 * M * N * K
        row = {"shape": name, "M": M, "N": N, "K": K, "epilogue": kind, "rounds": rounds, "bitwise_mismatching_rounds": mismatches}
        for c in cfgs:
            med, mn = statistics.median(times[c]), min(times[c])
            row[names[c]] = {"median_ms": round(med, 4), "min_ms": round(mn, 4), "median_tflops": round(flop / med / 1e9, 1),
                             "best_tflops": round(flop / mn / 1e9, 1)}
        out.append(row)
        print(json.dumps(row), flush=True)
        del a, w, outs, xs, x0
    print(json.dumps({"dtype": dtype, "data": "uniform [-1, 1) activations, scaled uniform weights", "results": out}))


if __name__ == "__main__":
    main()
