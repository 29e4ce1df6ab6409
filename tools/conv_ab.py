"""3x3 convolution, 256 -> 256 channels, at the decoder's two big map sizes: the two-group implicit-GEMM tile (config 0)
against its halo form (config 9), interleaved rounds in one process, random data, with the epilogue forms the decoder
uses (f32 + bordered 16-bit output, two f32 residual inputs, ReLU)."""
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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    cfgs = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "0,9").split(",")]
    ctx = m.Context(0, "f16", m.ModelConfig.tiny())
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lib, h = ctx.lib, ctx.handle
    for Hh, Cin, form in ((768, 256, "resid2+relu"), (768, 256, "16-bit only"), (384, 256, "resid2+relu"), (384, 512, "16-bit only")):
        xb = torch.zeros(1, Hh + 2, Hh + 2, Cin, dtype=torch.float16, device="cuda")
        xb[:, 1:-1, 1:-1] = torch.randn(1, Hh, Hh, Cin, device="cuda").half()
        w = (torch.randn(256, 9 * Cin, device="cuda") / math.sqrt(9 * Cin)).half()
        bias = torch.randn(256, device="cuda")
        full = form == "resid2+relu"
        r1 = torch.randn(Hh * Hh, 256, device="cuda") if full else None
        r2 = torch.randn(Hh * Hh, 256, device="cuda") if full else None
        outs = {c: (torch.empty(Hh * Hh, 256, device="cuda") if full else None,
                    torch.zeros(1, Hh + 2, Hh + 2, 256, dtype=torch.float16, device="cuda")) for c in cfgs}

        def run(c):
            o32, o16 = outs[c]
            assert lib.me_op_conv2d(h, ptr(xb), 1, Hh, Hh, Cin, ptr(w), 256, 3, 1, ptr(bias), ptr(r1), ptr(r2), ptr(o32), ptr(o16), 1, 2, 0, c) == 0

        times = {c: [] for c in cfgs}
        for c in cfgs:
            run(c)
        torch.cuda.synchronize()
        for r in range(rounds):
            for c in (cfgs if r % 2 == 0 else cfgs[::-1]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(c)
                e1.record()
                torch.cuda.synchronize()
                times[c].append(e0.elapsed_time(e1))
        a16, b16 = outs[cfgs[0]][1].float(), outs[cfgs[-1]][1].float()
        row = {"map": Hh, "Cin": Cin, "epilogue": form, "max_abs_diff_16bit_outputs": float((a16 - b16).abs().max())}
        flop = 2.0 * Hh * Hh * 256 * 9 * Cin
        for c in cfgs:
            med = statistics.median(times[c])
            row[lib.me_op_gemm_config_name(c).decode()] = {"median_ms": round(med, 4), "min_ms": round(min(times[c]), 4), "tflops": round(flop / med / 1e9, 1)}
        print(json.dumps(row), flush=True)
        del xb, w, outs, r1, r2


if __name__ == "__main__":
    main()
