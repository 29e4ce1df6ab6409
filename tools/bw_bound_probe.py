"""Development probe: the four launches of the step's tail that move the most bytes per FLOP (K <= 768 on 384^2 - 768^2 pixel
maps) on every tile configuration that takes them -- time, TFLOP/s and the HBM rate their algorithmic bytes come to.
    python3 tools/bw_bound_probe.py"""
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
    names = {c: lib.me_op_gemm_config_name(c).decode() for c in range(lib.me_op_gemm_config_count())}
    cfgs = [-1] + [c for c in names if "halo" not in names[c] and "352" not in names[c]]
    if os.environ.get("BW_PROBE_AUTO"):
        cfgs = [-1]
    print("ME_STAGGER_US =", os.environ.get("ME_STAGGER_US", "(unset)"), flush=True)
    # (label, H, Cin, Cout, out32, out16, border)
    for label, H, cin, cout, o32, o16, border in [("convt 384^2 512->256 f32 + bordered 16-bit", 384, 512, 256, 1, 1, 1),
                                                   ("convt 384^2 768->256 f32", 384, 768, 256, 1, 0, 0),
                                                   ("convt 768^2 128->128 bordered 16-bit", 768, 128, 128, 0, 1, 1)]:
        x = torch.randn(H * H, cin, device="cuda").half()
        w = (torch.randn(4 * cout, cin, device="cuda") / cin ** 0.5).half()
        out32 = torch.empty(4 * H * H, cout, device="cuda") if o32 else None
        out16 = torch.zeros((2 * H + 2) ** 2, cout, dtype=torch.float16, device="cuda") if o16 else None
        byts = H * H * cin * 2 + 4 * H * H * cout * (4 * o32 + 2 * o16)
        flop = 2.0 * H * H * 4 * cout * cin
        line = []
        for cfg in cfgs:
            call = lambda: lib.me_op_conv_transpose2x2(h, ptr(x), 1, H, H, cin, ptr(w), cout, None, ptr(out32), ptr(out16), border, cfg)
            if call() != 0:   # a configuration that does not take the shape
                continue
            ms = timeit(call, iters=10)
            line.append("%s %.1f us (%.0f TFLOP/s, %.2f TB/s)" % ("auto" if cfg < 0 else names[cfg], ms * 1e3, flop / ms / 1e9, byts / ms / 1e9))
        print(label + f" [{byts / 1e6:.0f} MB]:\n   " + "\n   ".join(line), flush=True)
    # the RCU convolutions of the last fusion level: plain, and with the f32 residual read + f32 and 16-bit outputs
    for label, H, res, o32 in [("conv3x3 768^2 256->256 16-bit out", 768, 0, 0), ("conv3x3 768^2 256->256 residual + f32 + 16-bit out", 768, 1, 1),
                               ("conv3x3 384^2 256->256 residual + f32 + 16-bit out", 384, 1, 1)]:
        xb = torch.randn(1, H + 2, H + 2, 256, device="cuda").half()
        w = (torch.randn(256, 9 * 256, device="cuda") / 48.0).half()
        bias = torch.randn(256, device="cuda")
        out16 = torch.zeros(1, H + 2, H + 2, 256, dtype=torch.float16, device="cuda")
        r32 = torch.randn(H * H, 256, device="cuda") if res else None
        out32 = torch.empty(H * H, 256, device="cuda") if o32 else None
        byts = (H + 2) ** 2 * 256 * 2 + H * H * 256 * (2 + 4 * res + 4 * o32)
        flop = 2.0 * H * H * 256 * 2304
        call = lambda: lib.me_op_conv2d(h, ptr(xb), 1, H, H, 256, ptr(w), 256, 3, 1, ptr(bias), ptr(r32), None, ptr(out32), ptr(out16), 1, 2, 0, -1)
        assert call() == 0
        ms = timeit(call, iters=10)
        print(label + f" [{byts / 1e6:.0f} MB]: auto {ms * 1e3:.1f} us ({flop / ms / 1e9:.0f} TFLOP/s, {byts / ms / 1e9:.2f} TB/s)", flush=True)
    M, K, N = 768 * 768, 512, 256
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    bias = torch.randn(N, device="cuda")
    out16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
    byts, flop = M * K * 2 + M * N * 2, 2.0 * M * N * K
    line = []
    for cfg in cfgs:
        call = lambda: lib.me_op_linear(h, M, N, K, ptr(x), ptr(w), ptr(bias), ptr(out16), None, 0, cfg)
        if call() != 0:
            continue
        ms = timeit(call, iters=10)
        line.append("%s %.1f us (%.0f TFLOP/s, %.2f TB/s)" % ("auto" if cfg < 0 else names[cfg], ms * 1e3, flop / ms / 1e9, byts / ms / 1e9))
    print(f"1x1 conv 768^2 512->256 16-bit [{byts / 1e6:.0f} MB]:\n   " + "\n   ".join(line), flush=True)


if __name__ == "__main__":
    main()
