"""Development probe: the head's ConvTranspose (768^2, 128 -> 128: M 589824, N 512, K 128) against the same GEMM with a
plain 16-bit row-major store -- what the pixel-shuffle addressing and the bordered layout cost in the epilogue."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
from tools.bench_kernels import ptr

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
lib, h = ctx.lib, ctx.handle
for H, cin, cout in ((768, 128, 128), (384, 512, 256)):
    M, N, K = H * H, 4 * cout, cin
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    bias = torch.randn(N, device="cuda")
    out_plain = torch.empty(M, N, dtype=torch.float16, device="cuda")
    out_b = torch.zeros((2 * H + 2) ** 2, cout, dtype=torch.float16, device="cuda")
    out_u = torch.zeros((2 * H) ** 2, cout, dtype=torch.float16, device="cuda")
    calls = {"plain 16-bit store [M][N]": lambda: lib.me_op_linear(h, M, N, K, ptr(x), ptr(w), ptr(bias), ptr(out_plain), None, 0, 0),
             "plain store, generic epilogue path": lambda: lib.me_op_linear(h, M, N, K, ptr(x), ptr(w), None, ptr(out_plain), None, 0, 0),
             "ConvTranspose, bordered 16-bit": lambda: lib.me_op_conv_transpose2x2(h, ptr(x), 1, H, H, cin, ptr(w), cout, None, None, ptr(out_b), 1, 0),
             "ConvTranspose, unbordered 16-bit": lambda: lib.me_op_conv_transpose2x2(h, ptr(x), 1, H, H, cin, ptr(w), cout, None, None, ptr(out_u), 0, 0)}
    ts = {k: [] for k in calls}
    for r in range(8):
        for k, f in calls.items():
            assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record(); torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) * 1e3 / 3)
    print(f"{H}^2 {cin} -> {cout} (M {M}, N {N}, K {K}; 256x256 two-group tile):")
    for k in calls:
        print(f"   {k:34s} median {statistics.median(ts[k]):7.1f} us  min {min(ts[k]):7.1f} us", flush=True)
