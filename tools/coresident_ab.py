"""VERDICT r3 item 3, the measurable part: the step's qkv and fc1 launches (M = 21760, K = 1024) on every tile
configuration -- among them the 4-wave tiles of which two or three workgroups are resident per CU (128x128: 64 KiB of LDS,
160x128: 72 KiB), i.e. the arrangement in which one workgroup's epilogue runs beside another's main loop -- against the
8-wave two-group tiles (one workgroup per CU).  One process, interleaved rounds; TFLOP/s on the launch's own FLOPs.
    python3 tools/coresident_ab.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import matrix_eyes_amd as m
from tools.bench_kernels import ptr

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
lib, h = ctx.lib, ctx.handle
names = {c: lib.me_op_gemm_config_name(c).decode() for c in range(lib.me_op_gemm_config_count())}
M, K = 21760, 1024
x = torch.randn(M, K, device="cuda").half()
for label, N, act in (("qkv (bias, 16-bit store)", 3072, 0), ("fc1 (bias + GELU, 16-bit store)", 4096, 1)):
    w = (torch.randn(N, K, device="cuda") / 32.0).half()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    cfgs = [c for c in names if "halo" not in names[c] and lib.me_op_linear(h, M, N, K, ptr(x), ptr(w), ptr(bias), ptr(out), None, act, c) == 0]
    ctx.synchronize()
    ts = {c: [] for c in cfgs}
    for r in range(8):
        for c in cfgs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                lib.me_op_linear(h, M, N, K, ptr(x), ptr(w), ptr(bias), ptr(out), None, act, c)
            e1.record(); torch.cuda.synchronize()
            ts[c].append(e0.elapsed_time(e1) * 1e3 / 3)
    print(label)
    for c in cfgs:
        med = statistics.median(ts[c])
        print(f"   {names[c]:24s} median {med:7.1f} us  min {min(ts[c]):7.1f} us  {2.0 * M * N * K / med / 1e6:6.0f} TFLOP/s", flush=True)
