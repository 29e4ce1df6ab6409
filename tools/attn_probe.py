"""Development probe: attention kernel time at the step's shape (35 + 2 windows of 577 tokens, 16 heads)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import matrix_eyes_amd as m
from tools.bench_kernels import ptr, timeit

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
lib, h = ctx.lib, ctx.handle
for W in (35, 37, 140):
    qkv = torch.randn(W * 577, 3072, device="cuda").half()
    out = torch.empty(W * 577, 1024, dtype=torch.float16, device="cuda")
    best = min(timeit(lambda: lib.me_op_attention(h, ptr(qkv), ptr(out), W, 577, 16), iters=30) for _ in range(3))
    print("attention windows", W, "us %.2f" % (best * 1e3), flush=True)
