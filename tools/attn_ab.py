"""attention kernel timing at the step's shape (37 windows x 577 tokens x 16 heads; 35 windows alone), random data"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
for W in (35, 37):
    M = W * 577
    qkv = (torch.randn(M, 3072, device="cuda") * 1.2).half()
    out = torch.empty(M, 1024, dtype=torch.float16, device="cuda")
    f = lambda: ctx.lib.me_op_attention(ctx.handle, C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), W, 577, 16)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    flop = 4.0 * W * 16 * 577 * 577 * 64
    print(f"windows {W}: median {statistics.median(ts):.1f} us  min {min(ts):.1f} us  {flop / statistics.median(ts) / 1e6:.0f} TFLOP/s", flush=True)
