"""Development probe: the head's final layers (me_op_head_final) at 1536 x 1536, the halo kernel against the implicit-GEMM
tile, one process, interleaved rounds."""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
from tools.bench_kernels import ptr

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
lib, h = ctx.lib, ctx.handle
B, S, Cin, Cmid = int(os.environ.get("HEAD_B", 1)), 1536, 128, 32
xb = torch.zeros(B, S + 2, S + 2, Cin, dtype=torch.float16, device="cuda")
xb[:, 1:S + 1, 1:S + 1] = torch.randn(B, S, S, Cin, device="cuda").half()
w = (torch.randn(Cmid, 9 * Cin, device="cuda") / math.sqrt(9 * Cin)).half()
bias, w2, b2 = torch.randn(Cmid, device="cuda") * 0.3, torch.randn(Cmid, device="cuda") / 6, torch.tensor([0.4], device="cuda")
fn = torch.ones(B, device="cuda")
out = torch.empty(B * S * S, device="cuda")
call = lambda cfg: lib.me_op_head_final(h, ptr(xb), B, S, S, Cin, ptr(w), Cmid, ptr(bias), ptr(w2), ptr(b2), ptr(fn), 1e-4, 1e4, ptr(out), cfg)
ts = {-1: [], 2: []}
for r in range(8):
    for cfg in ts:
        assert call(cfg) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            call(cfg)
        e1.record(); torch.cuda.synchronize()
        ts[cfg].append(e0.elapsed_time(e1) * 1e3 / 3)
flop = 2.0 * B * S * S * Cmid * 9 * Cin
byts = B * ((S + 2) ** 2 * Cin * 2 + S * S * 4)
for cfg, name in ((-1, "halo kernel"), (2, "implicit-GEMM tile")):
    med = statistics.median(ts[cfg])
    print(f"{name:20s} median {med:7.1f} us  min {min(ts[cfg]):7.1f} us  {flop / med / 1e6:6.0f} TFLOP/s  {byts / med / 1e6:5.2f} TB/s of algorithmic bytes", flush=True)
