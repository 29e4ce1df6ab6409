"""Diagnostic: does running the ViT block chain as TWO half-size chains on two streams, half a block apart, beat one
full-size chain?  (the token-stream phases of one half -- residual epilogues, LayerNorm -- would run beside the
MFMA phases of the other).  Run with ME_GEMM_GRID_LIMIT=128 for the dual case so each persistent GEMM takes half the
CUs.   python tools/dual_stream_probe.py single|dual [blocks]"""
import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m

mode = sys.argv[1]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 24
T, Cc, heads = 577, 1024, 16
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Chain:
    def __init__(self, windows):
        self.ctx = m.Context(0, "f16", m.ModelConfig.tiny())
        self.stream = torch.cuda.Stream()
        self.ctx.set_stream(self.stream.cuda_stream)
        self.W = windows
        M = self.M = windows * T
        with torch.cuda.stream(self.stream):
            g = lambda *s: torch.randn(*s, device="cuda")
            self.x = g(M, Cc)
            self.xn = torch.empty(M, Cc, dtype=torch.float16, device="cuda")
            self.qkv = torch.empty(M, 3 * Cc, dtype=torch.float16, device="cuda")
            self.att = torch.empty(M, Cc, dtype=torch.float16, device="cuda")
            self.hid = torch.empty(M, 4 * Cc, dtype=torch.float16, device="cuda")
            self.w = {k: (g(n, kk) / math.sqrt(kk)).half() for k, (n, kk) in
                      dict(qkv=(3 * Cc, Cc), proj=(Cc, Cc), fc1=(4 * Cc, Cc), fc2=(Cc, 4 * Cc)).items()}
            self.b = {k: g(n) * 0.1 for k, n in dict(qkv=3 * Cc, proj=Cc, fc1=4 * Cc, fc2=Cc).items()}
            self.gamma = torch.full((Cc,), 1e-3, device="cuda")
            self.lnw, self.lnb = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        self.stream.synchronize()

    def half_a(self):   # LN, qkv, attention, proj
        L, h, M = self.ctx.lib, self.ctx.handle, self.M
        assert L.me_op_layernorm(h, p(self.x), p(self.lnw), p(self.lnb), p(self.xn), None, M, Cc, C.c_float(1e-6)) == 0
        assert L.me_op_linear(h, M, 3 * Cc, Cc, p(self.xn), p(self.w["qkv"]), p(self.b["qkv"]), p(self.qkv), None, 0, 0) == 0
        assert L.me_op_attention(h, p(self.qkv), p(self.att), self.W, T, heads) == 0
        assert L.me_op_linear_residual(h, M, Cc, Cc, p(self.att), p(self.w["proj"]), p(self.b["proj"]), p(self.gamma), p(self.x), 0) == 0

    def half_b(self):   # LN, fc1, fc2
        L, h, M = self.ctx.lib, self.ctx.handle, self.M
        assert L.me_op_layernorm(h, p(self.x), p(self.lnw), p(self.lnb), p(self.xn), None, M, Cc, C.c_float(1e-6)) == 0
        assert L.me_op_linear(h, M, 4 * Cc, Cc, p(self.xn), p(self.w["fc1"]), p(self.b["fc1"]), p(self.hid), None, 1, 0) == 0
        assert L.me_op_linear_residual(h, M, Cc, 4 * Cc, p(self.hid), p(self.w["fc2"]), p(self.b["fc2"]), p(self.gamma), p(self.x), 0) == 0


def run(chains, offset):
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        if len(chains) == 2 and offset:
            chains[0].half_a()          # chain 0 runs half a block ahead
        for i in range(blocks):
            for k, c in enumerate(chains):
                if len(chains) == 2 and offset and k == 0:
                    c.half_b()
                    if i + 1 < blocks:
                        c.half_a()
                else:
                    c.half_a()
                    c.half_b()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3
    return ms


if mode == "single":
    print("single chain, 37 windows: %.3f ms for %d blocks" % (run([Chain(37)], False), blocks))
else:
    cs = [Chain(19), Chain(18)]
    print("two chains (19 + 18 windows), in phase:        %.3f ms" % run(cs, False))
    print("two chains (19 + 18 windows), half a block apart: %.3f ms" % run(cs, True))
