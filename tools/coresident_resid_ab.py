"""the step's RESIDUAL launches (proj: K = 1024, fc2: K = 4096; M = 21760, N = 1024, x += gamma * (A W^T + b) on the
f32 token stream, then LayerNorm of the updated rows as the next linear's 16-bit operand) on every tile configuration -- the 4-wave
tiles of which two or three workgroups are resident per CU, so that one workgroup's read-modify-write epilogue (43 % of proj's tile)
runs beside another's main loop, against the 8-wave two-group tiles (one workgroup per CU), and against the 352-row tile with
the LayerNorm inside the launch (what the forward pass runs).  Every unfused row is residual launch + stand-alone LayerNorm launch.
One process, interleaved rounds; TFLOP/s on the GEMM's own FLOPs over the time of BOTH launches.
    python3 tools/coresident_resid_ab.py"""
import ctypes as C, math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
lib, h = ctx.lib, ctx.handle
p = lambda t: C.c_void_p(t.data_ptr())
arr = lambda t: (C.c_void_p * 3)(t.data_ptr(), 0, 0)
names = {c: lib.me_op_gemm_config_name(c).decode() for c in range(lib.me_op_gemm_config_count())}
M, N = 21760, 1024
for label, K in (("proj (K = 1024)", 1024), ("fc2 (K = 4096)", 4096)):
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    b = torch.randn(N, generator=g).cuda(); gm = (0.05 + 0.15 * torch.rand(N, generator=g)).cuda()
    lw = (1.0 + 0.1 * torch.randn(N, generator=g)).cuda(); lb = (0.1 * torch.randn(N, generator=g)).cuda()
    x = (torch.randn(M, N, generator=g) * 2.0).cuda()
    xn = torch.empty(M, N, dtype=torch.float16, device="cuda")

    def unfused(c):
        rc = lib.me_op_linear_residual(h, M, N, K, p(a), p(w), p(b), p(gm), p(x), c)
        if rc == 0:
            rc = lib.me_op_layernorm(h, p(x), p(lw), p(lb), p(xn), None, M, N, C.c_float(1e-5))
        return rc

    def fused(_):
        return lib.me_op_linear_residual_layernorm(h, M, N, K, p(a), 0, 0, arr(w), arr(b), arr(gm), arr(lw), arr(lb), C.c_float(1e-5), p(x), p(xn))

    cases = [(f"{names[c]} + layernorm_kernel", unfused, c) for c in names if "halo" not in names[c] and unfused(c) == 0]
    cases.append(("352x256x64/8w-pp, LayerNorm in the launch", fused, 0))
    assert fused(0) == 0
    ctx.synchronize()
    ts = {nm: [] for nm, _, _ in cases}
    for r in range(8):
        for nm, fn, c in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn(c)
            e1.record(); torch.cuda.synchronize()
            ts[nm].append(e0.elapsed_time(e1) * 1e3 / 3)
    print(label)
    for nm, _, _ in cases:
        med = statistics.median(ts[nm])
        print(f"   {nm:44s} median {med:7.1f} us  min {min(ts[nm]):7.1f} us  {2.0 * M * N * K / med / 1e6:6.0f} TFLOP/s", flush=True)
