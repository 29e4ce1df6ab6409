"""Development A/B: the fused residual + LayerNorm launch (me_op_linear_residual_layernorm) at the step's proj and fc2
shapes, one library per process (MATRIX_EYES_HIP_LIB), interleaved by the calling script.  Prints the median time and a
checksum of the outputs (the two builds must agree bit for bit)."""
import ctypes as C, hashlib, math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
arr = lambda t: (C.c_void_p * 3)(t.data_ptr(), 0, 0)
M, N = int(os.environ.get("AB_M", 21760)), 1024
for K in (1024, 4096):
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K + int(os.environ.get("ME_DEV_LDA_PAD", 0)), generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    b = torch.randn(N, generator=g).cuda(); gm = (0.05 + 0.15 * torch.rand(N, generator=g)).cuda()
    lw = (1.0 + 0.1 * torch.randn(N, generator=g)).cuda(); lb = (0.1 * torch.randn(N, generator=g)).cuda()
    x0 = (torch.randn(M, N, generator=g) * 2.0).cuda()
    x = x0.clone()
    xn = torch.empty(M, N, dtype=torch.float16, device="cuda")
    call = lambda: ctx.lib.me_op_linear_residual_layernorm(ctx.handle, M, N, K, p(a), 0, 0, arr(w), arr(b), arr(gm), arr(lw), arr(lb), 1e-5, p(x), p(xn))
    assert call() == 0
    ctx.synchronize()
    h = hashlib.sha1(x.cpu().numpy().tobytes() + xn.cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for r in range(12):
        x.copy_(x0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            call()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 4)
    print(f"{os.environ.get('MATRIX_EYES_HIP_LIB', 'in-tree')[-24:]:24s} M {M} K {K}: median {statistics.median(ts):7.1f} us  min {min(ts):7.1f} us  sha {h}", flush=True)
