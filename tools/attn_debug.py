"""development probe: which rows of the planted-maximum test are non-finite, per kernel variant"""
import ctypes as C, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
tokens, heads, Cc = 577, 1, 64
g = torch.Generator().manual_seed(11)
x = torch.randn(tokens, 3 * Cc, generator=g)
qi = 40
for t in range(9):
    x[64 * t + 5, Cc:2 * Cc] = x[qi, 0:Cc] * (t + 2) * 64.0 / float(x[qi, 0:Cc].pow(2).sum())
x[576, Cc:2 * Cc] = x[200, 0:Cc] * 5.0
u = torch.full((Cc,), 0.125)
x[:, Cc:2 * Cc] += 4.0 * u
x[100, 0:Cc] = -175.0 * u
qkv = x.half().cuda()
QS = 0.125 * 1.4426950408889634
p = lambda t: C.c_void_p(t.data_ptr())
for name, env, pre in (("v1", {"ME_ATT_V": "1"}, False), ("v2", {}, False), ("v2pre", {}, True)):
    os.environ.pop("ME_ATT_V", None)
    os.environ.update(env)
    xin = qkv.clone()
    if pre:
        xin[:, :Cc] = (xin[:, :Cc].float() * QS).half()
    out = torch.empty(tokens, Cc, dtype=torch.float16, device="cuda")
    fn = ctx.lib.me_op_attention_prescaled if pre else ctx.lib.me_op_attention
    assert fn(ctx.handle, p(xin), p(out), 1, tokens, heads) == 0
    torch.cuda.synchronize()
    bad = (~torch.isfinite(out.float()).all(dim=1)).nonzero().flatten().tolist()
    xx = xin.double()
    q, k, v = xx[:, :Cc], xx[:, Cc:2 * Cc], xx[:, 2 * Cc:]
    s = q @ k.T * (math.log(2.0) if pre else 0.125)
    ref = torch.softmax(s, dim=1) @ v
    ok = torch.isfinite(out.float()).all(dim=1)
    err = (out.double() - ref).abs().max(dim=1).values
    print(name, "non-finite rows:", bad[:40], "count", len(bad), " max err over finite rows %.3e" % float(err[ok].max()),
          " worst finite row", int(torch.where(ok, err, torch.zeros_like(err)).argmax()), flush=True)
    for r in bad[:3]:
        sr = s[r] / math.log(2.0) if pre else s[r] * 1.4426950408889634
        print("   row", r, "scores (log2 units) per tile max:", [round(float(sr[64 * t:64 * t + 64].max()), 1) for t in range(9)], "tail", round(float(sr[576]), 1))
