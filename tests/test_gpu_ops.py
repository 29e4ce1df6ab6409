"""Kernel-level parity (-m gpu): each HIP kernel, called through the C ABI
(include/matrix_eyes_hip_ops.h), against a plain PyTorch fp32/fp64 evaluation of the same op on
the same 16-bit-rounded operands.  Tolerances are written per test: f32 accumulation differences
only (operands are identical), plus one rounding where the output is 16-bit."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import (TORCH16, bordered, ctx_for, dev16, max_abs_rel, max_err_over_max, pack_conv, pack_convt, ptr,
                  rel_l2)

pytestmark = pytest.mark.gpu

DTYPES = ["f16", "bf16"]
OUT_EPS = {"f16": 2.0 ** -11, "bf16": 2.0 ** -8}   # one rounding of a 16-bit output


def _check(ctx, rc):
    ctx._check(rc)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, -1])
@pytest.mark.parametrize("shape", [(577, 256, 128), (1000, 132, 192), (2309, 384, 256), (5000, 512, 64)])
def test_linear(dtype, cfg, shape):
    M, N, K = shape
    ctx = ctx_for("tiny", dtype)
    g = torch.Generator().manual_seed(M + N + K)
    a = dev16(torch.randn(M, K, generator=g), dtype)
    w = dev16(torch.randn(N, K, generator=g) / math.sqrt(K), dtype)
    bias = torch.randn(N, generator=g).cuda()
    out16 = torch.empty(M, N, dtype=TORCH16[dtype], device="cuda")
    out32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), ptr(out32), 0, cfg))
    ctx.synchronize()
    ref = a.double() @ w.double().T + bias.double()
    assert max_abs_rel(out32, ref) < 2e-5          # f32 accumulation over K <= 256
    assert max_abs_rel(out16.float(), ref) < 4 * OUT_EPS[dtype] * 4   # |x| up to ~4 rms


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_gelu(dtype):
    M, N, K = 700, 512, 128
    ctx = ctx_for("tiny", dtype)
    g = torch.Generator().manual_seed(7)
    a = dev16(torch.randn(M, K, generator=g) * 2, dtype)
    w = dev16(torch.randn(N, K, generator=g) / math.sqrt(K), dtype)
    bias = torch.randn(N, generator=g).cuda()
    out32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, N, K, ptr(a), ptr(w), ptr(bias), None, ptr(out32), 1, -1))
    ctx.synchronize()
    ref = F.gelu(a.double() @ w.double().T + bias.double())     # exact erf form (vit.rs:121)
    # the f32 accumulation over K dominates; the GELU itself is pinned by test_gelu_function below
    assert float((out32.double() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
def test_gelu_function(cfg):
    """The epilogue's GELU alone: identity weights and zero bias make the pre-activation the f16 input
    itself, so the outputs are gelu(x) for ~1 M inputs over [-9, 9], the f16 extremes and the denormals.
    x * Phi(x) with the exact erf (burn activation::gelu, vit.rs:121): the f32 result (degree-9 form) within 2.5e-7 +
    one f32 rounding of the exact value; the 16-bit result -- the fast path the ViT takes, a degree-6 form accurate to
    4.5e-5 relative / 6.5e-6 absolute, a tenth of an f16 code -- within 0.6 of an f16 code of the exact value, equal
    to the correctly rounded value for all but the inputs that sit within that tenth of a rounding boundary."""
    M, N = 4096, 256
    ctx = ctx_for("tiny", "f16")
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.linspace(-9, 9, M * N - 4096), torch.randn(4080, generator=g) * 1e-3,
                   torch.tensor([0.0, -0.0, 65504.0, -65504.0, 6e-8, -6e-8, 6.1e-5, -6.1e-5, 5.5, -5.5, 5.51, -5.51,
                                 1e4, -1e4, 0.75, -0.75])])
    a = x.reshape(M, N).half().cuda()
    w = torch.eye(N, dtype=torch.float16, device="cuda")
    bias = torch.zeros(N, device="cuda")
    out32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    out16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
    _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, N, N, ptr(a), ptr(w), ptr(bias), None, ptr(out32), 1, cfg))
    _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, N, N, ptr(a), ptr(w), ptr(bias), ptr(out16), None, 1, cfg))
    ctx.synchronize()
    ref = F.gelu(a.double())
    err = (out32.double() - ref).abs()
    assert bool((err <= 2.5e-7 + 6e-8 * ref.abs()).all()), float(err.max())
    # 16-bit copy: the correctly rounded value, or its neighbour where the f32 result sits on a rounding boundary
    want16 = ref.half()
    ulp = (want16.double().abs() * 2.0 ** -10).clamp_min(2.0 ** -24)
    assert bool(((out16.double() - ref).abs() <= 0.6 * ulp + 7e-6).all())
    assert float((out16 != want16).float().mean()) < 0.1


def test_linear_random_shapes():
    """120 random problems per run of the seed: M from 1 (a single row) to a few tiles and ragged edges, N any
    multiple of 4, K any multiple of 64, every tile configuration and the automatic choice, plain and residual
    epilogue.  Catches indexing at tile edges that the model's own shapes never reach."""
    import random
    ctx = ctx_for("tiny", "f16")
    rnd = random.Random(20240)
    ncfg = ctx.lib.me_op_gemm_config_count()
    for it in range(120):
        M = rnd.choice([1, 2, 7, 63, 64, 65, 127, 129, 255, 257, 300, 511, 513, 777, rnd.randrange(1, 1500)])
        N = 4 * rnd.choice([1, 2, 3, 7, 8, 15, 16, 31, 33, 64, 65, rnd.randrange(1, 160)])
        K = 64 * rnd.choice([1, 2, 3, 4, 5, 9, 16])
        cfg = rnd.randrange(-1, min(ncfg, 9))      # every configuration that takes a plain linear (9 is the conv halo tile)
        g = torch.Generator().manual_seed(it)
        a = dev16(torch.randn(M, K, generator=g), "f16")
        w = dev16(torch.randn(N, K, generator=g) / math.sqrt(K), "f16")
        bias = torch.randn(N, generator=g).cuda()
        ref = a.double() @ w.double().T + bias.double()
        if it % 3 == 2:
            gamma = torch.rand(N, generator=g).cuda()
            x32 = torch.randn(M, N, generator=g).cuda()
            want = ref * gamma.double() + x32.double()
            torch.cuda.synchronize()       # the context launches on its own stream, torch filled these on its
            _check(ctx, ctx.lib.me_op_linear_residual(ctx.handle, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(x32), cfg))
            got = x32
        else:
            got = torch.full((M + 1, N), 7.0, dtype=torch.float32, device="cuda")      # one guard row behind the output
            torch.cuda.synchronize()
            _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, N, K, ptr(a), ptr(w), ptr(bias), None, ptr(got), 0, cfg))
            ctx.synchronize()
            assert bool((got[M] == 7.0).all()), (it, M, N, K, cfg)
            got, want = got[:M], ref
        ctx.synchronize()
        err = float((got.double() - want).abs().max())
        assert err < 3e-5 * max(1.0, float(want.abs().max())), (it, M, N, K, cfg, err)


@pytest.mark.parametrize("cfg", [0, 1, 3])
@pytest.mark.parametrize("shape", [(20195, 1024, 256), (9000, 2304, 192), (70000, 256, 128)])
def test_linear_persistent_rounds(cfg, shape):
    """more tiles than resident workgroups: every workgroup walks several tiles, the K stream crosses tile
    boundaries (and, in the two-group kernel, the weight ring runs two slabs ahead across them)"""
    M, N, K = shape
    ctx = ctx_for("tiny", "f16")
    g = torch.Generator().manual_seed(M + N + K)
    a = dev16(torch.randn(M, K, generator=g), "f16")
    w = dev16(torch.randn(N, K, generator=g) / math.sqrt(K), "f16")
    bias = torch.randn(N, generator=g).cuda()
    out32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, N, K, ptr(a), ptr(w), ptr(bias), None, ptr(out32), 0, cfg))
    ctx.synchronize()
    ref = a.float() @ w.float().T + bias
    assert float((out32 - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    again = torch.empty_like(out32)
    _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, N, K, ptr(a), ptr(w), ptr(bias), None, ptr(again), 0, cfg))
    ctx.synchronize()
    assert torch.equal(out32, again)


def test_dynamic_tile_order_in_a_child_process():
    """ME_GEMM_DYNAMIC_TILES is read once per process: run the multi-round cases again with it set"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, ME_GEMM_DYNAMIC_TILES="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu",
                        "-k", "persistent_rounds or conv2d"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_linear_residual(dtype, cfg):
    M, N, K = 1154, 256, 512
    ctx = ctx_for("tiny", dtype)
    g = torch.Generator().manual_seed(11)
    a = dev16(torch.randn(M, K, generator=g), dtype)
    w = dev16(torch.randn(N, K, generator=g) / math.sqrt(K), dtype)
    bias = torch.randn(N, generator=g).cuda()
    gamma = (0.05 + 0.15 * torch.rand(N, generator=g)).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    ref = x.double() + gamma.double() * (a.double() @ w.double().T + bias.double())
    _check(ctx, ctx.lib.me_op_linear_residual(ctx.handle, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(x), cfg))
    ctx.synchronize()
    assert max_abs_rel(x, ref) < 2e-5


def _segmented_linear(ctx, dtype, M, N, K, seg1, seg2, cfg, resid, seed, act=0):
    """me_op_linear_segments on seeded operands; returns (result, fp64 reference)"""
    import ctypes as C
    g = torch.Generator().manual_seed(seed)
    a = dev16(torch.randn(M, K, generator=g), dtype)
    ws = [dev16(torch.randn(N, K, generator=g) / math.sqrt(K), dtype) for _ in range(3)]
    bs = [torch.randn(N, generator=g).cuda() for _ in range(3)]
    gs = [(0.05 + 0.15 * torch.rand(N, generator=g)).cuda() for _ in range(3)]
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    bounds = [0, seg1 if seg1 else M, (seg2 if seg2 else M) if seg1 else M, M]
    x0 = torch.randn(M, N, generator=g).cuda()
    ref = torch.empty(M, N, dtype=torch.float64, device="cuda")
    for i in range(3):
        lo, hi = bounds[i], bounds[i + 1]
        if hi <= lo:
            continue
        y = a[lo:hi].double() @ ws[i].double().T + bs[i].double()
        ref[lo:hi] = x0[lo:hi].double() + gs[i].double() * y if resid else (F.gelu(y) if act == 1 else y)
    if resid:
        x = x0.clone()
        torch.cuda.synchronize()      # torch's fills run on torch's stream, the op on the context's
        _check(ctx, ctx.lib.me_op_linear_segments(ctx.handle, M, N, K, ptr(a), seg1, seg2, arr(ws), arr(bs), arr(gs), None,
                                                  ptr(x), 0, cfg))
        ctx.synchronize()
        return x, ref
    out16 = torch.full((M, N), float("nan"), dtype=TORCH16[dtype], device="cuda")
    torch.cuda.synchronize()
    _check(ctx, ctx.lib.me_op_linear_segments(ctx.handle, M, N, K, ptr(a), seg1, seg2, arr(ws), arr(bs), arr(gs), ptr(out16),
                                              None, act, cfg))
    ctx.synchronize()
    return out16, ref


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [
    # (M, N, K, seg1, seg2): one column tile (no exchange); two and four column tiles; ragged last row tiles; three
    # segments with their own GEMM and LayerNorm weights; more row tiles than resident workgroups (the persistent
    # kernel's second and third rounds wait for neighbours of their own round); the step's own shape
    (700, 256, 128, 0, 0), (1000, 512, 256, 0, 0), (353, 1024, 128, 0, 0), (3000, 1024, 256, 768, 1536),
    (2500, 512, 128, 600, 1300), (60000, 1024, 128, 768, 1536), (21760, 1024, 1024, 768, 1536),
])
def test_linear_residual_layernorm_fused(dtype, shape):
    """vit.rs:165-169: the residual update with the next sublayer's LayerNorm in the same launch (gemm_core.h
    resid_ln_epilogue).  x32 is bit for bit what the plain residual launch on the same tile writes; the normalised rows
    are held to an fp64 LayerNorm of those x32 values (one 16-bit rounding, statistics summed in f32); a second and
    third launch on the same buffers (the arrival counters are never reset) reproduce themselves from the same input."""
    import ctypes as C
    M, N, K, seg1, seg2 = shape
    ctx = ctx_for("tiny", dtype)
    g = torch.Generator().manual_seed(M + N + K)
    a = dev16(torch.randn(M, K, generator=g), dtype)
    ws = [dev16(torch.randn(N, K, generator=g) / math.sqrt(K), dtype) for _ in range(3)]
    bs = [torch.randn(N, generator=g).cuda() for _ in range(3)]
    gs = [(0.05 + 0.15 * torch.rand(N, generator=g)).cuda() for _ in range(3)]
    lw = [(1.0 + 0.1 * torch.randn(N, generator=g)).cuda() for _ in range(3)]
    lb = [(0.1 * torch.randn(N, generator=g)).cuda() for _ in range(3)]
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    x0 = torch.randn(M, N, generator=g) * 2.0
    x0[:, 7] += 40.0                                 # an outlier channel: mean and variance far from 0 / 1
    x0[::3] *= 10.0                                   # rows of very different scale
    x0 = x0.cuda()
    plain = x0.clone()
    fused = x0.clone()
    xn = torch.full((M + 1, N), 7.0, dtype=TORCH16[dtype], device="cuda")     # a guard row behind the output
    torch.cuda.synchronize()
    _check(ctx, ctx.lib.me_op_linear_segments(ctx.handle, M, N, K, ptr(a), seg1, seg2, arr(ws), arr(bs), arr(gs), None,
                                              ptr(plain), 0, 10))
    eps = 1e-5
    _check(ctx, ctx.lib.me_op_linear_residual_layernorm(ctx.handle, M, N, K, ptr(a), seg1, seg2, arr(ws), arr(bs), arr(gs),
                                                        arr(lw), arr(lb), eps, ptr(fused), ptr(xn)))
    ctx.synchronize()
    assert ctx.status_flags() == 0
    assert torch.equal(fused, plain)
    assert bool((xn[M] == 7.0).all())
    bounds = [0, seg1 if seg1 else M, (seg2 if seg2 else M) if seg1 else M, M]
    ref = torch.empty(M, N, dtype=torch.float64, device="cuda")
    xd = fused.double()
    for i in range(3):
        lo, hi = bounds[i], bounds[i + 1]
        if hi > lo:
            ref[lo:hi] = F.layer_norm(xd[lo:hi], (N,), lw[i].double(), lb[i].double(), eps)
    err = (xn[:M].double() - ref).abs()
    tol = OUT_EPS[dtype] * ref.abs().clamp_min(1.0) + 2e-5 * ref.abs().clamp_min(1.0)
    assert bool((err <= 1.01 * tol).all()), float((err / tol).max())
    assert rel_l2(xn[:M].float(), ref) < OUT_EPS[dtype]
    # the stand-alone kernel on the same rows: the same values up to the rounding of the last bit of a few of them
    if seg1 == 0:
        alone = torch.empty(M, N, dtype=TORCH16[dtype], device="cuda")
        _check(ctx, ctx.lib.me_op_layernorm(ctx.handle, ptr(fused), ptr(lw[0]), ptr(lb[0]), ptr(alone), None, M, N, eps))
        ctx.synchronize()
        differ = float((alone != xn[:M]).float().mean())
        assert differ < 2e-3 and rel_l2(alone.float(), xn[:M].float()) < 0.2 * OUT_EPS[dtype], differ
    # again, twice, from the same input: same bits (counters keep counting, nothing is reset)
    for _ in range(2):
        again = x0.clone()
        xn2 = torch.empty(M, N, dtype=TORCH16[dtype], device="cuda")
        torch.cuda.synchronize()
        _check(ctx, ctx.lib.me_op_linear_residual_layernorm(ctx.handle, M, N, K, ptr(a), seg1, seg2, arr(ws), arr(bs),
                                                            arr(gs), arr(lw), arr(lb), eps, ptr(again), ptr(xn2)))
        ctx.synchronize()
        assert torch.equal(again, fused) and torch.equal(xn2, xn[:M])
    assert ctx.status_flags() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("resid", [False, True])
@pytest.mark.parametrize("shape", [
    # (M, N, K, seg1, seg2): one segment with a ragged last tile; the merged ViT row space of one image in small (three
    # segments on 256-row boundaries: 3 + 3 + ... row tiles of 352); boundaries that no other tile height allows
    (1000, 256, 128, 0, 0), (577, 512, 256, 0, 0), (5000, 260, 192, 0, 0), (352, 256, 128, 0, 0), (353, 256, 128, 0, 0),
    (3000, 512, 256, 768, 1536), (2500, 256, 128, 600, 1300), (1500, 256, 128, 700, 0), (90000, 256, 128, 768, 1536),
])
def test_tall_tile_352(dtype, resid, shape):
    """Tile config 10 (352x256, two weight slots, row tiles laid out per segment): against fp64, and -- where the
    256-row tile takes the same problem -- bit for bit against tile config 0 (same K order per output element)."""
    M, N, K, seg1, seg2 = shape
    ctx = ctx_for("tiny", dtype)
    got, ref = _segmented_linear(ctx, dtype, M, N, K, seg1, seg2, 10, resid, seed=M + N + K, act=0 if resid else 1)
    if resid:
        assert max_abs_rel(got, ref) < 2e-5
    else:
        assert not bool(torch.isnan(got.float()).any())          # every row of every segment was stored
        assert max_abs_rel(got.float(), ref) < 4 * OUT_EPS[dtype] * 4
    if seg1 % 256 == 0 and seg2 % 256 == 0 and N % 4 == 0:
        same, _ = _segmented_linear(ctx, dtype, M, N, K, seg1, seg2, 0, resid, seed=M + N + K, act=0 if resid else 1)
        assert torch.equal(got, same)


def test_linear_segments_argument_errors():
    """me_op_linear_segments: a segment without weights, or boundaries out of order, is ME_ERR_BAD_ARG, not a device fault"""
    import ctypes as C
    ctx = ctx_for("tiny", "f16")
    a = dev16(torch.randn(512, 128), "f16")
    w = dev16(torch.randn(256, 128), "f16")
    b = torch.zeros(256, device="cuda")
    out = torch.empty(512, 256, dtype=torch.float16, device="cuda")
    one = (C.c_void_p * 3)(w.data_ptr(), None, None)
    bias = (C.c_void_p * 3)(b.data_ptr(), b.data_ptr(), b.data_ptr())
    call = lambda s1, s2, ws: ctx.lib.me_op_linear_segments(ctx.handle, 512, 256, 128, ptr(a), s1, s2, ws, bias, None, ptr(out), None, 0, 10)
    assert call(0, 0, one) == 0
    assert call(256, 0, one) == 1            # segment 1 has no weights
    both = (C.c_void_p * 3)(w.data_ptr(), w.data_ptr(), None)
    assert call(256, 0, both) == 0
    assert call(256, 128, both) == 1         # seg2 <= seg1
    assert call(600, 0, both) == 1           # beyond M
    ctx.synchronize()


def test_tall_tile_352_at_the_step_shapes():
    """proj / fc2 / fc1 of one 1536x1536 image (M = 21760 in segments of 768 / 768 / 20224 rows): one exact round of
    256 tall tiles (proj, fc2), four rounds (fc1); bit-identical to the 256-row tile."""
    ctx = ctx_for("tiny", "f16")
    for (N, K, resid) in ((1024, 1024, True), (1024, 4096, True), (4096, 1024, False)):
        got, ref = _segmented_linear(ctx, "f16", 21760, N, K, 768, 1536, 10, resid, seed=N + K, act=0 if resid else 1)
        same, _ = _segmented_linear(ctx, "f16", 21760, N, K, 768, 1536, 0, resid, seed=N + K, act=0 if resid else 1)
        assert torch.equal(got, same)
        if resid:
            assert max_abs_rel(got, ref) < 1e-4        # f32 accumulation over K = 4096
        del got, ref, same


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tokens,windows,heads", [(577, 3, 2), (65, 5, 2), (577, 1, 16), (130, 2, 1)])
def test_attention(dtype, tokens, windows, heads):
    ctx = ctx_for("tiny", dtype)
    C = heads * 64
    g = torch.Generator().manual_seed(tokens + heads)
    qkv = dev16(torch.randn(windows * tokens, 3 * C, generator=g) * 1.5, dtype)
    out = torch.empty(windows * tokens, C, dtype=TORCH16[dtype], device="cuda")
    _check(ctx, ctx.lib.me_op_attention(ctx.handle, ptr(qkv), ptr(out), windows, tokens, heads))
    ctx.synchronize()
    x = qkv.double().reshape(windows, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0] * 0.125, x[1], x[2]                       # vit.rs:63-71
    ref = (torch.softmax(q @ k.transpose(3, 2), dim=3) @ v).transpose(1, 2).reshape(windows * tokens, C)
    # P and the output are rounded to 16 bit once each: <= ~1 ulp of the largest output, and an
    # rms error of a fraction of one rounding
    assert max_err_over_max(out.float(), ref) < 2 * OUT_EPS[dtype]
    assert rel_l2(out.float(), ref) < 1.5 * OUT_EPS[dtype]


QSCALE = 0.125 * 1.4426950408889634      # common.h kAttnQScale: 1/sqrt(64) * log2(e)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tokens,windows,heads", [(577, 3, 2), (65, 5, 2), (577, 1, 16), (130, 2, 1), (64, 2, 2), (1, 3, 1)])
def test_attention_prescaled(dtype, tokens, windows, heads):
    """The form the forward pass runs: Q arrives multiplied by 1/sqrt(64) * log2(e) (the qkv linear's epilogue does
    that before its one rounding) and the kernel's softmax runs on exp2 with the reference point inside the MFMA
    accumulator.  Reference: fp64 softmax of the 16-bit operands as given (exp2 of the scaled scores)."""
    ctx = ctx_for("tiny", dtype)
    C = heads * 64
    g = torch.Generator().manual_seed(tokens + 3 * heads)
    x = torch.randn(windows * tokens, 3 * C, generator=g) * 1.5
    x[:, :C] *= QSCALE
    qkv = dev16(x, dtype)
    out = torch.empty(windows * tokens, C, dtype=TORCH16[dtype], device="cuda")
    _check(ctx, ctx.lib.me_op_attention_prescaled(ctx.handle, ptr(qkv), ptr(out), windows, tokens, heads))
    ctx.synchronize()
    xx = qkv.double().reshape(windows, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = xx[0], xx[1], xx[2]
    s = (q @ k.transpose(3, 2)) * math.log(2.0)             # exp2(s) = exp(s ln 2)
    ref = (torch.softmax(s, dim=3) @ v).transpose(1, 2).reshape(windows * tokens, C)
    assert max_err_over_max(out.float(), ref) < 2 * OUT_EPS[dtype]
    assert rel_l2(out.float(), ref) < 1.5 * OUT_EPS[dtype]


def test_attention_prescaled_token_counts():
    """The forward pass's kernel over the query-block geometry: workgroups of six waves x 32 queries, the one query beyond the
    whole blocks (tokens = 32 n + 1: 33, 65, 193, 577, 609) on the trailing workgroups' vector-pipe path
    (attention_extra_query), every other remainder as one more partly filled block; key tiles with 1 to 64 valid rows;
    a guard row behind the output stays untouched."""
    ctx = ctx_for("tiny", "f16")
    for tokens in [1, 2, 31, 32, 33, 34, 63, 64, 65, 66, 97, 129, 191, 192, 193, 194, 225, 257, 385, 576, 577, 578, 609, 641]:
        windows, heads = (3, 2) if tokens < 400 else (2, 1)
        C = heads * 64
        g = torch.Generator().manual_seed(1000 + tokens)
        x = torch.randn(windows * tokens, 3 * C, generator=g) * 1.5
        x[:, :C] *= QSCALE
        qkv = dev16(x, "f16")
        out = torch.full((windows * tokens + 1, C), 7.0, dtype=torch.float16, device="cuda")
        _check(ctx, ctx.lib.me_op_attention_prescaled(ctx.handle, ptr(qkv), ptr(out), windows, tokens, heads))
        ctx.synchronize()
        xx = qkv.double().reshape(windows, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
        q, k, v = xx[0], xx[1], xx[2]
        ref = (torch.softmax((q @ k.transpose(3, 2)) * math.log(2.0), dim=3) @ v).transpose(1, 2).reshape(windows * tokens, C)
        assert bool((out[windows * tokens] == 7.0).all()), tokens
        assert max_err_over_max(out[:-1].float(), ref) < 2 * OUT_EPS["f16"], tokens
        assert rel_l2(out[:-1].float(), ref) < 1.5 * OUT_EPS["f16"], tokens
        # the last query of every window on its own (the vector-pipe path when tokens = 32 n + 1)
        last = out[:-1].float().reshape(windows, tokens, C)[:, -1]
        assert rel_l2(last, ref.reshape(windows, tokens, C)[:, -1]) < 2 * OUT_EPS["f16"], tokens


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_pipelined_kernel_is_bit_identical(dtype, monkeypatch):
    """attention2p_kernel (ME_ATT_V=4: tile t + 1's score MFMAs issued in front of tile t's softmax, three ring slots) performs the
    arithmetic of attention2_kernel in the same order: bit-identical outputs over the token-count geometry (ragged last tiles,
    the single tail key, one-tile inputs) at two and three waves per SIMD -- as long as the reference point does not move
    after the first tile.  Where it does (planted dominating keys), the block computed ahead is corrected by a subtraction
    where attention2_kernel starts its accumulator from the new reference point: the same value to the last rounding, so
    there the two are held to each other within the output's rounding and both to the fp64 softmax."""
    ctx = ctx_for("tiny", dtype)
    for tokens, windows, heads, planted in [(577, 3, 2, False), (65, 5, 2, False), (1, 3, 1, False), (2, 2, 1, False), (63, 2, 2, False),
                                            (64, 2, 2, False), (66, 2, 1, False), (128, 2, 2, False), (129, 2, 1, False), (130, 2, 1, False),
                                            (193, 2, 2, False), (300, 2, 1, False), (576, 1, 2, False), (578, 1, 1, False),
                                            (640, 1, 1, False), (641, 1, 2, False), (577, 2, 16, False), (577, 2, 2, True),
                                            (300, 2, 1, True), (641, 1, 2, True)]:
        C = heads * 64
        g = torch.Generator().manual_seed(9000 + tokens + heads)
        x = torch.randn(windows * tokens, 3 * C, generator=g) * (1.5 if planted else 1.0)
        if planted:     # dominating keys in late tiles: the reference point moves after the next block was computed
            x[tokens // 2 + 70, C:2 * C] *= 6.0
            x[tokens - 1, C:2 * C] *= 5.0
        x[:, :C] *= QSCALE
        qkv = dev16(x, dtype)
        outs = []
        for env in ({}, {"ME_ATT_V": "4"}, {"ME_ATT_V": "4", "ME_ATT_MINW": "2"}):
            for k in ("ME_ATT_V", "ME_ATT_MINW"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            out = torch.full((windows * tokens + 1, C), 7.0, dtype=TORCH16[dtype], device="cuda")
            _check(ctx, ctx.lib.me_op_attention_prescaled(ctx.handle, ptr(qkv), ptr(out), windows, tokens, heads))
            ctx.synchronize()
            outs.append(out.clone())
        assert bool(torch.isfinite(outs[0].float()).all()), tokens
        assert torch.equal(outs[1], outs[2]), tokens                       # two and three waves per SIMD: the same code
        if not planted:
            assert torch.equal(outs[0], outs[1]), tokens
        else:
            xx = qkv.double().reshape(windows, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
            ref = (torch.softmax((xx[0] @ xx[1].transpose(3, 2)) * math.log(2.0), dim=3) @ xx[2]).transpose(1, 2).reshape(windows * tokens, C)
            for o_ in outs[:2]:
                assert max_err_over_max(o_[:-1].float(), ref) < 2 * OUT_EPS[dtype], tokens
                assert rel_l2(o_[:-1].float(), ref) < 1.5 * OUT_EPS[dtype], tokens
            assert rel_l2(outs[1][:-1].float(), outs[0][:-1].float()) < OUT_EPS[dtype], tokens
    for k in ("ME_ATT_V", "ME_ATT_MINW"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention3_recut_kernel(dtype, monkeypatch):
    """Round 5's re-cut of the forward pass's attention (csrc/attention3.hip, ME_ATT_V=3: 48 queries per wave on 16x16x32 MFMAs,
    persistent workgroups over 192-query items, the query beyond whole wave units on the vector pipe).  It lost its A/B in the
    step (profiles/r05_attention_recut_ab.txt) and is not the default; it stays selectable, so it stays tested: the geometry
    sweep of test_attention_prescaled_token_counts' kind against fp64, more items than resident workgroups (items > grid via
    ME_ATT_GRID=8), and agreement with attention2_kernel to the rounding of the 16-bit output."""
    ctx = ctx_for("tiny", dtype)
    for tokens, windows, heads, grid in [(577, 3, 2, 0), (577, 2, 16, 8), (65, 5, 2, 0), (49, 3, 1, 8), (97, 2, 2, 0), (48, 2, 1, 0),
                                         (1, 3, 1, 0), (193, 2, 2, 8), (640, 1, 1, 0), (641, 1, 2, 0)]:
        C = heads * 64
        g = torch.Generator().manual_seed(7000 + tokens + heads)
        x = torch.randn(windows * tokens, 3 * C, generator=g) * 1.5
        x[:, :C] *= QSCALE
        qkv = dev16(x, dtype)
        outs = {}
        for v in ("2", "3"):
            monkeypatch.setenv("ME_ATT_V", v)
            if grid:
                monkeypatch.setenv("ME_ATT_GRID", str(grid))
            else:
                monkeypatch.delenv("ME_ATT_GRID", raising=False)
            out = torch.full((windows * tokens + 1, C), 7.0, dtype=TORCH16[dtype], device="cuda")
            _check(ctx, ctx.lib.me_op_attention_prescaled(ctx.handle, ptr(qkv), ptr(out), windows, tokens, heads))
            ctx.synchronize()
            assert bool((out[windows * tokens] == 7.0).all()), (tokens, v)
            outs[v] = out[:-1].float()
        xx = qkv.double().reshape(windows, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
        q, k, v_ = xx[0], xx[1], xx[2]
        ref = (torch.softmax((q @ k.transpose(3, 2)) * math.log(2.0), dim=3) @ v_).transpose(1, 2).reshape(windows * tokens, C)
        assert max_err_over_max(outs["3"], ref) < 2 * OUT_EPS[dtype], tokens
        assert rel_l2(outs["3"], ref) < 1.5 * OUT_EPS[dtype], tokens
        assert rel_l2(outs["3"], outs["2"]) < 1.5 * OUT_EPS[dtype], tokens
    monkeypatch.delenv("ME_ATT_V", raising=False)
    monkeypatch.delenv("ME_ATT_GRID", raising=False)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_running_max_branches(dtype):
    """The rescale branch of both kernels (cdna_hip_programming.md rule 26: a rare data-dependent branch needs an input
    that forces it): keys that dominate their query are planted in EVERY 64-key tile with growing scores (8 natural
    units = 11.5 exp2 units per tile: past the deferred-maximum threshold of 8 every time), so each tile shifts the
    reference point -- for query 40 and, more gently, for every query correlated with it; one query sees only strongly
    NEGATIVE scores (the first tile must set the reference point below zero); one query has its maximum in the single
    tail key of 577 = 9 x 64 + 1.  ME_ATT_THR=0 (exact running maximum) must agree with the shipped threshold."""
    ctx = ctx_for("tiny", dtype)
    tokens, heads, C = 577, 1, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(tokens, 3 * C, generator=g)
    qi = 40
    for t in range(9):                                   # key 64 t + 5 scores (t + 2) * 8 against query 40
        x[64 * t + 5, C:2 * C] = x[qi, 0:C] * (t + 2) * 64.0 / float(x[qi, 0:C].pow(2).sum())
    x[576, C:2 * C] = x[200, 0:C] * 5.0                  # query 200 attends to the tail key
    u = torch.full((C,), 0.125)                          # a unit vector: every key gets 4 u (a constant per query) ...
    x[:, C:2 * C] += 4.0 * u
    x[100, 0:C] = -175.0 * u                             # ... and query 100 scores every key at -87 +- 22 (natural units)
    qkv = dev16(x, dtype)
    for fn in ("me_op_attention", "me_op_attention_prescaled"):
        xin = qkv.clone()
        if fn.endswith("prescaled"):
            xin[:, :C] = (xin[:, :C].float() * QSCALE).to(xin.dtype)
        out = torch.empty(tokens, C, dtype=TORCH16[dtype], device="cuda")
        _check(ctx, getattr(ctx.lib, fn)(ctx.handle, ptr(xin), ptr(out), 1, tokens, heads))
        ctx.synchronize()
        xx = xin.double()
        q, k, v = xx[:, :C], xx[:, C:2 * C], xx[:, 2 * C:]
        s = q @ k.T * (math.log(2.0) if fn.endswith("prescaled") else 0.125)
        ref = torch.softmax(s, dim=1) @ v
        assert bool(torch.isfinite(out.float()).all()), fn
        assert max_err_over_max(out.float(), ref) < 2 * OUT_EPS[dtype], fn
        assert float((out[qi].double() - v[64 * 8 + 5]).abs().max()) < 0.03, fn    # the last planted key wins
        assert float((out[200].double() - v[576]).abs().max()) < 0.03, fn
        if fn.endswith("prescaled"):
            import os
            exact = torch.empty_like(out)
            os.environ["ME_ATT_THR"] = "0"
            try:
                _check(ctx, getattr(ctx.lib, fn)(ctx.handle, ptr(xin), ptr(exact), 1, tokens, heads))
                ctx.synchronize()
            finally:
                del os.environ["ME_ATT_THR"]
            assert max_err_over_max(out.float(), exact.double()) < 2 * OUT_EPS[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C,cfg", [(705, 128, -1), (2308, 1024, -1), (2308, 1024, 10), (1024, 256, 0)])
def test_linear_scaled_cols(dtype, M, C, cfg):
    """The qkv linear with its Q columns scaled in the epilogue: out16 = round16((A W^T + bias) * qscale) for the first C
    columns, round16(A W^T + bias) for the rest -- against fp64, and the unscaled columns bit for bit me_op_linear's."""
    ctx = ctx_for("tiny", dtype)
    g = torch.Generator().manual_seed(M + C)
    A = dev16(torch.randn(M, C, generator=g), dtype)
    W = dev16(torch.randn(3 * C, C, generator=g) / C ** 0.5, dtype)
    b = (torch.randn(3 * C, generator=g) * 0.5).cuda()
    got = torch.empty(M, 3 * C, dtype=TORCH16[dtype], device="cuda")
    plain = torch.empty_like(got)
    _check(ctx, ctx.lib.me_op_linear_scaled_cols(ctx.handle, M, 3 * C, C, ptr(A), ptr(W), ptr(b), ptr(got), C, QSCALE, cfg))
    _check(ctx, ctx.lib.me_op_linear(ctx.handle, M, 3 * C, C, ptr(A), ptr(W), ptr(b), ptr(plain), None, 0, cfg))
    ctx.synchronize()
    ref = A.double() @ W.double().T + b.double()
    ref[:, :C] *= QSCALE
    assert torch.equal(got[:, C:], plain[:, C:])
    assert max_err_over_max(got.float(), ref) < 1.5 * OUT_EPS[dtype]
    assert rel_l2(got[:, :C].float(), ref[:, :C]) < OUT_EPS[dtype]


def test_attention_token_counts():
    """Every way the sequence can end: one token, one short of / exactly / one past a 64-key tile (the single
    tail key is folded in as a rank-one update) and a 128-query block, the model's 577 and a few in between; a
    guard row behind the output stays untouched."""
    ctx = ctx_for("tiny", "f16")
    for tokens in [1, 2, 31, 63, 64, 65, 66, 127, 128, 129, 191, 192, 193, 256, 257, 300, 576, 577, 578, 640, 641]:
        windows, heads = (2, 2) if tokens < 400 else (1, 1)
        C = heads * 64
        g = torch.Generator().manual_seed(tokens)
        qkv = dev16(torch.randn(windows * tokens, 3 * C, generator=g) * 1.5, "f16")
        out = torch.full((windows * tokens + 1, C), 7.0, dtype=torch.float16, device="cuda")
        torch.cuda.synchronize()
        _check(ctx, ctx.lib.me_op_attention(ctx.handle, ptr(qkv), ptr(out), windows, tokens, heads))
        ctx.synchronize()
        x = qkv.double().reshape(windows, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4)
        q, k, v = x[0] * 0.125, x[1], x[2]
        ref = (torch.softmax(q @ k.transpose(3, 2), dim=3) @ v).transpose(1, 2).reshape(windows * tokens, C)
        assert bool((out[windows * tokens] == 7.0).all()), tokens
        assert max_err_over_max(out[:-1].float(), ref) < 2 * OUT_EPS["f16"], tokens
        assert rel_l2(out[:-1].float(), ref) < 1.5 * OUT_EPS["f16"], tokens


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_large_scores(dtype):
    """one key dominates from the middle of the sequence on: exercises the running-max rescale"""
    ctx = ctx_for("tiny", dtype)
    tokens, heads, C = 577, 1, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(tokens, 3 * C, generator=g)
    x[300, C:2 * C] = x[17, 0:C] * 6.0       # key 300 aligned with query 17, score ~ 6*64/8
    qkv = dev16(x, dtype)
    out = torch.empty(tokens, C, dtype=TORCH16[dtype], device="cuda")
    _check(ctx, ctx.lib.me_op_attention(ctx.handle, ptr(qkv), ptr(out), 1, tokens, heads))
    ctx.synchronize()
    xx = qkv.double()
    q, k, v = xx[:, :C] * 0.125, xx[:, C:2 * C], xx[:, 2 * C:]
    ref = torch.softmax(q @ k.T, dim=1) @ v
    assert max_err_over_max(out.float(), ref) < 2 * OUT_EPS[dtype]
    assert float((out[17].double() - v[300]).abs().max()) < 0.02   # query 17 attends to key 300 only


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dim", [128, 1024])
def test_layernorm(dtype, dim):
    ctx = ctx_for("tiny", dtype)
    rows = 1001
    g = torch.Generator().manual_seed(dim)
    x = (torch.randn(rows, dim, generator=g) * 3 + 0.5).cuda()
    w = (1 + 0.02 * torch.randn(dim, generator=g)).cuda()
    b = (0.02 * torch.randn(dim, generator=g)).cuda()
    y32 = torch.empty_like(x)
    y16 = torch.empty(rows, dim, dtype=TORCH16[dtype], device="cuda")
    _check(ctx, ctx.lib.me_op_layernorm(ctx.handle, ptr(x), ptr(w), ptr(b), ptr(y16), ptr(y32), rows, dim, 1e-5))
    ctx.synchronize()
    ref = F.layer_norm(x.double(), (dim,), w.double(), b.double(), 1e-5)
    assert float((y32.double() - ref).abs().max()) < 2e-5
    assert float((y16.double() - ref).abs().max()) < 4 * OUT_EPS[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    dict(B=2, H=24, W=24, Cin=64, Cout=128, k=3, stride=1),
    dict(B=1, H=40, W=40, Cin=128, Cout=256, k=3, stride=1, res=True, relu=True),
    dict(B=2, H=16, W=16, Cin=256, Cout=128, k=3, stride=2),
    dict(B=1, H=12, W=12, Cin=64, Cout=32, k=3, stride=2),
    dict(B=1, H=20, W=20, Cin=128, Cout=64, k=1, stride=1),
])
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4])
def test_conv2d(dtype, case, cfg):
    ctx = ctx_for("tiny", dtype)
    B, H, W, Cin, Cout, k, s = (case[n] for n in ("B", "H", "W", "Cin", "Cout", "k", "stride"))
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    bias = torch.randn(Cout, generator=g)
    xb = bordered(x, dtype)
    w16 = dev16(pack_conv(w), dtype)
    Ho, Wo = H // s, W // s
    res = torch.randn(B * Ho * Wo, Cout, generator=g).cuda() if case.get("res") else None
    res2 = torch.randn(B * Ho * Wo, Cout, generator=g).cuda() if case.get("res") else None
    out32 = torch.empty(B * Ho * Wo, Cout, dtype=torch.float32, device="cuda")
    out16 = torch.zeros(B, Ho + 2, Wo + 2, Cout, dtype=TORCH16[dtype], device="cuda")
    act = 2 if case.get("relu") else 0
    _check(ctx, ctx.lib.me_op_conv2d(ctx.handle, ptr(xb), B, H, W, Cin, ptr(w16), Cout, k, s, ptr(bias.cuda()),
                                     ptr(res), ptr(res2), ptr(out32), ptr(out16), 1, act, 0, cfg))
    ctx.synchronize()
    x16 = xb[:, 1:H + 1, 1:W + 1, :].permute(0, 3, 1, 2).double().cpu()
    w16f = dev16(w, dtype).double().cpu()
    ref = F.conv2d(x16, w16f, bias.double(), stride=s, padding=(k - 1) // 2)
    ref = ref.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)
    if res is not None:
        ref = ref + res.double().cpu() + res2.double().cpu()
    assert max_abs_rel(out32.cpu(), ref) < 3e-5
    ref16 = F.relu(ref) if act else ref
    got16 = out16[:, 1:Ho + 1, 1:Wo + 1, :].reshape(B * Ho * Wo, Cout).float().cpu()
    assert max_abs_rel(got16, ref16) < 6 * OUT_EPS[dtype] * 4
    # the zero border must stay zero
    assert float(out16[:, 0].abs().max()) == 0 and float(out16[:, :, 0].abs().max()) == 0
    assert float(out16[:, -1].abs().max()) == 0 and float(out16[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    dict(B=2, H=48, W=32, Cin=64, Cout=256),
    dict(B=1, H=32, W=64, Cin=256, Cout=512, res=True, relu=True),
    dict(B=1, H=16, W=16, Cin=128, Cout=256, relu=True),
    dict(B=1, H=272, W=272, Cin=128, Cout=256, res=True),       # 289 pixel tiles: workgroups walk on to a second tile
    dict(B=3, H=96, W=112, Cin=192, Cout=256, res=True, relu=True),
    dict(B=1, H=384, W=384, Cin=64, Cout=256, res=True, relu=True),   # the decoder's 384 x 384 level: 768 tiles of 12 x 16
    dict(B=2, H=48, W=80, Cin=128, Cout=128, relu=True, halo=12),     # tile config 12: 128 output channels per tile
    dict(B=1, H=272, W=272, Cin=64, Cout=384, res=True, halo=12),     # 289 pixel tiles x 3 channel tiles
])
def test_conv3x3_halo_tile(dtype, case):
    """Tile config 9 (gemm_core.h conv_halo_kernel): 16 x 16 pixel tiles whose 18 x 18 halo is staged once per 64 input
    channels and read at shifted rows by the nine taps -- against torch.nn.functional.conv2d on the same 16-bit operands,
    with the f32 residual inputs, the fused ReLU, the zero-bordered 16-bit output and the split [hi | lo] output form;
    maps larger than one round of tiles; the zero border stays zero.  The automatic choice picks it for such shapes."""
    ctx = ctx_for("tiny", dtype)
    B, H, W, Cin, Cout = (case[n] for n in ("B", "H", "W", "Cin", "Cout"))
    g = torch.Generator().manual_seed(H * W + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    xb, w16, bias_d = bordered(x, dtype), dev16(pack_conv(w), dtype), bias.cuda()
    res = torch.randn(B * H * W, Cout, generator=g).cuda() if case.get("res") else None
    res2 = torch.randn(B * H * W, Cout, generator=g).cuda() if case.get("res") else None
    act = 2 if case.get("relu") else 0
    x16 = xb[:, 1:H + 1, 1:W + 1, :].permute(0, 3, 1, 2).double().cpu()
    ref = F.conv2d(x16, dev16(w, dtype).double().cpu(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    if res is not None:
        ref = ref + res.double().cpu() + res2.double().cpu()
    ref16 = F.relu(ref) if act else ref
    outs = {}
    HALO = case.get("halo", 9)
    for cfg in (HALO, -1, 0):
        out32 = torch.empty(B * H * W, Cout, dtype=torch.float32, device="cuda")
        out16 = torch.zeros(B, H + 2, W + 2, Cout, dtype=TORCH16[dtype], device="cuda")
        torch.cuda.synchronize()
        _check(ctx, ctx.lib.me_op_conv2d(ctx.handle, ptr(xb), B, H, W, Cin, ptr(w16), Cout, 3, 1, ptr(bias_d), ptr(res), ptr(res2),
                                         ptr(out32), ptr(out16), 1, act, 0, cfg))
        ctx.synchronize()
        assert max_abs_rel(out32.cpu(), ref) < 3e-5, cfg
        got16 = out16[:, 1:H + 1, 1:W + 1, :].reshape(B * H * W, Cout).float().cpu()
        assert max_abs_rel(got16, ref16) < 6 * OUT_EPS[dtype] * 4, cfg
        assert float(out16[:, 0].abs().max()) == 0 and float(out16[:, :, 0].abs().max()) == 0
        assert float(out16[:, -1].abs().max()) == 0 and float(out16[:, :, -1].abs().max()) == 0
        outs[cfg] = (out32, out16)
    # every convolution tile walks K in the same order (input-channel slab outermost, taps inside: gemm_core.h SlabWalk),
    # so the halo tile and the implicit-GEMM tile agree bit for bit -- the result does not depend on which tile the
    # problem size selects
    outs[9] = outs[HALO]
    assert torch.equal(outs[9][0], outs[0][0]) and torch.equal(outs[9][1], outs[0][1])
    if H % 12 == 0 and HALO == 9:
        # tile config 11: the halo tile on 12 x 16 pixels (chosen where it fills the rounds of 256 workgroups better)
        out32 = torch.empty_like(outs[9][0])
        out16 = torch.zeros_like(outs[9][1])
        torch.cuda.synchronize()
        _check(ctx, ctx.lib.me_op_conv2d(ctx.handle, ptr(xb), B, H, W, Cin, ptr(w16), Cout, 3, 1, ptr(bias_d), ptr(res), ptr(res2),
                                         ptr(out32), ptr(out16), 1, act, 0, 11))
        ctx.synchronize()
        assert torch.equal(out32, outs[9][0]) and torch.equal(out16, outs[9][1])
    # a repeated launch of the halo tile reproduces itself bit for bit
    out32 = torch.empty_like(outs[9][0])
    out16 = torch.zeros_like(outs[9][1])
    torch.cuda.synchronize()
    _check(ctx, ctx.lib.me_op_conv2d(ctx.handle, ptr(xb), B, H, W, Cin, ptr(w16), Cout, 3, 1, ptr(bias_d), ptr(res), ptr(res2),
                                     ptr(out32), ptr(out16), 1, act, 0, HALO))
    ctx.synchronize()
    assert torch.equal(out32, outs[9][0]) and torch.equal(out16, outs[9][1])


def test_conv2d_random_shapes():
    """40 random convolutions: non-square and odd maps, 1x1 and 3x3, stride 1 and 2, any tile configuration, with
    and without the two f32 residual inputs and the fused ReLU -- against torch.nn.functional.conv2d on the same
    16-bit operands; the zero border of the 16-bit output stays zero."""
    import random
    ctx = ctx_for("tiny", "f16")
    rnd = random.Random(777)
    ncfg = ctx.lib.me_op_gemm_config_count()
    for it in range(40):
        k = rnd.choice([1, 3, 3])
        s = rnd.choice([1, 1, 2])
        B = rnd.choice([1, 1, 2, 3])
        H, W = (2 * rnd.randrange(1, 14), 2 * rnd.randrange(1, 14)) if s == 2 else (rnd.randrange(1, 27), rnd.randrange(1, 27))
        Cin, Cout = 64 * rnd.choice([1, 2, 3, 4]), 4 * rnd.choice([1, 2, 8, 9, 16, 33, 64])
        cfg = rnd.choice([-1, 0, 1, 2, 3, 4, 5, 7, 8])      # 6 is linear-only, 9 the halo tile (its own test below)
        with_res, relu = rnd.random() < 0.5, rnd.random() < 0.5
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        bias = torch.randn(Cout, generator=g)
        Ho, Wo = H // s, W // s
        res = torch.randn(B * Ho * Wo, Cout, generator=g).cuda() if with_res else None
        res2 = torch.randn(B * Ho * Wo, Cout, generator=g).cuda() if with_res else None
        out32 = torch.empty(B * Ho * Wo, Cout, dtype=torch.float32, device="cuda")
        out16 = torch.zeros(B, Ho + 2, Wo + 2, Cout, dtype=torch.float16, device="cuda")
        xb, w16, bias_d = bordered(x, "f16"), dev16(pack_conv(w), "f16"), bias.cuda()
        torch.cuda.synchronize()
        _check(ctx, ctx.lib.me_op_conv2d(ctx.handle, ptr(xb), B, H, W, Cin, ptr(w16), Cout, k, s, ptr(bias_d),
                                         ptr(res), ptr(res2), ptr(out32), ptr(out16), 1, 2 if relu else 0, 0, cfg))
        ctx.synchronize()
        x16 = xb[:, 1:H + 1, 1:W + 1, :].permute(0, 3, 1, 2).double().cpu()
        ref = F.conv2d(x16, dev16(w, "f16").double().cpu(), bias.double(), stride=s, padding=(k - 1) // 2)
        ref = ref.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)
        if with_res:
            ref = ref + res.double().cpu() + res2.double().cpu()
        tag = (it, B, H, W, Cin, Cout, k, s, cfg, with_res, relu)
        assert max_abs_rel(out32.cpu(), ref) < 3e-5, tag
        got16 = out16[:, 1:Ho + 1, 1:Wo + 1, :].reshape(B * Ho * Wo, Cout).float().cpu()
        assert max_abs_rel(got16, F.relu(ref) if relu else ref) < 6 * OUT_EPS["f16"] * 4, tag
        assert float(out16[:, 0].abs().max()) == 0 and float(out16[:, :, 0].abs().max()) == 0, tag
        assert float(out16[:, -1].abs().max()) == 0 and float(out16[:, :, -1].abs().max()) == 0, tag


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4])
def test_conv_transpose(dtype, cfg):
    ctx = ctx_for("tiny", dtype)
    B, H, W, Cin, Cout = 2, 18, 18, 128, 64
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin)
    bias = torch.randn(Cout, generator=g)
    x16 = dev16(x.permute(0, 2, 3, 1).reshape(B * H * W, Cin), dtype)
    w16 = dev16(pack_convt(w), dtype)
    out32 = torch.empty(B, 2 * H, 2 * W, Cout, dtype=torch.float32, device="cuda")
    out16 = torch.zeros(B, 2 * H + 2, 2 * W + 2, Cout, dtype=TORCH16[dtype], device="cuda")
    _check(ctx, ctx.lib.me_op_conv_transpose2x2(ctx.handle, ptr(x16), B, H, W, Cin, ptr(w16), Cout,
                                                ptr(bias.cuda()), ptr(out32), ptr(out16), 1, cfg))
    ctx.synchronize()
    xr = x16.double().cpu().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(xr, dev16(w, dtype).double().cpu(), bias.double(), stride=2).permute(0, 2, 3, 1)
    assert max_abs_rel(out32.cpu(), ref) < 2e-5
    assert max_abs_rel(out16[:, 1:-1, 1:-1, :].float().cpu(), ref) < 6 * OUT_EPS[dtype] * 4
    assert float(out16[:, 0].abs().max()) == 0 and float(out16[:, :, -1].abs().max()) == 0


def _vit_block_gpu(ctx, x32, w, i, C, T, windows, heads, dtype):
    """One Block::forward (vit.rs:163-170) from the kernel-level entry points, in place on x32 [rows, C]"""
    rows = x32.shape[0]
    t16 = TORCH16[dtype]
    xn = torch.empty(rows, C, dtype=t16, device="cuda")
    qkv = torch.empty(rows, 3 * C, dtype=t16, device="cuda")
    att = torch.empty(rows, C, dtype=t16, device="cuda")
    hid = torch.empty(rows, 4 * C, dtype=t16, device="cuda")
    p = f"encoder.patch_encoder.blocks.{i}."
    d = {k: torch.as_tensor(w[p + k]).cuda() for k in
         ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
          "ls1.gamma", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias",
          "ls2.gamma")}
    f = {k: v.float().contiguous() for k, v in d.items()}
    h = {k: v.to(t16).contiguous() for k, v in d.items() if k.endswith("weight") and v.dim() == 2}
    L = ctx.lib
    _check(ctx, L.me_op_layernorm(ctx.handle, ptr(x32), ptr(f["norm1.weight"]), ptr(f["norm1.bias"]), ptr(xn), None, rows, C, 1e-5))
    _check(ctx, L.me_op_linear(ctx.handle, rows, 3 * C, C, ptr(xn), ptr(h["attn.qkv.weight"]), ptr(f["attn.qkv.bias"]), ptr(qkv), None, 0, -1))
    _check(ctx, L.me_op_attention(ctx.handle, ptr(qkv), ptr(att), windows, T, heads))
    _check(ctx, L.me_op_linear_residual(ctx.handle, rows, C, C, ptr(att), ptr(h["attn.proj.weight"]), ptr(f["attn.proj.bias"]), ptr(f["ls1.gamma"]), ptr(x32), -1))
    _check(ctx, L.me_op_layernorm(ctx.handle, ptr(x32), ptr(f["norm2.weight"]), ptr(f["norm2.bias"]), ptr(xn), None, rows, C, 1e-5))
    _check(ctx, L.me_op_linear(ctx.handle, rows, 4 * C, C, ptr(xn), ptr(h["mlp.fc1.weight"]), ptr(f["mlp.fc1.bias"]), ptr(hid), None, 1, -1))
    _check(ctx, L.me_op_linear_residual(ctx.handle, rows, C, 4 * C, ptr(hid), ptr(h["mlp.fc2.weight"]), ptr(f["mlp.fc2.bias"]), ptr(f["ls2.gamma"]), ptr(x32), -1))
    ctx.synchronize()
    return qkv, hid


def test_outlier_activations_through_a_block():
    """All other parity runs on N(0, 1/fan_in) weights; real DINOv2-L checkpoints carry massive activations in a
    few channels of the residual stream.  What the f16 path does with them, one Block (vit.rs:163-170) on the
    tiny geometry:
      * outliers of 3e4 in the f32 residual stream (beyond anything LayerNorm lets through to a GEMM operand):
        the block's output matches the fp64 oracle as closely as without them -- the stream, the LayerNorm
        statistics and the residual adds are f32, and what reaches the 16-bit operands is normalised;
      * 16-bit operands up to the f16 maximum are exact: a qkv pre-activation of 6.0e4 is stored as 6.0e4;
      * past 65504 the 16-bit copy of a GEMM output is +-inf (round-to-nearest conversion, no saturation), the
        attention / MLP that consume it go non-finite and so does the depth: the overflow is loud (bench.py and
        the pipeline tests assert finiteness), never a silently clamped value.  ME_DTYPE_BF16 (f32's exponent
        range) is the operand type for a checkpoint that needs that range: the same block stays finite there."""
    import matrix_eyes_amd as m
    from oracle import depth_pro_oracle as O
    from util import oracle_cfg, weights_for
    cfg = m.ModelConfig.tiny()
    w = {k: torch.as_tensor(v) for k, v in weights_for("tiny").items()}
    C, T, heads, windows = cfg.embed_dim, cfg.tokens, cfg.num_heads, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(windows, T, C, generator=g)
    spiky = x.clone()
    spiky[:, 0, 7] = 3.0e4          # a massive activation in the cls token of every window
    spiky[:, 5, 100] = -2.0e4
    ocfg = oracle_cfg(cfg, dtype=torch.float64)
    w64 = {k: v.double() for k, v in w.items()}
    ctx = ctx_for("tiny", "f16")
    ctx.status_flags()
    for name, inp in (("plain", x), ("spiky", spiky)):
        ref = O.block_forward(inp.double(), w64, "encoder.patch_encoder.blocks.0.", ocfg)
        got = inp.reshape(-1, C).contiguous().cuda()
        _vit_block_gpu(ctx, got, w, 0, C, T, windows, heads, "f16")
        delta_ref = ref.reshape(-1, C) - inp.reshape(-1, C).double()       # what the block adds
        delta_got = got.cpu().double() - inp.reshape(-1, C).double()
        err = float((delta_got - delta_ref).norm() / delta_ref.norm())
        print("block update rel-L2", name, err)
        assert torch.isfinite(got).all() and err < 1.5e-3
    # operands at and past the f16 maximum: qkv weights scaled so that one pre-activation column reaches 6.0e4 / 7e4
    big = dict(w)
    for target, finite in ((6.0e4, True), (7.0e4, False)):
        wq = w["encoder.patch_encoder.blocks.0.attn.qkv.weight"].float().clone()
        bq = w["encoder.patch_encoder.blocks.0.attn.qkv.bias"].float().clone()
        wq[2 * C + 3] = 0.0
        bq[2 * C + 3] = target      # a value column (v of head 0, dim 3) pinned to `target`
        big["encoder.patch_encoder.blocks.0.attn.qkv.weight"] = wq.half()
        big["encoder.patch_encoder.blocks.0.attn.qkv.bias"] = bq
        got = x.reshape(-1, C).contiguous().cuda()
        qkv, _ = _vit_block_gpu(ctx, got, big, 0, C, T, windows, heads, "f16")
        col = qkv[:, 2 * C + 3].float()
        if finite:
            assert torch.equal(col, torch.full_like(col, float(torch.tensor(target).half()))) and torch.isfinite(got).all()
            assert ctx.status_flags() == 0
        else:
            assert torch.isinf(col).all() and not torch.isfinite(got).all()
            # ... and it is reported: the store that rounded 7e4 to +inf raised the context's status bit
            assert ctx.status_flags() == 1 and ctx.status_flags() == 0          # read and cleared
            bctx = ctx_for("tiny", "bf16")
            gotb = x.reshape(-1, C).contiguous().cuda()
            _vit_block_gpu(bctx, gotb, big, 0, C, T, windows, heads, "bf16")
            assert torch.isfinite(gotb).all() and bctx.status_flags() == 0


def test_outlier_activations_through_a_residual_conv_unit():
    """decoder.rs:35-44 ResidualConvUnit with inputs of 1e4 (maxima near 4.5e4): x + conv2(relu(conv1(relu(x)))) keeps
    x in f32, the 16-bit operands are relu(x) <= 65504 and the first conv's output stays below the f16 maximum by
    construction of the weights here, so the unit is as accurate as at unit scale.  Scaled 30x further the 16-bit
    operand overflows to +inf, the first conv sums infinities of both signs to NaN -- and the ReLU behind it maps
    NaN to 0 (fmaxf, exactly what f32::max does in the reference's relu): the branch silently drops out and the
    unit returns x + bias terms, finite.  So on the conv stages an f16 overflow is NOT loud; the operand type for
    a checkpoint with such magnitudes is ME_DTYPE_BF16.  (The fp32 reference has no such limit: this is a property
    of 16-bit operands, stated here so that it is known, with the magnitudes at which it starts.)"""
    ctx = ctx_for("tiny", "f16")
    B, H, Cc = 1, 24, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cc, H, H, generator=g)
    w1 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc)
    w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc)
    b1, b2 = torch.randn(Cc, generator=g) * 0.02, torch.randn(Cc, generator=g) * 0.02
    for scale, finite in ((1.0, True), (1.0e4, True), (3.0e5, False)):
        xs = x * scale
        x32 = xs.permute(0, 2, 3, 1).reshape(-1, Cc).contiguous().cuda()
        a16 = bordered(torch.relu(xs), "f16")
        t16 = torch.zeros(B, H + 2, H + 2, Cc, dtype=torch.float16, device="cuda")
        out32 = torch.empty(B * H * H, Cc, dtype=torch.float32, device="cuda")
        wa, wb = dev16(pack_conv(w1), "f16"), dev16(pack_conv(w2), "f16")
        b1d, b2d = b1.cuda(), b2.cuda()
        _check(ctx, ctx.lib.me_op_conv2d(ctx.handle, ptr(a16), B, H, H, Cc, ptr(wa), Cc, 3, 1, ptr(b1d), None, None,
                                         None, ptr(t16), 1, 2, 0, -1))
        _check(ctx, ctx.lib.me_op_conv2d(ctx.handle, ptr(t16), B, H, H, Cc, ptr(wb), Cc, 3, 1, ptr(b2d), ptr(x32), None,
                                         ptr(out32), None, 0, 0, 0, -1))
        ctx.synchronize()
        if not finite:
            assert torch.isinf(a16).any() and torch.isfinite(out32).all()
            assert not torch.isinf(t16).any() and float(t16.float().abs().max()) == 0.0      # NaN -> relu -> 0
            continue
        wa64, wb64 = w1.half().double(), w2.half().double()
        ref = xs.double() + F.conv2d(torch.relu(F.conv2d(torch.relu(xs.half().double()), wa64, b1.double(), padding=1)),
                                     wb64, b2.double(), padding=1)
        branch = ref - xs.double()
        got = out32.cpu().double().reshape(B, H, H, Cc).permute(0, 3, 1, 2)
        err = float(((got - xs.double()) - branch).norm() / branch.norm())
        print("RCU branch rel-L2 at scale", scale, err)
        assert torch.isfinite(out32).all() and err < 1.0e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    dict(B=1, H=12, W=16, Cin=128, Cmid=32),                 # one tile of the halo kernel
    dict(B=2, H=48, W=64, Cin=128, Cmid=32, f_norm=True),    # 2 x 16 tiles, per-image f_norm
    dict(B=1, H=384, W=400, Cin=128, Cmid=32, f_norm=True),  # 800 tiles: workgroups walk on (double-buffered halo)
    dict(B=3, H=36, W=48, Cin=128, Cmid=32, clamp=True),
    dict(B=2, H=24, W=128, Cin=128, Cmid=32, f_norm=True),   # 8 tile columns: one per XCD (the strip walk)
    dict(B=3, H=132, W=256, Cin=128, Cmid=32),               # strips of two tile columns, 528 tiles
    dict(B=1, H=20, W=24, Cin=128, Cmid=32),                 # not a multiple of 12 x 16: the implicit-GEMM tile
    dict(B=1, H=24, W=32, Cin=64, Cmid=32),                  # another channel count: the implicit-GEMM tile
])
def test_head_final(dtype, case):
    """mod.rs:83-94,329-362: conv3x3 (128 -> 32) + ReLU + conv1x1 (32 -> 1) + ReLU, / f_norm, clamp in one launch --
    the halo kernel (csrc/head_conv.hip) where the shape is the model's, the implicit-GEMM tile elsewhere, both against
    torch on the same 16-bit operands, and against each other where both apply."""
    ctx = ctx_for("tiny", dtype)
    B, H, W, Cin, Cmid = (case[n] for n in ("B", "H", "W", "Cin", "Cmid"))
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cmid, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cmid, generator=g) * 0.3
    w2 = torch.randn(Cmid, generator=g) / math.sqrt(Cmid)
    b2 = torch.tensor([0.4])
    f_norm = (0.5 + torch.rand(B, generator=g)) if case.get("f_norm") else None
    lo, hi = (0.2, 1.5) if case.get("clamp") else (-float("inf"), float("inf"))
    xb, w16 = bordered(x, dtype), dev16(pack_conv(w), dtype)
    bias_d, w2_d, b2_d = bias.cuda(), w2.cuda(), b2.cuda()
    fn_d = f_norm.cuda() if f_norm is not None else None
    outs = {}
    for cfg in (-1, 2):
        out = torch.full((B * H * W + 16,), -7.0, device="cuda")
        _check(ctx, ctx.lib.me_op_head_final(ctx.handle, ptr(xb), B, H, W, Cin, ptr(w16), Cmid, ptr(bias_d), ptr(w2_d), ptr(b2_d),
                                             ptr(fn_d), lo, hi, ptr(out), cfg))
        ctx.synchronize()
        assert bool((out[B * H * W:] == -7.0).all())
        outs[cfg] = out[:B * H * W].cpu().double().reshape(B, H, W)
    x16 = xb[:, 1:H + 1, 1:W + 1, :].permute(0, 3, 1, 2).double().cpu()
    mid = F.relu(F.conv2d(x16, dev16(w, dtype).double().cpu(), bias.double(), padding=1))
    ref = F.relu((mid * w2.double().view(1, Cmid, 1, 1)).sum(1) + b2.double())
    if f_norm is not None:
        ref = ref / f_norm.double().view(B, 1, 1)
    ref = ref.clamp(lo, hi)
    scale = float(ref.abs().max())
    for cfg, got in outs.items():
        assert float((got - ref).abs().max()) < 2e-5 * scale, (cfg, float((got - ref).abs().max()), scale)
    assert float((outs[-1] - outs[2]).abs().max()) < 2e-5 * scale
