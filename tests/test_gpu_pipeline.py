"""Forward-pass parity (-m gpu): the HIP path, through the C ABI, against the CPU oracle
(oracle/depth_pro_oracle.py, fp32) on the same seeded synthetic checkpoint and images.

Tolerances.  north_star asks for "depth within 1e-3 relative" of the fp32 CPU path.  Operands are
rounded to the MFMA input type (f16: 2^-11, bf16: 2^-8 per element) with f32 accumulation and an f32
residual stream, so the achievable error is a few 2^-11 for f16; the asserted bounds below are what
the stage under test is held to, written as relative L2 against the oracle."""
import os

import numpy as np
import pytest
import torch

import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_images
from oracle import depth_pro_oracle as O
from util import depth_error_report, loaded_ctx, oracle_cfg, rel_l2, weights_for

pytestmark = pytest.mark.gpu

# relative-L2 budget per stage and operand type
TOL = {"f16": 1.0e-3, "bf16": 8.0e-3}


def _img(cfg, batch=1, family="structured"):
    return O.preprocess_u8(synthetic_images(batch, cfg.img_size, family))


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_vit_forward_features_tiny(dtype):
    ctx = loaded_ctx("tiny", dtype)
    cfg, w = ctx.cfg, weights_for("tiny")
    xs = _img(cfg)[:, :, :cfg.window * 3, :cfg.window].reshape(1, 3, 3, cfg.window, cfg.window)
    xs = xs.permute(0, 2, 1, 3, 4).reshape(3, 3, cfg.window, cfg.window).contiguous()
    final, inter = ctx.vit_forward_features(0, xs.numpy(), [1, 3])
    rf, ri = O.vit_forward_features(xs, w, "encoder.patch_encoder.", oracle_cfg(cfg), [1, 3])
    assert rel_l2(final, rf) < TOL[dtype]
    assert rel_l2(inter[0], ri[0]) < TOL[dtype] and rel_l2(inter[1], ri[1]) < TOL[dtype]


def test_vit_missing_block_is_bad_shape():
    ctx = loaded_ctx("tiny", "f16")
    cfg = ctx.cfg
    xs = np.zeros((1, 3, cfg.window, cfg.window), np.float32)
    with pytest.raises(m.MatrixEyesError) as e:      # vit.rs:318-324 panics; here an error code
        ctx.vit_forward_features(0, xs, [cfg.depth])
    assert e.value.code == 2


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_stages_tiny(dtype):
    """encoder / decoder / head / fov, each fed the ORACLE's inputs so errors do not compound"""
    ctx = loaded_ctx("tiny", dtype)
    cfg, w = ctx.cfg, weights_for("tiny")
    ocfg = oracle_cfg(cfg)
    img = _img(cfg)
    inv, fov, parts = O.extract_depth(img, None, w, ocfg, return_parts=True)
    enc = ctx.encoder_forward_encodings(img.numpy())
    for got, ref in zip(enc, parts["encodings"]):
        assert got.shape == tuple(ref.shape)
        assert rel_l2(got, ref) < 1.5 * TOL[dtype]
    feat, low = ctx.decoder_forward([e.numpy() for e in parts["encodings"]])
    assert rel_l2(feat, parts["features"]) < TOL[dtype]
    assert rel_l2(low, parts["lowres"]) < TOL[dtype]
    canon = ctx.head_forward(parts["features"].numpy())
    assert rel_l2(canon, parts["canonical"]) < TOL[dtype]
    fov_gpu = ctx.fov_forward(img.numpy(), parts["lowres"].numpy())
    assert abs(float(fov_gpu[0]) - float(fov[0])) < 0.05 * (8 if dtype == "bf16" else 1)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("f_norm", [1.0, None])
def test_extract_depth_tiny(dtype, f_norm):
    ctx = loaded_ctx("tiny", dtype)
    cfg, w = ctx.cfg, weights_for("tiny")
    img = _img(cfg)
    ref, ref_fov = O.extract_depth(img, f_norm, w, oracle_cfg(cfg))
    got, fov = ctx.extract_depth(img.numpy(), f_norm, want_fov=True)
    rep = depth_error_report(got, ref.numpy())
    print(dtype, f_norm, rep)
    assert rep["rel_l2"] < 2 * TOL[dtype]
    if f_norm is None:
        assert abs(float(fov[0]) - float(ref_fov[0])) < 0.05 * (8 if dtype == "bf16" else 1)
    assert got.min() >= 1e-4 and got.max() <= 1e4          # mod.rs:362


def test_u8_entry_equals_f32_entry():
    ctx = loaded_ctx("tiny", "f16")
    rgb = synthetic_images(1, ctx.cfg.img_size)
    a = ctx.extract_depth(rgb, 1.0)
    b = ctx.extract_depth(O.preprocess_u8(rgb).numpy(), 1.0)
    assert np.array_equal(a, b)      # reconstruction.rs:116-124 reproduced operation for operation
    pre = ctx.preprocess_u8(rgb)
    assert np.array_equal(pre, O.preprocess_u8(rgb).numpy())


def test_batch_equals_loop_of_batch_one():
    """the reference is batch 1; a batch here must equal a loop of batch-1 calls"""
    ctx = loaded_ctx("tiny", "f16")
    rgb = synthetic_images(3, ctx.cfg.img_size)
    f_norm = np.array([0.9, 1.0, 1.3], np.float32)
    batch = ctx.extract_depth(rgb, f_norm)
    for i in range(3):
        one = ctx.extract_depth(rgb[i:i + 1], float(f_norm[i]))
        assert np.array_equal(batch[i], one[0])
    batch_fov, fovs = ctx.extract_depth(rgb, None, want_fov=True)
    one, fov1 = ctx.extract_depth(rgb[1:2], None, want_fov=True)
    assert np.array_equal(batch_fov[1], one[0]) and fovs[1] == fov1[0]


def test_device_pointers_and_determinism():
    ctx = loaded_ctx("tiny", "f16")
    S = ctx.cfg.img_size
    rgb = torch.from_numpy(synthetic_images(1, S)).cuda()
    out1 = torch.empty(1, S, S, dtype=torch.float32, device="cuda")
    out2 = torch.empty_like(out1)
    ctx.extract_depth(rgb, 1.0, out=out1)
    ctx.extract_depth(rgb, 1.0, out=out2)
    ctx.synchronize()
    assert torch.equal(out1, out2)                       # repeated launches are bit-identical
    host = ctx.extract_depth(rgb.cpu().numpy(), 1.0)
    assert np.array_equal(out1.cpu().numpy(), host)


@pytest.mark.parametrize("dtype", ["f16", "fp8"])
def test_captured_graph_replays_the_eager_step(dtype):
    """A call with device pointers only is eager, then captured, then one hipGraphLaunch (matrix_eyes_hip.h
    me_graph_launch_count): every replay is bit-identical to the eager result, with the FOV head and with a
    device-resident f_norm; another output pointer, a host pointer or a progress callback run eagerly."""
    from matrix_eyes_amd.synthetic import synthetic_checkpoint
    cfg = m.ModelConfig(grid=8, embed_dim=256, num_heads=4, depth=4, tap_blocks=(1, 2), enc_dims=(64, 128, 128, 128),
                        dec_dim=256, head_dims=(32, 1))
    ctx = m.Context(0, dtype, cfg)
    ctx.load_state_dict(synthetic_checkpoint(cfg))
    ctx.set_graph(True)
    S = cfg.img_size
    structured = torch.from_numpy(synthetic_images(2, S)).cuda()
    noise = torch.from_numpy(synthetic_images(2, S, "noise")).cuda()
    rgb = structured.clone()
    out = torch.empty(2, S, S, dtype=torch.float32, device="cuda")
    for f_norm in (None, torch.tensor([0.8, 1.3], device="cuda")):
        rgb.copy_(structured)
        n0 = ctx.graph_launch_count
        ctx.extract_depth(rgb, f_norm, out=out)              # first sight of this call: eager
        ctx.synchronize()
        eager = out.clone()
        assert ctx.graph_launch_count == n0
        for i in range(3):                                   # captured and launched, then replayed
            out.zero_()
            ctx.extract_depth(rgb, f_norm, out=out)
            ctx.synchronize()
            assert ctx.graph_launch_count == n0 + i + 1
            assert torch.equal(out, eager)
        rgb.copy_(noise)                                     # same pointers, new pixels: the graph reads them
        ctx.extract_depth(rgb, f_norm, out=out)
        ctx.synchronize()
        assert ctx.graph_launch_count == n0 + 4 and not torch.equal(out, eager)
        other = torch.empty_like(out)
        ctx.extract_depth(rgb, f_norm, out=other)            # another output pointer: eager again
        ctx.synchronize()
        assert ctx.graph_launch_count == n0 + 4 and torch.equal(other, out)
        # a key change right behind a replay, nothing synchronised in between: the exec is destroyed only once
        # its queued launch has drained (me_ctx::drop_graph), so the replay's result is intact
        ctx.extract_depth(rgb, f_norm, out=out)
        ctx.extract_depth(rgb, f_norm, out=out)
        n1 = ctx.graph_launch_count
        out.zero_()
        ctx.extract_depth(rgb, f_norm, out=out)              # replay, in flight ...
        third = torch.empty_like(out)
        ctx.extract_depth(rgb, f_norm, out=third)            # ... when the new pointer drops its exec
        ctx.synchronize()
        assert ctx.graph_launch_count == n1 + 1 and torch.equal(out, other) and torch.equal(third, other)
        host = ctx.extract_depth(noise.cpu().numpy(), None if f_norm is None else f_norm.cpu().numpy())
        assert np.array_equal(host, out.cpu().numpy())       # host pointers: eager, same bits
    n0 = ctx.graph_launch_count
    seen = []
    ctx.set_progress(lambda pos, msg: seen.append(pos))
    for _ in range(3):
        ctx.extract_depth(rgb, None, out=other)
    ctx.set_progress(None)
    ctx.synchronize()
    assert ctx.graph_launch_count == n0 and seen
    ctx.close()


def test_state_dict_with_extra_keys_loads_like_the_reference(tmp_path):
    """mod.rs:236-243 checks `result.errors` and `result.missing` only: keys the model does not use (wrapper
    siblings, EMA copies, integer buffers, bf16 tensors) are normal.  Both loaders skip and list them; the depth
    equals the plain checkpoint's."""
    cfg = m.ModelConfig.tiny()
    w = dict(weights_for("tiny"))
    extra = {"ema.decay": torch.tensor(0.999, dtype=torch.float64), "step": torch.tensor([7], dtype=torch.int64),
             "aux.proj.weight": torch.randn(4, 4, dtype=torch.bfloat16)}
    rgb = synthetic_images(1, cfg.img_size)
    want = loaded_ctx("tiny", "f16").extract_depth(rgb, 1.0)
    ctx = m.Context(0, "f16", cfg)
    ctx.load_state_dict({**extra, **w})
    assert sorted(ctx.unused_keys) == sorted(extra) and np.array_equal(ctx.extract_depth(rgb, 1.0), want)
    ctx.close()
    # the library's own reader (me_load_checkpoint_pt), a {"state_dict": ...} wrapper with siblings
    ck = {k: torch.as_tensor(v) for k, v in w.items()}
    torch.save({"state_dict": {**ck, **extra}, "epoch": 3}, tmp_path / "wrapped.pt")
    ctx = m.Context(0, "f16", cfg)
    ctx.load_checkpoint_pt(str(tmp_path / "wrapped.pt"))
    assert sorted(ctx.unused_keys) == sorted(extra) and np.array_equal(ctx.extract_depth(rgb, 1.0), want)
    ctx.close()
    # bf16 / f64 storage of a tensor the model does use is converted, a wrong shape stays an error
    ck2 = dict(ck)
    name = "head.4.bias"
    ck2[name] = ck[name].to(torch.float64)
    torch.save(ck2, tmp_path / "f64.pt")
    ctx = m.Context(0, "f16", cfg)
    ctx.load_checkpoint_pt(str(tmp_path / "f64.pt"))
    assert np.array_equal(ctx.extract_depth(rgb, 1.0), want)
    ctx.close()
    ck2[name] = torch.zeros(2)
    torch.save(ck2, tmp_path / "shape.pt")
    ctx = m.Context(0, "f16", cfg)
    with pytest.raises(m.MatrixEyesError) as e:
        ctx.load_checkpoint_pt(str(tmp_path / "shape.pt"))
    assert e.value.code == 4
    with pytest.raises(m.MatrixEyesError) as e:
        ctx.load_checkpoint_pt(str(tmp_path / "absent.pt"))
    assert e.value.code == 7          # LoaderError::Pytorch
    ctx.close()


def test_padding_rows_stay_finite_across_batch_sizes():
    """The three ViTs share one row space with each segment padded to 256 rows; the padding rows take part in
    every row-wise kernel.  They are re-zeroed per call: a batch of 2, then of 1, then of 2 reproduces itself."""
    ctx = loaded_ctx("tiny", "f16")
    rgb = synthetic_images(2, ctx.cfg.img_size, seed=5)
    first = [ctx.extract_depth(rgb, None, want_fov=True) for _ in range(1)][0]
    for _ in range(3):
        ctx.extract_depth(rgb[:1], None)
        again = ctx.extract_depth(rgb, None, want_fov=True)
        assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])


def test_bcast_weights_single_rank_runs_rccl():
    """me_bcast_weights with one rank still creates a communicator, broadcasts the arena in place and destroys the
    communicator: every RCCL call of the native start-up path executes on a one-GPU box."""
    src = loaded_ctx("tiny", "f16")
    rgb = synthetic_images(1, src.cfg.img_size)
    want = src.extract_depth(rgb, 1.0)
    before = src.weight_arena_tensor().clone()
    src.bcast_weights(src.rccl_unique_id(), 0, 1)
    assert torch.equal(src.weight_arena_tensor(), before) and np.array_equal(src.extract_depth(rgb, 1.0), want)
    fresh = m.Context(0, "f16", src.cfg)
    with pytest.raises(m.MatrixEyesError) as e:          # rank 0 must hold finalized weights
        fresh.bcast_weights(fresh.rccl_unique_id(), 0, 1)
    assert e.value.code == 8
    fresh.close()


def test_missing_and_unexpected_weights():
    cfg = m.ModelConfig.tiny()
    ctx = m.Context(0, "f16", cfg)
    w = dict(weights_for("tiny"))
    some = next(iter(w))
    with pytest.raises(m.MatrixEyesError) as e:          # mod.rs:238-240
        ctx.load_weight("encoder.nonexistent.weight", w[some])
    assert e.value.code == 4
    with pytest.raises(m.MatrixEyesError) as e:
        ctx.load_weight(some, w[some].reshape(-1))
    assert e.value.code == 4
    del w["head.4.bias"]
    with pytest.raises(m.MatrixEyesError) as e:          # mod.rs:241-243
        ctx.load_state_dict(w)
    assert e.value.code == 3 and "head.4.bias" in e.value.message
    with pytest.raises(m.MatrixEyesError) as e:
        ctx.extract_depth(np.zeros((1, 3, cfg.img_size, cfg.img_size), np.float32), 1.0)
    assert e.value.code == 8
    assert [n for n, _ in ctx.expected_weights()] == [n for n, _, _ in m.expected_weights(cfg)]
    assert [s for _, s in ctx.expected_weights()] == [tuple(s) for _, s, _ in m.expected_weights(cfg)]
    ctx.close()


def test_f16_overflow_is_reported_not_swallowed():
    """VERDICT r2 item 9.  The reference computes in f32 (decoder.rs:35-44); an f16 operand past 65504 is +-inf, and
    behind a conv + ReLU the branch silently drops out (test_outlier_activations_through_a_residual_conv_unit).  A
    checkpoint whose encoder projection is scaled by 3e5 drives the un-diluted conv chain past the range: the
    call that returns to the host fails with ME_ERR_OVERFLOW, the asynchronous (device-output) call raises
    ME_STATUS_OVERFLOW_16BIT for me_status_flags, the bf16 context computes the same model finitely with no flag,
    and the untouched checkpoint raises nothing."""
    cfg = m.ModelConfig.tiny()
    w = dict(weights_for("tiny"))
    rgb = synthetic_images(1, cfg.img_size)
    good = loaded_ctx("tiny", "f16")
    good.status_flags()
    good.extract_depth(rgb, None)
    assert good.status_flags() == 0
    w["encoder.upsample_latent0.0.weight"] = (torch.as_tensor(w["encoder.upsample_latent0.0.weight"]).float() * 3.0e5)
    ctx = m.Context(0, "f16", cfg)
    ctx.load_state_dict(w)
    with pytest.raises(m.MatrixEyesError) as e:
        ctx.extract_depth(rgb, 1.0)
    assert e.value.code == 10 and "65504" in e.value.message
    out = torch.empty(1, cfg.img_size, cfg.img_size, dtype=torch.float32, device="cuda")
    ctx.extract_depth(torch.from_numpy(rgb).cuda(), 1.0, out=out)      # asynchronous: no error here ...
    assert ctx.status_flags() == 1 and ctx.status_flags() == 0          # ... the flag says so, once
    # the flag is sticky over asynchronous calls: an overflowing step followed by clean steps on ANOTHER
    # context state would be lost if every step cleared it -- here three device-result calls, one poll
    for _ in range(3):
        ctx.extract_depth(torch.from_numpy(rgb).cuda(), 1.0, out=out)
    good_out = torch.empty_like(out)
    assert ctx.status_flags() == 1 and ctx.status_flags() == 0
    good.extract_depth(torch.from_numpy(rgb).cuda(), 1.0, out=good_out)
    good.extract_depth(torch.from_numpy(rgb).cuda(), 1.0, out=good_out)
    assert good.status_flags() == 0
    ctx.close()
    bctx = m.Context(0, "bf16", cfg)
    bctx.load_state_dict(w)
    d = bctx.extract_depth(rgb, 1.0)
    assert np.isfinite(d).all() and bctx.status_flags() == 0
    bctx.close()


def test_weight_arena_handover():
    """The multi-GPU start-up path on one GPU: a second context receives only the packed arena bytes
    (what the broadcast delivers), adopts them and must produce bit-identical depth."""
    src = loaded_ctx("tiny", "f16")
    dst = m.Context(0, "f16", src.cfg)
    a, b = src.weight_arena_tensor(), dst.weight_arena_tensor()
    assert a.dtype == torch.uint8 and a.numel() == src.weight_arena_bytes() == b.numel()
    with pytest.raises(m.MatrixEyesError):               # nothing loaded yet
        dst.extract_depth(synthetic_images(1, src.cfg.img_size), 1.0)
    b.copy_(a)
    torch.cuda.synchronize()
    dst.adopt_weights()
    rgb = synthetic_images(1, src.cfg.img_size)
    d0, f0 = src.extract_depth(rgb, None, want_fov=True)
    d1, f1 = dst.extract_depth(rgb, None, want_fov=True)
    assert np.array_equal(d0, d1) and f0[0] == f1[0]
    dst.close()


def test_one_reloaded_fusion_factor_recomposes_the_fused_weights():
    """a context whose arena came by me_weights_adopt holds no host copies of the deconv / out_conv
    factors that the composed fusion weights (split_operands & 2) are built from.  Reloading ONE of them and
    finalizing must recompose with the other factor read back from the arena -- the depth equals that of a context
    loaded with the modified checkpoint from scratch, not the stale composition."""
    src = loaded_ctx("tiny", "f16")
    assert src.cfg.split_operands & 2
    w = dict(weights_for("tiny"))
    name = "decoder.fusions.2.out_conv.weight"
    w[name] = (torch.as_tensor(w[name]).float() * 1.5).half()
    rgb = synthetic_images(1, src.cfg.img_size)
    fresh = m.Context(0, "f16", src.cfg)
    fresh.load_state_dict(w)
    want = fresh.extract_depth(rgb, 1.0)
    fresh.close()
    dst = m.Context(0, "f16", src.cfg)
    assert dst.weight_arena_layout() == src.weight_arena_layout() != 0
    dst.weight_arena_tensor().copy_(src.weight_arena_tensor())
    torch.cuda.synchronize()
    dst.adopt_weights()
    before = dst.extract_depth(rgb, 1.0)
    dst.load_weight(name, w[name])
    dst._check(dst.lib.me_weights_finalize(dst.handle))
    got = dst.extract_depth(rgb, 1.0)
    assert not np.array_equal(before, got) and np.array_equal(got, want)
    other = m.Context(0, "f16", m.ModelConfig(**{**src.cfg.__dict__, "split_operands": 0}))
    assert other.weight_arena_layout() != src.weight_arena_layout()      # what me_bcast_weights compares
    other.close()
    dst.close()


def test_progress_callback():
    ctx = loaded_ctx("tiny", "f16")
    seen = []
    ctx.set_progress(lambda pos, msg: seen.append((pos, msg)))
    ctx.extract_depth(synthetic_images(1, ctx.cfg.img_size), None)
    ctx.set_progress(None)
    assert seen and seen[-1][0] == 1.0 and any(msg == "encoding patches" for _, msg in seen)
    # mod.rs:265-293: one bar for the whole call -- the encoder ends at 64 %, the decoder at 79.68 % when
    # the FOV head runs; the positions only move forward
    pos = [p for p, msg in seen]
    assert all(b >= a - 1e-6 for a, b in zip(pos, pos[1:])) and 0.0 <= min(pos) and max(pos) == 1.0
    by_msg = {msg: p for p, msg in seen}
    assert abs(by_msg["fusing lowres"] - (0.032 + 0.95 * (0.64 - 0.032))) < 1e-5
    assert abs(by_msg["forwarding head"] - 0.99) < 1e-5


@pytest.fixture(scope="module")
def full_oracle():
    """One 1536x1536 image through the fp32 oracle (about 20 TFLOP of CPU work), shared by the full-size tests"""
    cfg, w = m.ModelConfig(), weights_for("full")
    img = _img(cfg)
    ref, ref_fov = O.extract_depth(img, None, w, oracle_cfg(cfg))
    return img, ref.numpy(), float(ref_fov[0])


# (image family, image seed, checkpoint seed): SURVEY 8d names `structured` / 4321 for parity and `noise` / 1234 for
# throughput; the third pair changes BOTH the image and the 952 M weights
FULL_PAIRS = [("structured", 4321, 2024), ("noise", 1234, 2024), ("structured", 77, 7)]
# Each pair costs a 70 - 95 s run of the CPU oracle and the driver's `-m gpu` tier has a time limit: the pair that shares pair 0's
# weights runs when ME_TEST_ALL_PAIRS=1 (the round's evidence runs set it; DESIGN 5.1 has its numbers)
_ALL_PAIRS = os.environ.get("ME_TEST_ALL_PAIRS", "0") not in ("", "0")
_PAIR_PARAMS = [pytest.param(*pr, marks=pytest.mark.skipif(i == 1 and not _ALL_PAIRS, reason="ME_TEST_ALL_PAIRS=1 runs the third oracle pass"))
                for i, pr in enumerate(FULL_PAIRS)]


@pytest.mark.parametrize("family,img_seed,ckpt_seed", _PAIR_PARAMS)
def test_extract_depth_full_size_pairs(family, img_seed, ckpt_seed, full_oracle):
    """north_star: depth within 1e-3 relative of the CPU reference -- held on three (image, checkpoint) pairs, not one:
    relative L2 < 1e-3 on each, and the per-pixel distribution bounded too (median, 99th percentile and maximum of
    |d - ref| / max(|ref|, 0.05 median(ref)); the floor keeps the pixels that the closing ReLU zeroes -- clamped to
    1e-4 on both sides -- from dividing by ~0).  Measured (f16, split_operands 3), round 3:
        structured/4321, ckpt 2024: rel-L2 7.11e-4, median 1.40e-4, p99 2.29e-2, max 0.45, FOV 54.90174 vs 54.90181
        noise/1234,      ckpt 2024: rel-L2 6.86e-4, median 0.94e-4, p99 2.21e-2, max 0.51, FOV 54.75515 vs 54.75537
        structured/77,   ckpt 7:    rel-L2 5.15e-4, median 3.31e-4, p99 1.19e-2, max 0.074, FOV 54.86040 vs 54.86039
    The per-pixel tail is NOT a ReLU-crossing effect (round 3's reading): the error is additive, the same absolute
    distribution at every reference value, so the relative figure is large exactly where the reference is small against
    the map's rms -- see the assertions at the end."""
    from matrix_eyes_amd.synthetic import synthetic_checkpoint
    cfg = m.ModelConfig()
    if (family, img_seed, ckpt_seed) == FULL_PAIRS[0]:
        img, ref, ref_fov = full_oracle
        ctx, own = loaded_ctx("full", "f16"), False
        assert cfg.img_size == m.IMG_SIZE == 1536 and ctx.cfg.split_operands == 3 and ctx.weight_arena_bytes() > 1.9e9
    else:
        w = weights_for("full") if ckpt_seed == 2024 else synthetic_checkpoint(cfg, seed=ckpt_seed)
        img = O.preprocess_u8(synthetic_images(1, cfg.img_size, family, seed=img_seed))
        r, rf = O.extract_depth(img, None, w, oracle_cfg(cfg))
        ref, ref_fov = r.numpy(), float(rf[0])
        if ckpt_seed == 2024:
            ctx, own = loaded_ctx("full", "f16"), False
        else:
            ctx, own = m.Context(0, "f16", cfg), True
            ctx.load_state_dict(w)
        del w
    got, fov = ctx.extract_depth(img.numpy(), None, want_fov=True)
    if own:
        ctx.close()
    rep = depth_error_report(got, ref)
    print("full-size f16 pair", family, img_seed, ckpt_seed, rep, float(fov[0]), ref_fov)
    assert rep["rel_l2"] < 1.0e-3
    assert rep["median"] < 5.0e-4
    assert rep["p99"] < 4.0e-2
    assert rep["max"] < 1.0
    assert abs(float(fov[0]) - ref_fov) < 0.01
    # What the per-pixel tail is (tools/tail_probe.py, profiles/r04_tail_probe.json): the error is
    # ADDITIVE -- |d - ref| has the same distribution in every band of the reference value (p99 0.011 median(ref) from the
    # darkest band to the brightest) -- a noise floor that the 2^-11 operand roundings upstream of the head leave in
    # proportion to the map's own scale.  So it is bounded against rms(ref) for EVERY pixel, and a relative error above
    # 1e-2 can only occur where the reference value is small against that scale.  Measured (three pairs): max 5.8e-3 /
    # 5.3e-3 / 3.2e-3 of the rms, p99 2.3e-3 / 2.3e-3 / 1.5e-3, median 1.5e-4 / 1.0e-4 / 2.8e-4; the brightest tail pixel
    # at 0.45 / 0.42 / 0.24 rms; worst relative error among the pixels at or above the rms 6.1e-3 / 9.0e-3 / 1.2e-3.
    assert rep["abs_over_rms_max"] < 8.0e-3 and rep["abs_over_rms_p99"] < 3.2e-3 and rep["abs_over_rms_median"] < 4.0e-4
    assert rep["tail_ref_over_rms_max"] < 0.8                # = abs_over_rms_max / 1e-2: no tail pixel above 0.8 rms
    assert rep["bright_rel_max"] < 1.0e-2 and rep["bright_fraction"] > 0.02


# Burn 0.21's LayerNorm eps and bilinear convention are ASSUMED (1e-5, align_corners = true); the other
# choice of each is a parameter of both the oracle and the HIP path, and both choices are held to the oracle here
@pytest.mark.parametrize("align_corners", [True, False])
@pytest.mark.parametrize("ln_eps", [1e-5, 1e-6])
def test_assumed_semantics_branches_tiny(align_corners, ln_eps):
    base = m.ModelConfig.tiny()
    cfg = m.ModelConfig(**{**base.__dict__, "align_corners": align_corners, "ln_eps": ln_eps})
    w = weights_for("tiny")
    ctx = m.Context(0, "f16", cfg)
    ctx.load_state_dict(w)
    ocfg = oracle_cfg(cfg)
    assert ocfg.align_corners == align_corners and ocfg.ln_eps == ln_eps
    img = _img(cfg, family="noise")          # high-frequency content: the two bilinear conventions differ visibly on it
    inv, fov, parts = O.extract_depth(img, None, w, ocfg, return_parts=True)
    enc = ctx.encoder_forward_encodings(img.numpy())
    for got, ref in zip(enc, parts["encodings"]):
        assert rel_l2(got, ref) < 1.5 * TOL["f16"]
    got, gfov = ctx.extract_depth(img.numpy(), None, want_fov=True)
    rep = depth_error_report(got, inv.numpy())
    print("branches", align_corners, ln_eps, rep)
    assert rep["rel_l2"] < 2 * TOL["f16"] and abs(float(gfov[0]) - float(fov[0])) < 0.05
    # the test means something only if the branch changes the answer by more than the tolerance it is held to
    other = O.extract_depth(img, None, w, oracle_cfg(m.ModelConfig(**{**cfg.__dict__, "align_corners": not align_corners})))[0]
    assert rel_l2(other, inv) > 10 * rep["rel_l2"]
    other = O.extract_depth(img, None, w, oracle_cfg(m.ModelConfig(**{**cfg.__dict__, "ln_eps": 1e-6 if ln_eps == 1e-5 else 1e-5})))[0]
    print("eps branch moves the depth by", rel_l2(other, inv))
    ctx.close()


def test_split_operand_stages_buy_the_margin(full_oracle):
    """me_model_config.split_operands: the stages of the un-diluted conv chain carried as hi + lo operands.
    Without them the same path sits on the 1e-3 bound (1.01e-3 measured); with every stage split it reaches
    5.2e-4.  Each context is created, measured and destroyed (1.9 GB of weights each)."""
    img, ref, _ = full_oracle
    errs = {}
    for mask in (0, 15):
        cfg = m.ModelConfig(split_operands=mask)
        ctx = m.Context(0, "f16", cfg)
        ctx.load_state_dict(weights_for("full"))
        errs[mask] = depth_error_report(ctx.extract_depth(img.numpy(), None), ref)["rel_l2"]
        ctx.close()
    print("split_operands 0 / 15: rel_l2", errs)
    assert errs[15] < 6.5e-4 < errs[0] < 1.2e-3


def test_extract_depth_full_size_bf16(full_oracle):
    """BASELINE configs[1] names bf16: the same image with bf16 MFMA operands (weights rounded from the fp16
    checkpoint to bf16, 8 significand bits).  Reported, not held to 1e-3: bf16 rounds every operand 8x coarser
    than f16 at the same MFMA rate, which is why f16 is the default.  Measured: 1.06e-2 relative L2 (median per-pixel
    2.1e-3); tolerance 1.5e-2."""
    ctx = loaded_ctx("full", "bf16")
    img, ref, ref_fov = full_oracle
    got, fov = ctx.extract_depth(img.numpy(), None, want_fov=True)
    rep = depth_error_report(got, ref)
    print("full-size bf16", rep, float(fov[0]), ref_fov)
    assert rep["rel_l2"] < 1.5e-2
    assert abs(float(fov[0]) - ref_fov) < 0.5


def test_extract_depth_full_size_fp8(full_oracle):
    """BASELINE configs[3]: the full-size model with the qkv / fc1 / fc2 linears of the three ViTs on MX
    block-scaled fp8 (e4m3 elements, one e8m0 scale per 32 K elements; 2x the 16-bit MFMA rate), everything else
    f16.  fp8 operands carry 3 significand bits, so this configuration is NOT held to north_star's 1e-3: its depth
    error against the fp32 oracle is reported here (and in DESIGN.md).  Each linear on fp8 costs ~4e-2 of relative L2
    on its own and they add in quadrature (profiles/r05_fp8_mask_budget.txt: 6.9e-2 for the default three, 7.8e-2 for
    all four) -- each e4m3 product carries ~5 % of noise, which the residual stream (LayerScale 0.05 - 0.2) dilutes
    over 24 blocks; random weights have no structure that would absorb it."""
    ctx = loaded_ctx("full", "fp8")
    img, ref, ref_fov = full_oracle
    got, fov = ctx.extract_depth(img.numpy(), None, want_fov=True)
    rep = depth_error_report(got, ref)
    print("full-size fp8", rep, float(fov[0]), ref_fov)
    # measured 6.9e-2 with the default fp8_linears = 13 (qkv + fc1 + fc2; 7.8e-2 with proj on fp8 too,
    # profiles/r05_fp8_mask_budget.txt): the bound is 1.3x that
    assert np.isfinite(got).all() and rep["rel_l2"] < 9.0e-2
    assert abs(float(fov[0]) - ref_fov) < 2.0


def test_full_size_fp8_batch_of_eight():
    """BASELINE configs[3] at its per-GPU batch: 8 images per step on the fp8 context.  Images 1 and 6 of the batch
    are bit for bit what a batch of one produces (other tile rounds, 280 + 8 + 8 windows), all depths finite and inside
    the clamp of mod.rs:362."""
    ctx = loaded_ctx("full", "fp8")
    rgb = synthetic_images(8, ctx.cfg.img_size, "structured", seed=123)
    batch, fovs = ctx.extract_depth(rgb, None, want_fov=True)
    assert batch.shape == (8, 1536, 1536) and np.isfinite(batch).all() and np.isfinite(fovs).all()
    assert batch.min() >= 1e-4 and batch.max() <= 1e4
    for i in (1, 6):
        one, fov1 = ctx.extract_depth(rgb[i:i + 1], None, want_fov=True)
        assert np.array_equal(batch[i], one[0]) and fovs[i] == fov1[0]
    assert len({batch[i].tobytes() for i in range(8)}) == 8


def test_full_size_batch_of_eight():
    """BASELINE configs[2]: 8 images per GPU per step.  Images 2 and 7 of the batch are bit for bit what a batch
    of one produces (280 + 8 + 8 windows in three row segments, other tile rounds than at batch 1), every depth
    is finite and inside the clamp of mod.rs:362."""
    ctx = loaded_ctx("full", "f16")
    rgb = synthetic_images(8, ctx.cfg.img_size, "structured", seed=99)
    batch, fovs = ctx.extract_depth(rgb, None, want_fov=True)
    assert batch.shape == (8, 1536, 1536) and np.isfinite(batch).all() and np.isfinite(fovs).all()
    assert batch.min() >= 1e-4 and batch.max() <= 1e4
    for i in (2, 7):
        one, fov1 = ctx.extract_depth(rgb[i:i + 1], None, want_fov=True)
        assert np.array_equal(batch[i], one[0]) and fovs[i] == fov1[0]
    assert len({batch[i].tobytes() for i in range(8)}) == 8


def test_full_size_batch_equals_loop_of_batch_one():
    """BASELINE configs[2] gives each GPU several images per step: at the full size (70 + 2 + 2 windows in three
    row segments, GEMM tiles in other rounds than at batch 1) image i of a batch is bit for bit what a batch of
    one produces, FOV head included, and a repeated call reproduces itself."""
    ctx = loaded_ctx("full", "f16")
    rgb = synthetic_images(2, ctx.cfg.img_size, "structured", seed=77)
    both, fovs = ctx.extract_depth(rgb, None, want_fov=True)
    again, fovs2 = ctx.extract_depth(rgb, None, want_fov=True)
    assert np.array_equal(both, again) and np.array_equal(fovs, fovs2)
    for i in range(2):
        one, fov1 = ctx.extract_depth(rgb[i:i + 1], None, want_fov=True)
        assert np.array_equal(both[i], one[0]) and fovs[i] == fov1[0]
    assert not np.array_equal(both[0], both[1]) and np.isfinite(both).all()


_TAIL_CHILD = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
cfg = m.ModelConfig()
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(synthetic_checkpoint(cfg))
d, fov = ctx.extract_depth(synthetic_images(1, cfg.img_size, "structured", seed=5), None, want_fov=True)
np.save(sys.argv[2], d)
"""


def test_tile_choices_change_no_bit(tmp_path):
    """The three ways proj / fc2 / fc1 of one image can run -- ONE round of 352-row tiles laid out per row segment
    (pipeline.hip tall_tile_wins, the default), whole rounds of 256x256 + a second launch of 96x256 tiles over the
    remaining rows (launch_with_short_tail; ME_GEMM_TALL=0), single 256x256 launches (and ME_GEMM_TAIL96=0) -- keep
    the same K order per output element: the full-size depth is bit for bit the same, with the LayerNorm launched on
    its own everywhere (ME_LN_FUSE=0).  LayerNorm inside the tall tile's residual epilogue (ME_LN_FUSE=1) sums a row's
    statistics in another order (1e-7 relative): a handful of 16-bit roundings fall the other way in the first block,
    every rounding downstream of them is drawn again, and the two depth maps end up two independent realisations of the
    same 16-bit rounding noise -- 8.9e-4 apart, each 7.1e-4 from the oracle.  Bounded by sqrt(2) x the 1e-3 each is held
    to."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # (each child initialises the 952 M-parameter synthetic checkpoint: 16 s; the single-launch form runs when ME_TEST_ALL_PAIRS=1)
    forms = [("tall", {"ME_LN_FUSE": "0"}), ("split", {"ME_GEMM_TALL": "0"}),
             ("single", {"ME_GEMM_TALL": "0", "ME_GEMM_TAIL96": "0"}), ("fused", {"ME_LN_FUSE": "1"})]
    if not _ALL_PAIRS:
        forms = [f for f in forms if f[0] != "single"]
    for name, extra in forms:
        path = str(tmp_path / (name + ".npy"))
        r = subprocess.run([sys.executable, "-c", _TAIL_CHILD, root, path], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    assert np.isfinite(outs[0]).all() and all(np.array_equal(outs[0], o) for o in outs[1:-1])
    fused_err = rel_l2(outs[-1], outs[0])
    print("LayerNorm in the residual epilogue against the stand-alone launches: rel-L2", fused_err)
    assert np.isfinite(outs[-1]).all() and 0 < fused_err < 1.5e-3


_HEAD_CHILD = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
cfg = m.ModelConfig.tiny()
w = dict(synthetic_checkpoint(cfg))
# a ConvTranspose bias large enough that a wrong border term would show: the composed form takes the share of every 3x3 tap
# that falls into the zero padding of the full-resolution map out of its bias
import torch
w["head.1.bias"] = torch.as_tensor(w["head.1.bias"]).float() * 0 + torch.linspace(-0.5, 0.5, cfg.dec_dim // 2)
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(w)
rgb = synthetic_images(2, cfg.img_size, "structured", seed=5)
d = ctx.extract_depth(rgb, 1.0)
canon = ctx.extract_depth(rgb, 1.0)       # repeatable
assert np.array_equal(d, canon)
np.save(sys.argv[2], d)
"""


def test_composed_head_equals_the_three_layers(tmp_path):
    """VERDICT r4 item 4b.  head[1] (ConvTranspose 128 -> 128) and head[2] (conv3x3 128 -> 32) composed at load time into one
    3x3 convolution with 4 phases x 32 channels on the half-resolution map (weights.hip compose_head, EPI_HEAD_COMPOSED) against
    the three launches it replaces (ME_HEAD_COMPOSED=0), tiny model, a batch of two, with a deliberately large ConvTranspose
    bias: the maps agree to the 16-bit rounding the composed form SKIPS (the ConvTranspose output is no longer rounded to an
    operand), on the one-pixel frame -- where taps of the 3x3 convolution fall into its zero padding and the composed bias
    changes -- as well as inside.  (test_extract_depth_tiny / the full-size pairs hold the composed form to the fp32 oracle.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, extra in (("composed", {}), ("layers", {"ME_HEAD_COMPOSED": "0"})):
        path = str(tmp_path / (name + ".npy"))
        r = subprocess.run([sys.executable, "-c", _HEAD_CHILD, root, path], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(path)
    a, b = outs["composed"], outs["layers"]
    assert np.isfinite(a).all() and a.shape == b.shape and not np.array_equal(a, b)
    assert rel_l2(a, b) < 6e-4
    frame = np.ones(a.shape[1:], bool)
    frame[1:-1, 1:-1] = False
    scale = np.sqrt((b ** 2).mean())
    inner_err = np.abs(a - b)[:, ~frame].max() / scale
    frame_err = np.abs(a - b)[:, frame].max() / scale
    print("composed head vs three layers: rel-L2", rel_l2(a, b), "max |d| / rms inside", inner_err, "on the frame", frame_err)
    assert frame_err < 3 * inner_err + 1e-3      # a missing border term would be of the order of the bias: 10 - 100 x this


_FEAT_CHILD = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
cfg = m.ModelConfig.tiny()
w = dict(synthetic_checkpoint(cfg))
# an out_conv bias large enough that a wrong border term would show: head[0] pads out_conv's OUTPUT with zeros, so the composed
# convolution takes the share of every tap that falls into that padding out of its bias
import torch
w["decoder.fusions.0.out_conv.bias"] = torch.linspace(-0.5, 0.5, cfg.dec_dim)
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(w)
rgb = synthetic_images(2, cfg.img_size, "structured", seed=5)
d = ctx.extract_depth(rgb, 1.0)
canon = ctx.extract_depth(rgb, 1.0)       # repeatable
assert np.array_equal(d, canon)
np.save(sys.argv[2], d)
"""


def test_composed_features_equal_the_two_layers(tmp_path):
    """VERDICT r4 item 4a, by composition instead of a second GEMM in the epilogue.  The last fusion block's out_conv (1x1) and
    head[0] (conv3x3) composed at load time into one 3x3 convolution of out_conv's input (weights.hip compose_features; the residual
    unit's last convolution writes that input as the zero-bordered 16-bit operand, GemmParams::tap_bias corrects the bias on the
    frame) against the two launches (ME_FEAT_COMPOSED=0): tiny model, a batch of two, a deliberately large out_conv bias.  The
    maps agree to the operand roundings that differ (the feature map is no longer rounded to 16 bits, its pre-image and the composed
    weights are), on the one-pixel frame of the half-resolution map -- two pixels of the depth map -- as well as inside."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, extra in (("composed", {}), ("layers", {"ME_FEAT_COMPOSED": "0"})):
        path = str(tmp_path / (name + ".npy"))
        r = subprocess.run([sys.executable, "-c", _FEAT_CHILD, root, path], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(path)
    a, b = outs["composed"], outs["layers"]
    assert np.isfinite(a).all() and a.shape == b.shape and not np.array_equal(a, b)
    assert rel_l2(a, b) < 6e-4
    frame = np.ones(a.shape[1:], bool)
    frame[4:-4, 4:-4] = False     # the half-resolution frame pixel and its 3x3 neighbourhood in the head's second convolution
    scale = np.sqrt((b ** 2).mean())
    inner_err = np.abs(a - b)[:, ~frame].max() / scale
    frame_err = np.abs(a - b)[:, frame].max() / scale
    print("composed features vs two layers: rel-L2", rel_l2(a, b), "max |d| / rms inside", inner_err, "on the frame", frame_err)
    assert frame_err < 3 * inner_err + 1e-3      # a missing border term would be of the order of the bias: 10 - 100 x this


_SPLITK_CHILD = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
cfg = m.ModelConfig.tiny()
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(synthetic_checkpoint(cfg))
rgb = synthetic_images(3, cfg.img_size, "structured", seed=11)
d, fov = ctx.extract_depth(rgb, None, want_fov=True)
again, fov2 = ctx.extract_depth(rgb, None, want_fov=True)     # the last arriver differs from run to run: the sum must not
assert np.array_equal(d, again) and np.array_equal(fov, fov2)
one, fov1 = ctx.extract_depth(rgb[1:2], None, want_fov=True)  # a batch is a loop of batch-one calls, bit for bit
assert np.array_equal(one[0], d[1]) and fov1[0] == fov[1]
np.save(sys.argv[2], d)
"""


def test_split_k_tail_launches_are_deterministic(tmp_path):
    """The deep, small launches of the decoder / upsample tail as tile x K-range work items (gemm_core.h gemm_kernel<..., SPLITK>,
    pipeline.hip maybe_split_k): every work item writes its f32 partial, the LAST arriver of a tile sums them in split order.
    Opt-in (ME_SPLIT_K=1: it does not pay); the test lowers its K bound so that a tiny model has such launches.
    Repeated runs and a batch against its images give the same bits (inside the child); against ME_SPLIT_K=0 only the order of
    the f32 sums differs -- which moves 16-bit operand roundings downstream, so the two maps differ like any two correct 16-bit
    evaluations do (the same bound as the fused LayerNorm against its stand-alone launches; test_extract_depth_tiny and the
    full-size pairs hold the split form to the fp32 oracle)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, extra in (("split", {"ME_SPLIT_K": "1", "ME_SPLIT_K_MINK": "1024"}), ("whole", {"ME_SPLIT_K": "0"})):
        path = str(tmp_path / (name + ".npy"))
        r = subprocess.run([sys.executable, "-c", _SPLITK_CHILD, root, path], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(path)
    a, b = outs["split"], outs["whole"]
    err = rel_l2(a, b)
    print("split-K against whole-K launches: rel-L2", err, "identical" if np.array_equal(a, b) else "")
    assert np.isfinite(a).all() and 0 < err < 1.5e-3


_LN_FALLBACK_CHILD = """
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import matrix_eyes_amd as m
from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
cfg = m.ModelConfig(grid=8, embed_dim=512, num_heads=8, depth=2, tap_blocks=(0, 1), enc_dims=(64, 128, 128, 128),
                    dec_dim=256, head_dims=(32, 1))
w = synthetic_checkpoint(cfg)
rgb = synthetic_images(1, cfg.img_size, "structured", seed=5)
res = {}
# (a) a host-result call: the step that times out is run again without the fusion, the call succeeds
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(w)
res["fused_before"], _ = ctx.ln_fusion_state()
d, fov = ctx.extract_depth(rgb, None, want_fov=True)
res["state_after"] = ctx.ln_fusion_state()
d2, fov2 = ctx.extract_depth(rgb, None, want_fov=True)
res["state_after2"] = ctx.ln_fusion_state()
res["flags"] = ctx.status_flags()
ctx.close()
# (b) device-result calls are asynchronous: the NEXT entry reports the lost step, the one after runs unfused
ctx = m.Context(0, "f16", cfg)
ctx.load_state_dict(w)
out = torch.empty(1, cfg.img_size, cfg.img_size, dtype=torch.float32, device="cuda")
dev_rgb = torch.from_numpy(rgb).cuda()
ctx.extract_depth(dev_rgb, None, out=out)
ctx.synchronize()
try:
    ctx.extract_depth(dev_rgb, None, out=out)
    res["async_error"] = None
except m.MatrixEyesError as e:
    res["async_error"] = (e.code, e.message)
ctx.extract_depth(dev_rgb, None, out=out)
ctx.synchronize()
res["async_state"] = ctx.ln_fusion_state()
res["async_flags"] = ctx.status_flags()
d3 = out.cpu().numpy()
ctx.close()
np.savez(sys.argv[2], d=d, d2=d2, d3=d3, fov=fov, res=np.array(repr(res)))
"""


def test_fused_layernorm_timeout_falls_back_to_the_stand_alone_launches(tmp_path):
    """VERDICT r4 item 7 / ADVICE r4.  The fused residual + LayerNorm launch waits on sibling workgroups inside the
    launch (gemm_core.h resid_ln_epilogue); when they are not co-resident the wait runs into its bound and raises
    ME_STATUS_SYNC_TIMEOUT.  Forced here by a persistent grid of 8 workgroups (ME_GEMM_GRID_LIMIT: a workgroup's partner
    tile is then its OWN next tile) and a short bound (ME_LN_SPIN_LIMIT): a host-result call runs its step again on the
    stand-alone LayerNorm launches and returns the ME_LN_FUSE=0 depth bit for bit, fusion stays off for the context, the
    flag does not leak; after a device-result call the next entry fails with ME_ERR_HIP and the one after it runs unfused."""
    import ast
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, extra in (("forced", {"ME_GEMM_GRID_LIMIT": "8", "ME_LN_SPIN_LIMIT": "2000"}), ("unfused", {"ME_LN_FUSE": "0"}),
                        ("fused", {})):
        path = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", _LN_FALLBACK_CHILD, root, path], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        z = np.load(path)
        runs[name] = (z, ast.literal_eval(str(z["res"])), r.stderr)
    z, res, err = runs["forced"]
    ref = runs["unfused"][0]
    assert res["fused_before"] is True and res["state_after"] == (False, 1) and res["state_after2"] == (False, 1)
    assert res["flags"] == 0 and "LayerNorm fusion switched off" in err and err.count("LayerNorm fusion switched off") == 2
    assert np.isfinite(z["d"]).all() and np.array_equal(z["d"], ref["d"]) and np.array_equal(z["d2"], ref["d"])
    assert z["fov"][0] == ref["fov"][0]
    assert res["async_error"] is not None and res["async_error"][0] == 5 and "ME_STATUS_SYNC_TIMEOUT" in res["async_error"][1]
    assert res["async_state"][0] is False and res["async_flags"] == 0 and np.array_equal(z["d3"], ref["d"])
    # the healthy device: fusion stays on, nothing is re-run, and the fused result is the other realisation of the rounding
    zf, resf, errf = runs["fused"]
    assert resf["state_after"] == (True, 0) and resf["async_error"] is None and resf["async_state"] == (True, 0)
    assert "switched off" not in errf and np.array_equal(zf["d"], zf["d3"]) and 0 < rel_l2(zf["d"], ref["d"]) < 1.5e-3
    # the unfused run never had anything to fall back from
    assert runs["unfused"][1]["state_after"] == (True, 0)


def test_reconstruction_end_to_end_with_pt_checkpoint(tmp_path):
    """reconstruction.rs:155-205 through the host mirror: photo file + PyTorch .pt checkpoint in,
    depth-map PNG / stereogram PNG / OBJ+MTL out"""
    from PIL import Image
    cfg = m.ModelConfig.tiny()
    ckpt = tmp_path / "depth_pro_tiny.pt"
    torch.save(weights_for("tiny"), ckpt)                      # fp16 state dict, PyTorch names
    loader = m.DepthProModelLoader(str(ckpt), False, cfg=cfg)
    S = cfg.img_size
    rgb = synthetic_images(1, S)[0]
    photo = tmp_path / "photo.png"
    Image.fromarray(rgb).save(photo)
    seen = []
    m.extract_depth(0, loader, str(photo), str(tmp_path / "depth.png"), None, m.ImageOutputFormat.DepthMap(),
                    m.VertexMode.Color, progress=lambda pos, msg: seen.append(pos))
    got = np.asarray(Image.open(tmp_path / "depth.png"))
    assert got.shape == (S, S, 3) and seen[-1] == 1.0
    # same depth as the synthetic-checkpoint context (the .pt path loads the same tensors)
    ref_depth = loaded_ctx("tiny", "f16").extract_depth(rgb[None], None)[0]
    dm = m.DepthMap(loaded_ctx("tiny", "f16"), ref_depth, (S, S))
    assert np.array_equal(got, dm.depth_map_rgb())
    m.extract_depth(0, loader, str(photo), str(tmp_path / "mesh.obj"), 50.0, m.ImageOutputFormat.DepthMap(),
                    m.VertexMode.Texture)
    text = (tmp_path / "mesh.obj").read_text()
    assert text.startswith("mtllib mesh.mtl\nusemtl Textured\nvt ") and "\nf " in text
    assert (tmp_path / "mesh.mtl").read_text().endswith(f"map_Kd {photo}\n\n")
    noise = np.random.default_rng(7).integers(0, 256, size=(S, S, 3), dtype=np.uint8)
    m.extract_depth(0, loader, str(photo), str(tmp_path / "stereo.png"), 50.0,
                    m.ImageOutputFormat.Stereogram(None, 1 / 16), m.VertexMode.Plain, noise=noise)
    assert np.asarray(Image.open(tmp_path / "stereo.png")).shape == (S, S, 3)
    # a checkpoint with a missing tensor is refused like mod.rs:241-243
    bad = dict(weights_for("tiny"))
    del bad["fov.head.4.bias"]
    torch.save(bad, tmp_path / "bad.pt")
    with pytest.raises(m.MatrixEyesError) as e:
        m.DepthProModelLoader(str(tmp_path / "bad.pt"), False, cfg=cfg).context(0)
    assert e.value.code == 3
