"""Output back end parity (-m gpu): the HIP kernels behind me_depth_clamp_minmax / me_stereogram /
me_depthmap_rgb / me_mesh_index / me_mesh_vertices / me_output_mesh, through the C ABI, BIT-EXACT
against the C oracle (oracle/output_oracle.c), the hand-derived known answers, and size-independent
properties at the full 1536x1536 size."""
import json
import os
import struct

import numpy as np
import pytest

import matrix_eyes_amd as m
from oracle import output_oracle as OO
from util import ctx_for

pytestmark = pytest.mark.gpu
KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "output_known_answers.json")))


def _ctx():
    return ctx_for("tiny", "f16")


def _depth(n, seed=0, kind="scene"):
    """inverse-depth-like maps: smooth background + a few nearer rectangles (discontinuities)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, n, dtype=np.float32), np.linspace(0, 1, n, dtype=np.float32),
                         indexing="ij")
    d = 0.2 + 0.15 * np.sin(3 * xx + 2 * yy) + 0.1 * yy
    if kind == "scene":
        for _ in range(6):
            x0, y0 = rng.integers(0, n - n // 4, size=2)
            w, h = rng.integers(n // 16, n // 4, size=2)
            d[y0:y0 + h, x0:x0 + w] += rng.uniform(0.2, 1.5)
        d += rng.normal(0, 0.002, size=d.shape).astype(np.float32)
    return np.ascontiguousarray(d.astype(np.float32))


def test_known_answers():
    ctx = _ctx()
    c = KA["clamp"]
    dm = m.DepthMap(ctx, np.array(c["input"], np.float32).reshape(1, -1), (5, 1))
    assert dm.data.flatten().tolist() == np.array(c["clamped"], np.float32).tolist()
    assert dm.inverse_depth_range() == (np.float32(c["min"]), np.float32(c["max"]))

    c = KA["stereogram_step"]
    dm = m.DepthMap(ctx, np.array(c["depth"], np.float32), (c["out_w"], c["out_h"]))
    noise = np.arange(c["out_w"] * c["out_h"] * 3, dtype=np.uint8).reshape(c["out_h"], c["out_w"], 3)
    out = dm.stereogram(None, c["amplitude"], noise)
    assert np.array_equal(out[0], noise[0][c["source_index"]])

    c = KA["stereogram_flat"]
    dm = m.DepthMap(ctx, np.array(c["depth"], np.float32), (c["out_w"], c["out_h"]))
    noise = np.random.default_rng(5).integers(0, 256, size=(c["out_h"], c["out_w"], 3), dtype=np.uint8)
    out = dm.stereogram(None, c["amplitude"], noise)
    for y in range(c["out_h"]):
        assert np.array_equal(out[y], noise[y][np.arange(c["out_w"]) % c["pattern_width"]])

    c = KA["mesh_3x3"]
    dm = m.DepthMap(ctx, np.array(c["depth"], np.float32), (3, 3))
    vi, nv, faces = dm.mesh_index()
    assert vi.tolist() == c["vertex_index"] and nv == c["nvertices"] and faces.tolist() == c["faces"]

    for which in ("keep", "drop"):
        c = KA["mesh_threshold"][which]
        a = struct.unpack("<f", struct.pack("<I", int(c["a_bits"], 16)))[0]
        dm = m.DepthMap(ctx, np.array([[1.0, a], [1.0, 1.0]], np.float32), (2, 2))
        assert len(dm.mesh_index()[2]) == c["nfaces"]

    c = KA["colormap"]
    dm = m.DepthMap(ctx, np.array(c["depth"], np.float32).reshape(1, -1), (2, 1))
    assert dm.depth_map_rgb().reshape(-1, 3).tolist() == c["rgb"]


@pytest.mark.parametrize("n,out_size,amplitude", [
    (1536, None, 1.0 / 16.0),          # BASELINE config 5 geometry
    (512, (640, 480), 1.0 / 16.0),     # non-square original size
    (512, (1001, 777), 0.125),
    (256, (300, 200), 0.01),           # small pattern
    (64, (40, 30), 0.0005),            # degenerate: pattern_width rounds to 0
])
def test_stereogram_bit_exact(n, out_size, amplitude):
    ctx = _ctx()
    d = _depth(n, seed=n)
    size = out_size or (n, n)
    dm = m.DepthMap(ctx, d, size)
    od, mn, mx = OO.clamp_minmax(d)
    assert np.array_equal(dm.data, od) and dm.inverse_depth_range() == (mn, mx)
    noise = np.random.default_rng(99).integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)
    got = dm.stereogram(None, amplitude, noise)
    want = OO.stereogram(od, mn, mx, size[0], size[1], amplitude, noise)
    assert np.array_equal(got, want)


def test_stereogram_resize_scale():
    ctx = _ctx()
    d = _depth(512, seed=3)
    dm = m.DepthMap(ctx, d, (801, 603))
    w, h = dm.stereogram_size(0.37)
    assert (w, h) == (296, 223)        # round(801*0.37) = round(296.37), round(603*0.37) = round(223.11)
    noise = np.random.default_rng(1).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    got = dm.stereogram(0.37, 1.0 / 16.0, noise)
    od, mn, mx = OO.clamp_minmax(d)
    assert np.array_equal(got, OO.stereogram(od, mn, mx, w, h, 1.0 / 16.0, noise))


def test_clamp_with_outliers_and_device_pointer():
    import torch
    ctx = _ctx()
    d = _depth(700, seed=11)
    d[5, 7], d[100, 3], d[8, 8] = 1e-4, 1e4, 0.0041
    od, mn, mx = OO.clamp_minmax(d)
    dm = m.DepthMap(ctx, d, (700, 700))
    assert np.array_equal(dm.data, od) and dm.inverse_depth_range() == (mn, mx)
    t = torch.from_numpy(d).cuda()
    import ctypes as C
    a, b = C.c_float(), C.c_float()
    ctx._check(ctx.lib.me_depth_clamp_minmax(ctx.handle, C.c_void_p(t.data_ptr()), t.numel(), C.byref(a), C.byref(b)))
    assert np.array_equal(t.cpu().numpy(), od) and (a.value, b.value) == (mn, mx)


@pytest.mark.parametrize("n", [1536, 333])
def test_depthmap_rgb_bit_exact(n):
    ctx = _ctx()
    d = _depth(n, seed=n + 1)
    dm = m.DepthMap(ctx, d, (n, n))
    od, mn, mx = OO.clamp_minmax(d)
    assert np.array_equal(dm.depth_map_rgb(), OO.depthmap_rgb(od, mn, mx))


@pytest.mark.parametrize("n,kind", [(1536, "scene"), (257, "scene"), (64, "smooth")])
def test_mesh_index_and_vertices_bit_exact(n, kind):
    ctx = _ctx()
    d = _depth(n, seed=n + 2, kind=kind)
    dm = m.DepthMap(ctx, d, (n * 2, n))
    od, _, _ = OO.clamp_minmax(d)
    vi, nv, faces = dm.mesh_index()
    ovi, onv, ofaces = OO.mesh_index(od)
    assert nv == onv and np.array_equal(vi, ovi) and np.array_equal(faces, ofaces)
    assert 0 < len(faces) < 2 * (n - 1) * (n - 1) or kind == "smooth"
    uv, xyz = dm.mesh_vertices(vi, nv)
    ouv, oxyz = OO.mesh_vertices(od, ovi, onv, (n * 2, n))
    assert np.array_equal(uv, ouv) and np.array_equal(xyz, oxyz)
    # counting only (faces == NULL)
    _, nv2, nf2 = dm.mesh_index(want_faces=False)
    assert (nv2, nf2) == (nv, len(faces))


def test_full_grid_mesh_closed_form_at_full_size():
    ctx = _ctx()
    n = 1536
    dm = m.DepthMap(ctx, np.full((n, n), 0.7, np.float32), (n, n))
    vi, nv, faces = dm.mesh_index()
    assert nv == n * n and len(faces) == 2 * (n - 1) * (n - 1)
    vi = vi.reshape(n, n)
    assert np.array_equal(vi[0], np.concatenate([[0], 2 * np.arange(1, n)]))
    assert np.array_equal(vi[1], np.concatenate([[1], 2 * np.arange(1, n) + 1]))
    assert np.array_equal(vi[2:], (2 * n + np.arange((n - 2) * n)).reshape(n - 2, n))
    # every face references valid, distinct vertices
    assert faces.min() == 0 and faces.max() == nv - 1
    assert np.all(faces[:, 0] != faces[:, 1]) and np.all(faces[:, 1] != faces[:, 2])


@pytest.mark.parametrize("mode", ["plain", "color", "texture"])
@pytest.mark.parametrize("ext", ["obj", "ply", "OBJ"])
def test_output_mesh_files(tmp_path, mode, ext):
    from PIL import Image
    ctx = _ctx()
    n = 48
    d = _depth(n, seed=8)
    src = tmp_path / "photo.png"
    rgb = np.random.default_rng(2).integers(0, 256, size=(n, n, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(src)
    dm = m.DepthMap(ctx, d, (n, n))
    dest = tmp_path / f"mesh.{ext}"
    vm = {"plain": m.VertexMode.Plain, "color": m.VertexMode.Color, "texture": m.VertexMode.Texture}[mode]
    dm.output_image(str(dest), str(src), m.ImageOutputFormat.DepthMap(), vm)
    od, _, _ = OO.clamp_minmax(d)
    vi, nv, faces = OO.mesh_index(od)
    uv, xyz = OO.mesh_vertices(od, vi, nv, (n, n))
    colors = None
    if mode == "color":      # source image is already n x n: the Lanczos resize is the identity
        colors = np.zeros((nv, 3), np.uint8)
        colors[vi[vi >= 0]] = rgb.reshape(-1, 3)[vi >= 0]
    if ext.lower() == "obj":
        assert dest.read_text() == OO.obj_text(uv, xyz, faces, mode, "mesh", colors)
        mtl = tmp_path / "mesh.mtl"
        if mode == "texture":
            assert mtl.read_text() == OO.mtl_text(str(src))
        else:
            assert not mtl.exists()
    else:
        assert dest.read_bytes() == OO.ply_bytes(xyz, faces, mode, colors)


def test_output_image_files(tmp_path):
    from PIL import Image
    ctx = _ctx()
    n = 96
    d = _depth(n, seed=4)
    dm = m.DepthMap(ctx, d, (n, n))
    od, mn, mx = OO.clamp_minmax(d)
    dm.output_image(str(tmp_path / "d.png"), "unused", m.ImageOutputFormat.DepthMap(), m.VertexMode.Plain)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "d.png")), OO.depthmap_rgb(od, mn, mx))
    noise = np.random.default_rng(0).integers(0, 256, size=(n, n, 3), dtype=np.uint8)
    dm.output_image(str(tmp_path / "s.png"), "unused", m.ImageOutputFormat.Stereogram(None, 1 / 16), m.VertexMode.Plain,
                    noise=noise)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "s.png")), OO.stereogram(od, mn, mx, n, n, 1 / 16, noise))


def test_bad_destination_is_an_error(tmp_path):
    import ctypes as C
    ctx = _ctx()
    d = _depth(16)
    rc = ctx.lib.me_output_mesh(ctx.handle, C.c_void_p(d.ctypes.data), 16, 16, 16, 16,
                                str(tmp_path / "x.txt").encode(), b"src", 0, None)
    assert rc == 1
    rc = ctx.lib.me_output_mesh(ctx.handle, C.c_void_p(d.ctypes.data), 16, 16, 16, 16,
                                b"/nonexistent-dir/x.obj", b"src", 0, None)
    assert rc == 7      # OutputError::Io


def test_write_behind_files_and_deferred_errors(tmp_path):
    """me_ctx_set_write_behind: the OBJ / MTL files written by a host thread behind the caller are the bytes of the
    synchronous call (five meshes through the two alternating buffers), me_output_flush waits for them, and a write
    that fails is reported by the call that next waits for it."""
    import ctypes as C
    ctx = m.Context(0, "f16", m.ModelConfig.tiny())
    n = 160
    depths = [_depth(n, seed=20 + i) for i in range(5)]
    want = []
    for i, d in enumerate(depths):
        m.DepthMap(ctx, d, (n, n)).output_mesh(str(tmp_path / f"sync{i}.obj"), "photo.jpg", m.VertexMode.Texture)
        want.append(((tmp_path / f"sync{i}.obj").read_bytes(), (tmp_path / f"sync{i}.mtl").read_bytes()))
    ctx.set_write_behind(True)
    for i, d in enumerate(depths):
        m.DepthMap(ctx, d, (n, n)).output_mesh(str(tmp_path / f"behind{i}.obj"), "photo.jpg", m.VertexMode.Texture)
    ctx.output_flush()
    for i in range(5):
        obj = (tmp_path / f"behind{i}.obj").read_bytes().replace(b"mtllib behind%d.mtl" % i, b"mtllib sync%d.mtl" % i)
        assert obj == want[i][0] and (tmp_path / f"behind{i}.mtl").read_bytes() == want[i][1]
    # a failing write: the call itself succeeds (the text reached pinned memory), the flush reports ME_ERR_IO = 7
    d = depths[0]
    rc = ctx.lib.me_output_mesh(ctx.handle, C.c_void_p(d.ctypes.data), n, n, n, n, b"/nonexistent-dir/x.obj", b"src", 0, None)
    assert rc == 0
    assert ctx.lib.me_output_flush(ctx.handle) == 7 and b"write-behind" in ctx.lib.me_last_error(ctx.handle)
    assert ctx.lib.me_output_flush(ctx.handle) == 0          # reported once
    # ... or the later call that has to wait for that buffer (the third after it)
    assert ctx.lib.me_output_mesh(ctx.handle, C.c_void_p(d.ctypes.data), n, n, n, n, b"/nonexistent-dir/y.obj", b"src", 0, None) == 0
    m.DepthMap(ctx, d, (n, n)).output_mesh(str(tmp_path / "ok1.obj"), "photo.jpg", m.VertexMode.Plain)
    with pytest.raises(m.MatrixEyesError) as e:
        m.DepthMap(ctx, d, (n, n)).output_mesh(str(tmp_path / "ok2.obj"), "photo.jpg", m.VertexMode.Plain)
    assert e.value.code == 7
    ctx.set_write_behind(False)
    m.DepthMap(ctx, d, (n, n)).output_mesh(str(tmp_path / "sync_again.obj"), "photo.jpg", m.VertexMode.Texture)
    assert (tmp_path / "sync_again.obj").read_bytes().replace(b"mtllib sync_again.mtl", b"mtllib sync0.mtl") == want[0][0]


def test_config4_chain_full_size(tmp_path):
    """BASELINE configs[4] for one image at full size: model -> DepthMap -> stereogram -> textured OBJ.  The depth
    stays on the GPU from the model's output tensor through DepthMap::new (clamp + range, output.rs:44-75) into the
    stereogram kernel (no host round trip); given that depth, the stereogram is bit-exact against the C oracle and
    the 450 MB OBJ + MTL are byte-identical to the oracle's writer (output.rs:484-630)."""
    import filecmp
    import torch
    from matrix_eyes_amd.synthetic import synthetic_images
    from util import loaded_ctx
    ctx = loaded_ctx("full", "f16")
    S = ctx.cfg.img_size
    rgb = torch.from_numpy(synthetic_images(1, S, "structured", seed=31)).cuda()
    depth = torch.empty(1, S, S, dtype=torch.float32, device="cuda")
    ctx.extract_depth(rgb, None, out=depth)
    ctx.synchronize()                                       # the context runs on its own stream, not torch's
    raw = depth[0].cpu().numpy().copy()                     # before DepthMap::new clamps it in place
    ddm = m.DeviceDepthMap(ctx, depth[0], (S, S))
    noise = np.random.default_rng(99).integers(0, 256, size=(S, S, 3), dtype=np.uint8)
    stereo = ddm.stereogram(1.0 / 16.0, torch.from_numpy(noise).cuda())
    colour = ddm.depth_map_rgb()
    od, mn, mx = OO.clamp_minmax(raw)
    assert ddm.inverse_depth_range() == (mn, mx) and np.array_equal(depth[0].cpu().numpy(), od)
    assert np.array_equal(stereo.cpu().numpy(), OO.stereogram(od, mn, mx, S, S, 1.0 / 16.0, noise))
    assert np.array_equal(colour.cpu().numpy(), OO.depthmap_rgb(od, mn, mx))
    # the host-side DepthMap of the same depth agrees with the device chain
    dm = m.DepthMap(ctx, raw, (S, S))
    assert dm.inverse_depth_range() == (mn, mx) and np.array_equal(dm.data, od)
    # output.rs:195-261: textured OBJ + MTL, whole file against the oracle's writer
    dest = tmp_path / "mesh.obj"
    dm.output_image(str(dest), "photo.jpg", m.ImageOutputFormat.DepthMap(), m.VertexMode.Texture)
    vi, nv, faces = OO.mesh_index(od)
    uv, xyz = OO.mesh_vertices(od, vi, nv, (S, S))
    assert nv > 100000 and len(faces) > 100000              # a real mesh (the random-weight depth is rough: many quads fail the 1.025 test)
    OO.write_obj(str(tmp_path / "oracle.obj"), uv, xyz, faces, "texture", "mesh")
    assert dest.stat().st_size == (tmp_path / "oracle.obj").stat().st_size > 10e6
    assert filecmp.cmp(dest, tmp_path / "oracle.obj", shallow=False)
    assert (tmp_path / "mesh.mtl").read_text() == OO.mtl_text("photo.jpg")


def test_config4_chain_batch_of_two(tmp_path):
    """BASELINE configs[4] in batch form: TWO images per step through depth -> DepthMap::new -> stereogram -> textured
    OBJ, everything from the model's batched output tensor on the device (DeviceDepthMap.output_mesh reads the device
    depth).  Per image: the clamp + range, stereogram and colour map are bit-exact against the C oracle and the
    OBJ + MTL byte-identical to the oracle's writer; the two images' products differ from each other."""
    import filecmp
    import torch
    from matrix_eyes_amd.synthetic import synthetic_images
    from util import loaded_ctx
    ctx = loaded_ctx("full", "f16")
    S, B = ctx.cfg.img_size, 2
    rgb = torch.from_numpy(synthetic_images(B, S, "structured", seed=57)).cuda()
    depth = torch.empty(B, S, S, dtype=torch.float32, device="cuda")
    ctx.extract_depth(rgb, None, out=depth)
    ctx.synchronize()
    raw = depth.cpu().numpy().copy()
    noise = np.random.default_rng(99).integers(0, 256, size=(S, S, 3), dtype=np.uint8)
    noise_dev = torch.from_numpy(noise).cuda()
    maps = [m.DeviceDepthMap(ctx, depth[b], (S, S)) for b in range(B)]
    stereo = [maps[b].stereogram(1.0 / 16.0, noise_dev) for b in range(B)]
    for b in range(B):
        maps[b].output_mesh(str(tmp_path / f"mesh{b}.obj"), f"photo{b}.jpg", m.VertexMode.Texture)
    sizes = []
    for b in range(B):
        od, mn, mx = OO.clamp_minmax(raw[b])
        assert maps[b].inverse_depth_range() == (mn, mx) and np.array_equal(depth[b].cpu().numpy(), od)
        assert np.array_equal(stereo[b].cpu().numpy(), OO.stereogram(od, mn, mx, S, S, 1.0 / 16.0, noise))
        vi, nv, faces = OO.mesh_index(od)
        uv, xyz = OO.mesh_vertices(od, vi, nv, (S, S))
        OO.write_obj(str(tmp_path / f"oracle{b}.obj"), uv, xyz, faces, "texture", f"mesh{b}")
        assert filecmp.cmp(tmp_path / f"mesh{b}.obj", tmp_path / f"oracle{b}.obj", shallow=False)
        assert (tmp_path / f"mesh{b}.mtl").read_text() == OO.mtl_text(f"photo{b}.jpg")
        sizes.append((tmp_path / f"mesh{b}.obj").stat().st_size)
    assert min(sizes) > 10e6 and not filecmp.cmp(tmp_path / "mesh0.obj", tmp_path / "mesh1.obj", shallow=False)
    assert not torch.equal(stereo[0], stereo[1])


def test_config4_chain_batch_of_eight(tmp_path):
    """BASELINE configs[4] at its per-GPU batch: EIGHT images per step (64 images over 8 GPUs) through depth ->
    DepthMap::new -> stereogram -> textured OBJ from the model's batched output tensor, the eight OBJ files written
    behind the caller as bench.py --chain runs it (two pinned buffers, flushed at the end).  Images 1 and 6 of the batch
    are held to the C oracle: clamp + range and stereogram bit-exact, OBJ + MTL byte-identical to the oracle's writer;
    their depth is bit for bit that of a batch of one; the eight files differ from one another."""
    import filecmp
    import torch
    from matrix_eyes_amd.synthetic import synthetic_images
    from util import loaded_ctx
    ctx = loaded_ctx("full", "f16")
    S, B = ctx.cfg.img_size, 8
    rgb = torch.from_numpy(synthetic_images(B, S, "structured", seed=211)).cuda()
    depth = torch.empty(B, S, S, dtype=torch.float32, device="cuda")
    ctx.extract_depth(rgb, None, out=depth)
    ctx.synchronize()
    checked = (1, 6)
    raw = {b: depth[b].cpu().numpy().copy() for b in checked}
    for b in checked:
        one = ctx.extract_depth(rgb[b:b + 1].cpu().numpy(), None)
        assert np.array_equal(one[0], raw[b])
    noise = np.random.default_rng(99).integers(0, 256, size=(S, S, 3), dtype=np.uint8)
    noise_dev = torch.from_numpy(noise).cuda()
    ctx.set_write_behind(2)
    try:
        maps = [m.DeviceDepthMap(ctx, depth[b], (S, S)) for b in range(B)]
        stereo = [maps[b].stereogram(1.0 / 16.0, noise_dev) for b in range(B)]
        for b in range(B):
            maps[b].output_mesh(str(tmp_path / f"mesh{b}.obj"), f"photo{b}.jpg", m.VertexMode.Texture)
        ctx.output_flush()
    finally:
        ctx.set_write_behind(0)
    for b in checked:
        od, mn, mx = OO.clamp_minmax(raw[b])
        assert maps[b].inverse_depth_range() == (mn, mx) and np.array_equal(depth[b].cpu().numpy(), od)
        assert np.array_equal(stereo[b].cpu().numpy(), OO.stereogram(od, mn, mx, S, S, 1.0 / 16.0, noise))
        vi, nv, faces = OO.mesh_index(od)
        uv, xyz = OO.mesh_vertices(od, vi, nv, (S, S))
        OO.write_obj(str(tmp_path / f"oracle{b}.obj"), uv, xyz, faces, "texture", f"mesh{b}")
        assert filecmp.cmp(tmp_path / f"mesh{b}.obj", tmp_path / f"oracle{b}.obj", shallow=False)
        assert (tmp_path / f"mesh{b}.mtl").read_text() == OO.mtl_text(f"photo{b}.jpg")
    sizes = [(tmp_path / f"mesh{b}.obj").stat().st_size for b in range(B)]
    assert min(sizes) > 10e6 and len(set(sizes)) == B
    assert len({stereo[b].cpu().numpy().tobytes() for b in range(B)}) == B
    for b in range(B):
        (tmp_path / f"mesh{b}.obj").unlink()               # 0.6 - 1.3 GB of text on the test box's tmpfs


def test_config4_chain_pipelined_across_images(tmp_path):
    """bench.py --chain's pipelined form (me_ctx_set_output_overlap) -- image i + 1's me_extract_depth
    is queued BEFORE image i's DepthMap::new / stereogram / OBJ calls, which run on the context's output stream behind the
    step that wrote their depth buffer; two depth buffers in turn, OBJ files written behind the caller.  Four different
    images: every stereogram and every OBJ + MTL is byte for byte what the serial, synchronised order produces, the depth
    buffers hold the clamped maps of the LAST two images (nothing overwrote a buffer that was still being read), and the
    serial products themselves are the oracle's for one image."""
    import filecmp
    import torch
    from matrix_eyes_amd.synthetic import synthetic_images
    from util import loaded_ctx
    ctx = loaded_ctx("full", "f16")
    S, N = ctx.cfg.img_size, 4
    rgbs = [torch.from_numpy(synthetic_images(1, S, "structured", seed=900 + i)).cuda() for i in range(N)]
    noise = np.random.default_rng(99).integers(0, 256, size=(S, S, 3), dtype=np.uint8)
    noise_dev = torch.from_numpy(noise).cuda()
    bufs = [torch.empty(1, S, S, dtype=torch.float32, device="cuda") for _ in range(2)]

    for tag in ("serial", "piped"):
        (tmp_path / tag).mkdir()

    def outputs(i, tag, stereos, clamped):
        dm = m.DeviceDepthMap(ctx, bufs[i & 1][0], (S, S))
        stereos.append(dm.stereogram(1.0 / 16.0, noise_dev))
        # (the same file name in both runs: the OBJ's first line names its .mtl after the file's stem)
        dm.output_mesh(str(tmp_path / tag / f"mesh{i}.obj"), f"photo{i}.jpg", m.VertexMode.Texture)
        clamped.append(dm)

    # serial: every step synchronised before the next begins
    serial_stereo, serial_maps, raw0 = [], [], None
    for i in range(N):
        ctx.extract_depth(rgbs[i], None, out=bufs[i & 1])
        ctx.synchronize()
        if i == 0:
            raw0 = bufs[0][0].cpu().numpy().copy()
        outputs(i, "serial", serial_stereo, serial_maps)
        ctx.synchronize()
    serial_stereo = [t.cpu().numpy() for t in serial_stereo]
    serial_last = [bufs[k][0].cpu().numpy().copy() for k in range(2)]
    # pipelined
    ctx.set_output_overlap(True)
    ctx.set_write_behind(2)
    try:
        piped_stereo, piped_maps = [], []
        ctx.extract_depth(rgbs[0], None, out=bufs[0])
        for i in range(N):
            if i + 1 < N:
                ctx.extract_depth(rgbs[i + 1], None, out=bufs[(i + 1) & 1])
            outputs(i, "piped", piped_stereo, piped_maps)
        ctx.output_flush()
        ctx.synchronize()
    finally:
        ctx.set_write_behind(0)
        ctx.set_output_overlap(False)
    assert ctx.status_flags() == 0
    for i in range(N):
        assert np.array_equal(piped_stereo[i].cpu().numpy(), serial_stereo[i]), i
        assert filecmp.cmp(tmp_path / "piped" / f"mesh{i}.obj", tmp_path / "serial" / f"mesh{i}.obj", shallow=False), i
        assert (tmp_path / "piped" / f"mesh{i}.mtl").read_text() == OO.mtl_text(f"photo{i}.jpg")
    for k in range(2):
        assert np.array_equal(bufs[k][0].cpu().numpy(), serial_last[k])
    assert len({t.tobytes() for t in serial_stereo}) == N           # four different images
    # the serial products are the oracle's (image 0)
    od, mn, mx = OO.clamp_minmax(raw0)
    assert np.array_equal(serial_stereo[0], OO.stereogram(od, mn, mx, S, S, 1.0 / 16.0, noise))
    vi, nv, faces = OO.mesh_index(od)
    uv, xyz = OO.mesh_vertices(od, vi, nv, (S, S))
    OO.write_obj(str(tmp_path / "oracle0.obj"), uv, xyz, faces, "texture", "mesh0")
    assert filecmp.cmp(tmp_path / "serial" / "mesh0.obj", tmp_path / "oracle0.obj", shallow=False)
    for f in tmp_path.rglob("*.obj"):
        f.unlink()


def test_device_number_formatter_prints_like_rust():
    """obj_format.hip's digit generator on the GPU (64 x 128-bit multiplications by __umul64hi, tables in device
    memory) against the oracle's rust_display_f64 (the shortest round-trip digits of CPython's repr laid out positionally): random
    bit patterns over the whole f64 range, widened f32 values, 1 - v, c / 255 and the edge cases."""
    import torch
    ctx = _ctx()
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 2 ** 64, size=200000, dtype=np.uint64)
    any_f64 = bits.view(np.float64)
    f32 = rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    f32 = f32[np.isfinite(f32)]
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e21, 1e22, 1e23, 1e-7, 5e-324, 2.2250738585072014e-308,
                        1.7976931348623157e308, 9007199254740993.0, 0.3, 1e15, 1e16, 1e17, np.inf, -np.inf, np.nan,
                        0.10000000149011612, 1 / 3, 2 / 3, 8.5e-323])
    vals = np.concatenate([special, any_f64[~np.isnan(any_f64)], f32.astype(np.float64), 1.0 - np.abs(f32[:50000]).astype(np.float64),
                           np.arange(256) / 255.0, rng.uniform(-250, 250, 100000).astype(np.float32).astype(np.float64)])
    stride = 352
    v = torch.from_numpy(vals).cuda()
    text = torch.zeros(len(vals) * stride, dtype=torch.uint8, device="cuda")
    lens = torch.zeros(len(vals), dtype=torch.int32, device="cuda")
    ctx._check(ctx.lib.me_op_format_f64(ctx.handle, v.data_ptr(), len(vals), text.data_ptr(), stride, lens.data_ptr()))
    ctx.synchronize()
    text, lens = text.cpu().numpy().reshape(len(vals), stride), lens.cpu().numpy()
    step = max(1, len(vals) // 60000)           # every value's length, a dense sample's characters
    for i in list(range(0, len(special))) + list(range(len(special), len(vals), step)):
        want = OO.rust_display_f64(float(vals[i]))
        got = text[i, :lens[i]].tobytes().decode()
        assert got == want, (i, repr(float(vals[i])), got, want)
    assert lens.min() >= 1 and lens.max() <= 344


@pytest.mark.parametrize("mode", ["plain", "color", "texture"])
def test_obj_text_on_device_equals_the_file_and_the_oracle(tmp_path, mode):
    """me_mesh_obj_text: the device buffer holds exactly the bytes me_output_mesh writes, and those are the oracle
    writer's -- on a non-square original size (xm != ym) and a depth map that spans the whole clamp range, so that
    coordinates of every magnitude from 1e-3 to 250 (and exact zeros on the optical axis) are printed."""
    import torch
    ctx = _ctx()
    n = 200
    rng = np.random.default_rng(3)
    d = np.exp(rng.uniform(np.log(0.004), np.log(10.0), size=(n, n))).astype(np.float32)
    d[40:120, 30:170] = 0.25                      # a flat patch: many kept faces
    d[:, 100] = d[:, 99]
    dm = m.DepthMap(ctx, d, (3000, 2000))
    vm = {"plain": m.VertexMode.Plain, "color": m.VertexMode.Color, "texture": m.VertexMode.Texture}[mode]
    colors = rng.integers(0, 256, size=(n, n, 3), dtype=np.uint8) if mode == "color" else None
    ddm = m.DeviceDepthMap(ctx, torch.from_numpy(dm.data).cuda(), (3000, 2000))
    cdev = torch.from_numpy(colors).cuda() if colors is not None else None
    text = ddm.obj_text("scene", vm, cdev).cpu().numpy().tobytes()
    ddm.output_mesh(str(tmp_path / "scene.obj"), "photo.jpg", vm, cdev)
    assert (tmp_path / "scene.obj").read_bytes() == text
    vi, nv, faces = OO.mesh_index(dm.data)
    uv, xyz = OO.mesh_vertices(dm.data, vi, nv, (3000, 2000))
    vcol = None
    if colors is not None:
        vcol = np.zeros((nv, 3), np.uint8)
        flat = colors.reshape(-1, 3)
        vcol[vi[vi >= 0]] = flat[vi >= 0]
    assert nv > 1000 and len(faces) > 1000
    assert text.decode() == OO.obj_text(uv, xyz, faces, mode, "scene", vcol)
