"""The C++ host layer (matrix-eyes_amd/host/): the compiled twin of the reference's main.rs /
reconstruction.rs / DepthProModelLoader above the C ABI.  CPU part: checkpoint reader against torch, PNG
codec, JPEG decoder, EXIF and Lanczos3 against Pillow, CLI usage and exit codes.  GPU part (-m gpu): the CLI end to end on the
test geometry, against the Python mirror driving the same library."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "matrix-eyes_amd")
SELFTEST = os.path.join(PKG, "host_selftest")
CLI = os.path.join(PKG, "matrix-eyes-hip")


@pytest.fixture(scope="module", autouse=True)
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
    assert os.path.exists(SELFTEST) and os.path.exists(CLI)


def _fnv1a64(b: bytes) -> int:
    h = 1469598103934665603
    for x in np.frombuffer(b, np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _tiny_checkpoint(path):
    import matrix_eyes_amd as m
    from matrix_eyes_amd.synthetic import synthetic_checkpoint
    ck = {k: torch.as_tensor(v) for k, v in synthetic_checkpoint(m.ModelConfig.tiny()).items()}
    torch.save(ck, path)
    return ck


def test_checkpoint_reader_matches_torch(tmp_path):
    ck = _tiny_checkpoint(str(tmp_path / "tiny.pt"))
    r = subprocess.run([SELFTEST, "pt", str(tmp_path / "tiny.pt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert [l[0] for l in lines] == list(ck.keys())            # state_dict order is kept
    names = {torch.float16: "f16", torch.float32: "f32"}
    for l in lines[:: max(1, len(lines) // 40)]:               # every ~7th tensor: shape, dtype, bytes
        t = ck[l[0]].contiguous()
        assert l[1] == names[t.dtype] and [int(x) for x in l[2:-1]] == list(t.shape)
        assert int(l[-1], 16) == _fnv1a64(t.numpy().tobytes())


def test_checkpoint_reader_errors(tmp_path):
    bad = tmp_path / "bad.pt"
    bad.write_bytes(b"not a zip archive at all, just some bytes" * 3)
    r = subprocess.run([SELFTEST, "pt", str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "zip" in r.stderr
    r = subprocess.run([SELFTEST, "pt", str(tmp_path / "missing.pt")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


def test_checkpoint_reader_survives_corrupt_files(tmp_path):
    """Byte flips (biased towards the pickle and the zip directory), deletions and truncations of torch.save
    archives, plain and wrapped ({"state_dict": ..., extras}): the reader either lists the tensors or reports a
    CheckpointError (LoaderError::Pytorch in the reference, mod.rs:229-233) -- exit code 0 or 1, never a signal.
    The same corpus ran clean under ASan + UBSan during development."""
    import collections
    import random
    sd = collections.OrderedDict((f"layer{i}.weight", torch.randn(5, 7).half()) for i in range(4))
    sd["bias"] = torch.zeros(3)
    torch.save(sd, tmp_path / "a.pt")
    torch.save({"state_dict": sd, "epoch": 3, "extra": [1, 2, {"k": (1.5, None)}]}, tmp_path / "b.pt")
    seeds = [(tmp_path / n).read_bytes() for n in ("a.pt", "b.pt")]
    rnd = random.Random(3)
    codes = set()
    for it in range(300):
        b = bytearray(seeds[it % 2])
        mode = rnd.random()
        if mode < 0.7:
            for _ in range(rnd.choice([1, 1, 2, 4, 8])):
                region = rnd.random()
                if region < 0.4:
                    i = rnd.randrange(0, min(len(b), 1200))
                elif region < 0.8:
                    i = rnd.randrange(max(0, len(b) - 1200), len(b))
                else:
                    i = rnd.randrange(len(b))
                b[i] = rnd.randrange(256)
        elif mode < 0.85:
            i = rnd.randrange(len(b))
            del b[i:i + rnd.randrange(1, 40)]
        else:
            b = b[:rnd.randrange(4, len(b))]
        (tmp_path / "f.bin").write_bytes(bytes(b))
        r = subprocess.run([SELFTEST, "pt", str(tmp_path / "f.bin")], capture_output=True, timeout=60)
        assert r.returncode in (0, 1), (it, r.returncode, r.stderr[-300:])
        codes.add(r.returncode)
    assert codes == {0, 1}


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "P"])
def test_png_codec_against_pillow(tmp_path, mode):
    from PIL import Image
    rng = np.random.default_rng(5)
    img = Image.fromarray(rng.integers(0, 256, (41, 67, 3), dtype=np.uint8)).convert(mode)
    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    img.save(src)
    r = subprocess.run([SELFTEST, "png", src, dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.asarray(Image.open(dst)), np.asarray(img.convert("RGB")))


def _encode_png(samples, depth, ctype, interlace, palette=None, seed=0):
    """A PNG file of `samples` [h, w, channels] (ints below 2^depth) written pass by pass, scanline by scanline, with
    filter types 0 / 1 / 2 chosen at random per line -- every bit depth, colour type and Adam7, which Pillow's
    writer does not produce."""
    import random
    import struct
    import zlib
    rnd = random.Random(seed)
    h, w, ch = samples.shape
    bits = ch * depth
    bpp = max(1, bits // 8)

    def pack(line):                                   # [pw, ch] -> bytes
        if depth == 16:
            return b"".join(struct.pack(">H", int(v)) for v in line.reshape(-1))
        if depth == 8:
            return bytes(int(v) for v in line.reshape(-1))
        out, acc, n = bytearray(), 0, 0
        for v in line.reshape(-1):
            acc, n = (acc << depth) | int(v), n + depth
            if n == 8:
                out.append(acc)
                acc, n = 0, 0
        if n:
            out.append(acc << (8 - n))
        return bytes(out)

    passes = ([(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
              if interlace else [(0, 0, 1, 1)])
    raw = bytearray()
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for line in sub:
            cur = pack(line)
            f = rnd.choice([0, 1, 2])
            if f == 0:
                enc = cur
            elif f == 1:
                enc = bytes((cur[i] - (cur[i - bpp] if i >= bpp else 0)) & 255 for i in range(len(cur)))
            else:
                enc = bytes((cur[i] - (prev[i] if prev else 0)) & 255 for i in range(len(cur)))
            raw += bytes([f]) + enc
            prev = cur

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette.reshape(-1).tolist()))
    return out + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")


@pytest.mark.parametrize("interlace", [0, 1])
@pytest.mark.parametrize("ctype,depth", [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (3, 1), (3, 2), (3, 4), (3, 8),
                                         (2, 8), (2, 16), (4, 8), (4, 16), (6, 8), (6, 16)])
def test_png_every_depth_colour_type_and_adam7(tmp_path, ctype, depth, interlace):
    """reconstruction.rs:95-105 decodes through the image crate: every PNG bit depth and colour type, Adam7 too,
    grey of 1 / 2 / 4 bits scaled to the full range, 16-bit samples rounded to 8 ((v + 128) / 257), alpha dropped."""
    from PIL import Image
    rng = np.random.default_rng(100 * ctype + depth + interlace)
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    for (h, w) in [(1, 1), (3, 9), (13, 11), (16, 24)]:
        palette = rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8) if ctype == 3 else None
        samples = rng.integers(0, 1 << depth, (h, w, ch))
        (tmp_path / "in.png").write_bytes(_encode_png(samples, depth, ctype, interlace, palette, seed=h * w))
        if ctype == 3:
            want = palette[samples[:, :, 0]]
        else:
            v = samples[:, :, :3] if ch >= 3 else np.repeat(samples[:, :, :1], 3, axis=2)
            want = (v + 128) // 257 if depth == 16 else v * (255 // ((1 << depth) - 1))
        r = subprocess.run([SELFTEST, "png", str(tmp_path / "in.png"), str(tmp_path / "out.png")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.asarray(Image.open(tmp_path / "out.png"))
        assert np.array_equal(got, want.astype(np.uint8)), (h, w)
        if depth <= 8:                                  # Pillow reads the same file to the same pixels
            assert np.array_equal(np.asarray(Image.open(tmp_path / "in.png").convert("RGB")), want.astype(np.uint8))


def test_png_depth_and_colour_type_must_agree(tmp_path):
    for ctype, depth in [(2, 4), (6, 2), (4, 1), (3, 16), (0, 3)]:
        (tmp_path / "bad.png").write_bytes(_encode_png(np.zeros((2, 2, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]), int), 8, ctype, 0,
                                                       np.zeros((2, 3), np.uint8) if ctype == 3 else None).replace(
            bytes([8, ctype, 0, 0, 0]), bytes([depth, ctype, 0, 0, 0]), 1))
        r = subprocess.run([SELFTEST, "png", str(tmp_path / "bad.png"), str(tmp_path / "o.png")], capture_output=True, text=True)
        assert r.returncode != 0 and "PNG" in r.stderr


@pytest.mark.parametrize("size", [(150, 100), (512, 384), (300, 200)])
def test_lanczos3_against_pillow(tmp_path, size):
    from PIL import Image
    rng = np.random.default_rng(6)
    big = Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)).resize((300, 200), Image.BICUBIC)
    src, dst = str(tmp_path / "big.png"), str(tmp_path / "rs.png")
    big.save(src)
    r = subprocess.run([SELFTEST, "resize", src, str(size[0]), str(size[1]), dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.asarray(Image.open(dst)).astype(int)
    want = np.asarray(big.resize(size, Image.LANCZOS)).astype(int)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1     # one code of rounding at most


@pytest.mark.parametrize("case", [((300, 200), (150, 100)), ((300, 200), (512, 384)), ((301, 199), (97, 333)),
                                  ((64, 48), (1536, 1536)), ((40, 30), (1, 1)), ((1, 7), (5, 3)), ((257, 129), (256, 128)),
                                  ((100, 100), (100, 100))])
def test_lanczos3_is_the_image_crates_sampler(tmp_path, case):
    """reconstruction.rs:107-113 / output.rs:133-137: resize_exact(.., Lanczos3) is the `image` crate's sampler (0.25.10,
    imageops/sample.rs: rows first into an f32 intermediate, then columns; weights in f32; round half away from zero).
    The C++ host layer against the C restatement of that algorithm in oracle/image_oracle.c: the same bytes."""
    import ctypes as C
    from PIL import Image
    from oracle import output_oracle
    output_oracle.build()
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libimage_oracle.so"))
    (w, h), (nw, nh) = case
    rng = np.random.default_rng(w * 7 + nh)
    img = _photo(w, h, w + h) if w > 8 and h > 8 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[rng.integers(0, h, 20), rng.integers(0, w, 20)] = rng.integers(0, 2, (20, 1), dtype=np.uint8) * 255   # hard edges: overshoot clamps
    src, dst = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    Image.fromarray(img).save(src)
    r = subprocess.run([SELFTEST, "resize", src, str(nw), str(nh), dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.asarray(Image.open(dst))
    want = np.empty((nh, nw, 3), np.uint8)
    a = np.ascontiguousarray(img)
    assert lib.oracle_resize_lanczos3_rgb8(C.c_void_p(a.ctypes.data), C.c_int64(w), C.c_int64(h), C.c_void_p(want.ctypes.data),
                                           C.c_int64(nw), C.c_int64(nh)) == 0
    assert got.shape == want.shape and np.array_equal(got, want)


def test_sixteen_bit_samples_round_like_the_image_crate():
    """into_rgb8 of a 16-bit picture: (v + 128) / 257 per sample (image 0.25 color.rs); the oracle's one-liner against
    the closed form over all 65536 values, and the PNG decoder applies it in test_png_every_depth_colour_type_and_adam7"""
    import ctypes as C
    from oracle import output_oracle
    output_oracle.build()
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libimage_oracle.so"))
    lib.oracle_u16_to_u8.restype = C.c_uint8
    v = np.arange(65536)
    want = np.rint(v / 257.0 - 1e-9).astype(int)                # nearest of v * 255 / 65535, ties cannot occur
    got = np.array([lib.oracle_u16_to_u8(C.c_uint16(int(x))) for x in v[::17]])
    assert np.array_equal(got, ((v[::17] + 128) // 257)) and np.abs(got - want[::17]).max() == 0


def _photo(w, h, seed):
    """a smooth, noisy, colourful picture: what JPEG is made for"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 100 * np.sin(x / 17.0 + y / 29.0), 127 + 100 * np.cos(x / 11.0 - y / 23.0),
                  127 + 90 * np.sin((x + y) / 31.0)], -1) + rng.normal(0, 6, (h, w, 3))
    return np.clip(a, 0, 255).astype(np.uint8)


def _decode(src, dst, oriented=False):
    r = subprocess.run([SELFTEST, "decode", src, dst] + (["oriented"] if oriented else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from PIL import Image
    return np.asarray(Image.open(dst)).astype(int), [int(v) for v in r.stdout.split()]


@pytest.mark.parametrize("progressive", [False, True])
@pytest.mark.parametrize("subsampling", [0, 1, 2])          # 4:4:4, 4:2:2, 4:2:0
@pytest.mark.parametrize("size", [(123, 77), (64, 48), (1, 19), (3, 2)])
def test_jpeg_decoder_against_pillow(tmp_path, size, subsampling, progressive):
    """reconstruction.rs:96,106: the source photo is usually a JPEG.  Two conforming decoders differ by IDCT
    and colour rounding: within 3 codes per sample and 0.1 on average of libjpeg-turbo (Pillow)."""
    from PIL import Image
    for k, quality in enumerate((95, 55)):
        src, dst = str(tmp_path / f"in{k}.jpg"), str(tmp_path / f"out{k}.ppm")
        Image.fromarray(_photo(*size, seed=3)).save(src, quality=quality, subsampling=subsampling,
                                                     progressive=progressive, optimize=bool(k))
        got, meta = _decode(src, dst)
        want = np.asarray(Image.open(src).convert("RGB")).astype(int)
        assert got.shape == want.shape and meta == [1, -1, size[0], size[1]]
        d = np.abs(got - want)
        assert d.max() <= 3 and (d.mean() < 0.1 or d.size < 3000), (d.max(), d.mean())   # a few pixels have no average


def test_jpeg_grey_restart_intervals_and_errors(tmp_path):
    from PIL import Image
    src, dst = str(tmp_path / "g.jpg"), str(tmp_path / "g.ppm")
    Image.fromarray(_photo(100, 60, seed=4)).convert("L").save(src, quality=90)
    got, _ = _decode(src, dst)
    assert np.abs(got - np.asarray(Image.open(src).convert("RGB")).astype(int)).max() <= 1
    # restart markers every 2 MCU rows / every 5 MCUs (libjpeg's `restart_marker_rows / _blocks`)
    for kw in ({"restart_marker_rows": 2}, {"restart_marker_blocks": 5}):
        try:
            Image.fromarray(_photo(203, 111, seed=5)).save(src, quality=85, **kw)
        except TypeError:      # an older Pillow without the option
            pytest.skip("Pillow cannot write restart markers")
        assert b"\xff\xdd" in open(src, "rb").read()
        got, _ = _decode(src, dst)
        assert np.abs(got - np.asarray(Image.open(src).convert("RGB")).astype(int)).max() <= 3
    cmyk = str(tmp_path / "cmyk.jpg")
    Image.fromarray(_photo(32, 32, seed=6)).convert("CMYK").save(cmyk)
    r = subprocess.run([SELFTEST, "decode", cmyk, dst], capture_output=True, text=True)
    assert r.returncode == 1 and "4 components" in r.stderr
    cut = str(tmp_path / "cut.jpg")
    open(cut, "wb").write(open(src, "rb").read()[:300])
    r = subprocess.run([SELFTEST, "decode", cut, dst], capture_output=True, text=True)
    assert r.returncode in (0, 1)          # a truncated scan decodes to grey or is refused: never a crash


def test_image_decoders_survive_corrupt_files(tmp_path):
    """Byte flips, deletions and truncations of JPEG / PNG files (with EXIF blocks): the decoders either decode
    or report an ImageError -- exit code 0 or 1, never a signal.  (Run under ASan/UBSan during development.)"""
    import random
    from PIL import Image
    rnd = random.Random(1)
    img = Image.fromarray(_photo(56, 40, seed=8))
    exif = Image.Exif()
    exif[0x0112] = 6
    exif.get_ifd(0x8769)[0xA405] = 28
    img.save(tmp_path / "s0.jpg", quality=90, subsampling=2, exif=exif)
    img.save(tmp_path / "s1.jpg", quality=80, subsampling=1, progressive=True)
    img.save(tmp_path / "s2.png", exif=exif)
    img.convert("P").save(tmp_path / "s3.png")
    seeds = [(tmp_path / n).read_bytes() for n in ("s0.jpg", "s1.jpg", "s2.png", "s3.png")]
    codes = set()
    for it in range(240):
        b = bytearray(seeds[it % 4])
        mode = rnd.random()
        if mode < 0.7:
            for _ in range(rnd.choice([1, 1, 2, 4, 8])):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif mode < 0.85:
            i = rnd.randrange(len(b))
            del b[i:i + rnd.randrange(1, 40)]
        else:
            b = b[:rnd.randrange(4, len(b))]
        (tmp_path / "f.bin").write_bytes(bytes(b))
        r = subprocess.run([SELFTEST, "decode", str(tmp_path / "f.bin"), str(tmp_path / "o.ppm"), "oriented"],
                           capture_output=True, timeout=60)
        assert r.returncode in (0, 1), (it, r.returncode, r.stderr[-300:])
        codes.add(r.returncode)
    assert codes == {0, 1}          # both outcomes occur: the corpus does exercise the error paths


@pytest.mark.parametrize("orientation", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("fmt", ["jpg", "png"])
def test_exif_orientation_and_focal_length(tmp_path, orientation, fmt):
    """reconstruction.rs:97-105,133-143: FocalLengthIn35mmFilm from the Exif IFD, the orientation applied to
    the decoded pixels -- against Pillow's exif_transpose."""
    from PIL import Image, ImageOps
    img = Image.fromarray(_photo(48, 32, seed=7))
    exif = Image.Exif()
    exif[0x0112] = orientation
    exif.get_ifd(0x8769)[0xA405] = 28            # FocalLengthIn35mmFilm lives in the Exif sub-IFD
    src, dst = str(tmp_path / f"o.{fmt}"), str(tmp_path / "o.ppm")
    img.save(src, exif=exif, **({"quality": 95, "subsampling": 0} if fmt == "jpg" else {}))
    got, meta = _decode(src, dst, oriented=True)
    want = np.asarray(ImageOps.exif_transpose(Image.open(src)).convert("RGB")).astype(int)
    assert meta[:2] == [orientation, 28] and meta[2:] == [want.shape[1], want.shape[0]]
    assert got.shape == want.shape and np.abs(got - want).max() <= (3 if fmt == "jpg" else 0)


def test_cli_usage_and_exit_codes(tmp_path):
    r = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage: matrix-eyes [OPTIONS] <IMG_SRC>... <IMG_OUT>" in r.stdout
    from matrix_eyes_amd.cli import USAGE_INSTRUCTIONS
    assert USAGE_INSTRUCTIONS in r.stdout                       # the same text as the Python twin
    for argv, msg in [([], "No source image provided"), (["a.png"], "No output image provided"),
                      (["--mesh=cubes", "a.png", "b.obj"], "Unsupported mesh vertex output mode cubes"),
                      (["--focal-length", "a.png", "b.png"], "Option flag --focal-length has no value"),
                      (["--resize-scale=big", "a.png", "b.png"], "Argument --resize-scale has an unsupported value big"),
                      (["a.png", "b.png", "c.png"], "Unexpected argument c.png")]:
        r = subprocess.run([CLI] + argv, capture_output=True, text=True)
        assert r.returncode == 2 and msg in r.stderr, (argv, r.stderr)


@pytest.mark.gpu
def test_cli_end_to_end_against_the_python_mirror(tmp_path):
    """depth map PNG, stereogram PNG and OBJ from the compiled CLI == the Python mirror on the same inputs"""
    from PIL import Image
    import matrix_eyes_amd as m
    from matrix_eyes_amd.synthetic import synthetic_images
    cfg = m.ModelConfig.tiny()
    ckpt = str(tmp_path / "tiny.pt")
    _tiny_checkpoint(ckpt)
    S = cfg.img_size
    src = str(tmp_path / "photo.png")
    Image.fromarray(synthetic_images(1, S, "structured", seed=11)[0]).save(src)
    env = dict(os.environ, MATRIX_EYES_MODEL="tiny", MATRIX_EYES_SEED="7")

    def cli(*args):
        r = subprocess.run([CLI, f"--checkpoint-path={ckpt}", "--focal-length=35", *args], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.startswith("Matrix Eyes version")

    loader = m.DepthProModelLoader(ckpt, False, cfg)
    from matrix_eyes_amd import reconstruction as R

    cli(src, str(tmp_path / "depth_cpp.png"))
    R.extract_depth(0, loader, src, str(tmp_path / "depth_py.png"), 35.0, m.ImageOutputFormat.DepthMap(),
                    m.VertexMode.Color)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "depth_cpp.png")),
                          np.asarray(Image.open(tmp_path / "depth_py.png")))

    cli("--mesh=texture-coordinates", src, str(tmp_path / "mesh_cpp.obj"))
    R.extract_depth(0, loader, src, str(tmp_path / "mesh_py.obj"), 35.0, m.ImageOutputFormat.DepthMap(),
                    m.VertexMode.Texture)
    cpp, py = (tmp_path / "mesh_cpp.obj").read_text(), (tmp_path / "mesh_py.obj").read_text()
    assert cpp.replace("mesh_cpp", "mesh") == py.replace("mesh_py", "mesh") and len(cpp) > 1000

    cli("--image-output-format=stereogram", "--resize-scale=0.5", src, str(tmp_path / "stereo_cpp.png"))
    st = np.asarray(Image.open(tmp_path / "stereo_cpp.png"))
    assert st.shape == (S // 2, S // 2, 3)
    # same seed, same picture; the autostereogram property: rows repeat with the pattern period where flat
    cli("--image-output-format=stereogram", "--resize-scale=0.5", src, str(tmp_path / "stereo_cpp2.png"))
    assert np.array_equal(st, np.asarray(Image.open(tmp_path / "stereo_cpp2.png")))

    r = subprocess.run([CLI, f"--checkpoint-path={tmp_path / 'none.pt'}", src, str(tmp_path / "x.png")], env=env,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Reconstruction failed" in r.stdout


@pytest.mark.gpu
def test_cli_jpeg_photo_with_exif(tmp_path):
    """A rotated JPEG whose EXIF block carries the orientation and the 35 mm focal length: the CLI reads both
    itself (no --focal-length), as reconstruction.rs:97-106 does.  The Python mirror decodes the same file with
    libjpeg-turbo, whose samples differ by a code or two, so the depth maps agree closely, not bit for bit."""
    from PIL import Image
    import matrix_eyes_amd as m
    from matrix_eyes_amd import reconstruction as R
    cfg = m.ModelConfig.tiny()
    ckpt = str(tmp_path / "tiny.pt")
    _tiny_checkpoint(ckpt)
    exif = Image.Exif()
    exif[0x0112] = 6                                  # rotate 90 clockwise on display
    exif.get_ifd(0x8769)[0xA405] = 50
    src = str(tmp_path / "photo.jpg")
    Image.fromarray(_photo(300, 200, seed=9)).save(src, quality=92, exif=exif)
    env = dict(os.environ, MATRIX_EYES_MODEL="tiny")
    r = subprocess.run([CLI, f"--checkpoint-path={ckpt}", src, str(tmp_path / "d_cpp.png")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    loader = m.DepthProModelLoader(ckpt, False, cfg)
    R.extract_depth(0, loader, src, str(tmp_path / "d_py.png"), None, m.ImageOutputFormat.DepthMap(), m.VertexMode.Color)
    a = np.asarray(Image.open(tmp_path / "d_cpp.png")).astype(int)
    b = np.asarray(Image.open(tmp_path / "d_py.png")).astype(int)
    assert a.shape == b.shape == (300, 200, 3)        # portrait after the rotation (height 300, width 200)
    assert np.abs(a - b).mean() < 2.0
