"""CPU tests of the 'next' rows: the image front end (reconstruction.rs:87-152) and the
CLI twin (main.rs:22-146)."""
import math

import numpy as np
import pytest

import matrix_eyes_amd as m
from matrix_eyes_amd import cli


def test_cli_defaults_and_flags():
    a = cli.parse(["in.jpg", "out.png"])
    assert (a.checkpoint_path, a.convert_checkpoints, a.focal_length) == ("./checkpoints/depth_pro.pt", False, None)
    assert a.output_format.kind == "depthmap" and a.vertex_mode == m.VertexMode.Color     # main.rs:41-43
    a = cli.parse(["--focal-length=28", "--image-output-format=Stereogram", "--stereo-amplitude=0.125",
                   "--resize-scale=0.5", "--mesh=texture-coordinates", "--checkpoint-path=x.pt",
                   "--convert-checkpoints", "a.jpg", "b.obj"])
    assert a.focal_length == 28.0 and a.checkpoint_path == "x.pt" and a.convert_checkpoints
    assert a.output_format.kind == "stereogram" and a.output_format.amplitude == 0.125
    assert a.output_format.resize_scale == 0.5 and a.vertex_mode == m.VertexMode.Texture
    # flags after the first positional argument are positional (main.rs:51)
    a = cli.parse(["a.jpg", "--mesh=plain"])
    assert a.img_out == "--mesh=plain"
    # the stereogram parameters apply wherever they appear (main.rs:131-133)
    a = cli.parse(["--image-output-format=stereogram", "--stereo-amplitude=0.25", "a", "b"])
    assert a.output_format.amplitude == 0.25


@pytest.mark.parametrize("argv,code", [
    (["--help"], 0), ([], 2), (["only-src"], 2), (["--focal-length"], 2), (["--focal-length=abc", "a", "b"], 2),
    (["--image-output-format=jpeg", "a", "b"], 2), (["--mesh=wire", "a", "b"], 2), (["a", "b", "c"], 2),
])
def test_cli_exit_codes(argv, code, capsys):
    with pytest.raises(SystemExit) as e:
        cli.parse(argv)
    assert e.value.code == code
    assert "Usage: matrix-eyes" in capsys.readouterr().out


def test_cli_unknown_flag_is_not_fatal(capsys):
    a = cli.parse(["--bogus=1", "a", "b"])          # main.rs:127-129 only prints
    assert a.img_src == "a" and "Unsupported argument --bogus=1" in capsys.readouterr().err


def test_cli_main_reports_failure_with_exit_code_1(tmp_path, capsys):
    rc = cli.main([str(tmp_path / "missing.jpg"), str(tmp_path / "out.png")])
    assert rc == 1 and "Reconstruction failed" in capsys.readouterr().out


def test_source_image_load_and_focal_length(tmp_path):
    from PIL import Image
    rgb = np.random.default_rng(0).integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
    img = Image.fromarray(rgb)
    exif = Image.Exif()
    exif[0x0112] = 6                       # Orientation: rotate 90 CW
    ifd = exif.get_ifd(0x8769)
    ifd[0xA405] = 35                       # FocalLengthIn35mmFilm
    path = tmp_path / "photo.jpg"
    img.save(path, exif=exif)
    src = m.SourceImage.load(str(path), None, size=32)
    assert src.rgb8.shape == (32, 32, 3) and src.rgb8.dtype == np.uint8
    assert src.original_size == (48, 64)   # (width, height) after applying the orientation
    assert src.focal_length_35mm == 35.0
    diag35 = math.sqrt(24 * 24 + 36 * 36)
    assert src.focal_length_px() == pytest.approx(35.0 * math.sqrt(48 * 48 + 64 * 64) / diag35)
    # an explicit focal length overrides EXIF (reconstruction.rs:96-97)
    assert m.SourceImage.load(str(path), 50.0, size=32).focal_length_35mm == 50.0
    # a native-size image is passed through untouched
    big = np.random.default_rng(1).integers(0, 256, size=(32, 32, 3), dtype=np.uint8)
    Image.fromarray(big).save(tmp_path / "native.png")
    s2 = m.SourceImage.load(str(tmp_path / "native.png"), None, size=32)
    assert np.array_equal(s2.rgb8, big) and s2.focal_length_35mm is None and s2.focal_length_px() is None
    with pytest.raises(m.ReconstructionError):
        m.SourceImage.load(str(tmp_path / "nope.png"))
