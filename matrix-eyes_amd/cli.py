"""Command-line twin of reference src/main.rs: the same `--name=value` flags,
defaults and exit codes, driving the HIP back end through the C ABI.

    python -m matrix_eyes_amd [OPTIONS] <IMG_SRC> <IMG_OUT>
"""
import sys
from dataclasses import dataclass
from typing import List, Optional

from .output import ImageOutputFormat, VertexMode

USAGE_INSTRUCTIONS = """Usage: matrix-eyes [OPTIONS] <IMG_SRC>... <IMG_OUT>

Arguments:
  <IMG_SRC>...  Source image
  <IMG_OUT>     Output image

Options:
      --focal-length=<FOCAL_LENGTH>       Focal length in 35mm equivalent
      --checkpoint-path=<CHECKPOINT_PATH> Path to checkpoint file [default: ./checkpoints/depth_pro.pt]
      --image-output-format=<FORMAT>      Format for output [default: depthmap] [possible values: depthmap, stereogram]
      --resize-scale=<SCALE>              Custom scale for stereogram output [default: 1.0]
      --stereo-amplitude=<AMPLITUDE>      Custom scale for stereogram output [default: 0.0625]
      --mesh=<MESH>                       Mesh options [default: vertex-colors] [possible values: plain, vertex-colors, texture-coordinates]
      --convert-checkpoints               Convert checkpoints into a more efficient format [default: disabled]
      --help                              Print help"""


class UsageExit(SystemExit):
    pass


@dataclass
class Args:                                  # main.rs:12-20
    focal_length: Optional[float] = None
    checkpoint_path: str = "./checkpoints/depth_pro.pt"
    convert_checkpoints: bool = False
    output_format: ImageOutputFormat = None
    vertex_mode: VertexMode = VertexMode.Color   # main.rs:43 (the code's default; quirk Q5)
    img_src: str = ""
    img_out: str = ""


def _fail(message: str):
    print(message, file=sys.stderr)
    print(USAGE_INSTRUCTIONS)
    raise UsageExit(2)


def _parse_float(name: str, value: str) -> float:
    try:
        return float(value)
    except ValueError as err:
        _fail(f"Argument {name} has an unsupported value {value}: {err}")


def parse(argv: List[str]) -> Args:
    """main.rs:37-146"""
    args = Args(output_format=ImageOutputFormat.DepthMap())
    resize_scale, stereo_amplitude, stereogram = None, 1.0 / 16.0, False
    for arg in argv:
        if arg.startswith("--") and not args.img_src and not args.img_out:
            if arg == "--convert-checkpoints":
                args.convert_checkpoints = True
                continue
            if arg == "--help":
                print(USAGE_INSTRUCTIONS)
                raise UsageExit(0)
            if "=" not in arg:
                _fail(f"Option flag {arg} has no value")
            name, value = arg.split("=", 1)
            if name == "--focal-length":
                args.focal_length = _parse_float(name, value)
            elif name == "--image-output-format":
                if value.lower() == "depthmap":
                    stereogram = False
                elif value.lower() == "stereogram":
                    stereogram = True
                else:
                    _fail(f"Unsupported output format {value}")
            elif name == "--resize-scale":
                resize_scale = _parse_float(name, value)
            elif name == "--stereo-amplitude":
                stereo_amplitude = _parse_float(name, value)
            elif name == "--mesh":
                modes = {"plain": VertexMode.Plain, "vertex-colors": VertexMode.Color,
                         "texture-coordinates": VertexMode.Texture}
                if value.lower() not in modes:
                    _fail(f"Unsupported mesh vertex output mode {value}")
                args.vertex_mode = modes[value.lower()]
            elif name == "--checkpoint-path":
                args.checkpoint_path = value
            else:
                print(f"Unsupported argument {arg}", file=sys.stderr)   # main.rs:128: not fatal
        elif not args.img_src:
            args.img_src = arg
        elif not args.img_out:
            args.img_out = arg
        else:
            _fail(f"Unexpected argument {arg}")
    args.output_format = (ImageOutputFormat.Stereogram(resize_scale, stereo_amplitude) if stereogram
                          else ImageOutputFormat.DepthMap())
    if not args.img_src:
        _fail("No source image provided")
    if not args.img_out:
        _fail("No output image provided")
    return args


def main(argv: Optional[List[str]] = None) -> int:
    """main.rs:149-173"""
    from . import __version__
    print(f"Matrix Eyes (HIP back end) version {__version__}")
    try:
        args = parse(sys.argv[1:] if argv is None else argv)
    except UsageExit as e:
        return int(e.code)
    from .depth_pro import DepthProModelLoader
    from .reconstruction import extract_depth
    loader = DepthProModelLoader(args.checkpoint_path, args.convert_checkpoints)
    try:
        extract_depth(0, loader, args.img_src, args.img_out, args.focal_length, args.output_format,
                      args.vertex_mode)
    except Exception as err:
        print(f"Reconstruction failed: {err}")
        return 1
    return 0
