"""Host-side mirror of reference src/reconstruction.rs: image front end and the
top-level `extract_depth`.  File decoding, EXIF and Lanczos resampling are Pillow's (the reference
uses the `image` and `kamadak-exif` crates); at the native 1536x1536 size the resize is the identity.
The u8 -> float normalisation and HWC -> CHW (reconstruction.rs:114-124) run on the GPU
(`me_extract_depth_u8`)."""
import math
import os
import sys
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .depth_pro import IMG_SIZE, DepthProModelLoader
from .output import DepthMap, ImageOutputFormat, VertexMode

_EXIF_IFD = 0x8769
_FOCAL_LENGTH_35MM = 0xA405   # exif::Tag::FocalLengthIn35mmFilm (reconstruction.rs:136-137)


class ReconstructionError(RuntimeError):   # reconstruction.rs:240-249
    pass


@dataclass
class SourceImage:                          # reconstruction.rs:74-81
    rgb8: np.ndarray                        # u8 [IMG_SIZE, IMG_SIZE, 3]
    original_size: Tuple[int, int]          # (width, height) after orientation
    focal_length_35mm: Optional[float]

    @staticmethod
    def load(path: str, focal_length_35mm: Optional[float] = None, size: int = IMG_SIZE) -> "SourceImage":
        """reconstruction.rs:87-131: decode, EXIF focal length, orientation, Lanczos3 to size x size"""
        from PIL import Image, ImageOps
        try:
            img = Image.open(path)
            img.load()
        except Exception as err:            # ImageError / io::Error
            raise ReconstructionError(f"Failed to load source image: {err}") from err
        if focal_length_35mm is None:
            focal_length_35mm = SourceImage.get_focal_length_35mm(img)
        img = ImageOps.exif_transpose(img)  # decoder.orientation() + apply_orientation (:103-105)
        original_size = img.size
        img = img.convert("RGB")
        if img.size != (size, size):        # resize_exact(.., Lanczos3) (:107-113)
            img = img.resize((size, size), Image.LANCZOS)
        return SourceImage(np.ascontiguousarray(np.asarray(img, dtype=np.uint8)), original_size,
                           focal_length_35mm)

    @staticmethod
    def get_focal_length_35mm(img) -> Optional[float]:
        """reconstruction.rs:133-143"""
        try:
            exif = img.getexif()
            value = exif.get_ifd(_EXIF_IFD).get(_FOCAL_LENGTH_35MM, exif.get(_FOCAL_LENGTH_35MM))
        except Exception:
            return None
        return float(value) if value is not None else None     # the reference keeps Some(0) (reconstruction.rs:136-143)

    def focal_length_px(self) -> Optional[float]:
        """reconstruction.rs:145-152: f_img / f_35mm == diagonal / diagonal(24 mm x 36 mm)"""
        if self.focal_length_35mm is None:
            return None
        diagonal_35mm = math.sqrt(24.0 * 24.0 + 36.0 * 36.0)
        w, h = float(self.original_size[0]), float(self.original_size[1])
        return float(self.focal_length_35mm) * math.sqrt(w * w + h * h) / diagonal_35mm


def extract_depth(device: int, model_loader: DepthProModelLoader, source_path: str, destination_path: str,
                  focal_length_35mm: Optional[float], image_format: ImageOutputFormat,
                  vertex_mode: VertexMode, progress=None, noise=None) -> None:
    """reconstruction.rs:155-205"""
    try:
        img = SourceImage.load(source_path, focal_length_35mm, model_loader.cfg.img_size)
    except ReconstructionError as err:
        print(err, file=sys.stderr)
        raise
    f_px = img.focal_length_px()
    f_norm = None if f_px is None else float(np.float32(f_px / float(img.original_size[0])))   # :174-176
    ctx = model_loader.context(device, os.environ.get("MATRIX_EYES_DTYPE", "f16"))   # f16 | bf16 | fp8, as the C++ twin
    ctx.set_progress(progress)
    try:
        inverse_depth = ctx.extract_depth(img.rgb8[None], f_norm)[0]
    except Exception as err:
        print(f"Failed to process image: {err}", file=sys.stderr)
        raise
    finally:
        ctx.set_progress(None)
    depth_map = DepthMap(ctx, inverse_depth, img.original_size)
    try:
        depth_map.output_image(destination_path, source_path, image_format, vertex_mode, noise=noise)
    except Exception as err:
        print(f"Failed to output result: {err}", file=sys.stderr)
        raise
