"""Host-side mirror of reference src/output.rs over the C ABI: DepthMap::new (output.rs:44-67),
output_image (:100-121), output_depth_map (:123-139), output_stereogram (:141-193),
output_mesh (:195-261)."""
import ctypes as C
import enum
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from .depth_pro import Context, _in_ptr


class VertexMode(enum.IntEnum):   # output.rs:33-38
    Plain = 0
    Color = 1
    Texture = 2


@dataclass
class ImageOutputFormat:           # output.rs:27-31
    kind: str = "depthmap"         # "depthmap" | "stereogram"
    resize_scale: Optional[float] = None
    amplitude: float = 1.0 / 16.0  # main.rs default --stereo-amplitude

    @staticmethod
    def DepthMap():
        return ImageOutputFormat("depthmap")

    @staticmethod
    def Stereogram(resize_scale: Optional[float], amplitude: float):
        return ImageOutputFormat("stereogram", resize_scale, amplitude)


def _f32_round(x: float) -> int:
    """f32::round (half away from zero, exact) of an f32 value, then `as u32` (saturating, NaN -> 0)."""
    x = np.float32(x)
    if not np.isfinite(x):
        return 0 if (np.isnan(x) or x < 0) else 4294967295
    t = np.trunc(x)
    if abs(x - t) >= np.float32(0.5):      # x - trunc(x) is exact in f32
        t += np.sign(x)
    if t <= 0:
        return 0
    return int(min(float(t), 4294967295.0))


class DepthMap:
    def __init__(self, ctx: Context, inverse_depth, original_size):
        """DepthMap::new: clamp to [1/250, 1/0.1] (output.rs:51-57).  inverse_depth [rows, cols]."""
        self.ctx = ctx
        data = np.array(inverse_depth, dtype=np.float32, copy=True)
        if data.ndim != 2:
            raise L.MatrixEyesError(2, "DepthMap: inverse_depth must be 2-D")
        self.data_width, self.data_height = data.shape   # output.rs:52 (names as in the reference)
        mn, mx = C.c_float(), C.c_float()
        ctx._check(ctx.lib.me_depth_clamp_minmax(ctx.handle, C.c_void_p(data.ctypes.data), data.size,
                                                 C.byref(mn), C.byref(mx)))
        self.data = data
        self._range = (mn.value, mx.value)
        self.original_width, self.original_height = original_size

    def inverse_depth_range(self):   # output.rs:69-75
        return self._range

    # -- raster products (arrays; files below)
    def depth_map_rgb(self) -> np.ndarray:
        """output.rs:123-131 before the Lanczos resize -> u8 [data_width, data_height, 3]"""
        mn, mx = self._range
        out = np.empty(self.data.shape + (3,), np.uint8)
        self.ctx._check(self.ctx.lib.me_depthmap_rgb(self.ctx.handle, C.c_void_p(self.data.ctypes.data),
                                                     self.data.size, mn, mx, C.c_void_p(out.ctypes.data)))
        return out

    def stereogram_size(self, resize_scale: Optional[float]):
        if resize_scale is not None:   # output.rs:147-151, f32 arithmetic
            w = _f32_round(np.float32(self.original_width) * np.float32(resize_scale))
            h = _f32_round(np.float32(self.original_height) * np.float32(resize_scale))
            return w, h
        return self.original_width, self.original_height

    def stereogram(self, resize_scale: Optional[float], amplitude: float, noise=None) -> np.ndarray:
        """output.rs:141-193 -> u8 [out_h, out_w, 3].  noise u8 [out_h, out_w, 3] stands in for the
        reference's rand::rng() stream (row by row, pixel by pixel); drawn from os.urandom-seeded
        numpy when omitted."""
        w, h = self.stereogram_size(resize_scale)
        if noise is None:
            noise = np.random.default_rng().integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        pn, keep = _in_ptr(noise, np.uint8)
        if tuple(noise.shape) != (h, w, 3):
            raise L.MatrixEyesError(2, f"noise must be [{h},{w},3]")
        mn, mx = self._range
        out = np.empty((h, w, 3), np.uint8)
        self.ctx._check(self.ctx.lib.me_stereogram(
            self.ctx.handle, C.c_void_p(self.data.ctypes.data), self.data_width, self.data_height, mn, mx,
            w, h, amplitude, pn, C.c_void_p(out.ctypes.data)))
        return out

    def mesh_index(self, want_faces=True):
        """IndexedMesh::new + remap_face (output.rs:264-363) -> (vertex_index [n], nverts, faces)"""
        w, h = self.data_width, self.data_height
        vi = np.empty((w * h,), np.int32)
        faces = np.empty((2 * (w - 1) * (h - 1), 3), np.int32) if want_faces else None
        nv, nf = C.c_int64(), C.c_int64()
        self.ctx._check(self.ctx.lib.me_mesh_index(
            self.ctx.handle, C.c_void_p(self.data.ctypes.data), w, h, C.c_void_p(vi.ctypes.data),
            C.byref(nv), C.byref(nf), C.c_void_p(faces.ctypes.data) if want_faces else None))
        return vi, nv.value, (faces[:nf.value] if want_faces else nf.value)

    def mesh_vertices(self, vertex_index, nverts):
        """output.rs:228-249 -> (uv [n,2], xyz [n,3]) in vertex-id order"""
        uv = np.empty((nverts, 2), np.float32)
        xyz = np.empty((nverts, 3), np.float32)
        pv, keep = _in_ptr(vertex_index, np.int32)
        self.ctx._check(self.ctx.lib.me_mesh_vertices(
            self.ctx.handle, C.c_void_p(self.data.ctypes.data), self.data_width, self.data_height, pv,
            nverts, self.original_width, self.original_height, C.c_void_p(uv.ctypes.data),
            C.c_void_p(xyz.ctypes.data)))
        return uv, xyz

    # -- files
    def output_image(self, destination_path: str, source_path: str, image_format: ImageOutputFormat,
                     vertex_mode: VertexMode, noise=None):
        """output.rs:100-121: dispatch on the destination suffix."""
        low = destination_path.lower()
        if low.endswith(".ply") or low.endswith(".obj"):
            return self.output_mesh(destination_path, source_path, vertex_mode)
        from PIL import Image
        if image_format.kind == "depthmap":
            img = Image.fromarray(self.depth_map_rgb())
            size = (self.original_width, self.original_height)
            if img.size != size:   # output.rs:133-137 (identity at the native size)
                img = img.resize(size, Image.LANCZOS)
            img.save(destination_path)
        else:
            Image.fromarray(self.stereogram(image_format.resize_scale, image_format.amplitude,
                                            noise)).save(destination_path)

    def output_mesh(self, destination_path: str, source_path: str, vertex_mode: VertexMode):
        """output.rs:195-261 with ObjWriter / PlyWriter."""
        colors = None
        if vertex_mode == VertexMode.Color:   # output.rs:206-218
            from PIL import Image
            img = Image.open(source_path).convert("RGB").resize(
                (self.data_width, self.data_height), Image.LANCZOS)
            colors = np.ascontiguousarray(np.asarray(img, dtype=np.uint8))
        self.ctx._check(self.ctx.lib.me_output_mesh(
            self.ctx.handle, C.c_void_p(self.data.ctypes.data), self.data_width, self.data_height,
            self.original_width, self.original_height, destination_path.encode(), source_path.encode(),
            int(vertex_mode), C.c_void_p(colors.ctypes.data) if colors is not None else None))


class DeviceDepthMap:
    """DepthMap::new -> output_image chained on the GPU (BASELINE configs[4]: depth -> stereogram / depth map
    without the host round trips of the reference's readback, output.rs:44-67).  `inverse_depth` is a CUDA f32
    tensor [rows, cols] (e.g. the `out=` tensor of Context.extract_depth); it is clamped IN PLACE on the context's
    stream and its range stays in device memory for the raster kernels queued behind it."""

    def __init__(self, ctx: Context, inverse_depth, original_size):
        import torch
        if not (inverse_depth.is_cuda and inverse_depth.dtype == torch.float32 and inverse_depth.dim() == 2
                and inverse_depth.is_contiguous()):
            raise L.MatrixEyesError(2, "DeviceDepthMap: a contiguous CUDA f32 [rows, cols] tensor is required")
        self.ctx, self.data = ctx, inverse_depth
        self.data_width, self.data_height = inverse_depth.shape
        self.original_width, self.original_height = original_size
        self.range_dev = torch.empty(2, dtype=torch.float32, device=inverse_depth.device)
        ctx._check(ctx.lib.me_depth_clamp_minmax_async(ctx.handle, C.c_void_p(inverse_depth.data_ptr()),
                                                       inverse_depth.numel(), C.c_void_p(self.range_dev.data_ptr())))

    def inverse_depth_range(self):
        self.ctx.synchronize()
        mn, mx = self.range_dev.tolist()
        return mn, mx

    def depth_map_rgb(self, out=None):
        import torch
        if out is None:
            out = torch.empty(self.data.shape + (3,), dtype=torch.uint8, device=self.data.device)
        self.ctx._check(self.ctx.lib.me_depthmap_rgb_dev_range(
            self.ctx.handle, C.c_void_p(self.data.data_ptr()), self.data.numel(), C.c_void_p(self.range_dev.data_ptr()),
            C.c_void_p(out.data_ptr())))
        return out

    def output_mesh(self, destination_path: str, source_path: str, vertex_mode: VertexMode = VertexMode.Texture,
                    colors=None):
        """output.rs:195-261 from the device-resident depth: mesh indexing, vertex coordinates and (OBJ) the text
        itself are produced on the GPU; only the finished file bytes cross to the host."""
        self.ctx._check(self.ctx.lib.me_output_mesh(
            self.ctx.handle, C.c_void_p(self.data.data_ptr()), self.data_width, self.data_height,
            self.original_width, self.original_height, destination_path.encode(), source_path.encode(),
            int(vertex_mode), C.c_void_p(colors.data_ptr()) if colors is not None else None))

    def obj_text(self, stem: str = "mesh", vertex_mode: VertexMode = VertexMode.Texture, colors=None):
        """The OBJ file's bytes as a CUDA uint8 tensor (me_mesh_obj_text: the text formatted on the GPU; a copy of the
        context-owned buffer)."""
        import torch

        class _DevMem:
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        ptr, n = C.c_void_p(), C.c_int64()
        self.ctx._check(self.ctx.lib.me_mesh_obj_text(
            self.ctx.handle, C.c_void_p(self.data.data_ptr()), self.data_width, self.data_height, self.original_width,
            self.original_height, stem.encode(), int(vertex_mode),
            C.c_void_p(colors.data_ptr()) if colors is not None else None, C.byref(ptr), C.byref(n)))
        return torch.as_tensor(_DevMem(int(ptr.value), int(n.value)), device=self.data.device).clone()

    def stereogram(self, amplitude: float, noise, out=None):
        """noise: CUDA u8 [out_h, out_w, 3]"""
        import torch
        h, w = noise.shape[0], noise.shape[1]
        if out is None:
            out = torch.empty((h, w, 3), dtype=torch.uint8, device=self.data.device)
        self.ctx._check(self.ctx.lib.me_stereogram_dev_range(
            self.ctx.handle, C.c_void_p(self.data.data_ptr()), self.data_width, self.data_height,
            C.c_void_p(self.range_dev.data_ptr()), w, h, amplitude, C.c_void_p(noise.data_ptr()),
            C.c_void_p(out.data_ptr())))
        return out
