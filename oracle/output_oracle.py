"""ctypes wrapper of oracle/output_oracle.c plus a pure-Python OBJ/PLY formatter that follows
reference src/output.rs:385-630 — TEST INFRASTRUCTURE, NOT THE PRODUCT (see output_oracle.c)."""
import ctypes as C
import os
import struct
import subprocess
from decimal import Decimal

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "_build", "liboutput_oracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboutput_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_stereogram.restype = C.c_int
    return _LIB


def _p(a):
    return C.c_void_p(a.ctypes.data)


def clamp_minmax(depth):
    d = np.array(depth, dtype=np.float32, copy=True)
    mn, mx = C.c_float(), C.c_float()
    lib().oracle_clamp_minmax(_p(d), C.c_int64(d.size), C.byref(mn), C.byref(mx))
    return d, mn.value, mx.value


def stereogram(depth, min_depth, max_depth, out_w, out_h, amplitude, noise):
    d = np.ascontiguousarray(depth, np.float32)
    nz = np.ascontiguousarray(noise, np.uint8)
    assert nz.shape == (out_h, out_w, 3)
    out = np.empty((out_h, out_w, 3), np.uint8)
    rc = lib().oracle_stereogram(_p(d), C.c_int32(d.shape[0]), C.c_int32(d.shape[1]), C.c_float(min_depth),
                                 C.c_float(max_depth), C.c_int32(out_w), C.c_int32(out_h),
                                 C.c_float(amplitude), _p(nz), _p(out))
    if rc != 0:
        raise IndexError("output_row index out of range (the reference panics here)")
    return out


def depthmap_rgb(depth, min_depth, max_depth):
    d = np.ascontiguousarray(depth, np.float32)
    out = np.empty(d.shape + (3,), np.uint8)
    lib().oracle_depthmap_rgb(_p(d), C.c_int64(d.size), C.c_float(min_depth), C.c_float(max_depth), _p(out))
    return out


def mesh_index(depth):
    """depth [height... indexed i = y*width + x with width = depth.shape[0] as the reference passes
    (data_width, data_height)] -> (vertex_index, nverts, faces [nf,3])"""
    d = np.ascontiguousarray(depth, np.float32)
    w, h = d.shape[0], d.shape[1]
    vi = np.empty((w * h,), np.int32)
    faces = np.empty((2 * (w - 1) * (h - 1), 3), np.int32)
    nv, nf = C.c_int64(), C.c_int64()
    lib().oracle_mesh_index(_p(d), C.c_int32(w), C.c_int32(h), _p(vi), C.byref(nv), C.byref(nf), _p(faces))
    return vi, nv.value, faces[:nf.value].copy()


def mesh_vertices(depth, vertex_index, nverts, original_size):
    d = np.ascontiguousarray(depth, np.float32)
    uv = np.zeros((nverts, 2), np.float32)
    xyz = np.zeros((nverts, 3), np.float32)
    lib().oracle_mesh_vertices(_p(d), C.c_int32(d.shape[0]), C.c_int32(d.shape[1]), _p(vertex_index),
                               C.c_uint32(original_size[0]), C.c_uint32(original_size[1]), _p(uv), _p(xyz))
    return uv, xyz


def rust_display_f64(v: float) -> str:
    """Rust `{}` for f64: shortest round-trip digits, positional, `1` for 1.0, `-0` for -0.0."""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    s = format(Decimal(repr(float(v))), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s


def obj_text(uv, xyz, faces, vertex_mode: str, stem: str, colors=None) -> str:
    """output.rs:550-630 ObjWriter (vertex_mode 'plain' | 'color' | 'texture')"""
    out = []
    if vertex_mode == "texture":
        out.append(f"mtllib {stem}.mtl\nusemtl Textured\n")
        for u, v in uv:
            out.append(f"vt {rust_display_f64(float(u))} {rust_display_f64(1.0 - float(v))}\n")
    for i, (x, y, z) in enumerate(xyz):
        ny, nz = np.float32(-y), np.float32(-z)
        line = f"v {rust_display_f64(float(x))} {rust_display_f64(float(ny))} {rust_display_f64(float(nz))}"
        if vertex_mode == "color" and colors is not None:
            line += "".join(" " + rust_display_f64(float(c) / 255.0) for c in colors[i])
        out.append(line + "\n")
    for f in faces:
        if vertex_mode == "texture":
            out.append("f " + " ".join(f"{int(i) + 1}/{int(i) + 1}" for i in f) + "\n")
        else:
            out.append("f " + " ".join(str(int(i) + 1) for i in f) + "\n")
    return "".join(out)


def rust_display_f64_c(v: float) -> str:
    """the C restatement of Rust's `{}` for f64 (output_oracle.c), used by write_obj"""
    buf = C.create_string_buffer(400)
    lib().oracle_rust_display_f64(C.c_double(v), buf)
    return buf.value.decode()


def write_obj(path, uv, xyz, faces, vertex_mode: str, stem: str, colors=None):
    """output.rs:484-630 ObjWriter straight to a file (C): for meshes too large for obj_text"""
    mode = {"plain": 0, "color": 1, "texture": 2}[vertex_mode]
    uv = np.ascontiguousarray(uv, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    col = None if colors is None else np.ascontiguousarray(colors, np.uint8)
    fn = lib().oracle_write_obj
    fn.restype = C.c_int
    rc = fn(os.fsencode(path), stem.encode(), C.c_int32(mode), _p(uv), _p(xyz), C.c_int64(len(xyz)), _p(faces),
            C.c_int64(len(faces)), _p(col) if col is not None else C.c_void_p(0))
    if rc != 0:
        raise OSError(f"cannot write {path}")


def mtl_text(image_path: str) -> str:
    """output.rs:536-544"""
    return ("newmtl Textured\nKa 0.2 0.2 0.2\nKd 0.8 0.8 0.8\nKs 1.0 1.0 1.0\nillum 2\nNs 0.000500\n"
            f"map_Ka {image_path}\nmap_Kd {image_path}\n\n")


def ply_bytes(xyz, faces, vertex_mode: str, colors=None) -> bytes:
    """output.rs:415-473 PlyWriter"""
    head = ["ply", "format binary_big_endian 1.0", "comment Matrix Eyes 3D surface",
            f"element vertex {len(xyz)}", "property double x", "property double y", "property double z"]
    if vertex_mode == "color":
        head += ["property uchar red", "property uchar green", "property uchar blue"]
    head += [f"element face {len(faces)}", "property list uchar int vertex_indices", "end_header"]
    out = bytearray(("\n".join(head) + "\n").encode())
    for i, (x, y, z) in enumerate(xyz):
        out += struct.pack(">ddd", float(x), float(np.float32(-y)), float(np.float32(-z)))
        if vertex_mode == "color" and colors is not None:
            out += bytes(int(c) for c in colors[i])
    for f in faces:
        out += struct.pack(">BIII", 3, int(f[0]), int(f[1]), int(f[2]))
    return bytes(out)
