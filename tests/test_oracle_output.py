"""CPU tests: the C oracle of the output back end (oracle/output_oracle.c) against the hand-derived
known answers in tests/golden/output_known_answers.json, plus properties the algorithm implies."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import output_oracle as OO

KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "output_known_answers.json")))


def f32_from_bits(s):
    return struct.unpack("<f", struct.pack("<I", int(s, 16)))[0]


def test_clamp_known_answer():
    c = KA["clamp"]
    d, mn, mx = OO.clamp_minmax(np.array(c["input"], np.float32))
    assert np.array_equal(d, np.array(c["clamped"], np.float32))
    assert mn == np.float32(c["min"]) and mx == np.float32(c["max"])


def test_stereogram_step_known_answer():
    c = KA["stereogram_step"]
    noise = np.arange(c["out_w"] * c["out_h"] * 3, dtype=np.uint8).reshape(c["out_h"], c["out_w"], 3)
    out = OO.stereogram(np.array(c["depth"], np.float32), c["min"], c["max"], c["out_w"], c["out_h"],
                        c["amplitude"], noise)
    assert np.array_equal(out[0], noise[0][c["source_index"]])


def test_stereogram_flat_is_periodic():
    c = KA["stereogram_flat"]
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, size=(c["out_h"], c["out_w"], 3), dtype=np.uint8)
    out = OO.stereogram(np.array(c["depth"], np.float32), c["min"], c["max"], c["out_w"], c["out_h"],
                        c["amplitude"], noise)
    P = c["pattern_width"]
    for y in range(c["out_h"]):
        assert np.array_equal(out[y], noise[y][np.arange(c["out_w"]) % P])


def test_mesh_3x3_known_answer():
    c = KA["mesh_3x3"]
    vi, nv, faces = OO.mesh_index(np.array(c["depth"], np.float32))
    assert vi.tolist() == c["vertex_index"] and nv == c["nvertices"]
    assert faces.tolist() == c["faces"]


@pytest.mark.parametrize("which", ["keep", "drop"])
def test_mesh_threshold_inclusive(which):
    c = KA["mesh_threshold"][which]
    a = f32_from_bits(c["a_bits"])
    _, _, faces = OO.mesh_index(np.array([[1.0, a], [1.0, 1.0]], np.float32))
    assert len(faces) == c["nfaces"]


def test_colormap_known_answer():
    c = KA["colormap"]
    rgb = OO.depthmap_rgb(np.array(c["depth"], np.float32), c["min"], c["max"])
    assert rgb.tolist() == c["rgb"]


def test_colormap_is_monotone_in_green():
    # the table's green channel decreases with the index (viridis, reversed): a sanity property
    d = np.linspace(1.0, 2.0, 1000, dtype=np.float32)
    g = OO.depthmap_rgb(d, 1.0, 2.0)[:, 1].astype(int)
    assert np.all(np.diff(g) >= 0)          # nearer (larger inverse depth) -> towards entry 0


def test_rust_display():
    for v, s in KA["rust_display_f64"]["cases"]:
        assert OO.rust_display_f64(v) == s


def test_full_grid_mesh_closed_form():
    """constant depth keeps every triangle: nf = 2(w-1)(h-1), nv = w*h, and the first-use order has
    the closed form id(0,0)=0, id(1,0)=1, id(0,1)=2, row 0: id(0,x)=2x, row 1: id(1,x)=2x+1 (x>=1: 2x+1),
    rows y>=2: id(y,x) = 2w + (y-2)*w + x"""
    w = h = 9
    vi, nv, faces = OO.mesh_index(np.full((w, h), 0.7, np.float32))
    assert nv == w * h and len(faces) == 2 * (w - 1) * (h - 1)
    vi = vi.reshape(h, w)
    assert vi[0].tolist() == [0] + [2 * x for x in range(1, w)]
    assert vi[1].tolist() == [1] + [2 * x + 1 for x in range(1, w)]
    for y in range(2, h):
        assert vi[y].tolist() == [2 * w + (y - 2) * w + x for x in range(w)]
    assert sorted(set(vi.flatten().tolist())) == list(range(w * h))


def test_mesh_vertices_follow_output_rs():
    d = np.array([[0.5, 0.25], [1.0, 2.0]], np.float32)
    vi, nv, _ = OO.mesh_index(np.full((2, 2), 1.0, np.float32))
    uv, xyz = OO.mesh_vertices(d, vi, nv, (200, 100))     # x_multiplier 1, y_multiplier 0.5
    for i in range(4):
        x, y = i % 2, i // 2
        z = np.float32(1.0) / d.flatten()[i]
        assert uv[vi[i]].tolist() == [x / 2, y / 2]
        assert xyz[vi[i]].tolist() == [np.float32(1.0) * np.float32(x / 2 - 0.5) * z,
                                       np.float32(0.5) * np.float32(y / 2 - 0.5) * z, z]


def test_obj_and_ply_writers():
    uv = np.array([[0.0, 0.0], [0.5, 0.25]], np.float32)
    xyz = np.array([[1.0, 2.0, 4.0], [0.1, -0.0, 0.5]], np.float32)
    faces = np.array([[0, 1, 1]], np.int32)
    t = OO.obj_text(uv, xyz, faces, "texture", "out")
    assert t == ("mtllib out.mtl\nusemtl Textured\nvt 0 1\nvt 0.5 0.75\n"
                 "v 1 -2 -4\nv 0.10000000149011612 0 -0.5\nf 1/1 2/2 2/2\n")
    assert OO.obj_text(uv, xyz, faces, "plain", "out").endswith("f 1 2 2\n")
    b = OO.ply_bytes(xyz, faces, "plain")
    assert b.startswith(b"ply\nformat binary_big_endian 1.0\ncomment Matrix Eyes 3D surface\nelement vertex 2\n")
    assert b.endswith(struct.pack(">BIII", 3, 0, 1, 1))
    assert len(b) == b.index(b"end_header\n") + 11 + 2 * 24 + 13


def test_c_obj_writer_equals_python_formatter(tmp_path):
    """The oracle holds the OBJ text twice: obj_text (Python, Decimal(repr(x)) for Rust's `{}` on f64) and
    oracle_write_obj (C, printf-based shortest round-trip search) for full-size meshes.  They must agree on
    numbers of every magnitude and on a whole file in each vertex mode."""
    rng = np.random.default_rng(0)
    vals = list(rng.standard_normal(4000).astype(np.float32).astype(np.float64)) + list(rng.random(4000))
    vals += list(np.ldexp(rng.random(2000), rng.integers(-70, 70, 2000)))
    vals += [0.0, -0.0, 1.0, -1.0, 0.1, 1e-7, 123456789.0, 1e21, 5e-324, 2.5e-310, 1.7976931348623157e308, 0.5, 100.0,
             1 / 3, 0.30000000000000004, float(np.float32(0.1)), float(np.float32(1 / 255)), 250.0, float("inf"), float("nan")]
    for v in vals:
        assert OO.rust_display_f64_c(float(v)) == OO.rust_display_f64(float(v)), repr(v)
    n = 24
    d, _, _ = OO.clamp_minmax(np.abs(rng.standard_normal((n, n))).astype(np.float32) * 0.02 + 0.5)
    vi, nv, faces = OO.mesh_index(d)
    uv, xyz = OO.mesh_vertices(d, vi, nv, (n, n))
    colors = rng.integers(0, 256, size=(nv, 3), dtype=np.uint8)
    for mode in ("plain", "color", "texture"):
        OO.write_obj(str(tmp_path / f"{mode}.obj"), uv, xyz, faces, mode, "mesh", colors if mode == "color" else None)
        assert (tmp_path / f"{mode}.obj").read_text() == OO.obj_text(uv, xyz, faces, mode, "mesh", colors if mode == "color" else None)
