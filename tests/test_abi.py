"""CPU tests of the drop-in boundary: libmatrixeyes_hip.so loads and exports every symbol that
include/*.h declares (no compute calls without a GPU), the ctypes prototypes cover exactly that set,
and a context cannot be created without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import matrix_eyes_amd as m
from matrix_eyes_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("matrix_eyes_hip.h", "matrix_eyes_hip_ops.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(me_[a-z0-9_]+)\s*\(", text))
    names.discard("me_progress_fn")
    return names


def test_library_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 35
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
    assert names == set(L.SIGNATURES), (names ^ set(L.SIGNATURES))


def test_abi_version_and_default_config(lib):
    assert lib.me_abi_version() == 4
    c = L.CModelConfig()
    assert lib.me_default_config(C.byref(c)) == 0
    d = m.ModelConfig().to_c()
    for f, _ in L.CModelConfig._fields_:
        a, b = getattr(c, f), getattr(d, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == pytest.approx(b)), f
    assert lib.me_default_config(None) == 1


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.me_ctx_create(0, 0, None, C.byref(h))
    assert rc == 5 and not h.value                       # ME_ERR_HIP
    assert b"no CPU fallback" in lib.me_last_error(None)
    with pytest.raises(m.MatrixEyesError) as e:
        m.Context(0, "f16", m.ModelConfig.tiny())
    assert e.value.code == 5
    # null-context calls are rejected, not crashed on
    assert lib.me_weights_finalize(None) == 1
    assert lib.me_extract_depth(None, None, 1, None, None, None) == 1
    assert lib.me_expected_weight_count(None) == 0
    assert lib.me_ctx_set_graph(None, 1) == 1 and lib.me_graph_launch_count(None) == 0
    assert lib.me_load_checkpoint_pt(None, b"x.pt") == 1 and lib.me_unused_weight_count(None) == 0
    lib.me_ctx_destroy(None)


def test_headers_cite_the_reference():
    text = open(os.path.join(ROOT, "include", "matrix_eyes_hip.h")).read()
    for cite in ("mod.rs:251-363", "vit.rs:328-346", "encoder.rs:218-335", "decoder.rs:153-208",
                 "fov.rs:40-88", "output.rs:141-193", "output.rs:264-363", "reconstruction.rs:114-124"):
        assert cite in text, cite
