"""The Rust binding shipped under integration/ cannot be compiled here (no Rust toolchain): these checks hold
its `extern "C"` block to include/matrix_eyes_hip.h textually -- the same symbol set, the same argument count per
function, the same struct fields and constants."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _c_functions():
    text = _strip_c(open(os.path.join(ROOT, "include", "matrix_eyes_hip.h")).read())
    out = {}
    for m in re.finditer(r"\b(me_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    out.pop("me_progress_fn", None)
    return out


def _rust_functions():
    text = re.sub(r"//.*", "", open(os.path.join(ROOT, "integration", "hip_ffi.rs")).read())
    out = {}
    for m in re.finditer(r"pub fn (me_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_rust_extern_block_matches_the_header():
    c, r = _c_functions(), _rust_functions()
    assert len(c) >= 35
    assert set(c) == set(r), sorted(set(c) ^ set(r))
    for name in c:
        assert c[name] == r[name], (name, c[name], r[name])


def test_rust_struct_and_constants_match_the_header():
    h = _strip_c(open(os.path.join(ROOT, "include", "matrix_eyes_hip.h")).read())
    rs = open(os.path.join(ROOT, "integration", "hip_ffi.rs")).read()
    body = re.search(r"typedef struct me_model_config \{(.*?)\} me_model_config;", h, flags=re.S).group(1)
    c_fields = re.findall(r"\b([a-z_0-9]+)(?:\[\d+\])?\s*;", body)
    r_body = re.search(r"pub struct MeModelConfig \{(.*?)\n\}", rs, flags=re.S).group(1)
    r_fields = re.findall(r"pub ([a-z_0-9]+):", r_body)
    assert c_fields == r_fields
    for name, value in re.findall(r"\b(ME_[A-Z0-9_]+)\s*=\s*(\d+)", h):
        m = re.search(rf"pub const {name}: i32 = (\d+);", rs)
        assert m and m.group(1) == value, name
    assert re.search(r"#define ME_ABI_VERSION (\d+)", h).group(1) == re.search(r"ME_ABI_VERSION: i32 = (\d+)", rs).group(1)
    # the safe wrapper only calls functions the extern block declares
    wrapper = open(os.path.join(ROOT, "integration", "hip_backend.rs")).read()
    assert set(re.findall(r"ffi::(me_[a-z0-9_]+)", wrapper)) <= set(_rust_functions())
