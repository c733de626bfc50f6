"""The Rust FFI crate under bindings/rust/ cannot be compiled here (no cargo/rustc in the image), so its
declarations are held against include/rtx950.h as text: every exported function is declared with the same
argument count, every struct has the same fields in the same order, every constant has the same value."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rtx950.h")).read()
RUST = open(os.path.join(ROOT, "bindings", "rust", "rtx950-sys", "src", "lib.rs")).read()


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _c_functions():
    text = _strip_c_comments(HEADER)
    out = {}
    for m in re.finditer(r"\b(rt_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def _rust_functions():
    block = RUST[RUST.index('extern "C" {'):]
    block = block[: block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (rt_\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def _c_structs():
    text = _strip_c_comments(HEADER)
    out = {}
    for m in re.finditer(r"typedef struct (rt_\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if decl:
                fields += [re.sub(r"[\s\*]", "", f) for f in re.split(r",", re.sub(r"^.*?(\w+(?:\s*,\s*\w+)*)$", r"\1", decl))]
        out[m.group(1)] = fields
    return out


def _rust_structs():
    out = {}
    for m in re.finditer(r"pub struct (rt_\w+)\s*\{(.*?)\n\}", RUST, flags=re.S):
        out[m.group(1)] = re.findall(r"pub (\w+)\s*:", m.group(2))
    return out


def test_every_header_function_is_declared_with_the_same_arity():
    c, r = _c_functions(), _rust_functions()
    assert len(c) >= 17
    assert c == r


def test_struct_fields_match_in_order():
    c, r = _c_structs(), _rust_structs()
    for name in ("rt_vec3", "rt_cube", "rt_sphere", "rt_material", "rt_light", "rt_camera", "rt_frame", "rt_debug_planes", "rt_counters"):
        assert c[name] == r[name], name


def test_constants_match():
    for name, value in re.findall(r"#define (RT\w+|RTX950_ABI_VERSION)\s+\(?(-?\w+)\)?", _strip_c_comments(HEADER)):
        if name == "RTX950_H":
            continue
        m = re.search(r"pub const %s: \w+ = ([^;]+);" % name, RUST)
        assert m, name
        assert int(m.group(1).replace("_", ""), 0) == int(value.rstrip("u"), 0), name
    for name, value in re.findall(r"(RT_(?:OK|ERR_\w+))\s*=\s*(-?\d+)", HEADER):
        m = re.search(r"pub const %s: c_int = (-?\d+);" % name, RUST)
        assert m and int(m.group(1)) == int(value), name
