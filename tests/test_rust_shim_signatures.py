"""The Rust binding (rust-shim/) against the C header, mechanically: no Rust toolchain exists in this image, so nothing compiles the
shim -- this test is what keeps its `extern "C"` block, its call sites and include/jubjub_hip.h in step (CPU only).
Reference boundary the shim stands at: /root/reference/src/lib.rs:1241-1454 (the group / ff trait surface)."""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//.*", "", text)


def _c_class(t):
    """class of one C parameter / return type: ('ptr', const?, depth) or a scalar name"""
    t = re.sub(r"\s+", " ", t.strip())
    if "[" in t:
        t = t[: t.index("[")].rstrip() + " *"
        t = re.sub(r"\b\w+ \*$", "*", t) if re.search(r"\w \w+ \*$", t) else t
    depth = t.count("*")
    if depth:
        return ("ptr", bool(re.match(r"^\s*const\b", t)), depth)
    base = re.sub(r"\bconst\b", "", t).strip()
    return {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "long long": "c_longlong", "void": "void"}[base]


def header_prototypes():
    text = _strip_c(open(os.path.join(ROOT, "include", "jubjub_hip.h")).read())
    out = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[A-Za-z_][\w\s\*]*?)\b(jj_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M):
        params = []
        args = args.strip()
        if args and args != "void":
            for p in args.split(","):
                p = p.strip()
                m = re.match(r"^(.*?)(\b[A-Za-z_]\w*)?\s*(\[\d*\])?$", p)
                ctype, pname, arr = m.group(1), m.group(2), m.group(3)
                if pname in (None, "int", "unsigned", "size_t", "uint64_t", "long", "void", "char") or ctype.strip() in ("", "const", "long"):
                    ctype, pname = p if not arr else p[: p.index("[")], None           # unnamed parameter: all of it is the type
                if arr:
                    ctype = ctype + "*"
                params.append(_c_class(ctype))
        assert name not in out, "prototype declared twice: " + name
        out[name] = (_c_class(ret), params)
    return out


def _rust_class(t):
    t = t.strip()
    depth, const0 = 0, None
    while t.startswith("*"):
        m = re.match(r"^\*(const|mut)\s+(.*)$", t)
        depth += 1
        const0 = m.group(1) == "const"            # the innermost qualifier wins (read last)
        t = m.group(2)
    if depth:
        return ("ptr", const0, depth)
    return {"c_int": "c_int", "c_uint": "c_uint", "usize": "usize", "u64": "u64", "c_longlong": "c_longlong", "i64": "c_longlong", "u32": "c_uint"}[t]


def rust_declarations():
    text = re.sub(r"//.*", "", open(os.path.join(ROOT, "rust-shim", "src", "ffi.rs")).read())
    block = text[text.index('extern "C" {'):]
    out = {}
    for name, args, ret in re.findall(r"pub fn (jj_\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        params = [_rust_class(p.split(":", 1)[1]) for p in args.split(",") if p.strip()]
        assert name not in out, "declared twice: " + name
        out[name] = (_rust_class(ret) if ret else "void", params)
    return out


def test_every_extern_declaration_matches_the_header():
    h, r = header_prototypes(), rust_declarations()
    assert len(h) >= 100, "header parser lost prototypes (%d)" % len(h)
    assert sorted(set(r) - set(h)) == [], "shim binds entry points the header lacks"
    assert sorted(set(h) - set(r)) == [], "header entry points the shim does not bind (python tools/gen_rust_ffi.py)"
    for name in sorted(h):
        (hret, hargs), (rret, rargs) = h[name], r[name]
        assert len(hargs) == len(rargs), "%s: %d parameters in the header, %d in the shim" % (name, len(hargs), len(rargs))
        assert hret == rret, "%s: return %r in the header, %r in the shim" % (name, hret, rret)
        for k, (a, b) in enumerate(zip(hargs, rargs)):
            assert a == b, "%s: parameter %d is %r in the header, %r in the shim" % (name, k, a, b)


def _calls(text):
    """(name, argument count) of every jj_*( ... ) call"""
    out = []
    for m in re.finditer(r"\b(jj_\w+)\(", text):
        i, depth, commas, empty = m.end(), 1, 0, True
        while depth:
            c = text[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                commas += 1
            if depth and not c.isspace():
                empty = False
            i += 1
        out.append((m.group(1), 0 if empty else commas + 1))
    return out


def test_every_call_in_the_safe_layer_matches_its_declaration():
    r = rust_declarations()
    text = re.sub(r"//.*", "", open(os.path.join(ROOT, "rust-shim", "src", "lib.rs")).read())
    calls = _calls(text)
    assert len(calls) >= 25
    for name, nargs in calls:
        assert name in r, "lib.rs calls %s, which ffi.rs does not declare" % name
        assert nargs == len(r[name][1]), "lib.rs calls %s with %d arguments, declared with %d" % (name, nargs, len(r[name][1]))
    text_types = set(re.findall(r"\b(Jj[A-Z]\w*)\b", text))
    declared = set(re.findall(r"pub struct (Jj\w+)", open(os.path.join(ROOT, "rust-shim", "src", "ffi.rs")).read()))
    assert text_types <= declared, "lib.rs names opaque types ffi.rs lacks: %s" % sorted(text_types - declared)


def test_ffi_rs_is_what_the_generator_writes_from_the_current_header(tmp_path):
    path = os.path.join(ROOT, "rust-shim", "src", "ffi.rs")
    before = open(path).read()
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py")], check=True, capture_output=True)
        assert open(path).read() == before, "rust-shim/src/ffi.rs is stale: python tools/gen_rust_ffi.py"
    finally:
        open(path, "w").write(before)


def test_shim_tree_is_complete():
    for f in ("Cargo.toml", "build.rs", "README.md", "src/lib.rs", "src/ffi.rs"):
        assert os.path.exists(os.path.join(ROOT, "rust-shim", f)), f
    lib = open(os.path.join(ROOT, "rust-shim", "src", "lib.rs")).read()
    assert "pub struct Pooled<'a>" in lib and "impl Drop for Pooled" in lib           # (round 5: existed only inside a comment of INTEGRATION.md)
    assert lib.count("{") == lib.count("}") and lib.count("(") == lib.count(")")
