"""Text-level view of the C ABI: parses the prototypes of include/omok_mi355x.h and the `extern "C"` block of
bindings/omok_mi355x.rs into comparable records (name, return type, argument types as (kind, width, pointer depth, const)),
and can print the Rust declarations for the header (`python tools/abi_text.py` -> the extern block of the .rs file).
Used by tests/test_abi.py; rustc is absent from the build image, so this comparison is what keeps the Rust text honest."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "omok_mi355x.h")
RUST = os.path.join(ROOT, "bindings", "omok_mi355x.rs")

# scalar C type -> (kind, bits); `int` is the platform C int (Rust: c_int)
C_SCALARS = {"int": ("cint", 32), "int32_t": ("int", 32), "int64_t": ("int", 64), "uint8_t": ("uint", 8), "uint16_t": ("uint", 16),
             "uint32_t": ("uint", 32), "uint64_t": ("uint", 64), "float": ("float", 32), "double": ("float", 64), "char": ("char", 8),
             "void": ("void", 0), "omok_engine": ("engine", 0), "omok_config": ("config", 0)}
RUST_SCALARS = {"c_int": ("cint", 32), "i32": ("int", 32), "i64": ("int", 64), "u8": ("uint", 8), "u16": ("uint", 16), "u32": ("uint", 32),
                "u64": ("uint", 64), "f32": ("float", 32), "f64": ("float", 64), "c_char": ("char", 8), "c_void": ("void", 0), "()": ("void", 0),
                "OmokEngine": ("engine", 0), "OmokConfig": ("config", 0)}
TO_RUST = {("cint", 32): "c_int", ("int", 32): "i32", ("int", 64): "i64", ("uint", 8): "u8", ("uint", 16): "u16", ("uint", 32): "u32",
           ("uint", 64): "u64", ("float", 32): "f32", ("float", 64): "f64", ("char", 8): "c_char", ("void", 0): "c_void",
           ("engine", 0): "OmokEngine", ("config", 0): "OmokConfig"}


def _strip_c(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def _c_type(t):
    """'const float*' -> (kind, bits, pointer levels as a tuple of 'const' / 'mut', outermost last)."""
    t = t.strip()
    depth = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = [w for w in base if w != "const"]
    assert len(base) == 1 and base[0] in C_SCALARS, t
    kind, bits = C_SCALARS[base[0]]
    # the header only uses `const T*`, `T*` and `T**` (no pointer-to-const-pointer): const applies to the pointee of the innermost level
    ptr = tuple(("const" if (const and i == 0) else "mut") for i in range(depth))
    return kind, bits, ptr


def parse_header(path=HEADER):
    text = _strip_c(open(path).read())
    text = text[text.index('extern "C"'):]
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(omok_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argl = []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", a)
                assert mm, a
                argl.append((mm.group(2), _c_type(mm.group(1))))
        out[name] = (_c_type(ret), argl)
    return out


def _rust_type(t):
    t = t.strip()
    ptr = []
    while t.startswith("*"):
        mm = re.match(r"\*(const|mut)\s+(.*)$", t)
        assert mm, t
        ptr.append(mm.group(1))
        t = mm.group(2).strip()
    t = t.split("::")[-1]
    assert t in RUST_SCALARS, t
    kind, bits = RUST_SCALARS[t]
    return kind, bits, tuple(reversed(ptr))  # innermost level first, like _c_type


def parse_rust(path=RUST):
    text = re.sub(r"//[^\n]*", "", open(path).read())
    i = text.index('extern "C" {')
    block = text[i:text.index("\n}", i)]
    out = {}
    for m in re.finditer(r"pub fn (omok_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        argl = []
        for a in [x for x in args.split(",") if x.strip()]:
            an, at = a.split(":", 1)
            argl.append((an.strip(), _rust_type(at)))
        out[name] = (_rust_type(ret) if ret else ("void", 0, ()), argl)
    return out


def parse_rust_config(path=RUST):
    text = re.sub(r"//[^\n]*", "", open(path).read())
    m = re.search(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct OmokConfig\s*\{([^}]*)\}", text)
    return [(f.split(":")[0].replace("pub", "").strip(), _rust_type(f.split(":")[1])) for f in m.group(1).split(",") if f.strip()]


def parse_header_config(path=HEADER):
    text = _strip_c(open(path).read())
    m = re.search(r"typedef struct\s*\{([^}]*)\}\s*omok_config\s*;", text)
    out = []
    for f in m.group(1).split(";"):
        f = f.strip()
        if f:
            mm = re.match(r"(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", f)
            out.append((mm.group(2), _c_type(mm.group(1))))
    return out


def parse_defines(path=HEADER):
    """#define OMOK_X <integer> -> {name: value}"""
    out = {}
    for m in re.finditer(r"^#define (OMOK_[A-Z0-9_]+) \(?(-?\d+)\)?", _strip_c(open(path).read()), flags=re.M):
        out[m.group(1)] = int(m.group(2))
    return out


def parse_rust_consts(path=RUST):
    text = re.sub(r"//[^\n]*", "", open(path).read())
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (OMOK_[A-Z0-9_]+)\s*:\s*[a-z0-9_]+\s*=\s*(-?\d+)\s*;", text)}


def rust_decl(name, sig):
    (rk, rb, rp), args = sig

    def ty(kind, bits, ptr):
        t = TO_RUST[(kind, bits)]
        for p in ptr:
            t = f"*{p} {t}"
        return t
    ret = "" if (rk == "void" and not rp) else f" -> {ty(rk, rb, rp)}"
    rs_args = ", ".join(f"{'input' if n == 'in' else n}: {ty(*t)}" for n, t in args)
    return f"    pub fn {name}({rs_args}){ret};"


if __name__ == "__main__":
    for name, sig in parse_header().items():
        print(rust_decl(name, sig))
    if "--consts" in sys.argv:
        for k, v in parse_defines().items():
            print(f"pub const {k}: i32 = {v};")
