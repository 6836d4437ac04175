#!/usr/bin/env python3
"""Generate shim/src/ffi.rs -- the Rust `extern "C"` block for libp2mt_hip.so -- from include/p2mt.h.

The image has no Rust toolchain, so the shim ships as source; what CAN be checked here is that the extern block
never drifts from the header: tests/test_abi_cpu.py regenerates it and compares with the committed file.

  python tools/gen_rust_ffi.py            # rewrite shim/src/ffi.rs
  python tools/gen_rust_ffi.py --check    # exit 1 if the committed file differs
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "p2mt.h")
OUT = os.path.join(ROOT, "shim", "src", "ffi.rs")

SCALARS = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "size_t": "usize", "uint64_t": "u64", "int64_t": "i64",
           "uint32_t": "u32", "int32_t": "i32", "uint8_t": "u8", "int8_t": "i8", "float": "f32", "double": "f64",
           "char": "c_char", "void": "c_void", "p2mt_target": "u64"}
RUST_KEYWORDS = {"in", "type", "ref", "box", "loop", "match", "move", "fn", "mod", "use", "as", "where", "impl", "self"}


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def rust_type(ctype, stars, array, outer_const=False):
    ctype = ctype.strip()
    const = False
    if ctype.startswith("const "):
        const, ctype = True, ctype[6:].strip()
    if ctype.startswith("struct "):
        ctype = ctype[7:].strip()
    base = SCALARS.get(ctype, ctype)
    t = base
    # a C array parameter is a pointer to its element type
    levels = stars + (1 if array else 0)
    for lv in range(levels):
        innermost = lv == 0
        t = ("*const " if ((const and innermost) or (outer_const and not innermost)) else "*mut ") + t
    return t


def parse_param(p):
    p = p.strip()
    if p in ("void", ""):
        return None
    m = re.match(r"^(.*?)(\**)\s*([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?$", p)
    if not m:
        raise ValueError("cannot parse parameter %r" % p)
    ctype, stars, name, array = m.group(1), len(m.group(2)), m.group(3), m.group(4)
    # `const T *const *name` style: stars may be split by const
    outer_const = "*const" in ctype
    ctype = ctype.replace("*const", "*").strip()
    extra = ctype.count("*")
    ctype = ctype.replace("*", "").strip()
    if name in RUST_KEYWORDS:
        name += "_"
    return name, rust_type(ctype, stars + extra, bool(array), outer_const)


def parse_header(text):
    text = strip_comments(text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = text.replace('extern "C" {', "")
    opaque, structs, funcs, consts = [], [], [], []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.append(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r"^(const\s+)?(\w+)\s+(.*)$", decl, flags=re.S)
            const, ctype, names = dm.group(1) or "", dm.group(2), dm.group(3)
            for n in names.split(","):
                nm = re.match(r"^\s*(\**)\s*(\w+)\s*(\[(\d+)\])?\s*$", n)
                stars, fname, alen = len(nm.group(1)), nm.group(2), nm.group(4)
                t = rust_type(const + ctype, stars, False)
                if alen:
                    t = "[%s; %s]" % (t, alen)
                fields.append((fname, t))
        structs.append((m.group(3), fields))
    # function-pointer typedefs: typedef int (*name)(params);  (callbacks a caller hands to the library)
    fnptrs = []
    for m in re.finditer(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ps = [parse_param(q) for q in " ".join(m.group(3).split()).split(",")]
        fnptrs.append((m.group(2), [q for q in ps if q], rust_type(m.group(1), 0, False)))
    text = re.sub(r"typedef\s+\w+\s*\(\s*\*\s*\w+\s*\)\s*\([^;]*?\)\s*;", " ", text, flags=re.S)
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"typedef\s+enum\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", body, flags=re.S)
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(p2mt_\w+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.S):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        rstars = ret.count("*")
        rtype = ret.replace("*", "").strip()
        rust_ret = rust_type(rtype, rstars, False)
        ps = [parse_param(p) for p in params.split(",")] if params.strip() else []
        funcs.append((name, [p for p in ps if p], None if rust_ret == "c_void" else rust_ret))
    for m in re.finditer(r"^\s*(P2MT_\w+)\s*=\s*(-?\d+)", text, flags=re.M):
        consts.append((m.group(1), m.group(2)))
    return opaque, structs, funcs, consts, fnptrs


def generate():
    opaque, structs, funcs, consts, fnptrs = parse_header(open(HEADER).read())
    struct_names = {s[0] for s in structs}
    out = ["//! GENERATED by tools/gen_rust_ffi.py from include/p2mt.h -- do not edit; `python tools/gen_rust_ffi.py` rewrites it and",
           "//! tests/test_abi_cpu.py::test_rust_ffi_matches_header fails when it is stale.",
           "//! The C ABI of libp2mt_hip.so: plain pointers and sizes, status codes (0 = ok, negative = the reference's panics).",
           "#![allow(non_camel_case_types, dead_code)]",
           "use std::os::raw::{c_char, c_void};",
           ""]
    for name, val in consts:
        out.append("pub const %s: i32 = %s;" % (name, val))
    out.append("pub const P2MT_GOLDILOCKS_FIELD_ORDER: u64 = 0xFFFF_FFFF_0000_0001;")
    out.append("pub const P2MT_MAX_PROOF_LEN: usize = 64;")
    out.append("")
    for o in sorted(set(opaque) - struct_names):
        out.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % o)
    out.append("")
    for name, fields in structs:
        out.append("#[repr(C)] #[derive(Clone, Copy)]")
        out.append("pub struct %s {" % name)
        for f, t in fields:
            out.append("    pub %s: %s," % (f, t))
        out.append("}")
    out.append("")
    for name, params, ret in fnptrs:
        sig = ", ".join("%s: %s" % q for q in params)
        out.append('pub type %s = Option<unsafe extern "C" fn(%s)%s>;' % (name, sig, " -> " + ret if ret != "c_void" else ""))
    if fnptrs:
        out.append("")
    out.append('#[link(name = "p2mt_hip")]')
    out.append('extern "C" {')
    for name, params, ret in funcs:
        sig = ", ".join("%s: %s" % p for p in params)
        out.append("    pub fn %s(%s)%s;" % (name, sig, " -> " + ret if ret else ""))
    out.append("}")
    out.append("")
    return "\n".join(out), [f[0] for f in funcs]


def main():
    text, _ = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote %s (%d lines)" % (OUT, text.count("\n")))


if __name__ == "__main__":
    main()
