#!/usr/bin/env python3
"""Generates the raw Rust binding of the C ABI (`latticefold-hip-sys/src/lib.rs`) from include/lfhip.h and include/lfplus.h.

    python tools/gen_rust_bindings.py            print the binding
    python tools/gen_rust_bindings.py --write    rewrite bindings/latticefold-hip-sys/src/lib.rs and the block between the
                                                 GENERATED markers of INTEGRATION.md

The reference is safe Rust with `#![forbid(unsafe_code)]` (crates/latticefold/src/lib.rs:4): the `extern "C"` block lives in a new
`-sys` crate.  No Rust toolchain exists in the build image, so the text is derived mechanically from the headers (every prototype, the
error enums, `lf_params`, the exchange callback) and tests/test_abi_cpu.py checks that header, binding file and INTEGRATION.md agree
symbol by symbol and that the shared library exports each of them."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = ("include/lfhip.h", "include/lfplus.h")
OUT = os.path.join(ROOT, "bindings", "latticefold-hip-sys", "src", "lib.rs")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED BINDING (tools/gen_rust_bindings.py) -->", "<!-- END GENERATED BINDING -->"

SCALARS = {
    "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "char": "c_char", "float": "f32", "double": "f64", "void": "c_void",
    "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int8_t": "i8", "int32_t": "i32", "int64_t": "i64",
}
OPAQUE = ["lf_ctx", "lf_witness", "lf_witness_job", "lf_transcript", "lfplus_ctx", "lfplus_transcript"]
KEYWORDS = {"in": "inp", "type": "ty", "ref": "r", "fn": "f", "mod": "m", "box": "b", "use": "u", "loop": "lp", "match": "mt", "move": "mv", "self": "this"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def rust_type(ctype):
    """`const uint64_t *const *` -> `*const *const u64`"""
    toks = re.findall(r"\*|[A-Za-z_][A-Za-z_0-9]*", ctype)
    base, quals, ptrs = [], False, []          # ptrs: constness of the pointee at each level, innermost first
    pending_const = False
    for t in toks:
        if t == "const":
            pending_const = True
        elif t == "*":
            ptrs.append(pending_const)
            pending_const = False
        elif t == "struct":
            continue
        else:
            base.append(t)
    name = " ".join(base)
    if name in SCALARS:
        r = SCALARS[name]
    elif name in OPAQUE or name in ("lf_params",):
        r = name
    elif name in ("lf_exchange_fn", "lfplus_exchange_fn"):
        r = "lf_exchange_fn"
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    # C reads inside-out: `const T *const *p`: first `*` points at const T, second `*` points at a const pointer.  The constness of level i's
    # pointee is the `const` seen before that `*` -- for the first level that is the base type's const.
    for is_const in ptrs:
        r = ("*const " if is_const else "*mut ") + r
    return r


def split_params(s):
    s = s.strip()
    if s in ("", "void"):
        return []
    out = []
    for i, p in enumerate(x.strip() for x in s.split(",")):
        m = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)?$", p)
        ctype, name = m.group(1).strip(), m.group(2)
        if name in SCALARS or name in OPAQUE or name in ("lf_params", "lf_exchange_fn", "lfplus_exchange_fn", "unsigned", "const") or not ctype:   # unnamed parameter
            ctype, name = p, None
        if name is None:
            base = re.findall(r"[A-Za-z_][A-Za-z_0-9]*", ctype)[-1]
            name = {"lf_ctx": "ctx", "lf_transcript": "t", "lf_witness": "w", "lf_params": "p", "lfplus_ctx": "ctx"}.get(base, f"a{i}")
        out.append((KEYWORDS.get(name, name), rust_type(ctype)))
    return out


def parse(path):
    text = strip_comments(open(os.path.join(ROOT, path)).read())
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = text.replace('extern "C" {', "").replace("\n}\n", "\n")
    enums = []
    for body in re.findall(r"enum\s*\{(.*?)\}\s*;", text, flags=re.S):
        for name, val in re.findall(r"([A-Z_0-9]+)\s*=\s*(-?\d+)", body):
            enums.append((name, int(val)))
    text = re.sub(r"enum\s*\{.*?\}\s*;", "", text, flags=re.S)
    text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    text = re.sub(r"typedef[^;]*;", "", text)
    fns = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(lf(?:plus)?_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        fns.append((name, split_params(params), None if ret == "void" else rust_type(ret)))
    return enums, fns


def defines(path):
    text = strip_comments(open(os.path.join(ROOT, path)).read())
    out = []
    for name, val in re.findall(r"^\s*#define\s+(LF[A-Z_0-9]*)\s+\(?(-?\d+)(?:ULL)?\)?\s*$", text, flags=re.M):
        out.append((name, int(val)))
    return out


def generate():
    L = ["// GENERATED by tools/gen_rust_bindings.py from include/lfhip.h and include/lfplus.h -- do not edit.",
         "// latticefold-hip-sys: raw binding of liblfhip.so (MI355X / gfx950).  Every pointer is a HOST pointer; every `c_int` result is 0 or a",
         "// negative LF_ERR_* / LFPLUS_E_* code.  The comments of the headers are the documentation.",
         "#![allow(non_camel_case_types, non_upper_case_globals, non_snake_case)]",
         "use core::ffi::{c_char, c_int, c_uint, c_void};", ""]
    for o in OPAQUE:
        L.append(f"#[repr(C)] pub struct {o} {{ _p: [u8; 0] }}")
    L += ["",
          "/// DecompositionParams + CCS shape (decomposition_parameters.rs:11-20, arith.rs:50-74)",
          "#[repr(C)] #[derive(Clone, Copy, Debug)]",
          "pub struct lf_params { pub s: u32, pub wit_len: u32, pub l: u32, pub L: u32, pub K: u32, pub b: u32, pub B: u64, pub kappa: u32, pub t: u32, pub q: u32, pub d: u32 }",
          "/// all-gather `words` u64 from every rank into recv_all[world * words] in rank order; 0 on success (lf_set_sharding)",
          "pub type lf_exchange_fn = Option<unsafe extern \"C\" fn(user: *mut c_void, send: *const u64, recv_all: *mut u64, words: usize) -> c_int>;", ""]
    allfns = []
    for h in HEADERS:
        enums, fns = parse(h)
        L.append(f"// ---- {h} " + "-" * (100 - len(h)))
        for name, val in defines(h):
            ty = "u64" if val > 2 ** 31 else "c_int"
            L.append(f"pub const {name}: {ty} = {val};")
        for name, val in enums:
            L.append(f"pub const {name}: c_int = {val};")
        L.append('#[link(name = "lfhip")]')
        L.append('extern "C" {')
        for name, params, ret in fns:
            ps = ", ".join(f"{n}: {t}" for n, t in params)
            L.append(f"    pub fn {name}({ps})" + (f" -> {ret}" if ret else "") + ";")
            allfns.append(name)
        L += ["}", ""]
    return "\n".join(L), allfns


def main():
    text, fns = generate()
    if "--write" in sys.argv:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        open(OUT, "w").write(text)
        doc = open(DOC).read()
        if BEGIN in doc and END in doc:
            a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
            doc = doc[:a] + "\n```rust\n" + text + "```\n" + doc[b:]
            open(DOC, "w").write(doc)
        print(f"{len(fns)} functions -> {os.path.relpath(OUT, ROOT)}" + (", INTEGRATION.md" if BEGIN in doc else ""))
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
