"""The drop-in boundary, mechanically: include/p2e.h, the Rust binding a maintainer of the reference would add
(rust/src/gpu/ffi.rs), the Python binding's EXPORTS list and the symbols the built library
really exports must describe ONE surface -- same symbol set, same argument count, same integer widths and constness
(`long` <-> `i64`, `size_t` <-> `usize`, `const uint64_t *` <-> `*const u64` ...).  Nobody compiles the Rust file in this
image (no rustc), so this parser is the only thing standing between it and drift.  Reference trait surface the
boundary stands in for: gates/mul_nonnative.rs:229-342, gadgets/nonnative.rs:608-895."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "p2e.h")
FFI = os.path.join(ROOT, "rust", "src", "gpu", "ffi.rs")

C_SCALARS = {"int": "i32", "unsigned": "u32", "long": "i64", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32",
             "int32_t": "i32", "uint8_t": "u8", "float": "f32", "void": "()", "char": "c_char"}
C_STRUCTS = {"p2e_ctx": "P2eCtx", "p2e_wire_map": "P2eWireMap", "p2e_curve_program": "P2eCurveProgram",
             "p2e_gen_desc": "P2eGenDesc", "p2e_gen_wiring": "P2eGenWiring", "p2e_aux_desc": "P2eAuxDesc",
             "p2e_ux_desc": "P2eUxDesc", "p2e_wire_map_entry": "P2eWireMapEntry", "p2e_segment_desc": "P2eSegmentDesc"}


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _c_type(t):
    """'const uint8_t *' -> '*const u8'; 'p2e_ctx **' -> '*mut *mut P2eCtx'; 'void *' -> '*mut c_void'."""
    t = t.strip()
    stars = t.count("*")
    core = t.replace("*", " ").split()
    const = "const" in core
    core = [w for w in core if w not in ("const", "struct")]
    assert len(core) == 1, t
    base = core[0]
    if base == "void" and stars:
        rust = "c_void"
    elif base in C_SCALARS:
        rust = C_SCALARS[base]
    else:
        rust = C_STRUCTS[base]
    for k in range(stars):
        innermost = k == 0
        rust = ("*const " if (const and innermost) else "*mut ") + rust
    return rust


def parse_header():
    src = _strip_c_comments(open(HEADER).read())
    src = re.sub(r"^\s*#[^\n]*(\\\n[^\n]*)*", " ", src, flags=re.M)           # preprocessor lines
    src = re.sub(r'extern\s+"C"\s*\{', " ", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(p2e_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = re.sub(r"\[\d*\]", "*", a.strip())                  # uint8_t x[32] is a pointer parameter
                mm = re.match(r"(.*?)(\w+)\s*(\**)$", a)                # "<type> name" (name may carry the trailing '*'s)
                typ = (mm.group(1) + mm.group(3)).strip()
                params.append(_c_type(typ))
        out[name] = (_c_type(ret), params)
    return out


def parse_ffi():
    src = re.sub(r"//[^\n]*", "", open(FFI).read())
    blocks = re.findall(r'((?:#\[[^\]]*\]\s*)*)extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S)
    out = {}
    for _attrs, body in blocks:
        for m in re.finditer(r"pub\s+fn\s+(p2e_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", body, flags=re.S):
            name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
            params = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
            out[name] = (ret, params)
    return out, blocks


def test_rust_binding_declares_exactly_the_header():
    h = parse_header()
    r, _ = parse_ffi()
    assert len(h) >= 60
    assert sorted(set(h) - set(r)) == [], "declared in include/p2e.h but missing from rust/src/gpu/ffi.rs"
    assert sorted(set(r) - set(h)) == [], "declared in rust/src/gpu/ffi.rs but not in include/p2e.h"
    for name in sorted(h):
        assert h[name] == r[name], f"{name}: header {h[name]} != ffi.rs {r[name]}"


def test_link_attribute_sits_on_the_extern_block():
    """rustc ignores #[link] on anything but an extern block: the extern block would then not link
    libp2e_hip.so at all."""
    _, blocks = parse_ffi()
    assert len(blocks) == 1 and '#[link(name = "p2e_hip")]' in blocks[0][0]
    src = open(FFI).read()
    assert src.count("#[link(") == 1


def test_python_exports_and_library_symbols_equal_the_header():
    import plonky2_ecdsa_amd as p2e
    h = parse_header()
    assert sorted(p2e.EXPORTS) == sorted(h)
    so = p2e.LIB_PATH
    if os.path.exists(so):
        syms = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
        exported = {ln.split()[-1] for ln in syms.splitlines() if " T " in ln and ln.split()[-1].startswith("p2e_")}
        assert exported == set(h)
