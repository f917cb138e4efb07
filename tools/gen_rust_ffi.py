#!/usr/bin/env python
"""Generate the Rust side of the C ABI from include/mixlab_gpu.h: every constant, every `#[repr(C)]` struct, every opaque handle
and the complete `extern "C"` block -- what a mixlab maintainer drops into `src/gpu/ffi.rs`.

There is no rustc in this image, so the block cannot be compiled here; what CAN be checked is, and is (tests/test_cpu_rust_ffi.py):
  * the block in INTEGRATION.md is this script's output verbatim (hand edits and header drift both fail the test);
  * every symbol the shared library exports (`nm -D`) has a declaration, and nothing is declared that is not exported;
  * the size and field offsets this script computes for each `#[repr(C)]` struct (Rust's repr(C) layout rule = C's) equal what gcc
    computes for the C struct (a generated C program prints sizeof / offsetof).

    python tools/gen_rust_ffi.py                 # the Rust source on stdout
    python tools/gen_rust_ffi.py --layout-c      # the C program that prints every struct's layout
    python tools/gen_rust_ffi.py --layout-json   # the layout this script computes (what the test compares with the C program's output)
    python tools/gen_rust_ffi.py --update-integration   # rewrite the block between the markers in INTEGRATION.md
"""
from __future__ import annotations

import json
import pathlib
import re
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "mixlab_gpu.h"
INTEGRATION = ROOT / "INTEGRATION.md"
BEGIN, END = "<!-- BEGIN GENERATED: tools/gen_rust_ffi.py -->", "<!-- END GENERATED: tools/gen_rust_ffi.py -->"

SCALARS = {  # C type -> (Rust type, size, alignment)
    "uint8_t": ("u8", 1, 1), "int8_t": ("i8", 1, 1), "uint16_t": ("u16", 2, 2), "int16_t": ("i16", 2, 2),
    "uint32_t": ("u32", 4, 4), "int32_t": ("i32", 4, 4), "uint64_t": ("u64", 8, 8), "int64_t": ("i64", 8, 8),
    "size_t": ("usize", 8, 8), "float": ("f32", 4, 4), "double": ("f64", 8, 8), "int": ("c_int", 4, 4),
    "unsigned": ("c_uint", 4, 4), "char": ("c_char", 1, 1), "void": ("c_void", 0, 1),
}


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


class Header:
    def __init__(self, text: str):
        self.consts: list[tuple[str, str, str]] = []      # (name, rust type, value)
        self.enums: dict[str, str] = {}                   # typedef'd enum name -> rust integer type
        self.structs: dict[str, list[tuple[str, str, int]]] = {}   # name -> [(field, c type, array len or 0)]
        self.opaque: list[str] = []
        self.funcs: list[tuple[str, str, list[tuple[str, str, int]]]] = []   # (name, c return type, [(param, c type, array len)])
        self.order: list[tuple[str, str]] = []            # declaration order: (kind, name)
        self.parse(text)

    # ---- parsing ----
    def parse(self, text: str):
        src = strip_comments(text)
        for m in re.finditer(r"^[ \t]*#define[ \t]+(MX_\w+)[ \t]+([-0-9xa-fA-F]+)[uUlL]*[ \t]*$", src, flags=re.M):
            self.consts.append((m.group(1), "u32", str(int(m.group(2), 0))))
            self.order.append(("const", m.group(1)))
        src = re.sub(r"^[ \t]*#.*$", "", src, flags=re.M)
        src = src.replace('extern "C" {', "")
        # top-level statements: split on ';' at brace depth 0
        depth, cur, stmts = 0, [], []
        for ch in src:
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
            if ch == ";" and depth == 0:
                stmts.append(" ".join("".join(cur).split()))
                cur = []
            else:
                cur.append(ch)
        for s in stmts:
            s = s.strip().lstrip("}").strip()
            if s:
                self.statement(s)

    def enum_body(self, body: str, ty: str | None):
        nxt, items = 0, []
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = (x.strip() for x in item.split("=", 1))
                nxt = int(val, 0)
            else:
                name = item
            items.append((name, nxt))
            nxt += 1
        if ty is None:   # an anonymous enum: its constants are used where the ABI takes uint32_t (kinds, modes) or returns int (status codes)
            ty = "c_int" if any(v < 0 for _n, v in items) else "u32"
        for name, v in items:
            self.consts.append((name, ty, str(v)))
            self.order.append(("const", name))

    def declarators(self, decl: str, where: str):
        """'double a, b' / 'uint8_t* data[3]' / 'const mx_node* nodes' -> [(name, c type, array len)]"""
        decl = decl.strip()
        m = re.match(r"^((?:const\s+)?(?:struct\s+)?\w+(?:\s+const)?)\s*(.*)$", decl)
        if not m:
            raise SystemExit(f"gen_rust_ffi: cannot parse declaration '{decl}' in {where}")
        base, rest = m.group(1), m.group(2)
        out = []
        for d in rest.split(","):
            d = d.strip()
            dm = re.match(r"^((?:\*\s*(?:const\s*)?)*)\s*(\w+)\s*(?:\[\s*(\w*)\s*\])?$", d)
            if not dm:
                raise SystemExit(f"gen_rust_ffi: cannot parse declarator '{d}' of '{decl}' in {where}")
            arr = dm.group(3)
            n = 0 if arr is None else (int(arr) if arr.isdigit() else int(dict((c[0], c[2]) for c in self.consts)[arr]))
            out.append((dm.group(2), self.ctype(base, dm.group(1)), n))
        return out

    @staticmethod
    def ctype(base: str, stars: str) -> str:
        """canonical spelling: '[const ]T' followed by one '*' or '*const' per pointer level (a const AFTER a star qualifies that pointer)"""
        words = base.replace("struct ", "").split()
        core = [w for w in words if w != "const"][0]
        levels = re.findall(r"\*\s*(const)?", stars)
        return ("const " if "const" in words else "") + core + "".join("*const" if c else "*" for c in levels)

    def statement(self, s: str):
        m = re.match(r"^enum\s*\{(.*)\}$", s)
        if m:
            self.enum_body(m.group(1), None)
            return
        m = re.match(r"^typedef\s+enum\s*\{(.*)\}\s*(\w+)$", s)
        if m:
            self.enums[m.group(2)] = "c_uint"
            self.order.append(("enum", m.group(2)))
            self.enum_body(m.group(1), m.group(2))
            return
        m = re.match(r"^typedef\s+struct\s*(\w*)\s*\{(.*)\}\s*(\w+)$", s)
        if m:
            fields = []
            for decl in m.group(2).split(";"):
                if decl.strip():
                    fields += self.declarators(decl, m.group(3))
            self.structs[m.group(3)] = fields
            self.order.append(("struct", m.group(3)))
            return
        m = re.match(r"^typedef\s+struct\s+(\w+)\s+(\w+)$", s)
        if m:
            self.opaque.append(m.group(2))
            self.order.append(("opaque", m.group(2)))
            return
        m = re.match(r"^((?:const\s+)?\w+\s*\**)\s*(mx_\w+)\s*\((.*)\)$", s)
        if m:
            params = []
            body = m.group(3).strip()
            if body and body != "void":
                for p in body.split(","):
                    params += self.declarators(p, m.group(2))
            rm = re.match(r"^((?:const\s+)?\w+)\s*((?:\*\s*)*)$", m.group(1).strip())
            self.funcs.append((m.group(2), self.ctype(rm.group(1), rm.group(2)), params))
            self.order.append(("fn", m.group(2)))
            return
        raise SystemExit(f"gen_rust_ffi: unrecognised declaration in {HEADER.name}: '{s[:120]}'")

    # ---- types ----
    def rust_type(self, c: str, arr: int, param: bool) -> str:
        """c: Header.ctype's spelling.  A pointer is *const when what it points AT is const: the base type for the first level, the
        pointer one level in (a `*const` level) for the others."""
        m = re.match(r"^(const )?(\w+)((?:\*(?:const)?)*)$", c)
        if not m:
            raise SystemExit(f"gen_rust_ffi: unknown C type spelling '{c}'")
        pointee_const, core = bool(m.group(1)), m.group(2)
        if core in SCALARS:
            r = SCALARS[core][0]
        elif core in self.enums or core in self.structs or core in self.opaque:
            r = core
        else:
            raise SystemExit(f"gen_rust_ffi: unknown C type '{c}'")
        for lvl in re.findall(r"\*(const)?", m.group(3)):
            r = ("*const " if pointee_const else "*mut ") + r
            pointee_const = bool(lvl)
        if arr:
            if param:   # an array parameter decays to a pointer to its element
                r = ("*const " if pointee_const else "*mut ") + r
            else:
                r = f"[{r}; {arr}]"
        return r

    def layout(self, name: str):
        """(size, align, [(field, offset, size)]) by C's (= Rust repr(C)'s) rule"""
        off, align, fields = 0, 1, []
        for fname, c, arr in self.structs[name]:
            core = c.replace("const ", "").replace("*const", "").replace("*", "").strip()
            if "*" in c:
                sz, al = 8, 8
            elif core in SCALARS:
                _, sz, al = SCALARS[core]
            elif core in self.enums:
                sz, al = 4, 4
            elif core in self.structs:
                sz, al, _ = self.layout(core)
            else:
                raise SystemExit(f"gen_rust_ffi: field of opaque / unknown type '{c}' in {name}")
            tot = sz * (arr if arr else 1)
            off = (off + al - 1) // al * al
            fields.append((fname, off, tot))
            off += tot
            align = max(align, al)
        return (off + align - 1) // align * align, align, fields

    # ---- output ----
    def rust(self) -> str:
        ver = dict((c[0], c[2]) for c in self.consts)["MX_ABI_VERSION"]
        out = [f"// GENERATED by tools/gen_rust_ffi.py from include/mixlab_gpu.h (MX_ABI_VERSION {ver}, {len(self.funcs)} entry points). Do not edit:",
               "// change the header and run `python tools/gen_rust_ffi.py --update-integration`.",
               "#![allow(non_camel_case_types, dead_code)]",
               "use std::os::raw::{c_char, c_int, c_uint, c_void};", ""]
        consts = dict((c[0], c) for c in self.consts)
        structs_done = False
        for kind, name in self.order:
            if kind == "const":
                _, ty, val = consts[name]
                out.append(f"pub const {name}: {ty} = {val};")
            elif kind == "enum":
                out.append(f"pub type {name} = {self.enums[name]};")
            elif kind == "struct":
                size, align, _ = self.layout(name)
                out.append(f"#[repr(C)] #[derive(Clone, Copy)] pub struct {name} {{ " +
                           ", ".join(f"pub {f}: {self.rust_type(c, a, False)}" for f, c, a in self.structs[name]) + f" }}   // {size} bytes, align {align}")
            elif kind == "opaque":
                out.append(f"#[repr(C)] pub struct {name} {{ _private: [u8; 0] }}")
        out += ["", '#[link(name = "mixlab_gpu")]', 'extern "C" {']
        for name, ret, params in self.funcs:
            args = ", ".join(f"{'r#' + p if p in ('in', 'type', 'fn', 'ref', 'mod', 'box', 'loop', 'match', 'move', 'use') else p}: {self.rust_type(c, a, True)}" for p, c, a in params)
            r = "" if ret == "void" else " -> " + self.rust_type(ret, 0, True)
            out.append(f"    pub fn {name}({args}){r};")
        out.append("}")
        del structs_done
        # Rust-style names for the code around the block (src/gpu/*.rs in INTEGRATION.md): MxGraph = mx_graph, ...
        out += ["", "// Rust-style aliases"]
        for name in list(self.structs) + self.opaque:
            out.append(f"pub type {camel(name)} = {name};")
        return "\n".join(out) + "\n"

    def layout_json(self) -> dict:
        return {n: {"size": self.layout(n)[0], "offsets": {f: o for f, o, _ in self.layout(n)[2]}} for n in self.structs}

    def layout_c(self) -> str:
        lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "mixlab_gpu.h"', "int main(void) {", '    printf("{");']
        first = True
        for n, fields in self.structs.items():
            lines.append(f'    printf("{"" if first else ", "}\\"{n}\\": {{\\"size\\": %zu, \\"offsets\\": {{", sizeof({n}));')
            for i, (f, _c, _a) in enumerate(fields):
                lines.append(f'    printf("{"" if i == 0 else ", "}\\"{f}\\": %zu", offsetof({n}, {f}));')
            lines.append('    printf("}}");')
            first = False
        lines += ['    printf("}\\n");', "    return 0;", "}"]
        return "\n".join(lines) + "\n"


def camel(name: str) -> str:
    return "".join(w.capitalize() for w in name.split("_"))


def integration_block(h: Header) -> str:
    return BEGIN + "\n```rust\n" + h.rust() + "```\n" + END


def main():
    h = Header(HEADER.read_text())
    if "--layout-c" in sys.argv:
        sys.stdout.write(h.layout_c())
    elif "--layout-json" in sys.argv:
        json.dump(h.layout_json(), sys.stdout)
    elif "--update-integration" in sys.argv:
        text = INTEGRATION.read_text()
        if BEGIN not in text or END not in text:
            raise SystemExit(f"INTEGRATION.md has no {BEGIN} ... {END} markers")
        a, b = text.index(BEGIN), text.index(END) + len(END)
        INTEGRATION.write_text(text[:a] + integration_block(h) + text[b:])
        print(f"INTEGRATION.md: {len(h.funcs)} entry points, {len(h.structs)} structs, {len(h.opaque)} opaque handles, {len(h.consts)} constants")
    else:
        sys.stdout.write(h.rust())


if __name__ == "__main__":
    main()
