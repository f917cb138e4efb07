"""The Rust side of the boundary is generated from include/mixlab_gpu.h by tools/gen_rust_ffi.py.  There is no
rustc in this image, so the block is checked from the other three sides instead: it is the generator's output for the header as it is now,
it declares exactly what the shared library exports, and the `#[repr(C)]` layouts it states are the ones gcc computes for the C structs."""
import json
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import gen_rust_ffi as gen  # noqa: E402


def _header():
    return gen.Header(gen.HEADER.read_text())


def test_integration_md_holds_the_generators_output_for_the_current_header():
    text = gen.INTEGRATION.read_text()
    assert gen.BEGIN in text and gen.END in text
    block = text[text.index(gen.BEGIN): text.index(gen.END) + len(gen.END)]
    assert block == gen.integration_block(_header()), "INTEGRATION.md is stale: run `python tools/gen_rust_ffi.py --update-integration`"


def test_every_exported_symbol_is_declared_and_nothing_else():
    h = _header()
    declared = {f[0] for f in h.funcs}
    out = subprocess.run(["nm", "-D", "--defined-only", str(ROOT / "mixlab_amd" / "libmixlab_gpu.so")], capture_output=True, text=True, check=True).stdout
    exported = {m.group(1) for m in re.finditer(r" T (mx_\w+)$", out, flags=re.M)}
    assert exported - declared == set(), f"exported but not declared in the header / Rust block: {sorted(exported - declared)}"
    assert declared - exported == set(), f"declared but not exported: {sorted(declared - exported)}"
    rust = h.rust()
    for name in exported:
        assert f"pub fn {name}(" in rust


def test_repr_c_layouts_equal_what_gcc_computes(tmp_path):
    h = _header()
    src = tmp_path / "layout.c"
    src.write_text(h.layout_c())
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", str(ROOT / "include"), "-o", str(exe), str(src)], check=True)
    c_layout = json.loads(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)
    mine = h.layout_json()
    assert set(c_layout) == set(mine) and len(mine) >= 20
    for name in mine:
        assert mine[name] == c_layout[name], f"{name}: generator {mine[name]} vs gcc {c_layout[name]}"
    # and every struct line of the Rust block states that size
    rust = h.rust()
    for name, lay in mine.items():
        assert re.search(rf"pub struct {name} \{{.*\}}   // {lay['size']} bytes", rust), name


def test_pointer_constness_and_array_parameters_are_translated():
    rust = _header().rust()
    assert "pub fn mx_last_error() -> *const c_char;" in rust
    assert "frames: *const *mut mx_dframe" in rust                      # mx_dframe* const* frames
    assert "pub fn mx_graph_eq_repair_stats(g: *mut mx_graph, out: *mut u64) -> c_int;" in rust   # uint64_t out[8] decays
    assert "inputs: *const mx_video_input" in rust                      # const mx_video_input inputs[4]
    assert "pub data: [*mut u8; 3]" in rust and "pub stride: [i32; 3]" in rust
    assert "pub const MX_ERR_FULL: c_int = -6;" in rust and "pub const MX_KIND_MIXER: u32 = 4;" in rust and "pub const MX_FLAG_FP_CONTRACT: u32 = 16;" in rust
