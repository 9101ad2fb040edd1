"""The C-ABI library loads and exports every symbol include/hpt.h declares; host-only entry points behave; and without
a GPU the product fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hpt_[a-z0-9_]+)\s*\(", text)))


def test_exports_match_header():
    import hpt
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(hpt.lib, s), "libhpt.so does not export %s" % s
    assert sorted(hpt.EXPORTS) == syms


def test_struct_layouts_match_oracle_binding():
    """the ctypes mirrors of the POD structs used by tests agree in size (same field order on both sides of the check)"""
    import hpt
    import oracle_lib as ol
    assert C.sizeof(hpt.BvhNode) == C.sizeof(ol.BvhNode) == 36
    assert C.sizeof(hpt.Texture) == C.sizeof(ol.Texture) == 92
    assert C.sizeof(hpt.Material) == C.sizeof(ol.Material) == 3 * 92 + 24
    assert C.sizeof(hpt.Camera) == C.sizeof(ol.Camera) == 56


def test_host_only_entry_points(olib):
    import hpt
    assert hpt.lib.hpt_version().startswith(b"hpt")
    for args in [(1, 0, 1, 0), (1, 5, 3, 2), (77, 123456, 1024, 33), (0xFFFFFFFF, 2073599, 512, 34)]:
        assert hpt.lib.hpt_rng_f32(*args) == olib.orc_rng_f32(*args)
    for i in range(0, 600, 7):
        assert hpt.lib.hpt_halton(i, 2) == olib.orc_halton(i, 2) and hpt.lib.hpt_halton(i, 3) == olib.orc_halton(i, 3)
    rgb = np.random.default_rng(0).uniform(-0.2, 1.5, size=(5, 7, 3)).astype(np.float32)
    import oracle_lib as ol
    want = np.zeros((5, 7, 4), np.uint8)
    olib.orc_tonemap_rgba8(ol._p(rgb, ol.F32P), 35, ol._p(want, ol.U8P))
    np.testing.assert_array_equal(hpt.tonemap_rgba8(rgb), want)


def test_no_gpu_fails_loudly():
    import hpt
    if hpt.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(hpt.HptError, match="no HIP device"):
        hpt.Context(0)


def test_product_does_not_touch_oracle():
    """libhpt.so and the hpt package must not link, load or import anything under oracle/."""
    import subprocess
    lib = os.path.join(ROOT, "path-tracer_amd", "libhpt.so")
    out = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "path-tracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "oracle_lib" not in src and "oracle/" not in src.replace("the CPU oracle", ""), f


def test_rust_ffi_crate_declares_every_export():
    """rust/hpt-sys/src/lib.rs (source only: no Rust toolchain here) binds every function include/hpt.h declares, and
    its #[repr(C)] structs have the header's field counts (sizes are checked against ctypes)"""
    import re
    import hpt
    src = open(os.path.join(ROOT, "rust", "hpt-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (hpt_\w+)\(", src))
    assert declared == set(hpt.EXPORTS)
    hdr = open(os.path.join(ROOT, "include", "hpt.h")).read()
    assert set(re.findall(r"\b(hpt_\w+)\(", hdr)) - {"hpt_scene_add_"} >= declared
    # struct sizes implied by the Rust field lists (all fields 4 or 8 bytes, natural alignment) == ctypes sizes
    sizes = {"hpt_bvh_node": hpt.BvhNode, "hpt_texture": hpt.Texture, "hpt_material": hpt.Material, "hpt_camera": hpt.Camera,
             "hpt_render_params": hpt.RenderParams, "hpt_render_stats": hpt.RenderStats, "hpt_trace_stats": hpt.TraceStats,
             "hpt_scene_info_t": hpt.SceneInfo, "hpt_group_stats": hpt.GroupStats, "hpt_accum": hpt.Accum}
    prim = {"i32": 4, "u32": 4, "f32": 4, "i64": 8, "u64": 8, "f64": 8}
    import ctypes
    known = {}
    for name, body in re.findall(r"pub struct (hpt_\w+)\s*\{(.*?)\n?\}", src, re.S):
        total, align = 0, 4
        for ftype in re.findall(r"pub \w+:\s*([^,}]+)", body):
            ftype = ftype.strip()
            m = re.match(r"\[(\w+); (\d+)\]", ftype)
            if m:
                sz, a = prim[m.group(1)] * int(m.group(2)), prim[m.group(1)]
            elif ftype in prim:
                sz = a = prim[ftype]
            else:
                sz, a = known[ftype]
            total = (total + a - 1) // a * a + sz
            align = max(align, a)
        total = (total + align - 1) // align * align
        known[name] = (total, align)
    for name, cls in sizes.items():
        assert known[name][0] == ctypes.sizeof(cls), name
