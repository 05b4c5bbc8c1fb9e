"""The C-ABI boundary (include/rdr.h): both libraries export every declared entry point, and the ctypes
mirror of the POD structs has the layout the C compiler gives them. No compute call is made here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from renderer_amd import _binding as B
from renderer_amd.api import HIP_LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rdr.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds():
    names = declared_functions()
    assert len(names) >= 29
    assert sorted("rdr_" + n for n in list(B.SIGNATURES) + list(B.HIP_ONLY_SIGNATURES)) == names
    assert f"#define RDR_NOALLOC_MAX_PIXELS (1920u * 1088u)" in open(HEADER).read() and B.NOALLOC_MAX_PIXELS == 1920 * 1088


def test_hip_library_exports_every_entry_point():
    assert os.path.exists(HIP_LIB_PATH), "build the HIP library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(HIP_LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/rdr.h but not exported by {HIP_LIB_PATH}"
    api = B.CApi(HIP_LIB_PATH, "rdr_")
    assert api.backend_name() == b"hip:gfx950"
    s = B.Settings()
    assert api.settings_default(C.byref(s)) == 0
    # RendererSettings{} defaults (Renderer/VulkanRenderer.h:44-59)
    assert (s.dynamic_shadows_enabled, s.point_shadow_map_resolution, s.point_shadow_map_count) == (1, 512, 8)
    assert (s.directional_shadow_map_resolution, s.directional_shadow_cascade_levels, s.msaa) == (2048, 4, 4)
    assert s.exposure_bias == 0.0 and abs(s.gamma - 2.2) < 1e-7


def test_checker_library_exports_the_same_abi(orc_capi):
    for name in B.SIGNATURES:
        assert hasattr(orc_capi.lib, "orc_" + name)
    assert orc_capi.backend_name() == b"cpu:oracle"


def test_hip_library_fails_loudly_without_a_device():
    """On a machine without a GPU rdr_create must return an error, never fall back to another path."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    api = B.CApi(HIP_LIB_PATH, "rdr_")
    s = B.Settings()
    api.settings_default(C.byref(s))
    ctx = C.c_void_p()
    rc = api.create(C.byref(s), 64, 64, 0, C.byref(ctx))
    assert rc != 0 and not ctx.value


STRUCTS = {
    "rdr_settings": B.Settings, "rdr_sampler": B.Sampler, "rdr_vertex_attribute": B.VertexAttribute,
    "rdr_index_buffer": B.IndexBuffer, "rdr_material": B.Material, "rdr_primitive_desc": B.PrimitiveDesc,
    "rdr_point_light": B.PointLight, "rdr_directional_light": B.DirectionalLight, "rdr_camera": B.Camera,
    "rdr_frame_params": B.FrameParams, "rdr_frame_stats": B.FrameStats,
}


def test_struct_layout_matches_the_c_compiler(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in STRUCTS.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for cname, cls in STRUCTS.items():
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, f"{cname}.{fname}"
