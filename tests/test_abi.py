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


def _refusals(api):
    """What the settings / resource calls refuse, on either library: (description, status) pairs."""
    out = []

    def create(**kw):
        s = B.Settings()
        api.settings_default(C.byref(s))
        for k, v in kw.items():
            setattr(s, k, v)
        ctx = C.c_void_p()
        rc = api.create(C.byref(s), 64, 64, 0, C.byref(ctx))
        if rc == 0:
            api.destroy(ctx)
        return rc
    out.append(("hdr output 2", create(hdr_output_enabled=2)))
    out.append(("msaa 2", create(msaa=2)))
    out.append(("5 cascades", create(directional_shadow_cascade_levels=5)))
    out.append(("cascade resolution 0", create(directional_shadow_map_resolution=0)))
    out.append(("point resolution 0", create(point_shadow_map_resolution=0)))
    out.append(("cascade resolution 65536", create(directional_shadow_map_resolution=65536)))
    out.append(("point resolution 16384", create(point_shadow_map_resolution=16384)))
    out.append(("1000 point shadow maps", create(point_shadow_map_count=1000)))
    return out


EXPECTED_REFUSALS = [("hdr output 2", B.RDR_ERR_INVALID), ("msaa 2", B.RDR_ERR_UNSUPPORTED), ("5 cascades", B.RDR_ERR_UNSUPPORTED),
                     ("cascade resolution 0", B.RDR_ERR_INVALID), ("point resolution 0", B.RDR_ERR_INVALID),
                     ("cascade resolution 65536", B.RDR_ERR_UNSUPPORTED), ("point resolution 16384", B.RDR_ERR_UNSUPPORTED),
                     ("1000 point shadow maps", B.RDR_ERR_UNSUPPORTED)]


def test_settings_the_checker_refuses(orc_capi):
    """RendererSettings values neither library implements are refused by rdr_create with a status, never silently ignored
    (VulkanRenderer.h:44-59: the shadow-map sizes and counts, the sample count)."""
    assert _refusals(orc_capi) == EXPECTED_REFUSALS


def test_resources_the_checker_refuses(orc_capi):
    """Every ImageFormat of Image.h:5-14 and every combination of the Sampler's filters (VulkanRenderer.cpp:363-385) is
    accepted; values outside the enums and buffers too small for the format are invalid."""
    _check_resource_refusals(orc_capi)


def _check_resource_refusals(api):
    import numpy as np

    from renderer_amd.api import Renderer, RendererError, default_settings
    r = Renderer(api, 32, 32, default_settings(msaa=1))
    px = np.zeros((4, 4, 4), np.uint8)
    for fmt in (B.FORMAT_R8G8B8A8_UNORM, B.FORMAT_R8G8B8A8_SRGB):
        r.load_image(px, fmt)
    for fmt in (B.FORMAT_R8G8B8_UNORM, B.FORMAT_R8G8B8_SRGB):
        r.load_image(px[..., :3], fmt)
    for fmt, texel_bytes in ((4, 6), (5, 8), (6, 12), (7, 16)):  # the 16- and 32-bit float formats of Image.h
        out = C.c_uint32()
        buf = np.zeros(4 * 4 * texel_bytes, np.uint8)
        assert api.load_image(r._ctx, buf.ctypes.data_as(C.c_void_p), buf.size, 4, 4, fmt, 1, C.byref(out)) == B.RDR_OK, fmt
        assert api.load_image(r._ctx, buf.ctypes.data_as(C.c_void_p), buf.size - 1, 4, 4, fmt, 1, C.byref(out)) == B.RDR_ERR_INVALID, fmt
    out = C.c_uint32()
    buf = np.zeros(4 * 4 * 16, np.uint8)
    assert api.load_image(r._ctx, buf.ctypes.data_as(C.c_void_p), buf.size, 4, 4, 8, 1, C.byref(out)) == B.RDR_ERR_INVALID
    img = r.load_image(px, B.FORMAT_R8G8B8A8_UNORM)
    for bits in range(8):
        filt = [B.FILTER_NEAREST if bits >> k & 1 else B.FILTER_LINEAR for k in range(3)]
        s = B.Sampler(filt[0], filt[1], filt[2], B.WRAP_REPEAT, B.WRAP_REPEAT)
        out = C.c_uint32()
        assert api.make_texture(r._ctx, img, C.byref(s), C.byref(out)) == B.RDR_OK
    for k in range(3):
        filt = [B.FILTER_LINEAR] * 3
        filt[k] = 2
        s = B.Sampler(filt[0], filt[1], filt[2], B.WRAP_REPEAT, B.WRAP_REPEAT)
        out = C.c_uint32()
        assert api.make_texture(r._ctx, img, C.byref(s), C.byref(out)) == B.RDR_ERR_INVALID
    s = B.Sampler(B.FILTER_LINEAR, B.FILTER_LINEAR, B.FILTER_LINEAR, 7, B.WRAP_REPEAT)
    out = C.c_uint32()
    assert api.make_texture(r._ctx, img, C.byref(s), C.byref(out)) == B.RDR_ERR_INVALID
    r.close()


@pytest.mark.gpu
def test_settings_and_resources_the_hip_library_refuses(hip_capi):
    assert _refusals(hip_capi) == EXPECTED_REFUSALS
    _check_resource_refusals(hip_capi)


def test_gather_library_exports_what_its_header_declares():
    """include/rdr_gather.h (the multi-GPU exchange step for C / C++ hosts): renderer_amd/lib/librdr_gather.so loads
    without a GPU (RCCL is only linked) and exports every function the header declares."""
    import ctypes
    text = open(os.path.join(ROOT, "include", "rdr_gather.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(rdr_gather_[a-z0-9_]+)\s*\(", text)))
    assert names == ["rdr_gather_all_floats", "rdr_gather_create", "rdr_gather_destroy", "rdr_gather_frames", "rdr_gather_last_error",
                     "rdr_gather_make_id", "rdr_gather_wait"]
    lib = ctypes.CDLL(os.path.join(ROOT, "renderer_amd", "lib", "librdr_gather.so"))
    for n in names:
        assert hasattr(lib, n), n
    # argument checks that need no device
    assert lib.rdr_gather_make_id(None) == 1
    out = ctypes.c_void_p()
    assert lib.rdr_gather_create(None, 0, 1, 0, ctypes.byref(out)) == 1
