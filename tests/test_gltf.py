"""glTF 2.0 reader (the part of Renderer/main.cpp:33-418 that feeds the renderer), checked by round trip: a scene
is built once through the API directly and once by writing it to .gltf / .glb and loading that; both must hand
the renderer the same bytes, so frames and skinned positions are compared exactly. The named sample assets are not on
disk and the reference's loader (tinygltf + stb) is not buildable here, so there is nothing of the reference to
compare with; material mapping follows main.cpp:135-166 as cited in the reader."""
import io
import json

import numpy as np
import pytest
from PIL import Image

from renderer_amd import _binding as B
from renderer_amd import animation as A
from renderer_amd import gltf, scenes
from renderer_amd.api import Renderer, default_settings
from tests.gltf_writer import GltfWriter

FRAME = dict(exposure_mode=1, temporal_luminance=0.5)
F32 = lambda a: np.asarray(a, np.float32).astype(np.float64)  # noqa: E731  what survives a float32 accessor


def new_renderer(capi, W=160, H=120):
    r = Renderer(capi, W, H, default_settings(msaa=1, directional_shadow_map_resolution=256, point_shadow_map_resolution=64))
    r.set_brdf_lut(scenes.brdf_lut(32, 32))
    r.set_environment_map(scenes.sky_faces(16))
    r.set_camera(position=(0.4, 1.7, 5.2), euler=(-0.2, 0.05, 0.0))
    r.set_lights([((1.5, 2.2, 1.5), (30.0, 24.0, 18.0), True), ((-2.0, 1.8, 2.0), (10.0, 14.0, 22.0), False)], scenes.default_directional_light())
    return r


class Room:
    """One description, two ways to build it."""

    def __init__(self):
        self.floor = scenes.plane(12, 12, 4, 4, uv_scale=4.0)
        self.box = scenes.box(1.0, 1.5, 1.0, 2)
        self.sphere = scenes.uv_sphere(0.6, 24, 12)
        self.quad = scenes.plane(1.4, 1.4, 1, 1)
        self.mat_floor = scenes.material_images(32, 11, tint=(0.9, 0.8, 0.7))
        self.mat_box = scenes.material_images(32, 21, tint=(0.7, 0.8, 1.0), metal_fraction=0.5)
        alpha = scenes.value_noise(32, 77, octaves=2, base=4)
        self.mat_sphere = scenes.material_images(32, 41, tint=(0.5, 1.0, 0.5), alpha=(alpha > 0.5).astype(np.float32))
        # the sphere's normal map travels as JPEG: both paths use the decoded pixels
        buf = io.BytesIO()
        Image.fromarray(self.mat_sphere[1][..., :3]).save(buf, "JPEG", quality=92)
        self.sphere_normal_jpeg = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"))
        self.parent_trs = ((-1.8, 0.0, -0.5), A.quat_from_axis_angle((0, 1, 0), 0.4), (1.0, 1.2, 1.0))
        self.child_matrix = scenes.translation(0.3, 0.75, 0.2) @ scenes.rotation_y(0.3)
        self.sphere_t = ([0.0, 1.0, 2.0], F32([[0.0, 0.9, 0.4], [0.6, 1.4, 0.2], [0.0, 0.9, 0.4]]))
        self.sphere_r = ([0.0, 1.0], F32([A.quat_from_axis_angle((0, 1, 0), 0.0), A.quat_from_axis_angle((0, 1, 0), 1.2)]))
        self.quad_model = scenes.translation(1.0, 0.9, 2.0) @ scenes.rotation_x(np.pi / 2)

    def graph(self):
        child = A.Node(matrix=self.child_matrix)
        parent = A.Node(*self.parent_trs, children=[child])
        sphere = A.Node(translation=self.sphere_t[1][0], rotation=self.sphere_r[1][0])
        sphere.translation_animation = A.Animation(self.sphere_t[0], self.sphere_t[1], A.LINEAR, A.MIRROR)
        sphere.rotation_animation = A.Animation(self.sphere_r[0], self.sphere_r[1], A.STEP, A.MIRROR, rotation=True)
        return parent, child, sphere

    def build_direct(self, r):
        def textures(imgs, wrap=(B.WRAP_REPEAT, B.WRAP_REPEAT), filters=(B.FILTER_LINEAR, B.FILTER_LINEAR, B.FILTER_LINEAR), mips=True):
            fmts = [B.FORMAT_R8G8B8A8_SRGB] + [B.FORMAT_R8G8B8A8_UNORM] * 4
            return [r.make_texture(r.load_image(i, f, 0xFFFFFFFF if mips else 1), *wrap, *filters) for i, f in zip(imgs, fmts)]

        parent, child, sphere = self.graph()
        for n in (parent, sphere):
            n.recalculate_model()
        # the floor's sampler: magFilter NEAREST, minFilter LINEAR without mip-mapping (9728 / 9729): an image of one level
        floor_tex = textures(self.mat_floor, filters=(B.FILTER_NEAREST, B.FILTER_LINEAR, B.FILTER_NEAREST), mips=False)
        r.add_mesh([scenes.upload_primitive(r, self.floor, scenes.make_material(floor_tex), B.UINT16)])
        # the box's: magFilter LINEAR, minFilter NEAREST_MIPMAP_LINEAR (9729 / 9986)
        box_tex = textures(self.mat_box, (B.WRAP_MIRROR, B.WRAP_CLAMP), filters=(B.FILTER_LINEAR, B.FILTER_NEAREST, B.FILTER_LINEAR))
        r.add_mesh([scenes.upload_primitive(r, self.box, scenes.make_material(box_tex, ao=0.7), B.UINT16)],
                   model=np.float32(child.model.T))
        imgs = list(self.mat_sphere)
        imgs[1] = self.sphere_normal_jpeg
        sph = r.add_mesh([scenes.upload_primitive(r, self.sphere, scenes.make_material(textures(imgs), alpha_mode=B.ALPHA_MASK, alpha_cutoff=0.4,
                                                                                        roughness=0.6), B.UINT32)],
                         model=np.float32(sphere.model.T), is_static=False)
        r.add_mesh([scenes.upload_primitive(r, self.quad, scenes.make_material(None, base_color=(0.4, 0.6, 1.0, 0.5), emissive=(0.2 * 3, 0.1 * 3, 0.0),
                                                                               metallic=0.0, roughness=0.8, alpha_mode=B.ALPHA_BLEND), B.UINT16)],
                   model=np.float32(self.quad_model.T))

        def set_time(t):
            sphere.set_animation_time(t)
            sphere.recalculate_model()
            r.set_mesh_transform(sph, np.float32(sphere.model.T))
        return set_time

    def write(self, path, glb=False, where="file"):
        w = GltfWriter()

        def textures(imgs, name, wrap=(10497, 10497), jpeg_slot=None, filters=(9729, 9987)):
            out = []
            for k, img in enumerate(imgs):
                i, _ = w.image(img, f"{name}_{k}", where=where, fmt="JPEG" if k == jpeg_slot else "PNG")
                out.append(w.texture(i, *wrap, *filters))
            return out

        m_floor = w.material(textures(self.mat_floor, "floor", filters=(9728, 9729)))
        m_box = w.material(textures(self.mat_box, "box", (33648, 33071), filters=(9729, 9986)), ao=0.7)
        m_sph = w.material(textures(self.mat_sphere, "sphere", jpeg_slot=1), alpha_mode="MASK", alpha_cutoff=0.4, roughness=0.6)
        m_quad = w.material(None, base_color=(0.4, 0.6, 1.0, 0.5), emissive=(0.2, 0.1, 0.0), emissive_strength=3.0, metallic=0.0, roughness=0.8,
                            alpha_mode="BLEND")
        w.node(mesh=w.mesh([w.primitive(self.floor, m_floor, np.uint16)]), root=True, name="floor")
        child = w.node(mesh=w.mesh([w.primitive(self.box, m_box, np.uint16, interleave=True)]), matrix=self.child_matrix)
        w.node(translation=self.parent_trs[0], rotation=self.parent_trs[1], scale=self.parent_trs[2], children=[child], root=True)
        sph = w.node(mesh=w.mesh([w.primitive(self.sphere, m_sph, np.uint32)]), translation=self.sphere_t[1][0], rotation=self.sphere_r[1][0], root=True)
        w.node(mesh=w.mesh([w.primitive(self.quad, m_quad, np.uint16)]), matrix=self.quad_model, root=True)
        w.animation([(sph, "translation", self.sphere_t[0], self.sphere_t[1], "LINEAR"), (sph, "rotation", self.sphere_r[0], self.sphere_r[1], "STEP")])
        (w.save_glb if glb else w.save_gltf)(path)
        return w


@pytest.mark.parametrize("container", ["gltf+files", "glb+embedded", "gltf+datauri"])
def test_room_round_trip(orc_capi, tmp_path, container):
    room = Room()
    ra, rb = new_renderer(orc_capi), new_renderer(orc_capi)
    set_time_a = room.build_direct(ra)
    if container == "gltf+files":
        path = str(tmp_path / "room.gltf"); room.write(path)
    elif container == "glb+embedded":
        path = str(tmp_path / "room.glb"); room.write(path, glb=True, where="embedded")
    else:
        path = str(tmp_path / "room.gltf"); w = room.write(path, where="datauri"); w.save_gltf(path, bin_as_datauri=True)
    scene = gltf.load_gltf(rb, path)
    assert scene.triangles == sum(g.triangles for g in (room.floor, room.box, room.sphere, room.quad))
    assert [i["static"] for i in scene.instances] == [True, True, False, True] and scene.duration == 2.0
    for t in (0.0, 0.7, 2.6):   # 2.6 mirrors back to 1.4
        set_time_a(t); scene.set_time(t)
        a, la = ra.render_frame(**FRAME)
        b, lb = rb.render_frame(**FRAME)
        assert np.array_equal(ra.read_depth(), rb.read_depth()) and np.array_equal(a, b) and la == lb, t
    ra.close(); rb.close()


def test_skinned_figure_round_trip(orc_capi, tmp_path):
    ra, rb = new_renderer(orc_capi), new_renderer(orc_capi)
    fig = scenes.SkinnedFigure(ra, origin=(0.8, 0.0, 1.5), tex_size=32)
    w = GltfWriter()
    tex = [w.texture(w.image(img, f"fig_{k}", where="embedded")[0]) for k, img in enumerate(scenes.material_images(**fig.material_args))]
    mat = w.material(tex)
    prims = []
    for geo, joints, weights, jt, wt in fig.parts:
        prims.append(w.primitive(geo, mat, np.uint32, joints=joints.astype(np.uint8 if jt == B.UINT8 else np.uint16), weights=weights,
                                 weights_normalized_dtype=np.uint8 if wt == B.UINT8 else None))
    bones = []
    for k in reversed(range(len(fig.nodes))):
        bones.insert(0, w.node(translation=fig.nodes[k].translation, children=bones[:1], name=f"bone{k}"))
    channels = []
    for k, n in enumerate(fig.nodes):
        an = n.rotation_animation
        channels.append((bones[k], "rotation", an.times, an.values, an.interpolation))
    sa = fig.nodes[-1].scale_animation
    channels.append((bones[-1], "scale", sa.times, sa.values, sa.interpolation))
    w.animation(channels)
    skin = w.skin(bones, fig.skin.inverse_bind)
    w.node(mesh=w.mesh(prims), skin=skin, matrix=fig.model, children=[bones[0]], root=True, name="figure")
    path = str(tmp_path / "figure.glb")
    w.save_glb(path)

    scene = gltf.load_gltf(rb, path, repeat=A.REPEAT)
    assert len(scene.skins) == 1 and scene.instances[0]["skin"] == 0 and not scene.instances[0]["static"]
    for t in (0.0, 0.3, 0.95, 1.6):   # inside the clip: the figure's own tracks mix repeat modes
        fig.pose(t); scene.set_time(t)
        for prim in (0, 1):
            assert np.array_equal(ra.read_primitive_positions(fig.mesh, prim), rb.read_primitive_positions(scene.instances[0]["mesh"], prim)), (t, prim)
        a, _ = ra.render_frame(**FRAME)
        b, _ = rb.render_frame(**FRAME)
        assert np.array_equal(a, b), t
    ra.close(); rb.close()


def test_reader_rejects_what_it_does_not_support(orc_capi, tmp_path):
    room = Room()
    path = str(tmp_path / "room.gltf")
    room.write(path)
    doc = json.load(open(path))
    r = new_renderer(orc_capi)
    for mutate, what in ((lambda d: d["meshes"][0]["primitives"][0].update(mode=1), "lines"),
                         (lambda d: d["accessors"][0].update(sparse={"count": 1}), "sparse"),
                         (lambda d: d["asset"].update(version="1.0"), "version"),
                         (lambda d: d["buffers"][0].update(byteLength=10 ** 9), "short buffer")):
        bad = json.loads(json.dumps(doc))
        mutate(bad)
        p = str(tmp_path / f"bad_{what.replace(' ', '_')}.gltf")
        json.dump(bad, open(p, "w"))
        with pytest.raises(gltf.GltfError):
            gltf.load_gltf(r, p)
    r.close()


@pytest.mark.gpu
def test_gltf_scene_on_the_hip_library(orc_capi, hip_capi, tmp_path):
    from tests import helpers as hp
    room = Room()
    path = str(tmp_path / "room.glb")
    room.write(path, glb=True, where="embedded")
    ro, rh = new_renderer(orc_capi, 320, 180), new_renderer(hip_capi, 320, 180)
    so, sh = gltf.load_gltf(ro, path), gltf.load_gltf(rh, path)
    for t in (0.0, 1.3):
        so.set_time(t); sh.set_time(t)
        a, _ = ro.render_frame(**FRAME)
        b, _ = rh.render_frame(**FRAME)
        assert np.array_equal(ro.read_depth(), rh.read_depth()) and np.array_equal(ro.read_visibility(), rh.read_visibility())
        hp.assert_image_close(a, b)
    ro.close(); rh.close()
