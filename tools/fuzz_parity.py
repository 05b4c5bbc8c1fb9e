"""Randomised parity runs: the HIP library against the CPU checker on scenes nobody designed.

  python tools/fuzz_parity.py [first_seed] [count] [frames]

Every seed makes one scene: a few triangle soups (small, huge, crossing the near plane, slivers, repeated vertices),
some of them skinned,
materials with random alpha modes, wrap modes, filters, image sizes and formats, up to six point lights of which some
cast shadows, a random camera, frame size and sample count. Depth, visibility, cascades and point-light faces must be
equal bit for bit; the 8-bit frame is compared pixel by pixel and the share of pixels further apart than one step is
reported (a discontinuous choice - alpha cut-off, NEAREST texel, mipmap rounding, cascade selection - can fall on
either side in two float32 implementations, so single pixels may differ; more than a few per thousand is a finding).
With more than two frames per seed, a mesh, the camera or the lights change between frames (static point-light maps,
cube slots and cascades are then reused, refreshed or reassigned). Prints one line per seed and exits non-zero at the
first exact mismatch."""
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import orc  # noqa: E402
from renderer_amd import _binding as B  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402


# Diagnosis of pixels further apart than one step: FUZZ_SMOOTH=1 builds the same scenes with every sampler filter LINEAR,
# 2 with MASK materials made OPAQUE, 4 with no roughness texel below 0.25, any sum of them (the same random draws otherwise): what is left then is not a texel or
# cut-off choice falling on either side.
SMOOTH = int(os.environ.get("FUZZ_SMOOTH", "0"))
DUMP = int(os.environ.get("FUZZ_DUMP", "0"))   # print up to this many of a frame's pixels that are further apart than one step
BIG_FROM = 100000   # seeds from here on: larger frames, more primitives, more triangles, sometimes an RGBA16F output
WIDE_FROM = 50000   # seeds from here on: also 1-4 cascades, up to 8 cubes, other gammas, all three exposure modes, orthographic cameras


def soup(rng, kind, n):
    """n random triangles as a Geo with per-vertex frames."""
    if kind == 0:      # pebbles around the origin
        centre = rng.normal(0, 2.0, (n, 1, 3)); size = rng.uniform(0.02, 0.8, (n, 1, 1))
    elif kind == 1:    # large sheets, some through the camera
        centre = rng.normal(0, 4.0, (n, 1, 3)); size = rng.uniform(2.0, 30.0, (n, 1, 1))
    elif kind == 2:    # slivers
        centre = rng.normal(0, 3.0, (n, 1, 3)); size = rng.uniform(0.5, 5.0, (n, 1, 1))
    else:              # dust: sub-pixel triangles far away
        centre = rng.normal(0, 1.0, (n, 1, 3)) + np.array([0, 0, -25.0]); size = rng.uniform(0.001, 0.05, (n, 1, 1))
    tri = centre + rng.normal(0, 1.0, (n, 3, 3)) * size
    if kind == 2:
        tri[:, 2] = tri[:, 0] + (tri[:, 1] - tri[:, 0]) * rng.uniform(0.2, 0.8, (n, 1)) + rng.normal(0, 1e-3, (n, 3))
    if n > 4:
        tri[1] = tri[0]            # the same triangle twice: a depth tie over a whole triangle
        tri[2, 2] = tri[2, 1]      # two equal vertices: no area
    pos = tri.reshape(-1, 3).astype(np.float32)
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    nrm = np.cross(e1, e2)
    ln = np.linalg.norm(nrm, axis=-1, keepdims=True)
    nrm = np.where(ln > 1e-12, nrm / np.maximum(ln, 1e-12), np.array([0.0, 1.0, 0.0]))
    le = np.linalg.norm(e1, axis=-1, keepdims=True)
    tan = np.where(le > 1e-12, e1 / np.maximum(le, 1e-12), np.array([1.0, 0.0, 0.0]))
    nrm = np.repeat(nrm, 3, 0).astype(np.float32)
    tan4 = np.concatenate([np.repeat(tan, 3, 0), np.where(rng.random((3 * n, 1)) < 0.5, -1.0, 1.0)], -1).astype(np.float32)
    uv = (rng.uniform(-3.0, 4.0, (3 * n, 2)) * rng.choice([0.05, 1.0, 8.0])).astype(np.float32)
    return scenes.Geo(pos, nrm, tan4, uv, np.arange(3 * n, dtype=np.uint32))


def random_images(rng):
    """Five images of one random size and a random format family."""
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    if rng.random() < 0.5:
        w = h = int(2 ** rng.integers(0, 7))
    family = rng.choice(["u8", "u8", "u8", "rgb8", "f16", "f32", "rgb32f"])
    imgs = []
    for k in range(5):
        a = rng.random((h, w, 4))
        if k == 1:
            a[..., :3] = a[..., :3] * 0.3 + np.array([0.35, 0.35, 0.7])   # normals that point outwards
        if k == 2 and (SMOOTH & 4):
            a[..., 1] = np.maximum(a[..., 1], 0.25)   # FUZZ_SMOOTH=4: no roughness texel near 0 (8-bit images store 0 for a < 1/510)
        if family == "u8":
            imgs.append((scenes.to_u8(a), B.FORMAT_R8G8B8A8_SRGB if k == 0 else B.FORMAT_R8G8B8A8_UNORM))
        elif family == "rgb8":
            imgs.append((scenes.to_u8(a)[..., :3], B.FORMAT_R8G8B8_SRGB if k == 0 else B.FORMAT_R8G8B8_UNORM))
        elif family == "f16":
            imgs.append((a.astype(np.float16), B.FORMAT_R16G16B16A16_SFLOAT))
        elif family == "f32":
            imgs.append((a.astype(np.float32), B.FORMAT_R32G32B32A32_SFLOAT))
        else:
            imgs.append((a[..., :3].astype(np.float32), B.FORMAT_R32G32B32_SFLOAT))
    return imgs


def build(r, seed, only=None):
    """Returns the meshes that may move: [(mesh id, model matrix, number of joints or 0)]. `only`: a set of primitive
    numbers to keep (diagnosis: the same random draws, the other primitives left out)."""
    rng = np.random.default_rng(seed)
    big = seed >= BIG_FROM
    n_prims = int(rng.integers(8, 40)) if big else int(rng.integers(1, 6))
    movable = []
    for p in range(n_prims):
        geo = soup(rng, int(rng.integers(0, 4)), int(rng.integers(50, 3000)) if big else int(rng.integers(5, 300)))
        imgs = random_images(rng)
        wrap = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        filt = [int(rng.integers(0, 2)) if rng.random() < 0.3 else B.FILTER_LINEAR for _ in range(3)]
        mips = 1 if rng.random() < 0.2 else 0xFFFFFFFF
        if SMOOTH & 1:
            filt = [B.FILTER_LINEAR] * 3
        tex = [r.make_texture(r.load_image(a, f, mips), wrap[0], wrap[1], *filt) for a, f in imgs]
        if rng.random() < 0.2:
            tex[3] = tex[2]    # occlusion and metal-roughness in one image
        if rng.random() < 0.2:
            tex[int(rng.integers(0, 5))] = B.INVALID_ID   # a slot left to the renderer's default
        mode = int(rng.choice([B.ALPHA_OPAQUE, B.ALPHA_OPAQUE, B.ALPHA_MASK, B.ALPHA_BLEND]))
        if (SMOOTH & 2) and mode == B.ALPHA_MASK:
            mode = B.ALPHA_OPAQUE
        mat = scenes.make_material(tex, base_color=tuple(rng.uniform(0.2, 1.0, 4)), emissive=tuple(rng.uniform(0, 0.5, 3) * (rng.random() < 0.5)),
                                   metallic=float(rng.random()), roughness=float(rng.random()), ao=float(rng.uniform(0.5, 1.0)),
                                   alpha_mode=mode, alpha_cutoff=float(rng.uniform(0.2, 0.8)))
        model = scenes.translation(*rng.normal(0, 1.0, 3)) @ scenes.rotation_y(float(rng.uniform(0, 6.28))) @ scenes.rotation_x(float(rng.uniform(0, 6.28)))
        is_static = bool(rng.random() < 0.6)
        skinned = rng.random() < 0.25
        if only is not None and p not in only:
            if skinned:   # the draws a skinned primitive would have made
                n_joints = int(rng.integers(2, 7)); nv = geo.pos.shape[0]
                rng.integers(0, n_joints, (nv, 4)); rng.random((nv, 4)); rng.random((nv, 4)); rng.random()
                rng.choice([B.UINT8, B.UINT16]); rng.choice([B.FLOAT, B.UINT8, B.UINT16]); joint_palette(rng, n_joints)
            continue
        if skinned:
            # a skinned primitive: 2-6 joints, four influences per vertex (some with zero weight, weights not normalised
            # in a tenth of the cases), joint indices and weights in a random storage type
            n_joints = int(rng.integers(2, 7))
            nv = geo.pos.shape[0]
            joints = rng.integers(0, n_joints, (nv, 4))
            weights = rng.random((nv, 4)) * (rng.random((nv, 4)) < 0.7)
            weights[:, 0] += 0.05
            if rng.random() < 0.9:
                weights /= weights.sum(-1, keepdims=True)
            prim = scenes.upload_skinned_primitive(r, geo, joints, weights, mat, joint_type=int(rng.choice([B.UINT8, B.UINT16])),
                                                   weight_type=int(rng.choice([B.FLOAT, B.UINT8, B.UINT16])))
            mesh = r.add_mesh([prim], model=np.ascontiguousarray(model.T, np.float32), is_static=False)
            r.set_mesh_joints(mesh, joint_palette(rng, n_joints))
            movable.append((mesh, model, n_joints))
            continue
        mesh = r.add_mesh([scenes.upload_primitive(r, geo, mat, B.UINT32)], model=np.ascontiguousarray(model.T, np.float32), is_static=is_static)
        if not is_static:
            movable.append((mesh, model, 0))
    if rng.random() < 0.9:
        r.set_brdf_lut(scenes.brdf_lut(16, 16))
        r.set_environment_map(scenes.sky_faces(int(rng.choice([4, 16])), seed=seed))
    lights = [(tuple(rng.normal(0, 3.0, 3)), tuple(rng.uniform(0, 60.0, 3)), bool(rng.random() < 0.6)) for _ in range(int(rng.integers(0, 7)))]
    r.fuzz_casters = 0   # shadow-casting point lights currently set: the cube slots in use are the first min(count, casters)
    if rng.random() < 0.9:
        r.set_lights(lights, scenes.default_directional_light())
        r.fuzz_casters = sum(1 for l in lights if l[2])
    if seed >= WIDE_FROM and rng.random() < 0.3:
        r.set_camera(position=tuple(rng.normal(0, 3.0, 3) + np.array([0, 0, 6.0])), euler=(float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.8, 0.8)), 0.0),
                     near=0.1, far=float(rng.choice([20.0, 60.0])), fov_or_height=float(rng.uniform(2.0, 12.0)), cam_type=B.CAMERA_ORTHOGRAPHIC)
        return movable
    r.set_camera(position=tuple(rng.normal(0, 3.0, 3) + np.array([0, 0, 4.0])), euler=(float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.2, 0.2))),
                 near=float(rng.choice([0.05, 0.1, 0.5])), far=float(rng.choice([20.0, 60.0, 200.0])), fov_or_height=float(np.radians(rng.choice([25.0, 40.0, 75.0]))))
    return movable


def joint_palette(rng, n):
    """n joint matrices near the identity, as the C ABI takes them."""
    out = []
    for _ in range(n):
        m = scenes.translation(*rng.normal(0, 0.3, 3)) @ scenes.rotation_y(float(rng.normal(0, 0.4))) @ scenes.rotation_x(float(rng.normal(0, 0.4)))
        out.append(m.T)
    return np.ascontiguousarray(np.stack(out), np.float32)


def change(r, movable, seed, frame):
    """What happens between two frames of a seed: any of a moved mesh, a moved camera, other lights - or nothing."""
    rng = np.random.default_rng(seed * 1000 + frame)
    if movable and rng.random() < 0.6:
        mesh, model, n_joints = movable[int(rng.integers(0, len(movable)))]
        moved = scenes.translation(*rng.normal(0, 0.5, 3)) @ model @ scenes.rotation_y(float(rng.uniform(0, 1.0)))
        r.set_mesh_transform(mesh, np.ascontiguousarray(moved.T, np.float32))
        if n_joints:
            r.set_mesh_joints(mesh, joint_palette(rng, n_joints))
    if rng.random() < 0.5:
        r.set_camera(position=tuple(rng.normal(0, 3.0, 3) + np.array([0, 0, 4.0])), euler=(float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.8, 0.8)), 0.0))
    if rng.random() < 0.4:
        lights = [(tuple(rng.normal(0, 3.0, 3)), tuple(rng.uniform(0, 60.0, 3)), bool(rng.random() < 0.6)) for _ in range(int(rng.integers(0, 7)))]
        r.set_lights(lights, scenes.default_directional_light())
        r.fuzz_casters = sum(1 for l in lights if l[2])


def draw_settings(seed):
    """Frame size, sample count, settings and frame parameters of a seed."""
    rng = np.random.default_rng(seed + 10 ** 6)
    W, H = int(rng.integers(17, 260)), int(rng.integers(17, 200))
    if seed >= BIG_FROM:
        W, H = int(rng.integers(128, 900)), int(rng.integers(128, 600))
    msaa = int(rng.choice([1, 4]))
    st = dict(msaa=msaa, hdr_output_enabled=int(seed >= BIG_FROM and rng.random() < 0.3), directional_shadow_map_resolution=int(rng.choice([64, 200, 512])), point_shadow_map_resolution=int(rng.choice([16, 50, 128])),
              point_shadow_map_count=int(rng.integers(0, 4)), dynamic_shadows_enabled=int(rng.random() < 0.8))
    if seed >= WIDE_FROM:   # every setting and every frame parameter
        st.update(directional_shadow_cascade_levels=int(rng.integers(1, 5)), point_shadow_map_count=int(rng.integers(0, 9)),
                  gamma=float(rng.choice([1.0, 2.2, 2.6])), exposure_bias=float(rng.uniform(-2, 2)))
    frame_kw = dict(exposure_mode=1, temporal_luminance=0.5)
    if seed >= WIDE_FROM:
        frame_kw = [dict(exposure_mode=0, delta_time=float(rng.uniform(0.001, 0.5))), dict(exposure_mode=1, temporal_luminance=float(rng.uniform(0.01, 4.0))),
                    dict(exposure_mode=2, exposure=float(rng.uniform(-3, 3)))][int(rng.integers(0, 3))]
    return W, H, msaa, st, frame_kw


def one(seed, orc_capi, hip_capi, frames=2):
    W, H, msaa, st, frame_kw = draw_settings(seed)
    ro = Renderer(orc_capi, W, H, default_settings(**st), keep_intermediates=True)
    rh = Renderer(hip_capi, W, H, default_settings(**st), keep_intermediates=True)
    mo, mh = build(ro, seed), build(rh, seed)
    worst = 0.0
    for frame in range(frames):   # the second frame reuses static maps and moves nothing; from the third on things change
        if frame >= 2:
            change(ro, mo, seed, frame); change(rh, mh, seed, frame)
        a, la = ro.render_frame(**frame_kw)
        b, lb = rh.render_frame(**frame_kw)
        exact = [("depth", ro.read_depth(), rh.read_depth()), ("visibility", ro.read_visibility(), rh.read_visibility())]
        readers = [(f"cascade {i}", lambda r, i=i: r.read_directional_shadow(i)) for i in range(4)]
        readers += [(f"point {slot}/{face}", lambda r, slot=slot, face=face: r.read_point_shadow(slot, face))
                    for slot in range(min(st["point_shadow_map_count"], ro.fuzz_casters)) for face in range(6)]
        for name, read in readers:
            got = []
            for r in (ro, rh):   # a map that does not exist (no lights set, a slot no light has taken) cannot be read
                try:
                    got.append(read(r))
                except Exception:
                    got.append(None)
            # (after lights without casters the checker has dropped its cubes and the library still holds the old ones:
            # nothing in the frame reads them, so only maps both sides hold are compared)
            if got[0] is not None and got[1] is not None:
                exact.append((name, got[0], got[1]))
        for name, x, y in exact:
            if not np.array_equal(x, y):
                print(f"seed {seed} frame {frame}: {name} differs in {int((x != y).sum())} of {x.size} values ({W}x{H} msaa {msaa} {st})")
                return False, 1.0
        so, sh = ro.frame_stats(), rh.frame_stats()
        # (only the first frame: the library does not redraw shadow views that nothing changed, the checker does)
        if frame == 0 and (so.triangle_instances, so.triangles_binned, so.views) != (sh.triangle_instances, sh.triangles_binned, sh.views):
            print(f"seed {seed} frame {frame}: counters differ {so.triangle_instances, so.triangles_binned, so.views} / {sh.triangle_instances, sh.triangles_binned, sh.views}")
            return False, 1.0
        if a.dtype == np.float16:   # an RGBA16F output image: in steps of 1/255 like the 8-bit one
            a, b = np.nan_to_num(a.astype(np.float32)) * 255.0, np.nan_to_num(b.astype(np.float32)) * 255.0
        diff = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(-1)
        if float((diff > 1).mean()) > 0.001:
            # A fragment whose colour is not finite on one side only: the soups give every vertex a random bitangent sign, so
            # the interpolated frame degenerates along a line inside a triangle, and a pixel centre that lands on it exactly
            # normalises a zero vector (undefined in GLSL; NaN here, on whichever side rounds to exactly zero). The bloom
            # chain spreads it over a 252-pixel square; such a frame says nothing about the other pixels and is left out.
            nf = [int((~np.isfinite(r.read_hdr()).all(-1)).sum()) for r in (ro, rh)]
            if nf[0] or nf[1]:
                print(f"  seed {seed} frame {frame}: pixels with a non-finite linear colour: checker {nf[0]}, library {nf[1]}; {int((diff > 1).sum())} pixels of the "
                      f"output apart (the bloom of a non-finite pixel); frame left out of the count")
                continue
            print(f"  seed {seed} frame {frame}: {int((diff > 1).sum())} of {diff.size} pixels apart, largest step {int(diff.max())}; avg log luminance {la} / {lb}; "
                  f"exposure {ro.frame_stats().exposure} / {rh.frame_stats().exposure}; {W}x{H} msaa {msaa} {st}")
        worst = max(worst, float((diff > 1).mean()))
        if DUMP and (diff > 1).any():   # the pixels themselves: both outputs, both linear colours, who covers them
            ho, hh = ro.read_hdr().astype(np.float64), rh.read_hdr().astype(np.float64)
            vis = ro.read_visibility()
            ys, xs = np.nonzero(diff > 1)
            print(f"  seed {seed} frame {frame}: {ys.size} pixels apart ({W}x{H} msaa {msaa} {st} {frame_kw}); pixels with a non-finite linear colour: "
                  f"{int((~np.isfinite(ho).all(-1)).sum())} / {int((~np.isfinite(hh).all(-1)).sum())}")
            for name, img, other in (("checker", ho, hh), ("library", hh, ho)):
                for y, x in list(zip(*np.nonzero(~np.isfinite(img).all(-1))))[:DUMP]:
                    print(f"    {name} non-finite at ({x},{y}): {img[y, x, :3].tolist()}, the other side {np.round(other[y, x, :3], 4).tolist()}, triangles {vis[:, y, x].tolist()}")
            for y, x in list(zip(ys, xs))[:DUMP]:
                print(f"    ({x},{y}) out {a[y, x].tolist()} / {b[y, x].tolist()}  linear {np.round(ho[y, x, :3], 4).tolist()} / {np.round(hh[y, x, :3], 4).tolist()}  triangles {vis[:, y, x].tolist()}")
    ro.close(); rh.close()
    return True, worst


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    orc_capi = orc.capi()
    hip_capi = orc_capi if os.environ.get("FUZZ_SELF") else load_hip_capi()  # FUZZ_SELF=1: the checker against itself, to try the script without a GPU
    off = []
    for seed in range(first, first + count):
        ok, share = one(seed, orc_capi, hip_capi, frames)
        if not ok:
            sys.exit(1)
        off.append(share)
        print(f"seed {seed}: exact images equal, {share * 1e3:.2f} per thousand pixels of the frame further apart than one step", flush=True)
    print(f"{count} scenes: worst {max(off) * 1e3:.2f} per thousand, mean {np.mean(off) * 1e3:.3f}")


if __name__ == "__main__":
    main()
