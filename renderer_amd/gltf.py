"""Minimal glTF 2.0 reader on top of the renderer API: the part of Renderer/main.cpp:33-418 (tinygltf + stb there)
that turns a `.gltf` / `.glb` file into loadBuffer / loadImage / makeTexture / addMesh calls, plus what the reference
does not read: skins (JOINTS_0 / WEIGHTS_0, inverse bind matrices) and correct key-frame sampling.

Follows the reference where it has a behaviour of its own:
  - every glTF buffer is uploaded whole and attributes are described by (buffer, offset, stride, count, types)
    exactly as main.cpp:353-405 builds VertexAttributeDescription / IndexBufferDescription; the culling box comes
    from the POSITION accessor's min / max (main.cpp:382-385);
  - materials map as main.cpp:135-166: base colour texture sampled as sRGB, the others as UNORM; a missing normal
    texture selects the renderer's default (128,128,255); ao factor = occlusionTexture.strength; alpha mode and
    cut-off as given; a texture's sampler (wrap modes, filters; the reference's loader uses one default sampler) becomes
    the Sampler of makeTexture (VulkanRenderer.cpp:363-385), a minFilter without mip-mapping an image of one level;
  - all channels of all animations are attached to their nodes, with the MIRROR repeat mode main.cpp:260-276 uses.
Departs from it where the reference is wrong or silent: node rotations are quaternions (main.cpp:207-209 reads three
components as Euler angles), interpolation follows the specification (see animation.py), and skins are read.

Images are decoded with PIL (PNG / JPEG). Not supported: sparse accessors, morph targets, cameras, lights,
KHR_* extensions other than emissive strength, primitive modes other than triangle lists."""
import base64
import io
import json
import os
import struct

import numpy as np
from PIL import Image as PILImage

from . import _binding as B
from . import animation as A

COMPONENT = {5120: (B.INT8, np.int8), 5121: (B.UINT8, np.uint8), 5122: (B.INT16, np.int16), 5123: (B.UINT16, np.uint16),
             5125: (B.UINT32, np.uint32), 5126: (B.FLOAT, np.float32)}
CONTAINER = {"SCALAR": (B.SCALAR, 1), "VEC2": (B.VEC2, 2), "VEC3": (B.VEC3, 3), "VEC4": (B.VEC4, 4), "MAT4": (None, 16)}
ATTRIBUTES = {"POSITION": B.ATTR_POSITION, "NORMAL": B.ATTR_NORMAL, "TANGENT": B.ATTR_TANGENT, "TEXCOORD_0": B.ATTR_TEXCOORD_0,
              "JOINTS_0": B.ATTR_JOINTS_0, "WEIGHTS_0": B.ATTR_WEIGHTS_0}
WRAP = {10497: B.WRAP_REPEAT, 33648: B.WRAP_MIRROR, 33071: B.WRAP_CLAMP}
# glTF sampler.minFilter -> (minFilter, mipmapFilter, whether the image gets a mip chain); magFilter 9728 / 9729 -> filter.
# A sampler without the property leaves the choice to the loader: the reference's own, LINEAR everywhere with mips.
MIN_FILTER = {9728: (B.FILTER_NEAREST, B.FILTER_NEAREST, False), 9729: (B.FILTER_LINEAR, B.FILTER_NEAREST, False),
              9984: (B.FILTER_NEAREST, B.FILTER_NEAREST, True), 9985: (B.FILTER_LINEAR, B.FILTER_NEAREST, True),
              9986: (B.FILTER_NEAREST, B.FILTER_LINEAR, True), 9987: (B.FILTER_LINEAR, B.FILTER_LINEAR, True)}
MAG_FILTER = {9728: B.FILTER_NEAREST, 9729: B.FILTER_LINEAR}
ALPHA = {"OPAQUE": B.ALPHA_OPAQUE, "MASK": B.ALPHA_MASK, "BLEND": B.ALPHA_BLEND}


class GltfError(ValueError):
    pass


def _read_container(path):
    """Returns (json dict, embedded binary chunk or None)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] != b"glTF":
            f.seek(0)
            return json.loads(f.read().decode("utf-8")), None
        magic, version, length = struct.unpack("<4sII", head)
        if version != 2:
            raise GltfError(f"GLB version {version} is not supported")
        doc, blob = None, None
        while f.tell() < length:
            chunk = f.read(8)
            if len(chunk) < 8:
                break
            n, kind = struct.unpack("<I4s", chunk)
            data = f.read(n)
            if kind == b"JSON":
                doc = json.loads(data.decode("utf-8"))
            elif kind == b"BIN\0" and blob is None:
                blob = data
        if doc is None:
            raise GltfError("GLB without a JSON chunk")
        return doc, blob


def _resolve_uri(uri, base_dir):
    if uri.startswith("data:"):
        return base64.b64decode(uri.split(",", 1)[1])
    from urllib.parse import unquote
    with open(os.path.join(base_dir, unquote(uri)), "rb") as f:
        return f.read()


class GltfScene:
    """What load_gltf returns: the node graph, the mesh instances it created and a clock."""

    def __init__(self, renderer):
        self.renderer = renderer
        self.nodes, self.roots, self.skins = [], [], []
        self.instances = []     # dicts: mesh (renderer id), node (index), skin (index or None), static (bool)
        self.duration = 0.0
        self.triangles = 0

    def update_graph(self, t=None):
        for n in self.roots:
            if t is not None:
                n.set_animation_time(t)
            n.recalculate_model()

    def set_time(self, t):
        """Node.setAnimationTime + recalculateModel over the roots (Node.cpp:9-38), then this frame's transforms and
        joint palettes are handed to the renderer."""
        self.update_graph(t)
        for inst in self.instances:
            if inst["static"]:
                continue
            model = self.nodes[inst["node"]].model
            self.renderer.set_mesh_transform(inst["mesh"], np.ascontiguousarray(model.T, np.float32))
            if inst["skin"] is not None:
                self.renderer.set_mesh_joints(inst["mesh"], self.skins[inst["skin"]].palette_for_abi(model))


class _Loader:
    def __init__(self, r, path, repeat):
        self.r, self.repeat = r, repeat
        self.dir = os.path.dirname(os.path.abspath(path))
        self.doc, self.blob = _read_container(path)
        if str(self.doc.get("asset", {}).get("version", "2.0")).split(".")[0] != "2":
            raise GltfError("only glTF 2.x assets are supported")
        self.buffers_raw = [self._buffer_bytes(i) for i in range(len(self.doc.get("buffers", [])))]
        self.buffer_ids = {}
        self.image_ids, self.texture_ids = {}, {}

    # -- raw data -------------------------------------------------------------
    def _buffer_bytes(self, i):
        b = self.doc["buffers"][i]
        if "uri" in b:
            data = _resolve_uri(b["uri"], self.dir)
        elif i == 0 and self.blob is not None:
            data = self.blob
        else:
            raise GltfError(f"buffer {i} has no data")
        if len(data) < b["byteLength"]:
            raise GltfError(f"buffer {i}: {len(data)} bytes, {b['byteLength']} declared")
        return data

    def buffer_id(self, i):  # main.cpp:353-355: each buffer is uploaded once, whole
        if i not in self.buffer_ids:
            self.buffer_ids[i] = self.r.load_buffer(np.frombuffer(self.buffers_raw[i], np.uint8))
        return self.buffer_ids[i]

    def accessor_layout(self, index):
        acc = self.doc["accessors"][index]
        if "sparse" in acc or "bufferView" not in acc:
            raise GltfError(f"accessor {index}: sparse / view-less accessors are not supported")
        view = self.doc["bufferViews"][acc["bufferView"]]
        value_type, dtype = COMPONENT[acc["componentType"]]
        container, comps = CONTAINER[acc["type"]]
        offset = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
        return acc, view, value_type, dtype, container, comps, offset

    def accessor_array(self, index):
        """Decoded accessor as [count][components] numpy (for skins, animation tracks and missing min / max)."""
        acc, view, _, dtype, _, comps, offset = self.accessor_layout(index)
        item = np.dtype(dtype).itemsize * comps
        stride = view.get("byteStride", 0) or item
        raw = np.frombuffer(self.buffers_raw[view["buffer"]], np.uint8)
        n = acc["count"]
        if offset + (n - 1) * stride + item > raw.size:
            raise GltfError(f"accessor {index} runs past its buffer")
        rows = np.lib.stride_tricks.as_strided(raw[offset:], shape=(n, item), strides=(stride, 1))
        out = np.ascontiguousarray(rows).view(dtype).reshape(n, comps)
        if acc.get("normalized") and dtype != np.float32:
            info = np.iinfo(dtype)
            out = np.maximum(out.astype(np.float64) / info.max, -1.0)
        return out

    # -- images, textures, materials -----------------------------------------
    def image_id(self, index, srgb, mips=True):
        key = (index, srgb, mips)
        if key not in self.image_ids:
            img = self.doc["images"][index]
            if "uri" in img:
                data = _resolve_uri(img["uri"], self.dir)
            else:
                view = self.doc["bufferViews"][img["bufferView"]]
                o = view.get("byteOffset", 0)
                data = self.buffers_raw[view["buffer"]][o:o + view["byteLength"]]
            pixels = np.asarray(PILImage.open(io.BytesIO(data)).convert("RGBA"))
            self.image_ids[key] = self.r.load_image(pixels, B.FORMAT_R8G8B8A8_SRGB if srgb else B.FORMAT_R8G8B8A8_UNORM,
                                                    0xFFFFFFFF if mips else 1)
        return self.image_ids[key]

    def texture_id(self, info, srgb=False):
        if info is None:
            return B.INVALID_ID
        tex = self.doc["textures"][info["index"]]
        if "source" not in tex:
            return B.INVALID_ID
        smp = self.doc.get("samplers", [])[tex["sampler"]] if "sampler" in tex else {}
        wrap = (WRAP.get(smp.get("wrapS", 10497), B.WRAP_REPEAT), WRAP.get(smp.get("wrapT", 10497), B.WRAP_REPEAT))
        min_filter, mipmap_filter, mips = MIN_FILTER.get(smp.get("minFilter"), (B.FILTER_LINEAR, B.FILTER_LINEAR, True))
        mag_filter = MAG_FILTER.get(smp.get("magFilter"), B.FILTER_LINEAR)
        key = (tex["source"], srgb, wrap, mag_filter, min_filter, mipmap_filter, mips)
        if key not in self.texture_ids:
            self.texture_ids[key] = self.r.make_texture(self.image_id(tex["source"], srgb, mips), wrap[0], wrap[1], mag_filter, min_filter,
                                                        mipmap_filter)
        return self.texture_ids[key]

    def material(self, index):
        import ctypes as C
        m = B.Material()
        src = self.doc["materials"][index] if index is not None else {}
        pbr = src.get("pbrMetallicRoughness", {})
        strength = src.get("extensions", {}).get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0)
        ef = [float(v) * strength for v in src.get("emissiveFactor", [0.0, 0.0, 0.0])]
        m.base_color_factor = (C.c_float * 4)(*[float(v) for v in pbr.get("baseColorFactor", [1, 1, 1, 1])])
        m.emissive_factor = (C.c_float * 4)(ef[0], ef[1], ef[2], 0.0)
        m.normal_scale = float((src.get("normalTexture") or {}).get("scale", 1.0))
        m.metallic_factor = float(pbr.get("metallicFactor", 1.0))
        m.roughness_factor = float(pbr.get("roughnessFactor", 1.0))
        m.ao_factor = float((src.get("occlusionTexture") or {}).get("strength", 1.0))
        m.alpha_mode = ALPHA.get(src.get("alphaMode", "OPAQUE"), B.ALPHA_OPAQUE)
        m.alpha_cutoff = float(src.get("alphaCutoff", 0.5))
        m.albedo_texture = self.texture_id(pbr.get("baseColorTexture"), srgb=True)
        m.normal_texture = self.texture_id(src.get("normalTexture"))
        m.metal_roughness_texture = self.texture_id(pbr.get("metallicRoughnessTexture"))
        m.ao_texture = self.texture_id(src.get("occlusionTexture"))
        m.emissive_texture = self.texture_id(src.get("emissiveTexture"))
        return m

    # -- geometry -------------------------------------------------------------
    def primitive(self, prim):
        if prim.get("mode", 4) != 4:
            raise GltfError("only triangle-list primitives (mode 4) are supported")
        if "POSITION" not in prim["attributes"]:
            raise GltfError("primitive without POSITION")
        attrs = []
        for name, acc_index in prim["attributes"].items():
            if name not in ATTRIBUTES:
                continue
            acc, view, value_type, _, container, _, offset = self.accessor_layout(acc_index)
            attrs.append(B.VertexAttribute(ATTRIBUTES[name], self.buffer_id(view["buffer"]), offset, view.get("byteStride", 0),
                                           container, acc["count"], value_type, 0))
        pos_acc = self.doc["accessors"][prim["attributes"]["POSITION"]]
        if "min" in pos_acc and "max" in pos_acc:
            bb_min, bb_max = pos_acc["min"][:3], pos_acc["max"][:3]
        else:
            p = self.accessor_array(prim["attributes"]["POSITION"])
            bb_min, bb_max = p.min(0), p.max(0)
        indices, n_idx = None, pos_acc["count"]
        if prim.get("indices") is not None:
            acc, view, value_type, _, _, _, offset = self.accessor_layout(prim["indices"])
            if value_type not in (B.UINT8, B.UINT16, B.UINT32):
                raise GltfError("index accessors must be unsigned")
            indices = B.IndexBuffer(self.buffer_id(view["buffer"]), value_type, offset, acc["count"])
            n_idx = acc["count"]
        skinned = "JOINTS_0" in prim["attributes"] and "WEIGHTS_0" in prim["attributes"]
        return {"attributes": attrs, "indices": indices, "bb_min": bb_min, "bb_max": bb_max,
                "material": self.material(prim.get("material"))}, n_idx // 3, skinned

    # -- nodes, animations, skins ----------------------------------------------
    def build_nodes(self, scene):
        for n in self.doc.get("nodes", []):
            if "matrix" in n:
                node = A.Node(matrix=np.asarray(n["matrix"], np.float64).reshape(4, 4).T, name=n.get("name", ""))
            else:
                node = A.Node(n.get("translation", (0, 0, 0)), n.get("rotation", (0, 0, 0, 1)), n.get("scale", (1, 1, 1)), name=n.get("name", ""))
            scene.nodes.append(node)
        for n, node in zip(self.doc.get("nodes", []), scene.nodes):
            node.children = [scene.nodes[c] for c in n.get("children", [])]
        for anim in self.doc.get("animations", []):
            for ch in anim.get("channels", []):
                target = ch.get("target", {})
                path, node_index = target.get("path"), target.get("node")
                if node_index is None or path not in ("translation", "rotation", "scale"):
                    continue
                smp = anim["samplers"][ch["sampler"]]
                times = self.accessor_array(smp["input"]).reshape(-1).astype(np.float64)
                values = self.accessor_array(smp["output"]).astype(np.float64)
                track = A.Animation(times, values, smp.get("interpolation", A.LINEAR), self.repeat, rotation=(path == "rotation"))
                setattr(scene.nodes[node_index], path + "_animation", track)
                scene.duration = max(scene.duration, float(times[-1]))
        for s in self.doc.get("skins", []):
            joints = [scene.nodes[j] for j in s["joints"]]
            ibm = None
            if "inverseBindMatrices" in s:
                ibm = self.accessor_array(s["inverseBindMatrices"]).astype(np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)
            scene.skins.append(A.Skin(joints, ibm))


def load_gltf(renderer, path, scene_index=None, repeat=A.MIRROR):
    """Loads the default (or the given) scene of a glTF 2.0 file into `renderer` and returns a GltfScene."""
    ld = _Loader(renderer, path, repeat)
    doc = ld.doc
    out = GltfScene(renderer)
    ld.build_nodes(out)
    scenes = doc.get("scenes", [])
    if scenes:
        root_indices = scenes[doc.get("scene", 0) if scene_index is None else scene_index].get("nodes", [])
    else:
        children = {c for n in doc.get("nodes", []) for c in n.get("children", [])}
        root_indices = [i for i in range(len(out.nodes)) if i not in children]
    out.roots = [out.nodes[i] for i in root_indices]
    out.update_graph()

    def walk(index, animated_above):
        n, node = doc["nodes"][index], out.nodes[index]
        animated = animated_above or not node.is_static()
        if "mesh" in n:
            prims, skinned_any = [], False
            for p in doc["meshes"][n["mesh"]]["primitives"]:
                desc, tris, skinned = ld.primitive(p)
                prims.append(desc)
                out.triangles += tris
                skinned_any |= skinned
            skin = n.get("skin") if skinned_any else None
            static = not animated and not skinned_any
            mesh_id = renderer.add_mesh(prims, model=np.ascontiguousarray(node.model.T, np.float32), is_static=static)
            out.instances.append({"mesh": mesh_id, "node": index, "skin": skin, "static": static})
        for c in n.get("children", []):
            walk(c, animated)

    for i in root_indices:
        walk(i, False)
    for inst in out.instances:  # bind-pose palettes, so that a skinned mesh is placed by its joints from the first frame
        if inst["skin"] is not None:
            renderer.set_mesh_joints(inst["mesh"], out.skins[inst["skin"]].palette_for_abi(out.nodes[inst["node"]].model))
    return out
