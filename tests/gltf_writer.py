"""Test-side glTF 2.0 writer: serialises the synthetic scenes so that the reader can be checked by round trip
(the named sample assets — DamagedHelmet, Sponza, CesiumMan, Fox — are not on disk)."""
import base64
import io
import json
import os
import struct

import numpy as np
from PIL import Image

COMPONENT = {np.dtype(np.int8): 5120, np.dtype(np.uint8): 5121, np.dtype(np.int16): 5122, np.dtype(np.uint16): 5123,
             np.dtype(np.uint32): 5125, np.dtype(np.float32): 5126}
TYPE = {1: "SCALAR", 2: "VEC2", 3: "VEC3", 4: "VEC4", 16: "MAT4"}


class GltfWriter:
    def __init__(self):
        self.doc = {"asset": {"version": "2.0", "generator": "tests/gltf_writer.py"}, "scene": 0, "scenes": [{"nodes": []}], "nodes": [],
                    "meshes": [], "materials": [], "textures": [], "images": [], "samplers": [], "accessors": [], "bufferViews": [],
                    "buffers": [], "animations": [], "skins": []}
        self.bin = bytearray()
        self.files = {}  # name -> bytes, written next to the .gltf

    # -- binary data ------------------------------------------------------------
    def view(self, data, stride=None, target=None):
        self.bin += b"\0" * (-len(self.bin) % 4)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        self.bin += data
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, minmax=False, normalized=False, view=None, byte_offset=0):
        a = np.ascontiguousarray(array)
        if a.ndim == 1:
            a = a[:, None]
        comps = a.shape[1]
        if view is None:
            view = self.view(a.tobytes())
        acc = {"bufferView": view, "componentType": COMPONENT[a.dtype], "count": int(a.shape[0]), "type": TYPE[comps]}
        if byte_offset:
            acc["byteOffset"] = byte_offset
        if normalized:
            acc["normalized"] = True
        if minmax:
            acc["min"], acc["max"] = [float(x) for x in a.min(0)], [float(x) for x in a.max(0)]
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def interleaved(self, arrays):
        """Packs float32 arrays of one vertex count into one strided buffer view; returns their accessors."""
        arrays = [np.ascontiguousarray(a, np.float32) for a in arrays]
        packed = np.concatenate(arrays, 1)
        stride = packed.shape[1] * 4
        v = self.view(packed.tobytes(), stride=stride, target=34962)
        out, off = [], 0
        for i, a in enumerate(arrays):
            out.append(self.accessor(a, minmax=(i == 0), view=v, byte_offset=off))
            off += a.shape[1] * 4
        return out

    # -- images / materials -------------------------------------------------------
    def image(self, pixels, name, where="file", fmt="PNG"):
        buf = io.BytesIO()
        img = Image.fromarray(pixels if fmt == "PNG" else pixels[..., :3])
        img.save(buf, fmt, **({"quality": 92} if fmt == "JPEG" else {}))
        data = buf.getvalue()
        mime = "image/png" if fmt == "PNG" else "image/jpeg"
        if where == "file":
            fname = f"{name}.{'png' if fmt == 'PNG' else 'jpg'}"
            self.files[fname] = data
            entry = {"uri": fname}
        elif where == "datauri":
            entry = {"uri": f"data:{mime};base64," + base64.b64encode(data).decode()}
        else:
            entry = {"bufferView": self.view(data), "mimeType": mime}
        self.doc["images"].append(entry)
        return len(self.doc["images"]) - 1, data

    def texture(self, image, wrap_s=10497, wrap_t=10497, mag_filter=9729, min_filter=9987):
        self.doc["samplers"].append({"magFilter": mag_filter, "minFilter": min_filter, "wrapS": wrap_s, "wrapT": wrap_t})
        self.doc["textures"].append({"source": image, "sampler": len(self.doc["samplers"]) - 1})
        return len(self.doc["textures"]) - 1

    def material(self, textures=None, base_color=(1, 1, 1, 1), emissive=(0, 0, 0), metallic=1.0, roughness=1.0, ao=1.0,
                 alpha_mode="OPAQUE", alpha_cutoff=0.5, emissive_strength=None):
        """textures: five glTF texture indices (albedo, normal, metalRough, ao, emissive) or None entries."""
        t = textures or [None] * 5
        m = {"pbrMetallicRoughness": {"baseColorFactor": list(map(float, base_color)), "metallicFactor": float(metallic),
                                      "roughnessFactor": float(roughness)},
             "emissiveFactor": list(map(float, emissive)), "alphaMode": alpha_mode, "alphaCutoff": float(alpha_cutoff)}
        if t[0] is not None:
            m["pbrMetallicRoughness"]["baseColorTexture"] = {"index": t[0]}
        if t[1] is not None:
            m["normalTexture"] = {"index": t[1], "scale": 1.0}
        if t[2] is not None:
            m["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": t[2]}
        if t[3] is not None:
            m["occlusionTexture"] = {"index": t[3], "strength": float(ao)}
        if t[4] is not None:
            m["emissiveTexture"] = {"index": t[4]}
        if emissive_strength is not None:
            m["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": float(emissive_strength)}}
        self.doc["materials"].append(m)
        return len(self.doc["materials"]) - 1

    # -- geometry / nodes -----------------------------------------------------------
    def primitive(self, geo, material, index_dtype=np.uint32, interleave=False, joints=None, weights=None, weights_normalized_dtype=None):
        if interleave:
            pos, nrm, tan, uv = self.interleaved([geo.pos, geo.nrm, geo.tan, geo.uv])
        else:
            pos, nrm, tan, uv = (self.accessor(geo.pos, minmax=True), self.accessor(geo.nrm), self.accessor(geo.tan), self.accessor(geo.uv))
        attrs = {"POSITION": pos, "NORMAL": nrm, "TANGENT": tan, "TEXCOORD_0": uv}
        if joints is not None:
            attrs["JOINTS_0"] = self.accessor(joints)
            if weights_normalized_dtype is None:
                attrs["WEIGHTS_0"] = self.accessor(np.asarray(weights, np.float32))
            else:
                top = np.iinfo(weights_normalized_dtype).max
                attrs["WEIGHTS_0"] = self.accessor(np.rint(np.asarray(weights, np.float64) * top).astype(weights_normalized_dtype), normalized=True)
        p = {"attributes": attrs, "mode": 4, "material": material}
        if index_dtype is not None:
            p["indices"] = self.accessor(geo.idx.astype(index_dtype))
        return p

    def mesh(self, primitives):
        self.doc["meshes"].append({"primitives": primitives})
        return len(self.doc["meshes"]) - 1

    def node(self, mesh=None, translation=None, rotation=None, scale=None, matrix=None, children=(), skin=None, root=False, name=None):
        n = {}
        if matrix is not None:
            n["matrix"] = [float(x) for x in np.asarray(matrix, np.float64).T.reshape(-1)]
        else:
            if translation is not None:
                n["translation"] = [float(x) for x in translation]
            if rotation is not None:
                n["rotation"] = [float(x) for x in rotation]
            if scale is not None:
                n["scale"] = [float(x) for x in scale]
        if mesh is not None:
            n["mesh"] = mesh
        if skin is not None:
            n["skin"] = skin
        if children:
            n["children"] = list(children)
        if name:
            n["name"] = name
        self.doc["nodes"].append(n)
        i = len(self.doc["nodes"]) - 1
        if root:
            self.doc["scenes"][0]["nodes"].append(i)
        return i

    def skin(self, joints, inverse_bind):
        ibm = np.asarray(inverse_bind, np.float64).transpose(0, 2, 1).reshape(-1, 16).astype(np.float32)
        self.doc["skins"].append({"joints": list(joints), "inverseBindMatrices": self.accessor(ibm)})
        return len(self.doc["skins"]) - 1

    def animation(self, channels):
        """channels: (node, path, times, values, interpolation)."""
        samplers, chans = [], []
        for node, path, times, values, interp in channels:
            samplers.append({"input": self.accessor(np.asarray(times, np.float32), minmax=True), "output": self.accessor(np.asarray(values, np.float32)),
                             "interpolation": interp})
            chans.append({"sampler": len(samplers) - 1, "target": {"node": node, "path": path}})
        self.doc["animations"].append({"samplers": samplers, "channels": chans})

    # -- output ------------------------------------------------------------------------
    def _clean(self):
        doc = {k: v for k, v in self.doc.items() if not (isinstance(v, list) and not v)}
        return doc

    def save_gltf(self, path, bin_as_datauri=False):
        doc = self._clean()
        d = os.path.dirname(path)
        if bin_as_datauri:
            doc["buffers"] = [{"byteLength": len(self.bin), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()}]
        else:
            name = os.path.splitext(os.path.basename(path))[0] + ".bin"
            doc["buffers"] = [{"byteLength": len(self.bin), "uri": name}]
            with open(os.path.join(d, name), "wb") as f:
                f.write(bytes(self.bin))
        for fname, data in self.files.items():
            with open(os.path.join(d, fname), "wb") as f:
                f.write(data)
        with open(path, "w") as f:
            json.dump(doc, f)

    def save_glb(self, path):
        doc = self._clean()
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        blob = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)))
            f.write(struct.pack("<I4s", len(js), b"JSON") + js)
            f.write(struct.pack("<I4s", len(blob), b"BIN\0") + blob)
        for fname, data in self.files.items():
            with open(os.path.join(os.path.dirname(path), fname), "wb") as f:
                f.write(data)
