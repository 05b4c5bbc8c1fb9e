"""Host-side scene graph, key-frame animation and skins: what produces the per-frame inputs of
`Renderer.set_mesh_transform` and `Renderer.set_mesh_joints`.

The class and method names follow the reference (Renderer/Animation.h:13-65, Renderer/Node.h:13-83,
Renderer/Node.cpp:9-38); the interpolation itself follows the glTF 2.0 specification instead of
Renderer/Animation.h:86-98, whose LINEAR branch divides by the wrong quantity (`t / (t1 - t0)`) and whose
CUBICSPLINE branch returns the nearest key. Skins do not exist in the reference at all.

Conventions: matrices are numpy math-convention 4x4 float64 (M @ column vector; transpose before handing them to the
C ABI, which is column-major); quaternions are (x, y, z, w) as in glTF."""
import bisect
import math

import numpy as np

LINEAR, STEP, CUBICSPLINE = "LINEAR", "STEP", "CUBICSPLINE"      # AnimationInterpolationCurve (Animation.h:13-17)
CLAMP, REPEAT, MIRROR = "clamp", "repeat", "mirror"              # AnimationRepeatMode (Animation.h:19-23)


def quat_normalize(q):
    q = np.asarray(q, np.float64)
    return q / np.linalg.norm(q)


def quat_slerp(a, b, u):
    """Shortest-arc spherical interpolation; falls back to a normalised lerp when the keys almost coincide."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = float(np.dot(a, b))
    if d < 0.0:
        b, d = -b, -d
    if d > 0.9995:
        return quat_normalize(a + (b - a) * u)
    th = np.arccos(min(d, 1.0))
    return (np.sin((1.0 - u) * th) * a + np.sin(u * th) * b) / np.sin(th)


def quat_to_matrix(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([axis * np.sin(angle / 2), [np.cos(angle / 2)]])


def trs_matrix(translation, rotation, scale):
    """T * R * S as one 4x4: column j of the rotation scaled by scale[j] (the products quat_to_matrix forms, on plain
    floats: a frame of an animated scene builds one of these per node)."""
    x, y, z, w = np.asarray(rotation, np.float64).tolist()
    sx, sy, sz = np.asarray(scale, np.float64).tolist()
    tx, ty, tz = np.asarray(translation, np.float64).tolist()
    return np.array([[(1 - 2 * (y * y + z * z)) * sx, (2 * (x * y - z * w)) * sy, (2 * (x * z + y * w)) * sz, tx],
                     [(2 * (x * y + z * w)) * sx, (1 - 2 * (x * x + z * z)) * sy, (2 * (y * z - x * w)) * sz, ty],
                     [(2 * (x * z - y * w)) * sx, (2 * (y * z + x * w)) * sy, (1 - 2 * (x * x + y * y)) * sz, tz],
                     [0.0, 0.0, 0.0, 1.0]], np.float64)


# The same interpolation on lists of floats, for Animation.value_at (called once per animated channel and frame: on
# arrays of three or four elements numpy's call overhead is ten times the arithmetic).
def _normalize_list(q):
    x, y, z, w = q
    n = math.sqrt(x * x + y * y + z * z + w * w)
    return [x / n, y / n, z / n, w / n]


def _slerp_list(a, b, u):
    d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]
    if d < 0.0:
        b, d = [-v for v in b], -d
    if d > 0.9995:
        return _normalize_list([x + (y - x) * u for x, y in zip(a, b)])
    th = math.acos(min(d, 1.0))
    s0, s1, st = math.sin((1.0 - u) * th), math.sin(u * th), math.sin(th)
    return [(s0 * x + s1 * y) / st for x, y in zip(a, b)]


class Animation:
    """One key-frame track (Animation<T>, Animation.h:25-65). `values` has one row per key, or three per key
    (in-tangent, value, out-tangent) for CUBICSPLINE, as glTF stores them. `rotation=True` interpolates unit
    quaternions (slerp / normalised spline)."""

    def __init__(self, keyframe_times, keyframe_values, interpolation=LINEAR, repeat=CLAMP, rotation=False):
        self.times = [float(t) for t in keyframe_times]
        self.values = np.asarray(keyframe_values, np.float64)
        self.interpolation, self.repeat, self.rotation = interpolation, repeat, rotation
        n = len(self.times)
        if n == 0 or any(b <= a for a, b in zip(self.times, self.times[1:])):
            raise ValueError("key-frame times must be non-empty and strictly increasing")
        want = 3 * n if interpolation == CUBICSPLINE else n
        if self.values.shape[0] != want:
            raise ValueError(f"{interpolation}: expected {want} value rows for {n} keys, got {self.values.shape[0]}")
        self._rows = self.values.reshape(want, -1).tolist()   # the keys as lists of floats, for value_at

    def _key(self, i):
        return self._rows[3 * i + 1] if self.interpolation == CUBICSPLINE else self._rows[i]

    def _wrap(self, t):
        t0, t1 = self.times[0], self.times[-1]
        span = t1 - t0
        if self.repeat == CLAMP or span <= 0.0:
            return min(max(t, t0), t1)
        u = (t - t0) / span
        loops = math.floor(u)
        frac = u - loops
        if self.repeat == MIRROR and loops % 2:
            frac = 1.0 - frac
        return t0 + frac * span

    def value_at(self, t):
        return np.array(self._value_at(t), np.float64)

    def _value_at(self, t):
        """value_at as a list of floats."""
        if len(self.times) == 1:
            return list(self._key(0))
        t = self._wrap(float(t))
        i1 = bisect.bisect_right(self.times, t)
        if i1 >= len(self.times):
            out = list(self._key(len(self.times) - 1))
            return _normalize_list(out) if self.rotation else out
        i0 = max(i1 - 1, 0)
        if i1 == 0:
            i1 = 1
        t0, t1 = self.times[i0], self.times[i1]
        dt = t1 - t0
        u = min(max((t - t0) / dt, 0.0), 1.0)
        if self.interpolation == STEP:
            out = list(self._key(i0))
        elif self.interpolation == LINEAR:
            k0, k1 = self._key(i0), self._key(i1)
            out = _slerp_list(k0, k1, u) if self.rotation else [a + (b - a) * u for a, b in zip(k0, k1)]
        else:  # cubic Hermite, tangents scaled by the key spacing
            v0, b0 = self._rows[3 * i0 + 1], self._rows[3 * i0 + 2]
            a1, v1 = self._rows[3 * i1], self._rows[3 * i1 + 1]
            u2, u3 = u * u, u * u * u
            c0, c1, c2, c3 = 2 * u3 - 3 * u2 + 1, dt * (u3 - 2 * u2 + u), -2 * u3 + 3 * u2, dt * (u3 - u2)
            out = [c0 * p + c1 * q + c2 * r + c3 * s for p, q, r, s in zip(v0, b0, v1, a1)]
        return _normalize_list(out) if self.rotation else out


class Node:
    """Scene-graph node (Node.h:13-83): TRS or matrix, children, optional per-channel animations."""

    def __init__(self, translation=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1), children=(), matrix=None, name=""):
        self.translation = np.asarray(translation, np.float64)
        self.rotation = np.asarray(rotation, np.float64)
        self.scale = np.asarray(scale, np.float64)
        self.matrix = None if matrix is None else np.asarray(matrix, np.float64)
        self.children = list(children)
        self.name = name
        self.translation_animation = self.rotation_animation = self.scale_animation = None
        self.model = np.eye(4)

    def set_transform(self, translation, rotation, scale):  # Node.h:44-49
        self.translation, self.rotation, self.scale = (np.asarray(v, np.float64) for v in (translation, rotation, scale))
        self.matrix = None

    def is_static(self):  # Node.cpp:65-75: a node with any animation is dynamic
        return self.translation_animation is None and self.rotation_animation is None and self.scale_animation is None

    def local_matrix(self):
        return self.matrix if self.matrix is not None else trs_matrix(self.translation, self.rotation, self.scale)

    def set_animation_time(self, t):  # Node.cpp:21-38
        if self.translation_animation is not None:
            self.translation, self.matrix = self.translation_animation.value_at(t), None
        if self.rotation_animation is not None:
            self.rotation, self.matrix = self.rotation_animation.value_at(t), None
        if self.scale_animation is not None:
            self.scale, self.matrix = self.scale_animation.value_at(t), None
        for c in self.children:
            c.set_animation_time(t)

    def recalculate_model(self, parent_model=None):  # Node.cpp:9-19
        local = self.local_matrix()
        self.model = local.copy() if parent_model is None else parent_model @ local
        for c in self.children:
            c.recalculate_model(self.model)

    def model_matrix(self):
        return self.model


class Skin:
    """glTF skin: joint nodes + inverse bind matrices. `joint_matrices(mesh_model)` is the palette that
    rdr_set_mesh_joints expects: inverse(mesh global) @ joint global @ inverse bind, per joint."""

    def __init__(self, joints, inverse_bind_matrices=None):
        self.joints = list(joints)
        n = len(self.joints)
        self.inverse_bind = (np.tile(np.eye(4), (n, 1, 1)) if inverse_bind_matrices is None
                             else np.asarray(inverse_bind_matrices, np.float64).reshape(n, 4, 4))

    def joint_matrices(self, mesh_model=None):
        if mesh_model is None:
            inv = np.eye(4)
        else:
            m = np.asarray(mesh_model, np.float64)
            key = m.tobytes()
            if getattr(self, "_inv_key", None) != key:   # the same mesh matrix frame after frame: inverted once
                self._inv_key, self._inv = key, np.linalg.inv(m)
            inv = self._inv
        return np.stack([inv @ j.model @ ib for j, ib in zip(self.joints, self.inverse_bind)])

    def palette_for_abi(self, mesh_model=None):
        """float32 [n][4][4] laid out as the C ABI wants it (each matrix column-major when flattened)."""
        return np.ascontiguousarray(self.joint_matrices(mesh_model).transpose(0, 2, 1), np.float32)
