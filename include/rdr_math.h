// Host-side vector/matrix/quaternion math of the library.
//
// The reference builds every matrix with glm (Renderer/Camera.cpp:136-139,
// 196-216; Renderer/VulkanRendererShadow.cpp:260-273, 404-420;
// Renderer/DirectionalLight.cpp:3-5). glm is not a dependency here; these are
// the same closed forms (right-handed, depth 0..1 as the reference's
// GLM_FORCE_DEPTH_ZERO_TO_ONE build, Renderer/Renderer.vcxproj:92), evaluated
// in float in glm's order of operations. Storage is column-major, m[c*4+r].
#pragma once
#include <math.h>
#include <string.h>

namespace rdr {

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct quat { float w, x, y, z; };
struct mat4 {
    float m[16];
    float& at(int c, int r) { return m[c * 4 + r]; }
    float at(int c, int r) const { return m[c * 4 + r]; }
};

static inline vec3 make3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 add(vec3 a, vec3 b) { return make3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 sub(vec3 a, vec3 b) { return make3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 mulf(vec3 a, float s) { return make3(a.x * s, a.y * s, a.z * s); }
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) { return make3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
static inline float distance(vec3 a, vec3 b) { vec3 d = sub(b, a); return sqrtf(dot(d, d)); }
static inline vec3 normalize(vec3 a) { return mulf(a, 1.0f / sqrtf(dot(a, a))); }

static inline mat4 mat_identity() {
    mat4 r;
    memset(r.m, 0, sizeof r.m);
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}

// column i of the product = a.col0*b[i][0] + a.col1*b[i][1] + a.col2*b[i][2] + a.col3*b[i][3], summed left to right
static inline mat4 mat_mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int c = 0; c < 4; c++) {
        const float b0 = b.at(c, 0), b1 = b.at(c, 1), b2 = b.at(c, 2), b3 = b.at(c, 3);
        for (int row = 0; row < 4; row++) {
            float s = a.at(0, row) * b0 + a.at(1, row) * b1;
            s = s + a.at(2, row) * b2;
            s = s + a.at(3, row) * b3;
            r.at(c, row) = s;
        }
    }
    return r;
}
// (col0*x + col1*y) + (col2*z + col3*w)
static inline vec4 mat_mul_vec(const mat4& a, vec4 v) {
    float o[4];
    for (int row = 0; row < 4; row++) {
        const float lo = a.at(0, row) * v.x + a.at(1, row) * v.y;
        const float hi = a.at(2, row) * v.z + a.at(3, row) * v.w;
        o[row] = lo + hi;
    }
    vec4 r = {o[0], o[1], o[2], o[3]};
    return r;
}

static inline mat4 mat_translate(vec3 t) { mat4 r = mat_identity(); r.at(3, 0) = t.x; r.at(3, 1) = t.y; r.at(3, 2) = t.z; return r; }
static inline mat4 mat_scale(vec3 s) { mat4 r = mat_identity(); r.at(0, 0) = s.x; r.at(1, 1) = s.y; r.at(2, 2) = s.z; return r; }

static inline mat4 mat_rotate(float angle, vec3 axis_in) {
    const float c = cosf(angle), s = sinf(angle);
    const vec3 axis = normalize(axis_in);
    const vec3 t = mulf(axis, 1.0f - c);
    mat4 r = mat_identity();
    r.at(0, 0) = c + t.x * axis.x;          r.at(0, 1) = t.x * axis.y + s * axis.z; r.at(0, 2) = t.x * axis.z - s * axis.y;
    r.at(1, 0) = t.y * axis.x - s * axis.z; r.at(1, 1) = c + t.y * axis.y;          r.at(1, 2) = t.y * axis.z + s * axis.x;
    r.at(2, 0) = t.z * axis.x + s * axis.y; r.at(2, 1) = t.z * axis.y - s * axis.x; r.at(2, 2) = c + t.z * axis.z;
    return r;
}

static inline mat4 mat_perspective(float fovy, float aspect, float zn, float zf) {
    const float th = tanf(fovy / 2.0f);
    mat4 r;
    memset(r.m, 0, sizeof r.m);
    r.at(0, 0) = 1.0f / (aspect * th);
    r.at(1, 1) = 1.0f / th;
    r.at(2, 2) = zf / (zn - zf);
    r.at(2, 3) = -1.0f;
    r.at(3, 2) = -(zf * zn) / (zf - zn);
    return r;
}
static inline mat4 mat_ortho(float l, float rgt, float b, float t, float zn, float zf) {
    mat4 r = mat_identity();
    r.at(0, 0) = 2.0f / (rgt - l);
    r.at(1, 1) = 2.0f / (t - b);
    r.at(2, 2) = -1.0f / (zf - zn);
    r.at(3, 0) = -(rgt + l) / (rgt - l);
    r.at(3, 1) = -(t + b) / (t - b);
    r.at(3, 2) = -zn / (zf - zn);
    return r;
}

// cofactor inverse, glm's detail::compute_inverse<4,4>
static inline mat4 mat_inverse(const mat4& a) {
    const float c00 = a.at(2, 2) * a.at(3, 3) - a.at(3, 2) * a.at(2, 3);
    const float c02 = a.at(1, 2) * a.at(3, 3) - a.at(3, 2) * a.at(1, 3);
    const float c03 = a.at(1, 2) * a.at(2, 3) - a.at(2, 2) * a.at(1, 3);
    const float c04 = a.at(2, 1) * a.at(3, 3) - a.at(3, 1) * a.at(2, 3);
    const float c06 = a.at(1, 1) * a.at(3, 3) - a.at(3, 1) * a.at(1, 3);
    const float c07 = a.at(1, 1) * a.at(2, 3) - a.at(2, 1) * a.at(1, 3);
    const float c08 = a.at(2, 1) * a.at(3, 2) - a.at(3, 1) * a.at(2, 2);
    const float c10 = a.at(1, 1) * a.at(3, 2) - a.at(3, 1) * a.at(1, 2);
    const float c11 = a.at(1, 1) * a.at(2, 2) - a.at(2, 1) * a.at(1, 2);
    const float c12 = a.at(2, 0) * a.at(3, 3) - a.at(3, 0) * a.at(2, 3);
    const float c14 = a.at(1, 0) * a.at(3, 3) - a.at(3, 0) * a.at(1, 3);
    const float c15 = a.at(1, 0) * a.at(2, 3) - a.at(2, 0) * a.at(1, 3);
    const float c16 = a.at(2, 0) * a.at(3, 2) - a.at(3, 0) * a.at(2, 2);
    const float c18 = a.at(1, 0) * a.at(3, 2) - a.at(3, 0) * a.at(1, 2);
    const float c19 = a.at(1, 0) * a.at(2, 2) - a.at(2, 0) * a.at(1, 2);
    const float c20 = a.at(2, 0) * a.at(3, 1) - a.at(3, 0) * a.at(2, 1);
    const float c22 = a.at(1, 0) * a.at(3, 1) - a.at(3, 0) * a.at(1, 1);
    const float c23 = a.at(1, 0) * a.at(2, 1) - a.at(2, 0) * a.at(1, 1);
    const vec4 f0 = {c00, c00, c02, c03}, f1 = {c04, c04, c06, c07}, f2 = {c08, c08, c10, c11};
    const vec4 f3 = {c12, c12, c14, c15}, f4 = {c16, c16, c18, c19}, f5 = {c20, c20, c22, c23};
    const vec4 v0 = {a.at(1, 0), a.at(0, 0), a.at(0, 0), a.at(0, 0)}, v1 = {a.at(1, 1), a.at(0, 1), a.at(0, 1), a.at(0, 1)};
    const vec4 v2 = {a.at(1, 2), a.at(0, 2), a.at(0, 2), a.at(0, 2)}, v3 = {a.at(1, 3), a.at(0, 3), a.at(0, 3), a.at(0, 3)};
#define RDR_LANE(V, i) ((i) == 0 ? (V).x : (i) == 1 ? (V).y : (i) == 2 ? (V).z : (V).w)
    mat4 inv;
    for (int i = 0; i < 4; i++) {
        const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
        const float i0 = (RDR_LANE(v1, i) * RDR_LANE(f0, i) - RDR_LANE(v2, i) * RDR_LANE(f1, i)) + RDR_LANE(v3, i) * RDR_LANE(f2, i);
        const float i1 = (RDR_LANE(v0, i) * RDR_LANE(f0, i) - RDR_LANE(v2, i) * RDR_LANE(f3, i)) + RDR_LANE(v3, i) * RDR_LANE(f4, i);
        const float i2 = (RDR_LANE(v0, i) * RDR_LANE(f1, i) - RDR_LANE(v1, i) * RDR_LANE(f3, i)) + RDR_LANE(v3, i) * RDR_LANE(f5, i);
        const float i3 = (RDR_LANE(v0, i) * RDR_LANE(f2, i) - RDR_LANE(v1, i) * RDR_LANE(f4, i)) + RDR_LANE(v2, i) * RDR_LANE(f5, i);
        inv.at(0, i) = i0 * sa; inv.at(1, i) = i1 * sb; inv.at(2, i) = i2 * sa; inv.at(3, i) = i3 * sb;
    }
#undef RDR_LANE
    const float d0 = a.at(0, 0) * inv.at(0, 0), d1 = a.at(0, 1) * inv.at(1, 0), d2 = a.at(0, 2) * inv.at(2, 0), d3 = a.at(0, 3) * inv.at(3, 0);
    const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int k = 0; k < 16; k++) inv.m[k] = inv.m[k] * one_over_det;
    return inv;
}

// glm::eulerAngles: (pitch, yaw, roll)
static inline vec3 quat_euler(quat q) {
    const float eps = 1.1920929e-07f;
    const float py = 2.0f * (q.y * q.z + q.w * q.x), px = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
    const float pitch = (fabsf(px) <= eps && fabsf(py) <= eps) ? 2.0f * atan2f(q.x, q.w) : atan2f(py, px);
    float sy = -2.0f * (q.x * q.z - q.w * q.y);
    if (sy < -1.0f) sy = -1.0f;
    if (sy > 1.0f) sy = 1.0f;
    const float yaw = asinf(sy);
    const float ry = 2.0f * (q.x * q.y + q.w * q.z), rx = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
    const float roll = (fabsf(rx) <= eps && fabsf(ry) <= eps) ? 0.0f : atan2f(ry, rx);
    return make3(pitch, yaw, roll);
}
// q * v
static inline vec3 quat_rotate(quat q, vec3 v) {
    const vec3 u = make3(q.x, q.y, q.z);
    const vec3 uv = cross(u, v), uuv = cross(u, uv);
    return add(v, mulf(add(mulf(uv, q.w), uuv), 2.0f));
}
// glm::quatLookAt (right-handed) followed by quat_cast
static inline quat quat_look_at(vec3 dir, vec3 up) {
    const vec3 c2 = mulf(dir, -1.0f);
    const vec3 right = cross(up, c2);
    const float l2 = dot(right, right);
    const vec3 c0 = mulf(right, 1.0f / sqrtf(l2 > 0.00001f ? l2 : 0.00001f));
    const vec3 c1 = cross(c2, c0);
    const float fx = c0.x - c1.y - c2.z, fy = c1.y - c0.x - c2.z, fz = c2.z - c0.x - c1.y, fw = c0.x + c1.y + c2.z;
    int big = 0;
    float fb = fw;
    if (fx > fb) { fb = fx; big = 1; }
    if (fy > fb) { fb = fy; big = 2; }
    if (fz > fb) { fb = fz; big = 3; }
    const float bv = sqrtf(fb + 1.0f) * 0.5f, mult = 0.25f / bv;
    quat q;
    switch (big) {
        case 0: q.w = bv; q.x = (c1.z - c2.y) * mult; q.y = (c2.x - c0.z) * mult; q.z = (c0.y - c1.x) * mult; break;
        case 1: q.w = (c1.z - c2.y) * mult; q.x = bv; q.y = (c0.y + c1.x) * mult; q.z = (c2.x + c0.z) * mult; break;
        case 2: q.w = (c2.x - c0.z) * mult; q.x = (c0.y + c1.x) * mult; q.y = bv; q.z = (c1.z + c2.y) * mult; break;
        default: q.w = (c0.y - c1.x) * mult; q.x = (c2.x + c0.z) * mult; q.y = (c1.z + c2.y) * mult; q.z = bv; break;
    }
    return q;
}

// glm::mat4_cast
static inline mat4 quat_to_mat4(quat q) {
    const float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    const float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    mat4 r = mat_identity();
    r.at(0, 0) = 1.0f - 2.0f * (qyy + qzz); r.at(0, 1) = 2.0f * (qxy + qwz);        r.at(0, 2) = 2.0f * (qxz - qwy);
    r.at(1, 0) = 2.0f * (qxy - qwz);        r.at(1, 1) = 1.0f - 2.0f * (qxx + qzz); r.at(1, 2) = 2.0f * (qyz + qwx);
    r.at(2, 0) = 2.0f * (qxz + qwy);        r.at(2, 1) = 2.0f * (qyz - qwx);        r.at(2, 2) = 1.0f - 2.0f * (qxx + qyy);
    return r;
}
// glm::quat(eulerAngles): pitch (x), yaw (y), roll (z)
static inline quat quat_from_euler(vec3 e) {
    const float cx = cosf(e.x * 0.5f), cy = cosf(e.y * 0.5f), cz = cosf(e.z * 0.5f);
    const float sx = sinf(e.x * 0.5f), sy = sinf(e.y * 0.5f), sz = sinf(e.z * 0.5f);
    quat q;
    q.w = cx * cy * cz + sx * sy * sz;
    q.x = sx * cy * cz - cx * sy * sz;
    q.y = cx * sy * cz + sx * cy * sz;
    q.z = cx * cy * sz - sx * sy * cz;
    return q;
}
// translate(t) * mat4_cast(r) * scale(s): the local matrix of a node (Renderer/Node.cpp:11)
static inline mat4 mat_trs(vec3 t, quat r, vec3 s) { return mat_mul(mat_mul(mat_translate(t), quat_to_mat4(r)), mat_scale(s)); }

// Camera of the reference (Renderer/Camera.h:61-82): state + crop matrix.
struct Camera {
    int type;  // 0 perspective, 1 orthographic
    vec3 position, euler;
    float near_plane, far_plane, fov_or_height, aspect;
    mat4 crop;
    // Camera::rebuildViewMatrixUnsafe (Camera.cpp:136-139)
    mat4 view() const {
        mat4 r = mat_mul(mat_rotate(-euler.x, make3(1, 0, 0)), mat_rotate(-euler.y, make3(0, 1, 0)));
        r = mat_mul(r, mat_rotate(-euler.z, make3(0, 0, 1)));
        return mat_mul(r, mat_translate(make3(-position.x, -position.y, -position.z)));
    }
    // Camera::rebuildProjMatrixUnsafe (Camera.cpp:196-206)
    mat4 proj() const {
        const mat4 flip = mat_scale(make3(1.0f, -1.0f, 1.0f));
        if (type == 0) return mat_mul(flip, mat_perspective(fov_or_height, aspect, near_plane, far_plane));
        return mat_mul(flip, mat_ortho(-fov_or_height * aspect / 2, fov_or_height * aspect / 2, -fov_or_height, fov_or_height, near_plane, far_plane));
    }
    // Camera::rebuildViewProjMatrixUnsafe (Camera.cpp:208-216)
    mat4 view_proj() const { return mat_mul(mat_mul(crop, proj()), view()); }
};

}  // namespace rdr
