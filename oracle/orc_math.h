// TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference's host
// math. Not part of the product: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may use anything under oracle/.
//
// glm (vcpkg dependency of the reference, Renderer/vcpkg.json:5, no version
// pin) is absent here; these are the published glm formulas for exactly the
// calls the hot path makes (call sites: Renderer/Camera.cpp:104-137,199-202,237;
// Renderer/Node.cpp:11; Renderer/VulkanRendererShadow.cpp:261-273,418;
// Renderer/VulkanRendererEnvironment.cpp:209-217,224; Renderer/main.cpp:452).
// Matrices are column-major: m.c[col][row]. GLM_FORCE_DEPTH_ZERO_TO_ONE is
// defined by the reference build (Renderer/Renderer.vcxproj:92), right-handed.
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct Quat { float w, x, y, z; };
struct M4 { float c[4][4]; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline float distance(V3 a, V3 b) { return length(b - a); }
// glm::normalize = v * inversesqrt(dot(v,v)); inversesqrt(x) = 1/sqrt(x)
inline V3 normalize(V3 a) { float s = 1.0f / std::sqrt(dot(a, a)); return a * s; }

inline M4 identity() { M4 m; std::memset(&m, 0, sizeof m); m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f; return m; }

// glm mat4*mat4: Result[i] = A0*B[i][0] + A1*B[i][1] + A2*B[i][2] + A3*B[i][3], left to right
inline M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int row = 0; row < 4; row++)
            r.c[i][row] = ((a.c[0][row] * b.c[i][0] + a.c[1][row] * b.c[i][1]) + a.c[2][row] * b.c[i][2]) + a.c[3][row] * b.c[i][3];
    return r;
}
// glm mat4*vec4: (m0*x + m1*y) + (m2*z + m3*w)
inline V4 mul(const M4& m, V4 v) {
    float r[4];
    for (int row = 0; row < 4; row++)
        r[row] = (m.c[0][row] * v.x + m.c[1][row] * v.y) + (m.c[2][row] * v.z + m.c[3][row] * v.w);
    return {r[0], r[1], r[2], r[3]};
}

inline M4 translate(V3 v) { M4 m = identity(); m.c[3][0] = v.x; m.c[3][1] = v.y; m.c[3][2] = v.z; return m; }
inline M4 scale(V3 v) { M4 m = identity(); m.c[0][0] = v.x; m.c[1][1] = v.y; m.c[2][2] = v.z; return m; }

// glm::rotate(angle, axis) (gtx/transform.hpp -> gtc/matrix_transform rotate on identity)
inline M4 rotate(float angle, V3 v) {
    const float c = std::cos(angle), s = std::sin(angle);
    V3 axis = normalize(v);
    V3 temp = axis * (1.0f - c);
    M4 m = identity();
    m.c[0][0] = c + temp.x * axis.x;
    m.c[0][1] = temp.x * axis.y + s * axis.z;
    m.c[0][2] = temp.x * axis.z - s * axis.y;
    m.c[1][0] = temp.y * axis.x - s * axis.z;
    m.c[1][1] = c + temp.y * axis.y;
    m.c[1][2] = temp.y * axis.z + s * axis.x;
    m.c[2][0] = temp.z * axis.x + s * axis.y;
    m.c[2][1] = temp.z * axis.y - s * axis.x;
    m.c[2][2] = c + temp.z * axis.z;
    return m;
}

// glm::perspectiveRH_ZO
inline M4 perspective(float fovy, float aspect, float zNear, float zFar) {
    const float tanHalfFovy = std::tan(fovy / 2.0f);
    M4 m; std::memset(&m, 0, sizeof m);
    m.c[0][0] = 1.0f / (aspect * tanHalfFovy);
    m.c[1][1] = 1.0f / tanHalfFovy;
    m.c[2][2] = zFar / (zNear - zFar);
    m.c[2][3] = -1.0f;
    m.c[3][2] = -(zFar * zNear) / (zFar - zNear);
    return m;
}
// glm::orthoRH_ZO
inline M4 ortho(float l, float r, float b, float t, float zNear, float zFar) {
    M4 m = identity();
    m.c[0][0] = 2.0f / (r - l);
    m.c[1][1] = 2.0f / (t - b);
    m.c[2][2] = -1.0f / (zFar - zNear);
    m.c[3][0] = -(r + l) / (r - l);
    m.c[3][1] = -(t + b) / (t - b);
    m.c[3][2] = -zNear / (zFar - zNear);
    return m;
}

// glm::inverse(mat4) (detail::compute_inverse, cofactor expansion)
inline M4 inverse(const M4& m) {
    const float (*a)[4] = m.c;
    float Coef00 = a[2][2] * a[3][3] - a[3][2] * a[2][3];
    float Coef02 = a[1][2] * a[3][3] - a[3][2] * a[1][3];
    float Coef03 = a[1][2] * a[2][3] - a[2][2] * a[1][3];
    float Coef04 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
    float Coef06 = a[1][1] * a[3][3] - a[3][1] * a[1][3];
    float Coef07 = a[1][1] * a[2][3] - a[2][1] * a[1][3];
    float Coef08 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
    float Coef10 = a[1][1] * a[3][2] - a[3][1] * a[1][2];
    float Coef11 = a[1][1] * a[2][2] - a[2][1] * a[1][2];
    float Coef12 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
    float Coef14 = a[1][0] * a[3][3] - a[3][0] * a[1][3];
    float Coef15 = a[1][0] * a[2][3] - a[2][0] * a[1][3];
    float Coef16 = a[2][0] * a[3][2] - a[3][0] * a[2][2];
    float Coef18 = a[1][0] * a[3][2] - a[3][0] * a[1][2];
    float Coef19 = a[1][0] * a[2][2] - a[2][0] * a[1][2];
    float Coef20 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    float Coef22 = a[1][0] * a[3][1] - a[3][0] * a[1][1];
    float Coef23 = a[1][0] * a[2][1] - a[2][0] * a[1][1];

    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float Vec0[4] = {a[1][0], a[0][0], a[0][0], a[0][0]};
    float Vec1[4] = {a[1][1], a[0][1], a[0][1], a[0][1]};
    float Vec2[4] = {a[1][2], a[0][2], a[0][2], a[0][2]};
    float Vec3[4] = {a[1][3], a[0][3], a[0][3], a[0][3]};
    const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
    M4 inv;
    for (int i = 0; i < 4; i++) {
        float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        inv.c[0][i] = Inv0 * SignA[i];
        inv.c[1][i] = Inv1 * SignB[i];
        inv.c[2][i] = Inv2 * SignA[i];
        inv.c[3][i] = Inv3 * SignB[i];
    }
    // Row0 = first row of Inverse = (inv[0][0], inv[1][0], inv[2][0], inv[3][0])
    float Dot0[4] = {a[0][0] * inv.c[0][0], a[0][1] * inv.c[1][0], a[0][2] * inv.c[2][0], a[0][3] * inv.c[3][0]};
    float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) inv.c[i][j] = inv.c[i][j] * OneOverDeterminant;
    return inv;
}

// glm::eulerAngles(quat) = (pitch, yaw, roll)
inline V3 eulerAngles(Quat q) {
    float y_ = 2.0f * (q.y * q.z + q.w * q.x);
    float x_ = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
    float pitch;
    // glm: if both ~0 avoid atan2(0,0): 2*atan2(q.x, q.w)
    if (std::fabs(x_) <= 1.1920929e-07f && std::fabs(y_) <= 1.1920929e-07f) pitch = 2.0f * std::atan2(q.x, q.w);
    else pitch = std::atan2(y_, x_);
    float sy = -2.0f * (q.x * q.z - q.w * q.y);
    sy = sy < -1.0f ? -1.0f : (sy > 1.0f ? 1.0f : sy);
    float yaw = std::asin(sy);
    float ry = 2.0f * (q.x * q.y + q.w * q.z);
    float rx = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
    float roll;
    if (std::fabs(rx) <= 1.1920929e-07f && std::fabs(ry) <= 1.1920929e-07f) roll = 0.0f;
    else roll = std::atan2(ry, rx);
    return {pitch, yaw, roll};
}

// glm quat * vec3
inline V3 rotate(Quat q, V3 v) {
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    V3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}

}  // namespace orc
