// C++ façade over the C ABI of rdr.h: the reference's own scene API
// (Camera, Node, Mesh/MeshPrimitive, PointLight, DirectionalLight, Object,
// handles, RendererSettings) and the public surface of its renderer class
// (Renderer/VulkanRenderer.h:94-110), header-only, C++17, no dependencies
// beyond rdr.h and rdr_math.h.
//
// A host program written against the reference keeps its scene code; the
// differences are the ones a head-less compute node forces:
//   - the window argument of the constructor becomes (width, height, device);
//   - there is no swap chain: renderFrame() returns the finished RGBA8 image
//     (start() still spawns a free-running render thread as in the reference,
//     VulkanRenderer.cpp:495-498, and keeps the latest frame);
//   - glm types become the small rdr:: structs of rdr_math.h (same layout:
//     column-major mat4, (w,x,y,z) quaternion);
//   - the material of a primitive, which the reference's half-refactored
//     Mesh.h no longer carries (PBRInfo/AlphaInfo are used at main.cpp:162-165
//     but defined nowhere), is a member of MeshPrimitive.
//
// Every entry point is reached through RDR_ABI(name), which pastes
// RDR_ABI_PREFIX (default rdr_) in front of the name, so the same façade can
// be compiled against any library exporting the ABI under another prefix.
#pragma once

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "rdr.h"
#include "rdr_math.h"

#ifndef RDR_ABI_PREFIX
#define RDR_ABI_PREFIX rdr_
#endif
#define RDR_ABI_CAT2(a, b) a##b
#define RDR_ABI_CAT(a, b) RDR_ABI_CAT2(a, b)
#define RDR_ABI(name) RDR_ABI_CAT(RDR_ABI_PREFIX, name)

extern "C" {
// the subset of rdr.h the façade calls, declared under the chosen prefix
int RDR_ABI(create)(const rdr_settings*, uint32_t, uint32_t, int, rdr_ctx**);
void RDR_ABI(destroy)(rdr_ctx*);
const char* RDR_ABI(last_error)(const rdr_ctx*);
int RDR_ABI(load_buffer)(rdr_ctx*, const void*, size_t, uint32_t*);
int RDR_ABI(load_image)(rdr_ctx*, const void*, size_t, uint32_t, uint32_t, int, uint32_t, uint32_t*);
int RDR_ABI(make_texture)(rdr_ctx*, uint32_t, const rdr_sampler*, uint32_t*);
int RDR_ABI(add_mesh)(rdr_ctx*, const rdr_primitive_desc*, uint32_t, const float*, int, uint32_t*);
int RDR_ABI(set_mesh_transform)(rdr_ctx*, uint32_t, const float*);
int RDR_ABI(set_mesh_joints)(rdr_ctx*, uint32_t, const float*, uint32_t);
int RDR_ABI(set_lights)(rdr_ctx*, const rdr_point_light*, uint32_t, const rdr_directional_light*);
int RDR_ABI(set_environment_map)(rdr_ctx*, const float* const*, uint32_t, uint32_t);
int RDR_ABI(set_brdf_lut)(rdr_ctx*, const uint8_t*, uint32_t, uint32_t, uint32_t);
int RDR_ABI(set_environment_equirect)(rdr_ctx*, const float*, uint32_t, uint32_t, uint32_t);
int RDR_ABI(generate_brdf_lut)(rdr_ctx*, uint32_t, uint32_t);
int RDR_ABI(set_camera)(rdr_ctx*, const rdr_camera*);
int RDR_ABI(set_exposure_bias)(rdr_ctx*, float, float);
int RDR_ABI(render_frame)(rdr_ctx*, const rdr_frame_params*, uint8_t*, int, float*);
int RDR_ABI(get_frame_stats)(rdr_ctx*, rdr_frame_stats*);
int RDR_ABI(wait_frame)(rdr_ctx*, uint64_t);
int RDR_ABI(synchronize)(rdr_ctx*);
}

namespace hiprenderer {

using rdr::mat4;
using rdr::quat;
using rdr::vec3;
using rdr::vec4;
struct vec2 { float x, y; };

// ---- Handle.h / Buffer.h / Image.h / Texture.h ------------------------------
template <class keyT, int uniqueID> class Handle {
    keyT m_key = static_cast<keyT>(~keyT());
public:
    Handle() = default;
    explicit Handle(keyT key) : m_key(key) {}
    using key_type = keyT;
    explicit operator keyT() const { return m_key; }
    bool operator==(const Handle& o) const { return m_key == o.m_key; }
    bool operator!=(const Handle& o) const { return m_key != o.m_key; }
    bool operator<(const Handle& o) const { return m_key < o.m_key; }
    bool operator>(const Handle& o) const { return m_key > o.m_key; }
    bool isValid() const { return m_key != static_cast<keyT>(~keyT()); }
};
typedef Handle<uint32_t, 0> Buffer;
typedef Handle<uint32_t, 1> Image;
typedef Handle<uint32_t, 2> Texture;

enum class ImageFormat {  // Image.h:5-14, same values as rdr_image_format
    eR8G8B8Unorm, eR8G8B8Srgb, eR8G8B8A8Unorm, eR8G8B8A8Srgb, eR16G16B16Sfloat, eR16G16B16A16Sfloat, eR32G32B32Sfloat, eR32G32B32A32Sfloat,
};

// Sampler.h is missing from the reference tree; these are the fields VulkanRenderer::makeTexture reads
// (VulkanRenderer.cpp:363-385).
enum class SamplerFilter { eNearest, eLinear };
enum class SamplerWrap { eRepeat, eMirroredRepeat, eClampToEdge };
struct Sampler {
    SamplerFilter magFilter = SamplerFilter::eLinear, minFilter = SamplerFilter::eLinear, mipmapFilter = SamplerFilter::eLinear;
    SamplerWrap wrapU = SamplerWrap::eRepeat, wrapV = SamplerWrap::eRepeat;
};

// ---- Camera.h -----------------------------------------------------------------
class Camera {
public:
    enum class CameraType { ePerspective, eOrthographic };
    Camera() = default;
    // the reference's copy drops the crop matrix (Camera.cpp:3-41); kept
    Camera(const Camera& o) { copyFrom(o); }
    Camera& operator=(const Camera& o) { if (this != &o) copyFrom(o); return *this; }

    static Camera Perspective(vec3 position, vec3 eulerRotation, float near_, float far_, float vfov, float aspectRatio) {
        Camera c; c.cameraType = CameraType::ePerspective; c._position = position; c._eulerRotation = eulerRotation;
        c._near = near_; c._far = far_; c._fovOrHeight = vfov; c._aspectRatio = aspectRatio; return c;
    }
    static Camera Ortographic(vec3 position, vec3 eulerRotation, float near_, float far_, float height, float aspectRatio) {
        Camera c; c.cameraType = CameraType::eOrthographic; c._position = position; c._eulerRotation = eulerRotation;
        c._near = near_; c._far = far_; c._fovOrHeight = height; c._aspectRatio = aspectRatio; return c;
    }

    vec3 position() const { std::lock_guard<std::mutex> g(mutex); return _position; }
    vec3 eulerRotation() const { std::lock_guard<std::mutex> g(mutex); return _eulerRotation; }
    mat4 viewMatrix() const { std::lock_guard<std::mutex> g(mutex); return state().view(); }
    mat4 viewProjMatrix() const { std::lock_guard<std::mutex> g(mutex); return state().view_proj(); }
    std::tuple<vec3, mat4> positionAndMatrix() const { std::lock_guard<std::mutex> g(mutex); return {_position, state().view_proj()}; }

    void setPosition(vec3 p) { std::lock_guard<std::mutex> g(mutex); _position = p; }
    // translation is given in camera space (Camera.cpp:104-109)
    void move(vec3 t) {
        std::lock_guard<std::mutex> g(mutex);
        const mat4 inv = rdr::mat_inverse(state().view());
        const vec3 w = rdr::make3(inv.at(0, 0) * t.x + inv.at(1, 0) * t.y + inv.at(2, 0) * t.z,
                                  inv.at(0, 1) * t.x + inv.at(1, 1) * t.y + inv.at(2, 1) * t.z,
                                  inv.at(0, 2) * t.x + inv.at(1, 2) * t.y + inv.at(2, 2) * t.z);
        _position = rdr::add(_position, w);
    }
    void pan(vec2 direction) { move(rdr::make3(direction.x, direction.y, 0.0f)); }
    void dolly(float distance) { move(rdr::make3(0.0f, 0.0f, -distance)); }
    void tilt(vec2 yawPitch) { std::lock_guard<std::mutex> g(mutex); _eulerRotation = rdr::add(_eulerRotation, rdr::make3(yawPitch.y, yawPitch.x, 0.0f)); }
    void setEulerRotation(vec3 e) { std::lock_guard<std::mutex> g(mutex); _eulerRotation = e; }

    float near() const { return _near; }
    float far() const { return _far; }
    float vfov() const { return _fovOrHeight; }
    float hfov() const { return _aspectRatio * _fovOrHeight; }  // sic, Camera.cpp:179-181
    float aspectRatio() const { return _aspectRatio; }
    CameraType type() const { return cameraType; }
    void setNear(float n) { std::lock_guard<std::mutex> g(mutex); _near = n; }
    void setFar(float f) { std::lock_guard<std::mutex> g(mutex); _far = f; }
    void setVFov(float fov) { std::lock_guard<std::mutex> g(mutex); _fovOrHeight = fov; }
    void setAspectRatio(float ar) { std::lock_guard<std::mutex> g(mutex); _aspectRatio = ar; }
    void setCropMatrix(mat4 m) { std::lock_guard<std::mutex> g(mutex); _cropMatrix = m; }
    void clearCropMatrix() { std::lock_guard<std::mutex> g(mutex); _cropMatrix = rdr::mat_identity(); }

    // Camera::getFrustumVertices (Camera.cpp:218-242): the 8 NDC corners through inverse(viewProj)
    std::array<vec3, 8> getFrustumVertices() const {
        static const float ndc[8][3] = {{-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}};
        const mat4 inv = rdr::mat_inverse(viewProjMatrix());
        std::array<vec3, 8> out;
        for (int k = 0; k < 8; k++) {
            const vec4 h = rdr::mat_mul_vec(inv, vec4{ndc[k][0], ndc[k][1], ndc[k][2], 1.0f});
            out[k] = rdr::make3(h.x / h.w, h.y / h.w, h.z / h.w);
        }
        return out;
    }

    rdr_camera toAbi() const {
        std::lock_guard<std::mutex> g(mutex);
        rdr_camera c;
        c.type = cameraType == CameraType::ePerspective ? RDR_CAMERA_PERSPECTIVE : RDR_CAMERA_ORTHOGRAPHIC;
        c.position[0] = _position.x; c.position[1] = _position.y; c.position[2] = _position.z;
        c.euler[0] = _eulerRotation.x; c.euler[1] = _eulerRotation.y; c.euler[2] = _eulerRotation.z;
        c.near_plane = _near; c.far_plane = _far; c.fov_or_height = _fovOrHeight; c.aspect_ratio = _aspectRatio;
        return c;
    }

private:
    void copyFrom(const Camera& o) {
        std::lock_guard<std::mutex> g(o.mutex);
        _position = o._position; _eulerRotation = o._eulerRotation; cameraType = o.cameraType;
        _near = o._near; _far = o._far; _fovOrHeight = o._fovOrHeight; _aspectRatio = o._aspectRatio;
        _cropMatrix = rdr::mat_identity();
    }
    rdr::Camera state() const {
        rdr::Camera s;
        s.type = cameraType == CameraType::ePerspective ? 0 : 1;
        s.position = _position; s.euler = _eulerRotation;
        s.near_plane = _near; s.far_plane = _far; s.fov_or_height = _fovOrHeight; s.aspect = _aspectRatio;
        s.crop = _cropMatrix;
        return s;
    }
    mutable std::mutex mutex;
    vec3 _position{0, 0, 0}, _eulerRotation{0, 0, 0};
    mat4 _cropMatrix = rdr::mat_identity();
    CameraType cameraType = CameraType::ePerspective;
    float _near = 0.1f, _far = 100.0f, _fovOrHeight = 35.0f, _aspectRatio = 16.0f / 9.0f;
};

// ---- Animation.h ------------------------------------------------------------------
// Same class, constructor and valueAt as the reference; the interpolation follows the glTF 2.0 specification, which
// the reference does not (Animation.h:96 divides by t1 - t0 without subtracting t0; :91-94 returns the nearest key for
// CUBICSPLINE). CUBICSPLINE tracks hold three values per key: in-tangent, value, out-tangent.
enum class AnimationInterpolationCurve { eLinear, eStep, eCubicSpline };
enum class AnimationRepeatMode { eClamp, eRepeat, eMirror };

namespace anim_detail {
inline vec3 scaled(vec3 a, float s) { return rdr::mulf(a, s); }
inline vec3 sum(vec3 a, vec3 b) { return rdr::add(a, b); }
inline quat scaled(quat a, float s) { return quat{a.w * s, a.x * s, a.y * s, a.z * s}; }
inline quat sum(quat a, quat b) { return quat{a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 finish(vec3 v) { return v; }
inline quat finish(quat q) { const float n = 1.0f / std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z); return scaled(q, n); }
inline vec3 mix(vec3 a, vec3 b, float u) { return sum(a, scaled(rdr::sub(b, a), u)); }
inline quat mix(quat a, quat b, float u) {  // shortest-arc slerp
    float d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
    if (d < 0.0f) { b = scaled(b, -1.0f); d = -d; }
    if (d > 0.9995f) return finish(sum(a, scaled(sum(b, scaled(a, -1.0f)), u)));
    const float th = std::acos(std::min(d, 1.0f));
    return scaled(sum(scaled(a, std::sin((1.0f - u) * th)), scaled(b, std::sin(u * th))), 1.0f / std::sin(th));
}
}  // namespace anim_detail

template <typename T> class Animation {
public:
    Animation(std::vector<float> keyframeTimes, std::vector<T> keyframeValues,
              AnimationInterpolationCurve interpolation = AnimationInterpolationCurve::eLinear,
              AnimationRepeatMode repeat = AnimationRepeatMode::eClamp)
        : interpolationCurve(interpolation), repeatMode(repeat), keyframeTimes(std::move(keyframeTimes)), keyframes(std::move(keyframeValues)) {
        const size_t want = this->keyframeTimes.size() * (interpolation == AnimationInterpolationCurve::eCubicSpline ? 3 : 1);
        if (this->keyframeTimes.empty() || keyframes.size() != want) throw std::invalid_argument("Animation: key times and values do not match");
    }
    T valueAt(float t) const {
        using namespace anim_detail;
        const size_t n = keyframeTimes.size();
        if (n == 1) return key(0);
        const float first = keyframeTimes.front(), last = keyframeTimes.back(), span = last - first;
        if (repeatMode == AnimationRepeatMode::eClamp || span <= 0.0f) t = std::min(std::max(t, first), last);
        else {
            const float u = (t - first) / span, loops = std::floor(u);
            float frac = u - loops;
            if (repeatMode == AnimationRepeatMode::eMirror && (static_cast<long long>(loops) % 2)) frac = 1.0f - frac;
            t = first + frac * span;
        }
        size_t i1 = std::upper_bound(keyframeTimes.begin(), keyframeTimes.end(), t) - keyframeTimes.begin();
        if (i1 >= n) return finish(key(n - 1));
        if (i1 == 0) i1 = 1;
        const size_t i0 = i1 - 1;
        const float dt = keyframeTimes[i1] - keyframeTimes[i0];
        const float u = std::min(std::max((t - keyframeTimes[i0]) / dt, 0.0f), 1.0f);
        switch (interpolationCurve) {
            case AnimationInterpolationCurve::eStep: return key(i0);
            case AnimationInterpolationCurve::eLinear: return mix(key(i0), key(i1), u);
            default: {
                const float u2 = u * u, u3 = u2 * u;
                const T v = sum(sum(scaled(keyframes[3 * i0 + 1], 2 * u3 - 3 * u2 + 1), scaled(keyframes[3 * i0 + 2], dt * (u3 - 2 * u2 + u))),
                                sum(scaled(keyframes[3 * i1 + 1], -2 * u3 + 3 * u2), scaled(keyframes[3 * i1], dt * (u3 - u2))));
                return finish(v);
            }
        }
    }
private:
    T key(size_t i) const { return interpolationCurve == AnimationInterpolationCurve::eCubicSpline ? keyframes[3 * i + 1] : keyframes[i]; }
    AnimationInterpolationCurve interpolationCurve;
    AnimationRepeatMode repeatMode;
    std::vector<float> keyframeTimes;
    std::vector<T> keyframes;
};

// ---- Node.h / Node.cpp ----------------------------------------------------------
class Node {
public:
    Node(const vec3& translation, const quat& rotation, const vec3& scale, const std::vector<std::shared_ptr<Node>>& children)
        : children(children), _translation(translation), _rotation(rotation), _scale(scale) { recalculateModel(rdr::mat_identity()); }
    // the reference decomposes the matrix into TRS (Node.h:25-36); the matrix itself stays the local transform here
    Node(const mat4& matrix, const std::vector<std::shared_ptr<Node>>& children) : children(children), _modelLocal(matrix), _model(matrix) {
        modelDirty = false; hasMatrix = true;
        for (auto& c : this->children) c->recalculateModel(matrix);
    }
    vec3 translation() const { return _translation; }
    quat rotation() const { return _rotation; }
    vec3 scale() const { return _scale; }
    void setTranslation(vec3 t) { _translation = t; modelDirty = true; hasMatrix = false; }
    void setRotation(quat r) { _rotation = r; modelDirty = true; hasMatrix = false; }
    void setScale(vec3 s) { _scale = s; modelDirty = true; hasMatrix = false; }
    void setTransform(vec3 t, quat r, vec3 s) { _translation = t; _rotation = r; _scale = s; modelDirty = true; hasMatrix = false; }
    mat4 modelMatrix() const { return _model; }
    // Node.cpp:9-19
    void recalculateModel(const mat4& parentModel) {
        if (modelDirty && !hasMatrix) _modelLocal = rdr::mat_trs(_translation, _rotation, _scale);
        modelDirty = false;
        _model = rdr::mat_mul(parentModel, _modelLocal);
        for (auto& c : children) c->recalculateModel(_model);
    }
    // Node.cpp:40-63
    void setTranslationAnimation(Animation<vec3> a) { translationAnimation = std::make_unique<Animation<vec3>>(std::move(a)); setStatic(false); }
    void setRotationAnimation(Animation<quat> a) { rotationAnimation = std::make_unique<Animation<quat>>(std::move(a)); setStatic(false); }
    void setScaleAnimation(Animation<vec3> a) { scaleAnimation = std::make_unique<Animation<vec3>>(std::move(a)); setStatic(false); }
    // Node.cpp:21-38
    void setAnimationTime(float t) {
        if (translationAnimation) setTranslation(translationAnimation->valueAt(t));
        if (rotationAnimation) setRotation(rotationAnimation->valueAt(t));
        if (scaleAnimation) setScale(scaleAnimation->valueAt(t));
        for (auto& c : children) c->setAnimationTime(t);
    }
    bool isStatic() const { return _isStatic; }
    // Node.cpp:65-75
    void setStatic(bool s) {
        if (s) { if (!translationAnimation && !rotationAnimation && !scaleAnimation) _isStatic = true; }
        else _isStatic = false;
        for (auto& c : children) c->setStatic(_isStatic);
    }
    const std::vector<std::shared_ptr<Node>>& childNodes() const { return children; }
    void addChild(std::shared_ptr<Node> c) { children.push_back(std::move(c)); }

private:
    std::vector<std::shared_ptr<Node>> children;
    vec3 _translation{0, 0, 0};
    quat _rotation{1, 0, 0, 0};
    vec3 _scale{1, 1, 1};
    bool _isStatic = true, modelDirty = true, hasMatrix = false;
    mat4 _modelLocal = rdr::mat_identity(), _model = rdr::mat_identity();
    std::unique_ptr<Animation<vec3>> translationAnimation, scaleAnimation;
    std::unique_ptr<Animation<quat>> rotationAnimation;
};

// A glTF skin; the reference has none. jointMatrices(meshModel) is the palette rdr_set_mesh_joints takes:
// inverse(mesh global) * joint global * inverse bind, per joint.
class Skin {
public:
    std::vector<std::shared_ptr<Node>> joints;
    std::vector<mat4> inverseBindMatrices;
    Skin(std::vector<std::shared_ptr<Node>> j, std::vector<mat4> ibm) : joints(std::move(j)), inverseBindMatrices(std::move(ibm)) {
        if (inverseBindMatrices.empty()) inverseBindMatrices.assign(joints.size(), rdr::mat_identity());
        if (inverseBindMatrices.size() != joints.size()) throw std::invalid_argument("Skin: one inverse bind matrix per joint");
    }
    std::vector<mat4> jointMatrices(const mat4& meshModel) const {
        const mat4 inv = rdr::mat_inverse(meshModel);
        std::vector<mat4> out(joints.size());
        for (size_t i = 0; i < joints.size(); i++) out[i] = rdr::mat_mul(rdr::mat_mul(inv, joints[i]->modelMatrix()), inverseBindMatrices[i]);
        return out;
    }
};

// ---- Mesh.h ---------------------------------------------------------------------
enum class AttributeValueType { eInt8, eInt16, eInt32, eInt64, eUint8, eUint16, eUint32, eUint64, eHalf, eFloat, eDouble };
enum class AttributeContainerType { eScalar, eVec2, eVec3, eVec4 };
enum class MeshPrimitiveMode { ePoints, eLines, eLineLoop, eLineStrip, eTriangles, eTriangleStrip, eTriangleFan };

struct VertexAttributeDescription {
    std::string attributeName;  // "POSITION", "NORMAL", "TANGENT", "TEXCOORD_0", "JOINTS_0", "WEIGHTS_0" (glTF semantics, main.cpp:367-400)
    Buffer buffer;
    size_t offset = 0;
    int32_t stride = 0;
    size_t count = 0;
    AttributeContainerType containerType = AttributeContainerType::eVec3;
    AttributeValueType valueType = AttributeValueType::eFloat;
};
struct IndexBufferDescription {
    Buffer buffer;
    size_t offset = 0;
    int32_t stride = 0;
    size_t count = 0;
    AttributeValueType indexType = AttributeValueType::eUint32;
};

enum class AlphaMode { eOpaque, eMask, eBlend };
struct PBRInfo {  // main.cpp:162, pbr.frag:21-30
    float baseColorFactor[4] = {1, 1, 1, 1};
    float emissiveFactor[4] = {0, 0, 0, 0};
    float normalScale = 1.0f, metallicFactor = 1.0f, roughnessFactor = 1.0f, aoFactor = 1.0f;
};
struct AlphaInfo { AlphaMode alphaMode = AlphaMode::eOpaque; float alphaCutoff = 0.5f; };
struct Material {
    PBRInfo materialInfo;
    AlphaInfo alphaInfo;
    Texture albedoTexture, normalTexture, metalRoughnessTexture, aoTexture, emissiveTexture;  // invalid handle = the reference's defaults
};

class MeshPrimitive {
    std::vector<VertexAttributeDescription> m_vertexBufferDescription;
    bool m_isIndexed = false;
    IndexBufferDescription m_indexBufferDescription;
    vec3 m_bbMin{0, 0, 0}, m_bbMax{0, 0, 0};
    MeshPrimitiveMode m_mode = MeshPrimitiveMode::eTriangles;
public:
    Material material;
    MeshPrimitive() = default;
    MeshPrimitive(std::vector<VertexAttributeDescription> v, vec3 bbMin, vec3 bbMax, MeshPrimitiveMode mode)
        : m_vertexBufferDescription(std::move(v)), m_bbMin(bbMin), m_bbMax(bbMax), m_mode(mode) {}
    MeshPrimitive(std::vector<VertexAttributeDescription> v, IndexBufferDescription i, vec3 bbMin, vec3 bbMax, MeshPrimitiveMode mode)
        : m_vertexBufferDescription(std::move(v)), m_isIndexed(true), m_indexBufferDescription(i), m_bbMin(bbMin), m_bbMax(bbMax), m_mode(mode) {}
    const std::vector<VertexAttributeDescription>& vertexBufferDescription() const { return m_vertexBufferDescription; }
    const IndexBufferDescription& indexBufferDescription() const { return m_indexBufferDescription; }
    bool isIndexed() const { return m_isIndexed; }
    vec3 bbMin() const { return m_bbMin; }
    vec3 bbMax() const { return m_bbMax; }
    MeshPrimitiveMode mode() const { return m_mode; }
};
class Mesh {
public:
    std::vector<MeshPrimitive> primitives;
    std::shared_ptr<Node> node;
    std::shared_ptr<Skin> skin;  // set for meshes whose primitives carry JOINTS_0 / WEIGHTS_0
    Mesh(std::vector<MeshPrimitive> p, std::shared_ptr<Node> n, std::shared_ptr<Skin> s = nullptr)
        : primitives(std::move(p)), node(std::move(n)), skin(std::move(s)) {}
};

// ---- PointLight.h / DirectionalLight.h -------------------------------------------
enum class PointLightFlagBits : uint8_t { eNone = 0, eCastShadows = 1, eHasShadowMap = 2, eStaticShadowMapRendered = 4 };
class PointLight {
public:
    vec3 point, intensity;
    uint8_t flags;
    uint16_t shadowMapIndex;
    PointLight(vec3 point, vec3 intensity, bool castShadows)
        : point(point), intensity(intensity), flags(castShadows ? (uint8_t)PointLightFlagBits::eCastShadows : 0), shadowMapIndex(0xFFFF) {}
};
class DirectionalLight {
public:
    vec3 position{0, 100, 0}, intensity{0, 0, 0};
    quat orientation{1, 0, 0, 0};
    bool castShadows = true;
    bool staticShadowMapRendered = false;
    // DirectionalLight.cpp:3-5
    Camera lightViewCamera() const { return Camera::Ortographic(position, rdr::quat_euler(orientation), 0.1f, 200.0f, 10.0f, 1.0f); }
};

// ---- Object.h (unused by the reference's renderer; kept for source compatibility) ---
class Object {
public:
    void setTranslation(vec3 t) { translation = t; transformDirty = true; }
    void setRotation(vec3 euler) { rotation = rdr::quat_from_euler(euler); transformDirty = true; }
    void setRotation(quat q) { rotation = q; transformDirty = true; }
    void setScale(vec3 s) { scale = s; transformDirty = true; }
    mat4 getTransformMatrix() {
        if (transformDirty) transform = rdr::mat_trs(translation, rotation, scale);
        transformDirty = false;
        return transform;
    }
private:
    vec3 translation{0, 0, 0};
    quat rotation{1, 0, 0, 0};
    vec3 scale{1, 1, 1};
    bool transformDirty = true;
    mat4 transform = rdr::mat_identity();
};

// ---- VulkanRenderer.h:28-59 -------------------------------------------------------
struct TextureInfo {  // one cube face: RGBA32F bytes (main.cpp:47-59)
    std::vector<uint8_t> data;
    uint32_t width = 1, height = 1;
};
enum class SampleCount : uint32_t { e1 = 1, e4 = 4 };
struct RendererSettings {
    bool hdrOutputEnabled = false;
    bool dynamicShadowsEnabled = true;
    uint32_t pointShadowMapResolution = 512;
    uint16_t pointShadowMapCount = 8;
    uint32_t directionalShadowMapResolution = 2048;
    uint8_t directionalShadowCascadeLevels = 4;
    bool vsync = false;
    float exposureBias = 0.0f;
    float gamma = 2.2f;
    SampleCount msaa = SampleCount::e4;
};

// How exposure is chosen for a frame (see rdr_frame_params in rdr.h)
struct FrameParams {
    int exposureMode = 0;  // 0 reference recurrence with deltaTime, 1 given adapted luminance, 2 given exposure
    float deltaTime = 1.0f / 60.0f, temporalLuminance = 0.25f, exposure = 0.0f;
};

// ---- the renderer (public surface of VulkanRenderer, VulkanRenderer.h:94-110) -----------
class Renderer {
public:
    Renderer(uint32_t width, uint32_t height, const RendererSettings& s, int device = 0)
        : _settings(s), _width(width), _height(height), hdrOutput(s.hdrOutputEnabled) {
        rdr_settings a;
        a.hdr_output_enabled = s.hdrOutputEnabled; a.dynamic_shadows_enabled = s.dynamicShadowsEnabled;
        a.point_shadow_map_resolution = s.pointShadowMapResolution; a.point_shadow_map_count = s.pointShadowMapCount;
        a.directional_shadow_map_resolution = s.directionalShadowMapResolution;
        a.directional_shadow_cascade_levels = s.directionalShadowCascadeLevels;
        a.vsync = s.vsync; a.exposure_bias = s.exposureBias; a.gamma = s.gamma; a.msaa = (uint32_t)s.msaa;
        const int rc = RDR_ABI(create)(&a, width, height, device, &ctx);
        if (rc != RDR_OK) throw std::runtime_error("renderer creation failed with status " + std::to_string(rc));
        _camera = Camera::Perspective(rdr::make3(0, 0, 0), rdr::make3(0, 0, 0), 0.1f, 60.0f, 40.0f * 0.01745329251994329577f,
                                      (float)width / (float)height);  // VulkanRenderer.h:117, VulkanRenderer.cpp:50
    }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    ~Renderer() {
        stop();
        if (ctx) RDR_ABI(destroy)(ctx);
    }

    // spawns the free-running render thread of the reference (VulkanRenderer.cpp:495-498)
    void start() {
        if (running.exchange(true)) return;
        renderThread = std::thread([this] {
            while (running.load()) {
                std::vector<uint8_t> img = renderFrame(FrameParams{});
                std::lock_guard<std::mutex> g(latestMutex);
                latest.swap(img);
                framesRendered++;
            }
        });
    }
    void stop() {
        if (!running.exchange(false)) return;
        if (renderThread.joinable()) renderThread.join();
    }
    std::vector<uint8_t> latestFrame(uint64_t* count = nullptr) {
        std::lock_guard<std::mutex> g(latestMutex);
        if (count) *count = framesRendered;
        return latest;
    }

    void setRootNodes(const std::vector<Node*>& nodes) { rootNodes = nodes; }
    void setEnvironmentMap(const std::array<TextureInfo, 6>& faces) {
        const float* ptrs[6];
        for (int i = 0; i < 6; i++) {
            if (faces[i].width != faces[0].width || faces[i].height != faces[0].height ||
                faces[i].data.size() < (size_t)faces[i].width * faces[i].height * 16)
                throw std::runtime_error("setEnvironmentMap: faces must be RGBA32F images of one size");
            ptrs[i] = reinterpret_cast<const float*>(faces[i].data.data());
        }
        check(RDR_ABI(set_environment_map)(ctx, ptrs, faces[0].width, faces[0].height), "setEnvironmentMap");
    }
    // the reference loads textures/ibl_brdf_lut.png itself (VulkanRenderer.cpp:44-45); here the decoded pixels are handed over
    void setBrdfLut(const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels) {
        check(RDR_ABI(set_brdf_lut)(ctx, pixels, width, height, channels), "setBrdfLut");
    }
    // No counterpart in the reference (main.cpp:434-441 decodes six face files): an equirectangular RGBA32F panorama is
    // resampled to faceRes^2 cube faces on the GPU, then baked as setEnvironmentMap does.
    void setEnvironmentPanorama(const TextureInfo& panorama, uint32_t faceRes) {
        if (panorama.data.size() < (size_t)panorama.width * panorama.height * 16)
            throw std::runtime_error("setEnvironmentPanorama: the panorama must be an RGBA32F image");
        check(RDR_ABI(set_environment_equirect)(ctx, reinterpret_cast<const float*>(panorama.data.data()), panorama.width, panorama.height, faceRes),
              "setEnvironmentPanorama");
    }
    // Builds the table that textures/ibl_brdf_lut.png holds instead of loading it.
    void generateBrdfLut(uint32_t size = 400, uint32_t samples = 1024) { check(RDR_ABI(generate_brdf_lut)(ctx, size, samples), "generateBrdfLut"); }
    void setLights(const std::vector<PointLight>& pointLights, const DirectionalLight& d) {
        std::vector<rdr_point_light> pl(pointLights.size());
        for (size_t i = 0; i < pl.size(); i++) {
            const PointLight& s = pointLights[i];
            pl[i] = rdr_point_light{{s.point.x, s.point.y, s.point.z}, {s.intensity.x, s.intensity.y, s.intensity.z},
                                    (uint32_t)(s.flags & (uint8_t)PointLightFlagBits::eCastShadows)};
        }
        const rdr_directional_light dl = {{d.position.x, d.position.y, d.position.z}, {d.intensity.x, d.intensity.y, d.intensity.z},
                                          {d.orientation.w, d.orientation.x, d.orientation.y, d.orientation.z}};
        // the cascade splits are fixed from the camera's near/far at this moment (VulkanRendererShadow.cpp:152-175)
        const rdr_camera cam = _camera.toAbi();
        check(RDR_ABI(set_camera)(ctx, &cam), "setLights/camera");
        check(RDR_ABI(set_lights)(ctx, pl.empty() ? nullptr : pl.data(), (uint32_t)pl.size(), &dl), "setLights");
    }
    Buffer loadBuffer(const void* ptr, size_t size) {
        uint32_t id = RDR_INVALID_ID;
        check(RDR_ABI(load_buffer)(ctx, ptr, size, &id), "loadBuffer");
        return Buffer(id);
    }
    Image loadImage(const void* ptr, size_t size, uint32_t width, uint32_t height, ImageFormat format, uint32_t maxMipLevels = UINT32_MAX) {
        uint32_t id = RDR_INVALID_ID;
        check(RDR_ABI(load_image)(ctx, ptr, size, width, height, (int)format, maxMipLevels, &id), "loadImage");
        return Image(id);
    }
    Texture makeTexture(Image image, Sampler sampler) {
        const rdr_sampler s = {(uint32_t)sampler.magFilter, (uint32_t)sampler.minFilter, (uint32_t)sampler.mipmapFilter,
                               (uint32_t)sampler.wrapU, (uint32_t)sampler.wrapV};
        uint32_t id = RDR_INVALID_ID;
        check(RDR_ABI(make_texture)(ctx, (uint32_t)image, &s, &id), "makeTexture");
        return Texture(id);
    }
    void addMesh(const Mesh& mesh) {
        std::vector<rdr_primitive_desc> descs(mesh.primitives.size());
        std::vector<std::vector<rdr_vertex_attribute>> attrs(mesh.primitives.size());
        for (size_t i = 0; i < descs.size(); i++) {
            const MeshPrimitive& p = mesh.primitives[i];
            for (const VertexAttributeDescription& v : p.vertexBufferDescription()) {
                uint32_t name;
                if (v.attributeName == "POSITION") name = RDR_ATTR_POSITION;
                else if (v.attributeName == "NORMAL") name = RDR_ATTR_NORMAL;
                else if (v.attributeName == "TANGENT") name = RDR_ATTR_TANGENT;
                else if (v.attributeName == "TEXCOORD_0") name = RDR_ATTR_TEXCOORD_0;
                else if (v.attributeName == "BITANGENT") name = RDR_ATTR_BITANGENT;
                else if (v.attributeName == "JOINTS_0") name = RDR_ATTR_JOINTS_0;
                else if (v.attributeName == "WEIGHTS_0") name = RDR_ATTR_WEIGHTS_0;
                else continue;  // attributes the shaders do not read
                attrs[i].push_back(rdr_vertex_attribute{name, (uint32_t)v.buffer, v.offset, v.stride, (uint32_t)v.containerType, v.count,
                                                        (uint32_t)v.valueType, 0});
            }
            rdr_primitive_desc& d = descs[i];
            std::memset(&d, 0, sizeof d);
            d.attributes = attrs[i].data(); d.attribute_count = (uint32_t)attrs[i].size();
            d.mode = (uint32_t)p.mode();
            const IndexBufferDescription& ib = p.indexBufferDescription();
            d.indices = rdr_index_buffer{p.isIndexed() ? (uint32_t)ib.buffer : RDR_INVALID_ID, (uint32_t)ib.indexType, ib.offset, ib.count};
            const vec3 mn = p.bbMin(), mx = p.bbMax();
            d.bb_min[0] = mn.x; d.bb_min[1] = mn.y; d.bb_min[2] = mn.z;
            d.bb_max[0] = mx.x; d.bb_max[1] = mx.y; d.bb_max[2] = mx.z;
            const Material& m = p.material;
            std::memcpy(d.material.base_color_factor, m.materialInfo.baseColorFactor, 16);
            std::memcpy(d.material.emissive_factor, m.materialInfo.emissiveFactor, 16);
            d.material.normal_scale = m.materialInfo.normalScale; d.material.metallic_factor = m.materialInfo.metallicFactor;
            d.material.roughness_factor = m.materialInfo.roughnessFactor; d.material.ao_factor = m.materialInfo.aoFactor;
            d.material.alpha_mode = (int32_t)m.alphaInfo.alphaMode; d.material.alpha_cutoff = m.alphaInfo.alphaCutoff;
            d.material.albedo_texture = (uint32_t)m.albedoTexture; d.material.normal_texture = (uint32_t)m.normalTexture;
            d.material.metal_roughness_texture = (uint32_t)m.metalRoughnessTexture; d.material.ao_texture = (uint32_t)m.aoTexture;
            d.material.emissive_texture = (uint32_t)m.emissiveTexture;
        }
        const mat4 model = mesh.node ? mesh.node->modelMatrix() : rdr::mat_identity();
        uint32_t id = 0;
        const int isStatic = !mesh.skin && (mesh.node ? mesh.node->isStatic() : true);
        check(RDR_ABI(add_mesh)(ctx, descs.data(), (uint32_t)descs.size(), model.m, isStatic, &id), "addMesh");
        meshes.push_back(MeshSlot{id, mesh.node, mesh.skin});
    }

    Camera& camera() { return _camera; }
    RendererSettings& settings() { return _settings; }
    uint32_t width() const { return _width; }
    uint32_t height() const { return _height; }

    // One iteration of renderLoop (VulkanRenderer.cpp:1013-1110): reads every mesh's node->modelMatrix() (:975), the
    // camera and the two settings the loop re-reads each frame, renders, returns width*height*4 bytes (RGBA8), or with
    // hdrOutputEnabled width*height*8 bytes (RGBA16F).
    size_t outputPixelBytes() const { return hdrOutput ? 8u : 4u; }
    std::vector<uint8_t> renderFrame(const FrameParams& fp, float* avgLogLuminance = nullptr) {
        std::vector<uint8_t> out((size_t)_width * _height * outputPixelBytes());
        renderFrameInto(fp, out.data(), false, avgLogLuminance);
        return out;
    }
    // `dst` is a host pointer, or a device pointer when dstIsDevice (the frame then stays in HBM)
    void renderFrameInto(const FrameParams& fp, uint8_t* dst, bool dstIsDevice, float* avgLogLuminance = nullptr) {
        for (const MeshSlot& m : meshes)
            if (m.node) {
                const mat4 model = m.node->modelMatrix();
                check(RDR_ABI(set_mesh_transform)(ctx, m.id, model.m), "mesh transform");
                if (m.skin) {
                    const std::vector<mat4> palette = m.skin->jointMatrices(model);
                    check(RDR_ABI(set_mesh_joints)(ctx, m.id, palette[0].m, (uint32_t)palette.size()), "mesh joints");
                }
            }
        const rdr_camera cam = _camera.toAbi();
        check(RDR_ABI(set_camera)(ctx, &cam), "camera");
        check(RDR_ABI(set_exposure_bias)(ctx, _settings.exposureBias, _settings.gamma), "settings");
        const rdr_frame_params p = {fp.exposureMode, fp.deltaTime, fp.temporalLuminance, fp.exposure};
        check(RDR_ABI(render_frame)(ctx, &p, dst, dstIsDevice ? 1 : 0, avgLogLuminance), "renderFrame");
    }
    // Frames rendered into device memory are only enqueued (up to two in flight, rdr.h): waitFrame(i) returns once the
    // i-th frame of this renderer (counted from 1) is complete, synchronize() once everything enqueued is.
    void waitFrame(uint64_t frameIndex) { check(RDR_ABI(wait_frame)(ctx, frameIndex), "waitFrame"); }
    void synchronize() { check(RDR_ABI(synchronize)(ctx), "synchronize"); }
    rdr_frame_stats frameStats() {
        rdr_frame_stats s;
        check(RDR_ABI(get_frame_stats)(ctx, &s), "frameStats");
        return s;
    }
    rdr_ctx* handle() { return ctx; }

private:
    struct MeshSlot { uint32_t id; std::shared_ptr<Node> node; std::shared_ptr<Skin> skin; };
    void check(int rc, const char* what) {
        if (rc != RDR_OK) throw std::runtime_error(std::string(what) + ": status " + std::to_string(rc) + ": " + RDR_ABI(last_error)(ctx));
    }
    rdr_ctx* ctx = nullptr;
    RendererSettings _settings;
    Camera _camera;
    uint32_t _width, _height;
    bool hdrOutput;  // fixed at creation: settings().hdrOutputEnabled is not re-read (the swap chain is not rebuilt either)
    std::vector<Node*> rootNodes;
    std::vector<MeshSlot> meshes;
    std::atomic<bool> running{false};
    std::thread renderThread;
    std::mutex latestMutex;
    std::vector<uint8_t> latest;
    uint64_t framesRendered = 0;
};

using VulkanRenderer = Renderer;  // the name host code written against the reference uses

}  // namespace hiprenderer
