// Host program written against the reference's scene API (include/renderer.hpp): builds a small scene the way
// Renderer/main.cpp does (buffers -> mesh primitives on nodes -> lights -> environment; plus a skinned flap driven by
// key-framed joints, which the reference cannot do), renders three frames and
// writes them, raw RGBA8, to argv[1]. tests/test_cpp_facade.py builds the same scene through the Python API and
// compares. Compiled twice: against the CPU checker (-DRDR_ABI_PREFIX=orc_) and against the HIP library.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../include/renderer.hpp"

using namespace hiprenderer;

struct Geo { std::vector<float> pos, nrm, tan, uv; std::vector<uint16_t> idx; };

// (n+1)^2 grid in the XZ plane facing +Y, size s, two triangles per cell wound like scenes.grid_indices
static Geo grid(float s, int n, float uvScale) {
    Geo g;
    for (int j = 0; j <= n; j++)
        for (int i = 0; i <= n; i++) {
            const float u = (float)i / n, v = (float)j / n;
            g.pos.insert(g.pos.end(), {(u - 0.5f) * s, 0.0f, (0.5f - v) * s});
            g.nrm.insert(g.nrm.end(), {0.0f, 1.0f, 0.0f});
            g.tan.insert(g.tan.end(), {1.0f, 0.0f, 0.0f, 1.0f});
            g.uv.insert(g.uv.end(), {u * uvScale, v * uvScale});
        }
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            const uint16_t a = (uint16_t)(j * (n + 1) + i), b = a + 1, c = (uint16_t)(a + n + 1), d = (uint16_t)(a + n + 2);
            g.idx.insert(g.idx.end(), {a, b, d, a, d, c});
        }
    return g;
}

static MeshPrimitive upload(Renderer& r, const Geo& g, const Material& m, const std::vector<uint8_t>* joints = nullptr,
                            const std::vector<float>* weights = nullptr) {
    std::vector<uint8_t> bin;
    auto append = [&](const void* p, size_t n) { const size_t o = bin.size(); bin.resize(o + n); std::memcpy(bin.data() + o, p, n); return o; };
    const size_t oPos = append(g.pos.data(), g.pos.size() * 4), oNrm = append(g.nrm.data(), g.nrm.size() * 4);
    const size_t oTan = append(g.tan.data(), g.tan.size() * 4), oUv = append(g.uv.data(), g.uv.size() * 4);
    const size_t oIdx = append(g.idx.data(), g.idx.size() * 2);
    const size_t oJoints = joints ? append(joints->data(), joints->size()) : 0, oWeights = weights ? append(weights->data(), weights->size() * 4) : 0;
    const Buffer buf = r.loadBuffer(bin.data(), bin.size());
    const size_t nv = g.pos.size() / 3;
    std::vector<VertexAttributeDescription> attrs = {
        {"POSITION", buf, oPos, 12, nv, AttributeContainerType::eVec3, AttributeValueType::eFloat},
        {"NORMAL", buf, oNrm, 12, nv, AttributeContainerType::eVec3, AttributeValueType::eFloat},
        {"TANGENT", buf, oTan, 16, nv, AttributeContainerType::eVec4, AttributeValueType::eFloat},
        {"TEXCOORD_0", buf, oUv, 8, nv, AttributeContainerType::eVec2, AttributeValueType::eFloat},
        {"COLOR_0", buf, 0, 0, 0, AttributeContainerType::eVec4, AttributeValueType::eFloat},  // ignored, as by the shaders
    };
    if (joints && weights) {
        attrs.push_back({"JOINTS_0", buf, oJoints, 0, nv, AttributeContainerType::eVec4, AttributeValueType::eUint8});
        attrs.push_back({"WEIGHTS_0", buf, oWeights, 0, nv, AttributeContainerType::eVec4, AttributeValueType::eFloat});
    }
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i < nv; i++)
        for (int k = 0; k < 3; k++) { mn[k] = std::fmin(mn[k], g.pos[i * 3 + k]); mx[k] = std::fmax(mx[k], g.pos[i * 3 + k]); }
    MeshPrimitive p(attrs, IndexBufferDescription{buf, oIdx, 2, g.idx.size(), AttributeValueType::eUint16}, rdr::make3(mn[0], mn[1], mn[2]),
                    rdr::make3(mx[0], mx[1], mx[2]), MeshPrimitiveMode::eTriangles);
    p.material = m;
    return p;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out.raw [device]\n", argv[0]); return 2; }
    const uint32_t W = 96, H = 64;
    RendererSettings st;
    st.directionalShadowMapResolution = 128;
    st.pointShadowMapResolution = 32;
    st.msaa = SampleCount::e4;
    try {
        Renderer r(W, H, st, argc > 2 ? std::atoi(argv[2]) : 0);

        // 16x16 checker, sRGB base colour
        std::vector<uint8_t> tex(16 * 16 * 4);
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 16; x++) {
                const bool on = ((x / 4) + (y / 4)) & 1;
                uint8_t* p = &tex[(y * 16 + x) * 4];
                p[0] = on ? 230 : 60; p[1] = on ? 200 : 80; p[2] = on ? 150 : 200; p[3] = 255;
            }
        const Texture checker = r.makeTexture(r.loadImage(tex.data(), tex.size(), 16, 16, ImageFormat::eR8G8B8A8Srgb), Sampler{});

        // node hierarchy: root -> arm -> plate; the floor hangs off an identity node
        auto plateNode = std::make_shared<Node>(rdr::make3(0.0f, 0.6f, 0.0f), rdr::quat_from_euler(rdr::make3(0.5f, 0.0f, 0.0f)), rdr::make3(1, 1, 1),
                                                std::vector<std::shared_ptr<Node>>{});
        plateNode->setStatic(false);
        auto armNode = std::make_shared<Node>(rdr::make3(0.5f, 0.0f, 0.0f), rdr::quat_from_euler(rdr::make3(0.0f, 0.4f, 0.0f)), rdr::make3(1, 1, 1),
                                              std::vector<std::shared_ptr<Node>>{plateNode});
        Node root(rdr::make3(0.0f, 0.2f, -0.5f), quat{1, 0, 0, 0}, rdr::make3(1.5f, 1.5f, 1.5f), {armNode});
        auto floorNode = std::make_shared<Node>(rdr::mat_identity(), std::vector<std::shared_ptr<Node>>{});
        r.setRootNodes({&root, floorNode.get()});

        Material mFloor;
        mFloor.albedoTexture = checker;
        mFloor.materialInfo.metallicFactor = 0.0f; mFloor.materialInfo.roughnessFactor = 0.7f;
        Material mPlate;
        mPlate.materialInfo.baseColorFactor[0] = 0.9f; mPlate.materialInfo.baseColorFactor[1] = 0.3f; mPlate.materialInfo.baseColorFactor[2] = 0.2f;
        mPlate.materialInfo.metallicFactor = 0.8f; mPlate.materialInfo.roughnessFactor = 0.3f;
        mPlate.materialInfo.emissiveFactor[2] = 0.2f;
        r.addMesh(Mesh({upload(r, grid(8.0f, 4, 4.0f), mFloor)}, floorNode));
        r.addMesh(Mesh({upload(r, grid(1.2f, 2, 1.0f), mPlate)}, plateNode));

        // a flap of two bones: bone 0 along the far edge (z = +0.5 of the flap), bone 1 at its middle, folding about x
        Geo flap = grid(1.0f, 4, 1.0f);
        std::vector<uint8_t> flapJoints;
        std::vector<float> flapWeights;
        for (size_t i = 0; i < flap.pos.size() / 3; i++) {
            const float v = 0.5f - flap.pos[i * 3 + 2];
            const float w1 = std::fmin(std::fmax(2.0f * v - 0.5f, 0.0f), 1.0f);
            flapJoints.insert(flapJoints.end(), {0, 1, 0, 0});
            flapWeights.insert(flapWeights.end(), {1.0f - w1, w1, 0.0f, 0.0f});
        }
        auto bone1 = std::make_shared<Node>(rdr::make3(0.0f, 0.0f, -0.5f), quat{1, 0, 0, 0}, rdr::make3(1, 1, 1), std::vector<std::shared_ptr<Node>>{});
        auto bone0 = std::make_shared<Node>(rdr::make3(0.0f, 0.0f, 0.5f), quat{1, 0, 0, 0}, rdr::make3(1, 1, 1), std::vector<std::shared_ptr<Node>>{bone1});
        auto flapNode = std::make_shared<Node>(rdr::make3(-1.5f, 0.8f, 0.5f), quat{1, 0, 0, 0}, rdr::make3(1, 1, 1), std::vector<std::shared_ptr<Node>>{bone0});
        bone1->setRotationAnimation(Animation<quat>({0.0f, 1.0f}, {quat{1, 0, 0, 0}, quat{std::cos(0.45f), std::sin(0.45f), 0, 0}},
                                                    AnimationInterpolationCurve::eLinear, AnimationRepeatMode::eMirror));
        bone0->setTranslationAnimation(Animation<vec3>({0.0f, 2.0f}, {rdr::make3(0, 0, 0), rdr::make3(0, 0, 0.5f), rdr::make3(0, 1, 0),
                                                                       rdr::make3(0, 0, 0), rdr::make3(0, 0.4f, 0.5f), rdr::make3(0, 0, 0)},
                                                       AnimationInterpolationCurve::eCubicSpline, AnimationRepeatMode::eClamp));
        auto skin = std::make_shared<Skin>(std::vector<std::shared_ptr<Node>>{bone0, bone1},
                                           std::vector<mat4>{rdr::mat_translate(rdr::make3(0, 0, -0.5f)), rdr::mat_identity()});
        Material mFlap;
        mFlap.albedoTexture = checker;
        mFlap.materialInfo.metallicFactor = 0.1f; mFlap.materialInfo.roughnessFactor = 0.5f;
        r.addMesh(Mesh({upload(r, flap, mFlap, &flapJoints, &flapWeights)}, flapNode, skin));

        // environment: 8x8 faces, a vertical gradient; BRDF table: 16x16 ramp
        std::array<TextureInfo, 6> env;
        for (int f = 0; f < 6; f++) {
            env[f].width = env[f].height = 8;
            std::vector<float> px(8 * 8 * 4);
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    float* p = &px[(y * 8 + x) * 4];
                    p[0] = 0.2f + 0.1f * f; p[1] = 0.3f + 0.05f * y; p[2] = 0.6f + 0.02f * x; p[3] = 1.0f;
                }
            env[f].data.resize(px.size() * 4);
            std::memcpy(env[f].data.data(), px.data(), env[f].data.size());
        }
        r.setEnvironmentMap(env);
        std::vector<uint8_t> lut(16 * 16 * 3);
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 16; x++) { lut[(y * 16 + x) * 3] = (uint8_t)(x * 16); lut[(y * 16 + x) * 3 + 1] = (uint8_t)(y * 8); lut[(y * 16 + x) * 3 + 2] = 0; }
        r.setBrdfLut(lut.data(), 16, 16, 3);

        r.camera().setPosition(rdr::make3(0.2f, 1.6f, 4.0f));
        r.camera().tilt(vec2{0.05f, -0.25f});
        DirectionalLight sun;
        sun.position = rdr::make3(0, 100, 0); sun.intensity = rdr::make3(10, 7, 2);
        sun.orientation = rdr::quat_look_at(rdr::normalize(rdr::make3(0.05f, -1.0f, 0.05f)), rdr::make3(1, 0, 0));
        r.setLights({PointLight(rdr::make3(1.0f, 2.0f, 1.0f), rdr::make3(12, 10, 8), true), PointLight(rdr::make3(-2.0f, 1.5f, 0.5f), rdr::make3(3, 4, 6), false)}, sun);

        FrameParams fp;
        fp.exposureMode = 1; fp.temporalLuminance = 0.5f;
        std::vector<uint8_t> f0 = r.renderFrame(fp);
        // move the arm: the plate follows through the hierarchy
        armNode->setTranslation(rdr::make3(-0.4f, 0.1f, 0.3f));
        root.recalculateModel(rdr::mat_identity());
        r.camera().dolly(0.5f);
        std::vector<uint8_t> f1 = r.renderFrame(fp);
        // advance the clip: the flap folds (slerp on bone 1) while bone 0 rides a spline
        flapNode->setAnimationTime(1.4f);
        flapNode->recalculateModel(rdr::mat_identity());
        std::vector<uint8_t> f2 = r.renderFrame(fp);
        const mat4 bm = bone1->modelMatrix();
        std::printf("bone1 model col3 = %.6f %.6f %.6f\n", bm.at(3, 0), bm.at(3, 1), bm.at(3, 2));
        const mat4 pm = plateNode->modelMatrix();
        std::printf("plate model col3 = %.6f %.6f %.6f\n", pm.at(3, 0), pm.at(3, 1), pm.at(3, 2));
        const vec3 cp = r.camera().position();
        std::printf("camera = %.6f %.6f %.6f\n", cp.x, cp.y, cp.z);
        FILE* fo = std::fopen(argv[1], "wb");
        if (!fo) return 3;
        std::fwrite(f0.data(), 1, f0.size(), fo);
        std::fwrite(f1.data(), 1, f1.size(), fo);
        std::fwrite(f2.data(), 1, f2.size(), fo);
        std::fclose(fo);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
