// A host in the style of Renderer/main.cpp:432-462 that renders a frame batch on several GPUs: one process per GPU,
// the scene built on each through the reference's scene API (include/renderer.hpp), frame k of the job on rank
// k mod world, finished frames brought to rank 0 through include/rdr_gather.h (RCCL point-to-point group), the
// per-frame average luminance exchanged between all ranks. The gather of a frame is started once the frame behind
// it has been enqueued, so that it overlaps that frame's rendering.
//
//   shard_host out.raw n_frames world rank id_file [device]
//
// world = 0: the same frames one at a time on one GPU into host memory, without the gather library (what the sharded
// run must reproduce byte for byte). Rank 0 writes n_frames RGBA8 images to out.raw and one line
// "luminance k value" per frame to stdout. The rendezvous token travels through `id_file`.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../include/rdr_gather.h"
#include "../../include/renderer.hpp"

using namespace hiprenderer;

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 4; } } while (0)
#define GATHER_OK(expr) do { int r_ = (expr); if (r_ != RDR_GATHER_OK) { std::fprintf(stderr, "%s: status %d: %s\n", #expr, r_, rdr_gather_last_error(comm)); return 5; } } while (0)

struct Geo { std::vector<float> pos, nrm, tan, uv; std::vector<uint16_t> idx; };

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

static MeshPrimitive upload(Renderer& r, const Geo& g, const Material& m) {
    std::vector<uint8_t> bin;
    auto append = [&](const void* p, size_t n) { const size_t o = bin.size(); bin.resize(o + n); std::memcpy(bin.data() + o, p, n); return o; };
    const size_t oPos = append(g.pos.data(), g.pos.size() * 4), oNrm = append(g.nrm.data(), g.nrm.size() * 4);
    const size_t oTan = append(g.tan.data(), g.tan.size() * 4), oUv = append(g.uv.data(), g.uv.size() * 4);
    const size_t oIdx = append(g.idx.data(), g.idx.size() * 2);
    const Buffer buf = r.loadBuffer(bin.data(), bin.size());
    const size_t nv = g.pos.size() / 3;
    std::vector<VertexAttributeDescription> attrs = {
        {"POSITION", buf, oPos, 12, nv, AttributeContainerType::eVec3, AttributeValueType::eFloat},
        {"NORMAL", buf, oNrm, 12, nv, AttributeContainerType::eVec3, AttributeValueType::eFloat},
        {"TANGENT", buf, oTan, 16, nv, AttributeContainerType::eVec4, AttributeValueType::eFloat},
        {"TEXCOORD_0", buf, oUv, 8, nv, AttributeContainerType::eVec2, AttributeValueType::eFloat},
    };
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i < nv; i++)
        for (int k = 0; k < 3; k++) { mn[k] = std::fmin(mn[k], g.pos[i * 3 + k]); mx[k] = std::fmax(mx[k], g.pos[i * 3 + k]); }
    MeshPrimitive p(attrs, IndexBufferDescription{buf, oIdx, 2, g.idx.size(), AttributeValueType::eUint16}, rdr::make3(mn[0], mn[1], mn[2]),
                    rdr::make3(mx[0], mx[1], mx[2]), MeshPrimitiveMode::eTriangles);
    p.material = m;
    return p;
}

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s out.raw n_frames world rank id_file [device]\n", argv[0]); return 2; }
    const int nFrames = std::atoi(argv[2]), world = std::atoi(argv[3]), rank = std::atoi(argv[4]);
    const char* idFile = argv[5];
    const int device = argc > 6 ? std::atoi(argv[6]) : rank;
    const uint32_t W = 160, H = 96;
    const size_t frameBytes = (size_t)W * H * 4;
    rdr_gather* comm = nullptr;
    try {
        RendererSettings st;
        st.directionalShadowMapResolution = 256;
        st.pointShadowMapResolution = 64;
        st.msaa = SampleCount::e4;
        Renderer r(W, H, st, device);

        std::vector<uint8_t> tex(32 * 32 * 4);
        for (int y = 0; y < 32; y++)
            for (int x = 0; x < 32; x++) {
                const bool on = ((x / 4) + (y / 4)) & 1;
                uint8_t* p = &tex[(y * 32 + x) * 4];
                p[0] = on ? 230 : 60; p[1] = on ? 200 : 80; p[2] = on ? 150 : 200; p[3] = 255;
            }
        const Texture checker = r.makeTexture(r.loadImage(tex.data(), tex.size(), 32, 32, ImageFormat::eR8G8B8A8Srgb), Sampler{});
        auto floorNode = std::make_shared<Node>(rdr::mat_identity(), std::vector<std::shared_ptr<Node>>{});
        auto plateNode = std::make_shared<Node>(rdr::make3(0.3f, 0.8f, 0.0f), rdr::quat_from_euler(rdr::make3(0.5f, 0.2f, 0.0f)), rdr::make3(1, 1, 1),
                                                std::vector<std::shared_ptr<Node>>{});
        plateNode->setStatic(false);
        r.setRootNodes({floorNode.get(), plateNode.get()});
        Material mFloor;
        mFloor.albedoTexture = checker;
        mFloor.materialInfo.metallicFactor = 0.0f; mFloor.materialInfo.roughnessFactor = 0.7f;
        Material mPlate;
        mPlate.materialInfo.baseColorFactor[0] = 0.9f; mPlate.materialInfo.baseColorFactor[1] = 0.3f; mPlate.materialInfo.baseColorFactor[2] = 0.2f;
        mPlate.materialInfo.metallicFactor = 0.8f; mPlate.materialInfo.roughnessFactor = 0.3f;
        r.addMesh(Mesh({upload(r, grid(8.0f, 8, 4.0f), mFloor)}, floorNode));
        r.addMesh(Mesh({upload(r, grid(1.2f, 4, 1.0f), mPlate)}, plateNode));
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
        r.generateBrdfLut(32, 64);
        DirectionalLight sun;
        sun.position = rdr::make3(0, 100, 0); sun.intensity = rdr::make3(10, 7, 2);
        sun.orientation = rdr::quat_look_at(rdr::normalize(rdr::make3(0.05f, -1.0f, 0.05f)), rdr::make3(1, 0, 0));
        r.setLights({PointLight(rdr::make3(1.0f, 2.0f, 1.0f), rdr::make3(12, 10, 8), true), PointLight(rdr::make3(-2.0f, 1.5f, 0.5f), rdr::make3(3, 4, 6), false)}, sun);

        auto setFrame = [&](int k) {  // camera path and the moving plate of frame k of the job
            r.camera().setPosition(rdr::make3(0.2f + 0.15f * k, 1.6f, 4.0f - 0.1f * k));
            plateNode->setTranslation(rdr::make3(0.3f - 0.1f * k, 0.8f + 0.05f * k, 0.0f));
            plateNode->recalculateModel(rdr::mat_identity());
        };
        FrameParams fp;
        fp.exposureMode = 1; fp.temporalLuminance = 0.5f;
        r.camera().tilt(vec2{0.05f, -0.25f});

        if (world == 0) {  // one GPU, one frame at a time
            FILE* fo = std::fopen(argv[1], "wb");
            if (!fo) return 3;
            for (int k = 0; k < nFrames; k++) {
                setFrame(k);
                float avg = 0.0f;
                const std::vector<uint8_t> img = r.renderFrame(fp, &avg);
                std::fwrite(img.data(), 1, img.size(), fo);
                std::printf("luminance %d %.9g\n", k, avg);
            }
            std::fclose(fo);
            return 0;
        }

        // ---- rendezvous: rank 0 makes the token, the others wait for the file
        uint8_t id[RDR_GATHER_ID_BYTES];
        if (rank == 0) {
            if (rdr_gather_make_id(id) != RDR_GATHER_OK) { std::fprintf(stderr, "rdr_gather_make_id failed\n"); return 5; }
            const std::string tmp = std::string(idFile) + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f) return 3;
            std::fwrite(id, 1, sizeof id, f);
            std::fclose(f);
            std::rename(tmp.c_str(), idFile);
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 600 && !(f = std::fopen(idFile, "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (!f || std::fread(id, 1, sizeof id, f) != sizeof id) { std::fprintf(stderr, "no rendezvous token at %s\n", idFile); return 5; }
            std::fclose(f);
        }
        GATHER_OK(rdr_gather_create(id, rank, world, device, &comm));

        // ---- frames: three destinations in rotation; the gather of round j starts once round j + 1 is enqueued
        const int rounds = (nFrames + world - 1) / world, depth = 3;
        uint8_t* dst[depth];
        uint8_t* recv[depth] = {nullptr, nullptr, nullptr};
        for (int s = 0; s < depth; s++) {
            HIP_OK(hipMalloc((void**)&dst[s], frameBytes));
            HIP_OK(hipMemset(dst[s], 0, frameBytes));  // ranks without a frame in the last round send this
            if (rank == 0) HIP_OK(hipMalloc((void**)&recv[s], frameBytes * world));
        }
        std::vector<uint8_t> all(rank == 0 ? frameBytes * (size_t)nFrames : 0);
        std::vector<float> lum((size_t)rounds, 0.0f);
        uint64_t rendered = 0;
        std::vector<uint64_t> frameOfRound((size_t)rounds, 0);
        auto collect = [&](int j) -> int {  // round j: wait for the frame, start its gather
            if (frameOfRound[j]) {
                r.waitFrame(frameOfRound[j]);
                lum[j] = r.frameStats().avg_log_luminance;
            }
            GATHER_OK(rdr_gather_frames(comm, dst[j % depth], recv[j % depth], frameBytes));
            return 0;
        };
        auto retire = [&](int j) -> int {  // round j: its gather has finished; rank 0 keeps the frames
            GATHER_OK(rdr_gather_wait(comm));
            if (rank == 0)
                for (int src = 0; src < world && j * world + src < nFrames; src++)
                    HIP_OK(hipMemcpy(all.data() + (size_t)(j * world + src) * frameBytes, recv[j % depth] + (size_t)src * frameBytes, frameBytes, hipMemcpyDeviceToHost));
            return 0;
        };
        for (int j = 0; j < rounds; j++) {
            if (j >= 2) { const int rc = retire(j - 2); if (rc) return rc; }  // frees destination (j - 2) % depth... and (j + 1) % depth with it
            const int k = j * world + rank;
            if (k < nFrames) {
                setFrame(k);
                r.renderFrameInto(fp, dst[j % depth], true);
                frameOfRound[j] = ++rendered;
            }
            if (j >= 1) { const int rc = collect(j - 1); if (rc) return rc; }
        }
        if (rounds >= 1) { const int rc = collect(rounds - 1); if (rc) return rc; }
        for (int j = std::max(0, rounds - 2); j < rounds; j++) { const int rc = retire(j); if (rc) return rc; }

        // ---- one float per frame to every rank (what the exposure recurrence over the sequence needs)
        float *dMine = nullptr, *dAll = nullptr;
        HIP_OK(hipMalloc((void**)&dMine, (size_t)rounds * 4));
        HIP_OK(hipMalloc((void**)&dAll, (size_t)rounds * world * 4));
        HIP_OK(hipMemcpy(dMine, lum.data(), (size_t)rounds * 4, hipMemcpyHostToDevice));
        GATHER_OK(rdr_gather_all_floats(comm, dMine, dAll, (size_t)rounds));
        GATHER_OK(rdr_gather_wait(comm));
        std::vector<float> table((size_t)rounds * world);
        HIP_OK(hipMemcpy(table.data(), dAll, table.size() * 4, hipMemcpyDeviceToHost));
        if (rank == 0) {
            for (int k = 0; k < nFrames; k++) std::printf("luminance %d %.9g\n", k, table[(size_t)(k % world) * rounds + k / world]);
            FILE* fo = std::fopen(argv[1], "wb");
            if (!fo) return 3;
            std::fwrite(all.data(), 1, all.size(), fo);
            std::fclose(fo);
        }
        (void)hipFree(dMine); (void)hipFree(dAll);
        for (int s = 0; s < depth; s++) { (void)hipFree(dst[s]); if (recv[s]) (void)hipFree(recv[s]); }
        rdr_gather_destroy(comm);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
