// Headless driver shaped like the reference's App (reference blok/src/app.cpp:65-192) with the backend
// switch extended by GraphicsApi::HIP: build a world through ChunkManager, rebuildDirtyChunks,
// packChunksToGpuSvo, addWorld, then a frame loop of drawFrame; writes the last frame as a PPM.
//   blok_headless [--n 256 | --vox model.vox] [--size 1280x720] [--pose 0|1|2] [--frames 10] [--out frame.ppm] [--rt [--spp 8]]
//   --rt: every frame goes through the reference's full ray-tracing path (path trace, denoise, TAA, tonemap, sharpen)
//   --devices 0,1,2,...: the frame is tile-partitioned over these devices of the node by ONE process (blok::HipMultiTracer:
//                        RCCL send / receive group or peer copies to the first device); an ordinal may repeat (rehearsal on one GPU)
//   --no-rccl: peer copies even when RCCL is there
//   --dense-exchange: whole RGBA8 tiles travel (RCCL / peer copies) instead of the root reading the ranks' sparse code records
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "blok/hip_tracer.hpp"

namespace {

struct Options {
    uint32_t n = 256, width = 1280, height = 720, frames = 10;     // reference window: 1280x720, app.cpp:95
    int pose = 0;
    std::string out = "frame.ppm";
    bool rt = false;                      // full ray-tracing path per frame instead of first-hit frames
    uint32_t spp = 8;                     // samples per pixel and frame in --rt mode (the reference forces 8)
    std::string vox;                      // optional .vox model instead of the synthetic scene (app.cpp:105-113)
    std::vector<int> devices;             // more than one entry: the multi-device tracer
    bool dense_exchange = false;
    bool rccl = true;
};

class App {
public:
    App(blok::GraphicsApi api, Options opt) : m_backend(api), m_opt(std::move(opt)), m_mgr(128, 1.0f) {}   // app.cpp:37

    void run() { init(); update(); shutdown(); }                                           // app.cpp:65-71

private:
    void init() {
        switch (m_backend) {
            case blok::GraphicsApi::HIP: {
                m_tracer = std::make_unique<blok::HipTracer>(m_opt.width, m_opt.height);
                m_tracer->init();
                if (!m_opt.vox.empty()) {
                    std::string err;
                    if (!blok::loadAndImportVox(m_opt.vox, m_mgr, &m_materials, nullptr, 0, &err))   // app.cpp:105-113
                        throw std::runtime_error("Failed to load VOX: " + err);
                    m_opt.n = 128;
                } else {
                    uint64_t writes = 0;
                    if (blok_scene_generate(m_mgr.handle(), m_opt.n, 0xB10C0001u, &writes) != BLOK_OK)
                        throw std::runtime_error("scene generation failed (n must be a power of two in [16, 4096])");
                }
                rebuildDirtyChunks(m_mgr, 1 << 30);                                       // app.cpp:120
                packChunksToGpuSvo(m_mgr, m_world);                                       // app.cpp:121
                if (!m_opt.vox.empty()) m_world.materials = m_materials.packForGpu();
                else {
                    m_world.materials.resize(256);
                    blok_scene_materials(0xB10C0001u, m_world.materials.data());
                }
                m_tracer->addWorld(m_world);                                              // app.cpp:122-124
                if (m_opt.devices.size() > 1) {
                    m_multi = std::make_unique<blok::HipMultiTracer>(m_opt.devices, m_opt.width, m_opt.height, 32, m_opt.rccl);
                    m_multi->addWorld(m_world);
                    if (m_opt.dense_exchange) m_multi->setExchange(0);
                    std::cout << "multi-device: " << m_multi->deviceCount() << " ranks, exchange " << m_multi->exchange() << ", transport " << m_multi->transport() << "\n";
                }
                const blok_world_stats s = m_tracer->worldStats();
                std::cout << "world: " << s.n_voxels << " voxels, " << s.n_ref_nodes << " SVO nodes, " << s.n_sub_chunks
                          << " sub-chunks -> " << s.n_tree_nodes << " tree nodes (" << s.tree_bytes / 1e6 << " MB), "
                          << s.levels << " levels\n";
                blok_camera c{};
                blok_scene_camera(m_opt.n, 0xB10C0001u, m_opt.pose, m_opt.width, m_opt.height, &c);
                for (int a = 0; a < 3; ++a) m_camera.position[a] = c.pos[a];
                m_camera.pitch = std::asin(c.fwd[1]) * 57.29577951308232f;
                m_camera.yaw = std::atan2(c.fwd[2], c.fwd[0]) * 57.29577951308232f;
                break;
            }
            default:
                throw std::runtime_error("this driver only carries the HIP backend");
        }
    }
    void update() {
        using clock = std::chrono::steady_clock;
        for (uint32_t f = 0; f < m_opt.frames; ++f) {
            const auto t0 = clock::now();
            m_tracer->beginFrame();
            if (m_multi) m_multi->drawFrame(m_camera, m_multiFrame);
            else if (m_opt.rt) m_tracer->drawFrameRT(m_camera, m_opt.spp);
            else m_tracer->drawFrame(m_camera);
            m_tracer->endFrame();
            const double ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
            if (m_multi) std::cout << "frame " << f << ": " << ms << " ms (" << m_multi->deviceCount() << " ranks, incl. device->host copy of the RGBA8 frame)\n";
            else if (m_opt.rt) std::cout << "frame " << f << ": " << ms << " ms (path trace " << m_opt.spp << " spp, denoise, TAA, tonemap, sharpen; incl. device->host copy)\n";
            else std::cout << "frame " << f << ": " << ms << " ms (incl. device->host copy of "
                           << m_tracer->hits().size() * sizeof(blok_hit) / 1e6 << " MB)\n";
            if (f + 1 < m_opt.frames || !m_opt.rt) m_camera.processKeyboard('W', 0.016f);
        }
        const auto& single = m_opt.rt ? m_tracer->drawFrameRT(m_camera, m_opt.spp) : m_tracer->drawFrameRgba8(m_camera);
        if (m_multi) {                                       // the partitioned frame must be the single-device frame
            m_multi->drawFrame(m_camera, m_multiFrame);
            size_t differ = 0;
            for (size_t i = 0; i < single.size(); ++i) differ += single[i] != m_multiFrame[i];
            std::cout << "multi-device frame vs single-device frame: " << differ << " pixels differ\n";
            if (differ) throw std::runtime_error("multi-device frame differs from the single-device frame");
        }
        const auto& px = m_multi ? m_multiFrame : single;
        std::ofstream ppm(m_opt.out, std::ios::binary);
        ppm << "P6\n" << m_opt.width << " " << m_opt.height << "\n255\n";
        for (uint32_t p : px) { const char rgb[3] = {char(p & 255), char((p >> 8) & 255), char((p >> 16) & 255)}; ppm.write(rgb, 3); }
        std::cout << "wrote " << m_opt.out << "\n";
    }
    void shutdown() { if (m_tracer) m_tracer->shutdown(); }

    blok::GraphicsApi m_backend;
    Options m_opt;
    blok::ChunkManager m_mgr;
    blok::MaterialLibrary m_materials;
    blok::WorldSvoGpu m_world;                     // App owns the world, the tracer its device copy (app.hpp:40)
    blok::Camera m_camera;
    std::unique_ptr<blok::HipTracer> m_tracer;
    std::unique_ptr<blok::HipMultiTracer> m_multi;
    std::vector<uint32_t> m_multiFrame;
};

}  // namespace

int main(int argc, char** argv) {
    Options opt;
    for (int i = 1; i < argc; ++i) {
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--n")) opt.n = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--size")) { if (std::sscanf(next(), "%ux%u", &opt.width, &opt.height) != 2) return 2; }
        else if (!std::strcmp(argv[i], "--pose")) opt.pose = std::atoi(next());
        else if (!std::strcmp(argv[i], "--frames")) opt.frames = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--out")) opt.out = next();
        else if (!std::strcmp(argv[i], "--vox")) opt.vox = next();
        else if (!std::strcmp(argv[i], "--rt")) opt.rt = true;
        else if (!std::strcmp(argv[i], "--spp")) opt.spp = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--no-rccl")) opt.rccl = false;
        else if (!std::strcmp(argv[i], "--dense-exchange")) opt.dense_exchange = true;
        else if (!std::strcmp(argv[i], "--devices")) { for (const char* p = next(); *p;) { opt.devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    try {
        App app(blok::GraphicsApi::HIP, opt);
        app.run();
    } catch (const std::exception& e) {
        std::cerr << "[FATAL] " << e.what() << std::endl;                                  // main.cpp:19-22
        return 1;
    }
    return 0;
}
