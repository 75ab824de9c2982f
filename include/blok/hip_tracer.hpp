// C++20 host side of the gfx950 backend, above the C ABI (blok_hip.h / blok_world.h).
//
// Mirrors the reference's own types so that an `App`-like driver calls it the way blok's App calls its
// backends (reference blok/src/app.cpp:73-128,130-192):
//   blok::GraphicsApi      reference blok/include/backend.hpp:9-12, plus the third enumerator HIP
//   blok::Camera           reference blok/include/camera.hpp:15-84 (same fields, defaults, key/mouse steps)
//   blok::WorldSvoGpu      reference blok/include/resources.hpp:195-203 (the three host arrays only)
//   blok::ChunkManager     reference blok/include/chunk_manager.hpp:18-52 (+ rebuildDirtyChunks, packChunksToGpuSvo)
//   blok::HipTracer        reference blok/include/cuda_tracer.hpp:23-58 (lifecycle) and
//                          blok/include/renderer.hpp:40-72 (addWorld / updateWorld / cleanupWorld)
// Failures throw std::runtime_error, as the reference's backends do (caught once in main,
// reference blok/src/main.cpp:19-22).  Header-only; link libblok_hip.so and libblok_host.so.
#ifndef BLOK_HIP_TRACER_HPP
#define BLOK_HIP_TRACER_HPP

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../blok_hip.h"
#include "../blok_world.h"

namespace blok {

enum class GraphicsApi { OpenGL, Vulkan, HIP };

struct Camera {
    float position[3] = {0.0f, 10.0f, -5.0f};
    float yaw = 0.0f;
    float pitch = 0.0f;
    float fov = 60.0f;
    mutable bool cameraChanged = false;

    // basis through the same code the tests pin (blok_camera_from_yaw_pitch)
    blok_camera basis(unsigned width, unsigned height) const {
        blok_camera c{};
        if (blok_camera_from_yaw_pitch(position, yaw, pitch, fov, width, height, &c) != BLOK_OK)
            throw std::runtime_error("Camera::basis: bad frame size");
        return c;
    }
    void processKeyboard(char key, float dt) {           // reference camera.hpp:61-71
        const blok_camera c = basis(1, 1);
        const float speed = 40.0f * dt;
        auto move = [&](const float d[3], float s) { for (int a = 0; a < 3; ++a) position[a] += d[a] * s; };
        const float worldUp[3] = {0.0f, 1.0f, 0.0f};
        if (key == 'W') move(c.fwd, speed);
        if (key == 'S') move(c.fwd, -speed);
        if (key == 'A') move(c.right, -speed);
        if (key == 'D') move(c.right, speed);
        if (key == 'X') move(worldUp, speed);
        if (key == 'Z') move(worldUp, -speed);
        cameraChanged = true;
    }
    void processMouse(float dx, float dy) {               // reference camera.hpp:72-81
        const float sens = 0.01f;
        yaw += dx * sens;
        pitch += dy * sens;
        if (pitch > 89.0f) pitch = 89.0f;
        if (pitch < -89.0f) pitch = -89.0f;
        cameraChanged = true;
    }
};

struct WorldSvoGpu {
    std::vector<blok_svo_node> globalNodes;
    std::vector<blok_sub_chunk> globalSubChunks;
    std::vector<blok_material> materials;
};

class ChunkManager {
public:
    uint32_t C;
    float voxelSize;

    ChunkManager(uint32_t C_, float voxelSize_) : C(C_), voxelSize(voxelSize_) {
        if (blok_world_create(&w_, C_, voxelSize_) != BLOK_OK) throw std::runtime_error("ChunkManager: bad chunk size");
    }
    ~ChunkManager() { blok_world_destroy(w_); }
    ChunkManager(const ChunkManager&) = delete;
    ChunkManager& operator=(const ChunkManager&) = delete;

    void setVoxelMaterial(const float worldPos[3], uint32_t materialId, float density = 1.0f) {
        check(blok_world_set_voxel(w_, worldPos, materialId, density));
    }
    uint32_t getVoxelMaterial(const float worldPos[3]) const { return blok_world_get_voxel_material(w_, worldPos); }
    blok_world* handle() { return w_; }

    friend int rebuildDirtyChunks(ChunkManager& mgr, int maxPerFrame) {
        const int n = blok_world_rebuild_dirty(mgr.w_, maxPerFrame);
        if (n < 0) mgr.check(n);
        return n;
    }
    friend void packChunksToGpuSvo(ChunkManager& mgr, WorldSvoGpu& gpuWorld) {
        mgr.check(blok_world_pack(mgr.w_));
        const blok_svo_node* n = blok_world_nodes(mgr.w_);
        const blok_sub_chunk* s = blok_world_sub_chunks(mgr.w_);
        gpuWorld.globalNodes.assign(n, n + blok_world_node_count(mgr.w_));
        gpuWorld.globalSubChunks.assign(s, s + blok_world_sub_chunk_count(mgr.w_));
    }

private:
    void check(int rc) const { if (rc < 0) throw std::runtime_error(std::string("ChunkManager: ") + blok_world_last_error(w_)); }
    blok_world* w_ = nullptr;
};

// = blok::Brush / applyBrush (reference blok/include/brush.hpp:14-21, blok/src/brush.cpp:13-63)
struct Brush {
    float centerWS[3];
    float radiusWS;
    float value;
    enum Mode { ADD, SUBTRACT } mode;
};
inline void applyBrush(ChunkManager& mgr, const Brush& brush) {
    if (blok_world_apply_brush(mgr.handle(), brush.centerWS, brush.radiusWS, brush.value, brush.mode == Brush::ADD ? 0 : 1) != BLOK_OK)
        throw std::runtime_error(std::string("applyBrush: ") + blok_world_last_error(mgr.handle()));
}

// = blok::MaterialLibrary (reference blok/include/material.hpp:116-163), RAII over the C ABI.
class MaterialLibrary {
public:
    MaterialLibrary() { if (blok_material_library_create(&lib_) != BLOK_OK) throw std::runtime_error("MaterialLibrary: allocation failed"); }
    ~MaterialLibrary() { blok_material_library_destroy(lib_); }
    MaterialLibrary(const MaterialLibrary&) = delete;
    MaterialLibrary& operator=(const MaterialLibrary&) = delete;
    uint32_t addMaterial(const blok_material_desc& m) { return blok_material_library_add(lib_, &m); }
    uint32_t getOrCreateFromColor(uint8_t r, uint8_t g, uint8_t b) { return blok_material_library_from_color(lib_, r, g, b); }
    uint32_t getMaterialFromVoxPalette(uint8_t i) const { return blok_material_library_from_vox_palette(lib_, i); }
    size_t size() const { return blok_material_library_size(lib_); }
    std::vector<blok_material> packForGpu() const {
        std::vector<blok_material> out(size());
        if (blok_material_library_pack(lib_, out.data(), out.size()) != BLOK_OK) throw std::runtime_error("MaterialLibrary::packForGpu");
        return out;
    }
    blok_material_library* handle() { return lib_; }
private:
    blok_material_library* lib_ = nullptr;
};

// = loadAndImportVox (reference blok/src/vox_loader.cpp:432-462)
inline bool loadAndImportVox(const std::string& filepath, ChunkManager& chunkMgr, MaterialLibrary* materialLib = nullptr,
                             const float worldOffset[3] = nullptr, uint32_t modelIndex = 0, std::string* errorMsg = nullptr) {
    char err[256] = {0};
    const int rc = blok_load_and_import_vox(filepath.c_str(), chunkMgr.handle(), materialLib ? materialLib->handle() : nullptr,
                                            worldOffset, modelIndex, err, sizeof(err));
    if (rc != BLOK_OK && errorMsg) *errorMsg = err;
    return rc == BLOK_OK;
}

class HipTracer {
public:
    HipTracer(unsigned int width, unsigned int height, int device = 0) : m_width(width), m_height(height), m_device(device) {}
    ~HipTracer() { shutdown(); }
    HipTracer(const HipTracer&) = delete;
    HipTracer& operator=(const HipTracer&) = delete;

    void init() {
        if (m_ctx) return;
        if (blok_hip_create(&m_ctx, m_device, m_width, m_height) != BLOK_OK)
            throw std::runtime_error(std::string("HipTracer::init: ") + blok_hip_last_error(nullptr));
        m_hits.resize(static_cast<size_t>(m_width) * m_height);
    }
    void shutdown() { if (m_ctx) { blok_hip_destroy(m_ctx); m_ctx = nullptr; } m_world = nullptr; }
    void beginFrame() {}
    void endFrame() {}
    void resize(unsigned int w, unsigned int h) {
        check(blok_hip_resize(m_ctx, w, h));
        m_width = w; m_height = h;
        m_hits.assign(static_cast<size_t>(w) * h, blok_hit{});
    }
    void resetAccum() { check(blok_hip_reset_accum(m_ctx)); m_frameIndex = 0; }

    // = Renderer::addWorld: keeps a non-owning pointer to the caller's world, owns the device copies
    // ChunkManager's voxelSize for the worlds added from now on (a power of two; reference chunk_manager.cpp:19-25, app.cpp:37)
    void setVoxelSize(float voxelSize) { check(blok_hip_set_voxel_size(m_ctx, voxelSize)); }
    void addWorld(WorldSvoGpu& gpuWorld) { m_world = &gpuWorld; updateWorld(); }
    void updateWorld() {
        if (!m_world) return;
        check(blok_hip_upload_world(m_ctx, m_world->globalNodes.data(), m_world->globalNodes.size(),
                                    m_world->globalSubChunks.data(), m_world->globalSubChunks.size(),
                                    m_world->materials.data(), m_world->materials.size()));
    }
    void cleanupWorld() {
        check(blok_hip_upload_world(m_ctx, nullptr, 0, nullptr, 0, nullptr, 0));
        m_world = nullptr;
    }

    // = CudaTracer::drawFrame(cam, ...): one frame of primary first-hit records, blocking
    void drawFrame(Camera& cam) {
        const blok_camera c = cam.basis(m_width, m_height);
        check(blok_hip_trace_primary(m_ctx, &c, 0, 0, m_width, m_height, m_hits.data()));
        cam.cameraChanged = false;
        ++m_frameIndex;
    }
    // RGBA8 view of the same frame (the CUDA backend's output format, reference cuda_tracer.cu:385-386)
    const std::vector<uint32_t>& drawFrameRgba8(Camera& cam) {
        const blok_camera c = cam.basis(m_width, m_height);
        m_pixels.resize(static_cast<size_t>(m_width) * m_height);
        check(blok_hip_shade_rgba8(m_ctx, &c, 0, 0, m_width, m_height, m_pixels.data()));
        return m_pixels;
    }

    // = CudaTracer::drawFrame in its progressive mode (reference cuda_tracer.cu:484-555): path-traced samples added to the
    // accumulation buffer, which is cleared when the camera moved; returns the ACES + gamma 2.2 RGBA8 running average
    const std::vector<uint32_t>& drawFrameProgressive(Camera& cam, uint32_t sppPerFrame = 1, uint32_t maxBounces = 2) {
        const blok_camera c = cam.basis(m_width, m_height);
        m_pixels.resize(static_cast<size_t>(m_width) * m_height);
        check(blok_hip_draw_frame_accumulate(m_ctx, &c, sppPerFrame, maxBounces, m_pixels.data(), &m_frameIndex));
        cam.cameraChanged = false;
        return m_pixels;
    }
    uint32_t framesAccumulated() const { return m_frameIndex; }

    // ---- device-resident dense store (blok_hip_volume_*): ChunkManager's edit API for one box of the world, with the
    // arrays, the edits and the rebuild in HBM.  rebuildVolume() = rebuildDirtyChunks + packChunksToGpuSvo + updateWorld.
    void createVolume(const int32_t origin[3], uint32_t nx, uint32_t ny, uint32_t nz, uint32_t chunkSize = 128, float voxelSize = 1.0f) {
        check(blok_hip_volume_create(m_ctx, origin, nx, ny, nz, chunkSize, voxelSize));
    }
    void uploadVolume(const float* density, const uint32_t* materialIds) { check(blok_hip_volume_upload(m_ctx, density, materialIds)); }
    void setVoxelMaterial(const int32_t voxel[3], uint32_t materialId, float density = 1.0f) {
        check(blok_hip_volume_set_voxels(m_ctx, voxel, &materialId, &density, 1));
    }
    void applyBrush(const Brush& brush) {
        check(blok_hip_volume_apply_brush(m_ctx, brush.centerWS, brush.radiusWS, brush.value, brush.mode == Brush::ADD ? 0 : 1));
    }
    void rebuildVolume(const std::vector<blok_material>& materials) { check(blok_hip_volume_rebuild(m_ctx, materials.data(), materials.size())); }

    // ---- image-space chain (Denoiser::denoise, PostProcess::process) over device planes; see include/blok_hip.h
    void denoise(const blok_gbuffer& planesDev, const float prevViewProj[16], uint32_t frameCount, float* outColorDev,
                 const blok_denoise_settings* settings = nullptr, const float* motionDev = nullptr, void* stream = nullptr) {
        check(blok_hip_denoise_device(m_ctx, &planesDev, motionDev, prevViewProj, frameCount, settings, outColorDev, stream));
    }
    void taa(const float* colorDev, float* outColorDev, uint32_t frameCount, float feedbackMin = 0.93f, float feedbackMax = 0.98f,
             const float* motionDev = nullptr, void* stream = nullptr) {
        check(blok_hip_taa_device(m_ctx, colorDev, motionDev, feedbackMin, feedbackMax, frameCount, outColorDev, stream));
    }
    // = Renderer::drawFrame's ray-tracing path: trace, denoise, TAA, tonemap, sharpen with the reference's default settings
    const std::vector<uint32_t>& drawFrameRT(Camera& cam, uint32_t sampleCount = 8, uint32_t maxBounces = 2) {
        const blok_camera c = cam.basis(m_width, m_height);
        m_pixels.resize(static_cast<size_t>(m_width) * m_height);
        check(blok_hip_draw_frame_rt(m_ctx, &c, sampleCount, maxBounces, nullptr, m_pixels.data(), &m_frameIndex));
        cam.cameraChanged = false;
        return m_pixels;
    }
    void sharpen(const uint32_t* rgba8Dev, uint32_t* outRgba8Dev, float strength = 0.5f, void* stream = nullptr) {
        check(blok_hip_sharpen_device(m_ctx, rgba8Dev, strength, outRgba8Dev, stream));
    }

    const std::vector<blok_hit>& hits() const { return m_hits; }     // output accessor (getGLTex analogue)
    unsigned int width() const { return m_width; }
    unsigned int height() const { return m_height; }
    blok_hip_ctx* handle() { return m_ctx; }
    blok_world_stats worldStats() const {
        blok_world_stats s{};
        if (blok_hip_world_stats(m_ctx, &s) != BLOK_OK) throw std::runtime_error("HipTracer: no world");
        return s;
    }

private:
    void check(int rc) const { if (rc != BLOK_OK) throw std::runtime_error(std::string("HipTracer: ") + blok_hip_last_error(m_ctx)); }

    unsigned int m_width = 0, m_height = 0;
    int m_device = 0;
    blok_hip_ctx* m_ctx = nullptr;
    WorldSvoGpu* m_world = nullptr;       // non-owning, like reference renderer.hpp:195
    uint32_t m_frameIndex = 0;
    std::vector<blok_hit> m_hits;
    std::vector<uint32_t> m_pixels;
};

// Several devices of one node, one process (blok_hip_multi_*, blok_hip.h): the frame is cut into tile x tile screen tiles dealt
// round-robin to the devices, the world is replicated, the RGBA8 tiles are gathered on the first device over xGMI (RCCL
// send / receive group, or peer copies) and un-permuted there.  No reference counterpart: blok is single-GPU.
class HipMultiTracer {
public:
    HipMultiTracer(const std::vector<int>& devices, unsigned width, unsigned height, unsigned tile = 32, bool allowRccl = true)
        : m_width(width), m_height(height) {
        if (blok_hip_multi_create(&m_multi, devices.data(), static_cast<uint32_t>(devices.size()), width, height, tile, allowRccl ? 1 : 0) != BLOK_OK)
            throw std::runtime_error(std::string("HipMultiTracer: ") + blok_hip_multi_last_error(nullptr));
    }
    ~HipMultiTracer() { shutdown(); }
    HipMultiTracer(const HipMultiTracer&) = delete;
    HipMultiTracer& operator=(const HipMultiTracer&) = delete;

    void shutdown() { if (m_multi) { blok_hip_multi_destroy(m_multi); m_multi = nullptr; } }
    unsigned deviceCount() const { return blok_hip_multi_device_count(m_multi); }
    std::string transport() const { return blok_hip_multi_transport(m_multi); }
    blok_hip_ctx* context(unsigned rank) { return blok_hip_multi_context(m_multi, rank); }

    void addWorld(const WorldSvoGpu& w) {                       // = Renderer::addWorld on every device
        check(blok_hip_multi_upload_world(m_multi, w.globalNodes.data(), w.globalNodes.size(), w.globalSubChunks.data(), w.globalSubChunks.size(),
                                          w.materials.data(), w.materials.size()));
    }
    // One frame: RGBA8, row-major, width x height, on the host.
    void drawFrame(const Camera& c, std::vector<uint32_t>& rgba8) {
        rgba8.resize(static_cast<size_t>(m_width) * m_height);
        const blok_camera basis = c.basis(m_width, m_height);
        check(blok_hip_multi_draw_frame(m_multi, &basis, rgba8.data()));
    }
    // Asynchronous form: the frame stays on the root device.
    const uint32_t* drawFrameDevice(const Camera& c) {
        const blok_camera basis = c.basis(m_width, m_height);
        const uint32_t* frame = nullptr;
        check(blok_hip_multi_draw_frame_device(m_multi, &basis, &frame));
        return frame;
    }
    void synchronize() { check(blok_hip_multi_synchronize(m_multi)); }
    // Several frames per call (1..BLOK_MAX_TILE_FRAMES cameras, one launch pair per device); the frames follow one another in rgba8.
    void drawFrames(const std::vector<Camera>& cams, std::vector<uint32_t>& rgba8) {
        std::vector<blok_camera> basis;
        for (const Camera& c : cams) basis.push_back(c.basis(m_width, m_height));
        rgba8.resize(cams.size() * static_cast<size_t>(m_width) * m_height);
        check(blok_hip_multi_draw_frames(m_multi, basis.data(), static_cast<uint32_t>(basis.size()), rgba8.data()));
    }
    // How the tiles reach the root: -1 = sparse-pull when possible (default), 0 = dense, 1 = sparse-pull or an exception (blok_hip.h).
    void setExchange(int mode) { check(blok_hip_multi_set_exchange(m_multi, mode)); }
    std::string exchange() const { return blok_hip_multi_exchange(m_multi); }

private:
    void check(int rc) const { if (rc != BLOK_OK) throw std::runtime_error(std::string("HipMultiTracer: ") + blok_hip_multi_last_error(m_multi)); }
    blok_hip_multi* m_multi = nullptr;
    unsigned m_width, m_height;
};

}  // namespace blok
#endif
