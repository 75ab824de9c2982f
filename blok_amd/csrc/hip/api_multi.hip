// Single-process, several-device frame (include/blok_hip.h: blok_hip_multi_*): the screen-tile partition of SURVEY.md §8(e)
// driven from ONE host thread — one context and one stream per device, every device traces its tiles of the frame from its own
// replica of the world, the RGBA8 tiles travel to the root device (device 0 of the list) and are un-permuted there.  No reference
// counterpart: blok is single-GPU.
//
// Exchange.  "sparse-pull" (the default when it can be had): every device compacts its tiles with at least one hit into 16-bit
// (material, face) code records (blok_hip_compact_hit_tile_frames_device) in its OWN memory, and the root's assembly kernel reads the
// counts and the records straight out of the peers' memory through peer mappings — fine-grained loads over each peer's own xGMI link —
// and expands them (blok_hip.h: the coded exchange).  Nothing is staged, no size ever reaches the host, only live records cross a
// link, and the frame is bit-identical.  Needs peer access from the root to every device and a material table that fits the codes.
// "dense": every rank's RGBA8 tiles travel whole, then an un-permute kernel — over:
// Transport.  RCCL (loaded at run time with dlopen, so the library has no link-time dependency on it): one communicator per
// device from ncclCommInitAll, one ncclGroupStart/End per frame holding every peer's ncclSend and the root's matching ncclRecvs —
// each peer's tiles cross its own xGMI link to the root.  Peer copy (hipMemcpyPeerAsync, same links, no RCCL): the fallback
// when RCCL is absent, and the only transport when a device appears twice in the list (RCCL refuses that; it is how the
// one-GPU test box rehearses several ranks).  One device: no transport at all.
#include "api_internal.h"

#include <dlfcn.h>

namespace {

// The six RCCL entry points used, by their documented C signatures (rccl.h); types reduced to what crosses the call.
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    int (*Recv)(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv; }
};
constexpr int kNcclUint32 = 3;          // ncclUint32 in ncclDataType_t (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3)

Rccl load_rccl() {
    Rccl r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return r;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    return r;
}

}  // namespace

struct blok_hip_multi {
    struct Rank {
        int device = 0;
        blok_hip_ctx* ctx = nullptr;
        hipStream_t stream = nullptr;
        uint32_t* d_rgba = nullptr;          // dense exchange: this rank's RGBA8 tiles, frames x per_rank tiles (rank 0: its slot of the gathered buffer)
        blok_hit* d_hits = nullptr;          // this rank's first-hit records, frames x per_rank tiles, tile order (stay on the device)
        uint32_t* d_codes = nullptr;         // sparse-pull exchange: counts + code records of the batch (blok_hip_compact_hit_tile_frames_device)
        void* comm = nullptr;
        hipEvent_t traced = nullptr;
    };
    std::vector<Rank> ranks;
    uint32_t width = 0, height = 0, tile = 32, per_rank = 0;
    uint32_t frames_capacity = 0;            // frames per call the buffers are sized for
    uint32_t* d_gathered = nullptr;          // root, dense: n_ranks x frames x per_rank x tile^2
    uint32_t* d_frame = nullptr;             // root: frames x width x height
    uint8_t* d_tile_state = nullptr;         // root, sparse-pull: which tiles of d_frame hold something other than sky
    const uint32_t** d_rank_ptrs = nullptr;  // root, sparse-pull: every rank's d_codes
    hipEvent_t assembled = nullptr;          // root stream: behind the assembly (scatter / un-permute) of the previous call — the last reader of every rank's d_codes / d_gathered
    bool has_assembled = false;
    bool frame_is_sky = false;               // d_frame / d_tile_state are in the state sparse-pull assumes (all sky / all zero, or left by it)
    bool peer_readable = false;              // the root can read every device's memory
    bool peer_denied = false, peer_readable_real = false;      // blok_hip_multi_debug_deny_peer_access
    int exchange = -1;                       // -1 = sparse-pull when possible, 0 = dense, 1 = sparse-pull (refused when impossible)
    Rccl rccl;
    bool use_rccl = false;
    std::string transport = "none";
    std::string error;
};

namespace {

thread_local std::string g_multi_create_error;

int fail(blok_hip_multi* m, int status, const std::string& msg) {
    if (m) m->error = msg; else g_multi_create_error = msg;
    return status;
}

#define MULTI_TRY(m, call)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(m, e_ == hipErrorOutOfMemory ? BLOK_ERR_OOM : BLOK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int rank_error(blok_hip_multi* m, size_t i, int rc) {
    return fail(m, rc, "device " + std::to_string(m->ranks[i].device) + " (rank " + std::to_string(i) + "): " + blok_hip_last_error(m->ranks[i].ctx));
}

void destroy(blok_hip_multi* m) {
    if (!m) return;
    for (auto& r : m->ranks) {
        if (!r.ctx) continue;                                   // a rank whose creation failed owns nothing (and may name no device)
        (void)hipSetDevice(r.device);
        if (r.comm && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(r.comm);
        if (r.traced) (void)hipEventDestroy(r.traced);
        if (r.d_hits) (void)hipFree(r.d_hits);
        if (r.d_codes) (void)hipFree(r.d_codes);
        if (r.d_rgba && &r != &m->ranks[0]) (void)hipFree(r.d_rgba);
        if (r.stream) { (void)hipStreamSynchronize(r.stream); (void)hipStreamDestroy(r.stream); }
        if (r.ctx) blok_hip_destroy(r.ctx);
    }
    if (!m->ranks.empty() && m->ranks[0].ctx) (void)hipSetDevice(m->ranks[0].device);
    if (m->assembled) (void)hipEventDestroy(m->assembled);
    if (m->d_gathered) (void)hipFree(m->d_gathered);
    if (m->d_frame) (void)hipFree(m->d_frame);
    if (m->d_tile_state) (void)hipFree(m->d_tile_state);
    if (m->d_rank_ptrs) (void)hipFree(m->d_rank_ptrs);
    if (m->rccl.lib) dlclose(m->rccl.lib);
    delete m;
}

// Buffers for `frames` frames per call (grown on demand; everything in flight is finished first).
int ensure_frames(blok_hip_multi* m, uint32_t frames) {
    if (frames <= m->frames_capacity) return BLOK_OK;
    const uint32_t G = static_cast<uint32_t>(m->ranks.size());
    const size_t tile_px = static_cast<size_t>(m->per_rank) * m->tile * m->tile;
    const size_t n_px = static_cast<size_t>(m->width) * m->height;
    const size_t tiles_total = static_cast<size_t>((m->width + m->tile - 1) / m->tile) * ((m->height + m->tile - 1) / m->tile);
    for (auto& r : m->ranks) { MULTI_TRY(m, hipSetDevice(r.device)); MULTI_TRY(m, hipStreamSynchronize(r.stream)); }
    for (uint32_t i = 0; i < G; ++i) {
        auto& r = m->ranks[i];
        MULTI_TRY(m, hipSetDevice(r.device));
        if (r.d_hits) (void)hipFree(r.d_hits);
        if (r.d_codes) (void)hipFree(r.d_codes);
        if (r.d_rgba && i != 0) (void)hipFree(r.d_rgba);
        r.d_hits = nullptr; r.d_codes = nullptr; r.d_rgba = nullptr;
        MULTI_TRY(m, hipMalloc(reinterpret_cast<void**>(&r.d_hits), frames * tile_px * sizeof(blok_hit)));
        MULTI_TRY(m, hipMalloc(reinterpret_cast<void**>(&r.d_codes), frames * blok_hip_compact_code_words(m->tile, m->per_rank) * sizeof(uint32_t)));
        if (i == 0) {
            for (void* p : {static_cast<void*>(m->d_gathered), static_cast<void*>(m->d_frame), static_cast<void*>(m->d_tile_state), static_cast<void*>(m->d_rank_ptrs)})
                if (p) (void)hipFree(p);
            m->d_gathered = nullptr; m->d_frame = nullptr; m->d_tile_state = nullptr; m->d_rank_ptrs = nullptr;
            MULTI_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->d_gathered), static_cast<size_t>(G) * frames * tile_px * sizeof(uint32_t)));
            MULTI_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->d_frame), frames * n_px * sizeof(uint32_t)));
            MULTI_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->d_tile_state), frames * tiles_total));
            MULTI_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->d_rank_ptrs), G * sizeof(uint32_t*)));
            r.d_rgba = m->d_gathered;                                              // the root traces straight into its slot
        } else {
            MULTI_TRY(m, hipMalloc(reinterpret_cast<void**>(&r.d_rgba), frames * tile_px * sizeof(uint32_t)));
        }
    }
    std::vector<const uint32_t*> ptrs(G);
    for (uint32_t i = 0; i < G; ++i) ptrs[i] = m->ranks[i].d_codes;
    MULTI_TRY(m, hipSetDevice(m->ranks[0].device));
    MULTI_TRY(m, hipMemcpy(m->d_rank_ptrs, ptrs.data(), G * sizeof(uint32_t*), hipMemcpyHostToDevice));
    m->frames_capacity = frames;
    m->frame_is_sky = false;
    return BLOK_OK;
}

}  // namespace

extern "C" {

const char* blok_hip_multi_last_error(const blok_hip_multi* m) { return m ? m->error.c_str() : g_multi_create_error.c_str(); }

int blok_hip_multi_create(blok_hip_multi** out, const int* device_ordinals, uint32_t n_devices, uint32_t width, uint32_t height,
                          uint32_t tile, int allow_rccl) {
    if (!out) return BLOK_ERR_INVALID_ARG;
    *out = nullptr;
    if (!device_ordinals || !n_devices || n_devices > 64 || !width || !height || tile < 16 || (tile & 15u))
        return fail(nullptr, BLOK_ERR_INVALID_ARG, "multi: 1..64 devices, a non-empty frame and a tile that is a multiple of 16");
    auto* m = new (std::nothrow) blok_hip_multi();
    if (!m) return fail(nullptr, BLOK_ERR_OOM, "host allocation failed");
    m->width = width; m->height = height; m->tile = tile;
    m->per_rank = blok_hip_tiles_for_rank(width, height, tile, 0, n_devices);          // rank 0 owns the most tiles
    m->ranks.resize(n_devices);
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; ++i)
        for (uint32_t j = 0; j < i; ++j) distinct = distinct && device_ordinals[i] != device_ordinals[j];
    int rc = BLOK_OK;
    for (uint32_t i = 0; i < n_devices && rc == BLOK_OK; ++i) {
        auto& r = m->ranks[i];
        r.device = device_ordinals[i];
        rc = blok_hip_create(&r.ctx, r.device, width, height);
        if (rc != BLOK_OK) { fail(nullptr, rc, std::string("multi: device ") + std::to_string(r.device) + ": " + blok_hip_last_error(nullptr)); break; }
        hipError_t e = hipSetDevice(r.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r.traced, hipEventDisableTiming);
        if (e != hipSuccess) { rc = fail(nullptr, e == hipErrorOutOfMemory ? BLOK_ERR_OOM : BLOK_ERR_HIP, std::string("multi: device ") + std::to_string(r.device) + ": " + hipGetErrorString(e)); break; }
    }
    if (rc == BLOK_OK && n_devices > 1) {
        if (allow_rccl && distinct) {
            m->rccl = load_rccl();
            if (m->rccl.ok()) {
                std::vector<void*> comms(n_devices, nullptr);
                const int st = m->rccl.CommInitAll(comms.data(), static_cast<int>(n_devices), device_ordinals);
                if (st == 0) {
                    for (uint32_t i = 0; i < n_devices; ++i) m->ranks[i].comm = comms[i];
                    m->use_rccl = true; m->transport = "rccl";
                }
            }
        }
        if (!m->use_rccl) {
            m->transport = "peer-copy";
            for (uint32_t i = 1; i < n_devices && distinct; ++i) {                      // let the peers write into the root's memory
                (void)hipSetDevice(m->ranks[i].device);
                const hipError_t e = hipDeviceEnablePeerAccess(m->ranks[0].device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // hipMemcpyPeerAsync stages through the host then
            }
        }
    }
    if (rc == BLOK_OK) {
        // can the root read every device's memory (sparse-pull)?  A device listed twice is trivially readable.
        m->peer_readable = true;
        (void)hipSetDevice(m->ranks[0].device);
        for (uint32_t i = 1; i < n_devices; ++i) {
            if (m->ranks[i].device == m->ranks[0].device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->ranks[0].device, m->ranks[i].device) != hipSuccess || !can) { m->peer_readable = false; (void)hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(m->ranks[i].device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { m->peer_readable = false; (void)hipGetLastError(); }
        }
        rc = ensure_frames(m, 1);
    }
    if (rc != BLOK_OK) { const std::string msg = m->error.empty() ? g_multi_create_error : m->error; destroy(m); g_multi_create_error = msg; return rc; }
    *out = m;
    return BLOK_OK;
}

void blok_hip_multi_destroy(blok_hip_multi* m) { destroy(m); }

uint32_t blok_hip_multi_device_count(const blok_hip_multi* m) { return m ? static_cast<uint32_t>(m->ranks.size()) : 0u; }
const char* blok_hip_multi_transport(const blok_hip_multi* m) { return m ? m->transport.c_str() : ""; }
blok_hip_ctx* blok_hip_multi_context(blok_hip_multi* m, uint32_t rank) { return m && rank < m->ranks.size() ? m->ranks[rank].ctx : nullptr; }

int blok_hip_multi_upload_world(blok_hip_multi* m, const blok_svo_node* nodes, size_t n_nodes, const blok_sub_chunk* sub_chunks,
                                size_t n_sub_chunks, const blok_material* materials, size_t n_materials) {
    if (!m) return BLOK_ERR_INVALID_ARG;
    for (size_t i = 0; i < m->ranks.size(); ++i) {                                      // the world is replicated (19 MB for 1024^3)
        const int rc = blok_hip_upload_world(m->ranks[i].ctx, nodes, n_nodes, sub_chunks, n_sub_chunks, materials, n_materials);
        if (rc != BLOK_OK) return rank_error(m, i, rc);
    }
    return BLOK_OK;
}

// Will this call use the sparse-pull exchange?  (Decided per call: the material table arrives with the world.)
static bool sparse_pull(const blok_hip_multi* m) {
    return m->exchange != 0 && m->peer_readable && blok_hip_exchange_code_bits(m->ranks[0].ctx) == 16u;
}

int blok_hip_multi_set_exchange(blok_hip_multi* m, int mode) {
    if (!m || mode < -1 || mode > 1) return BLOK_ERR_INVALID_ARG;
    if (mode == 1 && !m->peer_readable) return fail(m, BLOK_ERR_UNSUPPORTED, "multi: the root device cannot read every device's memory (no peer access)");
    m->exchange = mode;
    return BLOK_OK;
}

const char* blok_hip_multi_exchange(const blok_hip_multi* m) { return !m ? "" : sparse_pull(m) ? "sparse-pull" : "dense"; }

// Diagnostic: behave as if the root could not read the other devices' memory (what hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess
// report on a node without peer access): sparse-pull is then unavailable and every call takes the dense exchange over RCCL or peer copies.
int blok_hip_multi_debug_deny_peer_access(blok_hip_multi* m, int deny) {
    if (!m) return BLOK_ERR_INVALID_ARG;
    if (deny) { if (!m->peer_denied) m->peer_readable_real = m->peer_readable; m->peer_denied = true; m->peer_readable = false; }
    else if (m->peer_denied) { m->peer_denied = false; m->peer_readable = m->peer_readable_real; }
    if (m->exchange == 1 && !m->peer_readable) m->exchange = -1;          // a forced sparse-pull cannot stand
    return BLOK_OK;
}

int blok_hip_multi_draw_frames_device(blok_hip_multi* m, const blok_camera* cams, uint32_t n_frames, const uint32_t** out_rgba8_dev_on_root) {
    if (!m) return BLOK_ERR_INVALID_ARG;
    if (!cams || !n_frames || n_frames > BLOK_MAX_TILE_FRAMES) return fail(m, BLOK_ERR_INVALID_ARG, "multi: 1 to 8 cameras per call");
    int rc = ensure_frames(m, n_frames);
    if (rc != BLOK_OK) return rc;
    const uint32_t G = static_cast<uint32_t>(m->ranks.size());
    const size_t tile_px = static_cast<size_t>(m->per_rank) * m->tile * m->tile;
    const size_t n_px = static_cast<size_t>(m->width) * m->height;
    auto& root = m->ranks[0];
    const bool pull = sparse_pull(m);
    // The calls are asynchronous: the root's assembly of the PREVIOUS call may still be reading every rank's code records (through the
    // peer mappings) or the gathered buffer a peer copy is about to overwrite.  Each peer stream therefore waits for that assembly before
    // it traces, compacts or copies again (the root's own work is ordered by its stream).
    if (m->has_assembled)
        for (uint32_t i = 1; i < G; ++i) {
            MULTI_TRY(m, hipSetDevice(m->ranks[i].device));
            MULTI_TRY(m, hipStreamWaitEvent(m->ranks[i].stream, m->assembled, 0));
        }
    auto note_assembled = [&]() -> int {
        MULTI_TRY(m, hipSetDevice(root.device));
        if (!m->assembled) MULTI_TRY(m, hipEventCreateWithFlags(&m->assembled, hipEventDisableTiming));
        MULTI_TRY(m, hipEventRecord(m->assembled, root.stream));
        m->has_assembled = true;
        return BLOK_OK;
    };
    // every device traces its tiles of all the frames in one launch pair (and, sparse-pull, compacts them where they are)
    for (uint32_t i = 0; i < G; ++i) {
        auto& r = m->ranks[i];
        rc = blok_hip_trace_tile_frames_device(r.ctx, cams, n_frames, m->tile, i, G, m->per_rank, r.d_hits, pull ? nullptr : r.d_rgba, r.stream);
        if (rc == BLOK_OK && pull)
            rc = blok_hip_compact_hit_tile_frames_device(r.ctx, r.d_hits, m->tile, blok_hip_tiles_for_rank(m->width, m->height, m->tile, i, G), n_frames,
                                                         m->per_rank, r.d_codes, r.stream);
        if (rc != BLOK_OK) return rank_error(m, i, rc);
        if (pull && i != 0) { MULTI_TRY(m, hipSetDevice(r.device)); MULTI_TRY(m, hipEventRecord(r.traced, r.stream)); }
    }
    if (pull) {
        MULTI_TRY(m, hipSetDevice(root.device));
        if (!m->frame_is_sky) {            // the state the assembly kernel builds on: every tile sky, no tile marked
            MULTI_TRY(m, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m->d_frame), static_cast<int>(0xFF000000u | (230u << 16) | (200u << 8) | 160u),
                                           m->frames_capacity * n_px, root.stream));
            const size_t tiles_total = static_cast<size_t>((m->width + m->tile - 1) / m->tile) * ((m->height + m->tile - 1) / m->tile);
            MULTI_TRY(m, hipMemsetAsync(m->d_tile_state, 0, m->frames_capacity * tiles_total, root.stream));
            m->frame_is_sky = true;
        }
        for (uint32_t i = 1; i < G; ++i) MULTI_TRY(m, hipStreamWaitEvent(root.stream, m->ranks[i].traced, 0));
        // the root reads every rank's counts and live records where they lie (its own, and the peers' over xGMI) and expands them
        rc = blok_api::scatter_frames(root.ctx, true, nullptr, reinterpret_cast<const void* const*>(m->d_rank_ptrs), G, 0, m->tile, m->per_rank, n_frames,
                                      m->d_frame, m->d_tile_state, root.stream);
        if (rc != BLOK_OK) return rank_error(m, 0, rc);
        rc = note_assembled();
        if (rc != BLOK_OK) return rc;
        if (out_rgba8_dev_on_root) *out_rgba8_dev_on_root = m->d_frame;
        return BLOK_OK;
    }
    m->frame_is_sky = false;
    const size_t block = n_frames * tile_px;                       // a rank's tiles of all the frames: what travels, and the stride of the gathered buffer
    if (m->use_rccl) {
        // one group: every peer sends on its own stream (ordered behind its trace), the root posts the matching receives
        int st = m->rccl.GroupStart();
        for (uint32_t i = 1; i < G && st == 0; ++i) {
            auto& r = m->ranks[i];
            st = m->rccl.Send(r.d_rgba, block, kNcclUint32, 0, r.comm, r.stream);
            if (st == 0) st = m->rccl.Recv(m->d_gathered + i * block, block, kNcclUint32, static_cast<int>(i), root.comm, root.stream);
        }
        const int st_end = m->rccl.GroupEnd();
        if (st == 0) st = st_end;
        if (st != 0) return fail(m, BLOK_ERR_HIP, std::string("rccl: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(st) : "error"));
    } else {
        // peer copies: each peer pushes its tiles into the root's buffer on its own stream; the root waits for all of them
        for (uint32_t i = 1; i < G; ++i) {
            auto& r = m->ranks[i];
            MULTI_TRY(m, hipSetDevice(r.device));
            MULTI_TRY(m, hipMemcpyPeerAsync(m->d_gathered + i * block, root.device, r.d_rgba, r.device, block * sizeof(uint32_t), r.stream));
            MULTI_TRY(m, hipEventRecord(r.traced, r.stream));
        }
        MULTI_TRY(m, hipSetDevice(root.device));
        for (uint32_t i = 1; i < G; ++i) MULTI_TRY(m, hipStreamWaitEvent(root.stream, m->ranks[i].traced, 0));
    }
    rc = blok_hip_untile_frames_device(root.ctx, m->d_gathered, 4, m->tile, G, n_frames * m->per_rank, n_frames, m->per_rank, m->d_frame, root.stream);
    if (rc != BLOK_OK) return rank_error(m, 0, rc);
    rc = note_assembled();
    if (rc != BLOK_OK) return rc;
    if (out_rgba8_dev_on_root) *out_rgba8_dev_on_root = m->d_frame;
    return BLOK_OK;
}

int blok_hip_multi_draw_frame_device(blok_hip_multi* m, const blok_camera* cam, const uint32_t** out_rgba8_dev_on_root) {
    return blok_hip_multi_draw_frames_device(m, cam, 1, out_rgba8_dev_on_root);
}

int blok_hip_multi_synchronize(blok_hip_multi* m) {
    if (!m) return BLOK_ERR_INVALID_ARG;
    for (auto& r : m->ranks) { MULTI_TRY(m, hipSetDevice(r.device)); MULTI_TRY(m, hipStreamSynchronize(r.stream)); }
    return BLOK_OK;
}

int blok_hip_multi_draw_frames(blok_hip_multi* m, const blok_camera* cams, uint32_t n_frames, uint32_t* out_rgba8_host) {
    if (!m) return BLOK_ERR_INVALID_ARG;
    const uint32_t* frames = nullptr;
    int rc = blok_hip_multi_draw_frames_device(m, cams, n_frames, &frames);
    if (rc != BLOK_OK) return rc;
    rc = blok_hip_multi_synchronize(m);
    if (rc != BLOK_OK) return rc;
    if (out_rgba8_host) {
        MULTI_TRY(m, hipSetDevice(m->ranks[0].device));
        MULTI_TRY(m, hipMemcpy(out_rgba8_host, frames, static_cast<size_t>(n_frames) * m->width * m->height * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return BLOK_OK;
}

int blok_hip_multi_draw_frame(blok_hip_multi* m, const blok_camera* cam, uint32_t* out_rgba8_host) {
    return blok_hip_multi_draw_frames(m, cam, 1, out_rgba8_host);
}

// This rank's first-hit records of the FIRST frame of the last call, tile order (blok_hip_tiles_for_rank(...) x tile^2 records), after a
// synchronised call.
int blok_hip_multi_download_hits(blok_hip_multi* m, uint32_t rank, blok_hit* out_host, size_t capacity_records) {
    if (!m || rank >= m->ranks.size() || !out_host) return BLOK_ERR_INVALID_ARG;
    const size_t n = static_cast<size_t>(blok_hip_tiles_for_rank(m->width, m->height, m->tile, rank, static_cast<uint32_t>(m->ranks.size()))) * m->tile * m->tile;
    if (capacity_records < n) return fail(m, BLOK_ERR_INVALID_ARG, "multi: hit buffer too small");
    MULTI_TRY(m, hipSetDevice(m->ranks[rank].device));
    MULTI_TRY(m, hipMemcpy(out_host, m->ranks[rank].d_hits, n * sizeof(blok_hit), hipMemcpyDeviceToHost));
    return BLOK_OK;
}

}  // extern "C"
