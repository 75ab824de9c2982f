// Internals shared by the translation units of the C ABI (api.hip, api_post.hip, api_volume.hip): the context, the error
// macro and the helpers defined in api.hip.  Not installed; include/blok_hip.h is the interface.
#ifndef BLOK_API_INTERNAL_H
#define BLOK_API_INTERNAL_H
#include <hip/hip_runtime.h>

#include <cmath>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "blok_hip.h"
#include "blok_hip_debug.h"
#include "gpu_build.h"
#include <hip/hip_fp16.h>
#include "post_kernels.h"
#include "post_core.h"
#include "reference_world.h"
#include "trace_kernels.h"
#include "dense_kernels.h"
#include "launch_policy.h"
#include "tile_order.h"
#include "path_args.h"
#include "tree.h"

struct blok_hip_ctx {
    int device = 0;
    uint32_t width = 0, height = 0;
    // derived structure in HBM
    uint4* d_nodes = nullptr;
    uint32_t* d_tree_materials = nullptr;
    blok_material* d_materials = nullptr;
    size_t n_materials = 0;
    bool has_world = false;
    bool built_on_device = false;     // structure built by gpu_build.hip (else tree_build.cpp on the host)
    uint32_t world_version = 0;        // counts uploads and rebuilds: scheduling state measured on another world is dropped (TileOrder::key)
    bool tree_owned_by_volume = false; // d_nodes / d_tree_materials belong to the resident volume's scratch (gpu_build.h): never freed here
    bool force_host_build = false;
    blok_world_stats stats{};
    float voxel_size = 1.0f;            // for the next blok_hip_upload_world (blok_hip_set_voxel_size)
    float world_voxel_size = 1.0f;      // of the installed world
    float pending_voxel_size = 1.0f;    // handed from blok_hip_upload_world to install_tree
    // dense-grid path (dense_kernels.hip): the uploaded id grid in 8^3-cell tiles and one occupancy bit per tile; kept when
    // blok_hip_set_dense_dda was on at blok_hip_upload_dense, and then used by the rectangle entries of the primary trace
    bool dense_dda = false, has_dense = false;
    uint32_t* d_dense_tiled = nullptr;
    uint32_t* d_dense_bits = nullptr;
    uint32_t dense_tiles[3] = {0, 0, 0};
    int32_t dense_origin[3] = {0, 0, 0};
    // scratch frame for the host-output entry points
    blok_hit* d_frame = nullptr;
    size_t frame_capacity = 0;
    // progressive accumulation (CudaTracer::m_dAccum / m_frameIndex / m_prevCam, reference cuda_tracer.hpp:51-55)
    float* d_accum = nullptr;
    float* d_color = nullptr;
    size_t accum_pixels = 0;
    uint32_t accum_frames = 0;
    blok_camera prev_cam{};
    bool has_prev_cam = false;
    // image-space chain (post_core.h): history ping-pong [2] and per-frame planes, all width x height
    struct Post {
        size_t pixels = 0;
        uint32_t width = 0, height = 0;      // the frame shape the planes were allocated for
        float *hist_color[2] = {nullptr, nullptr}, *moments[2] = {nullptr, nullptr}, *world_pos[2] = {nullptr, nullptr};
        uint16_t* hist_len[2] = {nullptr, nullptr};
        float* unit_normals[2] = {nullptr, nullptr};   // float4: normalize(binary16 normal)
        uint16_t* motion = nullptr;          // half2
        float *variance = nullptr, *ping = nullptr, *pong = nullptr, *taa_hist[2] = {nullptr, nullptr};
        float* widen = nullptr;              // scratch for state downloads
        int cur = 0, taa_cur = 0;
        bool has_motion = false;
        // blok_hip_draw_frame_rt: the frame's own planes and its camera history
        float *rt_planes[4] = {nullptr, nullptr, nullptr, nullptr}, *rt_denoised = nullptr, *rt_resolved = nullptr;   // [0] colour, [1] world position
        uint16_t *rt_normal_roughness_h = nullptr, *rt_motion_h = nullptr;      // RGBA16F, RG16F
        uint32_t* rt_albedo_metallic_u8 = nullptr;                              // RGBA8
        uint32_t *rt_ldr = nullptr, *rt_final = nullptr;
        uint32_t rt_frame = 0;
        blok_camera rt_prev_cam{};
    } post;
    // device-resident dense store (gpu_build.h: GpuVolume)
    blok::GpuVolume volume;
    bool has_volume = false;
    bool volume_keyed_layout = true;   // blok_hip_set_volume_layout: the next blok_hip_volume_create may use the keyed brick layout (gpu_build.h)
    std::vector<unsigned char> volume_materials;      // the material table the last blok_hip_volume_rebuild installed (compared, not re-uploaded, when unchanged)
    // "last occluder" map of the shadow rays (beam.h: prism_far), rebuilt with every world
    float* d_sun_map = nullptr;
    bool sun_map_enabled = true, has_sun_map = false;
    uint32_t sun_loose_edits = 0; int32_t sun_loose_lo[3] = {0, 0, 0}, sun_loose_hi[3] = {0, 0, 0};      // edits since the map was last made tight, and the union of their boxes (update_sun_map)
    bool sun_tighten_pending = false; uint32_t sun_tighten_u0 = 0, sun_tighten_u1 = 0, sun_tighten_v0 = 0, sun_tighten_v1 = 0;      // texels still to be made tight, a band of rows per edit (update_sun_map)
    hipEvent_t sun_event = nullptr; bool sun_event_pending = false;                                      // behind the latest patch of the map
    uint32_t ray_batching = 3;          // PathArgs::batch_kinds (blok_hip_set_ray_batching); 3 = 2 + the bounce rounds' tail pool
    uint32_t tail_cap = 24, tail_cap_parked = 32;                 // trips after which a bounce round / a round over parked rays stops (BLOK_TAIL_CAPS overrides, experiments)
    bool path_resume = false, path_fine_beam = true;      // PathArgs::resume_secondary / fine_beam (blok_hip_set_path_start)
    blok::SunMapArgs sun{};
    // beam pre-pass (beam.h): start parameters per beam tile, one buffer per stream (launches on one stream are
    // ordered, frames in flight on different streams must not share)
    uint32_t beam_tile = 32;
    uint32_t beam_budget = 0;           // 0 = beam.h's default visit budget
    bool miss_in_walk = true;           // who writes the pixels of tiles the pre-pass found empty (blok_hip_set_miss_writer)
    struct StreamScratch {             // per launch stream
        float* beam = nullptr; size_t n_beam = 0;                       // two-launch form: start parameters per beam tile
        uint32_t* ctl = nullptr; unsigned long long* entries = nullptr; size_t capacity = 0;   // one-launch form: work queue (trace_kernels.h: FrameQueue)
        unsigned long long* slots = nullptr; size_t n_slots = 0; uint32_t serial = 0;   // joint launch: published beam results
        uint32_t* gave_up = nullptr;                                      // joint and list launches: waves that gave up a bounded wait (sticky; blok_hip_frame_queue_stalls)
        // list launches (trace_kernels.h: LiveList): entries, control words, serial, and the pinned words the searches leave the segments' lengths in
        unsigned long long* list_entries = nullptr; size_t list_capacity = 0;   // entries per segment
        unsigned long long* list_ctl = nullptr;
        uint32_t list_serial = 0;
        uint32_t* list_hint = nullptr; bool list_hint_valid = false; uint32_t list_hint_key = 0;
        uint32_t* tile_map = nullptr; size_t n_tile_map = 0;            // sparse exchange, root: frame tile -> record (zero between launches)
        void* tail_pool = nullptr; size_t tail_pool_bytes = 0;           // path launches: the bounce rounds' tail pool (path_core.h: TailRecord[blocks][kTailCapacity]), grown on demand
    };
    std::unordered_map<hipStream_t, StreamScratch> beam_buffers;
    // Longest-first scheduling of the walk for a camera at rest (tile_order.h; rectangle launches of the static forms): every walk wave
    // leaves the clocks it spent in d_cost (one buffer per context, for the launch geometry in `key`).  Every `interval` launches a
    // radix sort of a snapshot of those costs follows the frame on its stream: it writes the order buffer that is NOT in use, and a later
    // launch adopts it once hipEventQuery says it is complete — no launch ever waits.  Decisions: launch_policy.h plan_order.
    struct TileOrder {
        bool enabled = true;
        bool rank_tiles = true;                            // ... also for a rank's tile launches (blok_hip_set_rank_tile_ordering)
        uint32_t interval = 8, interval_now = 8;          // interval_now: of a view that has no order of its own yet (a slot carries its own)
        uint32_t *d_cost = nullptr, *d_iota = nullptr;
        uint32_t *d_keys_in = nullptr, *d_keys = nullptr;   // the sort's snapshot of the costs, and its sorted keys
        void* d_temp = nullptr;
        size_t temp_bytes = 0, capacity = 0;
        // Round 4: a small CACHE of orders per launch geometry, one slot per view (round 3 kept two buffers — the order in use and the one
        // being sorted — so a caller alternating between two fixed views, stereo eyes or a camera cycle, walked in row-major order for ever).
        // A slot holds an order sorted from the clocks of the view `cam` (own-view: used by launches from that view only) or from dilated
        // clocks (made to be carried to the next view by a shift: only the latest such order is of any use).
        static constexpr int kSlots = 4;
        struct Slot {
            uint32_t *d_order = nullptr, *d_rank_of = nullptr;
            bool valid = false, dilated = false;
            uint32_t live = 0, radius = 0;                 // leading entries that walked (as read at adoption); dilation it was sorted with
            blok_camera cam{};                             // the view its clocks were measured under
            float inv_depth[2] = {};                       // mean and sigma of that frame's live beam tiles' inverse start parameters (dilated orders)
            uint32_t frames_since_sort = 0, interval_now = 8;      // launches that walked in it; re-sort interval of its view (doubles while the view rests)
            uint64_t last_use = 0;                         // serial of the latest launch that walked in it
        } slots[kSlots];
        uint32_t* h_live = nullptr;                         // pinned, kSlots words: how many leading entries of the slot's order walked (written by the device)
        float* h_depth = nullptr;                           // pinned, kSlots x 64 x 3 floats: partial count, sum, sum of squares of the live beam tiles' inverse start parameters
        uint32_t key[7] = {};                               // launch rectangle, frame size, world version
        int current = -1, target = 0;                       // the slot adopted last (-1: none yet); the slot a pending sort writes
        bool pending = false;
        bool orphan = false;                                // a sort of a previous launch geometry may still be running (the next sort waits for it)
        uint32_t still_frames = 0, prefix_limit = 0;
        uint64_t launch_serial = 0, hold_serial = 0;        // orderable launches so far; ... at the latest hold_markers (what the held markers are behind)
        blok_camera last_cam{};                             // camera of the last launch
        blok_camera recent[4] = {}; uint32_t n_recent = 0, revisit_streak = 0;  // cameras of the latest launches; consecutive launches from a view seen among them (alternating views are views at rest)
        blok_camera pending_cam{}; bool pending_dilated = false; uint32_t pending_radius = 0, pending_interval = 8;      // of the sort in flight
        hipEvent_t done = nullptr;
        // a camera in motion (launch_policy.h): orders sorted from dilated clocks, carried to the next view by a whole-tile shift
        bool moving = true;                                 // blok_hip_set_moving_order
        void* d_class_scratch = nullptr;                    // tile_order.h: count table and class bytes of the counting sort
        bool have_residual = false; float last_residual = 0.0f;     // what the latest shift left over (sizes the next dilation)
        bool debug_shift = false; uint32_t debug_sx = 0, debug_sy = 0;      // blok_hip_debug_force_order_shift: every ordered launch uses this shift (tests)
        uint32_t* d_fallback = nullptr;                     // wave tiles the search waves of the latest prefix launch walked themselves (cleared in stream order before it)
        uint32_t* h_fallback = nullptr;                     // pinned: ... copied here behind the launch
        uint32_t last_fallback = 0;                         // ... as read before the next launch (blok_hip_last_fallback_tiles)
        bool alone_before = false;                          // the previous orderable launch had the device to itself
        int last_use = 0; uint32_t last_sx = 0, last_sy = 0;        // the latest launch: 0 natural order, 1 an order of its own view, 2 a carried one (blok_hip_last_order_use)
        int chosen = -1;                                    // the slot the latest launch walked in (-1: none)
    } order;
    // list launches, rectangle frames: clocks per wave tile of the last frame of this launch geometry, and the camera they were measured under (trace_kernels.h: cost classes)
    uint32_t* d_list_cost = nullptr; size_t list_cost_capacity = 0; uint32_t list_cost_key[6] = {};
    blok_camera list_prev_cam{}; bool list_has_prev_cam = false;
    bool list_classes = true;           // blok_hip_set_list_classes
    uint32_t* debug_clocks = nullptr;   // blok_hip_set_debug_wave_clocks (caller's device memory)
    int launch_form = blok::kFormAuto;  // blok_hip_set_fused (launch_policy.h: LaunchForm)
    int last_launch_kind = -1;          // launch_policy.h: LaunchKind of the latest rectangle / tile launch (blok_hip_last_launch_kind)
    uint32_t frame_parts = 32, frame_chunk = 1;      // FrameQueue::n_parts / chunk (BLOK_FRAME_PARTS / BLOK_FRAME_CHUNK override, for experiments)
    int cu_count = 0;
    int frame_blocks_per_cu[2][16] = {};   // [mode][levels], 0 = not asked yet
    // TAA jitter of the next frames' primary rays, in pixels (blok_hip_set_taa_jitter); blok_hip_draw_frame_rt sets it per frame
    float jitter_px[2] = {0.0f, 0.0f};
    bool rt_taa_jitter = true;          // PostProcess::Settings::enableTAA (renderer_postprocess.hpp:103)
    // timing
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool timing = false, timed = false;
    std::string error;
};

namespace blok_api {

int set_error(blok_hip_ctx* ctx, int status, const std::string& msg);

#define BLOK_HIP_TRY(ctx, call)                                                                    \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_error(ctx, e_ == hipErrorOutOfMemory ? BLOK_ERR_OOM : BLOK_ERR_HIP,         \
                             std::string(#call) + ": " + hipGetErrorString(e_));                   \
    } while (0)

void free_post(blok_hip_ctx* ctx);
void free_world(blok_hip_ctx* ctx);
int install_materials(blok_hip_ctx* ctx, const blok_material* materials, size_t n_materials);
int install_tree(blok_hip_ctx* ctx, const blok::HostTree& tree, const blok_material* materials, size_t n_materials);
int rebuild_sun_map(blok_hip_ctx* ctx);
int update_sun_map(blok_hip_ctx* ctx, const int32_t lo[3], const int32_t hi[3], bool same_lattice, bool may_add = true);
int ensure_frame(blok_hip_ctx* ctx, size_t records);
blok::TraceArgs base_args(const blok_hip_ctx* ctx, const blok_camera* cam);
int prepare_beam(blok_hip_ctx* ctx, blok::RayMode mode, blok::TraceArgs& args, hipStream_t stream, uint32_t tiles_of_rank, uint32_t* n_beams);
int prepare_queue(blok_hip_ctx* ctx, blok::RayMode mode, const blok::TraceArgs& args, hipStream_t stream, uint32_t n_beams, blok::FrameQueue* queue, uint32_t* n_blocks);
int check_trace(blok_hip_ctx* ctx, const blok_camera* cam);
// api_launch.hip
int beam_buffer(blok_hip_ctx* ctx, hipStream_t stream, size_t n, float** out);
void free_order(blok_hip_ctx* ctx);
int order_buffers(blok_hip_ctx* ctx, uint32_t blocks, hipStream_t stream);
int live_list(blok_hip_ctx* ctx, blok::TraceArgs& args, hipStream_t stream, uint32_t n_searches, uint32_t per_search);
int launch_timed(blok_hip_ctx* ctx, blok::RayMode mode, blok::TraceArgs args, uint32_t blocks, hipStream_t stream, uint32_t tiles_of_rank = 0,
                 const blok::TileFrames* frames = nullptr);
void forget_device_activity(const blok_hip_ctx* ctx, bool one_stream = false, hipStream_t stream = nullptr);
bool rect_inside(const blok_hip_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h);
// The root's assembly of a sparse exchange (blok_hip_scatter_*_tile_frames_device): the ranks' buffers either side by side in
// `gathered_dev` or, rank_ptrs_dev != null, wherever a device array of n_ranks pointers says (peer-mapped memory of other devices).
int scatter_frames(blok_hip_ctx* ctx, bool codes, const void* gathered_dev, const void* const* rank_ptrs_dev, uint32_t n_ranks, size_t rank_stride_words,
                   uint32_t tile, uint32_t max_records, uint32_t n_frames, void* out_frames_rgba_dev, void* tile_state_dev, void* hip_stream);

}  // namespace blok_api
#endif
