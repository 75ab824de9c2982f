// gfx950 trace kernels: one lane = one ray, walking the 64-tree (tree.h).
//
// SEMANTICS (what must equal the reference bit for bit).  For a ray (o, d, tmin, tmax), with
//   inv_a = 1 / (|d_a| < 1e-6 ? 1e-6 : d_a)                        intersect.rint:79
//   T(a, p) = fl( fl(p - o_a) * inv_a )      for an integer plane p  intersect.rint:48-49,179-180
// a filled unit voxel v is REPORTED by the reference iff
//   max(entry(v), tmin) < min(exit(v), tmax),   entry = max_a min(T(a,v_a), T(a,v_a+1)),
//                                               exit  = min_a max(T(a,v_a), T(a,v_a+1))
// (intersect.rint:179-193 at the leaf level; the same test on every ancestor box is implied, because
// box planes are integers — exactly representable — and fl(p - o), fl(x * inv) are monotone in p, so
// ancestor intervals contain the leaf's), and is reported with t = max(entry(v), tmin)
// (intersect.rint:189,141).  The ray's result is the reported voxel of smallest t
// (traceRayEXT closest-hit), which is unique.  The face is
// getHitFace(o + d*t, centre) (intersect.rint:58-68,138), the material the leaf's materialId.
//
// TRAVERSAL.  Sort all integer planes the ray crosses by T (a 3-way merge, one sorted list per
// axis).  Every reported voxel is a cell of that merge sequence (its near planes all have T strictly
// below its far planes), and along the sequence the entry T never decreases.  The kernel walks the
// sequence hierarchically: at level l it steps whole 4^l cells whose mask bit is clear, descends
// into set ones, and picks the start cell of a node by counting which of its three interior planes
// per axis have T <= current t.  All T's are evaluated with the formula above on integer planes
// (never accumulated), so the walk visits exactly the reference's reported voxels in t order and
// stops at the first.
#ifndef BLOK_TRACE_KERNELS_H
#define BLOK_TRACE_KERNELS_H

#ifdef BLOK_TRACE_HOST_HARNESS
#include "host_harness_shims.h"   // tests/host_harness: CPU stand-ins for the HIP types/intrinsics used
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "blok_hip.h"
#include "tree.h"

namespace blok {

#ifndef BLOK_BLOCK_THREADS
#define BLOK_BLOCK_THREADS 64
#endif
constexpr int kBlock = BLOK_BLOCK_THREADS;   // 64: one wave = one 8x8 pixel tile (measured 5 % faster than 4-wave blocks)
#ifndef BLOK_WAVE_W
#define BLOK_WAVE_W 8
#endif
constexpr uint32_t kWaveW = BLOK_WAVE_W, kWaveH = 64u / kWaveW;      // pixel footprint of one wave
constexpr uint32_t kTileW = kBlock >= 128 ? 2u * kWaveW : kWaveW;   // pixels per block, x
constexpr uint32_t kTileH = kBlock >= 256 ? 2u * kWaveH : kWaveH;   // pixels per block, y
static_assert(kBlock == 64 || kBlock == 128 || kBlock == 256, "block = 1, 2 or 4 waves of 8x8 pixels");

// Block -> tile mapping of the rectangle kernels.  Workgroups are dealt round-robin to the 8 XCDs (blocks b and
// b + 8 share an XCD and its L2); BLOK_XCD_MAP chooses what that means on screen:
//   0  row-major tiles (neighbouring tiles on different XCDs)
//   1  each XCD gets one contiguous eighth of the row-major tile sequence
//   2  as 1, and inside an XCD's share tiles are visited supertile by supertile (16x16 tiles, Morton order)
#ifndef BLOK_XCD_MAP
#define BLOK_XCD_MAP 0
#endif
constexpr uint32_t kSuper = 16u;          // tiles per supertile edge (mapping 2)

// Number of workgroups to launch for a w x h rectangle under the mapping.
inline uint32_t rect_grid_blocks(uint32_t w, uint32_t h) {
    const uint32_t bx = (w + kTileW - 1u) / kTileW, by = (h + kTileH - 1u) / kTileH;
#if BLOK_XCD_MAP == 0
    return bx * by;
#elif BLOK_XCD_MAP == 1
    return (bx * by + 7u) / 8u * 8u;
#else
    const uint32_t sx = (bx + kSuper - 1u) / kSuper, sy = (by + kSuper - 1u) / kSuper;
    return (sx * sy * kSuper * kSuper + 7u) / 8u * 8u;
#endif
}

enum class RayMode : int { Rect = 0, Tiles = 1, Rays = 2 };

constexpr float kBeamNone = 3.0e38f;   // beam pre-pass result: "no cell of the tree meets the tile's frustum" (beam.h)

// Live list of a frame (list launches: list_joint_kernel / list_walk_kernel, trace_kernels.hip).  The searches of THIS frame say which
// wave tiles need a walk at all: a search that finds its beam tile live appends one 64-bit entry per wave tile of it — serial | task |
// start parameter — to a list of its segment (beam tile b -> segment b mod 8; one 64-bit add per list reserves the slots, one more
// counts the search as done), and the walk waves of that segment take the entries.  The walk therefore has no wave for a dead tile, needs
// nothing from an earlier frame to be complete, and in the joint form starts when the first searches end.
//
// COST CLASSES.  A launch lasts as long as its last wave, and a wave of grazing rays takes 100+ us of a 200 us launch whatever its
// priority (a chain of dependent instructions): such waves must start first.  What a wave tile will cost is best predicted by what it
// cost in the previous frame — walk waves leave their clock count per wave tile — looked up where the tile's point at its start
// parameter was on the previous frame's screen (a camera in motion), and used in four coarse classes (a full sort by a stale cost is
// worse than none: profiles/r03_stale_cost_order_experiment.txt).  Each segment has one list per class; walk workgroups are dealt to the
// classes heaviest first, as many per class as the previous launch's lists were long plus a margin (a hint: walk wave k of a class
// takes entries k, k + n, ... of its list, so any numbers walk every entry).  Only the ORDER depends on the past; no pixel does.
//
// Segments follow the round-robin dispatch of workgroups to the 8 XCDs (workgroup i -> XCD i mod 8), so a walk wave only ever waits for
// searches that were dispatched to its own XCD before it; nothing depends on that for exactness (a wave that gives up a bounded wait
// leaves its entries to list_cleanup_kernel).
constexpr uint32_t kListSegments = 8, kListClasses = 4;
// 64-bit control words per segment, two 128-byte lines: what the searches add to — searches done, entries reserved per class — on one,
// what the walk waves poll on the other — per class final (serial << 32 | length), and the serial of the last launch in which a walk wave gave up
constexpr uint32_t kListCtlWords = 32;
constexpr uint32_t kListDone = 0, kListTally = 1, kListFinal = 16, kListGaveUp = 20;
constexpr uint32_t kListTaskBits = 21, kListT0Bits = 23, kListSerialBits = 20;
constexpr uint32_t kListUnknownClass = 1;         // a wave tile nobody has a cost for
// clocks (s_memtime) a walk wave took in the previous frame -> class 1, 2, 3: the median wave, 2.25 and 5 times as much
constexpr uint32_t kListCostClass1 = 36000u, kListCostClass2 = 81000u, kListCostClass3 = 182000u;
struct LiveList {
    unsigned long long* entries;           // [kListSegments][kListClasses][seg_capacity]; null = not a list launch
    unsigned long long* ctl;               // kListSegments x kListCtlWords
    uint32_t* hint;                        // pinned host memory, [kListSegments][kListClasses]: the lists' lengths of the last launch (size the next walk grid); may be null
    uint32_t* cost;                        // per wave tile (task): clocks its walk wave took last time (0 = unknown); null = no classes (everything kListUnknownClass)
    uint32_t seg_capacity;                 // entries per list (>= the segment's beam tiles x wave tiles per beam tile)
    uint32_t serial;                       // of this launch, 1 .. 2^20 - 1: an entry / a final word is valid when it carries it
    uint32_t walkers[kListClasses];        // walk workgroups per segment and class
    uint32_t n_searches;                   // search workgroups of this launch (segment x has those with index = x mod 8)
    uint32_t has_prev;                     // prev_cam is the camera the costs were measured under (else: looked up at the same screen position)
    blok_camera prev_cam;
};

struct TraceArgs {
    const uint4*    nodes;
    const uint32_t* materials;
    int32_t  origin[3];                    // corner of the tree in VOXEL units (world coordinate = voxel coordinate * voxel_size)
    uint32_t levels;
    float voxel_size, inv_voxel_size;      // ChunkManager's voxelSize (reference chunk_manager.cpp:19-25): a power of two, so that
                                           // every plane (integer * voxel_size) stays exactly representable; 1 for the reference app
    blok_camera cam;
    // TAA jitter in clip space, (2 jx / frame_w, 2 jy / frame_h) (getJitterClipSpace, reference blok/src/renderer_postprocess.cpp:234-241):
    // what getJitteredProjection adds to proj[2][0..1] (:264-265), i.e. to the NDC coordinate every primary ray is formed from
    float jitter_clip[2];
    uint32_t frame_w, frame_h;
    uint32_t x0, y0, w, h;                 // Rect
    uint32_t tile, rank, n_ranks, tiles_x, tiles_total;   // Tiles
    const blok_ray* rays;                  // Rays
    uint32_t n_rays;
    blok_hit* out;                         // may be null
    uint32_t* out_rgba;                    // may be null: RGBA8 through the material table
    const blok_material* mat_table;
    uint32_t n_materials;
    float tmin, tmax;
    // beam pre-pass (beam.h): one conservative start parameter per beam_tile x beam_tile pixels, written by
    // beam_kernel and read by the Rect / Tiles trace kernels of the same stream; null = no pre-pass
    float* beam;
    uint32_t beam_tile, beam_bx;           // beam_bx: beam tiles per row of the rectangle (Rect)
    const uint32_t* order;                 // Rect / Tiles: workgroup b (of a frame) walks wave tile order[b] (null = b): longest-first scheduling
    uint32_t* cost_out;                    // Rect / Tiles: per wave tile, the clocks its wave spent (null = not recorded)
    // joint launch (joint_kernel): the pre-pass waves and the walk waves are ONE grid; a beam tile's result is published as
    // (serial << 32 | start parameter bits) and a walk wave waits for its tile's word to carry this launch's serial
    unsigned long long* beam_slots;        // null = the two-launch form (TraceArgs::beam holds plain floats)
    uint32_t beam_serial;
    // launch over a PREFIX of the order: walk waves are dispatched only for the first `launched` tiles of `order` (the tiles that walked when
    // the order was made); rank_of[tile] >= launched = no walk wave exists for that tile, and the search wave of a live beam tile walks
    // such tiles itself (a view that has changed; exact either way).  null = every tile has its walk wave.
    const uint32_t* rank_of;
    uint32_t launched;
    // an order made for ANOTHER view, carried over by a whole-tile shift (a camera in motion, api.hip): entry (tx, ty) of the order names
    // tile ((tx + order_sx) mod tiles_x, (ty + order_sy) mod tiles_y) of this launch — still a permutation.  0, 0 = none.
    uint32_t order_sx, order_sy;
    // ... and the part of the screen the shift brings in from outside the old view has no entry of its own (the entries that land there come
    // from the opposite edge): columns [strip_x0, strip_x0 + strip_nx) and rows [strip_y0, strip_y0 + strip_ny) of the tile grid.  A prefix
    // launch gives EVERY tile of these strips a walk workgroup — the first n_strip of the walk's workgroups, in strip order — and the
    // order's entries that land there stand down; left to the search waves (one tile after the other) a pan that brings the world into the
    // frame took 0.5-0.9 ms per frame instead of 0.27 in row-major order.
    uint32_t strip_x0, strip_nx, strip_y0, strip_ny, n_strip;
    uint32_t* fallback_tiles;              // prefix launches: how many wave tiles the search waves walked themselves is added here (null = not counted); device memory
    uint32_t* joint_gave_up;               // waves that gave up a bounded wait: joint form, for their tile's search (they start at the ray origin instead); list forms, for an entry (walked by the clean-up).  0 in a working system
    uint32_t miss_in_walk;                 // two-launch form: 1 = the walk's waves write the miss pixels of tiles the pre-pass found empty (they are launched anyway), 0 = the pre-pass does
    uint32_t beam_budget;                  // node visits a search may spend (0 = kBeamMaxVisits); running out is answered conservatively
    // diagnostics (null in production; blok_hip_beam_prepass / blok_hip_set_debug_wave_clocks): node visits of every beam search, clocks
    // every listed walk wave spent (indexed by wave tile)
    uint32_t* debug_visits;
    uint32_t* debug_clocks;
    LiveList list;                         // list launches (entries != null): the searches publish the live wave tiles, the walk takes them from the list
};

// Work queues of the one-launch frame (frame_kernel, trace_kernels.hip), per stream: n_parts independent parts, each with its
// own counters and its own slice of 64-bit task entries.
constexpr uint32_t kFrameParts = 256;        // most parts a launch may be cut into (FrameQueue::n_parts <= this)
constexpr uint32_t kFramePartWords = 192;    // per part: five counters and a sticky error word, one 128-byte line each
constexpr uint32_t kFrameCtlWords = kFrameParts * kFramePartWords;
constexpr uint32_t kFrameStalledWord = 160;  // within a part: waves that gave up waiting for a queue entry (0 in a working system); host-readable, never reset by the kernel
constexpr uint32_t kFramePollBudget = 1u << 19;   // polls (~1 us apart) before a waiting wave gives up
struct FrameQueue {
    uint32_t* ctl;                  // kFrameCtlWords words, zero between launches (but for the sticky error words)
    unsigned long long* entries;    // n_parts slices of part_capacity entries: task id | start parameter bits << 32; all ones = empty
    uint32_t n_beam;                // beam tasks of this launch; part p owns tasks p, p + n_parts, ...
    uint32_t part_capacity;         // entries per part (>= its beam tasks * sub-tiles per beam tile)
    uint32_t n_parts;               // parts of this launch (workgroup b works in part b mod n_parts)
    uint32_t chunk;                 // consecutive entries per trace ticket
};

// Several frames of one rank's tile share in ONE beam + trace launch pair (Tiles mode; multi-GPU).  At N ranks a frame's share is
// 1/N of a launch whose duration is mostly latency (the pre-pass is one round of waves; the walk ends in a tail of grazing-ray
// waves), so a rank traces kMaxTileFrames consecutive frames — each with its own camera — as one launch of the single-GPU size.
constexpr uint32_t kMaxTileFrames = 8;
struct TileFrames {
    blok_camera cam[kMaxTileFrames];
    uint32_t n_frames;
    uint32_t blocks_per_frame;             // trace workgroups of one frame
    uint32_t beams_per_frame;              // beam tiles (and floats of TraceArgs::beam) of one frame
    size_t   frame_stride;                 // records / pixels between consecutive frames in out and out_rgba
};

struct UntileArgs {
    const void* gathered;
    void* frame;
    uint32_t frame_w, frame_h, tile, n_ranks, tiles_per_rank_max, tiles_x, elem_bytes;
    size_t gathered_frame_stride;          // elements between consecutive frames (grid y) inside the gathered buffer; output frames are contiguous
};

// Sparse exchange buffers, for n_frames frames at once: n_frames count words, then the records INTERLEAVED by frame — slot j of
// frame f at word n_frames + (j * n_frames + f) * (1 + tile^2) — so that the first S slots of every frame are ONE contiguous
// prefix (what travels); one frame: the count word, then its records.
struct CompactArgs {                       // rank's dense RGBA8 tiles -> {counts, records {local tile index, tile^2 pixels}}
    const uint32_t* tiles;
    uint32_t* out;                         // count words zero before the launch
    uint32_t tile, n_tiles, n_frames;
    size_t tiles_frame_stride;             // words between consecutive frames (grid y) of the dense tiles
};
// The same exchange with 16 bits per pixel instead of 32: a shaded pixel is a function of (material id, face) — shade_rgba,
// trace_core.h — or the sky, and every rank holds the material table, so a pixel travels as
//   code = min(material id, n_materials) * 8 + face,  0xFFFF = sky        (needs (n_materials + 1) * 8 <= 0xFFFF)
// made from the rank's first-hit records and expanded by the root through the very function the trace kernels shade with: the
// assembled frame is bit-identical, the bytes on the wire are half (kCodeSky / pixel_code / code_rgba in trace_kernels.hip).
struct CompactHitArgs {                    // rank's dense first-hit tiles -> {counts, records {local tile index, tile^2 16-bit codes}}
    const blok_hit* hits;
    uint32_t* out;
    uint32_t tile, n_tiles, n_frames, n_materials;
    size_t hits_frame_stride;              // records between consecutive frames (grid y)
};
struct ScatterArgs {                       // gathered compact buffers of all ranks -> row-major frames
    const uint32_t* gathered;
    const uint32_t* const* rank_ptrs;      // optional (device array of n_ranks pointers): rank r's buffer is rank_ptrs[r] instead of gathered + r * rank_stride —
                                           // buffers that live on OTHER devices, read over xGMI through peer mappings (single-process entry, api_multi.hip)
    uint32_t* frame;                       // n_frames contiguous frames
    uint32_t* tile_map;                    // n_frames * tiles_total words, zero before and after: 1 + (rank * max_records + slot) of the record that holds a tile
    uint8_t* tile_state;                   // optional, n_frames * tiles_total: does the frame buffer's tile hold anything but sky?  Lets sky tiles that stay sky go unwritten
    uint32_t frame_w, frame_h, tile, n_ranks, tiles_x, tiles_total, max_records, n_frames;
    size_t rank_stride;                    // words between the ranks' buffers
    uint32_t record_words;                 // 1 + tile^2 (RGBA8 pixels) or 1 + tile^2 / 2 (16-bit codes)
    const blok_material* mat_table;        // codes only
    uint32_t n_materials;
};
void launch_compact_hit_tiles(const CompactHitArgs& args, hipStream_t stream);
void launch_compact_tiles(const CompactArgs& args, hipStream_t stream);
void launch_scatter_tiles(const ScatterArgs& args, bool codes, hipStream_t stream);      // map kernel, then one wave per frame tile
uint32_t sky_rgba();                       // the RGBA8 a miss is shaded with (trace_core.h: kSkyRgba)

struct SunMapArgs {                        // beam.h: prism_far, one wave per texel
    TraceArgs trace;
    float u[3], v[3], s[3];
    float u0, v0, texel;
    uint32_t nu, nv;
    float* map;
    uint32_t iu0, iv0, sub_nu, sub_nv;     // the texels this launch computes: [iu0, iu0 + sub_nu) x [iv0, iv0 + sub_nv) (the whole map, or what an edit can have changed)
};
void launch_sun_map(const SunMapArgs& args, hipStream_t stream);
void launch_sun_map_raise(const SunMapArgs& args, float far_depth, hipStream_t stream);      // texels of the sub-rectangle: max(value, far_depth)

struct PathArgs;
struct TonemapArgs;
struct AccumArgs;
void launch_accumulate(const AccumArgs& args, hipStream_t stream);
void launch_tonemap(const TonemapArgs& args, hipStream_t stream);
void launch_paths(const PathArgs& args, uint32_t n_blocks, hipStream_t stream);
void launch_trace(RayMode mode, const TraceArgs& args, uint32_t n_blocks, hipStream_t stream);
// Pre-pass and walk in one grid, statically: workgroups [0, n_beam_tiles) search, the others walk (Rect and Tiles).
void launch_joint(RayMode mode, const TraceArgs& args, uint32_t n_beam_tiles, uint32_t n_blocks, hipStream_t stream);
// Polls (>= 0.5 us apart: s_sleep 16 + the load) before a walk wave stops waiting for its tile's search and starts at the ray origin instead.  A search ends
// within ~80 us of its launch's start, so a longer wait means the search has not been DISPATCHED yet — possible with several joint
// launches in flight: workgroup i goes to XCD i mod 8 and each XCD works through its share on its own, so walk waves of launch A can
// fill an XCD on which launch B's searches are still pending while B's walk waves do the same to A elsewhere.  Such a circle ends
// when the waiting waves give up, so the budget is what it may cost: >= 0.25 ms (three times the longest search), not the seconds a
// "never happens" budget would.  The automatic form never has two joint launches in flight (api.hip).
constexpr uint32_t kJointPollBudget = 512u;
// List launches (TraceArgs::list): joint = searches and list-fed walk waves in ONE grid (n_walkers a multiple of kListSegments), followed
// by the clean-up of entries whose walk wave gave up waiting (none in a working system); otherwise launch_beam (which fills the list),
// then launch_list_walk.  frames: several frames of a rank's tiles in one launch (Tiles), or null.
void launch_list_joint(RayMode mode, const TraceArgs& args, const TileFrames* frames, uint32_t n_beam_tiles, uint32_t n_walkers, hipStream_t stream);
void launch_list_walk(RayMode mode, const TraceArgs& args, const TileFrames* frames, uint32_t n_walkers, hipStream_t stream);
// Number of beam tiles of a launch (= floats of TraceArgs::beam) and the pre-pass itself; Rect and Tiles only.
uint32_t beam_tiles(RayMode mode, const TraceArgs& args, uint32_t tiles_of_rank);
void launch_beam(RayMode mode, const TraceArgs& args, uint32_t n_beam_tiles, hipStream_t stream);
void launch_untile(const UntileArgs& args, uint32_t n_frames, hipStream_t stream);
// Beam pre-pass (if args.beam) and walk of frames.n_frames frames of the rank's tiles, one launch each (Tiles mode).
void launch_tile_frames(const TraceArgs& args, const TileFrames& frames, hipStream_t stream, uint32_t walk_blocks_per_frame = 0);
void launch_beam_frames(const TraceArgs& args, const TileFrames& frames, hipStream_t stream);      // the pre-pass alone (list launches: it fills the list)
// One-launch frame: pre-pass and walk in one persistent grid of n_blocks waves (Rect and Tiles).
void launch_frame(RayMode mode, const TraceArgs& args, const FrameQueue& queue, uint32_t n_blocks, hipStream_t stream);
int frame_blocks_per_cu(RayMode mode, const TraceArgs& args);      // resident workgroups per CU for the launch's LDS size (0 on error)

}  // namespace blok
#endif
