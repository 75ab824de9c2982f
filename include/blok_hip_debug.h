/* blok_hip_debug.h — diagnostics, test hooks and tuning knobs of libblok_hip.so.
 *
 * NOT part of the drop-in surface: a reference-side binding includes blok_hip.h only, whose entries are the ones
 * SURVEY.md section 8(b) derives from CudaTracer (blok/include/cuda_tracer.hpp:23-58) and Renderer (blok/include/renderer.hpp:29-77).
 * What is here has no reference counterpart: switches between launch forms and schedules that were measured against each other
 * (every setting gives the same frames), read-backs the tests and the experiment scripts use, and hooks that force
 * paths a working system rarely takes.  The library exports them all the same; nothing here is needed to render.
 */
#ifndef BLOK_HIP_DEBUG_H
#define BLOK_HIP_DEBUG_H

#include "blok_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------ structure build */
/* The traversal structure is built on the device (blok_amd/csrc/hip/gpu_build.hip).  Worlds outside what those
 * kernels cover (empty worlds, sub-chunks smaller than 4 voxels or of mixed sizes) are built by the general host
 * path instead; blok_hip_set_host_build(ctx, 1) forces that path (used by tests to compare the two). */
int blok_hip_set_host_build(blok_hip_ctx* ctx, int enabled);
int blok_hip_world_built_on_device(const blok_hip_ctx* ctx);   /* 1 / 0 */
/* Copies the device-resident structure back: n_tree_nodes 16-byte nodes and n_voxels material ids
 * (see blok_world_stats); either output may be NULL.  For tests and debugging. */
int blok_hip_download_tree(const blok_hip_ctx* ctx, void* nodes_out, size_t node_capacity,
                           uint32_t* materials_out, size_t material_capacity);

/* ------------------------------------------------------------ resident volume */
/* Diagnostic: 0 = the next blok_hip_volume_create keeps its brick masks in row-major order and rebuilds by scanning, keying and sorting
 * all bricks (the general path, ~2.4 ms for a 1024^3 box); 1 (default) = bricks indexed by their tree key under a pyramid of occupancy
 * words, rebuilt by scans over the pyramid (~0.2 ms), whenever the box's 64^(levels-1) mask words are affordable.  Both give the same
 * node and material arrays, byte for byte (tests/test_brush.py). */
int blok_hip_set_volume_layout(blok_hip_ctx* ctx, int keyed);

/* ------------------------------------------------------------ path kernel */
/* Scheduling knob of the path kernel (no reference counterpart): 0 = every lane walks whatever ray it has pending; 1 = a wave walks
 * one kind of ray at a time (primary, else shadow, else bounce); 2 = one kind at a time and the oldest sample first, so the
 * pixels of a wave stay in step sample by sample.  Per-lane work and results are identical in these three (tests/test_paths.py).
 * 3 (default, round 4) = 2 and the bounce rounds' tail pool (from 4 samples per pixel on, two bounces or more, rays from the root): a round
 * of last-segment rays is cut off after 24 trips of the walk, the rays still walking are parked and walked 64 at a time in rounds of
 * their own, and their terms reach the pixels' sums later — the same terms in another order (colour inside a hundredth of the test
 * tolerance of mode 2's, G-buffer bit-identical, launches deterministic; 64 spp at 4K 35 -> 30 ms). */
int blok_hip_set_ray_batching(blok_hip_ctx* ctx, int enabled);
/* Where the path kernel's walks start (round 4; no reference counterpart: raygen.rgen:217-229, 282-298 hand every ray to traceRayEXT).
 * wave_tile_beam (default 1): every 8x8-pixel wave tile searches a start parameter of its own once per launch, on top of its 32x32 beam
 * tile's, from 8 samples per pixel on: all spp primary rays of its pixels start behind it (4K over 1024^3, 64 spp: 46.7 -> 44.5 ms; 8 spp: 6.06 -> 5.90;
 * 2 spp would lose: 1.68 -> 1.82).
 * resume_from_anchor (default 0 — built, exact, and slower): a pixel's rays enter the walk from the ancestors of its latest reported
 * voxel — the shadow and bounce rays start a hair off it (raygen.rgen:284, :376), the next sample's primary ray ends near it — instead
 * of descending from the root: the start voxel is found with the walk's own plane rule and verified, the lowest common ancestor comes
 * from an LDS copy of the hit's ancestor stack, and ancestors that hold nothing the ray can still reach are left at once
 * (blok_amd/csrc/hip/trace_core.h: walk_resume).  It takes 13 % off the secondary rays' iterations and gives nothing back: the descents it
 * saves are the ones every lane of the wave takes together, the set-up it adds runs per ray (profiles/r04_paths_start_ab.txt).  Kept as a
 * second instantiation of the kernel, so the default carries none of its state.
 * Frames are identical bit for bit in all four combinations (tests/test_paths.py). */
int blok_hip_set_path_start(blok_hip_ctx* ctx, int resume_from_anchor, int wave_tile_beam);
/* "Last occluder" map of the path kernel's shadow rays (no reference counterpart): all shadow rays share the shader's constant sun
 * direction (raygen.rgen:142,185), so with every world the backend records, per 4-voxel texel of the plane perpendicular to
 * it, how far along that direction voxels exist at all; a shadow ray's tmax is capped there (rays above the last occluder skip
 * their walk).  Results are identical with and without it (tests/test_paths.py).  Default on. */
int blok_hip_set_sun_map(blok_hip_ctx* ctx, int enabled);

/* ------------------------------------------------------------ frame launches: pre-pass, order, launch form */
/* Beam pre-pass of the frame kernels (no reference counterpart; the reference culls per ray in Vulkan RT hardware,
 * blok/src/renderer_raytracing.cpp:15-254): before a rectangle / tile launch, one wave per beam_tile_pixels^2 pixels finds a
 * conservative start parameter for that tile's rays, and tiles whose frustum meets no voxel are written as misses without
 * a walk.  Results are identical with and without it (tests/test_gpu_parity.py).  0 turns it off; default 32. */
int blok_hip_set_beam(blok_hip_ctx* ctx, uint32_t beam_tile_pixels);
/* Longest-first scheduling of the walk for a camera at rest (default: on, re-sorted every 8 launches; rectangle launches of at least
 * 4096 wave tiles behind the pre-pass, static launch forms 0, 2, 3): every wave of the walk leaves the clocks it spent; every N-th launch a
 * 16-bit radix sort of a snapshot of those costs (129 600 keys at 4K) follows the frame on its stream, a later launch adopts the finished
 * order, and the walk's workgroups take their tiles in it, so the long grazing-ray waves — each a quarter of the launch long, whatever
 * their priority — start first instead of forming the launch's tail.  The order also tells which tiles need a walk wave at all: launch
 * forms 2 and 3 dispatch walk waves for the tiles that walked when the order was made only, and a tile that has become live since is
 * walked by its search wave.  An order of a view's own clocks is used for that view only (the same camera up to float noise: 0.0006 degree,
 * a thousandth of a voxel): ordering by a stale cost is no better than row-major even one frame later, and the prefix of a view that has crept
 * away leaves the new silhouettes to the search waves (profiles/r03_stale_cost_order_experiment.txt, r03_moving_order_solitary_frames.txt);
 * any camera in motion, however slow, gets an order of another kind (blok_hip_set_moving_order below).
 * A view at rest is re-sorted ever less often (the interval doubles up to 64 launches).
 * Pure scheduling: any order gives the same frame (tests/test_gpu_parity.py).  No reference counterpart
 * (renderer_raytracing.cpp:666-685 leaves scheduling to the driver).  0 = off; N > 0 = re-sort every N launches.
 * Measured (4K over 1024^3, one frame at a time): walk alone 216 us row-major, 168 us in this order. */
int blok_hip_set_tile_ordering(blok_hip_ctx* ctx, int resort_every_n_frames);
/* The same for a camera in MOTION (round 3; default on): the heavy tiles of the next frame are near the heavy tiles of this one, not on
 * them.  After a frame that had the device to itself (and whose predecessor had too), a counting sort in three small launches keys every tile by the largest clocks within a few tiles of it (2-8, sized
 * by what the previous shift left over), and the next launch carries that order to its own view by ONE whole-tile shift of the screen
 * — entry (tx, ty) names tile (tx + sx, ty + sy) modulo the grid, still a permutation — computed on the host from the two cameras and
 * the depth range of the frame the order was measured on (blok_amd/csrc/hip/launch_policy.h: plan_shift).  Used while what the shift
 * leaves over (parallax, the stretch of a rotation towards the screen's edge) stays within twice the dilation; beyond that, and beside
 * frames in flight on other streams, a moving camera's launches keep row-major order as before.  Pure scheduling, as above.
 * Measured (4K over 1024^3, walk alone, camera orbiting by 1-2 degrees per frame): 209-216 us row-major, 171-178 us carried order, 167-170
 * us in the order of the frame's own clocks (profiles/r03_moving_order_experiment.txt).  0 = off. */
int blok_hip_set_moving_order(blok_hip_ctx* ctx, int enabled);
/* The same scheduling for a RANK's tile launches (blok_hip_trace_tiles_device / _tile_frames_device, 4096 wave tiles per frame or more;
 * round 4, default on): the rank's wave tiles are walked longest first and, in the automatic launch form, walk workgroups are dispatched for
 * the live prefix only — for a view at rest (all cameras of a several-frame launch the same view): a whole-tile shift of the screen does
 * not map a rank's round-robin share of the tiles onto itself, so a camera in motion keeps the natural order there.  Pure scheduling
 * (tests/test_gpu_parity.py).  0 = off (round 3's launches). */
int blok_hip_set_rank_tile_ordering(blok_hip_ctx* ctx, int enabled);
/* Diagnostic / test hook: the counting sort that follows a moving camera's frames, run on the caller's host arrays — cost[tiles_x * tiles_y]
 * (row-major wave tiles), dilation radius <= 8, optionally n_beams start parameters (>= 1e38 = none) — and read back: out_order (tiles by
 * descending class of the largest cost within `radius` tiles: 64 classes, four to the octave from 256 up, 0 = nothing near; within a class by
 * 16x16-tile block of the grid, row-major, then row-major inside the block), its inverse, *out_live = entries of classes > 0, and
 * out_depth_sums3 = (count, sum, sum of squares) of 1 / max(start parameter, 1) over the beam tiles that have one (may be null). */
int blok_hip_debug_class_order(blok_hip_ctx* ctx, const uint32_t* cost_host, uint32_t tiles_x, uint32_t tiles_y, uint32_t radius, const float* beam_host, uint32_t n_beams,
                               uint32_t* out_order_host, uint32_t* out_rank_of_host, uint32_t* out_live, float* out_depth_sums3);
/* Diagnostic: wave tiles the search waves of the latest TIMED prefix launch (blok_hip_set_timing) walked themselves (tiles live now that had
 * no walk workgroup: a changed view); read it after synchronising.  -1 = null context or nothing allocated yet. */
int64_t blok_hip_last_fallback_tiles(const blok_hip_ctx* ctx);
/* Test hook: every launch that walks in an order (of either kind) applies this whole-tile shift to it (taken modulo the launch's grid)
 * instead of the one the cameras give — any shift of any order is a permutation of the tiles, so the frames must not change. */
int blok_hip_debug_force_order_shift(blok_hip_ctx* ctx, int enabled, uint32_t shift_x, uint32_t shift_y);
/* Diagnostic: what the latest rectangle launch walked in — 0 row-major order, 1 an order of its own view, 2 an order carried over from
 * another view by the shift returned through the pointers (wave tiles, modulo the grid; either may be null); -1 = null context. */
int blok_hip_last_order_use(const blok_hip_ctx* ctx, int32_t* out_shift_x, int32_t* out_shift_y);
/* Diagnostic: the most walk waves a launch over the order's live prefix dispatches (0 = no limit).  Whatever is cut off is walked by the
 * search waves; the frame is the same (the tests use it to exercise that path). */
int blok_hip_set_joint_prefix_limit(blok_hip_ctx* ctx, uint32_t max_walk_waves);
/* Who writes the miss pixels of the tiles the pre-pass found empty (two-launch form): 1 (default) = the walk launch's waves of
 * those tiles — they are launched anyway and have nothing else to do — 0 = the pre-pass wave of the tile, 1 024 pixels each, which
 * puts ~120 MB of stores on the pre-pass's critical path (4K, 73 % sky).  Never changes a result. */
int blok_hip_set_miss_writer(blok_hip_ctx* ctx, int in_walk);
/* Node visits one beam search may spend (0 = the default, 256; searches average 35).  A search that runs out answers with the
 * lower bound over the cells it has not visited yet — valid, only less tight — never "none", so the frame is the same whatever
 * the budget (tests/test_gpu_parity.py runs with budgets of 1-64 visits against an unlimited search).  The pre-pass lasts as
 * long as its longest search, so the budget bounds its duration; too small a budget is paid for by the walk (beam.h). */
int blok_hip_set_beam_budget(blok_hip_ctx* ctx, uint32_t max_node_visits);
/* Launch form of a rectangle / tile frame (no reference counterpart: blok/src/renderer_raytracing.cpp:666-685 issues one
 * traceRaysKHR per frame); results are identical in every form (tests/test_gpu_parity.py):
 *   0  two launches: the beam kernel, then one walk wave per 8x8 pixels;
 *   1  one persistent launch with work queues: resident waves first take beam tiles, append the wave-sized sub-tiles of the live
 *      ones to per-part queues with an atomic reservation, then take walk tasks from those queues with one ticket each (measures
 *      slower on MI355X: same-address atomics run at 88 M/s);
 *   2  joint launch: the search waves and the walk waves are ONE grid, statically — workgroups are dispatched in index order, the
 *      searches first; a walk wave waits (bounded) only while its own tile's search is still running, so the chip starts walking
 *      when the first searches end, not when the last one does.  With a longest-first order in force (blok_hip_set_tile_ordering)
 *      walk waves are dispatched only for the tiles that walked when the order was made; a tile that has become live since is walked
 *      by its search wave.  Not for frames in flight on several streams: the waiting waves hold slots other frames' waves would
 *      use, and several joint launches in flight can wait for each other's searches in a circle until they give up (bounded; the
 *      frame stays exact);
 *   3  (default) automatic: 2 for a launch that has the DEVICE to itself — no frame launch of another stream or context of this
 *      process still pending on it — else 0, over the order's live prefix when an order is in force;
 *   4  list-fed joint launch: as 2, but the walk waves take their 8x8-pixel tiles from the frame's LIVE LISTS — a search that finds
 *      its beam tile live appends the tile's wave tiles, with their start parameter, to lists by cost class (blok_hip_set_list_classes;
 *      one 64-bit add per list reserves the slots), walk wave k of a list takes its entries k, k + n, ... — so no wave is launched for
 *      a dead tile and no pixel depends on an earlier frame: the walk grid is sized from the previous launch's lists (a hint only; any
 *      size walks every entry).  A walk wave that waits in vain (bounded, ~1 ms) leaves its entries to a clean-up launch behind the frame;
 *   5  the same lists in two launches: the beam kernel fills them, the walk waves take them (nothing waits).
 * Measured at 4K over 1024^3, a launch alone, camera at rest / orbiting by 1 degree per frame (profiles/r03_*): form 0 0.29 / 0.29 ms,
 * 2 with the order 0.20 / (no order) 0.24, 4 0.23 / 0.26, 5 0.26 / 0.29 — the lists lose to the measured order because entries arrive in
 * the order the searches finish, the heavy tiles last; they stay as the forms that need nothing from earlier frames. */
int blok_hip_set_fused(blok_hip_ctx* ctx, int enabled);
/* Health check of forms 1, 2 and 4: synchronises the device and returns how many waves ever gave up a bounded wait (0 in a working
 * system).  Form 1: for a queue entry (~0.5 s; frames since context creation may then be incomplete).  Form 2: for their tile's
 * search (such a wave starts at the ray origin instead: the frame is still exact).  Form 4: for a list entry (walked by the clean-up
 * launch: the frame is still exact). */
int blok_hip_frame_queue_stalls(blok_hip_ctx* ctx, uint32_t* out_stalled_waves);
/* Diagnostics of the pre-pass and the walk's scheduling (no reference counterpart).  blok_hip_beam_prepass: the pre-pass alone over a
 * rectangle — per beam tile (row-major, blok_hip_set_beam pixels each) its start parameter (>= 3e38 = no ray of the tile can hit
 * anything) and, optionally, the node visits its search spent — to host arrays of `capacity` elements.
 * blok_hip_trace_wave_tiles_device: walks exactly the listed 8x8-pixel wave tiles of the rectangle (index = row * ceil(w / 8) + column),
 * walk workgroup j taking entry j, each ray starting at its tile's t0 (NULL = at the ray origin; a value beyond the true first hit
 * would lose it — use what blok_hip_beam_prepass returned for the tile's beam tile); pixels of tiles not listed are left as they are.
 * A frame assembled from the two equals blok_hip_trace_primary_device's (tests/test_gpu_parity.py).
 * blok_hip_set_debug_wave_clocks: device array of one word per wave tile that list-fed walk waves leave their clock count in (NULL = off). */
int blok_hip_beam_prepass(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                          float* out_t0_host, uint32_t* out_visits_host_or_null, size_t capacity);
int blok_hip_trace_wave_tiles_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                     const uint32_t* tiles_host, const float* t0_host_or_null, size_t n_tiles,
                                     void* out_hits_dev_or_null, void* out_rgba8_dev_or_null, void* hip_stream);
int blok_hip_set_debug_wave_clocks(blok_hip_ctx* ctx, void* clocks_dev_or_null);
/* Which kernels the latest rectangle / tile launch of the context was issued as: 0 walk alone (no pre-pass), 1 two launches, 2 queues,
 * 3 joint, 4 list-fed joint, 5 beam launch + list-fed walk; -1 before the first launch (what form 3 chose; tests and diagnostics). */
int blok_hip_last_launch_kind(const blok_hip_ctx* ctx);
/* Cost classes of the list launches (rectangle frames; default on): walk waves leave the clocks they took per wave tile, and the next
 * frame's searches put every live wave tile into one of four lists by what its place on the previous frame's screen cost, heaviest
 * list first — so the waves of grazing rays, a quarter of the launch long each, start first instead of forming its tail.  Only the
 * order depends on the previous frame; 0 = one list in the order the searches finish.  Never changes a result. */
int blok_hip_set_list_classes(blok_hip_ctx* ctx, int enabled);

/* ------------------------------------------------------------ several devices */
/* Diagnostic: deny != 0 = behave as if the root had no peer access to the other devices (sparse-pull unavailable, mode 1 refused, every
 * call takes the dense exchange over blok_hip_multi_transport); 0 = back to what the node really offers.  The frames are the same. */
int blok_hip_multi_debug_deny_peer_access(blok_hip_multi* m, int deny);

#ifdef __cplusplus
}
#endif
#endif /* BLOK_HIP_DEBUG_H */
