// See trace_kernels.h for the semantics this file implements.
// Compiled with -ffp-contract=off; the float ops that define results additionally use the
// explicitly rounded intrinsics so that no flag can fuse them.
#include "trace_kernels.h"
#include "trace_core.h"
#include "path_core.h"
#include "beam.h"

namespace blok {

namespace {

// Tile (bx, by) of workgroup b for a rectangle of bx_count x by_count tiles; false if the workgroup has no tile.
__device__ __forceinline__ bool block_to_tile(uint32_t b, uint32_t grid, uint32_t bx_count, uint32_t by_count,
                                              uint32_t& bx, uint32_t& by) {
#if BLOK_XCD_MAP == 0
    (void)grid;
    bx = b % bx_count; by = b / bx_count;
    return by < by_count;
#else
    const uint32_t vb = (b & 7u) * (grid >> 3) + (b >> 3);        // XCD k owns virtual blocks [k*grid/8, (k+1)*grid/8)
#if BLOK_XCD_MAP == 1
    bx = vb % bx_count; by = vb / bx_count;
    return by < by_count;
#else
    const uint32_t super_x = (bx_count + kSuper - 1u) / kSuper;
    const uint32_t st = vb / (kSuper * kSuper), in = vb % (kSuper * kSuper);
    // Morton order inside the supertile: even bits -> x, odd bits -> y
    uint32_t mx = in & 0x55u, my = (in >> 1) & 0x55u;
    mx = (mx | (mx >> 1)) & 0x33u; mx = (mx | (mx >> 2)) & 0x0Fu;
    my = (my | (my >> 1)) & 0x33u; my = (my | (my >> 2)) & 0x0Fu;
    bx = (st % super_x) * kSuper + mx; by = (st / super_x) * kSuper + my;
    return bx < bx_count && by < by_count;
#endif
#endif
}

// joint launch: a beam tile's published word (trace_kernels.h).  Relaxed agent-scope atomics on the word itself — it validates
// itself — as in the one-launch frame below (sc1 loads see another XCD's publication: scripts/microbench/xcd_poll.hip).
__device__ __forceinline__ void publish_beam(unsigned long long* slot, uint32_t serial, float t0) {
    (void)__hip_atomic_exchange(slot, (static_cast<unsigned long long>(serial) << 32) | __float_as_uint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// All 64 lanes call this with the same slot; the result is wave-uniform.  A wave that has waited kJointPollBudget polls starts
// at the ray origin instead (always a valid start parameter), so every wave reaches its exit whatever the dispatch order.
__device__ __forceinline__ float await_beam(unsigned long long* slot, uint32_t serial, uint32_t* gave_up) {
    uint32_t lo = 0u, hi = 0u, budget = kJointPollBudget;
    for (;;) {
        const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)); hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
        if (hi == serial) return __uint_as_float(lo);
        if (--budget == 0u) break;
        __builtin_amdgcn_s_sleep(16);
    }
    if (gave_up && threadIdx.x == 0) (void)__hip_atomic_fetch_add(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0.0f;
}


// ---- live list (trace_kernels.h: LiveList) ---------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long list_ld(unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void list_st(unsigned long long* p, unsigned long long v) { (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One 64-bit fetch-add per WAVE, performed by lane 0 inside one opaque instruction sequence (see wave_fetch_add below for why), result
// wave-uniform.  Must be called with all 64 lanes active.
__device__ __forceinline__ unsigned long long wave_fetch_add64(unsigned long long* counter, unsigned long long v) {
    unsigned long long result, saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add_x2 %[res], %[zero], %[val], %[ptr] sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, %[saved]\n\t"
        "s_nop 4"
        : [res] "=&v"(result), [saved] "=&s"(saved)
        : [val] "v"(v), [zero] "v"(0u), [ptr] "s"(counter)
        : "memory");
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(result)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(result >> 32));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
// entry = serial (20 bits) | task (21 bits) | start parameter (the 23 bits 30..8 of a non-negative float: rounded DOWN, still a lower bound)
__device__ __forceinline__ unsigned long long list_entry(uint32_t serial, uint32_t task, float t0) {
    return (static_cast<unsigned long long>(serial) << 44) | (static_cast<unsigned long long>(task) << 23) | (__float_as_uint(t0) >> 8);
}
constexpr uint32_t kListPollBudget = 1024u;      // polls (~1 us apart) before a walk wave stops waiting for an entry: ~1 ms, ten times the longest search

// Which cost class a wave tile of a rectangle launch goes to: by the clocks the walk wave of the tile took last time, looked up where
// the tile's point at its start parameter was on the previous frame's screen (trace_kernels.h: cost classes).  (bx, by): the wave tile.
__device__ __forceinline__ uint32_t list_class_rect(const TraceArgs& A, const uint32_t bx, const uint32_t by, const uint32_t bx_count, const uint32_t by_count, const float t0) {
    const LiveList& Q = A.list;
    if (!Q.cost) return kListUnknownClass;
    uint32_t was = by * bx_count + bx;                                    // the same screen position: a camera at rest, or no camera to compare with
    if (Q.has_prev) {
        const float inv_w = __builtin_amdgcn_rcpf(static_cast<float>(A.frame_w)), inv_h = __builtin_amdgcn_rcpf(static_cast<float>(A.frame_h));
        const BeamVec d = beam_unit(beam_dir(A.cam, static_cast<float>(A.x0 + bx * kTileW) + 0.5f * kTileW, static_cast<float>(A.y0 + by * kTileH) + 0.5f * kTileH, inv_w, inv_h));
        const blok_camera& P = Q.prev_cam;
        const float rx = __builtin_fmaf(d.x, t0, A.cam.pos[0]) - P.pos[0], ry = __builtin_fmaf(d.y, t0, A.cam.pos[1]) - P.pos[1], rz = __builtin_fmaf(d.z, t0, A.cam.pos[2]) - P.pos[2];
        const float z = rx * P.fwd[0] + ry * P.fwd[1] + rz * P.fwd[2];
        if (!(z > 0.0f)) return kListUnknownClass;                        // behind the previous camera
        const float iz = __builtin_amdgcn_rcpf(z * P.tan_half_fov);
        const float u = (rx * P.right[0] + ry * P.right[1] + rz * P.right[2]) * iz * __builtin_amdgcn_rcpf(P.aspect), v = (rx * P.up[0] + ry * P.up[1] + rz * P.up[2]) * iz;
        const float px = (u + 1.0f) * 0.5f * static_cast<float>(A.frame_w) - static_cast<float>(A.x0), py = (1.0f - v) * 0.5f * static_cast<float>(A.frame_h) - static_cast<float>(A.y0);
        if (!(px >= 0.0f && py >= 0.0f && px < static_cast<float>(A.w) && py < static_cast<float>(A.h))) return kListUnknownClass;      // was off the rectangle (NaN included)
        was = min(static_cast<uint32_t>(py) / kTileH, by_count - 1u) * bx_count + min(static_cast<uint32_t>(px) / kTileW, bx_count - 1u);
    }
    const uint32_t c = Q.cost[was];
    if (c == 0u) return kListUnknownClass;
    return c >= kListCostClass3 ? 3u : (c >= kListCostClass2 ? 2u : (c >= kListCostClass1 ? 1u : 0u));
}

// The search wave of beam tile b (search workgroup `search` of the launch) has found start parameter t0: a live tile's wave tiles
// go onto the lists of the search's segment, by cost class, and the search is counted as done.  All 64 lanes.
template <RayMode MODE>
__device__ __forceinline__ void list_publish(const TraceArgs& A, const uint32_t search, const uint32_t task_base, const uint32_t b, const float t0, const uint32_t lane) {
    const LiveList& Q = A.list;
    const uint32_t seg = search & (kListSegments - 1u);
    const uint32_t subs_x = A.beam_tile / kWaveW, subs_y = A.beam_tile / kWaveH;
    bool valid = false;
    uint32_t task = 0, cls = kListUnknownClass;
    if (t0 < kBeamNone && lane < subs_x * subs_y) {
        const uint32_t sx = lane % subs_x, sy = lane / subs_x;
        if constexpr (MODE == RayMode::Rect) {
            const uint32_t bx_count = (A.w + kTileW - 1u) / kTileW, by_count = (A.h + kTileH - 1u) / kTileH;
            const uint32_t bx = (b % A.beam_bx) * subs_x + sx, by = (b / A.beam_bx) * subs_y + sy;
            valid = bx < bx_count && by < by_count;                     // an edge tile of the rectangle may be cut
            task = by * bx_count + bx;
            if (valid) cls = list_class_rect(A, bx, by, bx_count, by_count, t0);
        } else {
            const uint32_t per_side = A.tile / kTileW, bps = A.tile / A.beam_tile;
            const uint32_t local_tile = b / (bps * bps), rem = b % (bps * bps);
            task = local_tile * per_side * (A.tile / kTileH) + ((rem / bps) * subs_y + sy) * per_side + (rem % bps) * subs_x + sx;
            valid = true;
        }
    }
    unsigned long long* const ctl = Q.ctl + seg * kListCtlWords;
    for (uint32_t c = 0; c < kListClasses; ++c) {                         // wave-uniform: one reservation per class this tile has entries for
        const unsigned long long m = __ballot(valid && cls == c);
        if (m == 0ull) continue;
        const uint32_t n = static_cast<uint32_t>(__builtin_popcountll(m));
        const uint32_t pos = static_cast<uint32_t>(wave_fetch_add64(ctl + kListTally + c, n));
        if (valid && cls == c) {
            const uint32_t slot = pos + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
            if (slot < Q.seg_capacity) list_st(Q.entries + (static_cast<size_t>(seg) * kListClasses + c) * Q.seg_capacity + slot, list_entry(Q.serial, task_base + task, t0));
        }
    }
    // every reservation of this search has returned (wave_fetch_add64 waits for its value) before the search counts itself as done
    const uint32_t done = static_cast<uint32_t>(wave_fetch_add64(ctl + kListDone, 1ull));
    const uint32_t in_seg = (Q.n_searches + kListSegments - 1u - seg) / kListSegments;       // searches of this segment
    if (done + 1u == in_seg && lane < kListClasses) {
        // the last search of the segment: every reservation has been made, so the lengths are final; the counters are ready for the next launch
        const uint32_t total = min(static_cast<uint32_t>(list_ld(ctl + kListTally + lane)), Q.seg_capacity);
        list_st(ctl + kListFinal + lane, (static_cast<unsigned long long>(Q.serial) << 32) | total);
        list_st(ctl + kListTally + lane, 0ull);
        if (lane == 0) list_st(ctl + kListDone, 0ull);
        if (Q.hint) Q.hint[seg * kListClasses + lane] = total;
    }
}

// Entry k of a segment, for all 64 lanes (wave-uniform result): true with the entry, false when the list is final and shorter, or when
// the wait ran out (counted: the entry, should it still come, is left to list_cleanup_kernel).
__device__ __forceinline__ bool list_await(unsigned long long* slot, unsigned long long* final_word, unsigned long long* gave_up_word, const uint32_t serial, const uint32_t k,
                                           uint32_t* gave_up, uint32_t& task, float& t0) {
    // The entry has its own word; the segment's `final` word is shared by every waiting wave of the segment, so it is looked at only
    // every 8th poll (it matters to the waves beyond the end of the list alone, and those have nothing to do anyway).
    for (uint32_t polls = 0; polls < kListPollBudget; ++polls) {
        const unsigned long long v = list_ld(slot);
        const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
        if ((hi >> 12) == serial) {
            task = ((hi & 0xFFFu) << 9) | (lo >> 23);
            t0 = __uint_as_float((lo & 0x7FFFFFu) << 8);
            return true;
        }
#ifndef BLOK_LIST_FINAL_EVERY
#define BLOK_LIST_FINAL_EVERY 8
#endif
#ifndef BLOK_LIST_SLEEP
#define BLOK_LIST_SLEEP 32
#endif
        if ((polls & (BLOK_LIST_FINAL_EVERY - 1u)) == 0u) {
            const unsigned long long f = list_ld(final_word);
            const uint32_t flo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(f)), fhi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(f >> 32));
            if (fhi == serial && k >= flo) return false;
        }
        __builtin_amdgcn_s_sleep(BLOK_LIST_SLEEP);
    }
    list_st(gave_up_word, serial);               // "a wave of launch `serial` gave up in this segment": what list_cleanup_kernel looks at (every lane, same word)
    if (gave_up && threadIdx.x == 0) (void)__hip_atomic_fetch_add(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// Workgroup `block` of a launch of `grid` workgroups (the kernels below differ in where the arguments come from).  kListed: the block
// comes from the frame's live list together with its start parameter (list launches): no order, no beam lookup, no cost.
template <RayMode MODE, bool kListed = false>
__device__ __forceinline__ void trace_block(const TraceArgs& A, const uint32_t block, const uint32_t grid, uint4* lds_stack, const float listed_t0 = 0.0f) {
    const uint32_t tid = threadIdx.x;
    uint4* stk = lds_stack + tid;

    if constexpr (MODE == RayMode::Rays) {
        const uint32_t i = block * kBlock + tid;
        if (i >= A.n_rays) return;
        const blok_ray ray = A.rays[i];
        RayIn r{ray.org[0], ray.org[1], ray.org[2], ray.dir[0], ray.dir[1], ray.dir[2], ray.tmin, ray.tmax};
        trace_one(A, r, stk, Sink{A.out ? A.out + i : nullptr, A.out_rgba ? A.out_rgba + i : nullptr});
        return;
    } else {
        // a block is a 16x16 pixel tile, a wave an 8x8 sub-tile (coherent rays per wave)
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t lx = (wave & 1u) * kWaveW + (lane % kWaveW);
        const uint32_t ly = (wave >> 1) * kWaveH + (lane / kWaveW);
        // where this wave's 8x8 pixels start inside the block (wave-uniform)
        const uint32_t wave_x = __builtin_amdgcn_readfirstlane((wave & 1u) * kWaveW), wave_y = __builtin_amdgcn_readfirstlane((wave >> 1) * kWaveH);
        uint32_t x, y;
        [[maybe_unused]] uint64_t clock0 = 0;      // the wave's cost (clocks from here to its last record) goes to cost_slot
        [[maybe_unused]] uint32_t* cost_slot = nullptr;
        float t0 = 0.0f;                           // start parameter of this wave's beam tile (beam.h); kBeamNone: the pre-pass
        size_t out_index;                          // has already written the tile's pixels as misses, nothing left to do
        bool inside;
        if constexpr (MODE == RayMode::Rect) {
            const uint32_t bx_count = (A.w + kTileW - 1u) / kTileW, by_count = (A.h + kTileH - 1u) / kTileH;
            uint32_t bx, by;
            // longest first: workgroup i walks tile order[i] (the tiles by descending cost of their wave in earlier frames of the same view,
            // api.hip); any permutation gives the same frame.  A list entry names the tile itself.
            if constexpr (kListed) { bx = block % bx_count; by = block / bx_count; if (by >= by_count) return; }
            else {
                if (A.n_strip && block < A.n_strip) {
                    // a tile of the strips the shift brought in: columns first (strip_nx wide, every row), then what is left of the rows
                    const uint32_t in_columns = A.strip_nx * by_count;
                    if (block < in_columns) { bx = A.strip_x0 + block % A.strip_nx; by = block / A.strip_nx; }
                    else { const uint32_t k = block - in_columns, w = bx_count - A.strip_nx; bx = k % w; if (bx >= A.strip_x0) bx += A.strip_nx; by = A.strip_y0 + k / w; }
                    if (bx >= bx_count || by >= by_count) return;
                } else {
                    const uint32_t entry = block - A.n_strip;
                    const uint32_t tile = A.order ? A.order[entry] : entry;
                    if (!block_to_tile(tile, grid, bx_count, by_count, bx, by)) return;
                    if (A.order) {                                       // the order's tile, carried to this view (wave-uniform; 0, 0 for an order of this view)
                        bx += A.order_sx; if (bx >= bx_count) bx -= bx_count;
                        by += A.order_sy; if (by >= by_count) by -= by_count;
                        if (A.n_strip && (bx - A.strip_x0 < A.strip_nx || by - A.strip_y0 < A.strip_ny)) return;      // landed in a strip: that tile has its own workgroup
                    }
                }
                cost_slot = A.cost_out ? A.cost_out + (by * bx_count + bx) : nullptr;
            }
            if constexpr (kListed) t0 = listed_t0;
            else if (A.beam) {
                const uint32_t beam_index = ((by * kTileH + wave_y) / A.beam_tile) * A.beam_bx + (bx * kTileW + wave_x) / A.beam_tile;
                t0 = A.beam_slots ? await_beam(A.beam_slots + beam_index, A.beam_serial, A.joint_gave_up) : A.beam[beam_index];
                if (t0 >= kBeamNone) {
                    if (cost_slot && tid == 0) *cost_slot = 0u;
                    if (A.miss_in_walk) {
                        const uint32_t mx = bx * kTileW + lx, my = by * kTileH + ly;
                        const size_t at = static_cast<size_t>(my) * A.w + mx;
                        if (mx < A.w && my < A.h) write_miss(Sink{A.out ? A.out + at : nullptr, A.out_rgba ? A.out_rgba + at : nullptr});
                    }
                    return;
                }
            }
            const uint32_t rx = bx * kTileW + lx, ry = by * kTileH + ly;
            inside = rx < A.w && ry < A.h;
            x = A.x0 + rx; y = A.y0 + ry;
            out_index = static_cast<size_t>(ry) * A.w + rx;
        } else {
            const uint32_t per_side = A.tile / kTileW;                  // blocks per tile row
            const uint32_t per_tile = per_side * (A.tile / kTileH);
            // longest first, as for rectangles: workgroup i of a frame walks the rank's wave tile order[i] (a listed block names its wave tile itself)
            const uint32_t wave_tile = (!kListed && A.order) ? A.order[block] : block;
            if constexpr (!kListed) cost_slot = A.cost_out ? A.cost_out + wave_tile : nullptr;
            const uint32_t local_tile = wave_tile / per_tile, sub = wave_tile % per_tile;
            const uint32_t global_tile = A.rank + local_tile * A.n_ranks;
            const uint32_t tx = global_tile % A.tiles_x, ty = global_tile / A.tiles_x;
            const uint32_t ix = (sub % per_side) * kTileW + lx, iy = (sub / per_side) * kTileH + ly;
            x = tx * A.tile + ix; y = ty * A.tile + iy;
            out_index = static_cast<size_t>(local_tile) * A.tile * A.tile + static_cast<size_t>(iy) * A.tile + ix;
            inside = x < A.frame_w && y < A.frame_h;
            if (global_tile >= A.tiles_total) {                        // whole workgroup: padding tile of the last round
                write_miss(Sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr});
                if (cost_slot && tid == 0) *cost_slot = 0u;            // (a prefix launch leaves these to the tile's search wave, which writes them as misses)
                return;
            }
            if constexpr (kListed) t0 = listed_t0;
            else if (A.beam) {
                const uint32_t beams_per_side = A.tile / A.beam_tile;
                const uint32_t beam_index = (local_tile * beams_per_side + ((sub / per_side) * kTileH + wave_y) / A.beam_tile) * beams_per_side +
                                            ((sub % per_side) * kTileW + wave_x) / A.beam_tile;
                t0 = A.beam_slots ? await_beam(A.beam_slots + beam_index, A.beam_serial, A.joint_gave_up) : A.beam[beam_index];
                if (t0 >= kBeamNone) {
                    if (cost_slot && tid == 0) *cost_slot = 0u;
                    if (A.miss_in_walk) write_miss(Sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr});   // the tile buffer is dense
                    return;
                }
            }
        }
        const Sink sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr};
        if (!inside) {
            if constexpr (MODE == RayMode::Tiles) {
                write_miss(sink);                                       // the tile buffer is dense
                // a wave tile that straddles or lies beyond the frame's edge always keeps its walk workgroup in a prefix launch (someone has to write these)
                if (cost_slot && tid == 0) *cost_slot = 256u;
            }
            return;
        }
        if constexpr (MODE == RayMode::Rect) {
            if constexpr (kListed) cost_slot = A.debug_clocks ? A.debug_clocks + block : nullptr;
        }
        if (cost_slot && tid == 0) clock0 = __builtin_amdgcn_s_memtime();      // the walk's cost: any wait for the tile's search is not part of it
        RayIn r = primary_ray(A, x, y);
        r.tmin = fmaxf(r.tmin, t0);
#ifndef BLOK_WAVE_START
#define BLOK_WAVE_START 1
#endif
        trace_one<BLOK_WAVE_START != 0>(A, r, stk, sink);
        if (cost_slot && tid == 0) *cost_slot = max(static_cast<uint32_t>(__builtin_amdgcn_s_memtime() - clock0), 256u);     // it walked: any key >= 256 clocks counts as live in the next order
    }
}

template <RayMode MODE>
__global__ __launch_bounds__(kBlock) void trace_kernel(const TraceArgs A) {
    extern __shared__ uint4 lds_stack[];       // [levels-1][kBlock]
    trace_block<MODE>(A, blockIdx.x, gridDim.x, lds_stack);
}

// The arguments of frame f of a several-frame launch: its camera, its slice of the outputs and of the beam buffer.
__device__ __forceinline__ TraceArgs frame_args(const TraceArgs& A, const TileFrames& F, const uint32_t f) {
    TraceArgs L = A;
    L.cam = F.cam[f];
    if (L.out) L.out += f * F.frame_stride;
    if (L.out_rgba) L.out_rgba += f * F.frame_stride;
    if (L.beam) L.beam += static_cast<size_t>(f) * F.beams_per_frame;
    return L;
}

// TileFrames::n_frames frames of the rank's tiles in one launch: workgroups [f * blocks_per_frame, (f + 1) * blocks_per_frame) are frame f.
__global__ __launch_bounds__(kBlock) void trace_frames_kernel(const TraceArgs A, const TileFrames F) {
    extern __shared__ uint4 lds_stack[];
    const uint32_t f = blockIdx.x / F.blocks_per_frame;
    if (f >= F.n_frames) return;
    const TraceArgs L = frame_args(A, F, f);
    trace_block<RayMode::Tiles>(L, blockIdx.x - f * F.blocks_per_frame, F.blocks_per_frame, lds_stack);
}

// One wave per beam tile: TraceArgs::beam[tile] = conservative start parameter of the tile's rays, or kBeamNone.
// list_search / list_task_base (list launches): this search's index in the launch and the first task id of its frame.
// lds_stack: the walk's stack for the wave tiles a search walks itself (launches over a prefix of the order, TraceArgs::rank_of), else null.
template <RayMode MODE>
__device__ __forceinline__ void beam_block(const TraceArgs& A, const uint32_t b, const uint32_t n_beam_tiles, [[maybe_unused]] uint4* lds_stack = nullptr,
                                           const uint32_t list_search = 0u, const uint32_t list_task_base = 0u) {
    const uint32_t lane = threadIdx.x;
    if (b >= n_beam_tiles) return;
    const uint32_t B = A.beam_tile;
    uint32_t px, py, px_end, py_end;                                   // frame pixels [px, px_end) x [py, py_end)
    [[maybe_unused]] uint32_t tile_x0 = 0, tile_y0 = 0;                // Tiles: frame origin of the screen tile and its slot
    [[maybe_unused]] size_t tile_base = 0;                             // in the rank's dense tile buffer
    bool padding = false;
    if constexpr (MODE == RayMode::Rect) {
        px = A.x0 + (b % A.beam_bx) * B; py = A.y0 + (b / A.beam_bx) * B;
        px_end = min(px + B, A.x0 + A.w); py_end = min(py + B, A.y0 + A.h);
    } else {
        const uint32_t per_side = A.tile / B, per_tile = per_side * per_side;
        const uint32_t local_tile = b / per_tile, sub = b % per_tile;
        const uint32_t global_tile = A.rank + local_tile * A.n_ranks;
        padding = global_tile >= A.tiles_total;                        // a tile slot beyond the frame's last tile: all misses
        // (a padding tile's miss records are written by its walk workgroups — unless this launch has none for it: a prefix launch, miss_in_walk 0)
        if (padding && !A.list.entries && A.miss_in_walk) { if (lane == 0) { if (A.beam_slots) publish_beam(A.beam_slots + b, A.beam_serial, kBeamNone); else A.beam[b] = kBeamNone; } return; }
        tile_x0 = (global_tile % A.tiles_x) * A.tile; tile_y0 = (global_tile / A.tiles_x) * A.tile;
        tile_base = static_cast<size_t>(local_tile) * A.tile * A.tile;
        px = tile_x0 + (sub % per_side) * B; py = tile_y0 + (sub / per_side) * B;
        px_end = px + B; py_end = py + B;
    }
    uint32_t visits = 0;
    const float t0 = padding ? kBeamNone : beam_start(A, static_cast<float>(px), static_cast<float>(py), static_cast<float>(px_end), static_cast<float>(py_end), lane, kBeamNone,
                                                      A.debug_visits ? &visits : nullptr);
    if (A.debug_visits && lane == 0) A.debug_visits[b] = visits;
    if (A.list.entries) list_publish<MODE>(A, list_search, list_task_base, b, t0, lane);      // list launches: the live wave tiles go onto the frame's list
    else if (lane == 0) { if (A.beam_slots) publish_beam(A.beam_slots + b, A.beam_serial, t0); else A.beam[b] = t0; }
    if (t0 >= kBeamNone && (A.list.entries || !A.miss_in_walk) && (A.out || A.out_rgba)) {
        // no ray of this tile can hit anything: its pixels are written here, 64 at a time (a list launch has no walk wave for the tile;
        // in the other forms the tile's trace waves exit at once)
        const uint32_t tw = px_end - px, n = tw * (py_end - py);
        for (uint32_t i = lane; i < n; i += 64u) {
            const uint32_t x = px + i % tw, y = py + i / tw;
            size_t out_index;
            if constexpr (MODE == RayMode::Rect) out_index = static_cast<size_t>(y - A.y0) * A.w + (x - A.x0);
            else out_index = tile_base + static_cast<size_t>(y - tile_y0) * A.tile + (x - tile_x0);
            write_miss(Sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr});
        }
    }
    if constexpr (MODE == RayMode::Rect && kBlock == 64) {
        // joint launch over a prefix of the order: the wave-sized tiles of this (live) beam tile that no walk wave was dispatched for —
        // the view has changed since the order was made — are walked here, one after the other
        if (A.rank_of && lds_stack && t0 < kBeamNone) {
            const uint32_t bx_count = (A.w + kTileW - 1u) / kTileW, by_count = (A.h + kTileH - 1u) / kTileH;
            const uint32_t bx0 = (px - A.x0) / kTileW, bx1 = (px_end - A.x0 + kTileW - 1u) / kTileW;
            const uint32_t by0 = (py - A.y0) / kTileH, by1 = (py_end - A.y0 + kTileH - 1u) / kTileH;
            uint32_t walked = 0u;
            for (uint32_t by = by0; by < by1; ++by)
                for (uint32_t bx = bx0; bx < bx1; ++bx) {
                    const uint32_t tile = by * bx_count + bx;
                    // where the order knows this tile: the shift taken off again
                    if (A.n_strip && (bx - A.strip_x0 < A.strip_nx || by - A.strip_y0 < A.strip_ny)) continue;      // a strip tile: it has a walk workgroup of its own
                    const uint32_t ox = bx >= A.order_sx ? bx - A.order_sx : bx + bx_count - A.order_sx, oy = by >= A.order_sy ? by - A.order_sy : by + by_count - A.order_sy;
                    if (__builtin_amdgcn_readfirstlane(A.rank_of[oy * bx_count + ox]) < A.launched) continue;
                    const uint64_t clock0 = __builtin_amdgcn_s_memtime();
                    const uint32_t rx = bx * kTileW + lane % kWaveW, ry = by * kTileH + lane / kWaveW;
                    if (rx < A.w && ry < A.h) {
                        const size_t at = static_cast<size_t>(ry) * A.w + rx;
                        RayIn r = primary_ray(A, A.x0 + rx, A.y0 + ry);
                        r.tmin = fmaxf(r.tmin, t0);
                        trace_one(A, r, lds_stack + lane, Sink{A.out ? A.out + at : nullptr, A.out_rgba ? A.out_rgba + at : nullptr});
                    }
                    // it walked: the next sort puts it into the prefix (any key >= 256 clocks counts as live)
                    if (A.cost_out && lane == 0) A.cost_out[tile] = max(static_cast<uint32_t>(__builtin_amdgcn_s_memtime() - clock0), 256u);
                    ++walked;
                }
            if (walked && A.fallback_tiles && lane == 0) (void)__hip_atomic_fetch_add(A.fallback_tiles, walked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (MODE == RayMode::Tiles && kBlock == 64) {
        // the same for a rank's tiles (round 4): the wave tiles of this live beam tile that the order's prefix has no walk workgroup for
        if (A.rank_of && lds_stack && t0 < kBeamNone) {
            const uint32_t per_side = A.tile / kTileW, per_tile_blocks = per_side * (A.tile / kTileH), bps = A.tile / B;
            const uint32_t local_tile = b / (bps * bps), rem = b % (bps * bps);
            const uint32_t subs_x = B / kWaveW, subs_y = B / kWaveH;
            uint32_t walked = 0u;
            for (uint32_t sy = 0; sy < subs_y; ++sy)
                for (uint32_t sx = 0; sx < subs_x; ++sx) {
                    const uint32_t wave_tile = local_tile * per_tile_blocks + ((rem / bps) * subs_y + sy) * per_side + (rem % bps) * subs_x + sx;
                    if (__builtin_amdgcn_readfirstlane(A.rank_of[wave_tile]) < A.launched) continue;
                    const uint64_t clock0 = __builtin_amdgcn_s_memtime();
                    trace_block<RayMode::Tiles, true>(A, wave_tile, 0u, lds_stack, t0);
                    if (A.cost_out && lane == 0) A.cost_out[wave_tile] = max(static_cast<uint32_t>(__builtin_amdgcn_s_memtime() - clock0), 256u);
                    ++walked;
                }
            if (walked && A.fallback_tiles && lane == 0) (void)__hip_atomic_fetch_add(A.fallback_tiles, walked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <RayMode MODE>
__global__ __launch_bounds__(64) void beam_kernel(const TraceArgs A, const uint32_t n_beam_tiles) {
    extern __shared__ uint4 lds_stack[];       // launch_beam sizes it whenever the searches may walk (TraceArgs::rank_of), else 0 bytes and unused
    beam_block<MODE>(A, blockIdx.x, n_beam_tiles, A.rank_of ? lds_stack : nullptr, blockIdx.x, 0u);
}

// ---- joint launch: the pre-pass waves and the walk waves in ONE grid, statically ------------------------------------------
// Workgroups [0, n_beam) are the searches, the others the walk's waves in their usual order.  Workgroups are dispatched in index
// order, so all searches are resident before the first walk wave starts; a walk wave takes the slot of a search that has ended and
// waits (a bounded spin on one word) only if its own tile's search is still running.  The chip therefore starts walking when the
// FIRST searches end instead of when the LAST one does — the pre-pass is a 77 us tail of a 13 us average — without a queue, a
// ticket or any atomic read-modify-write on a shared address (what the one-launch frame below pays for), and without the gap
// between two dependent launches.  No reference counterpart (one traceRaysKHR per frame, renderer_raytracing.cpp:666-685).
template <RayMode MODE>
__global__ __launch_bounds__(kBlock) void joint_kernel(const TraceArgs A, const uint32_t n_beam_tiles) {
    extern __shared__ uint4 lds_stack[];
    if (blockIdx.x < n_beam_tiles) {
        if (threadIdx.x < 64u) beam_block<MODE>(A, blockIdx.x, n_beam_tiles, lds_stack);
        return;
    }
    trace_block<MODE>(A, blockIdx.x - n_beam_tiles, gridDim.x - n_beam_tiles, lds_stack);
}

__global__ __launch_bounds__(64) void beam_frames_kernel(const TraceArgs A, const TileFrames F) {
    extern __shared__ uint4 lds_stack[];       // sized by the launch whenever the searches may walk (TraceArgs::rank_of), else 0 bytes and unused
    const uint32_t f = blockIdx.x / F.beams_per_frame;
    if (f >= F.n_frames) return;
    const TraceArgs L = frame_args(A, F, f);
    beam_block<RayMode::Tiles>(L, blockIdx.x - f * F.beams_per_frame, F.beams_per_frame, A.rank_of ? lds_stack : nullptr, blockIdx.x, f * F.blocks_per_frame);
}

// ---- list launches: the walk takes its wave tiles from the frame's live list (trace_kernels.h: LiveList) -------------------------
// No reference counterpart (one traceRaysKHR per frame, renderer_raytracing.cpp:666-685).  Walk workgroup `walker` of the launch (on
// XCD `seg` if workgroups go round the XCDs in index order) takes entries walker / 8, + walkers_per_seg, ... of segment `seg`: normally
// one entry — the grid is sized from the previous launch's list, with a margin — and more only when the view has changed a lot.
// One listed wave tile: walked, and (rectangle launches) the clocks it took left for the next frame's classes.
template <RayMode MODE, bool kFrames>
__device__ __forceinline__ void list_task(const TraceArgs& A, const TileFrames& F, const uint32_t task, const float t0, uint4* lds_stack) {
    if constexpr (kFrames) {
        const uint32_t f = task / F.blocks_per_frame;
        if (f >= F.n_frames) return;
        const TraceArgs L = frame_args(A, F, f);
        trace_block<MODE, true>(L, task - f * F.blocks_per_frame, F.blocks_per_frame, lds_stack, t0);
    } else {
        [[maybe_unused]] uint64_t clock0 = 0;
        if constexpr (MODE == RayMode::Rect) { if (A.list.cost) clock0 = __builtin_amdgcn_s_memtime(); }
        trace_block<MODE, true>(A, task, 0u, lds_stack, t0);
        if constexpr (MODE == RayMode::Rect) {
            if (A.list.cost && threadIdx.x == 0) A.list.cost[task] = max(static_cast<uint32_t>(__builtin_amdgcn_s_memtime() - clock0), 1u);
        }
    }
}

template <RayMode MODE, bool kFrames>
__device__ __forceinline__ void list_walk(const TraceArgs& A, const TileFrames& F, const uint32_t walker, const uint32_t seg, uint4* lds_stack) {
    const LiveList& Q = A.list;
    // a launch of fewer searches than segments (a rectangle of < 8 beam tiles, a rank with few tiles): segment `seg` has no search, nobody
    // will ever write its final word for this serial, and its lists are empty — nothing to wait for (such waves spun their whole
    // poll budget, ~1 ms, and tripped the stall counter)
    if (seg >= Q.n_searches) return;
    // the segment's walk workgroups are dealt to the classes heaviest first
    uint32_t k = walker / kListSegments, cls = kListClasses - 1u;
    while (cls != 0u && k >= Q.walkers[cls]) { k -= Q.walkers[cls]; cls -= 1u; }
    const uint32_t stride = Q.walkers[cls];
    if (k >= stride) return;                                               // more workgroups than the classes asked for
    unsigned long long* const entries = Q.entries + (static_cast<size_t>(seg) * kListClasses + cls) * Q.seg_capacity;
    unsigned long long* const ctl = Q.ctl + seg * kListCtlWords;
    for (; k < Q.seg_capacity; k += stride) {
        uint32_t task; float t0;
        if (!list_await(entries + k, ctl + kListFinal + cls, ctl + kListGaveUp, Q.serial, k, A.joint_gave_up, task, t0)) return;
        __hip_atomic_store(entries + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // taken (every lane, same word)
        list_task<MODE, kFrames>(A, F, task, t0, lds_stack);
    }
}

template <RayMode MODE, bool kFrames>
__device__ __forceinline__ void list_search(const TraceArgs& A, const TileFrames& F, const uint32_t search) {
    if constexpr (kFrames) {
        const uint32_t f = search / F.beams_per_frame;
        if (f >= F.n_frames) return;
        const TraceArgs L = frame_args(A, F, f);
        beam_block<MODE>(L, search - f * F.beams_per_frame, F.beams_per_frame, nullptr, search, f * F.blocks_per_frame);
    } else {
        beam_block<MODE>(A, search, A.list.n_searches, nullptr, search, 0u);
    }
}

// Searches and list-fed walk waves in ONE grid: workgroups [0, n_searches) search, the others walk.  Workgroups are dispatched in index
// order, so a walk wave finds every search of its segment resident or finished, and waits (bounded) only for entries they still owe.
template <RayMode MODE, bool kFrames>
__global__ __launch_bounds__(kBlock) void list_joint_kernel(const TraceArgs A, const TileFrames F) {
    static_assert(kBlock == 64, "one wave per workgroup");
    extern __shared__ uint4 lds_stack[];
    const uint32_t n = A.list.n_searches;
    if (blockIdx.x < n) { list_search<MODE, kFrames>(A, F, blockIdx.x); return; }
    list_walk<MODE, kFrames>(A, F, blockIdx.x - n, blockIdx.x & (kListSegments - 1u), lds_stack);
}

// The walk alone, behind a beam launch that has filled the list (nothing to wait for).
template <RayMode MODE, bool kFrames>
__global__ __launch_bounds__(kBlock) void list_walk_kernel(const TraceArgs A, const TileFrames F) {
    extern __shared__ uint4 lds_stack[];
    list_walk<MODE, kFrames>(A, F, blockIdx.x, blockIdx.x & (kListSegments - 1u), lds_stack);
}

// Behind a joint list launch: entries whose walk wave gave up waiting (none unless workgroups were dispatched in an order nobody has
// seen yet, or several joint launches starved each other's searches) are walked here, so the frame is complete whatever happened.
template <RayMode MODE, bool kFrames>
__global__ __launch_bounds__(kBlock) void list_cleanup_kernel(const TraceArgs A, const TileFrames F) {
    extern __shared__ uint4 lds_stack[];
    const LiveList& Q = A.list;
    const uint32_t seg = blockIdx.x & (kListSegments - 1u), part = blockIdx.x / kListSegments, parts = gridDim.x / kListSegments;
    unsigned long long* const ctl = Q.ctl + seg * kListCtlWords;
    if (seg >= Q.n_searches) return;                                                                               // a segment without searches (list_walk)
    if (__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(list_ld(ctl + kListGaveUp))) != Q.serial) return;      // nobody gave up in this launch
    for (uint32_t cls = 0; cls < kListClasses; ++cls) {
        const unsigned long long f = list_ld(ctl + kListFinal + cls);
        const uint32_t total = static_cast<uint32_t>(f >> 32) == Q.serial ? min(static_cast<uint32_t>(f), Q.seg_capacity) : Q.seg_capacity;
        unsigned long long* const entries = Q.entries + (static_cast<size_t>(seg) * kListClasses + cls) * Q.seg_capacity;
        for (uint32_t k = part; k < total; k += parts) {
            const unsigned long long v = list_ld(entries + k);
            const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
            if ((hi >> 12) != Q.serial) continue;
            __hip_atomic_store(entries + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            list_task<MODE, kFrames>(A, F, ((hi & 0xFFFu) << 9) | (lo >> 23), __uint_as_float((lo & 0x7FFFFFu) << 8), lds_stack);
        }
    }
}

// ---- one-launch frame: pre-pass and walk in ONE persistent grid ------------------------------------------------------
// No reference counterpart (the reference hands "which boxes does this ray meet" to RT hardware and issues one
// traceRaysKHR per frame, blok/src/renderer_raytracing.cpp:666-685).  The two-launch form (beam_kernel, then trace_kernel)
// leaves the chip idle twice per frame: the pre-pass is one round of latency-bound waves whose slowest search sets its
// duration, and the trace launch spends its first instruction in ~3/4 of its waves (sky tiles the pre-pass has already
// written) and ends in a tail of long grazing-ray waves.  Here a grid of resident waves pulls work from two counters:
//   phase 1  beam tasks, ctl[kBeamNext]++: search one beam tile; "none" -> write the tile's miss pixels; else append
//            the tile's wave-sized sub-tiles — only those, the live ones — to a queue (ctl[kReserved] += n, one 64-bit
//            entry each = task id | start parameter << 32), then ctl[kBeamDone]++;
//   phase 2  trace tickets, ctl[kHead]++: wait for entry i (producers publish it with one 8-byte store), walk its 64 rays.
// Early finishers of phase 1 start walking while the long searches are still running, and the queue holds no dead tile.
// Termination: a ticket beyond the final queue length (known once ctl[kBeamDone] == n_beam) ends the wave.  A waiting wave
// only ever waits for beam tasks that running waves hold (it has seen ctl[kBeamNext] >= n_beam), and those never wait, so
// every wave reaches its exit whatever part of the grid is resident, and a wait is bounded besides (kFramePollBudget).  All
// communication is relaxed agent-scope atomics on the words themselves (an entry validates itself), no fence: values are
// published with an exchange and polled with sc1 loads, which see another XCD's publication (the eight XCDs have one L2
// each; scripts/microbench/xcd_poll.hip measures every publish / poll pairing across XCDs: all work).  Consumers put the
// entry back to kNoTask and the last wave out zeroes the counters, so the buffers are ready for the next launch on the same
// stream.  Every counter has its own 128-byte line.
//
// SEVERAL queues, not one.  Atomics on one address are served at ~88 M/s, 11.5 ns each, whichever XCDs they come from
// (scripts/microbench/atomic_rate.hip; different addresses proceed in parallel): the first version, with one ticket counter
// for the chip, spent 0.9 ms per 4K frame on its ~80 K atomics.  So the launch is cut into Q.n_parts parts — workgroup b works
// in part b mod n_parts, which owns beam tiles part, part + n_parts, ..., its own slice of the entry array and its own
// counters — and a trace ticket is good for Q.chunk consecutive entries.  Interleaving the tiles balances the parts the way the
// round-robin dispatch of the two-launch form balances the XCDs.
enum : uint32_t { kBeamNext = 0, kBeamDone = 32, kReserved = 64, kHead = 96, kExited = 128, kStalled = 160, kPartWords = kFramePartWords };
static_assert(kStalled < kPartWords, "counter block");
constexpr unsigned long long kNoTask = ~0ull;

__device__ __forceinline__ uint32_t add_agent(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent(uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) { (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One fetch-add per WAVE, performed by lane 0, result wave-uniform (an SGPR).  Written as ONE opaque instruction sequence
// because the source-level form — if (lane == 0) v = atomic; v = readfirstlane(v) — inside a loop is restructured by the
// compiler (the condition is loop-invariant per lane, so the back edge of lanes 1-63 is threaded past the block): lane 0
// leaves to perform the atomic while the other lanes go round the body again with the stale value.  Measured: the first
// version of this kernel hung that way, also with the value handed over through LDS behind a (single-wave, hence elided)
// barrier.  Must be called with all 64 lanes active.
__device__ __forceinline__ uint32_t wave_fetch_add(uint32_t* counter, uint32_t v) {
    uint32_t result, tmp;
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "v_mov_b32 %[tmp], %[val]\n\t"
        "global_atomic_add %[tmp], %[zero], %[tmp], %[ptr] sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, %[saved]\n\t"
        "s_nop 4\n\t"
        "v_readfirstlane_b32 %[res], %[tmp]\n\t"
        "s_nop 3"
        : [res] "=s"(result), [tmp] "=&v"(tmp), [saved] "=&s"(saved)
        : [val] "s"(v), [zero] "v"(0u), [ptr] "s"(counter)
        : "memory");
    return result;
}

// Pixels and output slots of beam tile b.
struct BeamRect {
    uint32_t px, py, px_end, py_end;       // frame pixels [px, px_end) x [py, py_end)
    uint32_t ox, oy, stride;               // output index = obase + (y - oy) * stride + (x - ox)
    size_t obase;
    bool padding;                          // Tiles: a tile slot beyond the frame's last tile (all misses)
};
template <RayMode MODE>
__device__ __forceinline__ BeamRect beam_rect(const TraceArgs& A, uint32_t b) {
    const uint32_t B = A.beam_tile;
    BeamRect r;
    r.padding = false;
    if constexpr (MODE == RayMode::Rect) {
        r.px = A.x0 + (b % A.beam_bx) * B; r.py = A.y0 + (b / A.beam_bx) * B;
        r.px_end = min(r.px + B, A.x0 + A.w); r.py_end = min(r.py + B, A.y0 + A.h);
        r.ox = A.x0; r.oy = A.y0; r.stride = A.w; r.obase = 0;
    } else {
        const uint32_t per_side = A.tile / B, per_tile = per_side * per_side;
        const uint32_t local_tile = b / per_tile, sub = b % per_tile;
        const uint32_t global_tile = A.rank + local_tile * A.n_ranks;
        r.padding = global_tile >= A.tiles_total;
        r.ox = (global_tile % A.tiles_x) * A.tile; r.oy = (global_tile / A.tiles_x) * A.tile;
        r.stride = A.tile; r.obase = static_cast<size_t>(local_tile) * A.tile * A.tile;
        r.px = r.ox + (sub % per_side) * B; r.py = r.oy + (sub / per_side) * B;
        r.px_end = r.px + B; r.py_end = r.py + B;
    }
    return r;
}

template <RayMode MODE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void frame_kernel(const TraceArgs A, const FrameQueue Q) {
    static_assert(kBlock == 64, "one wave per workgroup");
    extern __shared__ uint4 lds_stack[];       // [levels-1][kBlock]
    const uint32_t lane = threadIdx.x;
    const uint32_t B = A.beam_tile;
    const uint32_t subs_x = B / kWaveW, subs_y = B / kWaveH;
    // this wave's part: its counters, its beam tiles (part, part + kFrameParts, ...), its slice of the entries
    const uint32_t n_parts = Q.n_parts;
    const uint32_t part = blockIdx.x % n_parts;
    uint32_t* const ctl = Q.ctl + part * kPartWords;
    const uint32_t n_beam = Q.n_beam > part ? (Q.n_beam - part + n_parts - 1u) / n_parts : 0u;
    unsigned long long* const entries = Q.entries + static_cast<size_t>(part) * Q.part_capacity;

    // ---- phase 1: beam tasks.  A search is one long chain of dependent instructions; at raised priority its instructions issue
    // ahead of the walking waves of the same SIMD (VALU-bound, any order will do), so the searches finish as early as alone.
    __builtin_amdgcn_s_setprio(3);
    for (;;) {
        if (__builtin_amdgcn_readfirstlane(ld_agent(ctl + kBeamNext)) >= n_beam) break;      // a load before the atomic: no wave takes a ticket only to learn that it is late
        const uint32_t k = wave_fetch_add(ctl + kBeamNext, 1u);
        if (k >= n_beam) break;
        const uint32_t b = part + k * n_parts;
        const BeamRect r = beam_rect<MODE>(A, b);
        float t0 = kBeamNone;
        if (!r.padding) t0 = beam_start(A, static_cast<float>(r.px), static_cast<float>(r.py), static_cast<float>(r.px_end), static_cast<float>(r.py_end), lane);
        if (t0 >= kBeamNone) {
            // no ray of this tile can hit anything: its pixels are written here, 64 at a time
            const uint32_t tw = r.px_end - r.px, n = tw * (r.py_end - r.py);
            for (uint32_t i = lane; i < n; i += 64u) {
                const uint32_t x = r.px + i % tw, y = r.py + i / tw;
                const size_t out_index = r.obase + static_cast<size_t>(y - r.oy) * r.stride + (x - r.ox);
                write_miss(Sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr});
            }
        } else {
            // the tile's wave-sized sub-tiles that lie inside the rectangle (an edge tile of a Rect launch may be cut)
            const uint32_t nx = (r.px_end - r.px + kWaveW - 1u) / kWaveW, ny = (r.py_end - r.py + kWaveH - 1u) / kWaveH;
            const uint32_t n = nx * ny;                                   // <= subs_x * subs_y <= 64
            const uint32_t pos = wave_fetch_add(ctl + kReserved, n);
            if (lane < n && pos + lane < Q.part_capacity) {
                const uint32_t task = b * (subs_x * subs_y) + (lane / nx) * subs_x + lane % nx;
                st_agent(entries + pos + lane, (static_cast<unsigned long long>(__float_as_uint(t0)) << 32) | task);
            }
        }
        (void)wave_fetch_add(ctl + kBeamDone, 1u);
    }

    // ---- phase 2: trace tickets
    __builtin_amdgcn_s_setprio(0);
    uint4* stk = lds_stack + lane;
    for (;;) {
        {   // nothing left for a new ticket?  (loads, so that the waves of a part do not each spend an atomic to find out)
            const uint32_t done = __builtin_amdgcn_readfirstlane(ld_agent(ctl + kBeamDone));
            if (done >= n_beam) {
                const uint32_t total = __builtin_amdgcn_readfirstlane(ld_agent(ctl + kReserved));
                if (__builtin_amdgcn_readfirstlane(ld_agent(ctl + kHead)) >= total) break;
            }
        }
        // a ticket is good for Q.chunk consecutive entries (sub-tiles of one beam tile lie together): fewer same-address atomics
        const uint32_t first = wave_fetch_add(ctl + kHead, Q.chunk);
        bool ended = false;
        for (uint32_t i = first; i < first + Q.chunk && !ended; ++i) {
            if (i >= Q.part_capacity) { ended = true; break; }            // more tickets than tasks can exist
            unsigned long long e = kNoTask;
            for (uint32_t polls = 0;; ++polls) {
                if (polls >= kFramePollBudget) {                          // never reached in a working system: give up rather than hang
                    (void)wave_fetch_add(ctl + kStalled, 1u);
                    ended = true;
                    break;
                }
                // every lane loads the same address: one broadcast request, and no lane-0 branch for the compiler to thread
                e = ld_agent(entries + i);
                const uint32_t e_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(e)), e_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(e >> 32));
                e = (static_cast<unsigned long long>(e_hi) << 32) | e_lo;
                if (e != kNoTask) break;
                if ((polls & 3u) == 3u) {
                    const uint32_t done = __builtin_amdgcn_readfirstlane(ld_agent(ctl + kBeamDone));
                    // a wave adds to kReserved before it issues its kBeamDone increment (the add has returned its value by then), so
                    // once done == n_beam the queue length read AFTER it is final, and an entry below that length is on its way
                    if (done >= n_beam) {
                        const uint32_t total = __builtin_amdgcn_readfirstlane(ld_agent(ctl + kReserved));
                        if (i >= total) { ended = true; break; }
                    }
                }
                __builtin_amdgcn_s_sleep(32);                             // ~2000 clocks between polls of this wave
            }
            if (ended) break;
            __hip_atomic_store(entries + i, kNoTask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every lane, same word: the slot is clean for the next launch
            const uint32_t task = static_cast<uint32_t>(e);
            const float t0 = __uint_as_float(static_cast<uint32_t>(e >> 32));
            const uint32_t per_beam = subs_x * subs_y;
            const uint32_t b = task / per_beam, sub = task % per_beam;
            const BeamRect r = beam_rect<MODE>(A, b);
            const uint32_t x = r.px + (sub % subs_x) * kWaveW + lane % kWaveW, y = r.py + (sub / subs_x) * kWaveH + lane / kWaveW;
            const size_t out_index = r.obase + static_cast<size_t>(y - r.oy) * r.stride + (x - r.ox);
            const Sink sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr};
            if (x < r.px_end && y < r.py_end) {
                if (MODE == RayMode::Tiles && !(x < A.frame_w && y < A.frame_h)) {
                    write_miss(sink);                                     // the tile buffer is dense
                } else {
                    RayIn ray = primary_ray(A, x, y);
                    ray.tmin = fmaxf(ray.tmin, t0);
                    trace_one(A, ray, stk, sink);
                }
            }
        }
        if (ended) break;
    }

    // ---- exit: the last wave of the part re-arms its counters
    const uint32_t part_waves = (gridDim.x - part + n_parts - 1u) / n_parts;
    const uint32_t gone = wave_fetch_add(ctl + kExited, 1u);
    if (gone == part_waves - 1u && lane == 0) {
        st_agent(ctl + kBeamNext, 0u); st_agent(ctl + kBeamDone, 0u); st_agent(ctl + kReserved, 0u);
        st_agent(ctl + kHead, 0u); st_agent(ctl + kExited, 0u);
    }
}

__global__ __launch_bounds__(64) void sun_map_kernel(const SunMapArgs a) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= a.sub_nu * a.sub_nv) return;
    const uint32_t iu = a.iu0 + b % a.sub_nu, iv = a.iv0 + b / a.sub_nu;
    if (iu >= a.nu || iv >= a.nv) return;
    const SunFrame f{{a.u[0], a.u[1], a.u[2]}, {a.v[0], a.v[1], a.v[2]}, {a.s[0], a.s[1], a.s[2]}};
    const float u_lo = a.u0 + static_cast<float>(iu) * a.texel, v_lo = a.v0 + static_cast<float>(iv) * a.texel;
    const float last = prism_far(a.trace, f, u_lo, u_lo + a.texel, v_lo, v_lo + a.texel, lane);
    if (lane == 0) a.map[static_cast<size_t>(iv) * a.nu + iu] = last;
}

// After an edit that may have FILLED voxels inside a box: every texel whose prism can meet the box is raised to the box's farthest corner
// along the sun — an upper bound of whatever the edit put there.  The map is conservative by construction (a larger "last occluder" only caps
// a shadow ray's tmax later: path_core.h), so this keeps every answer; an edit that only EMPTIES voxels needs nothing at all.
__global__ __launch_bounds__(64) void sun_map_raise_kernel(const SunMapArgs a, const float far_depth) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= a.sub_nu * a.sub_nv) return;
    const uint32_t iu = a.iu0 + i % a.sub_nu, iv = a.iv0 + i / a.sub_nu;
    if (iu >= a.nu || iv >= a.nv) return;
    float* t = a.map + static_cast<size_t>(iv) * a.nu + iu;
    *t = fmaxf(*t, far_depth);
}

// raygen.rgen main(): one lane per pixel of the rectangle, same 16x16 / 8x8 pixel mapping as the trace kernel.
#ifndef BLOK_PATH_WAVES
// waves per SIMD the path kernel is compiled for (register budget 512 / waves).  Round 4 sweep, 4K 64 spp, poses A / B (profiles/r04_paths_start_ab.txt): 4 waves (115 VGPRs,
// no spill) 50.9 ms, 5 (96, 12 spilled) 48.7, 6 (80, 28) 44.2-44.6 / 96.5, 7 (72, 64) 43.8-44.5 / 95.8, 8 (64, 104) 44.3-44.8 / 94.7: occupancy beats spill-freedom — the
// spills sit in the state machine around the walk, none in the walk loop.  With the bounce rounds' tail pool (path_core.h) the state machine is larger: at 7 waves two
// scratch loads land INSIDE the walk loop (37.2 / 82.6 ms), at 6 (80 VGPRs, 49 spilled) the loop is clean: 29.6 / 67.6; 5: 30.8 / 70.6 (profiles/r04_tail_pool_ab.txt)
#define BLOK_PATH_WAVES 6
#endif
// kResume: PathArgs::resume_secondary honoured (two kernels, so that the default — off, it measures slower — carries none of its state).
template <bool kResume>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(BLOK_PATH_WAVES, BLOK_PATH_WAVES))) void path_kernel(const PathArgs P) {
    // [levels - 1][kBlock] uint4: the walk's stack; then, with PathArgs::resume_secondary, the side area of the pixels' anchors (path_core.h):
    // [levels - 2][kBlock] uint2 and [levels - 2][kBlock] words
    extern __shared__ uint4 lds_stack[];
    const TraceArgs& A = P.trace;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t lx = (wave & 1u) * kWaveW + (lane % kWaveW);
    const uint32_t ly = (wave >> 1) * kWaveH + (lane / kWaveW);
    const uint32_t bx_count = (A.w + kTileW - 1u) / kTileW, by_count = (A.h + kTileH - 1u) / kTileH;
    uint32_t bx, by;
    if (!block_to_tile(blockIdx.x, gridDim.x, bx_count, by_count, bx, by)) return;
    const uint32_t rx = bx * kTileW + lx, ry = by * kTileH + ly;
    float t0 = 0.0f;
    if (A.beam) {
        const uint32_t wx = rx - lane % kWaveW, wy = ry - lane / kWaveW;          // the wave's 8x8 pixels start here (wave-uniform, inside the rectangle)
        t0 = A.beam[__builtin_amdgcn_readfirstlane((wy / A.beam_tile) * A.beam_bx + wx / A.beam_tile)];
        // The wave tile's OWN start parameter: the same cooperative search over its 8x8 pixels (all 64 lanes, before any leaves), looking
        // only beyond what the beam tile found.  For one primary ray per pixel it costs what it saves (fine bounds);
        // here every pixel sends spp primary rays through it.  Exact for the same reason as the beam tile's (beam.h): the frustum is
        // the tile grown by a pixel, the sub-pixel jitter stays inside.
        // (from 8 samples per pixel on: 2 spp 1.68 -> 1.82 ms with it, 8 spp 6.06 -> 5.90 (pose A) / 12.51 -> 12.61 (B), 64 spp 46.7 -> 44.5)
        if (P.fine_beam != 0u && t0 < kBeamNone && P.spp >= 8u) {
            const uint32_t fx0 = A.x0 + wx, fy0 = A.y0 + wy;
            const float fine = beam_start(A, static_cast<float>(fx0), static_cast<float>(fy0), static_cast<float>(min(fx0 + kWaveW, A.x0 + A.w)),
                                          static_cast<float>(min(fy0 + kWaveH, A.y0 + A.h)), lane);
            t0 = fmaxf(t0, fine);                                                   // (kBeamNone: no ray of this wave tile can hit anything)
        }
    }
    if (rx >= A.w || ry >= A.h) return;
    uint2* keep_lohi = nullptr; uint32_t* keep_base = nullptr;
    if (kResume && P.resume_secondary != 0u && A.levels >= 2u) {
        const uint32_t slots = A.levels - 1u, kept = A.levels - 2u;
        keep_lohi = reinterpret_cast<uint2*>(lds_stack + slots * kBlock) + tid;
        keep_base = reinterpret_cast<uint32_t*>(reinterpret_cast<uint2*>(lds_stack + slots * kBlock) + kept * kBlock) + tid;
    }
    // the bounce rounds' tail pool (path_core.h): this wave's records in global memory, its answers and two counters in LDS behind the stack
    // (and the side area); one wave per workgroup
    TailRecord* pool = nullptr; TailAnswer* tail_results = nullptr;
    if (kBlock == 64u && P.tail_pool != nullptr) {
        pool = P.tail_pool + static_cast<size_t>(blockIdx.x) * kTailCapacity;
        const uint32_t slots = A.levels > 1u ? A.levels - 1u : 1u;
        size_t words4 = static_cast<size_t>(slots) * kBlock;                                       // uint4 units
        if (kResume && P.resume_secondary != 0u && A.levels >= 2u) words4 += (static_cast<size_t>(A.levels - 2u) * kBlock * 12u + 15u) / 16u;
        tail_results = reinterpret_cast<TailAnswer*>(lds_stack + words4);
    }
    if (t0 >= kBeamNone && P.max_bounces != 0u) shade_pixel<kResume, true>(P, A.x0 + rx, A.y0 + ry, static_cast<size_t>(ry) * A.w + rx, lds_stack + tid, t0);      // (wave-uniform)
    else shade_pixel<kResume>(P, A.x0 + rx, A.y0 + ry, static_cast<size_t>(ry) * A.w + rx, lds_stack + tid, t0, keep_lohi, keep_base, pool, tail_results);
}

__global__ __launch_bounds__(256) void tonemap_kernel(const TonemapArgs T) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < T.n) T.ldr[i] = tonemap_pixel(T, i);
}

__global__ __launch_bounds__(256) void accumulate_kernel(const AccumArgs T) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < T.n) accumulate_pixel(T, i);
}

template <typename Elem>
__global__ __launch_bounds__(256) void untile_kernel(const UntileArgs U) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, n_px = U.frame_w * U.frame_h;
    if (i >= n_px) return;
    const uint32_t x = i % U.frame_w, y = i / U.frame_w;
    const uint32_t g = (y / U.tile) * U.tiles_x + x / U.tile;
    const uint32_t rank = g % U.n_ranks, local = g / U.n_ranks;
    const size_t src = (static_cast<size_t>(rank) * U.tiles_per_rank_max + local) * U.tile * U.tile +
                       static_cast<size_t>(y % U.tile) * U.tile + (x % U.tile);
    static_cast<Elem*>(U.frame)[static_cast<size_t>(blockIdx.y) * n_px + i] = static_cast<const Elem*>(U.gathered)[blockIdx.y * U.gathered_frame_stride + src];
}

// ---- sparse framebuffer exchange (multi-GPU): only the tiles with at least one pixel that is not sky travel --------------
// No reference counterpart (blok is single-GPU).  compact: one wave per (tile of the rank's dense RGBA8 tile
// buffer, frame); a tile with a non-sky pixel takes the next record slot {local tile index, tile^2 pixels} of its frame
// (trace_kernels.h: CompactArgs for the layout).  scatter (root): a map kernel notes which record holds which frame tile, then one
// wave per (frame tile, frame) writes the tile — its record, or sky — so that no pixel is written twice and, with a tile-state
// array, a sky tile that stays sky is not written at all (the root's assembly is on the critical path of every frame at N ranks).
__global__ __launch_bounds__(64) void compact_tiles_kernel(const CompactArgs a) {
    const uint32_t b = blockIdx.x, f = blockIdx.y, lane = threadIdx.x, px = a.tile * a.tile;
    if (b >= a.n_tiles) return;
    const uint32_t* src = a.tiles + f * a.tiles_frame_stride + static_cast<size_t>(b) * px;
    bool live = false;
    for (uint32_t i = lane; i < px; i += 64u) live |= src[i] != kSkyRgba;
    if (__ballot(live) == 0ull) return;
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(a.out + f, 1u);
    slot = __builtin_amdgcn_readfirstlane(slot);
    uint32_t* rec = a.out + a.n_frames + (static_cast<size_t>(slot) * a.n_frames + f) * (1u + px);
    if (lane == 0) rec[0] = b;
    for (uint32_t i = lane; i < px; i += 64u) rec[1u + i] = src[i];
}

constexpr uint32_t kCodeSky = 0xFFFFu;
__device__ __forceinline__ uint32_t pixel_code(const uint4 hit, uint32_t n_materials) {          // hit: one blok_hit as four words
    if ((hit.w >> 24) == 0u) return kCodeSky;
    return (hit.y < n_materials ? hit.y : n_materials) * 8u + ((hit.w >> 16) & 7u);                 // ids beyond the table share its "out of table" colour
}
__device__ __forceinline__ uint32_t code_rgba(uint32_t code, const blok_material* table, uint32_t n_materials) {
    return code == kCodeSky ? kSkyRgba : shade_rgba(table, n_materials, code >> 3, code & 7u);
}

__global__ __launch_bounds__(64) void compact_hit_tiles_kernel(const CompactHitArgs a) {
    const uint32_t b = blockIdx.x, f = blockIdx.y, lane = threadIdx.x, px = a.tile * a.tile;
    if (b >= a.n_tiles) return;
    const uint4* src = reinterpret_cast<const uint4*>(a.hits) + f * a.hits_frame_stride + static_cast<size_t>(b) * px;
    bool live = false;
    for (uint32_t i = lane; i < px; i += 64u) live |= (src[i].w >> 24) != 0u;
    if (__ballot(live) == 0ull) return;
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(a.out + f, 1u);
    slot = __builtin_amdgcn_readfirstlane(slot);
    uint32_t* rec = a.out + a.n_frames + (static_cast<size_t>(slot) * a.n_frames + f) * (1u + px / 2u);
    if (lane == 0) rec[0] = b;
    for (uint32_t i = lane; i < px / 2u; i += 64u)                      // two pixels per word, the even one in the low half
        rec[1u + i] = pixel_code(src[2u * i], a.n_materials) | (pixel_code(src[2u * i + 1u], a.n_materials) << 16);
}

// root, step 1: one thread per (rank, record slot, frame): which record holds which frame tile
__global__ __launch_bounds__(64) void scatter_map_kernel(const ScatterArgs a) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x, f = blockIdx.y;
    if (i >= a.n_ranks * a.max_records) return;
    const uint32_t r = i / a.max_records, j = i % a.max_records;
    const uint32_t* base = a.rank_ptrs ? a.rank_ptrs[r] : a.gathered + static_cast<size_t>(r) * a.rank_stride;
    if (j >= base[f]) return;
    const uint32_t* rec = base + a.n_frames + (static_cast<size_t>(j) * a.n_frames + f) * a.record_words;
    const uint64_t g = r + static_cast<uint64_t>(rec[0]) * a.n_ranks;
    if (g < a.tiles_total) a.tile_map[static_cast<size_t>(f) * a.tiles_total + g] = i + 1u;
}

// root, step 2: one wave per (frame tile, frame): the tile's record, or sky; every pixel of the frame is written at most once
template <bool kCodes>
__global__ __launch_bounds__(64) void scatter_tiles_kernel(const ScatterArgs a) {
    const uint32_t g = blockIdx.x, f = blockIdx.y, lane = threadIdx.x, px = a.tile * a.tile;
    const size_t slot = static_cast<size_t>(f) * a.tiles_total + g;
    const uint32_t m = a.tile_map[slot];                       // all lanes read it before lane 0 puts the word back to 0
    uint8_t* state = a.tile_state ? a.tile_state + slot : nullptr;
    const bool was_live = state ? *state != 0 : true;
    if (m == 0u && !was_live) return;                          // sky before, sky now
    if (lane == 0) { if (m) a.tile_map[slot] = 0u; if (state) *state = m ? 1 : 0; }
    uint32_t* frame = a.frame + static_cast<size_t>(f) * a.frame_w * a.frame_h;
    const uint32_t x0 = (g % a.tiles_x) * a.tile, y0 = (g / a.tiles_x) * a.tile;
    const uint32_t* rec = nullptr;
    if (m) {
        const uint32_t r = (m - 1u) / a.max_records, j = (m - 1u) % a.max_records;
        rec = (a.rank_ptrs ? a.rank_ptrs[r] : a.gathered + static_cast<size_t>(r) * a.rank_stride) + a.n_frames + (static_cast<size_t>(j) * a.n_frames + f) * a.record_words;
    }
    for (uint32_t i = lane; i < px; i += 64u) {
        const uint32_t x = x0 + i % a.tile, y = y0 + i / a.tile;
        uint32_t v = kSkyRgba;
        if (rec) {
            if constexpr (kCodes) v = code_rgba((rec[1u + i / 2u] >> ((i & 1u) * 16u)) & 0xFFFFu, a.mat_table, a.n_materials);
            else v = rec[1u + i];
        }
        if (x < a.frame_w && y < a.frame_h) frame[static_cast<size_t>(y) * a.frame_w + x] = v;
    }
}

}  // namespace

uint32_t sky_rgba() { return kSkyRgba; }

size_t frame_lds_bytes(const TraceArgs& args);

void launch_compact_tiles(const CompactArgs& args, hipStream_t stream) {
    if (args.n_tiles && args.n_frames) hipLaunchKernelGGL(compact_tiles_kernel, dim3(args.n_tiles, args.n_frames), dim3(64), 0, stream, args);
}

void launch_compact_hit_tiles(const CompactHitArgs& args, hipStream_t stream) {
    if (args.n_tiles && args.n_frames) hipLaunchKernelGGL(compact_hit_tiles_kernel, dim3(args.n_tiles, args.n_frames), dim3(64), 0, stream, args);
}

void launch_scatter_tiles(const ScatterArgs& args, bool codes, hipStream_t stream) {
    if (!args.n_frames || !args.tiles_total) return;
    const uint32_t n = args.n_ranks * args.max_records;
    if (n) hipLaunchKernelGGL(scatter_map_kernel, dim3((n + 63u) / 64u, args.n_frames), dim3(64), 0, stream, args);
    if (codes) hipLaunchKernelGGL(scatter_tiles_kernel<true>, dim3(args.tiles_total, args.n_frames), dim3(64), 0, stream, args);
    else hipLaunchKernelGGL(scatter_tiles_kernel<false>, dim3(args.tiles_total, args.n_frames), dim3(64), 0, stream, args);
}

// walk_blocks_per_frame: the walk's workgroups per frame when it runs over the prefix of an order (TraceArgs::rank_of), else blocks_per_frame
void launch_tile_frames(const TraceArgs& args, const TileFrames& frames, hipStream_t stream, uint32_t walk_blocks_per_frame) {
    if (frames.n_frames == 0 || frames.blocks_per_frame == 0) return;
    const size_t lds = static_cast<size_t>(args.levels > 1 ? args.levels - 1 : 1) * kBlock * sizeof(uint4);
    if (args.beam && frames.beams_per_frame)
        hipLaunchKernelGGL(beam_frames_kernel, dim3(frames.n_frames * frames.beams_per_frame), dim3(64), args.rank_of ? lds : 0, stream, args, frames);
    TileFrames walk = frames;
    if (args.rank_of) walk.blocks_per_frame = walk_blocks_per_frame;          // frame f = workgroups [f * prefix, (f + 1) * prefix)
    if (walk.blocks_per_frame) hipLaunchKernelGGL(trace_frames_kernel, dim3(walk.n_frames * walk.blocks_per_frame), dim3(kBlock), lds, stream, args, walk);
}

void launch_beam_frames(const TraceArgs& args, const TileFrames& frames, hipStream_t stream) {
    if (args.beam_tile && frames.n_frames && frames.beams_per_frame)
        hipLaunchKernelGGL(beam_frames_kernel, dim3(frames.n_frames * frames.beams_per_frame), dim3(64), 0, stream, args, frames);
}

void launch_trace(RayMode mode, const TraceArgs& args, uint32_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0) return;
    const size_t lds = static_cast<size_t>(args.levels > 1 ? args.levels - 1 : 1) * kBlock * sizeof(uint4);
    switch (mode) {
        case RayMode::Rect:  hipLaunchKernelGGL(trace_kernel<RayMode::Rect>,  dim3(n_blocks), dim3(kBlock), lds, stream, args); break;
        case RayMode::Tiles: hipLaunchKernelGGL(trace_kernel<RayMode::Tiles>, dim3(n_blocks), dim3(kBlock), lds, stream, args); break;
        case RayMode::Rays:  hipLaunchKernelGGL(trace_kernel<RayMode::Rays>,  dim3(n_blocks), dim3(kBlock), lds, stream, args); break;
    }
}

void launch_joint(RayMode mode, const TraceArgs& args, uint32_t n_beam_tiles, uint32_t n_blocks, hipStream_t stream) {
    if (n_beam_tiles + n_blocks == 0 || mode == RayMode::Rays) return;
    const size_t lds = static_cast<size_t>(args.levels > 1 ? args.levels - 1 : 1) * kBlock * sizeof(uint4);
    if (mode == RayMode::Rect) hipLaunchKernelGGL(joint_kernel<RayMode::Rect>, dim3(n_beam_tiles + n_blocks), dim3(kBlock), lds, stream, args, n_beam_tiles);
    else hipLaunchKernelGGL(joint_kernel<RayMode::Tiles>, dim3(n_beam_tiles + n_blocks), dim3(kBlock), lds, stream, args, n_beam_tiles);
}

void launch_list_joint(RayMode mode, const TraceArgs& args, const TileFrames* frames, uint32_t n_beam_tiles, uint32_t n_walkers, hipStream_t stream) {
    if (n_beam_tiles == 0 || mode == RayMode::Rays || !args.list.entries) return;
    const size_t lds = frame_lds_bytes(args);
    const dim3 grid(n_beam_tiles + n_walkers), clean(kListSegments * 8u);
    const TileFrames none{};
    if (mode == RayMode::Rect) {
        hipLaunchKernelGGL((list_joint_kernel<RayMode::Rect, false>), grid, dim3(kBlock), lds, stream, args, none);
        hipLaunchKernelGGL((list_cleanup_kernel<RayMode::Rect, false>), clean, dim3(kBlock), lds, stream, args, none);
    } else if (!frames) {
        hipLaunchKernelGGL((list_joint_kernel<RayMode::Tiles, false>), grid, dim3(kBlock), lds, stream, args, none);
        hipLaunchKernelGGL((list_cleanup_kernel<RayMode::Tiles, false>), clean, dim3(kBlock), lds, stream, args, none);
    } else {
        hipLaunchKernelGGL((list_joint_kernel<RayMode::Tiles, true>), grid, dim3(kBlock), lds, stream, args, *frames);
        hipLaunchKernelGGL((list_cleanup_kernel<RayMode::Tiles, true>), clean, dim3(kBlock), lds, stream, args, *frames);
    }
}

void launch_list_walk(RayMode mode, const TraceArgs& args, const TileFrames* frames, uint32_t n_walkers, hipStream_t stream) {
    if (n_walkers == 0 || mode == RayMode::Rays || !args.list.entries) return;
    const size_t lds = frame_lds_bytes(args);
    const TileFrames none{};
    if (mode == RayMode::Rect) hipLaunchKernelGGL((list_walk_kernel<RayMode::Rect, false>), dim3(n_walkers), dim3(kBlock), lds, stream, args, none);
    else if (!frames) hipLaunchKernelGGL((list_walk_kernel<RayMode::Tiles, false>), dim3(n_walkers), dim3(kBlock), lds, stream, args, none);
    else hipLaunchKernelGGL((list_walk_kernel<RayMode::Tiles, true>), dim3(n_walkers), dim3(kBlock), lds, stream, args, *frames);
}

uint32_t beam_tiles(RayMode mode, const TraceArgs& a, uint32_t tiles_of_rank) {
    if (mode == RayMode::Rect) return ((a.w + a.beam_tile - 1u) / a.beam_tile) * ((a.h + a.beam_tile - 1u) / a.beam_tile);
    return tiles_of_rank * (a.tile / a.beam_tile) * (a.tile / a.beam_tile);
}

void launch_beam(RayMode mode, const TraceArgs& args, uint32_t n_beam_tiles, hipStream_t stream) {
    if (n_beam_tiles == 0 || mode == RayMode::Rays) return;
    // The search itself keeps its stack in registers (beam.h) and uses no LDS; the walk's stack is needed only when the search waves may
    // walk wave tiles themselves (rank_of: a launch over a prefix of the order) — the ONLY use of lds_stack in beam_block, guarded by the
    // same rank_of test there, so LDS is never indexed without having been sized here.
    const size_t lds = args.rank_of ? frame_lds_bytes(args) : 0;
    if (mode == RayMode::Rect) hipLaunchKernelGGL(beam_kernel<RayMode::Rect>, dim3(n_beam_tiles), dim3(64), lds, stream, args, n_beam_tiles);
    else hipLaunchKernelGGL(beam_kernel<RayMode::Tiles>, dim3(n_beam_tiles), dim3(64), lds, stream, args, n_beam_tiles);
}

size_t frame_lds_bytes(const TraceArgs& args) { return static_cast<size_t>(args.levels > 1 ? args.levels - 1 : 1) * kBlock * sizeof(uint4); }

int frame_blocks_per_cu(RayMode mode, const TraceArgs& args) {
    int n = 0;
    const hipError_t e = mode == RayMode::Rect
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, frame_kernel<RayMode::Rect>, kBlock, frame_lds_bytes(args))
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, frame_kernel<RayMode::Tiles>, kBlock, frame_lds_bytes(args));
    return e == hipSuccess ? n : 0;
}

void launch_frame(RayMode mode, const TraceArgs& args, const FrameQueue& queue, uint32_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0 || mode == RayMode::Rays) return;
    const size_t lds = frame_lds_bytes(args);
    if (mode == RayMode::Rect) hipLaunchKernelGGL(frame_kernel<RayMode::Rect>, dim3(n_blocks), dim3(kBlock), lds, stream, args, queue);
    else hipLaunchKernelGGL(frame_kernel<RayMode::Tiles>, dim3(n_blocks), dim3(kBlock), lds, stream, args, queue);
}

void launch_sun_map(const SunMapArgs& args, hipStream_t stream) {
    const uint32_t n = args.sub_nu * args.sub_nv;
    if (n) hipLaunchKernelGGL(sun_map_kernel, dim3(n), dim3(64), 0, stream, args);
}

void launch_sun_map_raise(const SunMapArgs& args, float far_depth, hipStream_t stream) {
    const uint32_t n = args.sub_nu * args.sub_nv;
    if (n) hipLaunchKernelGGL(sun_map_raise_kernel, dim3((n + 63u) / 64u), dim3(64), 0, stream, args, far_depth);
}

void launch_paths(const PathArgs& args, uint32_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0) return;
    const uint32_t levels = args.trace.levels;
    size_t lds = static_cast<size_t>(levels > 1 ? levels - 1 : 1) * kBlock * sizeof(uint4);
    const size_t tail_lds = args.tail_pool ? (kTailBatch + 1u) * sizeof(TailAnswer) : 0u;               // the tail pool's answers and counters
    if (args.resume_secondary && levels >= 2) {
        lds += (static_cast<size_t>(levels - 2) * kBlock * (sizeof(uint2) + sizeof(uint32_t)) + 15u) / 16u * 16u;      // the anchors' side area
        hipLaunchKernelGGL(path_kernel<true>, dim3(n_blocks), dim3(kBlock), lds + tail_lds, stream, args);
    } else hipLaunchKernelGGL(path_kernel<false>, dim3(n_blocks), dim3(kBlock), lds + tail_lds, stream, args);
}

void launch_tonemap(const TonemapArgs& args, hipStream_t stream) {
    if (args.n) hipLaunchKernelGGL(tonemap_kernel, dim3((args.n + 255u) / 256u), dim3(256), 0, stream, args);
}

void launch_accumulate(const AccumArgs& args, hipStream_t stream) {
    if (args.n) hipLaunchKernelGGL(accumulate_kernel, dim3((args.n + 255u) / 256u), dim3(256), 0, stream, args);
}

void launch_untile(const UntileArgs& args, uint32_t n_frames, hipStream_t stream) {
    const uint32_t n = args.frame_w * args.frame_h;
    if (!n || !n_frames) return;
    if (args.elem_bytes == 16) hipLaunchKernelGGL(untile_kernel<uint4>, dim3((n + 255u) / 256u, n_frames), dim3(256), 0, stream, args);
    else hipLaunchKernelGGL(untile_kernel<uint32_t>, dim3((n + 255u) / 256u, n_frames), dim3(256), 0, stream, args);
}

}  // namespace blok
