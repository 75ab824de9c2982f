// How a frame is launched (include/blok_hip.h: the trace entries; launch_policy.h: the decisions): who else is on the device, the
// orders cached per view and what keeps them (markers, sorts, shifts), the joint launch's slots, the list forms' buffers, and
// launch_timed, through which every rectangle / tile launch goes.  Split from api.hip in round 4.
#include "api_internal.h"
#include <cstdlib>

namespace blok_api {

// ---- who else is using the device -----------------------------------------------------------------------------------------------
// A joint launch only pays when it has the device to itself (launch_policy.h), and "itself" is a property of the DEVICE, not of a
// context: two contexts on one device (two HipTracers; blok_hip_multi_* with an ordinal listed twice) must see each other's frames.
// So every frame launch leaves an event behind on its stream in a process-wide table per device, and a launch asks whether any OTHER
// (context, stream) of the device still has one pending.
namespace {
// Two markers per (context, stream), both recorded by the stream's own launches only (nothing here ever records on a stream other than
// the one being launched on: a caller may have destroyed it, or be capturing it into a graph): `last` = behind its latest frame launch;
// `held` = what `last` was when the context last asked for its launches to be remembered (hold_markers: the adoption of an order, a change
// of launch geometry) — a marker that stays put while the stream goes on launching, which is what a later sort waits for.
struct StreamMarks { hipEvent_t last = nullptr, held = nullptr; bool fresh = false; };
struct DeviceActivity {
    std::mutex lock;
    std::map<std::pair<const blok_hip_ctx*, hipStream_t>, StreamMarks> marks;
};
DeviceActivity& device_activity(int device) {
    static std::mutex table_lock;
    static std::map<int, DeviceActivity> table;
    std::lock_guard<std::mutex> g(table_lock);
    return table[device];
}
}  // namespace

static bool device_busy_elsewhere(const blok_hip_ctx* ctx, hipStream_t stream) {
    DeviceActivity& act = device_activity(ctx->device);
    std::lock_guard<std::mutex> g(act.lock);
    bool busy = false;
    for (auto& kv : act.marks) {
        if (kv.first.first == ctx && kv.first.second == stream) continue;
        for (hipEvent_t ev : {kv.second.last, kv.second.held})
            if (ev && hipEventQuery(ev) == hipErrorNotReady) { busy = true; break; }
        if (busy) break;
    }
    (void)hipGetLastError();                               // hipErrorNotReady is an answer, not a failure
    return busy;
}

static int note_frame_launch(blok_hip_ctx* ctx, hipStream_t stream) {
    DeviceActivity& act = device_activity(ctx->device);
    std::lock_guard<std::mutex> g(act.lock);
    StreamMarks& m = act.marks[{ctx, stream}];
    if (!m.last) BLOK_HIP_TRY(ctx, hipEventCreateWithFlags(&m.last, hipEventDisableTiming));
    BLOK_HIP_TRY(ctx, hipEventRecord(m.last, stream));
    m.fresh = true;
    return BLOK_OK;
}

// Every launch of this context issued so far, on any of its streams, is in front of a `held` marker from here on.  No HIP call: the
// marker a stream recorded behind its latest launch changes places with the one it held before (which that stream's next launch re-records).
static void hold_markers(const blok_hip_ctx* ctx) {
    DeviceActivity& act = device_activity(ctx->device);
    std::lock_guard<std::mutex> g(act.lock);
    for (auto& kv : act.marks)
        if (kv.first.first == ctx && kv.second.fresh) { std::swap(kv.second.last, kv.second.held); kv.second.fresh = false; }
}

// `stream` waits for the held markers of the context's OTHER streams (its own earlier work is in front of it anyway).
static int wait_for_held_markers(blok_hip_ctx* ctx, hipStream_t stream) {
    DeviceActivity& act = device_activity(ctx->device);
    std::lock_guard<std::mutex> g(act.lock);
    for (auto& kv : act.marks)
        if (kv.first.first == ctx && kv.first.second != stream && kv.second.held) BLOK_HIP_TRY(ctx, hipStreamWaitEvent(stream, kv.second.held, 0));
    return BLOK_OK;
}

void forget_device_activity(const blok_hip_ctx* ctx, bool one_stream, hipStream_t stream) {
    DeviceActivity& act = device_activity(ctx->device);
    std::lock_guard<std::mutex> g(act.lock);
    for (auto it = act.marks.begin(); it != act.marks.end();)
        if (it->first.first == ctx && (!one_stream || it->first.second == stream)) {
            if (it->second.last) (void)hipEventDestroy(it->second.last);
            if (it->second.held) (void)hipEventDestroy(it->second.held);
            it = act.marks.erase(it);
        } else ++it;
}

// ---- longest-first order of the walk's wave tiles (api_internal.h: TileOrder; decisions in launch_policy.h: plan_order, plan_shift) ------
// Is it the same view?  An order sorted from a view's own clocks — and above all the set of tiles that walked in it, the only ones a prefix
// launch dispatches walk waves for — is that view's: a camera creeping by a quarter of a degree per frame (round 2's window) kept such
// an order in force while the silhouettes moved out from under it, and every tile that had become live was walked by its search wave,
// one after the other (measured: 0.32-0.37 ms per frame for a slow pan against 0.25 in row-major order).  So "at rest" means at rest: the
// basis within 1e-5 per component (0.0006 degree), the position within a thousandth of a voxel (and a millionth of its distance to the
// world), the same lens.  Whatever moves more is a camera in motion and gets the order made for that (launch_policy.h).
static bool camera_near(const blok_hip_ctx* ctx, const blok_camera& a, const blok_camera& b) {
    const float half = 0.5f * std::ldexp(1.0f, 2 * static_cast<int>(ctx->stats.levels)) * ctx->world_voxel_size;
    float d2 = 0.0f, r2 = 0.0f, turn = 0.0f;
    for (int k = 0; k < 3; ++k) {
        const float centre = static_cast<float>(ctx->stats.origin[k]) * ctx->world_voxel_size + half;
        d2 += (a.pos[k] - b.pos[k]) * (a.pos[k] - b.pos[k]);
        r2 += (a.pos[k] - centre) * (a.pos[k] - centre);
        turn = std::max(turn, std::max(std::fabs(a.fwd[k] - b.fwd[k]), std::max(std::fabs(a.right[k] - b.right[k]), std::fabs(a.up[k] - b.up[k]))));
    }
    const float still = 1.0e-3f * ctx->world_voxel_size;
    return turn <= 1.0e-5f && d2 <= std::max(still * still, 1.0e-12f * r2) && std::fabs(a.tan_half_fov - b.tan_half_fov) < 1e-6f && std::fabs(a.aspect - b.aspect) < 1e-6f;
}

void free_order(blok_hip_ctx* ctx) {
    auto& O = ctx->order;
    for (void* p : {static_cast<void*>(O.d_cost), static_cast<void*>(O.d_iota), static_cast<void*>(O.d_keys_in), static_cast<void*>(O.d_keys), O.d_class_scratch, O.d_temp})
        if (p) (void)hipFree(p);
    for (auto& sl : O.slots) {
        if (sl.d_order) (void)hipFree(sl.d_order);
        if (sl.d_rank_of) (void)hipFree(sl.d_rank_of);
        sl = blok_hip_ctx::TileOrder::Slot{};
    }
    O.d_cost = O.d_iota = O.d_keys_in = O.d_keys = nullptr; O.d_class_scratch = nullptr;
    O.d_temp = nullptr; O.capacity = 0;
}

// The order's buffers, for at least `blocks` wave tiles and never fewer than the full frame has: allocated by blok_hip_create / _resize, so
// that no *_device launch ever allocates or waits for the device on their account (the regrow below is what is left for a caller that
// resizes between launches: a device-wide wait, because frames in flight and a pending sort may use the old buffers).
int order_buffers(blok_hip_ctx* ctx, uint32_t blocks, hipStream_t stream) {
    auto& O = ctx->order;
    constexpr int kSlots = blok_hip_ctx::TileOrder::kSlots;
    const uint32_t want = std::max<uint32_t>(blocks, blok::rect_grid_blocks(ctx->width, ctx->height));
    if (O.capacity >= want) return BLOK_OK;
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    free_order(ctx);
    const size_t bytes = static_cast<size_t>(want) * sizeof(uint32_t);
    for (uint32_t** p : {&O.d_cost, &O.d_iota, &O.d_keys_in, &O.d_keys})
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(p), bytes));
    for (auto& sl : O.slots) {
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&sl.d_order), bytes));
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&sl.d_rank_of), bytes));
    }
    if (!O.h_live) BLOK_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&O.h_live), kSlots * sizeof(uint32_t), hipHostMallocDefault));
    if (!O.h_depth) BLOK_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&O.h_depth), kSlots * blok::kOrderDepthPartials * 3 * sizeof(float), hipHostMallocDefault));
    if (!O.h_fallback) { BLOK_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&O.h_fallback), sizeof(uint32_t), hipHostMallocDefault)); *O.h_fallback = 0u; }
    if (!O.d_fallback) BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&O.d_fallback), sizeof(uint32_t)));
    O.temp_bytes = blok::tile_order_temp_bytes(want);
    BLOK_HIP_TRY(ctx, hipMalloc(&O.d_temp, O.temp_bytes ? O.temp_bytes : 16));
    BLOK_HIP_TRY(ctx, hipMalloc(&O.d_class_scratch, blok::tile_order_class_sort_bytes_max(want)));
    if (!O.done) BLOK_HIP_TRY(ctx, hipEventCreateWithFlags(&O.done, hipEventDisableTiming));
    BLOK_HIP_TRY(ctx, blok::launch_iota(O.d_iota, want, stream));              // the identity, whatever the geometry: written once
    BLOK_HIP_TRY(ctx, hipMemsetAsync(O.d_cost, 0, bytes, stream));
    O.capacity = want;
    std::memset(O.key, 0xFF, sizeof(O.key));
    O.current = -1; O.chosen = -1; O.pending = false; O.orphan = false; O.n_recent = 0;
    return BLOK_OK;
}

// Before an orderable launch: adoption of a finished sort, the plan; fills args.order / rank_of / launched / cost_out.
// rect: a rectangle launch (else a rank's tiles: orders of a view's own clocks only — a whole-tile shift of the screen does not map a rank's
// round-robin share of the tiles onto itself); uniform_view: every camera of the launch is the same view (several frames per launch).
static int order_before_launch(blok_hip_ctx* ctx, blok::TraceArgs& args, uint32_t blocks, hipStream_t stream, bool alone, bool rect, bool uniform_view, blok::OrderPlan* plan) {
    auto& O = ctx->order;
    constexpr int kSlots = blok_hip_ctx::TileOrder::kSlots;
    const uint32_t key[7] = {rect ? args.x0 : (0x80000000u | args.tile), rect ? args.y0 : args.rank, rect ? args.w : args.n_ranks, rect ? args.h : 0u, ctx->width, ctx->height, ctx->world_version};
    if (O.capacity < blocks) { const int rc = order_buffers(ctx, blocks, stream); if (rc != BLOK_OK) return rc; }      // (never after create / resize: a rectangle has no more tiles than the frame)
    O.launch_serial += 1;
    if (std::memcmp(key, O.key, sizeof(key)) != 0) {
        // another launch geometry starts in natural order with no costs and no views; in stream order, nothing waits (a sort still pending
        // for the old geometry finishes into a slot nobody will adopt: its event is simply never asked again)
        BLOK_HIP_TRY(ctx, hipMemsetAsync(O.d_cost, 0, static_cast<size_t>(blocks) * sizeof(uint32_t), stream));
        std::memcpy(O.key, key, sizeof(key));
        O.orphan = O.orphan || O.pending;                                    // (a sort of the old geometry may still be writing an order buffer)
        // ... and frames in flight on other streams may still be READING the slots that were in use: the first sort of the new geometry,
        // whichever slot it targets, waits for the markers held from here (without this it waited for the markers of the last
        // adoption only, which cover the readers of the other buffer)
        hold_markers(ctx); O.hold_serial = O.launch_serial;
        for (auto& sl : O.slots) { sl.valid = false; sl.last_use = 0; }
        O.current = -1; O.pending = false; O.still_frames = 0; O.interval_now = O.interval; O.have_residual = false; O.n_recent = 0; O.revisit_streak = 0;
    }
    if (O.pending && hipEventQuery(O.done) == hipSuccess) {              // the sort launched some frames ago has finished
        auto& sl = O.slots[O.target];
        sl.valid = true; sl.dilated = O.pending_dilated; sl.radius = O.pending_radius; sl.cam = O.pending_cam;
        sl.live = O.h_live[O.target];                                    // written by the device before the event
        sl.frames_since_sort = 0; sl.interval_now = O.pending_interval; sl.last_use = 0;
        if (sl.dilated) {
            // the frame's depths: mean and standard deviation of its live beam tiles' inverse start parameters, from the sort's partial sums
            double cnt = 0.0, s1 = 0.0, s2 = 0.0;
            const float* part = O.h_depth + O.target * blok::kOrderDepthPartials * 3;
            for (uint32_t k = 0; k < blok::kOrderDepthPartials; ++k) { cnt += part[k * 3]; s1 += part[k * 3 + 1]; s2 += part[k * 3 + 2]; }
            const double mean = cnt > 0.0 ? s1 / cnt : 0.0, var = cnt > 0.0 ? s2 / cnt - mean * mean : 0.0;
            sl.inv_depth[0] = static_cast<float>(mean); sl.inv_depth[1] = static_cast<float>(var > 0.0 ? std::sqrt(var) : 0.0);
        }
        // an older order of the same view, and older orders made to be carried (only the latest is of use), make room
        for (int k = 0; k < kSlots; ++k)
            if (k != O.target && O.slots[k].valid && (O.slots[k].dilated || (!sl.dilated && camera_near(ctx, sl.cam, O.slots[k].cam)))) O.slots[k].valid = false;
        O.current = O.target; O.pending = false;
        // From here on the slots just retired are read by no new launch; the launches that may still be reading them are those already
        // issued, on any stream of this context.  The marker each stream left behind its latest launch is held from now: what a later sort —
        // which overwrites such a slot, many launches from now — has to wait for: long past by then, so the sort never holds up the frames in
        // flight (waiting for the streams' LATEST launches instead did: a bubble in the three-deep pipeline per sort, 5 % of a 20-frame run).
        hold_markers(ctx); O.hold_serial = O.launch_serial;
    }
    (void)hipGetLastError();                                             // hipErrorNotReady is an answer, not a failure
    // the order this launch may walk in: its own view's if the cache has one, else the one adopted last (a carried order, if it is dilated)
    int own = -1;
    for (int k = 0; k < kSlots && uniform_view; ++k)
        if (O.slots[k].valid && !O.slots[k].dilated && camera_near(ctx, args.cam, O.slots[k].cam) && (own < 0 || O.slots[k].last_use > O.slots[own].last_use)) own = k;
    const int chosen = own >= 0 ? own : O.current;
    // At rest: the previous launch's view — or, for a caller that alternates between fixed views, one of the few before it, once that has
    // happened four launches running (a camera that swings back and forth passes through a view of two launches ago at every turn: that is
    // motion, and keeps the moving camera's carried order)
    const bool rest = uniform_view && camera_near(ctx, args.cam, O.last_cam);
    bool revisit = false;
    for (uint32_t k = 0; k < O.n_recent && uniform_view && !rest && !revisit; ++k) revisit = camera_near(ctx, args.cam, O.recent[k]);
    O.revisit_streak = revisit ? O.revisit_streak + 1u : (rest ? O.revisit_streak : 0u);
    const bool seen = rest || (revisit && O.revisit_streak >= 4u);      // (a swing back through a view the camera rested in makes three in a row: four)
    blok::OrderFacts f{};
    f.enabled = true; f.have_order = chosen >= 0 && O.slots[chosen].valid;
    f.near_order_view = f.have_order && own >= 0;
    f.near_last_view = seen;
    f.sort_pending = O.pending; f.still_frames = O.still_frames;
    f.frames_since_sort = own >= 0 ? O.slots[own].frames_since_sort : 0u; f.interval = O.interval; f.interval_now = own >= 0 ? O.slots[own].interval_now : O.interval_now;
    f.moving_enabled = O.moving && rect; f.alone = alone; f.alone_before = O.alone_before; O.alone_before = alone;
    f.order_dilated = f.have_order && O.slots[chosen].dilated;
    blok::ShiftPlan shift{};
    if (f.order_dilated && f.moving_enabled && f.alone) {
        blok::ShiftFacts sf{};
        static_assert(sizeof(blok::PolicyCamera) == sizeof(blok_camera), "launch_policy.h: PolicyCamera is blok_camera");
        std::memcpy(&sf.then, &O.slots[chosen].cam, sizeof(blok_camera)); std::memcpy(&sf.now, &args.cam, sizeof(blok_camera));
        sf.inv_depth_mean = O.slots[chosen].inv_depth[0]; sf.inv_depth_sigma = O.slots[chosen].inv_depth[1];
        sf.frame_w = ctx->width; sf.frame_h = ctx->height;
        sf.tiles_x = (args.w + blok::kTileW - 1u) / blok::kTileW; sf.tiles_y = (args.h + blok::kTileH - 1u) / blok::kTileH; sf.tile_w = blok::kTileW; sf.tile_h = blok::kTileH;
        sf.radius = O.slots[chosen].radius;
        shift = blok::plan_shift(sf);
        f.shift_ok = shift.ok;
        // what the shift leaves over sizes the next dilation and the strips — also when it is too large to use, the next dilation grows with it —
        // but only when it was computed: a view plan_shift could not compare (another lens, a point behind the camera) says nothing, and
        // the next sort takes the default radius (a residual of 0 from an early exit gave the smallest radius after the largest change)
        O.have_residual = shift.measured; O.last_residual = shift.measured ? shift.residual : 0.0f;
    }
    *plan = blok::plan_order(f);
    O.still_frames = plan->still_frames; O.last_cam = args.cam;
    { for (uint32_t k = std::min<uint32_t>(O.n_recent, 3u); k > 0; --k) O.recent[k] = O.recent[k - 1]; O.recent[0] = args.cam; O.n_recent = std::min<uint32_t>(O.n_recent + 1u, 4u); }
    O.chosen = plan->use_order ? chosen : -1;
    args.order = plan->use_order ? O.slots[chosen].d_order : nullptr;
    args.cost_out = plan->measure ? O.d_cost : nullptr;
    args.order_sx = args.order_sy = 0u;
    O.last_use = plan->use_order ? (plan->shifted ? 2 : 1) : 0; O.last_sx = O.last_sy = 0u;
    if (args.order) {
        O.slots[chosen].last_use = O.launch_serial;
        args.rank_of = O.slots[chosen].d_rank_of; args.launched = O.slots[chosen].live;
        if (plan->shifted) { args.order_sx = O.last_sx = shift.sx; args.order_sy = O.last_sy = shift.sy; }
        if (O.debug_shift) {                                             // any shift of any order is a permutation: the frame must not change
            const uint32_t tiles_x = (args.w + blok::kTileW - 1u) / blok::kTileW, tiles_y = (args.h + blok::kTileH - 1u) / blok::kTileH;
            args.order_sx = O.last_sx = O.debug_sx % tiles_x; args.order_sy = O.last_sy = O.debug_sy % tiles_y;
        }
    }
    return BLOK_OK;
}

// After it: the sort, if the plan says so — on the LAUNCH stream, behind the frame (a stream of its own would be one HIP stream more than
// the hardware queues the frame streams and the null stream occupy: measured, that alone costs 18 % of the pipelined rate).
static int order_after_launch(blok_hip_ctx* ctx, const blok::TraceArgs& args, uint32_t blocks, uint32_t n_beams, hipStream_t stream, const blok::OrderPlan& plan) {
    auto& O = ctx->order;
    constexpr int kSlots = blok_hip_ctx::TileOrder::kSlots;
    if (O.chosen >= 0) { auto& sl = O.slots[O.chosen]; sl.frames_since_sort += 1; if (!sl.dilated) sl.interval_now = plan.next_interval_now; }
    else O.interval_now = plan.next_interval_now;
    if (!plan.start_sort) return BLOK_OK;
    // The slot to sort into: not the one this launch walks in; an empty one if there is one, else a retired-in-all-but-name dilated order, else
    // the view walked in longest ago.
    int target = -1;
    for (int pass = 0; pass < 3 && target < 0; ++pass)
        for (int k = 0; k < kSlots; ++k) {
            if (k == O.chosen) continue;
            const auto& sl = O.slots[k];
            const bool fits = pass == 0 ? !sl.valid : (pass == 1 ? (sl.dilated && k != O.current) : true);
            if (fits && (target < 0 || (pass == 2 && sl.last_use < O.slots[target].last_use))) { target = k; if (pass != 2) break; }
        }
    if (target < 0) return BLOK_OK;
    // Nothing still running may read the target: whatever walked in it was issued before the latest holding of the markers (an adoption, a
    // change of geometry) — or else they are held again now, behind the streams' latest launches (a cache with more views cycling than slots).
    // (>=: the launch during which the markers were held was itself issued behind them)
    if (O.slots[target].last_use >= O.hold_serial) { hold_markers(ctx); O.hold_serial = O.launch_serial; }
    O.slots[target].valid = false;
    if (O.current == target) O.current = -1;
    { const int rc = wait_for_held_markers(ctx, stream); if (rc != BLOK_OK) return rc; }
    // ... and nothing may still be WRITING it: a sort left behind by a change of launch geometry, possibly on another stream
    if (O.orphan) { BLOK_HIP_TRY(ctx, hipStreamWaitEvent(stream, O.done, 0)); O.orphan = false; }
    uint32_t radius = 0;
    auto& T = O.slots[target];
    if (plan.dilate) {
        // a camera in motion: a counting sort of the dilated clocks (three small launches; it reads the live cost buffer — any mixture of old
        // and new costs is as good a key, and what it sorts is its own copy), and the frame's depths go along for the next launch's shift
        radius = blok::plan_dilation(O.have_residual, O.last_residual);
        const uint32_t tiles_x = (args.w + blok::kTileW - 1u) / blok::kTileW, tiles_y = (args.h + blok::kTileH - 1u) / blok::kTileH;
        BLOK_HIP_TRY(ctx, blok::launch_tile_order_class_sort(O.d_cost, tiles_x, tiles_y, radius, O.d_class_scratch, T.d_order, T.d_rank_of, O.h_live + target,
                                                             args.beam, args.beam_slots, args.beam_serial, n_beams, O.h_depth + target * blok::kOrderDepthPartials * 3, stream));
    } else {
        // the sort reads a SNAPSHOT of the costs: frames in flight on other streams keep writing the live buffer, and a radix sort that saw a
        // key change between its histogram and its scatter would not produce a permutation
        BLOK_HIP_TRY(ctx, hipMemcpyAsync(O.d_keys_in, O.d_cost, static_cast<size_t>(blocks) * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        BLOK_HIP_TRY(ctx, blok::launch_tile_order_sort(O.d_keys_in, O.d_keys, O.d_iota, T.d_order, O.d_temp, O.temp_bytes, blocks, stream));
        BLOK_HIP_TRY(ctx, blok::launch_tile_order_finish(T.d_order, O.d_keys, blocks, T.d_rank_of, O.h_live + target, stream));
    }
    BLOK_HIP_TRY(ctx, hipEventRecord(O.done, stream));
    O.target = target; O.pending_cam = args.cam; O.pending_dilated = plan.dilate; O.pending_radius = radius; O.pending_interval = plan.next_interval_now; O.pending = true;
    return BLOK_OK;
}

// The stream's give-up counter (joint and list forms).
static int gave_up_counter(blok_hip_ctx* ctx, hipStream_t stream, uint32_t** out) {
    auto& slot = ctx->beam_buffers[stream];
    if (!slot.gave_up) {
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.gave_up), sizeof(uint32_t)));
        BLOK_HIP_TRY(ctx, hipMemsetAsync(slot.gave_up, 0, sizeof(uint32_t), stream));
    }
    *out = slot.gave_up;
    return BLOK_OK;
}

// The stream's published-result words of a joint launch: valid when they carry this launch's serial.
static int joint_slots(blok_hip_ctx* ctx, blok::TraceArgs& args, hipStream_t stream, uint32_t n_beams) {
    auto& slot = ctx->beam_buffers[stream];
    if (slot.n_slots < n_beams) {
        if (slot.slots) { BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(slot.slots); }
        slot.slots = nullptr; slot.n_slots = 0;
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.slots), n_beams * sizeof(unsigned long long)));
        BLOK_HIP_TRY(ctx, hipMemsetAsync(slot.slots, 0, n_beams * sizeof(unsigned long long), stream));
        slot.n_slots = n_beams; slot.serial = 0;
    }
    if (++slot.serial == 0u) slot.serial = 1u;
    args.beam_slots = slot.slots; args.beam_serial = slot.serial;
    return gave_up_counter(ctx, stream, &args.joint_gave_up);
}

// The stream's live list for a launch of n_searches search workgroups with per_search wave tiles each (trace_kernels.h: LiveList).
int live_list(blok_hip_ctx* ctx, blok::TraceArgs& args, hipStream_t stream, uint32_t n_searches, uint32_t per_search) {
    auto& slot = ctx->beam_buffers[stream];
    const size_t seg_capacity = static_cast<size_t>((n_searches + blok::kListSegments - 1u) / blok::kListSegments) * per_search;
    if (!slot.list_ctl) {
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.list_ctl), blok::kListSegments * blok::kListCtlWords * sizeof(unsigned long long)));
        BLOK_HIP_TRY(ctx, hipMemsetAsync(slot.list_ctl, 0, blok::kListSegments * blok::kListCtlWords * sizeof(unsigned long long), stream));
        BLOK_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&slot.list_hint), blok::kListSegments * blok::kListClasses * sizeof(uint32_t), hipHostMallocDefault));
        std::memset(slot.list_hint, 0, blok::kListSegments * blok::kListClasses * sizeof(uint32_t));      // read as a hint before the first launch has written it
        slot.list_hint_valid = false;
    }
    const bool wrapped = slot.list_serial + 1u >= (1u << blok::kListSerialBits);
    if (slot.list_capacity < seg_capacity || wrapped) {
        if (slot.list_capacity < seg_capacity) {
            if (slot.list_entries) { BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(slot.list_entries); }
            slot.list_entries = nullptr; slot.list_capacity = 0;
            BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.list_entries), seg_capacity * blok::kListSegments * blok::kListClasses * sizeof(unsigned long long)));
            slot.list_capacity = seg_capacity;
        }
        // every entry empty; also when the 20-bit serial starts over, so that no entry of a million launches ago can pass for a new one
        BLOK_HIP_TRY(ctx, hipMemsetAsync(slot.list_entries, 0, slot.list_capacity * blok::kListSegments * blok::kListClasses * sizeof(unsigned long long), stream));
        if (wrapped) slot.list_serial = 0;
    }
    slot.list_serial += 1u;
    args.list.entries = slot.list_entries; args.list.ctl = slot.list_ctl; args.list.hint = slot.list_hint;
    args.list.seg_capacity = static_cast<uint32_t>(slot.list_capacity);      // the buffer's stride (>= this launch's need)
    args.list.serial = slot.list_serial; args.list.n_searches = n_searches;
    return gave_up_counter(ctx, stream, &args.joint_gave_up);
}

// What the previous list launch on this stream left in pinned memory: per class its longest list (sizing hints, launch_policy.h).
static bool list_hint(const blok_hip_ctx* ctx, hipStream_t stream, uint32_t geometry_key, uint32_t per_class[blok::kListClasses]) {
    auto it = ctx->beam_buffers.find(stream);
    if (it == ctx->beam_buffers.end() || !it->second.list_hint || !it->second.list_hint_valid || it->second.list_hint_key != geometry_key) return false;
    for (uint32_t c = 0; c < blok::kListClasses; ++c) {
        per_class[c] = 0;
        for (uint32_t k = 0; k < blok::kListSegments; ++k) per_class[c] = std::max(per_class[c], it->second.list_hint[k * blok::kListClasses + c]);      // plain reads of words the device may be writing: hints
    }
    return true;
}

// The context's cost buffer for a rectangle launch (trace_kernels.h: cost classes): one word per wave tile of the launch geometry.
static int list_costs(blok_hip_ctx* ctx, blok::TraceArgs& args, uint32_t wave_tiles, hipStream_t stream) {
    const uint32_t key[6] = {args.x0, args.y0, args.w, args.h, ctx->width, ctx->height};
    if (ctx->list_cost_capacity < wave_tiles || std::memcmp(key, ctx->list_cost_key, sizeof(key)) != 0) {
        if (ctx->list_cost_capacity < wave_tiles) {
            if (ctx->d_list_cost) { BLOK_HIP_TRY(ctx, hipDeviceSynchronize()); (void)hipFree(ctx->d_list_cost); }      // frames in flight on other streams still write it
            ctx->d_list_cost = nullptr; ctx->list_cost_capacity = 0;
            BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_list_cost), static_cast<size_t>(wave_tiles) * sizeof(uint32_t)));
            ctx->list_cost_capacity = wave_tiles;
        }
        BLOK_HIP_TRY(ctx, hipMemsetAsync(ctx->d_list_cost, 0, ctx->list_cost_capacity * sizeof(uint32_t), stream));      // another geometry: nothing is known
        std::memcpy(ctx->list_cost_key, key, sizeof(key));
        ctx->list_has_prev_cam = false;
    }
    args.list.cost = ctx->d_list_cost;
    args.list.has_prev = ctx->list_has_prev_cam ? 1u : 0u;
    args.list.prev_cam = ctx->list_prev_cam;
    ctx->list_prev_cam = args.cam; ctx->list_has_prev_cam = true;        // the costs this launch leaves are those of this camera
    return BLOK_OK;
}

// Rect / Tiles launches run the beam pre-pass first, on the same stream (tiles_of_rank: Tiles only; frames: several frames of a rank's
// tiles in one launch, `blocks` and the beam tiles then count ONE frame).
int launch_timed(blok_hip_ctx* ctx, blok::RayMode mode, blok::TraceArgs args, uint32_t blocks, hipStream_t stream,
                 uint32_t tiles_of_rank, const blok::TileFrames* frames) {
    uint32_t n_beams = 0;
    if (blocks) { const int rc = prepare_beam(ctx, mode, args, stream, tiles_of_rank, &n_beams); if (rc != BLOK_OK) return rc; }
    const uint32_t n_frames = frames ? frames->n_frames : 1u;
    if (frames && n_beams) { const int rc = beam_buffer(ctx, stream, static_cast<size_t>(n_beams) * n_frames, &args.beam); if (rc != BLOK_OK) return rc; }
    blok::LaunchFacts facts{};
    facts.form = ctx->launch_form;
    facts.has_beam = n_beams != 0;
    facts.one_wave_blocks = blok::kBlock == 64;
    facts.wave_tiles = blocks * n_frames;
    const uint32_t geometry_key = blocks * 31u + n_beams * n_frames;
    facts.have_hint = list_hint(ctx, stream, geometry_key, facts.hint);
    bool busy = false;
    // A launch that is being captured into a hipGraph is replayed with these very arguments: it gets the plain two-launch form and none of
    // the per-frame bookkeeping (event queries and records outside the graph, serial numbers, orders adopted between frames).
    bool capturing = false;
    if (stream) {
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &status) == hipSuccess) capturing = status != hipStreamCaptureStatusNone;
        else (void)hipGetLastError();
    }
    if (capturing) facts.form = blok::kFormTwoLaunches;
    else busy = facts.has_beam && (ctx->launch_form == blok::kFormAuto || (ctx->order.enabled && ctx->order.moving)) && device_busy_elsewhere(ctx, stream);
    facts.device_busy = busy && ctx->launch_form == blok::kFormAuto;
    blok::LaunchPlan plan = blok::plan_launch(facts);
    if (frames && (plan.kind == blok::LaunchKind::Queues || plan.kind == blok::LaunchKind::Joint)) plan.kind = blok::LaunchKind::TwoLaunches;      // several frames per launch: the two-launch or the list forms
    ctx->last_launch_kind = static_cast<int>(plan.kind);

    blok::FrameQueue queue{};
    uint32_t frame_blocks = 0;
    blok::TileFrames fr{};
    if (frames) { fr = *frames; fr.blocks_per_frame = blocks; fr.beams_per_frame = n_beams; }
    if (plan.kind == blok::LaunchKind::Joint) { const int rc = joint_slots(ctx, args, stream, n_beams); if (rc != BLOK_OK) return rc; }
    if (plan.kind == blok::LaunchKind::Queues) { const int rc = prepare_queue(ctx, mode, args, stream, n_beams, &queue, &frame_blocks); if (rc != BLOK_OK) return rc; }
    if (plan.kind == blok::LaunchKind::ListJoint || plan.kind == blok::LaunchKind::ListTwoLaunches) {
        const uint32_t per_search = (args.beam_tile / blok::kWaveW) * (args.beam_tile / blok::kWaveH);
        const int rc = live_list(ctx, args, stream, n_beams * n_frames, per_search);
        if (rc != BLOK_OK) return rc;
        for (uint32_t c = 0; c < blok::kListClasses; ++c) args.list.walkers[c] = plan.walkers_per_class[c];
        if (mode == blok::RayMode::Rect && ctx->list_classes) { const int rc2 = list_costs(ctx, args, blocks, stream); if (rc2 != BLOK_OK) return rc2; }
        auto& slot = ctx->beam_buffers[stream];
        slot.list_hint_valid = true; slot.list_hint_key = geometry_key;      // the searches of this launch write the hint
    }
    // static forms over a rectangle: longest-first order of a camera at rest, and walk waves for its live prefix only
    const bool static_form = plan.kind == blok::LaunchKind::TwoLaunches || plan.kind == blok::LaunchKind::Joint;
    // (round 4: a rank's tile launches too — one frame or several per launch — for a view at rest: ctx->order.rank_tiles)
    const bool rect = mode == blok::RayMode::Rect;
    const bool orderable = !capturing && ctx->order.enabled && static_form && ((rect && !frames) || (mode == blok::RayMode::Tiles && ctx->order.rank_tiles)) && n_beams &&
                           blocks >= blok::kOrderMinTiles && BLOK_XCD_MAP == 0 && blok::kBlock == 64;
    bool uniform_view = true;
    if (frames) for (uint32_t f = 1; f < frames->n_frames; ++f) uniform_view = uniform_view && camera_near(ctx, frames->cam[0], frames->cam[f]);
    blok::OrderPlan order_plan{};
    if (orderable) { const int rc = order_before_launch(ctx, args, blocks, stream, !busy, rect, uniform_view, &order_plan); if (rc != BLOK_OK) return rc; }
    else ctx->order.last_use = 0;
    uint32_t walk_blocks = blocks;
    if (args.order && args.rank_of && plan.may_use_prefix && args.launched <= blocks) {
        // the search wave of a beam tile that is live now walks any wave tile of its own without a walk wave (a changed view), and writes
        // the miss pixels of the empty ones
        if (ctx->order.prefix_limit && args.launched > ctx->order.prefix_limit) args.launched = ctx->order.prefix_limit;      // tests: more work for the search waves
        walk_blocks = args.launched;
        // what the previous prefix launch left to its search waves (read now: that launch is over, or nearly), counted afresh for this one
        // (a diagnostic, counted only for timed launches — blok_hip_set_timing: a clear in front of the frame and a copy behind it are two more
        // operations on the stream of every frame)
        if (ctx->timing) {
            ctx->order.last_fallback = *static_cast<volatile uint32_t*>(ctx->order.h_fallback);
            BLOK_HIP_TRY(ctx, hipMemsetAsync(ctx->order.d_fallback, 0, sizeof(uint32_t), stream));
            args.fallback_tiles = ctx->order.d_fallback;
        }
        // an order carried over by a shift says nothing about the strips of the screen the shift brings in: every tile there gets a walk
        // workgroup of its own, in front of the prefix (trace_kernels.h: TraceArgs::n_strip)
        if (args.order_sx || args.order_sy) {
            const uint32_t tiles_x = (args.w + blok::kTileW - 1u) / blok::kTileW, tiles_y = (args.h + blok::kTileH - 1u) / blok::kTileH;
            const bool left = args.order_sx * 2u <= tiles_x, top = args.order_sy * 2u <= tiles_y;      // the shift as a signed number: towards +x / +y brings in the low columns / rows
            // (each strip wider by what the shift leaves over near the edge it comes in through: a rotation stretches the screen there)
            const float res = ctx->order.have_residual && ctx->order.last_residual >= 0.0f ? std::min(ctx->order.last_residual, 8.0f) : 0.0f;
            const uint32_t extra = static_cast<uint32_t>(std::ceil(res));
            args.strip_nx = left ? args.order_sx : tiles_x - args.order_sx; if (args.strip_nx) args.strip_nx = std::min(args.strip_nx + extra, tiles_x);
            args.strip_ny = top ? args.order_sy : tiles_y - args.order_sy; if (args.strip_ny) args.strip_ny = std::min(args.strip_ny + extra, tiles_y);
            args.strip_x0 = left ? 0u : tiles_x - args.strip_nx;
            args.strip_y0 = top ? 0u : tiles_y - args.strip_ny;
            args.n_strip = args.strip_nx * tiles_y + (tiles_x - args.strip_nx) * args.strip_ny;
            walk_blocks += args.n_strip;
        }
    } else { args.rank_of = nullptr; args.launched = 0u; }
    if (ctx->timing && !capturing) BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
    args.miss_in_walk = static_form && !args.rank_of && ctx->miss_in_walk ? 1u : 0u;
    switch (plan.kind) {
        case blok::LaunchKind::Walk:
            if (frames) blok::launch_tile_frames(args, fr, stream); else blok::launch_trace(mode, args, blocks, stream);
            break;
        case blok::LaunchKind::TwoLaunches:
            if (frames) blok::launch_tile_frames(args, fr, stream, walk_blocks);
            else { blok::launch_beam(mode, args, n_beams, stream); blok::launch_trace(mode, args, walk_blocks, stream); }
            break;
        case blok::LaunchKind::Queues: blok::launch_frame(mode, args, queue, frame_blocks, stream); break;
        case blok::LaunchKind::Joint: blok::launch_joint(mode, args, n_beams, walk_blocks, stream); break;
        case blok::LaunchKind::ListJoint: blok::launch_list_joint(mode, args, frames ? &fr : nullptr, n_beams * n_frames, plan.walkers, stream); break;
        case blok::LaunchKind::ListTwoLaunches:
            if (frames) blok::launch_beam_frames(args, fr, stream); else blok::launch_beam(mode, args, n_beams, stream);
            blok::launch_list_walk(mode, args, frames ? &fr : nullptr, plan.walkers, stream);
            break;
    }
    BLOK_HIP_TRY(ctx, hipGetLastError());
    if (ctx->timing && !capturing) { BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream)); ctx->timed = true; }
    if (args.fallback_tiles) BLOK_HIP_TRY(ctx, hipMemcpyAsync(ctx->order.h_fallback, ctx->order.d_fallback, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (orderable) { const int rc = order_after_launch(ctx, args, blocks, n_beams, stream, order_plan); if (rc != BLOK_OK) return rc; }
    if (facts.has_beam && !capturing) return note_frame_launch(ctx, stream);          // (behind the sort, if one was started: it belongs to this launch)
    return BLOK_OK;
}

}  // namespace blok_api
