// Which kernels a rectangle / tile frame is launched as: a pure function of a few facts, so that the table can be tested on the host
// (tests/test_launch_policy.py compiles this header with g++; nothing here touches HIP).  No reference counterpart: the reference issues
// one traceRaysKHR per frame (blok/src/renderer_raytracing.cpp:666-685).
#ifndef BLOK_LAUNCH_POLICY_H
#define BLOK_LAUNCH_POLICY_H
#include <stdint.h>

namespace blok {

// blok_hip_set_fused (include/blok_hip.h)
enum LaunchForm : int {
    kFormTwoLaunches = 0,      // beam kernel, then one walk wave per wave tile
    kFormQueues = 1,           // one persistent launch with work queues (measured slower; kept for comparison)
    kFormJoint = 2,            // searches and one walk wave per wave tile in one grid
    kFormAuto = 3,             // joint when the device is otherwise idle, two launches when not
    kFormListJoint = 4,        // searches and list-fed walk waves in one grid
    kFormListTwoLaunches = 5,  // beam kernel (fills the list), then list-fed walk waves
};

enum class LaunchKind : int { Walk, TwoLaunches, Queues, Joint, ListJoint, ListTwoLaunches };

struct LaunchFacts {
    int form;                  // LaunchForm asked for
    bool has_beam;             // the launch has a beam pre-pass (Rect / Tiles with a beam tile that fits)
    bool one_wave_blocks;      // the library was built with one wave per workgroup (the list and joint kernels need it)
    bool device_busy;          // launches of OTHER streams or contexts may still be running on this device
    uint32_t wave_tiles;       // walk workgroups of the static forms = wave tiles of the launch
    uint32_t hint[4];          // per cost class: the longest of the eight lists of the previous list launch on this stream
    bool have_hint;            // ... if there was one, of the same launch geometry
};

constexpr uint32_t kPolicySegments = 8, kPolicyClasses = 4, kPolicyUnknownClass = 1;
constexpr uint32_t kPolicyMinWalkers = 16;     // per segment and class: whatever turns up in a class nobody expected is walked by these, striding

struct LaunchPlan {
    LaunchKind kind;
    uint32_t walkers;                          // walk workgroups: list forms 8 x the sum of walkers_per_class; else wave_tiles
    uint32_t walkers_per_class[kPolicyClasses];        // list forms: per segment
    bool may_use_prefix;                       // static forms under a longest-first order: walk waves for the order's live prefix only
};

// ---- longest-first order of a camera at rest (static forms; api.hip: TileOrder) -----------------------------------------------------
// An order sorted from another view is worse than none (measured: a camera orbiting by 1 degree per frame loses 5 % under an order up to 8
// frames old; ordering by a stale cost is no better than row-major even one frame later, profiles/r03_stale_cost_order_experiment.txt),
// so an order is used only within ~0.25 degree of the view it was measured in, and a camera in motion is neither measured nor sorted for.
struct OrderFacts {
    bool enabled;              // blok_hip_set_tile_ordering, and the launch is a rectangle of >= kOrderMinTiles wave tiles behind the pre-pass in a static form
    bool have_order;           // a finished sort has been adopted for this launch geometry
    bool near_order_view;      // the camera is within the window of the view that order was measured in
    bool near_last_view;       // ... of the previous launch's view (the camera is at rest)
    bool sort_pending;         // a sort has been launched and not adopted yet
    uint32_t still_frames;     // consecutive launches at rest before this one
    uint32_t frames_since_sort, interval, interval_now;
};
struct OrderPlan {
    bool use_order;            // walk waves take their tiles in the adopted order
    bool measure;              // walk waves leave their clocks (the camera is at rest)
    bool start_sort;           // a sort of the clocks follows this launch on its stream
    uint32_t still_frames;     // updated count
    uint32_t next_interval_now;
};
constexpr uint32_t kOrderMinTiles = 4096;      // smaller launches have no tail worth a sort

inline OrderPlan plan_order(const OrderFacts& f) {
    OrderPlan p{false, false, false, 0u, f.interval};
    if (!f.enabled) return p;
    p.use_order = f.have_order && f.near_order_view;
    p.still_frames = f.near_last_view ? f.still_frames + 1u : 0u;
    p.measure = p.still_frames >= 1u;
    // sort when there is no order for this view yet, or the current one is `interval` launches old; a view at rest is re-sorted ever
    // less often (its costs do not change): the interval doubles with every re-sort of the same view, up to 64
    const uint32_t base = f.interval_now > f.interval ? f.interval_now : f.interval;
    p.next_interval_now = p.use_order ? f.interval_now : f.interval;
    const bool due = !p.use_order || f.frames_since_sort + 1u >= (p.use_order ? base : f.interval);
    p.start_sort = !f.sort_pending && f.interval != 0u && p.measure && due;
    if (p.start_sort && p.use_order) { const uint32_t twice = base * 2u, cap = f.interval > 64u ? f.interval : 64u; p.next_interval_now = twice < cap ? twice : cap; }
    return p;
}

// List forms size the walk grid class by class from the previous launch's lists — an eighth more, and a floor — never beyond one workgroup
// per wave tile of the segment.  Without a previous launch every wave tile is of the unknown class, which then gets them all.  The sizes
// are hints only: walk wave k of a class strides over its list, so any numbers walk every entry.
inline void list_walkers(uint32_t wave_tiles, bool have_hint, const uint32_t hint[kPolicyClasses], uint32_t out[kPolicyClasses]) {
    const uint32_t all = (wave_tiles + kPolicySegments - 1u) / kPolicySegments;
    for (uint32_t c = 0; c < kPolicyClasses; ++c) {
        uint64_t want = kPolicyMinWalkers;
        if (have_hint) want = static_cast<uint64_t>(hint[c]) + hint[c] / 8u + kPolicyMinWalkers;
        else if (c == kPolicyUnknownClass) want = all;
        out[c] = static_cast<uint32_t>(want > all ? all : want);
        if (out[c] == 0u) out[c] = 1u;
    }
}

inline LaunchPlan plan_launch(const LaunchFacts& f) {
    LaunchPlan p{LaunchKind::Walk, f.wave_tiles, {0u, 0u, 0u, 0u}, false};
    if (!f.has_beam) return p;                                            // no pre-pass: the walk alone
    const bool list_ok = f.one_wave_blocks && f.wave_tiles < (1u << 21);      // a list entry names its wave tile in 21 bits (trace_kernels.h)
    switch (f.form) {
        case kFormQueues: p.kind = LaunchKind::Queues; break;
        case kFormJoint: p.kind = LaunchKind::Joint; break;
        case kFormTwoLaunches: p.kind = LaunchKind::TwoLaunches; break;
        case kFormListJoint: p.kind = list_ok ? LaunchKind::ListJoint : LaunchKind::Joint; break;
        case kFormListTwoLaunches: p.kind = list_ok ? LaunchKind::ListTwoLaunches : LaunchKind::TwoLaunches; break;
        default:                                                           // automatic
            // A joint launch's waiting walk waves hold wave slots: alone on the device that is what starts the walk under the searches'
            // tail; beside other launches it only takes slots from them — and two joint launches can starve each other's searches — so a
            // launch that may not be alone keeps the searches and the walk as two launches.
            p.kind = (!f.one_wave_blocks || f.device_busy) ? LaunchKind::TwoLaunches : LaunchKind::Joint;
            break;
    }
    // an explicit two-launch form keeps one walk wave per wave tile (what it is there to be compared with)
    p.may_use_prefix = f.one_wave_blocks && (p.kind == LaunchKind::Joint || (p.kind == LaunchKind::TwoLaunches && f.form == kFormAuto));
    if (p.kind == LaunchKind::ListJoint || p.kind == LaunchKind::ListTwoLaunches) {
        list_walkers(f.wave_tiles, f.have_hint, f.hint, p.walkers_per_class);
        p.walkers = 0u;
        for (uint32_t c = 0; c < kPolicyClasses; ++c) p.walkers += p.walkers_per_class[c] * kPolicySegments;
    }
    return p;
}

}  // namespace blok
#endif
