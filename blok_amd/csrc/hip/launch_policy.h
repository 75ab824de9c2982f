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
    kFormAuto = 3,             // list launches: joint when the device is otherwise idle, two launches when not
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
    uint32_t hint_per_segment; // longest segment of the previous list launch on this stream (0 = none yet)
    bool have_hint;
};

struct LaunchPlan {
    LaunchKind kind;
    uint32_t walkers;          // list forms: walk workgroups (a multiple of 8); else wave_tiles
};

constexpr uint32_t kPolicySegments = 8;
constexpr uint32_t kPolicyMinWalkersPerSegment = 1024;     // 8192 walk waves: one resident round of the chip

// List forms size the walk grid from the previous launch's list — an eighth more, and a floor — never beyond one workgroup per wave tile.
// The size is a hint only: a walk wave strides over its segment, so any grid walks every entry.
inline uint32_t list_walkers(uint32_t wave_tiles, bool have_hint, uint32_t hint_per_segment) {
    const uint32_t all = (wave_tiles + kPolicySegments - 1u) / kPolicySegments;
    uint32_t per_seg = all;
    if (have_hint) {
        const uint64_t want = static_cast<uint64_t>(hint_per_segment) + hint_per_segment / 8u + 64u;
        per_seg = want < kPolicyMinWalkersPerSegment ? kPolicyMinWalkersPerSegment : static_cast<uint32_t>(want > all ? all : want);
        if (per_seg > all) per_seg = all;
    }
    if (per_seg == 0u) per_seg = 1u;
    return per_seg * kPolicySegments;
}

inline LaunchPlan plan_launch(const LaunchFacts& f) {
    LaunchPlan p{LaunchKind::Walk, f.wave_tiles};
    if (!f.has_beam) return p;                                            // no pre-pass: the walk alone
    const bool list_ok = f.one_wave_blocks && f.wave_tiles < (1u << 21);      // a list entry names its wave tile in 21 bits (trace_kernels.h)
    switch (f.form) {
        case kFormQueues: p.kind = LaunchKind::Queues; return p;
        case kFormJoint: p.kind = LaunchKind::Joint; return p;
        case kFormTwoLaunches: p.kind = LaunchKind::TwoLaunches; return p;
        case kFormListJoint: p.kind = list_ok ? LaunchKind::ListJoint : LaunchKind::Joint; break;
        case kFormListTwoLaunches: p.kind = list_ok ? LaunchKind::ListTwoLaunches : LaunchKind::TwoLaunches; break;
        default:                                                           // automatic
            // A joint launch's waiting walk waves hold wave slots: alone on the device that is what starts the walk under the searches'
            // tail; beside other launches it only takes slots from them — and two joint launches can starve each other's searches — so a
            // launch that may not be alone keeps the searches and the walk as two launches.
            p.kind = !list_ok ? LaunchKind::TwoLaunches : (f.device_busy ? LaunchKind::ListTwoLaunches : LaunchKind::ListJoint);
            break;
    }
    if (p.kind == LaunchKind::ListJoint || p.kind == LaunchKind::ListTwoLaunches) p.walkers = list_walkers(f.wave_tiles, f.have_hint, f.hint_per_segment);
    return p;
}

}  // namespace blok
#endif
