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

// ---- longest-first order of the walk's wave tiles (static forms; api.hip: TileOrder) ---------------------------------------------------
// A camera at rest: the tiles sorted by the clocks their waves took in this very view.  An order sorted from another view's clocks is
// worse than none — the heavy tiles are silhouettes and grazing rays, a quarter of a degree moves them (measured: ordering by a stale cost
// is no better than row-major even one frame later, profiles/r03_stale_cost_order_experiment.txt) — so that order is used for the view it
// was measured in only (api.hip: camera_near — the same camera up to float noise).
// A camera in motion (round 3): the heavy tiles of the next frame are NEAR the heavy tiles of this one.  The sort that follows a frame
// alone on the device keys every tile by the largest clocks within `radius` tiles of it (tile_order.hip), and the next launch carries that
// order to its own view by ONE whole-tile shift of the screen (plan_shift: entry (tx, ty) names tile (tx + sx, ty + sy) modulo the
// grid — still a permutation, so nothing but a launch argument changes): the walk alone takes 171-178 us instead of 209-216 in row-major
// order for a camera orbiting by 1-2 degrees per frame, against 167-170 in the order of the frame's own clocks
// (profiles/r03_moving_order_experiment.txt).
struct OrderFacts {
    bool enabled;              // blok_hip_set_tile_ordering, and the launch is a rectangle of >= kOrderMinTiles wave tiles behind the pre-pass in a static form
    bool have_order;           // a finished sort has been adopted for this launch geometry
    bool near_order_view;      // the camera is within the window of the view that order was measured in
    bool near_last_view;       // ... of the previous launch's view (the camera is at rest)
    bool sort_pending;         // a sort has been launched and not adopted yet
    uint32_t still_frames;     // consecutive launches at rest before this one
    uint32_t frames_since_sort, interval, interval_now;
    bool moving_enabled;       // blok_hip_set_moving_order
    bool alone;                // no other stream or context has a frame pending on the device
    bool alone_before;         // ... and the same held for this context's previous launch
    bool order_dilated;        // the adopted order was sorted from dilated clocks (made to be carried to another view)
    bool shift_ok;             // plan_shift: a whole-tile shift carries the adopted order to this view
};
struct OrderPlan {
    bool use_order;            // walk waves take their tiles in the adopted order
    bool measure;              // walk waves leave their clocks
    bool start_sort;           // a sort of the clocks follows this launch on its stream
    uint32_t still_frames;     // updated count
    uint32_t next_interval_now;
    bool shifted;              // ... through plan_shift's shift (an order of another view)
    bool dilate;               // the sort keys are dilated clocks (the camera is in motion)
};
constexpr uint32_t kOrderMinTiles = 4096;      // smaller launches have no tail worth a sort

inline OrderPlan plan_order(const OrderFacts& f) {
    OrderPlan p{false, false, false, 0u, f.interval, false, false};
    if (!f.enabled) return p;
    const bool own_view = f.have_order && !f.order_dilated && f.near_order_view;      // sorted from this view's own clocks
    // a dilated order serves whatever view a shift carries it to — but only launches that have the device to themselves, one after the other
    // (this one and the one before): beside frames in flight the tail is filled anyway, and a pipeline's launches that merely find the
    // device idle for a moment (its frames run in lockstep and end together) gain nothing from an order several frames old, while the
    // sorts behind them would delay the next frame of their stream (measured: -1 to -2 % of the pipelined rate of an orbiting camera)
    const bool solitary = f.alone && f.alone_before;
    const bool carried = f.have_order && f.order_dilated && f.moving_enabled && solitary && f.shift_ok;
    p.use_order = own_view || carried;
    p.shifted = !own_view && carried;
    p.still_frames = f.near_last_view ? f.still_frames + 1u : 0u;
    const bool at_rest = p.still_frames >= 1u;
    p.measure = at_rest || (f.moving_enabled && solitary);
    if (at_rest) {
        // sort when there is no order of this view yet, or the current one is `interval` launches old; a view at rest is re-sorted ever
        // less often (its costs do not change): the interval doubles with every re-sort of the same view, up to 64
        const uint32_t base = f.interval_now > f.interval ? f.interval_now : f.interval;
        p.next_interval_now = own_view ? f.interval_now : f.interval;
        const bool due = !own_view || f.frames_since_sort + 1u >= base;
        p.start_sort = !f.sort_pending && f.interval != 0u && due;
        if (p.start_sort && own_view) { const uint32_t twice = base * 2u, cap = f.interval > 64u ? f.interval : 64u; p.next_interval_now = twice < cap ? twice : cap; }
    } else {
        // in motion: a solitary frame leaves the order for the next one
        p.start_sort = p.dilate = f.moving_enabled && solitary && !f.sort_pending && f.interval != 0u;
    }
    return p;
}

// The whole-tile shift that carries an order made under camera `then` to camera `now`, and how far off it is at worst.  Points on the
// central ray and towards the corners of `then`'s screen, at a near and a far depth of the frame the order was measured on (mean -/+ two
// standard deviations of its live beam tiles' inverse start parameters), are projected into both views: the shift is the displacement of
// the central point at the mean inverse depth, in tiles; `residual` the largest displacement any sample keeps after the shift
// (Chebyshev, tiles) — parallax and the stretch of a rotation towards the screen's edge.  The order was sorted from costs dilated by
// `radius` tiles; it is used while the residual stays within twice that (a residual of three radii still beats row-major order, it
// no longer pays for its sort) and the shift within a quarter of the screen.
struct PolicyCamera { float pos[3], fwd[3], right[3], up[3], tan_half_fov, aspect; };      // = blok_camera (include/blok_hip.h)
struct ShiftFacts {
    PolicyCamera then, now;
    float inv_depth_mean, inv_depth_sigma;     // of the frame under `then` (0, 0: unknown — the far field only)
    uint32_t frame_w, frame_h;                 // pixels of the whole frame (what the cameras project onto)
    uint32_t tiles_x, tiles_y, tile_w, tile_h; // the launch's grid of wave tiles
    uint32_t radius;                           // dilation the order was sorted with
};
struct ShiftPlan { bool ok; uint32_t sx, sy; float residual; bool measured; };      // sx, sy already modulo the grid; measured: residual was computed (false on the early exits: another lens, a sample behind the camera, not finite)

inline bool policy_project(const PolicyCamera& c, const float p[3], float& u, float& v) {
    const float r[3] = {p[0] - c.pos[0], p[1] - c.pos[1], p[2] - c.pos[2]};
    const float z = r[0] * c.fwd[0] + r[1] * c.fwd[1] + r[2] * c.fwd[2];
    if (!(z > 0.0f)) return false;
    u = (r[0] * c.right[0] + r[1] * c.right[1] + r[2] * c.right[2]) / (z * c.tan_half_fov * c.aspect);
    v = (r[0] * c.up[0] + r[1] * c.up[1] + r[2] * c.up[2]) / (z * c.tan_half_fov);
    return true;
}

inline ShiftPlan plan_shift(const ShiftFacts& f) {
    ShiftPlan p{false, 0u, 0u, 0.0f, false};
    if (!f.tiles_x || !f.tiles_y || !f.tile_w || !f.tile_h) return p;
    // A change of lens (a zoom) is a scale of the screen about its centre: the samples below are projected through each camera's own lens, so
    // the stretch it puts on the tiles towards the edges is part of `residual` like a rotation's — a slow zoom (0.4 % per frame moves an
    // edge tile of a 4K frame by one tile) stays within the dilation, a fast one is refused by the residual test like any other jump.
    // (Round 3 refused every lens change here: a zooming camera walked in row-major order, profiles/r04_views_probe.txt.)
    if (!(f.now.tan_half_fov > 0.0f && f.then.tan_half_fov > 0.0f && f.now.aspect > 0.0f && f.then.aspect > 0.0f)) return p;
    const float near_inv = f.inv_depth_mean + 2.0f * f.inv_depth_sigma;
    const float far_inv = f.inv_depth_mean > 2.0f * f.inv_depth_sigma ? f.inv_depth_mean - 2.0f * f.inv_depth_sigma : 0.0f;
    const float depths[3] = {f.inv_depth_mean > 0.0f ? 1.0f / f.inv_depth_mean : 1.0e7f, near_inv > 0.0f ? 1.0f / near_inv : 1.0e7f, far_inv > 0.0f ? 1.0f / far_inv : 1.0e7f};
    static const float at[5][2] = {{0.0f, 0.0f}, {-0.8f, -0.8f}, {0.8f, -0.8f}, {-0.8f, 0.8f}, {0.8f, 0.8f}};
    float dx[15], dy[15];
    int n = 0;
    for (int d = 0; d < 3; ++d)
        for (int k = 0; k < 5; ++k) {
            const float u = at[k][0], v = at[k][1];
            float dir[3], len2 = 0.0f, q[3], u2, v2;
            for (int a = 0; a < 3; ++a) {
                dir[a] = f.then.fwd[a] + f.then.right[a] * (u * f.then.tan_half_fov * f.then.aspect) + f.then.up[a] * (v * f.then.tan_half_fov);
                len2 += dir[a] * dir[a];
            }
            float inv_len = 1.0f; { float x = len2 > 0.0f ? len2 : 1.0f, g = x; for (int it = 0; it < 24; ++it) g = 0.5f * (g + x / g); inv_len = 1.0f / g; }      // no <cmath>: Newton's square root
            for (int a = 0; a < 3; ++a) q[a] = f.then.pos[a] + dir[a] * inv_len * depths[d];
            if (!policy_project(f.now, q, u2, v2)) return p;                  // behind the new camera: another view altogether
            dx[n] = (u2 - u) * 0.5f * static_cast<float>(f.frame_w) / static_cast<float>(f.tile_w);      // in tiles, x to the right
            dy[n] = -(v2 - v) * 0.5f * static_cast<float>(f.frame_h) / static_cast<float>(f.tile_h);     // y down the screen
            ++n;
        }
    const float rx = dx[0] < 0.0f ? dx[0] - 0.5f : dx[0] + 0.5f, ry = dy[0] < 0.0f ? dy[0] - 0.5f : dy[0] + 0.5f;
    if (!(rx > -1.0e6f && rx < 1.0e6f && ry > -1.0e6f && ry < 1.0e6f)) return p;       // NaN included
    const int sx = static_cast<int>(rx), sy = static_cast<int>(ry);               // rounded to nearest
    for (int k = 0; k < n; ++k) {
        const float ex = dx[k] - static_cast<float>(sx), ey = dy[k] - static_cast<float>(sy);
        const float e = (ex < 0.0f ? -ex : ex) > (ey < 0.0f ? -ey : ey) ? (ex < 0.0f ? -ex : ex) : (ey < 0.0f ? -ey : ey);
        if (!(e <= p.residual)) p.residual = e;                                  // NaN sticks
    }
    const int ax = sx < 0 ? -sx : sx, ay = sy < 0 ? -sy : sy;
    p.measured = true;
    p.ok = p.residual <= 2.0f * static_cast<float>(f.radius) + 0.5f && static_cast<uint32_t>(ax) * 4u <= f.tiles_x && static_cast<uint32_t>(ay) * 4u <= f.tiles_y;
    p.sx = static_cast<uint32_t>((sx % static_cast<int>(f.tiles_x) + static_cast<int>(f.tiles_x)) % static_cast<int>(f.tiles_x));
    p.sy = static_cast<uint32_t>((sy % static_cast<int>(f.tiles_y) + static_cast<int>(f.tiles_y)) % static_cast<int>(f.tiles_y));
    return p;
}

// The dilation of the next sort: what the latest shift left over, rounded up, between 2 and 8 tiles (4 when nothing has been observed yet).
inline uint32_t plan_dilation(bool have_residual, float residual) {
    if (!have_residual || !(residual >= 0.0f)) return 4u;
    uint32_t r = static_cast<uint32_t>(residual < 100.0f ? residual : 100.0f);
    if (static_cast<float>(r) < residual) ++r;
    return r < 2u ? 2u : (r > 8u ? 8u : r);
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
