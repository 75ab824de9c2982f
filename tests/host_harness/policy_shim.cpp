// TEST INFRASTRUCTURE: exposes blok_amd/csrc/hip/launch_policy.h (pure C++, no HIP) to tests/test_launch_policy.py.
#include "launch_policy.h"

extern "C" {
// -> kind | may_use_prefix << 8; walkers (total) and per class through the pointers
int policy_plan_launch(int form, int has_beam, int one_wave_blocks, int device_busy, unsigned wave_tiles, int have_hint, const unsigned* hint4,
                       unsigned* walkers, unsigned* per_class4) {
    blok::LaunchFacts f{};
    f.form = form; f.has_beam = has_beam != 0; f.one_wave_blocks = one_wave_blocks != 0; f.device_busy = device_busy != 0;
    f.wave_tiles = wave_tiles; f.have_hint = have_hint != 0;
    for (int c = 0; c < 4; ++c) f.hint[c] = hint4 ? hint4[c] : 0u;
    const blok::LaunchPlan p = blok::plan_launch(f);
    *walkers = p.walkers;
    for (int c = 0; c < 4; ++c) per_class4[c] = p.walkers_per_class[c];
    return static_cast<int>(p.kind) | (p.may_use_prefix ? 256 : 0);
}
// -> use_order | measure << 1 | start_sort << 2 | shifted << 3 | dilate << 4; still_frames and next interval through the pointers
int policy_plan_order(int enabled, int have_order, int near_order_view, int near_last_view, int sort_pending, unsigned still_frames,
                      unsigned frames_since_sort, unsigned interval, unsigned interval_now, int moving_enabled, int alone, int alone_before, int order_dilated, int shift_ok,
                      unsigned* still_out, unsigned* interval_now_out) {
    blok::OrderFacts f{};
    f.enabled = enabled != 0; f.have_order = have_order != 0; f.near_order_view = near_order_view != 0; f.near_last_view = near_last_view != 0;
    f.sort_pending = sort_pending != 0; f.still_frames = still_frames; f.frames_since_sort = frames_since_sort; f.interval = interval; f.interval_now = interval_now;
    f.moving_enabled = moving_enabled != 0; f.alone = alone != 0; f.alone_before = alone_before != 0; f.order_dilated = order_dilated != 0; f.shift_ok = shift_ok != 0;
    const blok::OrderPlan p = blok::plan_order(f);
    *still_out = p.still_frames; *interval_now_out = p.next_interval_now;
    return (p.use_order ? 1 : 0) | (p.measure ? 2 : 0) | (p.start_sort ? 4 : 0) | (p.shifted ? 8 : 0) | (p.dilate ? 16 : 0);
}
// cameras: 14 floats each (blok_camera).  -> ok; shift (modulo the grid) and residual through the pointers
int policy_plan_shift(const float* then14, const float* now14, float inv_depth_mean, float inv_depth_sigma, unsigned frame_w, unsigned frame_h,
                      unsigned tiles_x, unsigned tiles_y, unsigned radius, unsigned* sx, unsigned* sy, float* residual) {
    blok::ShiftFacts f{};
    for (int i = 0; i < 14; ++i) { reinterpret_cast<float*>(&f.then)[i] = then14[i]; reinterpret_cast<float*>(&f.now)[i] = now14[i]; }
    f.inv_depth_mean = inv_depth_mean; f.inv_depth_sigma = inv_depth_sigma; f.frame_w = frame_w; f.frame_h = frame_h;
    f.tiles_x = tiles_x; f.tiles_y = tiles_y; f.tile_w = 8; f.tile_h = 8; f.radius = radius;
    const blok::ShiftPlan p = blok::plan_shift(f);
    *sx = p.sx; *sy = p.sy; *residual = p.residual;
    return (p.ok ? 1 : 0) | (p.measured ? 2 : 0);      // bit 1: the residual was computed (false on the early exits)
}
unsigned policy_plan_dilation(int have_residual, float residual) { return blok::plan_dilation(have_residual != 0, residual); }
}
