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
// -> use_order | measure << 1 | start_sort << 2; still_frames and next interval through the pointers
int policy_plan_order(int enabled, int have_order, int near_order_view, int near_last_view, int sort_pending, unsigned still_frames,
                      unsigned frames_since_sort, unsigned interval, unsigned interval_now, unsigned* still_out, unsigned* interval_now_out) {
    blok::OrderFacts f{};
    f.enabled = enabled != 0; f.have_order = have_order != 0; f.near_order_view = near_order_view != 0; f.near_last_view = near_last_view != 0;
    f.sort_pending = sort_pending != 0; f.still_frames = still_frames; f.frames_since_sort = frames_since_sort; f.interval = interval; f.interval_now = interval_now;
    const blok::OrderPlan p = blok::plan_order(f);
    *still_out = p.still_frames; *interval_now_out = p.next_interval_now;
    return (p.use_order ? 1 : 0) | (p.measure ? 2 : 0) | (p.start_sort ? 4 : 0);
}
}
