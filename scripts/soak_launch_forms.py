"""Soak test of the default launch form (automatic: joint launch alone, two launches over the order's live prefix with frames in
flight): minutes of frames with a camera that rests, creeps, orbits, jumps and returns, one and three streams, and TWO contexts with frames
in flight at the same time (the device-wide "is anybody else launching" registry: no two joint launches may overlap) — every frame compared
with the plain two-launch form of a reference context; reports the slowest frame of each phase (a stall shows there), how often the carried
order of a moving camera was in use, and the walk waves that gave up waiting (must be 0)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
tr2 = HipTracer(Wd, Ht).init(); tr2.add_world(pw)                       # a second context in the automatic form, for the "contexts" phases
ref = HipTracer(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0); ref.set_tile_ordering(0)
carried = 0; joint_pairs = 0
streams = [torch.cuda.Stream() for _ in range(3)]
bufs = [(torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"), torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")) for _ in streams]
want = (torch.zeros_like(bufs[0][0]), torch.zeros_like(bufs[0][1]))
rng = np.random.default_rng(7)
centre = np.array([512.0, 60.0, 512.0])
frames = 0; worst = 0.0; t_end = time.time() + seconds; phase = 0
while time.time() < t_end:
    phase += 1
    kind = rng.choice(["rest", "creep", "orbit", "jump", "flight", "contexts", "rects"])
    a0 = rng.uniform(0, 2 * np.pi); r = rng.uniform(500, 1100); hgt = rng.uniform(150, 600)
    def cam_at(a):
        return W.camera_look_at((centre[0] + r * np.cos(a), hgt, centre[2] + r * np.sin(a)), tuple(centre), 60.0, Wd, Ht)
    cams = {"rest": [cam_at(a0)] * 40, "creep": [cam_at(a0 + np.radians(0.04 * k)) for k in range(40)],
            "jump": [cam_at(a0 + (k // 5) * 0.7) for k in range(30)], "flight": [cam_at(a0)] * 45,
            "orbit": [cam_at(a0 + np.radians(rng.choice([0.5, 1.0, 2.0]) * k)) for k in range(40)], "contexts": [cam_at(a0)] * 24 + [cam_at(a0 + np.radians(k)) for k in range(24)],
            "rects": [cam_at(a0 + np.radians(0.7 * k)) for k in range(30)]}[kind]
    slow = 0.0
    if kind == "flight":
        ref.draw_frame_device(cams[0], want[0].data_ptr(), want[1].data_ptr())
        t0 = time.perf_counter()
        for k, c in enumerate(cams):
            tr.draw_frame_device(c, bufs[k % 3][0].data_ptr(), bufs[k % 3][1].data_ptr(), stream=streams[k % 3].cuda_stream)
        torch.cuda.synchronize()
        slow = (time.perf_counter() - t0) / len(cams) * 1e3
        for b in bufs:
            assert torch.equal(b[0], want[0]) and torch.equal(b[1], want[1]), (phase, kind)
        frames += len(cams)
    elif kind == "rects":
        # the launch geometry changes from frame to frame and from stream to stream (orders are per geometry: sorts of the geometry just
        # left may still be running), camera moving and resting
        t0 = time.perf_counter()
        for k, c in enumerate(cams):
            c = cams[k if k < 20 else 20]
            x0, y0 = int(rng.integers(0, 40)) * 32, int(rng.integers(0, 20)) * 32
            w, h = int(rng.integers(1500, Wd - x0 - 100)), int(rng.integers(900, Ht - y0 - 50))
            rect = None if k % 3 == 0 else (x0, y0, w, h)
            for b in (bufs[k % 3], want):
                b[0].fill_(7); b[1].fill_(7)
            ref.draw_frame_device(c, want[0].data_ptr(), want[1].data_ptr(), rect=rect)
            tr.draw_frame_device(c, bufs[k % 3][0].data_ptr(), bufs[k % 3][1].data_ptr(), rect=rect, stream=streams[k % 3].cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(bufs[k % 3][0], want[0]) and torch.equal(bufs[k % 3][1], want[1]), (phase, kind, k, rect)
            carried += int(tr.last_order_use()[0] == 2)
            frames += 1
        slow = (time.perf_counter() - t0) / len(cams) * 1e3
    elif kind == "contexts":
        # both contexts launch at once, each on its own stream, 3 frames each per round: whichever launches second sees the other's frame pending
        t0 = time.perf_counter()
        for k in range(0, len(cams), 3):
            ref.draw_frame_device(cams[k], want[0].data_ptr(), want[1].data_ptr())
            for j in range(3):
                (tr if j % 2 == 0 else tr2).draw_frame_device(cams[k], bufs[j][0].data_ptr(), bufs[j][1].data_ptr(), stream=streams[j].cuda_stream)
                if j == 1:
                    joint_pairs += int(tr.last_launch_kind() == 3 and tr2.last_launch_kind() == 3)      # both joint can only mean the first had finished
            torch.cuda.synchronize()
            for b in bufs:
                assert torch.equal(b[0], want[0]) and torch.equal(b[1], want[1]), (phase, kind, k)
            frames += 3
        slow = (time.perf_counter() - t0) / len(cams) * 1e3
    else:
        for k, c in enumerate(cams):
            ref.draw_frame_device(c, want[0].data_ptr(), want[1].data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.draw_frame_device(c, bufs[0][0].data_ptr(), bufs[0][1].data_ptr())
            torch.cuda.synchronize()
            slow = max(slow, (time.perf_counter() - t0) * 1e3)
            assert torch.equal(bufs[0][0], want[0]) and torch.equal(bufs[0][1], want[1]), (phase, kind, k)
            carried += int(tr.last_order_use()[0] == 2)
            frames += 1
    worst = max(worst, slow)
    if phase % 10 == 0:
        print(f"phase {phase} ({kind}): {frames} frames so far, slowest frame of this phase {slow:.3f} ms, of all {worst:.3f} ms, waves that gave up {tr.frame_queue_stalls() + tr2.frame_queue_stalls()}", flush=True)
print(f"soak ok: {frames} frames in {phase} phases, every frame identical to the two-launch form; slowest frame {worst:.3f} ms (host clock, incl. launch + sync); waves that gave up: {tr.frame_queue_stalls() + tr2.frame_queue_stalls()}; frames walked in a carried order: {carried}; consecutive joint launches of the two contexts (the first had finished): {joint_pairs}")
tr.shutdown(); tr2.shutdown(); ref.shutdown()
