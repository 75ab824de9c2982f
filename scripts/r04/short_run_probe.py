"""Where does a short timed region (the driver's --steps 20) lose against a long one?  bench.py's loop with an event behind every frame on
its stream: completion time of each frame since the head event, for K = 20 and K = 200, frames in flight as in bench.py, after the same
settle + warmup frames.  usage: short_run_probe.py [frames_in_flight]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from blok_amd.multi_gpu import FramePipeline, HipBackend
import bench

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, Wd, Ht = 1024, 3840, 2160
packed = bench.build_world(n, 0xB10C0001)
cam = W.scene_camera(n, 0, Wd, Ht, 0xB10C0001)
tr = HipTracer(Wd, Ht, device=0).init(); tr.add_world(packed)
tr.set_beam(32); tr.set_fused(3); tr.set_tile_ordering(8); tr.set_moving_order(True)
pipe = FramePipeline(HipBackend(tr, cam), Wd, Ht, 0, 1, None, tile=32, depth=depth, sparse=2, batch=1)
for _ in range(37): pipe.step()
pipe.flush(); torch.cuda.synchronize()
for K in (20, 20, 200, 20):
    ev0 = torch.cuda.Event(enable_timing=True)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    kinds = []
    t0 = time.perf_counter()
    ev0.record(pipe.streams[0])
    host = []
    for k in range(K):
        pipe.step()
        kinds.append(tr.last_launch_kind())
        evs[k].record(pipe.streams[pipe._cur])
        host.append(time.perf_counter() - t0)
    pipe.flush(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    done = np.array([ev0.elapsed_time(e) for e in evs])
    gaps = np.diff(np.concatenate([[0.0], np.sort(done)]))
    print(f"K = {K}: wall {wall * 1e3:.3f} ms = {wall / K * 1e3:.4f} per frame; last frame done at {done.max():.3f} ms; host enqueued the last frame at {host[-1] * 1e3:.3f} ms")
    print("   frame completion (ms):", np.round(done[:12], 3), "...", np.round(done[-6:], 3))
    print("   gaps between completions: first six", np.round(gaps[:6], 3), " median", round(float(np.median(gaps)), 4), " last four", np.round(gaps[-4:], 3))
    print("   launch kinds of the first six frames:", kinds[:6], " host time of the first six launches (ms):", np.round(np.array(host[:6]) * 1e3, 3))
