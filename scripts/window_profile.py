"""Per-frame completion times (HIP events) inside a 24-frame window of the bench's pipeline, after the usual settle + warmup."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from blok_amd.multi_gpu import FramePipeline, HipBackend
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(n, 0, Wd, Ht)
pipe = FramePipeline(HipBackend(tr, cam), Wd, Ht, depth=3)
for rep in range(6):
    offset_us = (0, 0, 0, 60, 60, 90)[rep]      # lockstep is an attractor: frames that start late catch up at once (profiles/r02_stagger_experiment.txt)
    for _ in range(37):
        pipe.step()
    pipe.flush(); torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); evs = []
    ev0.record(pipe.streams[0])
    if offset_us:           # break the symmetry once: streams 1 and 2 start one and two offsets late
        for i in (1, 2):
            with torch.cuda.stream(pipe.streams[i]):
                torch.cuda._sleep(int(i * offset_us * 1e-6 * 2.0e9))
    for k in range(24):
        pipe.step()
        e = torch.cuda.Event(enable_timing=True); e.record(pipe.streams[(pipe.frames_submitted - 1) % 3]); evs.append(e)
    pipe.flush(); torch.cuda.synchronize()
    t = [ev0.elapsed_time(e) for e in evs]
    d = np.diff([0.0] + t)
    print("initial offset", offset_us, "us; window", rep, "frame completion deltas (ms):", " ".join(f"{x:.3f}" for x in d), " total", f"{t[-1]:.3f}", flush=True)
tr.shutdown()
