"""GPU probe of the one-launch frame: small frame, compares with the two-launch form and prints the queue's stall counter."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 128)
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(w, h).init(); tr.add_world(pw)
cam = W.scene_camera(n, 0, w, h)
tr.set_fused(False); a = tr.draw_frame(cam).reshape(-1)
tr.set_fused(True)
t = time.perf_counter(); b = tr.draw_frame(cam).reshape(-1); dt = time.perf_counter() - t
same = (a.view(np.uint8).reshape(-1, 16) == b.view(np.uint8).reshape(-1, 16)).all(axis=1)
print(f"n={n} {w}x{h}: fused frame took {dt*1e3:.1f} ms, stalls={tr.frame_queue_stalls()}, records equal: {same.all()} ({(~same).sum()} differ)", flush=True)
for k in range(3):
    b = tr.draw_frame(cam).reshape(-1)
    same = (a.view(np.uint8).reshape(-1, 16) == b.view(np.uint8).reshape(-1, 16)).all(axis=1)
    print(f"  again: stalls={tr.frame_queue_stalls()}, equal: {same.all()}", flush=True)
tr.shutdown()
