"""GPU A/B of longest-first tile ordering (blok_hip_set_tile_ordering): the 4K frame alone (HIP events around single launches)
and back to back on three streams, ordering on / off, for the three poses and an orbiting camera."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
hits = [torch.empty((Wd * Ht, 4), dtype=torch.int32, device="cuda") for _ in range(3)]
rgba = [torch.empty(Wd * Ht, dtype=torch.int32, device="cuda") for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
centre = np.array([512.0, 256.0, 512.0]); start = np.array([-0.35 * n, 0.85 * n, -0.35 * n]) - centre
def orbit(i, deg):
    a = np.radians(deg * i)
    p = centre + np.array([start[0] * np.cos(a) - start[2] * np.sin(a), start[1], start[0] * np.sin(a) + start[2] * np.cos(a)])
    return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), 60.0, Wd, Ht)
cases = [(f"pose {'ABC'[p]}", [W.scene_camera(n, p, Wd, Ht)]) for p in (0, 1, 2)] + [("orbit 0.2 deg/frame", [orbit(i, 0.2) for i in range(40)]), ("orbit 2 deg/frame", [orbit(i, 2.0) for i in range(40)])]
for name, cams in cases:
    for ordering in (8, 0, 1, 0):
        tr = HipTracer(Wd, Ht).init(); tr.add_world(pw); tr.set_tile_ordering(ordering)
        st = streams[0].cuda_stream
        for k in range(6):
            tr.draw_frame_device(cams[k % len(cams)], hits[0].data_ptr(), rgba[0].data_ptr(), stream=st)
        torch.cuda.synchronize()
        tr.set_timing(True); ms = []
        for k in range(40):
            tr.draw_frame_device(cams[k % len(cams)], hits[0].data_ptr(), rgba[0].data_ptr(), stream=st); torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
        tr.set_timing(False)
        t = time.perf_counter()
        for k in range(120):
            j = k % 3
            tr.draw_frame_device(cams[k % len(cams)], hits[j].data_ptr(), rgba[j].data_ptr(), stream=streams[j].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 120 * 1e3
        print(f"{name:22s} ordering every {ordering}: alone {np.mean(ms):.4f} ms (min {np.min(ms):.4f})   3 streams {dt:.4f} ms/frame = {Wd * Ht / dt / 1e6:.1f} Grays/s", flush=True)
        tr.shutdown()
