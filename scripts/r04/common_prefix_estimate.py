"""VERDICT r3 item 2, proposal (i): the 64 lanes of a wave walking the pencil's COMMON prefix cooperatively.  How deep is the common prefix?  For sampled live
8x8-pixel wave tiles of the 4K benchmark frame: the lowest tree level at which the start voxels of its 64 rays (the points at the beam tile's start parameter,
emulated as in scripts/r04/path_start_sim.py) still share a node = how many of the walk's descents from the root a cooperative start could take for the wave at once."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
hk = H.HostKernel(pw.nodes, pw.sub_chunks); L = H.lib()
L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
for pose in (0, 1, 2):
    cam = W.scene_camera(n, pose, Wd, Ht)
    rays = O.primary_rays(cam, Wd, Ht).reshape(Ht, Wd)
    levels = []; iters = []
    for ty in range(8, 270, 9):
        for tx in range(5, 480, 11):
            bx, by = (tx * 8) // 32 * 32, (ty * 8) // 32 * 32
            x0, y0 = max(bx - 1, 0), max(by - 1, 0); w, h = min(bx + 33, Wd) - x0, min(by + 33, Ht) - y0
            out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
            L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, x0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
            t = np.where(out['hit'] == 1, out['t'], np.inf)
            if not np.isfinite(t.min()): continue
            t0 = max(float(t.min()) - 2.0, 0.001)
            r = rays[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8].reshape(-1)
            p = r['org'].astype(np.float64) + r['dir'].astype(np.float64) * t0
            v = np.floor(p).astype(np.int64)
            if (v < 0).any() or (v >= n).any(): continue      # (a start outside the box: the world's entry decides, not counted)
            diff = np.bitwise_or.reduce((v ^ v[0]).reshape(-1))
            lvl = 0 if diff == 0 else (int(diff).bit_length() + 1) // 2       # cells of 4^lvl voxels hold all 64 start voxels
            levels.append(min(lvl, 5))
            ts = np.full(64, t0, dtype=np.float32); o2 = np.zeros(64, dtype=O.HIT); i2 = np.zeros(64, dtype=np.uint32)
            L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, tx * 8, ty * 8, 8, 8, C.c_void_p(ts.ctypes.data), C.c_void_p(o2.ctypes.data), C.c_void_p(i2.ctypes.data))
            iters.append(int(i2.max()))
    lv = np.array(levels); saved = 5 - lv
    print(f"pose {'ABC'[pose]}: {len(lv)} live wave tiles; level of the common ancestor of the 64 start voxels: " + ", ".join(f"{k}: {np.mean(lv == k):.2f}" for k in range(6)) +
          f"; descents a cooperative start takes for the wave: mean {saved.mean():.2f} of 5; the wave's walk: mean {np.mean(iters):.1f} iterations (longest lane)")
