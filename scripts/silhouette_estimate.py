"""How much of the walk is spent in wave tiles (8x8) whose rays ALL miss although their 32x32 beam tile is live (silhouettes)?  CPU
harness, 4K frame sampled in strips of beam-tile rows; per-ray iterations behind an ideal per-beam-tile start parameter; wave cost =
the longest ray of the wave (what a wave64 pays)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = 1024; pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
Wd, Ht = 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_trace_rect_stats.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 6 + [C.c_void_p] * 3
cam = W.scene_camera(n, pose, Wd, Ht)
tot_cost = miss_cost = 0.0; n_waves = n_miss_waves = 0; live_tiles = 0
for y0 in range(0, Ht - 31, 32 * 6):            # every 6th row of beam tiles
    out = np.zeros(32 * Wd, dtype=O.HIT); it = np.zeros(32 * Wd, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, Wd, 32, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    hit = out["hit"].reshape(32, Wd); t = out["t"].reshape(32, Wd)
    # ideal beam: per 32x32 tile the smallest hit t (minus a voxel), then re-trace with it
    tstart = np.zeros((32, Wd), dtype=np.float32)
    live = np.zeros(Wd // 32, dtype=bool)
    for bx in range(Wd // 32):
        hh = hit[:, bx * 32:(bx + 1) * 32] == 1
        if hh.any():
            live[bx] = True
            tstart[:, bx * 32:(bx + 1) * 32] = max(0.0, float(t[:, bx * 32:(bx + 1) * 32][hh].min()) - 2.0)
    it2 = np.zeros(32 * Wd, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, Wd, 32, C.c_void_p(tstart.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(it2.ctypes.data))
    it2 = it2.reshape(32, Wd)
    for bx in np.flatnonzero(live):
        live_tiles += 1
        for sy in range(4):
            for sx in range(4):
                sl = (slice(sy * 8, sy * 8 + 8), slice(bx * 32 + sx * 8, bx * 32 + sx * 8 + 8))
                c = float(it2[sl].max())
                tot_cost += c; n_waves += 1
                if not (hit[sl] == 1).any():
                    miss_cost += c; n_miss_waves += 1
print(f"pose {'ABC'[pose]}: live beam tiles sampled {live_tiles}, walk waves {n_waves}, all-miss waves {n_miss_waves} ({n_miss_waves / n_waves:.1%}), "
      f"their share of the wave-iterations {miss_cost / tot_cost:.1%}; mean iterations of an all-miss wave {miss_cost / max(1, n_miss_waves):.1f} vs {(tot_cost - miss_cost) / max(1, n_waves - n_miss_waves):.1f}")
