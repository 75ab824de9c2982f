"""CPU statistics of the kernel's walk (host harness): iterations per ray, per 8x8 wave tile, per level."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pose = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w, h = 3840 // 4, 2160 // 4     # quarter-res frame with the full-res camera footprint
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_trace_primary_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 2 + [C.c_void_p] * 3
cam = W.scene_camera(n, pose, w, h)
out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32); tot = np.zeros((5, 8), dtype=np.uint64)
L.hh_trace_primary_stats(hk.h, C.c_void_p(cam.ctypes.data), w, h, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data), C.c_void_p(tot.ctypes.data))
it2 = it.reshape(h, w)
print("rays", w * h, "hit frac", out['hit'].mean(), "iters/ray mean", it.mean(), "max", it.max(), "hit rays mean", it[out['hit'] == 1].mean(), "miss rays mean", it[out['hit'] == 0].mean())
tiles = it2[:h // 8 * 8, :w // 8 * 8].reshape(h // 8, 8, w // 8, 8)
wmax = tiles.max(axis=(1, 3)); wmean = tiles.mean(axis=(1, 3))
print("wave(8x8) max-iter mean", wmax.mean(), " => lane utilisation", wmean.mean() / wmax.mean())
names = ["iter", "descend", "step", "ascend"]
for e in range(4):
    print(names[e], "per ray by level", np.round(tot[e] / (w * h), 2))
