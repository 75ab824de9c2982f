"""Per 8x8 wave tile (primary rays behind an emulated beam pre-pass): how the per-ray iteration counts are distributed inside the
long waves — are they a few stragglers or whole tiles of long rays?"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = 1024
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
Wd, Ht, B = 3840, 2160, 32
cam = W.scene_camera(n, pose, Wd, Ht)
rows = []
for y0 in range(0, Ht - B + 1, B * 2):
    w, h = Wd, B
    out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf).reshape(h, w // B, B)
    tmin_tile = t.min(axis=(0, 2))
    if not np.isfinite(tmin_tile).any(): continue
    ts = np.repeat(np.maximum(tmin_tile - 2.0, 0)[None, :], h, axis=0).repeat(B, axis=1)
    tsf = np.where(np.isfinite(ts), ts, 3.0e38).astype(np.float32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, C.c_void_p(tsf.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    tiles = it.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    live = np.repeat(np.isfinite(tmin_tile), 4)[None, :].repeat(h // 8, axis=0).reshape(-1)
    rows.append(np.sort(tiles[live], axis=1)[:, ::-1])
s = np.concatenate(rows)
mx = s[:, 0]
print(f"pose {pose}: {len(s)} walking waves sampled; max iterations per wave: mean {mx.mean():.1f}, median {np.median(mx):.0f}, 90% {np.percentile(mx, 90):.0f}, 99% {np.percentile(mx, 99):.0f}, max {mx.max()}")
for lo, hi in ((0, 32), (32, 64), (64, 100), (100, 160), (160, 1000)):
    sel = s[(mx >= lo) & (mx < hi)]
    if len(sel) == 0: continue
    print(f"  waves with max in [{lo},{hi}): {len(sel):6d} ({len(sel) / len(s):.1%}), sum of max {sel[:, 0].sum() / mx.sum():.1%} of all; "
          f"per wave: mean ray {sel.mean():.1f}; 4th longest {sel[:, 3].mean():.0f}, 8th {sel[:, 7].mean():.0f}, 16th {sel[:, 15].mean():.0f}, 32nd {sel[:, 31].mean():.0f}; "
          f"lanes still walking at trip 24: {(sel > 24).sum(axis=1).mean():.1f}, at 48: {(sel > 48).sum(axis=1).mean():.1f}, at 96: {(sel > 96).sum(axis=1).mean():.1f}")
