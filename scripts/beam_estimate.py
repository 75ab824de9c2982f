"""Estimate what a per-tile conservative start parameter (beam pre-pass) saves: walk iterations per 8x8 wave tile at
full 4K resolution with tmin' = (nearest hit in the BxB beam tile) - margin against the plain walk, on bands of the frame.
usage: beam_estimate.py [n] [pose] [margin] [beam tile B = 8|16|32]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pose = int(sys.argv[2]) if len(sys.argv) > 2 else 0
margin = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
W_, H_ = 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
cam = W.scene_camera(n, pose, W_, H_)
plain = beam = 0; waves = sky_waves = 0
for y0 in [int(a) for a in (sys.argv[5].split(",") if len(sys.argv) > 5 else range(0, H_ - B + 1, 216))]:
    w, h = W_, B
    out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), W_, H_, 0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf).reshape(h, w // B, B)
    tmin_tile = t.min(axis=(0, 2))
    ts = np.repeat(np.maximum(tmin_tile - margin, 0)[None, :], h, axis=0).repeat(B, axis=1)
    tsf = np.where(np.isfinite(ts), ts, 9999.0).astype(np.float32)
    out2 = np.zeros(w * h, dtype=O.HIT); it2 = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), W_, H_, 0, y0, w, h, C.c_void_p(tsf.ctypes.data), C.c_void_p(out2.ctypes.data), C.c_void_p(it2.ctypes.data))
    assert np.array_equal(out.view(np.uint8), out2.view(np.uint8))
    a = it.reshape(h // 8, 8, w // 8, 8).max(axis=(1, 3)); b = it2.reshape(h // 8, 8, w // 8, 8).max(axis=(1, 3))
    sky = np.repeat(np.repeat(~np.isfinite(tmin_tile)[None, :], h // 8, axis=0), B // 8, axis=1)
    plain += a.sum(); beam += b[~sky].sum(); waves += a.size; sky_waves += sky.sum()
print(f"pose {pose} beam tile {B}: wave-iterations plain {plain} ({plain / waves:.1f}/wave), with beam start {beam} ({beam / waves:.1f}/wave); sky waves {sky_waves / waves:.2f}")
# per-level event totals of the last band with the beam start (g_stat of the harness): iter / descend / step / ascend
tot = np.zeros((5, 8), dtype=np.uint64)
L.hh_stat_totals.argtypes = [C.c_void_p]
L.hh_stat_totals(C.c_void_p(tot.ctypes.data))
rays_hit_tiles = (~sky).sum() * 64
for e, name in enumerate(["iter", "descend", "step", "ascend"]):
    print(name, "per ray of the band by level", np.round(tot[e] / (w * h), 2))
