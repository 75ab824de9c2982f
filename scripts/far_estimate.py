"""Optimistic estimate of what a per-tile far bound (tmax' = farthest visible hit of the 32x32 tile + 2) would save on top of
the near bound: wave-max walk iterations per 8x8 wave on bands of the 4K frame."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = 1024; pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_trace_rect_stats2.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 4
Wd, Ht, B = 3840, 2160, 32
cam = W.scene_camera(n, pose, Wd, Ht)
tot = dict(plain=0, near=0, both=0); waves = 0
for y0 in range(0, Ht - B + 1, 216):
    w, h = Wd, B
    def run(ts, tf):
        out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
        L.hh_trace_rect_stats2(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, None if ts is None else C.c_void_p(ts.ctypes.data),
                               None if tf is None else C.c_void_p(tf.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
        return out, it.reshape(h // 8, 8, w // 8, 8).max(axis=(1, 3))
    out, a = run(None, None)
    hit = (out['hit'] == 1).reshape(h, w // B, B)
    t = out['t'].reshape(h, w // B, B)
    tmin = np.where(hit, t, np.inf).min(axis=(0, 2)); tmax = np.where(hit, t, -np.inf).max(axis=(0, 2))
    ts = np.repeat(np.maximum(tmin - 2, 0)[None, :], h, axis=0).repeat(B, axis=1)
    ts = np.where(np.isfinite(ts), ts, 9999.0).astype(np.float32)
    tf = np.repeat((tmax + 2)[None, :], h, axis=0).repeat(B, axis=1)
    tf = np.where(np.isfinite(tf), tf, 0.0).astype(np.float32)
    out1, b = run(ts, None)
    out2, c = run(ts, tf)
    assert np.array_equal(out.view(np.uint8), out1.view(np.uint8)) and np.array_equal(out.view(np.uint8), out2.view(np.uint8))
    tot['plain'] += a.sum(); tot['near'] += b.sum(); tot['both'] += c.sum(); waves += a.size
print(f"pose {pose}: wave-max iterations per wave: plain {tot['plain'] / waves:.2f}, near bound {tot['near'] / waves:.2f}, near + far bound {tot['both'] / waves:.2f}")
