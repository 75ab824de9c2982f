"""Two-pass estimate: pass 1 caps every wave at K walk iterations and queues unfinished rays; pass 2 walks the queued
rays packed 64 to a wave.  Wave-iterations (cost proxy) from per-ray iteration counts of the CPU harness, with the
per-tile start parameter of the beam pre-pass (nearest hit of the 32x32 tile - 2)."""
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
per_wave = []        # arrays of 64 per-lane iteration counts for walking waves
for y0 in range(0, Ht - B + 1, 108):
    w, h = Wd, B
    def run(ts):
        out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
        L.hh_trace_rect_stats2(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, None if ts is None else C.c_void_p(ts.ctypes.data), None,
                               C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
        return out, it
    out, _ = run(None)
    hit = (out['hit'] == 1).reshape(h, w // B, B); t = out['t'].reshape(h, w // B, B)
    tmin = np.where(hit, t, np.inf).min(axis=(0, 2))
    ts = np.repeat(np.maximum(tmin - 2, 0)[None, :], h, axis=0).repeat(B, axis=1)
    sky = ~np.isfinite(ts)
    tsf = np.where(sky, 9999.0, ts).astype(np.float32)
    _, it = run(tsf)
    it = it.reshape(h, w).astype(np.int64); it[sky] = 0
    tiles = it.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    per_wave.append(tiles[tiles.max(axis=1) > 0])
pw_ = np.concatenate(per_wave)
base = pw_.max(axis=1).sum()
print(f"pose {pose}: {len(pw_)} walking waves, lane mean {pw_.mean():.1f}, wave max mean {pw_.max(axis=1).mean():.1f} -> utilisation {pw_.mean() / pw_.max(axis=1).mean():.2f}")
rng = np.random.default_rng(0)
for K in (6, 8, 10, 12, 16, 20, 24):
    p1 = np.minimum(pw_.max(axis=1), K).sum()
    rest = (pw_ - K)[pw_ > K]
    rest = rest[rng.permutation(len(rest))]                      # queue order is arbitrary
    pad = (-len(rest)) % 64
    packed = np.concatenate([rest, np.zeros(pad, dtype=rest.dtype)]).reshape(-1, 64)
    p2 = packed.max(axis=1).sum()
    srt = np.sort(rest)[::-1]; srt = np.concatenate([srt, np.zeros(pad, dtype=srt.dtype)]).reshape(-1, 64)
    print(f"  K={K:2d}: pass 1 {p1 / base:.2f} + pass 2 {p2 / base:.2f} (sorted queue {srt.max(axis=1).sum() / base:.2f}) = {(p1 + p2) / base:.2f} of the one-pass cost; {len(rest) / pw_.size * 100:.1f} % of the rays are queued")
