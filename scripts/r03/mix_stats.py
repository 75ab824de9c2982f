"""Primary rays behind an emulated beam pre-pass: per wave-iteration, how often only the descend path, only the step path or both are live."""
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
L.hh_trace_rect_events.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p, C.c_uint32, C.c_void_p]
Wd, Ht, B, cap = 3840, 2160, 32, 260
cam = W.scene_camera(n, pose, Wd, Ht)
tot = dict(waves=0, iters=0, pureD=0, pureS=0, mixed=0, laneD=0, laneS=0, lanes_in_mixed_D=0, lanes_in_mixed_S=0, first_pureD=0)
lvl_hist = np.zeros((2, 8))
for y0 in range(44, Ht - B + 1, 216):
    w, h = Wd, B
    out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf).reshape(h, w // B, B)
    tmin_tile = t.min(axis=(0, 2))
    ts = np.repeat(np.maximum(tmin_tile - 2.0, 0)[None, :], h, axis=0).repeat(B, axis=1)
    tsf = np.where(np.isfinite(ts), ts, 9999.0).astype(np.float32)
    ev = np.zeros((h, w, cap), dtype=np.uint8)
    L.hh_trace_rect_events(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, C.c_void_p(tsf.ctypes.data), cap, C.c_void_p(ev.ctypes.data))
    full = ev[..., 1:]
    kind = full & 3
    live_tile = np.isfinite(tmin_tile)
    tiles = kind.reshape(h // 8, 8, w // 8, 8, cap - 1).transpose(0, 2, 1, 3, 4).reshape(h // 8, w // 8, 64, cap - 1)
    lv = (full >> 2).reshape(h // 8, 8, w // 8, 8, cap - 1).transpose(0, 2, 1, 3, 4).reshape(h // 8, w // 8, 64, cap - 1)
    for ty in range(h // 8):
        for tx in range(w // 8):
            if not live_tile[tx * 8 // B]: continue
            seqs = tiles[ty, tx]
            if not seqs.any(): continue
            m = int((seqs != 0).sum(axis=1).max())
            s = seqs[:, :m]
            nD = (s == 1).sum(axis=0); nS = (s == 2).sum(axis=0)
            tot['waves'] += 1; tot['iters'] += m
            tot['pureD'] += int(((nD > 0) & (nS == 0)).sum()); tot['pureS'] += int(((nS > 0) & (nD == 0)).sum())
            mix = (nD > 0) & (nS > 0)
            tot['mixed'] += int(mix.sum()); tot['laneD'] += int(nD.sum()); tot['laneS'] += int(nS.sum())
            tot['lanes_in_mixed_D'] += int(nD[mix].sum()); tot['lanes_in_mixed_S'] += int(nS[mix].sum())
            l = lv[ty, tx][:, :m]
            for k in (1, 2):
                sel = s == k
                lvl_hist[k - 1] += np.bincount(l[sel].ravel(), minlength=8)[:8]
wv = tot['waves']
print(f"pose {pose}: {wv} walking waves; per wave: iterations {tot['iters'] / wv:.1f} = pure descend {tot['pureD'] / wv:.1f} + pure step {tot['pureS'] / wv:.1f} + mixed {tot['mixed'] / wv:.1f}")
print(f"  lane events per wave: descend {tot['laneD'] / wv:.0f}, step {tot['laneS'] / wv:.0f}; in mixed iterations: {tot['lanes_in_mixed_D'] / max(1, tot['mixed']):.1f} lanes descend, {tot['lanes_in_mixed_S'] / max(1, tot['mixed']):.1f} step")
print("  descends by level (of the cell entered from):", np.round(lvl_hist[0] / wv, 1), " steps by level:", np.round(lvl_hist[1] / wv, 1))
T, D, S = 18, 70, 42
cur = tot['iters'] * T + (tot['pureD'] + tot['mixed']) * D + (tot['pureS'] + tot['mixed']) * S
print(f"  model loop VALU per wave: {cur / wv:.0f}; perfectly packed {(tot['laneD'] * (T + D) + tot['laneS'] * (T + S)) / 64 / wv:.0f}")
