"""Simulates a persistent 'lane refill' wave (lanes that finish pull the next ray of the wave's strip of tiles)
on per-ray event logs from the CPU harness, with a VALU cost model calibrated on the measured kernel."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H
n = 1024
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_trace_primary_events.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + [C.c_void_p]
Wd, Ht = 3840, 2160
cam = W.scene_camera(n, 0, Wd, Ht)
cap = 160
T, D, S, SETUP, OVH = 14, 72, 47, 200, 4

def tile_logs(ty):
    ev = np.zeros((8, Wd, cap), dtype=np.uint8)
    L.hh_trace_primary_events(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, ty * 8, Wd, 8, cap, C.c_void_p(ev.ctypes.data))
    return (ev & 3).reshape(8, Wd // 8, 8, cap).transpose(1, 0, 2, 3).reshape(Wd // 8, 64, cap)

def baseline_cost(tiles):
    isD = tiles == 1; isS = tiles == 2
    n_iter = (isD | isS).any(axis=1).sum(axis=1)
    return (n_iter * T + isD.any(axis=1).sum(axis=1) * D + isS.any(axis=1).sum(axis=1) * S + SETUP).sum()

def refill_cost(tiles, K, refill_min):
    total = 0
    for t0 in range(0, len(tiles) - K + 1, K):
        rays = tiles[t0:t0 + K].reshape(K * 64, cap)
        nxt = 0
        lane_ray = np.full(64, -1); ptr = np.zeros(64, dtype=int)
        while True:
            idle = lane_ray < 0
            n_idle = idle.sum(); remaining = len(rays) - nxt
            if remaining > 0 and (n_idle >= refill_min or n_idle == 64):
                take = min(n_idle, remaining)
                lanes = np.nonzero(idle)[0][:take]
                lane_ray[lanes] = np.arange(nxt, nxt + take); ptr[lanes] = 0; nxt += take
                total += SETUP
            act = lane_ray >= 0
            if not act.any():
                break
            cur = np.where(act, rays[np.maximum(lane_ray, 0), np.minimum(ptr, cap - 1)], 0)
            anyD, anyS = (cur == 1).any(), (cur == 2).any()
            done = act & (cur == 0)
            lane_ray[done] = -1                     # rays with no (more) events finish this round
            if anyD or anyS:
                total += T + OVH + (D if anyD else 0) + (S if anyS else 0)
                ptr[act & (cur != 0)] += 1
        
    return total

rows = [20, 60, 100, 140, 180, 220, 260]
base = 0; res = {}
for ty in rows:
    tiles = tile_logs(ty)[:240]        # half a row to keep the python loop affordable
    base += baseline_cost(tiles)
    for K, m in [(4, 16), (4, 32), (8, 16), (8, 32), (16, 24), (16, 32), (16, 48)]:
        res[(K, m)] = res.get((K, m), 0) + refill_cost(tiles, K, m)
print("baseline VALU:", base)
for k, v in res.items():
    print("K=%d refill_min=%d: %.3f of baseline" % (k[0], k[1], v / base))
