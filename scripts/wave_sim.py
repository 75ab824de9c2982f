"""Wave-level cost model from per-ray event logs (CPU harness): how often descend/step paths are live per
wave-iteration, lane utilisation, and what alternative schedules would cost."""
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
cam = W.scene_camera(n, int(sys.argv[1]) if len(sys.argv) > 1 else 0, Wd, Ht)
cap = 192
x0, y0, w, h = 0, 0, 3840, 2160
step = 4   # sample every 4th tile row to save time
tot = dict(iters=0, anyD=0, anyS=0, laneD=0, laneS=0, waves=0, lanes_active=0, maxD=0, maxS=0)
for ty in range(0, h // 8, step):
    ev = np.zeros((8, w, cap), dtype=np.uint8)
    L.hh_trace_primary_events(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, x0, y0 + ty * 8, w, 8, cap, C.c_void_p(ev.ctypes.data))
    kind = ev & 3
    tiles = kind.reshape(8, w // 8, 8, cap).transpose(1, 0, 2, 3).reshape(w // 8, 64, cap)   # (tile, lane, iter)
    isD = tiles == 1; isS = tiles == 2
    live = isD | isS
    n_iter = live.any(axis=1).sum(axis=1)
    tot['iters'] += n_iter.sum(); tot['waves'] += len(n_iter)
    tot['anyD'] += isD.any(axis=1).sum(); tot['anyS'] += isS.any(axis=1).sum()
    tot['laneD'] += isD.sum(); tot['laneS'] += isS.sum()
    tot['maxD'] += isD.sum(axis=2).max(axis=1).sum(); tot['maxS'] += isS.sum(axis=2).max(axis=1).sum()
wv = tot['waves']
print({k: (v / wv) for k, v in tot.items() if k != 'waves'})
T, D, S = 18, 72, 47
cur = (tot['iters'] * T + tot['anyD'] * D + tot['anyS'] * S) / wv
ideal = (tot['laneD'] * (T + D) + tot['laneS'] * (T + S)) / 64 / wv
prio = (tot['maxD'] * (T + D) + tot['maxS'] * (T + S)) / wv
print(f"model VALU/wave loop: current {cur:.0f}, perfectly packed {ideal:.0f}, phase-separated lower bound {prio:.0f}")

# ---- schedule simulation on a subset of tiles
def simulate(seqs, policy, T=18, D=72, S=47):
    """seqs: (64, cap) uint8 of 1 (descend) / 2 (step) / 0 (end).  Returns VALU cost of the loop for this wave."""
    ptr = np.zeros(64, dtype=np.int64)
    cost = 0
    cap = seqs.shape[1]
    rounds = 0
    while True:
        nxt = np.where(ptr < cap, seqs[np.arange(64), np.minimum(ptr, cap - 1)], 0)
        wantD = nxt == 1; wantS = nxt == 2
        nd, ns = wantD.sum(), wantS.sum()
        if nd + ns == 0:
            break
        rounds += 1
        if policy == "both":
            doD, doS = nd > 0, ns > 0
        elif policy == "S_first":
            doS = ns > 0; doD = not doS
        elif policy == "D_first":
            doD = nd > 0; doS = not doD
        elif policy == "majority":
            doD = nd >= ns; doS = not doD
        elif policy.startswith("thresh"):
            th = int(policy[6:])
            # run a path only if at least th lanes want it, unless nothing else can run
            doD = nd >= th; doS = ns >= th
            if not doD and not doS:
                doD = nd >= ns; doS = not doD
        cost += T + (D if doD else 0) + (S if doS else 0)
        if doD: ptr[wantD] += 1
        if doS: ptr[wantS] += 1
    return cost, rounds

rng = np.random.default_rng(0)
ev = np.zeros((8, w, cap), dtype=np.uint8)
res = {}
for ty in (40, 90, 135, 180, 230):
    L.hh_trace_primary_events(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, ty * 8, w, 8, cap, C.c_void_p(ev.ctypes.data))
    tiles = (ev & 3).reshape(8, w // 8, 8, cap).transpose(1, 0, 2, 3).reshape(w // 8, 64, cap)
    for t in range(0, w // 8, 3):
        for pol in ("both", "S_first", "D_first", "majority", "thresh8", "thresh16", "thresh24"):
            c, r = simulate(tiles[t], pol)
            a = res.setdefault(pol, [0, 0, 0]); a[0] += c; a[1] += r; a[2] += 1
for pol, (c, r, k) in res.items():
    print(f"{pol:10s} VALU/wave {c / k:7.0f}  rounds {r / k:5.1f}")
