"""Wave-level simulation of the path kernel on logged per-pixel event sequences (CPU harness):
 (a) current structure: one whole walk per round (round length = longest ray among the lanes), shading between rounds;
 (b) step-granular interleave: lanes that finished a walk wait; shading runs when >= TH lanes wait or nobody walks."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H
n = 1024
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_render_paths_events.argtypes = [C.c_void_p] * 3 + [C.c_uint32] * 10 + [C.c_void_p]
Wd, Ht, spp, cap = 3840, 2160, 8, 1536
cam = W.scene_camera(n, 0, Wd, Ht)
mats = pw.materials
T, D, S = 14, 72, 47

def tile_events(tx, ty):
    ev = np.zeros((8, 8, cap), dtype=np.uint8)
    L.hh_render_paths_events(hk.h, C.c_void_p(cam.ctypes.data), C.c_void_p(mats.ctypes.data), len(mats), Wd, Ht, tx * 8, ty * 8, 8, 8,
                             spp, 2, cap, C.c_void_p(ev.ctypes.data))
    lanes = []
    for e in (ev & 7).reshape(64, cap):
        e = e[e != 0]
        starts = np.nonzero(e == 4)[0]
        rays = [e[a + 1:b] for a, b in zip(starts, list(starts[1:]) + [len(e)])]
        lanes.append(rays)
    return lanes

def sim_a(lanes, SH):
    cost = 0; rounds = 0; walk = 0
    k = 0
    while True:
        cur = [l[k] for l in lanes if k < len(l)]
        if not cur: break
        rounds += 1
        m = max(len(r) for r in cur)
        for i in range(m):
            anyD = any(i < len(r) and r[i] == 1 for r in cur); anyS = any(i < len(r) and r[i] == 2 for r in cur)
            walk += T + D * anyD + S * anyS
        k += 1
    return walk + rounds * SH, rounds, walk

def sim_b(lanes, SH, TH):
    ray = [0] * 64; pos = [0] * 64
    waiting = [True] * 64          # need (initial) ray setup / shading
    done = [len(l) == 0 for l in lanes]
    cost = 0
    while True:
        n_wait = sum(1 for i in range(64) if waiting[i] and not done[i])
        n_walk = sum(1 for i in range(64) if not waiting[i] and not done[i])
        if n_wait == 0 and n_walk == 0: break
        if n_wait and (n_wait >= TH or n_walk == 0):
            cost += SH
            for i in range(64):
                if waiting[i] and not done[i]:
                    if ray[i] >= len(lanes[i]): done[i] = True
                    else: waiting[i] = False; pos[i] = 0
            continue
        anyD = anyS = False
        for i in range(64):
            if waiting[i] or done[i]: continue
            r = lanes[i][ray[i]]
            if pos[i] < len(r):
                if r[pos[i]] == 1: anyD = True
                else: anyS = True
                pos[i] += 1
            if pos[i] >= len(r):
                waiting[i] = True; ray[i] += 1
        cost += T + 3 + D * anyD + S * anyS
    return cost

tiles = [(tx, ty) for ty in (60, 110, 150, 190, 230) for tx in (40, 160, 300, 420)]
res_a = []; tot_rounds = 0; tot_walk = 0
logs = [tile_events(tx, ty) for tx, ty in tiles]
for SH in (300, 500, 800):
    a = [sim_a(l, SH) for l in logs]
    line = f"SHADE={SH}: (a) {np.mean([x[0] for x in a]):8.0f} VALU/wave (walk {np.mean([x[2] for x in a]):.0f}, rounds {np.mean([x[1] for x in a]):.1f})"
    for TH in (8, 16, 32):
        b = [sim_b(l, SH, TH) for l in logs]
        line += f" | (b,TH={TH}) {np.mean(b) / np.mean([x[0] for x in a]):.2f}x"
    print(line)
