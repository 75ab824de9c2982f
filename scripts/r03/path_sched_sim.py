"""Wave-level schedule simulation of the path kernel on logged per-pixel event sequences (CPU harness, kernel body compiled for
the host), primary rays started behind an emulated beam pre-pass (nearest hit of the 32x32 tile - 2):
 (A) the kernel as it is: one whole walk per round, one KIND of ray per round (primary, else shadow, else bounce), shading between rounds;
 (B) resumable walk + refill: all lanes iterate together; a lane whose walk ended waits; when >= TH lanes wait (or nobody walks) the
     waiting lanes are shaded and given their next ray (any kind), the others keep their walk state."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = 1024
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_render_paths_events2.argtypes = [C.c_void_p] * 3 + [C.c_uint32] * 9 + [C.c_void_p, C.c_uint32, C.c_void_p]
L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
Wd, Ht, cap = 3840, 2160, 320 * spp
cam = W.scene_camera(n, pose, Wd, Ht)
mats = pw.materials
T, D, S = 18, 70, 42

def tile_events(tx, ty):
    # emulated beam: nearest hit of the enclosing 32x32 tile, less 2 voxels
    bx, by = (tx * 8) // 32 * 32, (ty * 8) // 32 * 32
    out = np.zeros(32 * 32, dtype=O.HIT); it = np.zeros(32 * 32, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, bx, by, 32, 32, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf)
    t0 = max(float(t.min()) - 2.0, 0.0) if np.isfinite(t.min()) else 3.0e38
    ts = np.full(64, t0, dtype=np.float32)
    ev = np.zeros((64, cap), dtype=np.uint8)
    L.hh_render_paths_events2(hk.h, C.c_void_p(cam.ctypes.data), C.c_void_p(mats.ctypes.data), len(mats), Wd, Ht, tx * 8, ty * 8, 8, 8,
                              spp, 2, C.c_void_p(ts.ctypes.data), cap, C.c_void_p(ev.ctypes.data))
    lanes = []
    for e in ev:
        e = e[e != 0]
        assert len(e) < cap - 1, "event log overflow"
        starts = np.nonzero((e & 3) == 0)[0]
        rays = [(int(e[a]) >> 3, (e[a + 1:b] & 3).astype(np.uint8)) for a, b in zip(starts, list(starts[1:]) + [len(e)])]
        lanes.append(rays)
    return lanes

def sample_index(lanes):
    # sample index of every ray of a lane: a primary ray (kind 0) starts a sample
    out = []
    for l in lanes:
        s = -1; idx = []
        for k, _ in l:
            if k == 0: s += 1
            idx.append(s)
        out.append(idx)
    return out

def sim_a(lanes, SH, batch_kinds=True, lockstep=False):
    nxt = [0] * 64
    sidx = sample_index(lanes) if lockstep else None
    walk = 0; rounds = 0; lane_iters = 0; wave_iters = 0
    per_kind = [0, 0, 0]
    while True:
        pend = [(i, lanes[i][nxt[i]]) for i in range(64) if nxt[i] < len(lanes[i])]
        if not pend: break
        if lockstep:
            # the oldest (sample, kind) pending in the wave goes first: lanes stay in step sample by sample
            key = min((sidx[i][nxt[i]], r[0]) for i, r in pend)
            k = key[1]
            pend = [(i, r) for i, r in pend if (sidx[i][nxt[i]], r[0]) == key]
        elif batch_kinds:
            kinds = {r[0] for _, r in pend}
            k = 0 if 0 in kinds else (1 if 1 in kinds else 2)
            pend = [(i, r) for i, r in pend if r[0] == k]
        else:
            k = 3
        rounds += 1
        m = max(len(r[1]) for _, r in pend)
        arr = np.zeros((len(pend), m), dtype=np.uint8)
        for j, (_, r) in enumerate(pend): arr[j, :len(r[1])] = r[1]
        anyD = (arr == 1).any(axis=0); anyS = (arr == 2).any(axis=0)
        c = m * T + int(anyD.sum()) * D + int(anyS.sum()) * S
        walk += c; wave_iters += m; lane_iters += int((arr != 0).sum())
        if k < 3: per_kind[k] += c
        for i, _ in pend: nxt[i] += 1
    return dict(cost=walk + rounds * SH, walk=walk, rounds=rounds, util=lane_iters / max(1, 64 * wave_iters), wave_iters=wave_iters, per_kind=per_kind)

def sim_b(lanes, SH, TH):
    # concatenated streams
    lens = [[len(r[1]) for r in l] for l in lanes]
    n_rays = np.array([len(l) for l in lanes])
    stream = [np.concatenate([r[1] for r in l]) if l else np.zeros(0, np.uint8) for l in lanes]
    mx = max(len(s) for s in stream) + 1
    evs = np.zeros((64, mx), dtype=np.uint8)
    for i, s in enumerate(stream): evs[i, :len(s)] = s
    ends = [np.cumsum(x) if x else np.zeros(0, int) for x in lens]
    ray = np.zeros(64, dtype=np.int64); pos = np.zeros(64, dtype=np.int64); end = np.zeros(64, dtype=np.int64)
    waiting = n_rays > 0; done = n_rays == 0; walking = np.zeros(64, dtype=bool)
    cost = 0; shades = 0; wave_iters = 0; lane_iters = 0
    idx = np.arange(64)
    while True:
        nw = int(waiting.sum()); nk = int(walking.sum())
        if nw == 0 and nk == 0: break
        if nw and (nw >= TH or nk == 0):
            cost += SH; shades += 1
            for i in np.nonzero(waiting)[0]:
                if ray[i] >= n_rays[i]: done[i] = True
                else:
                    end[i] = ends[i][ray[i]]; walking[i] = True
                    if pos[i] >= end[i]: pass      # zero-length walk (immediate miss): ends at once below
            waiting[:] = False
            # zero-length walks
            z = walking & (pos >= end)
            if z.any():
                ray[z] += 1; walking[z] = False; waiting[z] = True
            continue
        cur = evs[idx, pos]
        anyD = bool((walking & (cur == 1)).any()); anyS = bool((walking & (cur == 2)).any())
        cost += T + 4 + D * anyD + S * anyS
        wave_iters += 1; lane_iters += nk
        pos[walking] += 1
        fin = walking & (pos >= end)
        if fin.any():
            ray[fin] += 1; walking[fin] = False; waiting[fin] = True
    return dict(cost=cost, shades=shades, util=lane_iters / max(1, 64 * wave_iters), wave_iters=wave_iters)

def sim_d(lanes, SH, TH, RF=120):
    """Lockstep: one primary round per sample, then ONE secondary round in which every lane walks its shadow ray and then its bounce
    ray (both known once the primary hit is shaded), going on to the second when >= TH lanes wait (RF VALU per refill event)."""
    sidx = sample_index(lanes)
    n_s = max((x[-1] + 1) if x else 0 for x in sidx)
    cost = 0; rounds = 0; wave_iters = 0; lane_iters = 0
    for sm in range(n_s):
        rays = [r[1] for i in range(64) for j, r in enumerate(lanes[i]) if sidx[i][j] == sm and r[0] == 0]
        if rays:
            m = max(len(r) for r in rays)
            arr = np.zeros((len(rays), m), dtype=np.uint8)
            for j, r in enumerate(rays): arr[j, :len(r)] = r
            cost += m * T + int((arr == 1).any(axis=0).sum()) * D + int((arr == 2).any(axis=0).sum()) * S + SH
            rounds += 1; wave_iters += m; lane_iters += int((arr != 0).sum())
        queue = [[r[1] for j, r in enumerate(lanes[i]) if sidx[i][j] == sm and r[0] != 0] for i in range(64)]
        if not any(queue): continue
        rounds += 1; cost += SH
        ptr = [0] * 64; pos = [0] * 64
        walking = [len(q) > 0 for q in queue]; waiting = [False] * 64
        while True:
            nk = sum(walking); nw = sum(waiting)
            if nk == 0 and nw == 0: break
            if nw and (nw >= TH or nk == 0):
                cost += RF
                for i in range(64):
                    if waiting[i]: waiting[i] = False; walking[i] = True
                continue
            anyD = anyS = False
            for i in range(64):
                if not walking[i]: continue
                r = queue[i][ptr[i]]
                if pos[i] < len(r):
                    if r[pos[i]] == 1: anyD = True
                    else: anyS = True
                    pos[i] += 1
                if pos[i] >= len(r):
                    ptr[i] += 1; pos[i] = 0; walking[i] = False
                    if ptr[i] < len(queue[i]): waiting[i] = True
            cost += T + 4 + D * anyD + S * anyS; wave_iters += 1; lane_iters += nk
    return dict(cost=cost, rounds=rounds, util=lane_iters / max(1, 64 * wave_iters), wave_iters=wave_iters)

def sim_c(lanes, SH, G, TH, RF=120):
    """Lockstep primary / shadow rounds sample by sample; the bounce rays of G consecutive samples are walked in ONE round in which a
    lane that finished a ray goes on to its next one (refill when >= TH lanes wait, RF VALU per refill event)."""
    sidx = sample_index(lanes)
    n_s = max((x[-1] + 1) if x else 0 for x in sidx)
    cost = 0; rounds = 0; wave_iters = 0; lane_iters = 0
    def round_cost(rays):
        m = max(len(r) for r in rays)
        arr = np.zeros((len(rays), m), dtype=np.uint8)
        for j, r in enumerate(rays): arr[j, :len(r)] = r
        return m * T + int((arr == 1).any(axis=0).sum()) * D + int((arr == 2).any(axis=0).sum()) * S, m, int((arr != 0).sum())
    for g0 in range(0, n_s, G):
        queue = [[] for _ in range(64)]
        for sm in range(g0, min(g0 + G, n_s)):
            for kind in (0, 1):
                rays = [r[1] for i in range(64) for j, r in enumerate(lanes[i]) if sidx[i][j] == sm and r[0] == kind]
                if rays:
                    c, m, li = round_cost(rays); cost += c + SH; rounds += 1; wave_iters += m; lane_iters += li
            for i in range(64):
                for j, r in enumerate(lanes[i]):
                    if sidx[i][j] == sm and r[0] == 2: queue[i].append(r[1])
        if not any(queue): continue
        rounds += 1; cost += SH
        # refill simulation over the queued bounce rays
        ptr = [0] * 64; pos = [0] * 64
        walking = [len(q) > 0 for q in queue]; waiting = [False] * 64
        while True:
            nk = sum(walking); nw = sum(waiting)
            if nk == 0 and nw == 0: break
            if nw and (nw >= TH or nk == 0):
                cost += RF
                for i in range(64):
                    if waiting[i]: waiting[i] = False; walking[i] = True
                continue
            anyD = anyS = False
            for i in range(64):
                if not walking[i]: continue
                r = queue[i][ptr[i]]
                if pos[i] < len(r):
                    if r[pos[i]] == 1: anyD = True
                    else: anyS = True
                    pos[i] += 1
                if pos[i] >= len(r):
                    ptr[i] += 1; pos[i] = 0; walking[i] = False
                    if ptr[i] < len(queue[i]): waiting[i] = True
            cost += T + 4 + D * anyD + S * anyS; wave_iters += 1; lane_iters += nk
    return dict(cost=cost, rounds=rounds, util=lane_iters / max(1, 64 * wave_iters), wave_iters=wave_iters)

rows = {0: (60, 110, 150, 190, 230, 262), 1: (20, 80, 140, 200, 250), 2: (30, 90, 150, 210, 260)}[pose]
tiles = [(tx, ty) for ty in rows for tx in (25, 100, 175, 250, 325, 400, 470)]
logs = []
for tx, ty in tiles:
    l = tile_events(tx, ty)
    if sum(len(x) for x in l) > 64 * spp * 1.2:      # a tile with hits
        logs.append(l)
print(f"pose {pose}, {spp} spp: {len(logs)} walking tiles with hits of {len(tiles)} sampled")
kinds = np.zeros(3); iters = np.zeros(3)
for l in logs:
    for lane in l:
        for k, e in lane: kinds[k] += 1; iters[k] += len(e)
print("rays per pixel and sample by kind (primary, shadow, bounce):", np.round(kinds / (64 * len(logs) * spp), 3), " iterations per ray:", np.round(iters / np.maximum(kinds, 1), 1))
for SH in (500,):
    a = [sim_a(l, SH) for l in logs]
    a0 = [sim_a(l, SH, False) for l in logs]
    al = [sim_a(l, SH, True, True) for l in logs]
    print(f"    lockstep by sample: {np.mean([x['cost'] for x in al]) / np.mean([x['cost'] for x in a]):.2f}x  wave iterations {np.mean([x['wave_iters'] for x in al]):.0f} rounds {np.mean([x['rounds'] for x in al]):.0f}, walk by kind {np.round(np.mean([x['per_kind'] for x in al], axis=0))}, lane util {np.mean([x['util'] for x in al]):.2f}")
    ca = np.mean([x['cost'] for x in a])
    print(f"SHADE={SH}: (A) {ca:9.0f} VALU/wave (walk {np.mean([x['walk'] for x in a]):.0f}, rounds {np.mean([x['rounds'] for x in a]):.0f}, lane util in walks {np.mean([x['util'] for x in a]):.2f}, "
          f"walk VALU by kind {np.round(np.mean([x['per_kind'] for x in a], axis=0))});  without kind batching {np.mean([x['cost'] for x in a0]) / ca:.2f}x")
    for TH in (4, 8, 16):
        d = [sim_d(l, SH, TH) for l in logs]
        print(f"    (D, TH={TH:2d}) {np.mean([x['cost'] for x in d]) / ca:.2f}x  rounds {np.mean([x['rounds'] for x in d]):.0f}, lane util {np.mean([x['util'] for x in d]):.2f}, wave iterations {np.mean([x['wave_iters'] for x in d]):.0f}")
    for G in (4,):
        for TH in (8,):
            c = [sim_c(l, SH, G, TH) for l in logs]
            print(f"    (C, G={G}, TH={TH:2d}) {np.mean([x['cost'] for x in c]) / ca:.2f}x  rounds {np.mean([x['rounds'] for x in c]):.0f}, lane util {np.mean([x['util'] for x in c]):.2f}, wave iterations {np.mean([x['wave_iters'] for x in c]):.0f}")
    for TH in (16, 32):
        b = [sim_b(l, SH, TH) for l in logs]
        print(f"    (B, TH={TH:2d}) {np.mean([x['cost'] for x in b]) / ca:.2f}x  shading rounds {np.mean([x['shades'] for x in b]):.0f}, lane util {np.mean([x['util'] for x in b]):.2f}, wave iterations {np.mean([x['wave_iters'] for x in b]):.0f} vs {np.mean([x['wave_iters'] for x in a]):.0f}")
