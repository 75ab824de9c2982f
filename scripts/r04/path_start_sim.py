"""What do a wave tile's own start parameter and walk_resume do to the path loop's walks?  Kernel body on the CPU (tests/host_harness), 4K over
1024^3, sampled 8x8-pixel wave tiles, primary rays behind an emulated beam (nearest pixel-centre hit of the B x B tile grown by a pixel, less
2 voxels).  Per kind of ray: iterations per walk; and the lockstep wave cost model of scripts/r03/path_sched_sim.py (T + D anyD + S anyS per
wave iteration, P per launch-pad level)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = 1024
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_render_paths_events2.argtypes = [C.c_void_p] * 3 + [C.c_uint32] * 9 + [C.c_void_p, C.c_uint32, C.c_void_p]
L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
L.hh_set_path_resume.argtypes = [C.c_uint32]
Wd, Ht, cap = 3840, 2160, 400 * spp
cam = W.scene_camera(n, pose, Wd, Ht)
mats = pw.materials
T, D, S, P = 18, 70, 42, 25

def beam_t0(tx, ty, B):
    bx, by = (tx * 8) // B * B, (ty * 8) // B * B
    x0, y0 = max(bx - 1, 0), max(by - 1, 0); w, h = min(bx + B + 1, Wd) - x0, min(by + B + 1, Ht) - y0
    out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, x0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf)
    return max(float(t.min()) - 2.0, 0.0) if np.isfinite(t.min()) else 3.0e38

def fine_t0(tx, ty, B):
    """per-lane start parameters of the 8x8 wave tile from B x B-pixel sub-tiles (B < 8), each grown by a pixel like a beam tile"""
    x0, y0 = max(tx * 8 - 1, 0), max(ty * 8 - 1, 0); w, h = min(tx * 8 + 9, Wd) - x0, min(ty * 8 + 9, Ht) - y0
    out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, x0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf).reshape(h, w)
    ts = np.zeros((8, 8), dtype=np.float32)
    for sy in range(0, 8, B):
        for sx in range(0, 8, B):
            ax, ay = tx * 8 + sx - x0, ty * 8 + sy - y0
            m = t[max(ay - 1, 0):ay + B + 1, max(ax - 1, 0):ax + B + 1].min()
            ts[sy:sy + B, sx:sx + B] = max(float(m) - 2.0, 0.0) if np.isfinite(m) else 3.0e38
    return ts.reshape(-1)

def tile_events(tx, ty, B, resume):
    L.hh_set_path_resume(resume)
    ts = np.full(64, beam_t0(tx, ty, B), dtype=np.float32) if B >= 8 else np.maximum(fine_t0(tx, ty, B), np.float32(beam_t0(tx, ty, 8)))
    ev = np.zeros((64, cap), dtype=np.uint8)
    L.hh_render_paths_events2(hk.h, C.c_void_p(cam.ctypes.data), C.c_void_p(mats.ctypes.data), len(mats), Wd, Ht, tx * 8, ty * 8, 8, 8,
                              spp, 2, C.c_void_p(ts.ctypes.data), cap, C.c_void_p(ev.ctypes.data))
    lanes = []
    for e in ev:
        e = e[e != 0]
        assert len(e) < cap - 1, "event log overflow"
        starts = np.nonzero((e & 3) == 0)[0]
        lanes.append([(int(e[a]) >> 3, (e[a + 1:b] & 3).astype(np.uint8)) for a, b in zip(starts, list(starts[1:]) + [len(e)])])
    return lanes

def lockstep_cost(lanes):
    """rounds keyed by (sample, kind); per round: pad events first (all lanes together), then the loop's iterations"""
    nxt = [0] * 64; sidx = []
    for l in lanes:
        s = -1; idx = []
        for k, _ in l:
            if k == 0: s += 1
            idx.append(s)
        sidx.append(idx)
    cost = np.zeros(3); iters = np.zeros(3); lane_iters = np.zeros(3); rays = np.zeros(3)
    while True:
        pend = [(i, lanes[i][nxt[i]]) for i in range(64) if nxt[i] < len(lanes[i])]
        if not pend: break
        key = min((sidx[i][nxt[i]], r[0]) for i, r in pend)
        cur = [(i, r) for i, r in pend if (sidx[i][nxt[i]], r[0]) == key]
        k = key[1]
        seqs = [r[1] for _, r in cur]
        pads = max((int((q == 3).sum()) for q in seqs), default=0)
        loops = [q[q != 3] for q in seqs]
        m = max((len(q) for q in loops), default=0)
        arr = np.zeros((len(loops), max(m, 1)), dtype=np.uint8)
        for j, q in enumerate(loops): arr[j, :len(q)] = q
        cost[k] += pads * P + m * T + int((arr == 1).any(axis=0).sum()) * D + int((arr == 2).any(axis=0).sum()) * S
        iters[k] += m; lane_iters[k] += int((arr != 0).sum()); rays[k] += len(cur)
        for i, _ in cur: nxt[i] += 1
    return cost, iters, lane_iters, rays

rows = {0: (60, 110, 150, 190, 230, 262), 1: (20, 80, 140, 200, 250), 2: (30, 90, 150, 210, 260)}[pose]
tiles = [(tx, ty) for ty in rows for tx in (25, 100, 175, 250, 325, 400, 470)]
live = [t for t in tiles if beam_t0(t[0], t[1], 8) < 1e38]
print(f"pose {'ABC'[pose]}, {spp} spp: {len(live)} live wave tiles of {len(tiles)} sampled")
base = None
for name, B, resume in (("beam 32, from the root (round 3)", 32, 0), ("beam 8, from the root", 8, 0), ("beam 32, resumed", 32, 1), ("beam 8, resumed", 8, 1),
                        ("beam 4, from the root", 4, 0), ("beam 4, resumed", 4, 1), ("beam 2, from the root", 2, 0), ("beam 2, resumed", 2, 1), ("beam 1, from the root", 1, 0), ("beam 1, resumed", 1, 1)):
    tot = np.zeros(3); it = np.zeros(3); li = np.zeros(3); ry = np.zeros(3)
    for tx, ty in live:
        c, i, l, r = lockstep_cost(tile_events(tx, ty, B, resume)); tot += c; it += i; li += l; ry += r
    if base is None: base = tot.sum()
    print(f"  {name:34s} model VALU by kind (primary, shadow, bounce) {np.round(tot / len(live) / spp).astype(int)} per sample and wave, total {tot.sum() / base:.3f}x;"
          f" lane iterations per ray {np.round(li / np.maximum(ry, 1), 1)}; wave iterations per round {np.round(it / (len(live) * spp), 1)}")
L.hh_set_path_resume(0)
