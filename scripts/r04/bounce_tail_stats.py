"""What are a bounce round's iterations spent on?  Kernel body on the CPU (tests/host_harness), 4K over 1024^3, sampled 8x8-pixel wave
tiles, 2 bounces.  Per bounce ray the event log (descend / step with its level); a ray whose last event is a step at a level >= 2 left
the world (a miss), any other ended in a voxel.  Printed: share of misses, iterations of either, iterations by level, the "coarse tail"
of a ray (its iterations after the last one below level 2), and the wave's round length with the tails cut off (what a conservative
look-ahead over coarse cells could save at best)."""
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
Wd, Ht, cap = 3840, 2160, 400 * spp
cam = W.scene_camera(n, pose, Wd, Ht)
mats = pw.materials

def beam_t0(tx, ty, B=8):
    bx, by = (tx * 8) // B * B, (ty * 8) // B * B
    x0, y0 = max(bx - 1, 0), max(by - 1, 0); w, h = min(bx + B + 1, Wd) - x0, min(by + B + 1, Ht) - y0
    out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, x0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf)
    return max(float(t.min()) - 2.0, 0.0) if np.isfinite(t.min()) else 3.0e38

def tile_rays(tx, ty):
    ts = np.full(64, beam_t0(tx, ty), dtype=np.float32)
    ev = np.zeros((64, cap), dtype=np.uint8)
    L.hh_render_paths_events2(hk.h, C.c_void_p(cam.ctypes.data), C.c_void_p(mats.ctypes.data), len(mats), Wd, Ht, tx * 8, ty * 8, 8, 8,
                              spp, 2, C.c_void_p(ts.ctypes.data), cap, C.c_void_p(ev.ctypes.data))
    lanes = []
    for e in ev:
        e = e[e != 0]
        starts = np.nonzero((e & 3) == 0)[0]
        rays = []; s = -1
        for a, b in zip(starts, list(starts[1:]) + [len(e)]):
            k = int(e[a]) >> 3
            if k == 0: s += 1
            rays.append((s, k, e[a + 1:b]))
        lanes.append(rays)
    return lanes

rows = {0: (60, 110, 150, 190, 230, 262), 1: (20, 80, 140, 200, 250), 2: (30, 90, 150, 210, 260)}[pose]
tiles = [(tx, ty) for ty in rows for tx in (25, 100, 175, 250, 325, 400, 470)]
n_rays = n_miss = 0; it_miss = []; it_hit = []; lvl_hist = np.zeros(8); tail_miss = []; fine_miss = []
round_full = round_cut = round_hits_only = 0; rounds = 0
for tx, ty in tiles:
    if beam_t0(tx, ty) > 1e38: continue
    lanes = tile_rays(tx, ty)
    per_round = {}
    for lane in lanes:
        for s, k, q in lane:
            if k != 2: continue
            ev = q & 3; lv = q >> 2
            keep = ev != 3
            ev, lv = ev[keep], lv[keep]
            m = len(ev)
            miss = m > 0 and ev[-1] == 2 and lv[-1] >= 2
            n_rays += 1; n_miss += miss
            for l in lv: lvl_hist[l] += 1
            fine = np.nonzero(lv < 2)[0]
            last_fine = int(fine[-1]) + 1 if len(fine) else 0
            if miss:
                it_miss.append(m); tail_miss.append(m - last_fine); fine_miss.append(last_fine)
            else:
                it_hit.append(m)
            per_round.setdefault(s, []).append((m, last_fine if miss else m, miss))
    for s, rs in per_round.items():
        rounds += 1
        round_full += max(r[0] for r in rs)
        round_cut += max(r[1] for r in rs)
        round_hits_only += max([r[0] for r in rs if not r[2]], default=0)
it_miss, it_hit, tail_miss, fine_miss = map(np.array, (it_miss, it_hit, tail_miss, fine_miss))
print(f"pose {'ABC'[pose]}, {spp} spp: {n_rays} bounce rays in {rounds} rounds; misses {n_miss / n_rays:.3f}")
print(f"  iterations per ray: misses {it_miss.mean():.1f} (p90 {np.percentile(it_miss, 90):.0f}), hits {it_hit.mean():.1f} (p90 {np.percentile(it_hit, 90):.0f})")
print(f"  iterations by level 0..5: {np.round(lvl_hist[:6] / lvl_hist.sum(), 3)}")
print(f"  a miss: {fine_miss.mean():.1f} iterations up to its last one below level 2, {tail_miss.mean():.1f} after it (coarse tail)")
print(f"  wave iterations per round: {round_full / rounds:.1f}; with every miss's coarse tail cut {round_cut / rounds:.1f}; hits alone {round_hits_only / rounds:.1f}")
