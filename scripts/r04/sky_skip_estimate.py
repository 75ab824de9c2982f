"""What would a height map give the path loop's bounce rays?  Estimate on the CPU (tests/host_harness) over a 512^3 scene: bounce rays from the
primary hits of sampled 8x8 wave tiles (cosine-weighted directions, numpy's generator: statistically the kernel's), walked (a) from the root as
now, (b) with tmin raised to where a march over the columns' max heights (a hundredth of a voxel of margin) first finds the ray at
or below a column's top, or dropped when the march leaves the world.  Reports iterations per ray and the wave's round length (max over its lanes).
    python3 scripts/r04/sky_skip_estimate.py [pose=0]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = 512
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dense = W.scene_dense(n)                                   # [z][y][x]
occ = dense != 0
ys = np.arange(1, n + 1, dtype=np.int32)[None, :, None]
height = (occ * ys).max(axis=1).astype(np.int32)           # [z][x]: 1 + highest filled y, 0 = empty column
del dense
hd = height.copy()                                         # dilated by one column
for dz in (-1, 0, 1):
    for dx in (-1, 0, 1):
        hd = np.maximum(hd, np.roll(np.roll(height, dz, axis=0), dx, axis=1))
hmax = int(height.max())
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
RAY = np.dtype([('org', '<f4', 3), ('tmin', '<f4'), ('dir', '<f4', 3), ('tmax', '<f4')])
L.hh_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
L.hh_stat_totals8.argtypes = [C.c_void_p]
L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
Wd, Ht = 3840, 2160
cam = W.scene_camera(n, pose, Wd, Ht)
rng = np.random.default_rng(7)

def iterations(rays):
    """per-ray loop iterations of the kernel's walk, and hit flags"""
    out = np.zeros(len(rays), dtype=O.HIT); its = np.zeros(len(rays), dtype=np.int64)
    tot = np.zeros((8, 8), dtype=np.uint64)
    for i in range(len(rays)):
        L.hh_stat_reset()
        L.hh_trace_rays(hk.h, C.c_void_p(rays[i:i + 1].ctypes.data), 1, C.c_void_p(out[i:i + 1].ctypes.data))
        L.hh_stat_totals8(C.c_void_p(tot.ctypes.data))
        its[i] = int(tot[0].sum())
    return its, out['hit'] == 1

def skip(org, d):
    """march over columns (finest level, one at a time: the estimate is of the bound, not of the march's cost): t at which the ray is first
    at or below a dilated column's top + 1, or None when it leaves the world first"""
    ox, oy, oz = org; dx, dy, dz = d
    t = 0.0
    x, z = int(np.floor(ox)), int(np.floor(oz))
    sx, sz = (1 if dx > 0 else -1), (1 if dz > 0 else -1)
    tx = ((x + (sx > 0)) - ox) / dx if abs(dx) > 1e-9 else np.inf
    tz = ((z + (sz > 0)) - oz) / dz if abs(dz) > 1e-9 else np.inf
    steps = 0
    while True:
        if not (0 <= x < n and 0 <= z < n): return None, steps
        t_out = min(tx, tz)
        y_lo = min(oy + dy * (t - 1e-3), oy + dy * (t_out + 1e-3)) if np.isfinite(t_out) else oy + dy * t
        if oy + dy * t > hmax + 1 and dy >= 0: return None, steps
        if y_lo < height[z, x] + 0.01: return max(t - 0.01, 0.0), steps
        if not np.isfinite(t_out): return None, steps
        if tx <= tz: t = tx; x += sx; tx += abs(1.0 / dx)
        else: t = tz; z += sz; tz += abs(1.0 / dz)
        steps += 1

rows = {0: (60, 110, 150, 190, 230, 262), 1: (20, 80, 140, 200, 250), 2: (30, 90, 150, 210, 260)}[pose]
tiles = [(tx, ty) for ty in rows for tx in (25, 100, 175, 250, 325, 400, 470)]
tot = dict(rays=0, base=0, skipped=0, round_base=0, round_skip=0, rounds=0, dropped=0, march=0, hit_base=0, march_max=0)
for tx, ty in tiles:
    out = np.zeros(64, dtype=O.HIT); it = np.zeros(64, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, tx * 8, ty * 8, 8, 8, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    hit = out['hit'] == 1
    if not hit.any(): continue
    # primary rays of the tile, to get hit points
    prim = np.zeros(64, dtype=RAY)
    from blok_amd import _ffi
    for rep in range(4):                                   # four samples' worth of bounce rays per tile
        rays = []; lanes = []
        for i in np.nonzero(hit)[0]:
            px, py = tx * 8 + i % 8, ty * 8 + i // 8
            r = W.primary_ray(cam, px, py, Wd, Ht) if hasattr(W, 'primary_ray') else None
            if r is None: break
        # hit point from the record: voxel + face (no need for the ray): a point on the face's centre
        for i in np.nonzero(hit)[0]:
            v = out['voxel'][i].astype(np.float64) if 'voxel' in out.dtype.names else None
            if v is None: break
            f = int(out['face'][i])
            nrm = np.zeros(3); nrm[f // 2] = 1.0 if f % 2 == 0 else -1.0
            p = v + 0.5 + nrm * 0.5 + nrm * 0.002
            p[(f // 2 + 1) % 3] += rng.uniform(-0.45, 0.45); p[(f // 2 + 2) % 3] += rng.uniform(-0.45, 0.45)
            u1, u2 = rng.uniform(), rng.uniform()
            r_, phi = np.sqrt(u1), 2 * np.pi * u2
            up = np.array([0.0, 0.0, 1.0]) if abs(nrm[2]) < 0.999 else np.array([1.0, 0.0, 0.0])
            t_ = np.cross(up, nrm); b_ = np.cross(nrm, t_)
            dvec = t_ * r_ * np.cos(phi) + b_ * r_ * np.sin(phi) + nrm * np.sqrt(max(0.0, 1 - u1))
            dvec /= np.linalg.norm(dvec)
            rays.append((p, dvec)); lanes.append(i)
        if not rays: continue
        base = np.zeros(len(rays), dtype=RAY); skp = np.zeros(len(rays), dtype=RAY)
        dropped = np.zeros(len(rays), dtype=bool); march = np.zeros(len(rays), dtype=np.int64)
        for k, (p, dvec) in enumerate(rays):
            base[k] = (p, 0.001, dvec, 10000.0)
            t0, steps = skip(p, dvec); march[k] = steps
            if t0 is None: dropped[k] = True; skp[k] = (p, 0.001, dvec, 0.0)
            else: skp[k] = (p, max(0.001, t0), dvec, 10000.0)
        ib, hb = iterations(base); isk, hs = iterations(skp)
        assert (hb == hs).all(), "a skipped ray changed its answer"
        tot['rays'] += len(rays); tot['base'] += ib.sum(); tot['skipped'] += isk.sum(); tot['dropped'] += dropped.sum(); tot['march'] += march.sum()
        tot['round_base'] += ib.max(); tot['round_skip'] += isk.max(); tot['rounds'] += 1; tot['hit_base'] += hb.sum(); tot['march_max'] += march.max()
print(f"pose {'ABC'[pose]} at {n}^3: {tot['rays']} bounce rays in {tot['rounds']} rounds, {tot['hit_base'] / tot['rays']:.2f} hit")
print(f"  iterations per ray: {tot['base'] / tot['rays']:.1f} from the root, {tot['skipped'] / tot['rays']:.1f} behind the height map's bound ({tot['dropped'] / tot['rays']:.2f} of the rays dropped outright)")
print(f"  wave iterations per round: {tot['round_base'] / tot['rounds']:.1f} -> {tot['round_skip'] / tot['rounds']:.1f}; columns marched per ray {tot['march'] / tot['rays']:.0f}, longest of a round {tot['march_max'] / tot['rounds']:.0f} (finest level only: a pyramid takes far fewer)")
