"""Is the previous frame's measured cost a good enough order for THIS frame when the camera moves?  4K over 1024^3, camera orbiting the
world's centre by `step` degrees per frame (bench.py --orbit): the wave tiles of frame k+1 are walked (blok_hip_trace_wave_tiles_device,
walk kernel alone) in the order of
  natural       row-major
  own clocks    their own measured clocks, descending (the ideal, unknowable before the walk)
  stale         the clocks frame k measured at the same screen position
  reprojected   the clocks frame k measured where the tile's point at its start parameter was on frame k's screen
each also quantised to 8 classes (powers of 1.5 above the median: what a list with classes would give)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
steps = [float(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1.0, 4.0]
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda")
bx, by, gx = Wd // 8, Ht // 8, (Wd + 31) // 32
clocks = torch.zeros(bx * by, dtype=torch.int32, device="cuda")

def orbit_cam(deg):
    nf = float(n)
    centre = np.array([0.5 * nf, 0.25 * nf, 0.5 * nf]); start = np.array([-0.35 * nf, 0.85 * nf, -0.35 * nf]) - centre
    a = np.radians(deg)
    p = centre + np.array([start[0] * np.cos(a) - start[2] * np.sin(a), start[1], start[0] * np.sin(a) + start[2] * np.cos(a)])
    return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), 60.0, Wd, Ht)

def live_tiles(cam):
    t0, _ = tr.beam_prepass(cam)
    live_b = np.nonzero(t0 < 1e38)[0]
    bxs = (live_b % gx)[:, None] * 4 + np.arange(16)[None, :] % 4
    bys = (live_b // gx)[:, None] * 4 + np.arange(16)[None, :] // 4
    ok = (bxs < bx) & (bys < by)
    return (bys * bx + bxs)[ok], np.repeat(t0[live_b], 16).reshape(-1, 16)[ok]

def run(cam, tiles, tt, reps=4):
    tr.set_timing(True); ms = []
    for _ in range(reps):
        tr.trace_wave_tiles_device(cam, tiles, tt, hits_ptr=hits.data_ptr())
        torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
    tr.set_timing(False)
    return float(np.mean(ms[1:])) * 1e3

def measure(cam, tiles, tt):
    clocks.zero_()
    tr.set_debug_wave_clocks(clocks.data_ptr()); run(cam, tiles, tt, 2); tr.set_debug_wave_clocks(0)
    torch.cuda.synchronize()
    return clocks.cpu().numpy().astype(np.float64)

def classes(c):
    med = max(np.median(c[c > 0]), 1.0) if (c > 0).any() else 1.0
    return np.clip(np.floor(np.log(np.maximum(c, 1.0) / med) / np.log(1.5)) + 2, 0, 7)

def vec(cam, name): return np.array(cam[name][0], dtype=np.float64)

for step in steps:
    cam0, cam1 = orbit_cam(10.0), orbit_cam(10.0 + step)
    tiles0, tt0 = live_tiles(cam0)
    full0 = measure(cam0, tiles0, tt0)                       # clocks of frame k by wave tile (0 where nothing walked)
    tiles1, tt1 = live_tiles(cam1)
    own = measure(cam1, tiles1, tt1)[tiles1]
    # reprojection of every wave tile of frame k+1: its centre ray at the tile's start parameter, onto frame k's screen
    px = (tiles1 % bx) * 8 + 4.0; py = (tiles1 // bx) * 8 + 4.0
    th, asp = float(cam1["tan_half_fov"][0]), float(cam1["aspect"][0])
    u = (2 * px / Wd - 1) * th * asp; v = (1 - 2 * py / Ht) * th
    d = vec(cam1, "fwd")[None] + vec(cam1, "right")[None] * u[:, None] + vec(cam1, "up")[None] * v[:, None]
    d /= np.linalg.norm(d, axis=1)[:, None]
    P = vec(cam1, "pos")[None] + d * np.maximum(tt1, 1.0)[:, None]
    rel = P - vec(cam0, "pos")[None]
    z = rel @ vec(cam0, "fwd"); xx = rel @ vec(cam0, "right"); yy = rel @ vec(cam0, "up")
    with np.errstate(divide="ignore", invalid="ignore"):
        u0 = xx / z / (th * asp); v0 = yy / z / th
    qx = np.clip(((u0 + 1) * 0.5 * Wd) // 8, 0, bx - 1).astype(np.int64); qy = np.clip(((1 - v0) * 0.5 * Ht) // 8, 0, by - 1).astype(np.int64)
    behind = ~(z > 0)
    repro = np.where(behind, 0.0, full0[qy * bx + qx])
    from scipy.ndimage import maximum_filter
    grid0 = full0.reshape(by, bx)
    dil = {k: np.where(behind, 0.0, maximum_filter(grid0, size=k, mode="nearest")[qy, qx]) for k in (3, 5, 9, 17)}
    stale = full0[tiles1]
    nat = np.argsort(tiles1, kind="stable")
    print(f"camera step {step} degrees per frame: {len(tiles1)} live wave tiles; correlation of own clocks with stale {np.corrcoef(own, stale)[0, 1]:.2f}, with reprojected {np.corrcoef(own, repro)[0, 1]:.2f}", flush=True)
    def show(name, key):
        o = np.argsort(-key, kind="stable")
        print(f"   {name:34s} {run(cam1, tiles1[o], tt1[o]):7.1f} us", flush=True)
    print(f"   {'natural':34s} {run(cam1, tiles1[nat], tt1[nat]):7.1f} us", flush=True)
    t_nat, tt_nat, own_n, stale_n, repro_n = tiles1[nat], tt1[nat], own[nat], stale[nat], repro[nat]
    def show_n(name, key):                                   # stable sort of the natural order: ties stay row-major
        o = np.argsort(-key, kind="stable")
        print(f"   {name:34s} {run(cam1, t_nat[o], tt_nat[o]):7.1f} us", flush=True)
    show_n("own clocks", own_n); show_n("own clocks, 8 classes", classes(own_n))
    show_n("stale clocks", stale_n); show_n("stale clocks, 8 classes", classes(stale_n))
    show_n("reprojected clocks", repro_n); show_n("reprojected clocks, 8 classes", classes(repro_n))
    show_n("reprojected, 4 classes", np.floor(classes(repro_n) / 2))
    show_n("reprojected, 2 classes (heavy first)", (classes(repro_n) >= 5).astype(np.float64))
    # ONE whole-tile shift for the frame (median displacement of the live tiles) instead of a reprojection per tile: an order made
    # behind frame k over frame k's screen can then be used for frame k+1 by adding the shift to every entry (a bijection on the torus)
    sx = int(np.median(qx - tiles1 % bx)); sy = int(np.median(qy - tiles1 // bx))
    tx1 = (tiles1 % bx + sx) % bx; ty1 = (tiles1 // bx + sy) % by
    print(f"   whole-frame shift {sx}, {sy} wave tiles; residual displacement after it: median {np.median(np.hypot(qx - tiles1 % bx - sx, qy - tiles1 // bx - sy)):.1f}, 99th percentile {np.percentile(np.hypot(qx - tiles1 % bx - sx, qy - tiles1 // bx - sy), 99):.1f} tiles", flush=True)
    for k in (1, 3, 5, 9, 17):
        g = maximum_filter(grid0, size=k, mode="nearest") if k > 1 else grid0
        sh = g[ty1, tx1][nat]
        show_n(f"shifted frame, max over {k}x{k}", sh); show_n(f"shifted frame, max over {k}x{k}, 8 classes", classes(sh))
    def octave_classes(c, per_octave):                       # absolute classes: what a kernel can compute without a median
        return np.where(c > 0, np.floor(np.log2(np.maximum(c, 1.0)) * per_octave), 0.0)
    for k in (9,):
        g = maximum_filter(grid0, size=k, mode="nearest")
        sh = g[ty1, tx1][nat]
        for po in (1, 2, 4):
            show_n(f"shifted frame, max over {k}x{k}, {po} classes per octave", octave_classes(sh, po))
    heavy = np.argsort(-own_n)[:200]                          # where the 200 truly heaviest tiles stand in each predicted order
    def standing(key):
        rank = np.empty(len(key), dtype=np.int64); rank[np.argsort(-key, kind="stable")] = np.arange(len(key))
        r = rank[heavy]; return f"median rank {int(np.median(r))}, worst {int(r.max())} of {len(key)}"
    print(f"   the 200 heaviest tiles: reprojected {standing(repro_n)}", flush=True)
    for k, dk in dil.items():
        dn = dk[nat]
        print(f"   the 200 heaviest tiles: max over {k}x{k} {standing(dn)}", flush=True)
        show_n(f"reprojected, max over {k}x{k}", dn); show_n(f"reprojected, max over {k}x{k}, 8 classes", classes(dn))
tr.shutdown()
