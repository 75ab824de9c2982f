"""What order of the live wave tiles does the walk want?  4K over 1024^3: the pre-pass gives the live beam tiles and their start
parameters; the walk is then launched over explicit lists of the live wave tiles (blok_hip_trace_wave_tiles_device) in different
orders, HIP events around the walk kernel alone:
  natural        row-major wave tiles (what the static forms dispatch, minus the dead tiles)
  beam-major     beam tile by beam tile (row-major beam tiles, their 16 wave tiles together) = what a list in tile order would be
  measured       descending measured clocks of the same frame (the ideal longest-first order, unknowable before the walk)
  measured/beam  beam tiles by descending sum of their wave tiles' clocks
  visits         beam tiles by descending node visits of their search
  span           beam tiles by descending (exit of the tile's central ray from the world's box - start parameter)
  random, shortest (ascending measured clocks)
"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
import ctypes as C
from tests import harness_ffi as H, oracle_ffi as O
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
HL = H.lib()
HL.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
def probe_iterations(cam, px, py, tstart):
    """Loop trips of the kernel's walk (CPU harness) for the primary ray of pixel (px, py) started at tstart."""
    out = np.zeros(1, dtype=O.HIT); it = np.zeros(1, dtype=np.uint32); ts = np.array([tstart], dtype=np.float32)
    HL.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, int(px), int(py), 1, 1, C.c_void_p(ts.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    return int(it[0])
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
ref = torch.zeros_like(hits)
bx, by = Wd // 8, Ht // 8
clocks = torch.zeros(bx * by, dtype=torch.int32, device="cuda")
rng = np.random.default_rng(1)
for pose in (0, 1, 2):
    cam = W.scene_camera(n, pose, Wd, Ht, seed)
    tr.set_fused(0)
    tr.draw_frame_device(cam, ref.data_ptr(), 0)
    t0, visits = tr.beam_prepass(cam, want_visits=True)
    gx = (Wd + 31) // 32
    live_b = np.nonzero(t0 < 1e38)[0]
    # wave tiles of the live beam tiles
    def tiles_of(beams):
        bxs = (beams % gx)[:, None] * 4 + np.arange(16)[None, :] % 4
        bys = (beams // gx)[:, None] * 4 + np.arange(16)[None, :] // 4
        ok = (bxs < bx) & (bys < by)
        return (bys * bx + bxs)[ok], np.repeat(t0[beams], 16).reshape(-1, 16)[ok]
    tiles, tt = tiles_of(live_b)
    beam_of = {int(t): None for t in tiles}
    def run(order_tiles, order_t0, reps=5):
        tr.set_timing(True); ms = []
        for _ in range(reps):
            tr.trace_wave_tiles_device(cam, order_tiles, order_t0, hits_ptr=hits.data_ptr(), rgba_ptr=rgba.data_ptr())
            torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
        tr.set_timing(False)
        return float(np.mean(ms[1:]))
    # measured clocks of every live wave tile (and the frame check: listed tiles equal the frame; the others stay as they were)
    hits.copy_(ref); hits[:, 0] = 0
    tr.set_debug_wave_clocks(clocks.data_ptr())
    base = run(tiles, tt, 2)
    tr.set_debug_wave_clocks(0)
    torch.cuda.synchronize()
    ck = clocks.cpu().numpy().astype(np.int64)[tiles]
    hv = hits.view(Ht // 8, 8, Wd // 8, 8, 4).permute(0, 2, 1, 3, 4).reshape(-1, 64, 4)[torch.from_numpy(tiles.astype(np.int64)).cuda()]
    rv = ref.view(Ht // 8, 8, Wd // 8, 8, 4).permute(0, 2, 1, 3, 4).reshape(-1, 64, 4)[torch.from_numpy(tiles.astype(np.int64)).cuda()]
    assert torch.equal(hv, rv), "listed tiles differ from the frame"
    print(f"pose {'ABC'[pose]}: {len(live_b)} live beam tiles of {len(t0)}, {len(tiles)} wave tiles; clocks per wave: median {np.median(ck):.0f}, mean {ck.mean():.0f}, max {ck.max()}, sum {ck.sum() / 1e6:.1f} M", flush=True)
    res = {}
    nat = np.argsort(tiles, kind="stable")
    res["natural (row-major wave tiles)"] = run(tiles[nat], tt[nat])
    res["beam-major (row-major beam tiles)"] = run(tiles, tt)
    o = np.argsort(-ck, kind="stable"); res["measured clocks, descending"] = run(tiles[o], tt[o])
    o = np.argsort(ck, kind="stable"); res["measured clocks, ascending"] = run(tiles[o], tt[o])
    o = rng.permutation(len(tiles)); res["random"] = run(tiles[o], tt[o])
    # beam-tile level keys
    per_beam_tiles = [tiles_of(np.array([b])) for b in live_b]
    idx_of = {int(t): i for i, t in enumerate(tiles)}
    sums = np.array([sum(ck[idx_of[int(t)]] for t in pt[0]) for pt in per_beam_tiles])
    def by_beam(key):
        order = np.argsort(-key, kind="stable")
        tl = np.concatenate([per_beam_tiles[i][0] for i in order]); t0s = np.concatenate([per_beam_tiles[i][1] for i in order])
        return run(tl, t0s)
    res["beam tiles by measured clocks"] = by_beam(sums.astype(np.float64))
    res["beam tiles by search visits"] = by_beam(visits[live_b].astype(np.float64))
    # span: exit of the central ray of the beam tile from the world box minus t0
    c = cam[0] if cam.dtype.names else None
    pos = np.array(cam["pos"][0], dtype=np.float64); fwd = np.array(cam["fwd"][0], dtype=np.float64); right = np.array(cam["right"][0], dtype=np.float64); up = np.array(cam["up"][0], dtype=np.float64)
    th = float(cam["tan_half_fov"][0]); asp = float(cam["aspect"][0])
    cxp = (live_b % gx) * 32 + 16.0; cyp = (live_b // gx) * 32 + 16.0
    u = (2 * cxp / Wd - 1) * th * asp; v = (1 - 2 * cyp / Ht) * th
    d = fwd[None, :] + right[None, :] * u[:, None] + up[None, :] * v[:, None]; d /= np.linalg.norm(d, axis=1)[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        tfar = np.where(d > 0, (n - pos[None, :]) / d, np.where(d < 0, (0 - pos[None, :]) / d, np.inf)).min(axis=1)
    res["beam tiles by span (world exit - t0)"] = by_beam(tfar - t0[live_b])
    res["beam tiles by visits x span"] = by_beam(visits[live_b] * np.maximum(tfar - t0[live_b], 1.0))
    # probes: the walk's own trip count for the ray through the beam tile's centre (and the worst of five rays) from t0
    cxi = np.minimum((live_b % gx) * 32 + 16, Wd - 1); cyi = np.minimum((live_b // gx) * 32 + 16, Ht - 1)
    centre = np.array([probe_iterations(cam, x, y, t) for x, y, t in zip(cxi, cyi, t0[live_b])], dtype=np.float64)
    res["beam tiles by trips of the centre ray"] = by_beam(centre)
    five = centre.copy()
    for dx, dy in ((-12, -12), (12, -12), (-12, 12), (12, 12)):
        five = np.maximum(five, [probe_iterations(cam, min(max(x + dx, 0), Wd - 1), min(max(y + dy, 0), Ht - 1), t) for x, y, t in zip(cxi, cyi, t0[live_b])])
    res["beam tiles by worst trips of five rays"] = by_beam(five)
    print(f"   correlation of beam-tile clocks with: visits {np.corrcoef(sums, visits[live_b])[0, 1]:.2f}, span {np.corrcoef(sums, tfar - t0[live_b])[0, 1]:.2f}, centre-ray trips {np.corrcoef(sums, centre)[0, 1]:.2f}, five-ray trips {np.corrcoef(sums, five)[0, 1]:.2f}")
    for k, v_ in res.items():
        print(f"   {k:40s} {v_ * 1e3:7.1f} us", flush=True)
tr.shutdown()
