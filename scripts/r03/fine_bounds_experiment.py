"""What would start parameters FINER than the 32x32-pixel beam tile be worth to the walk, and where?  4K over 1024^3: the walk kernel alone
(blok_hip_trace_wave_tiles_device, longest first by the tiles' own clocks) with the start parameter of every wave tile taken from the
32x32 pre-pass, from a 16x16 or an 8x8 pre-pass, and with the finer bound given only to the heaviest X % of the wave tiles (what an order
carried over from the previous frame could single out).  Records are checked to be the same.  The finer searches' own cost is printed
beside it (pre-pass alone at each granularity, by HIP events)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(n, pose, Wd, Ht, seed)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda")
bx, by = Wd // 8, (Ht + 7) // 8
clocks = torch.zeros(bx * by, dtype=torch.int32, device="cuda")
t0s = {}
for B in (32, 16, 8):
    tr.set_beam(B)
    t0, _ = tr.beam_prepass(cam)
    gx = (Wd + B - 1) // B
    ty, tx = np.divmod(np.arange(bx * by), bx)
    t0s[B] = t0[(ty * 8 // B) * gx + (tx * 8 // B)]                       # per wave tile
    import time
    ms = []
    for _ in range(5):
        t = time.perf_counter(); tr.beam_prepass(cam); ms.append((time.perf_counter() - t) * 1e3)
    print(f"pre-pass alone, {B}x{B}-pixel beam tiles: {min(ms[1:]) * 1e3:7.1f} us by the host's clock (launch, kernel, read-back); live wave tiles {(t0s[B] < 1e38).sum()}", flush=True)
tr.set_beam(32)
live = np.nonzero(t0s[32] < 1e38)[0]

def run(tiles, tt, reps=5):
    tr.set_timing(True); ms = []
    for _ in range(reps):
        tr.trace_wave_tiles_device(cam, tiles, tt, hits_ptr=hits.data_ptr())
        torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
    tr.set_timing(False)
    return float(np.mean(ms[1:])) * 1e3

clocks.zero_(); tr.set_debug_wave_clocks(clocks.data_ptr()); run(live, t0s[32][live], 2); tr.set_debug_wave_clocks(0); torch.cuda.synchronize()
own = clocks.cpu().numpy().astype(np.float64)[live]
order = np.argsort(-own, kind="stable")
tiles = live[order]
base = run(tiles, t0s[32][tiles]); ref = hits.clone()
print(f"pose {'ABC'[pose]}: walk alone, longest first, 32x32 bounds: {base:7.1f} us  ({len(tiles)} wave tiles, clocks sum {own.sum() / 1e6:.0f} M)", flush=True)
for B in (16, 8):
    finite = np.where(t0s[B][tiles] < 1e38, t0s[B][tiles], 0.0)          # a wave tile the finer search finds empty: walked from the coarse bound here (a product would skip it)
    fine = np.maximum(finite, t0s[32][tiles]).astype(np.float32)
    dead = (t0s[B][tiles] >= 1e38).sum()
    for frac in (0.02, 0.05, 0.1, 0.25, 1.0):
        k = int(len(tiles) * frac)
        tt = t0s[32][tiles].copy(); tt[:k] = fine[:k]
        us = run(tiles, tt)
        assert torch.equal(hits, ref), (B, frac)
        share = own[order][:k].sum() / own.sum()
        print(f"   {B}x{B} bounds for the heaviest {frac * 100:5.1f} % of the wave tiles ({share * 100:4.1f} % of the clocks): {us:7.1f} us", flush=True)
    print(f"   ({dead} of the live wave tiles are empty by the {B}x{B} search: all-miss waves)", flush=True)
tr.shutdown()
