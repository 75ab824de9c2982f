"""Soak of the per-view order cache: a caller hopping at random among 7 fixed views (more than the 4 slots) and a zooming view, on one and on three streams,
two rectangles, with rests in between — every frame compared with the frame of a context with all ordering off; reports frames, mismatches, waves that gave up.
    python3 scripts/r04/views_soak.py [seconds=40]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world, SEED
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
n, Wd, Ht = 1024, 3840, 2160
cm, pw = make_scene_world(n)
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
ref = HipTracer(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0); ref.set_tile_ordering(0)
rng = np.random.default_rng(3)
centre = np.array([512.0, 200.0, 512.0])
views = [W.scene_camera(n, p, Wd, Ht, SEED) for p in (0, 1, 2)]
for k in range(4):
    a = rng.uniform(0, 2 * np.pi); r = rng.uniform(500, 1100); h = rng.uniform(200, 700)
    views.append(W.camera_look_at((centre[0] + r * np.cos(a), h, centre[2] + r * np.sin(a)), tuple(centre), 60.0, Wd, Ht))
rects = [None, (256, 128, 3072, 1728)]
want = {}
def expected(vi, ri, cam=None):
    key = (vi, ri)
    if key not in want or cam is not None:
        r = rects[ri]; px = (r[2] * r[3]) if r else Wd * Ht
        h = torch.zeros((px, 4), dtype=torch.int32, device="cuda")
        ref.draw_frame_device(cam if cam is not None else views[vi], h.data_ptr(), 0, rect=r); torch.cuda.synchronize()
        if cam is not None: return h
        want[key] = h
    return want[key]
streams = [torch.cuda.Stream() for _ in range(3)]
frames = mismatches = 0; uses = {0: 0, 1: 0, 2: 0}; by_mode = {}
def note(mode, ok, out, exp, info):
    global mismatches
    by_mode.setdefault(mode, [0, 0]); by_mode[mode][0] += 1
    if not ok:
        mismatches += 1; by_mode[mode][1] += 1
        if by_mode[mode][1] <= 3:
            d = (out != exp).any(dim=1).nonzero().flatten()
            print('MISMATCH', mode, info, 'records differing', len(d), 'first', d[:6].tolist(), 'order use', tr.last_order_use(), 'kind', tr.last_launch_kind(), flush=True)
t_end = time.time() + seconds
while time.time() < t_end:
    ri = int(rng.integers(2)); r = rects[ri]; px = (r[2] * r[3]) if r else Wd * Ht
    mode = rng.choice(["hop", "hop3", "zoom", "rest"])
    if mode == "zoom":
        fov = 60.0
        for k in range(12):
            fov -= rng.choice([0.2, 0.5, 2.0])
            cam = W.camera_look_at((-358.0, 870.0, -358.0), (512.0, 256.0, 512.0), max(fov, 20.0), Wd, Ht)
            out = torch.zeros((px, 4), dtype=torch.int32, device="cuda")
            tr.draw_frame_device(cam, out.data_ptr(), 0, rect=r); torch.cuda.synchronize()
            uses[tr.last_order_use()[0]] += 1
            e = expected(-1, ri, cam); frames += 1; note(mode, torch.equal(out, e), out, e, (ri, k, fov))
        continue
    seq = [int(rng.integers(len(views))) for _ in range(24)] if mode != "rest" else [int(rng.integers(len(views)))] * 12
    if mode == "hop3":
        outs = [torch.zeros((px, 4), dtype=torch.int32, device="cuda") for _ in streams]
        torch.cuda.synchronize()          # (torch zeroes them on ITS stream: the side streams do not wait for that)
        for base in range(0, len(seq), 3):
            for j in range(3):
                tr.draw_frame_device(views[seq[base + j]], outs[j].data_ptr(), 0, rect=r, stream=streams[j].cuda_stream)
            torch.cuda.synchronize()
            for j in range(3):
                e = expected(seq[base + j], ri); frames += 1; note(mode, torch.equal(outs[j], e), outs[j], e, (ri, base, j, seq[base:base + 3]))
    else:
        out = torch.zeros((px, 4), dtype=torch.int32, device="cuda")
        for vi in seq:
            tr.draw_frame_device(views[vi], out.data_ptr(), 0, rect=r); torch.cuda.synchronize()
            uses[tr.last_order_use()[0]] += 1
            e = expected(vi, ri); frames += 1; note(mode, torch.equal(out, e), out, e, (ri, vi))
print(by_mode)
print(f"{frames} frames in {seconds:.0f} s, mismatches {mismatches}, walk waves that gave up {tr.frame_queue_stalls()}, order use on the one-stream frames: row-major {uses[0]}, own view {uses[1]}, carried {uses[2]}")
tr.shutdown(); ref.shutdown()
