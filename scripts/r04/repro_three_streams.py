import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world, SEED
n, Wd, Ht = 1024, 3840, 2160
cm, pw = make_scene_world(n)
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
ref = HipTracer(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0); ref.set_tile_ordering(0)
views = [W.scene_camera(n, p, Wd, Ht, SEED) for p in (0, 2)]
want = []
for v in views:
    h = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); ref.draw_frame_device(v, h.data_ptr(), 0); torch.cuda.synchronize(); want.append(h)
out = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda")
for k in range(24):
    tr.draw_frame_device(views[k % 2], out.data_ptr(), 0); torch.cuda.synchronize()
print("warm: order use", tr.last_order_use())
streams = [torch.cuda.Stream() for _ in range(3)]
for trial in range(6):
    seq = [(0, 1, 1), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 0), (1, 1, 0)][trial]
    outs = [torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda") for _ in streams]
    if len(sys.argv) < 2: torch.cuda.synchronize()      # without it torch's zero fill (on ITS stream) races the frames on the side streams: 'never written' records
    kinds = []
    for j in range(3):
        tr.draw_frame_device(views[seq[j]], outs[j].data_ptr(), 0, stream=streams[j].cuda_stream); kinds.append((tr.last_launch_kind(), tr.last_order_use()[0]))
    torch.cuda.synchronize()
    for j in range(3):
        d = (outs[j] != want[seq[j]]).any(dim=1)
        nd = int(d.sum())
        if nd:
            idx = d.nonzero().flatten()
            zeros = int((outs[j][idx] == 0).all(dim=1).sum())
            ys, xs = (idx // Wd), (idx % Wd)
            print(f"trial {trial} seq {seq} stream {j} kind/use {kinds[j]}: {nd} records differ, {zeros} of them all-zero (never written); rows {int(ys.min())}-{int(ys.max())} cols {int(xs.min())}-{int(xs.max())}; distinct 8x8 tiles {len(set(((ys // 8) * 480 + xs // 8).tolist()))}, distinct 32x32 beam tiles {len(set(((ys // 32) * 120 + xs // 32).tolist()))}")
        else:
            print(f"trial {trial} seq {seq} stream {j} kind/use {kinds[j]}: ok")
tr.shutdown(); ref.shutdown()
