"""The heaviest tiles in segments (blok_hip_set_heavy_split): the launch alone at rest (HIP events, one frame at a time), 4K over 1024^3, poses A / B / C,
for several (segments, threshold) settings; every frame compared with the frame of a context with all ordering off.
    python3 scripts/r04/split_probe.py [poses=0,1,2] [frames=40]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world, SEED
n, Wd, Ht = 1024, 3840, 2160
poses = [int(p) for p in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 2]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cm, pw = make_scene_world(n)
ref = HipTracer(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0); ref.set_tile_ordering(False)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
want_h = torch.zeros_like(hits); want_c = torch.zeros_like(rgba)
settings = [(0, 0), (2, 100000), (2, 150000), (2, 180000), (2, 220000), (0, 0)]
for pose in poses:
    cam = W.scene_camera(n, pose, Wd, Ht, SEED)
    ref.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr()); torch.cuda.synchronize()
    for segs, clocks in settings:
        tr = HipTracer(Wd, Ht).init(); tr.add_world(pw); tr.set_timing(True)
        tr.set_heavy_split(segs, clocks)
        ms = []; same = True; ks = []
        for k in range(frames):
            hits.fill_(5); rgba.fill_(5)
            tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize()
            ms.append(tr.last_kernel_ms()); ks.append(tr.last_split_tiles())
            same = same and torch.equal(hits, want_h) and torch.equal(rgba, want_c)
        m = np.array(ms[24:])
        print(f"pose {'ABC'[pose]} segments {segs} threshold {clocks:6d} clocks: alone mean {m.mean():.4f} ms median {np.median(m):.4f} min {m.min():.4f}; frames identical: {same}; gave up {tr.frame_queue_stalls()}; tiles split per frame {ks[8:40:4]}", flush=True)
        tr.shutdown()
ref.shutdown()
