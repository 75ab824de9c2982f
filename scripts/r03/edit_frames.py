"""A camera at rest over a world that is edited every frame (sphere brush at a visible point + rebuild of the resident volume): the frame's
launch alone, by HIP events.  Scheduling state measured on the previous contents must not outlive them."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
ids = W.scene_dense(n, seed)
tr = HipTracer(Wd, Ht).init()
tr.volume_create((0, 0, 0), (n, n, n), 128, 1.0)
tr.volume_upload((ids != 0).astype(np.float32), ids)
mats = W.scene_materials(seed)
tr.volume_rebuild(mats)
cam = W.scene_camera(n, 0, Wd, Ht, seed)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
base = tr.draw_frame(cam)
ys, xs = np.nonzero(base["hit"])
rng = np.random.default_rng(3)
tr.set_timing(True)
for phase, edit in (("no edits", False), ("a brush of radius 10 at a visible voxel + rebuild before every frame", True), ("no edits again", False)):
    ms = []
    for k in range(48):
        if edit:
            j = rng.integers(len(ys))
            c = tuple(min(max(float(v) + 0.5, 12.0), n - 13.0) for v in base["voxel"][ys[j], xs[j]])
            tr.volume_apply_brush(c, 10.0, 0.0 if k % 2 else 1.0, 1 if k % 2 else 0)
            tr.volume_rebuild(mats)
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr())
        torch.cuda.synchronize()
        if k >= 8:
            ms.append(tr.last_kernel_ms())
    print(f"{phase}: launch {np.mean(ms) * 1e3:6.1f} us (median {np.median(ms) * 1e3:6.1f}, max {np.max(ms) * 1e3:6.1f})", flush=True)
tr.shutdown()
