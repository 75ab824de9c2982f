"""The bounce rounds' tail pool (blok_hip_set_ray_batching 3) against the plain rounds (2): time of a 4K frame and the colour plane's largest deviation in units of
tests/test_paths.py's tolerance (1e-4 + 1e-3 |ref|).  usage: tail_pool_check.py [spp=16] [poses=0,1]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
poses = [int(p) for p in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1]
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
tr.set_timing(True)
for pose in poses:
    cam = W.scene_camera(n, pose, Wd, Ht, seed)
    planes = {}
    for mode in (2, 3, 2, 3):
        tr.set_ray_batching(mode)
        color = torch.zeros((Wd * Ht, 4), dtype=torch.float32, device="cuda")
        ms = []
        for f in range(2):
            tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=2, frame_index=1)
            torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
        planes[mode] = color
        print(f"pose {'ABC'[pose]} {spp} spp, mode {mode}: {ms[-1]:8.3f} ms", flush=True)
    a, b = planes[2][:, :3], planes[3][:, :3]
    dev = ((a - b).abs() / (1e-4 + 1e-3 * a.abs())).max().item()
    print(f"  largest deviation of the colour plane: {dev:.4f} of the tolerance; NaNs {int(torch.isnan(b).sum())}", flush=True)
tr.shutdown()
