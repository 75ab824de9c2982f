"""Path kernel at BASELINE configs[4] (4K, 64 spp, 2 bounces, 1024^3): where the walks start (blok_hip_set_path_start: entered from the pixel's
anchor or from the root; wave-tile beam on or off), HIP events around the kernel; every combination must give the same frame bit for bit."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
combos = [tuple(int(c) for c in m) for m in sys.argv[2].split(',')] if len(sys.argv) > 2 else [(0, 0), (1, 0), (0, 1), (1, 1)]
poses = [int(p) for p in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 1, 2]
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
color = torch.empty((Wd * Ht, 4), dtype=torch.float32, device="cuda")
tr.set_timing(True)
for pose in poses:
    cam = W.scene_camera(n, pose, Wd, Ht, seed)
    ref = None
    for resume, fine in combos:
        tr.set_path_start(resume, fine)
        ms = []
        for f in range(3):
            tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=2, frame_index=1)
            torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
        digest = (color.view(torch.int32).to(torch.int64) * torch.arange(1, color.numel() + 1, device="cuda").view(-1, 4) % 1000003).sum().item()
        if ref is None: ref = digest
        print(f"pose {'ABC'[pose]} {spp} spp, resume {resume} wave-tile beam {fine}: {np.mean(ms[1:]):8.3f} ms   same frame as the first: {digest == ref}", flush=True)
tr.shutdown()
