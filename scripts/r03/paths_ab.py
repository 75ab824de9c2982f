"""Path kernel at BASELINE configs[4] (4K, 64 spp, 2 bounces, 1024^3): scheduling modes of blok_hip_set_ray_batching, HIP events around the kernel."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
modes = [int(m) for m in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 0]
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
color = torch.empty((Wd * Ht, 4), dtype=torch.float32, device="cuda")
tr.set_timing(True)
for pose in (0, 1, 2):
    cam = W.scene_camera(n, pose, Wd, Ht, seed)
    ref = None
    for mode in modes:
        tr.set_ray_batching(mode)
        ms = []
        for f in range(3):
            tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=2, frame_index=1)
            torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
        digest = color.view(torch.int32).sum().item()
        if ref is None: ref = digest
        print(f"pose {'ABC'[pose]} {spp} spp, batching mode {mode}: {np.mean(ms[1:]):8.3f} ms   same frame as first mode: {digest == ref}" + (f"   digest {digest & 0xFFFFFFFF:08x}" if len(sys.argv) > 3 else ""), flush=True)
tr.shutdown()
