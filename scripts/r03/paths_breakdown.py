"""Where a path-traced 4K frame's time goes: 64 spp with 0, 1 and 2 bounces (the differences are the bounce segments with their shadow rays),
with the sun map and the beam pre-pass on and off.  HIP events around the kernel; one MI355X."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
color = torch.empty((Wd * Ht, 4), dtype=torch.float32, device="cuda")
tr.set_timing(True)
for pose in (0, 1):
    cam = W.scene_camera(n, pose, Wd, Ht, seed)
    for bounces in (1, 2, 3, 4):
        ms = []
        for f in range(3):
            tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=bounces, frame_index=1)
            torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
        print(f"pose {'ABC'[pose]} {spp} spp, {bounces} bounce(s): {np.mean(ms[1:]):8.3f} ms", flush=True)
tr.shutdown()
