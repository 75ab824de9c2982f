"""Where the path kernel's time goes at 4K / 1024^3: bounces 1 vs 2, beam pre-pass and sun map on/off (8 spp)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(n, 0, Wd, Ht, seed)
color = torch.empty((Wd * Ht, 4), dtype=torch.float32, device="cuda")
tr.set_timing(True)
for bounces in (1, 2):
    for beam in (0, 32):
        for sun in (False, True):
            tr.set_beam(beam); tr.set_sun_map(sun)
            ms = []
            for f in range(3):
                tr.trace_paths_device(cam, color.data_ptr(), spp=8, max_bounces=bounces, frame_index=f)
                torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
            print(f"8 spp, bounces {bounces}, beam {beam:2d}, sun map {int(sun)}: {np.mean(ms[1:]):7.3f} ms", flush=True)
tr.shutdown()
