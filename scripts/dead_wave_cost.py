"""What do the trace launch's exit-at-once waves cost?  A camera that sees only sky: every beam tile is 'none', the pre-pass writes the
frame, and the trace kernel is 129 600 waves that read one float and exit.  Run under rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.camera_look_at((512.0, 900.0, 512.0), (512.0, 2000.0, 530.0), 60.0, Wd, Ht)      # above the world, looking up
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
tr.set_timing(True)
ms = []
for _ in range(30):
    tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
print(f"all-sky frame: launch pair alone {np.mean(ms[5:]):.4f} ms; hits {(hits[:, 3] >> 24).sum().item()}")
tr.shutdown()
