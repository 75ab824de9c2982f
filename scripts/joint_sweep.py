"""Joint launch alone on an idle chip: visit budget of the searches x launch form, 4K over 1024^3, poses A, B, C."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
for pose in (0, 1, 2):
    cam = W.scene_camera(n, pose, Wd, Ht)
    tr.set_fused(0); tr.set_beam_budget(1 << 20)
    tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize()
    want = hits.clone()
    for form in (0, 2):
        for budget in (1 << 20, 256, 192, 128, 96, 64):
            tr.set_fused(form); tr.set_beam_budget(budget)
            for _ in range(40):
                tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr())
            torch.cuda.synchronize()
            same = bool(torch.equal(hits, want))
            tr.set_timing(True)
            ms = []
            for _ in range(30):
                tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
            tr.set_timing(False)
            print(f"pose {'ABC'[pose]} {'joint' if form else 'two launches'} budget {budget if budget < 100000 else 'unlimited':>9}: alone {np.mean(ms):.4f} ms (min {np.min(ms):.4f}) "
                  f"frac {843.5e6 / (np.mean(ms) * 1e-3) / 8e12:.3f}, records equal: {same}, gave up: {tr.frame_queue_stalls()}", flush=True)
            assert same
tr.shutdown()
