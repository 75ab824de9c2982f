import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, SEED = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, SEED); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(SEED))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
tr.set_timing(True)
def frame(cam, tag):
    tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize()
    print(f"{tag}: {tr.last_kernel_ms() * 1e3:8.1f} us, kind {tr.last_launch_kind()}, order use {tr.last_order_use()}, fallback tiles {tr.last_fallback_tiles()}, gave up so far {tr.frame_queue_stalls()}", flush=True)
centre = np.array([512.0, 60.0, 512.0])
for form in (3, 2):
    tr.set_fused(form)
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    for k in range(6): frame(cam, (form, "static", k))
    pos = np.array(cam["pos"][0], dtype=np.float64) - centre
    for k in range(1, 6):
        a = np.radians(0.05 * k)
        p = centre + np.array([pos[0] * np.cos(a) - pos[2] * np.sin(a), pos[1], pos[0] * np.sin(a) + pos[2] * np.cos(a)])
        frame(W.camera_look_at(tuple(p), tuple(centre), 60.0, Wd, Ht), (form, "creep", k))
    for k in range(4): frame(cam, (form, "back", k))
    for limit in (20000, 1000, 1):
        tr.set_joint_prefix_limit(limit)
        for k in range(3): frame(cam, (form, "limit", limit, k))
    tr.set_joint_prefix_limit(0)
    frame(W.scene_camera(1024, 1, Wd, Ht, SEED), (form, "jump"))
    frame(cam, (form, "return"))
tr.shutdown()
