"""GPU sweep of the one-launch frame's queue parameters (parts x ticket chunk): solitary launch time and back-to-back
throughput on one stream, against the two-launch form.  usage: fused_sweep.py [pose]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
cam = W.scene_camera(n, pose, Wd, Ht)
hits = torch.empty((Wd * Ht, 4), dtype=torch.int32, device="cuda")
rgba = torch.empty(Wd * Ht, dtype=torch.int32, device="cuda")
configs = [("two-launch", None, None)] + [(f"parts={p} chunk={c}", p, c) for p in (8, 32, 128, 256) for c in (1, 2, 4)] + [("two-launch", None, None)]
if len(sys.argv) > 2:
    configs = [(f"parts={p} chunk={c}", p, c) for p, c in (map(int, a.split(",")) for a in sys.argv[2:])]
ref = None
for name, parts, chunk in configs:
    if parts:
        os.environ["BLOK_FRAME_PARTS"] = str(parts); os.environ["BLOK_FRAME_CHUNK"] = str(chunk)
    tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
    tr.set_fused(parts is not None)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=st)
    torch.cuda.synchronize()
    tr.set_timing(True)
    ms = []
    for _ in range(20):
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=st)
        torch.cuda.synchronize()
        ms.append(tr.last_kernel_ms())
    tr.set_timing(False)
    t = time.perf_counter()
    for _ in range(100):
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 100 * 1e3
    h = hits.cpu().numpy()
    if ref is None:
        ref = h.copy()
    print(f"{name:22s} alone {np.mean(ms):.4f} ms (min {np.min(ms):.4f})  back-to-back {dt:.4f} ms/frame  stalls {tr.frame_queue_stalls()}  equal {bool((h == ref).all())}", flush=True)
    tr.shutdown()
