"""Can a frame launch be captured into a hipGraph and replayed?  (INTEGRATION.md says the *_device forms neither allocate nor synchronise.)"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 256, 1920, 1080, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(n, 0, Wd, Ht, seed)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
want = torch.zeros_like(hits)
tr.draw_frame_device(cam, want.data_ptr(), 0); torch.cuda.synchronize()
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for label, setup in [(("everything on (default)", lambda: None), ("tile ordering off", lambda: tr.set_tile_ordering(0)), ("two-launch form, ordering off", lambda: (tr.set_fused(0), tr.set_tile_ordering(0))))[which]]:
    setup()
    s = torch.cuda.Stream()
    for _ in range(3):                                   # warm: buffers of this stream exist before the capture
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=s.cuda_stream)
        hits.zero_()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        print(f"{label}: captured and replayed; frame equals the direct launch: {torch.equal(hits, want)}", flush=True)
    except Exception as e:
        print(f"{label}: capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
        torch.cuda.synchronize()
tr.shutdown()
