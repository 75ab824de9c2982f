"""From frames one at a time to frames in flight and back, camera at rest: per-chunk frame times and what the launches were (diagnostic)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
if len(sys.argv) > 1: tr.set_moving_order(bool(int(sys.argv[1])))
cam = W.scene_camera(n, 0, Wd, Ht, seed)
streams = [torch.cuda.Stream() for _ in range(3)]
bufs = [(torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"), torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")) for _ in streams]
def go(slot):
    tr.draw_frame_device(cam, bufs[slot][0].data_ptr(), bufs[slot][1].data_ptr(), stream=streams[slot].cuda_stream)
for phase in range(3):
    t = time.perf_counter()
    for k in range(24):
        go(0); torch.cuda.synchronize()
    print(f"phase {phase}: 24 frames one at a time: {(time.perf_counter() - t) / 24 * 1e3:.4f} ms/frame (wall); last launch kind {tr.last_launch_kind()}, order use {tr.last_order_use()}", flush=True)
    for chunk in range(6):
        kinds = []
        t = time.perf_counter()
        for k in range(24):
            go(k % 3); kinds.append((tr.last_launch_kind(), tr.last_order_use()[0]))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 24 * 1e3
        print(f"   in flight, chunk {chunk}: {dt:.4f} ms/frame; (kind, order use) counts {sorted((k, kinds.count(k)) for k in set(kinds))}", flush=True)
tr.shutdown()
