"""Why does a frame-equivalent of a rank's tile-frame launches cost what it costs?  One GPU plays rank `r` of N with F frames per launch
(launch form 0): the launch alone and `depth` in flight, for several (N, F, depth); also all F cameras the same against F different ones."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
tr.set_fused(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cam = W.scene_camera(n, 0, Wd, Ht, seed)
for N, F, rank, depth in [(1, 1, 0, 3), (1, 1, 0, 4), (8, 8, 1, 3), (8, 8, 1, 4), (8, 8, 0, 4), (8, 1, 1, 4), (8, 2, 1, 4), (8, 4, 1, 4), (2, 2, 1, 3), (4, 4, 1, 4), (1, 8, 0, 3)]:
    per = tr.tiles_for_rank(32, 0, N)
    streams = [torch.cuda.Stream() for _ in range(depth)]
    bufs = [(torch.zeros((F, per * 1024, 4), dtype=torch.int32, device="cuda"), torch.zeros((F, per * 1024), dtype=torch.int32, device="cuda")) for _ in streams]
    cams = np.concatenate([cam] * F)
    def go(slot):
        tr.draw_tile_frames_device(cams, 32, rank, N, per, hits_ptr=bufs[slot][0].data_ptr(), rgba_ptr=bufs[slot][1].data_ptr(), stream=streams[slot].cuda_stream)
    for k in range(2 * depth):
        go(k % depth)
    torch.cuda.synchronize()
    tr.set_timing(True); ms = []
    for _ in range(10):
        go(0); torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
    tr.set_timing(False)
    reps = 60
    t = time.perf_counter()
    for k in range(reps):
        go(k % depth)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps * 1e3
    fe = F / N                                     # frame-equivalents per launch
    print(f"rank {rank} of {N}, {F} frames per launch ({fe:.2f} frame-equivalents), {depth} in flight: alone {np.mean(ms) * 1e3:7.1f} us = {np.mean(ms) / fe * 1e3:7.1f} per frame-equivalent; "
          f"in flight {dt * 1e3:7.1f} us = {dt / fe * 1e3:7.1f} per frame-equivalent", flush=True)
    del bufs
tr.shutdown()
