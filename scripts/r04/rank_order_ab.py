"""A rank's launches at N GPUs (bench.py --gpus N: tile-frame launches, F = N frames of the rank's 32x32 tiles per launch pair), played by ONE
GPU: round 3's launches (blok_hip_set_rank_tile_ordering 0: natural order, a walk workgroup per wave tile) against round 4's (the rank's own
longest-first order, walk workgroups for its live prefix) — the launch alone and with `depth` launches in flight; same frames required.
    python3 scripts/r04/rank_order_ab.py [pose=0]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cam = W.scene_camera(n, pose, Wd, Ht, seed)
for N in (8, 4, 2):
    F, rank, depth = N, 1, (3 if N <= 2 else 4)
    per = tr.tiles_for_rank(32, 0, N)
    streams = [torch.cuda.Stream() for _ in range(depth)]
    bufs = [(torch.zeros((F, per * 1024, 4), dtype=torch.int32, device="cuda"), torch.zeros((F, per * 1024), dtype=torch.int32, device="cuda")) for _ in streams]
    cams = np.concatenate([cam] * F)
    def go(slot):
        tr.draw_tile_frames_device(cams, 32, rank, N, per, hits_ptr=bufs[slot][0].data_ptr(), rgba_ptr=bufs[slot][1].data_ptr(), stream=streams[slot].cuda_stream)
    ref = None
    for ordered in (0, 1, 0, 1):
        tr.set_rank_tile_ordering(bool(ordered))
        for _ in range(24):
            go(0); torch.cuda.synchronize()
        if ref is None:
            ref = (bufs[0][0].clone(), bufs[0][1].clone())
        else:
            assert torch.equal(bufs[0][0], ref[0]) and torch.equal(bufs[0][1], ref[1]), ordered
        tr.set_timing(True); ms = []
        for _ in range(12):
            go(0); torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
        use_alone = tr.last_order_use()[0]
        tr.set_timing(False)
        reps = 48
        for k in range(depth):
            go(k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(reps):
            go(k % depth)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps * 1e3
        print(f"N = {N}, rank {rank}, {F} frames per launch, rank-tile ordering {ordered}: alone {np.mean(ms) * 1e3:7.1f} us per launch ({np.mean(ms) / F * 1e3:6.1f} per frame-share); "
              f"{depth} in flight {dt * 1e3:7.1f} us per launch ({dt / F * 1e3:6.1f} per frame-share); order in use {use_alone} / {tr.last_order_use()[0]}, launch kind {tr.last_launch_kind()}, gave up {tr.frame_queue_stalls()}", flush=True)
tr.shutdown()
