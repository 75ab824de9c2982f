"""One rank's share of an N-rank frame (tiles i % N == 0) on one GPU, with d frames in flight on alternating streams:
how deep must the pipeline be to hide the kernel tail when the per-frame work is small?"""
import sys, time
sys.path.insert(0, '.')
import torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world
cm, pw = make_scene_world(1024)
Wd, Ht, tile = 3840, 2160, 32
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(1024, 0, Wd, Ht)
for n_ranks in (2, 4, 8):
    per = tr.tiles_for_rank(tile, 0, n_ranks)
    for depth in (1, 2, 3, 4, 6):
        streams = [torch.cuda.Stream() for _ in range(depth)]
        hits = [torch.empty((per * tile * tile, 4), dtype=torch.int32, device="cuda") for _ in range(depth)]
        rgba = [torch.empty(per * tile * tile, dtype=torch.int32, device="cuda") for _ in range(depth)]
        def run(k):
            for f in range(k):
                s = f % depth
                tr.draw_tiles_device(cam, tile, 0, n_ranks, hits[s].data_ptr(), rgba[s].data_ptr(), stream=streams[s].cuda_stream)
            torch.cuda.synchronize()
        run(20)
        t0 = time.perf_counter(); run(400); dt = (time.perf_counter() - t0) / 400
        print(f"N={n_ranks} depth={depth}: {dt * 1e6:7.1f} us per frame-share  -> {Wd * Ht / n_ranks / dt / 1e9:6.2f} Grays/s per rank, x{n_ranks} = {Wd * Ht / dt / 1e9:6.1f}")
