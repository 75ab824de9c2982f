"""Who writes the miss pixels of the tiles the pre-pass found empty: the pre-pass wave of the tile or the walk launch's (otherwise idle)
waves of the tile (blok_hip_set_miss_writer).  Launch pair alone (HIP events), three frames in flight, frames identical."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
streams = [torch.cuda.Stream() for _ in range(3)]
bufs = [(torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"), torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")) for _ in streams]
for ordering in (8, 0):
    tr.set_tile_ordering(ordering)
    for pose in (0, 1, 2):
        cam = W.scene_camera(n, pose, Wd, Ht)
        want = None
        for in_walk in (False, True, False, True):
            tr.set_miss_writer(in_walk)
            for b in bufs:
                b[0].fill_(7); b[1].fill_(7)
            for k in range(30):
                tr.draw_frame_device(cam, bufs[k % 3][0].data_ptr(), bufs[k % 3][1].data_ptr(), stream=streams[k % 3].cuda_stream)
            torch.cuda.synchronize()
            if want is None:
                want = (bufs[0][0].clone(), bufs[0][1].clone())
            same = all(bool(torch.equal(b[0], want[0])) and bool(torch.equal(b[1], want[1])) for b in bufs)
            tr.set_timing(True)
            ms = []
            for _ in range(20):
                tr.draw_frame_device(cam, bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), stream=streams[0].cuda_stream); torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
            tr.set_timing(False)
            t = time.perf_counter()
            for k in range(150):
                tr.draw_frame_device(cam, bufs[k % 3][0].data_ptr(), bufs[k % 3][1].data_ptr(), stream=streams[k % 3].cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 150 * 1e3
            print(f"ordering {ordering} pose {'ABC'[pose]} misses written by {'the walk' if in_walk else 'the pre-pass'}: alone {np.mean(ms):.4f} ms, 3 in flight {dt:.4f} ms/frame, frames identical: {same}", flush=True)
            assert same
tr.shutdown()
