"""Joint launch (blok_hip_set_fused(ctx, 2)): the searches and the walk waves in one grid.  A small frame first, then 4K: records
against the two-launch form, waves that gave up waiting, the launch alone and three frames in flight."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n = 1024
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
for (Wd, Ht) in ((256, 256), (3840, 2160)):
    tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [(torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"), torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")) for _ in streams]
    for pose in (0, 1, 2):
        cam = W.scene_camera(n, pose, Wd, Ht)
        want = None
        for form in (0, 2, 0, 2):
            tr.set_fused(form)
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
            print(f"{Wd}x{Ht} pose {'ABC'[pose]} {'joint launch' if form == 2 else 'two launches'}: alone {np.mean(ms):.4f} ms (min {np.min(ms):.4f}), 3 in flight {dt:.4f} ms/frame, "
                  f"frames identical: {same}, waves that gave up: {tr.frame_queue_stalls()}", flush=True)
            assert same
    tr.shutdown()
