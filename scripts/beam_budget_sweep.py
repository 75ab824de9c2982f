"""The pre-pass's visit budget (blok_hip_set_beam_budget): a search that runs out answers with the lower bound over its pending cells
(beam.h), so the budget bounds the pre-pass's longest wave.  Per budget: the launch pair alone (HIP events), three frames in flight,
and that the records are those of the unlimited search."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
tr.set_tile_ordering(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
streams = [torch.cuda.Stream() for _ in range(3)]
bufs = [(torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"), torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")) for _ in streams]
for pose in (0, 1, 2):
    cam = W.scene_camera(n, pose, Wd, Ht)
    tr.set_beam_budget(1 << 20)
    tr.draw_frame_device(cam, bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), stream=streams[0].cuda_stream); torch.cuda.synchronize()
    want = bufs[0][0].clone()
    for budget in (1 << 20, 1 << 20, 512, 384, 256, 192, 128, 96, 64, 32):
        tr.set_beam_budget(budget)
        for k in range(30):
            tr.draw_frame_device(cam, bufs[k % 3][0].data_ptr(), bufs[k % 3][1].data_ptr(), stream=streams[k % 3].cuda_stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(bufs[0][0], want))
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
        print(f"pose {'ABC'[pose]} budget {budget if budget < 100000 else 'unlimited':>9}: alone {np.mean(ms):.4f} ms (min {np.min(ms):.4f}), 3 in flight {dt:.4f} ms/frame, records equal: {same}", flush=True)
        assert same
tr.shutdown()
