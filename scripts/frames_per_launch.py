"""How long one beam + trace launch pair takes when it carries F frames (Tiles entry, one rank owning every tile), alone and with
three launches in flight, next to the rectangle entry (one frame per launch pair)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cam = W.scene_camera(n, pose, Wd, Ht)
per = tr.tiles_for_rank(32, 0, 1)
streams = [torch.cuda.Stream() for _ in range(3)]
bufs = [(torch.zeros((8, per * 1024, 4), dtype=torch.int32, device="cuda"), torch.zeros((8, per * 1024), dtype=torch.int32, device="cuda")) for _ in streams]
want = torch.from_numpy(tr.shade_rgba8(cam).reshape(-1).view(np.int32)).cuda()
def rect(slot):
    tr.draw_frame_device(cam, bufs[slot][0].data_ptr(), bufs[slot][1].data_ptr(), stream=streams[slot].cuda_stream)
def tiles(F):
    cams = np.concatenate([cam] * F)
    def go(slot):
        tr.draw_tile_frames_device(cams, 32, 0, 1, per, hits_ptr=bufs[slot][0].data_ptr(), rgba_ptr=bufs[slot][1].data_ptr(), stream=streams[slot].cuda_stream)
    return go
for name, fn, F in [("rectangle entry, 1 frame", rect, 1)] + [(f"tile entry, {F} frame(s) per launch pair", tiles(F), F) for F in (1, 2, 3, 4, 6, 8)]:
    for _ in range(12):
        fn(0)
    torch.cuda.synchronize()
    tr.set_timing(True)
    ms = []
    for _ in range(12):
        fn(0); torch.cuda.synchronize(); ms.append(tr.last_kernel_ms())
    tr.set_timing(False)
    reps = max(4, 96 // F)
    t = time.perf_counter()
    for k in range(reps):
        fn(k % 3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / (reps * F) * 1e3
    alone = float(np.mean(ms)) / F
    print(f"{name:42s} alone {alone:.4f} ms/frame (frac of 8 TB/s at 843.5 MB/frame: {843.5e6 / (alone * 1e-3) / 8e12:.3f}); 3 in flight {dt:.4f} ms/frame", flush=True)
tr.shutdown()
