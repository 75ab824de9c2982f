"""The distributed frame pipeline over RCCL on a ONE-GPU box: a process group of one rank (backend "nccl"), the frame cut into
tiles, the dense gather + un-permute and the sparse exchange (all-reduce of the record count, prefix gather, scatter), 30 frames
each with 3 frames in flight, checked against the single-launch frame.  RCCL refuses two ranks on one device, so this is as much
of the RCCL path as one GPU can run: process-group creation, every collective call the pipeline makes, their stream semantics."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from blok_amd.multi_gpu import FramePipeline, HipBackend
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = 1024
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
# the 4K frame, and a frame of 1/8 of its pixels: about the GPU work one rank of eight has per frame, where the host cost of
# the exchange is what bounds the rate
for (Wd, Ht) in ((3840, 2160), (1344, 768)):
    tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
    cam = W.scene_camera(n, 0, Wd, Ht)
    want = torch.from_numpy(tr.shade_rgba8(cam).reshape(-1).view(np.int32)).cuda()
    for sparse in (0, 1, 2):
        for batch in (1, 2, 4, 8):
            pipe = FramePipeline(HipBackend(tr, cam), Wd, Ht, 0, 1, dist, tile=32, depth=3, sparse=sparse, partition=True, batch=batch)
            for _ in range(8):
                pipe.step()
            pipe.flush(); torch.cuda.synchronize()
            frames = 96
            t = time.perf_counter()
            for _ in range(frames):
                pipe.step()
            pipe.flush(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / frames * 1e3
            ok = all(bool(torch.equal(f, want)) for f in pipe.last_frames)
            extra = f", {pipe.records_gathered / pipe.frames_done:.0f} of {pipe.per_rank} tiles travel per frame" if sparse else ""
            print(f"RCCL world_size 1, {Wd}x{Ht}, {('dense gather', 'sparse exchange', 'sparse exchange, 16-bit codes')[sparse]}, {batch} frame(s) per exchange: "
                  f"frames equal the single-launch frame: {ok}; {dt:.3f} ms/frame incl. host{extra}", flush=True)
            assert ok
    tr.shutdown()
dist.destroy_process_group()
print("rccl smoke ok")
