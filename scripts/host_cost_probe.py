"""Host time per call of the pieces one distributed frame is made of (enqueue only, one GPU, RCCL group of one rank)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from blok_amd.multi_gpu import FramePipeline, HipBackend
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29632")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = 1024
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
Wd, Ht = 1344, 768
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(n, 0, Wd, Ht)
b = HipBackend(tr, cam)
pipe = FramePipeline(b, Wd, Ht, 0, 1, dist, tile=32, depth=3, sparse=True, partition=True, batch=1)
s = pipe.streams[0]
K = 300

def timed(name, fn, sync_every=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    acc = 0.0
    for i in range(K):
        t = time.perf_counter(); fn(); acc += time.perf_counter() - t
        if i % sync_every == sync_every - 1:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{name:48s} {acc / K * 1e6:8.1f} us per call (host)", flush=True)

h = s.cuda_stream
timed("trace_tiles (beam + trace launches)", lambda: b.trace_tiles(32, 0, 1, pipe._hits[0][0], pipe.rgba[0][0], h))
timed("trace_full", lambda: b.trace_full(pipe._hits[0][0], pipe.rgba[0][0], h))
timed("compact", lambda: b.compact(pipe.rgba[0][0], 32, pipe.mine, pipe.compacted[0], h))
def ctx():
    with torch.cuda.stream(s):
        pass
timed("with torch.cuda.stream(s): pass", ctx)
gl = [pipe.gathered[0][0][:pipe.n_tile_px]] if True else None
def gather():
    with torch.cuda.stream(s):
        w = dist.gather(pipe.rgba[0].view(-1)[:pipe.n_tile_px], gather_list=[pipe.gathered[0][0][:pipe.n_tile_px]], dst=0, async_op=True)
        w.wait()
timed("dist.gather async + wait (8 MB... 4 MB)", gather)
def allred():
    with torch.cuda.stream(s):
        w = dist.all_reduce(pipe.smax[0], op=dist.ReduceOp.MAX, async_op=True); w.wait()
timed("dist.all_reduce MAX of one int", allred)
def smax_copy():
    with torch.cuda.stream(s):
        pipe.smax[0][:1].copy_(pipe.compacted[0][:1]); pipe.smax_host[0].copy_(pipe.smax[0], non_blocking=True); pipe.smax_event[0].record()
timed("smax zero + copy + D2H + event", smax_copy)
timed("untile", lambda: b.untile(pipe.gathered[0].view(-1), 4, 32, 1, pipe.per_rank, pipe._frame[0][0], h))
timed("scatter", lambda: b.scatter(pipe.gathered[0].view(-1), 1, pipe.gathered[0].shape[1], 32, 409, pipe._frame[0][0], h))
timed("pipe.step (sparse, batch 1)", pipe.step, sync_every=1000)
pipe.flush()
p2 = FramePipeline(b, Wd, Ht, 0, 1, dist, tile=32, depth=3, sparse=False, partition=True, batch=1)
timed("pipe.step (dense, batch 1)", p2.step, sync_every=1000)
p2.flush()
p3 = FramePipeline(b, Wd, Ht, 0, 1, None, tile=32, depth=3)
timed("pipe.step (no partition)", p3.step, sync_every=1000)
p3.flush()
dist.destroy_process_group(); tr.shutdown()
