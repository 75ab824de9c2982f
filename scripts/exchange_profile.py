"""One-GPU rehearsal of what the ROOT of N ranks does per frame besides tracing its share: run under rocprofv3 --kernel-trace --stats.
A one-rank RCCL group; the rank owns every tile (so compaction is N times a rank's; scatter / fill / un-permute are the root's)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from blok_amd.multi_gpu import FramePipeline, HipBackend
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29633")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, Wd, Ht = 1024, 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(n, 0, Wd, Ht)
sparse = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pipe = FramePipeline(HipBackend(tr, cam), Wd, Ht, 0, 1, dist, tile=32, depth=3, sparse=sparse, partition=True, batch=8)
for _ in range(16):
    pipe.step()
pipe.flush(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(96):
    pipe.step()
pipe.flush(); torch.cuda.synchronize()
print(f"sparse={sparse}: {(time.perf_counter() - t) / 96 * 1e3:.4f} ms/frame", flush=True)
dist.destroy_process_group(); tr.shutdown()
