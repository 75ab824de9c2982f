"""N3 end to end: a radius-r sphere brush on the resident 1024^3 volume -> a tree the tracer can walk, wall clock from the
brush call to the return of the rebuild, repeated at different places; under rocprofv3 --kernel-trace --stats this gives the kernels' shares.
    python3 scripts/r04/edit_latency.py [radius=8] [repeats=40]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, seed = 1024, 0xB10C0001
radius = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ids = W.scene_dense(n, seed)
tr = HipTracer(3840, 2160).init()
tr.volume_create((0, 0, 0), (n, n, n), 128, 1.0)
tr.volume_upload((ids != 0).astype(np.float32), ids)
mats = W.scene_materials(seed)
st = tr.volume_rebuild(mats)
cam = W.scene_camera(n, 0, 3840, 2160, seed)
base = tr.draw_frame(cam)
ys, xs = np.nonzero(base["hit"])
rng = np.random.default_rng(1)
edit, rebuild = [], []
for k in range(reps):
    j = rng.integers(len(ys))
    c = tuple(min(max(float(v) + 0.5, radius + 1), n - radius - 2) for v in base["voxel"][ys[j], xs[j]])
    t0 = time.perf_counter(); tr.volume_apply_brush(c, radius, 0.0 if k % 2 == 0 else 1.0, 1 if k % 2 == 0 else 0); t1 = time.perf_counter()
    st = tr.volume_rebuild(mats); t2 = time.perf_counter()
    edit.append(t1 - t0); rebuild.append(t2 - t1)
e, r = np.array(edit[4:]) * 1e3, np.array(rebuild[4:]) * 1e3
print(f"radius {radius:g}, {reps - 4} edits: brush call {e.mean():.3f} ms (median {np.median(e):.3f}) + rebuild {r.mean():.3f} ms (median {np.median(r):.3f}) = {(e + r).mean():.3f} ms "
      f"(median {np.median(e + r):.3f}, max {(e + r).max():.3f}) from the call to a tree the tracer can walk; {st.n_voxels} voxels, {st.n_tree_nodes} nodes", flush=True)
f = tr.draw_frame(cam)
print("frame after the edits differs from the first in", int((f["hit"] != base["hit"]).sum() + ((f["t"] != base["t"]) & (f["hit"] == base["hit"])).sum()), "pixels")
tr.shutdown()
