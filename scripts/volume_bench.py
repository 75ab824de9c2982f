"""Edit -> rebuild latency of the device-resident dense store on the benchmark world:
the 1024^3 scene as a dense volume in HBM (4 GiB density + 4 GiB material ids + 128 MiB brick masks), a sphere
brush, the rebuild of the traversal structure, and a 4K frame of the edited world.  Run on the GPU box."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from blok_amd.tracer import HipTracer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seed = 0xB10C0001
t = time.perf_counter(); ids = W.scene_dense(n, seed); print(f"host: dense scene {n}^3 generated in {time.perf_counter() - t:.2f} s, {int((ids != 0).sum())} voxels", flush=True)
tr = HipTracer(3840, 2160).init()
tr.volume_create((0, 0, 0), (n, n, n), 128, 1.0)
dens = (ids != 0).astype(np.float32)
t = time.perf_counter(); tr.volume_upload(dens, ids); up = time.perf_counter() - t
print(f"upload of both arrays + all brick masks: {up * 1e3:.1f} ms ({2 * ids.nbytes / up / 1e9:.1f} GB/s host->device incl. mask kernel)", flush=True)
del dens
mats = W.scene_materials(seed)
for rep in range(3):
    t = time.perf_counter(); st = tr.volume_rebuild(mats); dt = time.perf_counter() - t
    print(f"rebuild #{rep}: {dt * 1e3:.2f} ms -> {st.n_voxels} voxels, {st.n_tree_nodes} nodes, {st.levels} levels", flush=True)
cam = W.scene_camera(n, 0, 3840, 2160, seed)
base = tr.draw_frame(cam)
rng = np.random.default_rng(1)
for radius in (8.0, 32.0, 96.0):
    ys, xs = np.nonzero(base["hit"])                     # dig where a ray of the current frame lands
    k = rng.integers(len(ys))
    c = tuple(min(max(float(v) + 0.5, radius + 1), n - radius - 2) for v in base["voxel"][ys[k], xs[k]])     # keep the brush in the box
    t = time.perf_counter(); tr.volume_apply_brush(c, radius, 0.0, 1); tb = time.perf_counter() - t
    t = time.perf_counter(); st = tr.volume_rebuild(mats); trb = time.perf_counter() - t
    t = time.perf_counter(); f = tr.draw_frame(cam); tf = time.perf_counter() - t
    changed = int((f["hit"] != base["hit"]).sum() + ((f["t"] != base["t"]) & (f["hit"] == base["hit"])).sum())
    print(f"brush r={radius:g} (subtract): edit (enqueued) {tb * 1e3:.2f} ms + rebuild {trb * 1e3:.2f} ms = {(tb + trb) * 1e3:.2f} ms from the call to a tree the tracer can walk, voxels now {st.n_voxels}; frame differs in {changed} pixels", flush=True)
    base = f
tr.shutdown()
