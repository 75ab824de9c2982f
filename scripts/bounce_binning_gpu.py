"""Hardware measurement behind the wavefront question for BASELINE configs[4]: what do the SECOND-segment (bounce) rays of one
sample of the 4K frame cost when 64 of them are walked by one wave (a) in pixel order (what a per-pixel kernel sees), (b)
compacted (only the pixels whose path continues), (c) compacted and binned by direction octant, (d) binned by octant and sorted
by the Morton code of the origin's 16^3 cell, (e) additionally sorted by direction (octahedral 8x8 bins)?  Bounce rays are
generated here the way raygen.rgen:338-376 does for a diffuse surface (cosine-weighted around the face normal, origin pushed
out by 0.002), from the GPU's own first hits; numpy's RNG stands in for the shader's PCG (same distribution).  Timing = HIP
events around the Rays-mode trace kernel (blok_hip_trace_rays), rays already on the device."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from blok_amd._ffi import RAY
n, Wd, Ht = 1024, 3840, 2160
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
cam = W.scene_camera(n, pose, Wd, Ht)
hits = tr.draw_frame(cam).reshape(-1)
c = cam[0]
# primary ray of every pixel (double is fine here: only the statistics of the bounce rays matter)
y, x = np.mgrid[0:Ht, 0:Wd]
u = (2 * (x.reshape(-1) + 0.5) / Wd - 1) * float(c["tan_half_fov"]) * float(c["aspect"]); v = (1 - 2 * (y.reshape(-1) + 0.5) / Ht) * float(c["tan_half_fov"])
d = c["fwd"][None, :] + c["right"][None, :] * u[:, None] + c["up"][None, :] * v[:, None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
hit = hits["hit"] == 1
pos = c["pos"][None, :] + d * hits["t"][:, None]
normals = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
N = normals[np.minimum(hits["face"], 5)]
rng = np.random.default_rng(1)
r1, r2 = rng.random(len(hits)), rng.random(len(hits))
phi, ct = 2 * np.pi * r1, np.sqrt(1 - r2); st = np.sqrt(r2)
t1 = np.where(np.abs(N[:, [1]]) < 0.99, np.cross(N, [0.0, 1.0, 0.0]), np.cross(N, [1.0, 0.0, 0.0])); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
t2 = np.cross(N, t1)
nd = t1 * (st * np.cos(phi))[:, None] + t2 * (st * np.sin(phi))[:, None] + N * ct[:, None]
rays = np.zeros(len(hits), dtype=RAY)
rays["org"] = (pos + N * 0.002).astype(np.float32); rays["dir"] = nd.astype(np.float32); rays["tmin"] = 0.001; rays["tmax"] = 10000.0
dead = ~hit
rays["tmax"][dead] = 0.0005          # a pixel whose primary ray missed has no bounce ray: tmax < tmin ends its walk at once
print(f"pose {pose}: {hit.sum()} of {len(hits)} pixels have a bounce ray", flush=True)

def timed(r, label, n_real):
    tr.set_timing(True)
    ms = []
    for _ in range(4):
        tr.trace_rays(r); ms.append(tr.last_kernel_ms())
    tr.set_timing(False)
    m = float(np.mean(ms[1:]))
    print(f"  {label:70s} {m:7.3f} ms for {n_real} rays = {n_real / m / 1e3:7.1f} Mrays/s", flush=True)
    return m

# (a) pixel order, 8x8 tiles per wave like the path kernel: reorder the frame into 8x8 tile order
tile_order = np.arange(Wd * Ht).reshape(Ht // 8, 8, Wd // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
a = rays[tile_order]
timed(a, "(a) 8x8-pixel tiles, dead lanes in place (the per-pixel kernel's view)", int(hit.sum()))
live = a[a["tmax"] > 1.0]
timed(live, "(b) compacted in tile order", len(live))
octant = (live["dir"][:, 0] < 0).astype(np.int64) | ((live["dir"][:, 1] < 0).astype(np.int64) << 1) | ((live["dir"][:, 2] < 0).astype(np.int64) << 2)
timed(live[np.argsort(octant, kind="stable")], "(c) compacted, binned by direction octant (tile order inside)", len(live))
cell = np.clip((live["org"] / 16).astype(np.int64), 0, 63)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v
morton = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
timed(live[np.lexsort((morton, octant))], "(d) binned by octant, sorted by the origin's 16^3-cell Morton code", len(live))
ad = np.abs(live["dir"]); o = live["dir"][:, :2] / ad.sum(axis=1, keepdims=True)
dbin = (np.clip(((o[:, 0] * 0.5 + 0.5) * 8).astype(np.int64), 0, 7) << 3) | np.clip(((o[:, 1] * 0.5 + 0.5) * 8).astype(np.int64), 0, 7)
timed(live[np.lexsort((morton, dbin, octant))], "(e) octant, 8x8 direction bins, then origin Morton", len(live))
timed(live[np.lexsort((dbin, morton >> 6, octant))], "(f) octant, origin 64^3-cell Morton, then direction bin", len(live))
tr.shutdown()
