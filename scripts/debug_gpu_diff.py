import sys
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests import oracle_ffi as O, harness_ffi as H
n, w, h = 64, 128, 128
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
tr = HipTracer(w, h).init(); st = tr.add_world(pw)
print("levels", st.levels, "origin", list(st.origin), "voxels", st.n_voxels, "nodes", st.n_tree_nodes)
cam = W.scene_camera(n, 0, w, h)
rays = O.primary_rays(cam, w, h)
lat = O.Lattice(pw.nodes, pw.sub_chunks)
ref, _ = lat.trace(rays)
g1 = tr.draw_frame(cam).reshape(-1)
g2 = tr.trace_rays(rays)
print("primary bad", (g1 != ref).sum(), "rays bad", (g2 != ref).sum(), "g1 vs g2", (g1 != g2).sum())
bad = np.nonzero(g2 != ref)[0]
for i in bad[:12]:
    print(i, divmod(i, w), "ref", ref[i], "gpu", g2[i], "ray", rays[i])
print("hit flags: ref", ref['hit'].sum(), "g1", g1['hit'].sum(), "g2", g2['hit'].sum())
