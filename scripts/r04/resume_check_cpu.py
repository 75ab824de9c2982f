"""CPU (kernel body compiled for the host): the path loop with secondary rays entered from the hit's ancestors (walk_resume + launch pad)
against the same loop with every ray from the root — the planes must be bit-identical — and what it does to the walks' iteration counts."""
import sys, ctypes as C, time
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Wd, Ht = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (192, 108)
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 4
bounces = int(sys.argv[5]) if len(sys.argv) > 5 else 2
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo(W.scene_materials())
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_set_path_resume.argtypes = [C.c_uint32]; L.hh_stat_totals8.argtypes = [C.c_void_p]; L.hh_stat_by_kind.argtypes = [C.c_void_p]
names = ["iterations", "descends", "steps", "ascents", "walks", "resumed", "pad levels"]
for pose in (0, 1, 2):
    cam = W.scene_camera(n, pose, Wd, Ht)
    res = {}
    for mode in (0, 1):
        L.hh_set_path_resume(mode); L.hh_stat_reset()
        t = time.time()
        planes = hk.render_paths(cam, pw.materials, Wd, Ht, spp=spp, max_bounces=bounces, frame_index=3)
        tot = np.zeros((8, 8), dtype=np.uint64); L.hh_stat_totals8(C.c_void_p(tot.ctypes.data))
        kt = np.zeros((3, 8), dtype=np.uint64); L.hh_stat_by_kind(C.c_void_p(kt.ctypes.data))
        res[mode] = (planes, tot.sum(axis=1), time.time() - t, kt)
    same = all(np.array_equal(res[0][0][k].view(np.uint32), res[1][0][k].view(np.uint32)) for k in res[0][0])
    print(f"pose {'ABC'[pose]} {n}^3 {Wd}x{Ht} {spp} spp {bounces} bounces: planes bit-identical: {same}")
    for mode in (0, 1):
        print("   resume", mode, {k: int(v) for k, v in zip(names, res[mode][1])}, f"{res[mode][2]:.1f} s")
        for k, kind in enumerate(("primary", "shadow", "bounce")):
            kt = res[mode][3][k]
            print(f"      {kind:8s} walks {int(kt[4]):8d}  per walk: iterations {kt[0] / max(1, kt[4]):6.2f} descends {kt[1] / max(1, kt[4]):5.2f} steps {kt[2] / max(1, kt[4]):6.2f} ascents {kt[3] / max(1, kt[4]):5.2f} resumed {kt[5] / max(1, kt[4]):4.2f} pad levels {kt[6] / max(1, kt[4]):4.2f} start voxel not verified {kt[7] / max(1, kt[4]):5.3f}")
    assert same
L.hh_set_path_resume(0)
