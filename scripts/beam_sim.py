"""CPU emulation of beam.h's cooperative DFS: node visits per 8x8 tile (sample of tile rows of the 4K frame)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from blok_amd.tracer import HipTracer  # noqa (records only)
from tests import harness_ffi as H
import ctypes as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pose = int(sys.argv[2]) if len(sys.argv) > 2 else 0
STOP = int(sys.argv[3]) if len(sys.argv) > 3 else 2
Wd, Ht = 3840, 2160
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
# tree download from the harness
L.hh_tree_nodes.restype = C.c_void_p; L.hh_tree_nodes.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
cnt = C.c_size_t(); org = (C.c_int32 * 3)()
ptr = L.hh_tree_nodes(hk.h, C.byref(cnt), org)
nodes = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(cnt.value, 4)).copy()
levels = L.hh_levels(hk.h)
origin = np.array(list(org), dtype=np.int64)
cam = W.scene_camera(n, pose, Wd, Ht)[0]
pos = cam['pos'].astype(np.float64); fwd = cam['fwd'].astype(np.float64); right = cam['right'].astype(np.float64); up = cam['up'].astype(np.float64)
tanh, asp = float(cam['tan_half_fov']), float(cam['aspect'])
lanes = np.arange(64)

def direction(px, py):
    u = (2 * px / Wd - 1) * tanh * asp; v = (1 - 2 * py / Ht) * tanh
    return fwd + right * u + up * v

def unit(a): return a / np.linalg.norm(a)

def beam(px0, py0):
    a, b, c, d = direction(px0 - 1, py0 - 1), direction(px0 + 9, py0 - 1), direction(px0 + 9, py0 + 9), direction(px0 - 1, py0 + 9)
    mid = unit(a + b + c + d)
    ns = []
    for p, q in ((a, b), (b, c), (c, d), (d, a)):
        nn = unit(np.cross(p, q)); ns.append(nn if nn @ mid >= 0 else -nn)
    mirror = (3 if mid[0] < 0 else 0) | (0xC if mid[1] < 0 else 0) | (0x30 if mid[2] < 0 else 0)
    child = lanes ^ mirror
    cc = np.stack([child & 3, (child >> 2) & 3, child >> 4], axis=1).astype(np.float64)
    bs = [((cc + (nn > 0)) * nn).sum(axis=1) for nn in ns]
    b4 = ((cc + (mid < 0)) * mid).sum(axis=1)
    level, node, m, best = levels, 0, np.zeros(3, dtype=np.int64), np.inf
    stack = {}
    visits = [0] * 8; fresh = True; cand = None
    while True:
        mlo, mhi, base = int(nodes[node, 0]), int(nodes[node, 1]), int(nodes[node, 2])
        mask = mlo | (mhi << 32)
        shift = 2 * (level - 1); s = float(1 << shift)
        r = (origin + m) - pos
        depth = mid @ r + s * b4
        if fresh:
            visits[level] += 1
            outside = np.zeros(64, dtype=bool)
            for nn, bb in zip(ns, bs): outside |= (nn @ r + s * bb) < -0.05
            filled = ((np.uint64(mask) >> child.astype(np.uint64)) & np.uint64(1)).astype(bool)
            cand = filled & ~outside & ~(depth >= best)
            if level <= STOP:
                if cand.any(): best = min(best, max(depth[cand].min(), 0.0))
                cand[:] = False
        else:
            cand &= ~(depth >= best)
        if not cand.any():
            if level == levels: break
            level += 1
            node, cand = stack[level]
            keep = ~((1 << (2 * level)) - 1)
            m &= keep; fresh = False
            continue
        j = int(np.argmax(cand)); cand[j] = False
        stack[level] = (node, cand.copy())
        cj = j ^ mirror
        node = base + bin(mask & ((1 << cj) - 1)).count('1')
        m = m + (np.array([cj & 3, (cj >> 2) & 3, cj >> 4]) << shift)
        level -= 1; fresh = True
    return best, visits

tot = np.zeros(8); ntiles = 0; sky = 0; per_tile = []
for ty in range(0, Ht // 8, 27):
    for tx in range(0, Wd // 8, 4):
        best, v = beam(tx * 8, ty * 8)
        tot += v; ntiles += 1; sky += not np.isfinite(best); per_tile.append((sum(v), np.isfinite(best)))
pt = np.array(per_tile)
print("tiles", ntiles, "sky", sky, "visits/tile", tot.sum() / ntiles, "by level", np.round(tot / ntiles, 2))
print("visits/tile: sky tiles", pt[pt[:, 1] == 0, 0].mean(), "hit tiles", pt[pt[:, 1] == 1, 0].mean(), "max", pt[:, 0].max())
