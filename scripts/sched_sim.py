"""Wave-level schedule simulation behind the beam pre-pass: the current loop (one descend-or-step event per lane and
iteration, both paths executed when lanes disagree) against "step, then descend while the new cell is occupied".
Event logs come from the CPU harness with tmin' = (nearest hit of the 32x32 tile) - 2."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
n = 1024
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo()
hk = H.HostKernel(pw.nodes, pw.sub_chunks)
L = H.lib()
L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
L.hh_trace_rect_events.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p, C.c_uint32, C.c_void_p]
Wd, Ht, B, cap = 3840, 2160, 32, 200
cam = W.scene_camera(n, pose, Wd, Ht)
T, D, S = 19, 73, 40          # header, descend path, step path (VALU, from the ISA)
cost = dict(current=0.0, step_then_descend=0.0, descend_then_step=0.0)
waves = 0
for y0 in range(108, Ht - B + 1, 432):
    w, h = Wd, B
    out = np.zeros(w * h, dtype=O.HIT); it = np.zeros(w * h, dtype=np.uint32)
    L.hh_trace_rect_stats(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
    t = np.where(out['hit'] == 1, out['t'], np.inf).reshape(h, w // B, B)
    tmin_tile = t.min(axis=(0, 2))
    ts = np.repeat(np.maximum(tmin_tile - 2.0, 0)[None, :], h, axis=0).repeat(B, axis=1)
    tsf = np.where(np.isfinite(ts), ts, 9999.0).astype(np.float32)
    ev = np.zeros((h, w, cap), dtype=np.uint8)
    L.hh_trace_rect_events(hk.h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, h, C.c_void_p(tsf.ctypes.data), cap, C.c_void_p(ev.ctypes.data))
    kind = ev[..., 1:] & 3                     # entry 0 is the walk-start marker (event 4)
    tiles = kind.reshape(h // 8, 8, w // 8, 8, cap - 1).transpose(0, 2, 1, 3, 4).reshape(-1, 64, cap - 1)
    for seqs in tiles[::16]:
        if not seqs.any():
            continue
        waves += 1
        lens = (seqs != 0).sum(axis=1)
        # current: every round each live lane consumes one event
        for r in range(lens.max()):
            col = seqs[:, r]
            cost["current"] += T + (D if (col == 1).any() else 0) + (S if (col == 2).any() else 0)
        for name in ("step_then_descend", "descend_then_step"):
            ptr = np.zeros(64, dtype=np.int64)
            idx = np.arange(64)
            c = 0.0
            while (ptr < lens).any():
                nxt = np.where(ptr < lens, seqs[idx, np.minimum(ptr, cap - 2)], 0)
                c += 6                                                      # loop overhead
                if name == "step_then_descend":
                    st = nxt == 2
                    if st.any(): c += T + S; ptr[st] += 1
                    while True:
                        nxt = np.where(ptr < lens, seqs[idx, np.minimum(ptr, cap - 2)], 0)
                        de = nxt == 1
                        if not de.any(): break
                        c += T + D; ptr[de] += 1
                else:
                    de = nxt == 1
                    if de.any(): c += T + D; ptr[de] += 1
                    nxt = np.where(ptr < lens, seqs[idx, np.minimum(ptr, cap - 2)], 0)
                    st = nxt == 2
                    if st.any(): c += T + S; ptr[st] += 1
            cost[name] += c
print(f"pose {pose}: walking waves sampled {waves}")
for k, v in cost.items():
    print(f"  {k:22s} {v / waves:8.0f} loop VALU per walking wave  ({v / cost['current']:.2f}x)")
