"""[round 4: the BLOK_WALK_SKIP block was deleted from trace_core.h (it lost, profiles/r03_skip_ahead_estimate.txt); this script needs the tree at feeb08d to run]
Would "leave the node at once when nothing else of it lies ahead" pay?  The walk's body compiled for the CPU with and without
-DBLOK_WALK_SKIP (trace_core.h), bands of the 4K benchmark frame behind a beam-like start parameter: same records required, then
iterations per ray and per wave (the longest ray of each 8x8 tile), and the event totals by level.
    python3 scripts/r03/skip_ahead_estimate.py [pose=0]"""
import sys, os, subprocess, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from pathlib import Path
from blok_amd import world as W
from tests import harness_ffi as H, oracle_ffi as O
ROOT = Path('.').resolve(); SRC = ROOT / "tests/host_harness"
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 0
libs = {}
for name, flags in (("plain", []), ("skip", ["-DBLOK_WALK_SKIP"])):
    out = ROOT / "build/skip" / f"libhh_{name}.so"
    subprocess.run(["g++", "-O2", "-std=c++20", "-fPIC", "-ffp-contract=off", f"-I{ROOT / 'include'}", f"-I{ROOT / 'blok_amd/csrc/hip'}", f"-I{SRC}", "-shared", "-o", os.fspath(out),
                    os.fspath(SRC / "harness.cpp"), os.fspath(ROOT / "blok_amd/csrc/hip/tree_build.cpp")] + flags, check=True)
    libs[name] = C.CDLL(os.fspath(out))
n, Wd, Ht, B, seed = 1024, 3840, 2160, 32, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks(); pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
cam = W.scene_camera(n, pose, Wd, Ht, seed)
res = {}
for name, L in libs.items():
    L.hh_build.restype = C.c_void_p
    L.hh_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    why = C.c_char_p()
    h = L.hh_build(C.c_void_p(pw.nodes.ctypes.data), len(pw.nodes), C.c_void_p(pw.sub_chunks.ctypes.data), len(pw.sub_chunks), C.byref(why))
    L.hh_trace_rect_stats.argtypes = [C.c_void_p] * 2 + [C.c_uint32] * 6 + [C.c_void_p] * 3
    L.hh_stat_totals.argtypes = [C.c_void_p]
    tot_all = np.zeros((5, 8), dtype=np.uint64); ray_it = 0; wave_it = 0; waves = 0; outs = []
    for y0 in range(108, Ht - B + 1, 216):
        w, hgt = Wd, B
        out = np.zeros(w * hgt, dtype=O.HIT); it = np.zeros(w * hgt, dtype=np.uint32)
        if name == "plain":
            L.hh_trace_rect_stats(h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, hgt, None, C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
            t = np.where(out['hit'] == 1, out['t'], np.inf).reshape(hgt, w // B, B)
            ts = np.repeat(np.maximum(t.min(axis=(0, 2)) - 2.0, 0)[None, :], hgt, axis=0).repeat(B, axis=1)
            res.setdefault("tstart", {})[y0] = np.where(np.isfinite(ts), ts, 9999.0).astype(np.float32)
        tsf = res["tstart"][y0]
        out = np.zeros(w * hgt, dtype=O.HIT); it = np.zeros(w * hgt, dtype=np.uint32)
        L.hh_trace_rect_stats(h, C.c_void_p(cam.ctypes.data), Wd, Ht, 0, y0, w, hgt, C.c_void_p(tsf.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(it.ctypes.data))
        live = (tsf < 9000.0)
        it = np.where(live.reshape(-1), it, 0)
        tot = np.zeros((5, 8), dtype=np.uint64); L.hh_stat_totals(C.c_void_p(tot.ctypes.data)); tot_all += tot
        ray_it += int(it.sum()); wmax = it.reshape(hgt // 8, 8, w // 8, 8).max(axis=(1, 3)); wave_it += int(wmax.sum()); waves += int((wmax > 0).sum())
        outs.append(out.copy())
    res[name] = dict(ray_it=ray_it, wave_it=wave_it, waves=waves, tot=tot_all, outs=outs)
for a, b in zip(res["plain"]["outs"], res["skip"]["outs"]):
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), "records differ"
print(f"pose {pose}: records identical on {len(res['plain']['outs'])} bands of {Wd}x{B}")
for name in ("plain", "skip"):
    r = res[name]
    print(f"{name:6s} iterations per ray-in-live-tiles (sum) {r['ray_it']}, wave-iterations {r['wave_it']} over {r['waves']} walking waves = {r['wave_it'] / max(1, r['waves']):.2f} per wave")
    for e, nm in enumerate(["iter", "descend", "step", "ascend", "walks / skips by level+1"]):
        print(f"        {nm:26s}", r['tot'][e].tolist())
p, s = res["plain"], res["skip"]
print(f"wave-iterations {s['wave_it'] / p['wave_it']:.3f} x, ray iterations {s['ray_it'] / p['ray_it']:.3f} x")
