"""The window bench.py measures roofline.frac_moving on: 20 frames at rest, then 18 frames of the 1-degree orbit starting at its turning point (arc[2], arc[1],
arc[0], arc[1], ...), each alone; per frame the launch's ms and what order it walked in.  For A/B of libraries (BLOK_HIP_LIB)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world, SEED
n, Wd, Ht = 1024, 3840, 2160
cm, pw = make_scene_world(n)
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw); tr.set_timing(True)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
centre = np.array([512.0, 256.0, 512.0]); start = np.array([-358.4, 870.4, -358.4]) - centre
def arc(i):
    a = np.radians(1.0 * i)
    p = centre + np.array([start[0] * np.cos(a) - start[2] * np.sin(a), start[1], start[0] * np.sin(a) + start[2] * np.cos(a)])
    return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), 60.0, Wd, Ht)
cams = [arc(i) for i in range(64)]; cams = cams + cams[-2:0:-1]
for rep in range(3):
    for k in range(52):
        tr.draw_frame_device(W.scene_camera(n, 0, Wd, Ht, SEED), hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize()
    rest = tr.last_kernel_ms()
    ms = []; use = []
    for k in range(-2, 16):
        tr.draw_frame_device(cams[k % len(cams)], hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize()
        ms.append(tr.last_kernel_ms()); use.append(tr.last_order_use()[0])
    print(f"rep {rep}: at rest {rest:.4f}; orbit window mean of the 16 {np.mean(ms[2:]):.4f} ms; per frame {[round(m, 3) for m in ms]}; order use {use}", flush=True)
tr.shutdown()
