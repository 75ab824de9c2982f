"""The launch alone (HIP events, one frame at a time) for callers whose view is not one resting camera: two fixed views alternating, a cycle
of three, the first frame of a never-seen view (cold), a zoom.  4K over 1024^3.  usage: views_probe.py [frames]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world, SEED
n, Wd, Ht = 1024, 3840, 2160
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 48
cm, pw = make_scene_world(n)
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw); tr.set_timing(True)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
centre = np.array([512.0, 256.0, 512.0]); start = np.array([-358.0, 870.0, -358.0]) - centre
def orbit(deg, fov=60.0):
    r = np.radians(deg)
    p = centre + np.array([start[0] * np.cos(r) - start[2] * np.sin(r), start[1], start[0] * np.sin(r) + start[2] * np.cos(r)])
    return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), fov, Wd, Ht)
def run(cams, tag, skip):
    ms = []; uses = []
    for cam in cams:
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize()
        ms.append(tr.last_kernel_ms()); uses.append(tr.last_order_use()[0])
    m = np.array(ms[skip:])
    print(f"{tag:58s} mean {m.mean():.4f} ms  median {np.median(m):.4f}  max {m.max():.4f}   order use of the last 8: {uses[-8:]}", flush=True)
    return m
for ordering in (True, False):
    tr.set_tile_ordering(8 if ordering else 0)
    print("== ordering", "on" if ordering else "off")
    A, B, C = W.scene_camera(n, 0, Wd, Ht, SEED), orbit(6.0), W.scene_camera(n, 2, Wd, Ht, SEED)
    run([A] * frames, "one view at rest", 16)
    run([A, B] * (frames // 2), "two views alternating (stereo / cut back and forth)", 16)
    run([A, B, C] * (frames // 3), "three views in a cycle", 18)
    cold = [run([orbit(50.0 + 17.0 * k)], f"cold: first frame of a never-seen view #{k}", 0)[0] for k in range(4)]
    print(f"cold mean {np.mean(cold):.4f} ms")
    run([orbit(0.0, fov=60.0 - 0.25 * k) for k in range(frames)], "zoom in by 0.25 degree of field of view per frame", 8)
tr.shutdown()
