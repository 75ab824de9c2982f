"""Frames ALONE on the device with the camera in motion: 4K over the 1024^3 benchmark world, the camera orbiting the world's centre by
`step` degrees per frame (bench.py --orbit), one frame at a time with a host synchronisation after each — what roofline.frac_moving of
the bench line is measured on.  Prints, with the carried order (blok_hip_set_moving_order) on and off: the frame's launch by HIP events,
and the wall-clock period per frame including the order's upkeep behind the frame and the synchronisation.
    python3 scripts/r03/solitary_orbit.py [step_deg=1] [frames=64] [moving_order=both|0|1]
Under rocprofv3 --kernel-trace --stats the kernel table shows the upkeep kernels (order_class / order_scan / order_scatter) beside joint_kernel."""
import sys
import time
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
step = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
which = sys.argv[3] if len(sys.argv) > 3 else "both"
begin = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0        # yaw / dolly: where the motion begins (degrees / voxels)
motion = sys.argv[4] if len(sys.argv) > 4 else "orbit"      # orbit: around the world's centre; yaw: turning on the spot; dolly: flying forward by `step` voxels per frame
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")

def orbit_cam(deg):
    nf = float(n)
    if motion != "orbit":
        pos = np.array([-0.35 * nf, 0.85 * nf, -0.35 * nf]); centre = np.array([0.5 * nf, 0.25 * nf, 0.5 * nf])
        if motion == "yaw":
            d = centre - pos; a = np.radians(deg - 3.0 + begin)
            target = pos + np.array([d[0] * np.cos(a) - d[2] * np.sin(a), d[1], d[0] * np.sin(a) + d[2] * np.cos(a)])
            return W.camera_look_at(tuple(float(v) for v in pos), tuple(float(v) for v in target), 60.0, Wd, Ht)
        fwd = (centre - pos) / np.linalg.norm(centre - pos)
        p = pos + fwd * (deg - 3.0 + begin)
        return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in (p + fwd * 100.0)), 60.0, Wd, Ht)
    centre = np.array([0.5 * nf, 0.25 * nf, 0.5 * nf]); start = np.array([-0.35 * nf, 0.85 * nf, -0.35 * nf]) - centre
    a = np.radians(deg)
    p = centre + np.array([start[0] * np.cos(a) - start[2] * np.sin(a), start[1], start[0] * np.sin(a) + start[2] * np.cos(a)])
    return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), 60.0, Wd, Ht)

for mo in ([1, 0] if which == "both" else [int(which)]):
    tr.set_moving_order(bool(mo)); tr.set_timing(True)
    ms, uses, shifts = [], [], []
    t_begin = None
    for k in range(-4, frames):
        if k == 0:
            torch.cuda.synchronize(); t_begin = time.perf_counter()
        tr.draw_frame_device(orbit_cam(3.0 + step * (k + 4)), hits.data_ptr(), rgba.data_ptr())
        torch.cuda.synchronize()
        if k >= 0:
            ms.append(tr.last_kernel_ms()); u = tr.last_order_use(); uses.append(u[0]); shifts.append((u[1], u[2]))
    period = (time.perf_counter() - t_begin) / frames * 1e3
    tr.set_timing(False)
    print(f"{motion} {step} per frame, moving order {'on ' if mo else 'off'}: launch {np.mean(ms) * 1e3:6.1f} us (median {np.median(ms) * 1e3:6.1f}, max {np.max(ms) * 1e3:6.1f}); "
          f"wall period {period * 1e3:6.1f} us per frame incl. upkeep and synchronisation; carried order on {sum(1 for u in uses if u == 2)} of {frames} frames, "
          f"shifts {sorted(set(shifts))[:6]}", flush=True)
tr.shutdown()
