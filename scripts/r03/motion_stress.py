"""Does the carried order ever cost more than it saves?  A camera wandering through and around the 1024^3 world on a smooth random path
(random accelerations in yaw, pitch and velocity; speeds from a crawl to several degrees / tens of voxels per frame), every frame drawn
alone by two contexts — carried order on, and all ordering off — and the launch times compared frame by frame."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
a = HipTracer(Wd, Ht).init(); a.add_world(pw)
b = HipTracer(Wd, Ht).init(); b.add_world(pw); b.set_tile_ordering(0); b.set_moving_order(False)
hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); want = torch.zeros_like(hits)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
pos = np.array([-300.0, 800.0, -300.0]); yaw, pitch = np.radians(45.0), np.radians(-30.0)
vyaw = vpitch = 0.0; vel = np.zeros(3)
rows = []
a.set_timing(True); b.set_timing(True)
for k in range(frames):
    if k % 60 == 0:                                        # a new leg of the journey: another pace
        pace = rng.choice([0.05, 0.3, 1.0, 3.0])
    vyaw = 0.9 * vyaw + 0.1 * rng.normal(0, np.radians(1.0)) * pace
    vpitch = 0.9 * vpitch + 0.1 * rng.normal(0, np.radians(0.5)) * pace
    yaw += vyaw; pitch = float(np.clip(pitch + vpitch, np.radians(-80), np.radians(30)))
    fwd = np.array([np.cos(pitch) * np.cos(yaw), np.sin(pitch), np.cos(pitch) * np.sin(yaw)])
    vel = 0.9 * vel + 0.1 * (fwd * rng.normal(4.0, 4.0) + rng.normal(0, 2.0, 3)) * pace
    pos = np.clip(pos + vel, [-600, 40, -600], [1600, 1200, 1600])
    cam = W.camera_look_at(tuple(float(v) for v in pos), tuple(float(v) for v in pos + fwd * 100.0), 60.0, Wd, Ht)
    a.draw_frame_device(cam, hits.data_ptr(), 0); torch.cuda.synchronize(); ta = a.last_kernel_ms(); fb = a.last_fallback_tiles()
    b.draw_frame_device(cam, want.data_ptr(), 0); torch.cuda.synchronize(); tb = b.last_kernel_ms()
    assert torch.equal(hits, want), k
    rows.append((ta, tb, a.last_order_use()[0], np.degrees(np.hypot(vyaw, vpitch)), float(np.linalg.norm(vel)), fb))
r = np.array(rows)
on, off = r[:, 0] * 1e3, r[:, 1] * 1e3
print(f"{frames} frames: carried order in use on {int((r[:, 2] == 2).sum())}; launch alone, mean {on.mean():.1f} us with the order machinery, {off.mean():.1f} us row-major; "
      f"per frame: better on {int((on < off).sum())}, worse by > 10 % on {int((on > 1.1 * off).sum())}, worst {np.max(on / off):.2f}x", flush=True)
for i in np.argsort(-(on / off))[:6]:
    print(f"   frame {i}: {on[i]:.0f} us against {off[i]:.0f} us row-major; order use {int(r[i, 2])}; turning {r[i, 3]:.2f} deg/frame, moving {r[i, 4]:.1f} voxels/frame; tiles left to the search waves {int(r[i, 5])}", flush=True)
used = r[:, 2] == 2
if used.any():
    fbs = r[used, 5]
    print(f"   tiles left to the search waves on frames in a carried order: median {np.median(fbs):.0f}, 90th percentile {np.percentile(fbs, 90):.0f}, max {fbs.max():.0f}; correlation with launch / row-major ratio {np.corrcoef(fbs, (on / off)[used])[0, 1]:.2f}", flush=True)
a.shutdown(); b.shutdown()
