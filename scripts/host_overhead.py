"""Host-side cost of enqueueing one trace launch through the Python mirror (no synchronisation in the loop)."""
import sys, time
sys.path.insert(0, '.')
import torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from blok_amd.multi_gpu import HipBackend
from tests.conftest import make_scene_world
cm, pw = make_scene_world(64)
tr = HipTracer(64, 64).init(); tr.add_world(pw)
cam = W.scene_camera(64, 0, 64, 64)
hits = torch.empty((64 * 64, 4), dtype=torch.int32, device="cuda"); rgba = torch.empty(64 * 64, dtype=torch.int32, device="cuda")
be = HipBackend(tr, cam)
s = torch.cuda.current_stream().cuda_stream
for name, fn in (("draw_frame_device", lambda: tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=s)),
                 ("HipBackend.trace_full", lambda: be.trace_full(hits, rgba, s))):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name}: {(t1 - t0) / 3000 * 1e6:.1f} us per call (enqueue only), {(time.perf_counter() - t0) / 3000 * 1e6:.1f} us incl. drain")
