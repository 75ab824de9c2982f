"""Workload for rocprofv3: launches of the path kernel at BASELINE configs[4] (4K, 64 spp, 2 bounces) on the 1024^3 scene."""
import sys
sys.path.insert(0, '.')
import torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cm, pw = make_scene_world(1024)
tr = HipTracer(3840, 2160).init(); tr.add_world(pw)
cam = W.scene_camera(1024, 0, 3840, 2160)
color = torch.empty((3840 * 2160, 4), dtype=torch.float32, device="cuda")
for f in range(frames):
    tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=2, frame_index=f)
    torch.cuda.synchronize()
