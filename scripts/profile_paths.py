"""Workload for rocprofv3: a few launches of the path kernel (4K, 8 spp, 2 bounces) on the 1024^3 scene."""
import sys
sys.path.insert(0, '.')
import torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world
cm, pw = make_scene_world(1024)
tr = HipTracer(3840, 2160).init(); tr.add_world(pw)
cam = W.scene_camera(1024, 0, 3840, 2160)
color = torch.empty((3840 * 2160, 4), dtype=torch.float32, device="cuda")
for f in range(4):
    tr.trace_paths_device(cam, color.data_ptr(), spp=8, max_bounces=2, frame_index=f)
torch.cuda.synchronize()
