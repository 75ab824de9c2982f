"""Workload for rocprofv3: launches of the path kernel (4K, 2 bounces, 1024^3, pose A) with one setting of blok_hip_set_path_start.
    python3 scripts/r04/paths_start_one.py <spp> <frames> <resume 0|1> <wave_tile_beam 0|1>"""
import sys
sys.path.insert(0, '.')
import torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests.conftest import make_scene_world
spp, frames, resume, fine = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
pose = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cm, pw = make_scene_world(1024)
tr = HipTracer(3840, 2160).init(); tr.add_world(pw)
tr.set_path_start(resume, fine)
cam = W.scene_camera(1024, pose, 3840, 2160)
color = torch.empty((3840 * 2160, 4), dtype=torch.float32, device="cuda")
for f in range(frames):
    tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=2, frame_index=f)
    torch.cuda.synchronize()
