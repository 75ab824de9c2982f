"""Where the path kernel's wave time goes by kind of ray: a diagnostic build (-DBLOK_PATH_CLOCKS: scripts/build_variant.sh pathclocks -DBLOK_PATH_CLOCKS)
books the clocks of every walk round under primary / shadow / bounce, with the number of rounds and of active lanes.  4K, 64 spp, 2 bounces."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
color = torch.empty((Wd * Ht, 4), dtype=torch.float32, device="cuda")
acc = torch.zeros(12, dtype=torch.int64, device="cuda")
tr.set_timing(True)
for pose in (0, 1):
    cam = W.scene_camera(n, pose, Wd, Ht, seed)
    for bounces in (2,):
        tr.set_debug_wave_clocks(0)
        tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=bounces, frame_index=1); torch.cuda.synchronize()
        acc.zero_(); tr.set_debug_wave_clocks(acc.data_ptr())
        tr.trace_paths_device(cam, color.data_ptr(), spp=spp, max_bounces=bounces, frame_index=1); torch.cuda.synchronize()
        ms = tr.last_kernel_ms()
        whole = acc.cpu().numpy().astype(np.float64); a = whole[:9].reshape(3, 3)
        total = a[:, 0].sum()
        print(f"pose {'ABC'[pose]}, {spp} spp, {bounces} bounces: {ms:.2f} ms; wave clocks in walks by kind (primary, shadow, bounce): "
              + ", ".join(f"{a[k, 0] / total * 100:.1f} %" for k in range(3))
              + "; rounds: " + ", ".join(f"{a[k, 1] / 1e6:.2f} M" for k in range(3))
              + "; clocks per round: " + ", ".join(f"{a[k, 0] * 16 / max(a[k, 1], 1):.0f}" for k in range(3))
              + "; active lanes per round: " + ", ".join(f"{a[k, 2] / max(a[k, 1], 1):.1f}" for k in range(3))
              + (f"; walks are {total / whole[9] * 100:.1f} % of the {whole[10] / 1e3:.1f} K waves' clocks ({whole[9] * 16 / max(whole[10], 1):.0f} per wave)" if whole[9] else ""), flush=True)
tr.set_debug_wave_clocks(0)
tr.shutdown()
