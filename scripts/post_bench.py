"""Per-pass timing of the image-space chain at 4K over the benchmark world (one 1-spp path-traced frame sequence),
with the algorithmic bytes per pixel of every pass and the fraction of the 8 TB/s HBM roofline.  Run on the GPU box."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer

n, Wd, Ht, seed = 1024, 3840, 2160, 0xB10C0001
cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, seed); cm.rebuild_dirty_chunks()
mats = W.scene_materials(seed)
pw = cm.pack_chunks_to_gpu_svo(mats)
tr = HipTracer(Wd, Ht).init(); tr.add_world(pw)
px = Wd * Ht
P = {k: torch.zeros((px, 4), dtype=torch.float32, device="cuda") for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
den = torch.zeros((px, 4), dtype=torch.float32, device="cuda"); taa = torch.zeros_like(den)
ldr = torch.zeros(px, dtype=torch.int32, device="cuda"); sharp = torch.zeros_like(ldr)
base = W.scene_camera(n, 0, Wd, Ht, seed)
cams = []
for k in range(6):
    c = base.copy(); c["pos"][0][0] += 0.2 * k; cams.append(c)
ev = lambda: torch.cuda.Event(enable_timing=True)
rows = {"paths 1 spp": [], "denoise (temporal + variance + 4 a-trous)": [], "taa": [], "tonemap": [], "sharpen": []}
for k, cam in enumerate(cams):
    e = [ev() for _ in range(6)]
    e[0].record()
    tr.trace_paths_device(cam, P["color"].data_ptr(), spp=1, max_bounces=2, frame_index=k, world_pos_ptr=P["world_pos"].data_ptr(),
                          normal_roughness_ptr=P["normal_roughness"].data_ptr(), albedo_metallic_ptr=P["albedo_metallic"].data_ptr())
    e[1].record()
    tr.denoise_device(P["color"].data_ptr(), P["world_pos"].data_ptr(), P["normal_roughness"].data_ptr(), W.view_proj_from_camera(cams[max(k - 1, 0)]), k, den.data_ptr())
    e[2].record()
    tr.taa_device(den.data_ptr(), taa.data_ptr(), k)
    e[3].record()
    tr.tonemap_device(taa.data_ptr(), ldr.data_ptr())
    e[4].record()
    tr.sharpen_device(ldr.data_ptr(), sharp.data_ptr())
    e[5].record()
    torch.cuda.synchronize()
    if k >= 2:
        for name, i in zip(rows, range(5)):
            rows[name].append(e[i].elapsed_time(e[i + 1]))
# algorithmic bytes per pixel (read + write, each plane once per pass): see DESIGN.md
# temporal: colour, position, normal in (48) + history colour, position, unit normal, moments, length in (74) + colour, moments,
#           length, position, unit normal, motion out (62); variance: 16 + 8 + 2 + 16 + 16 in, 4 out; a-trous: 16 + 4 + 16 + 16 in, 16 out
alg = {"denoise (temporal + variance + 4 a-trous)": (48 + 74 + 62) + 62 + 4 * 68,
       "taa": 16 + 16 + 4 + 32, "tonemap": 16 + 4, "sharpen": 4 + 4}
hl = tr.denoise_state()[2]
print(f"4K ({Wd}x{Ht}), 1024^3 world, history length > 1 on {(hl > 1).mean() * 100:.0f} % of the pixels")
for name, ms in rows.items():
    t = float(np.mean(ms))
    extra = ""
    if name in alg:
        gbs = alg[name] * px / (t * 1e-3) / 1e9
        extra = f"  {alg[name]} B/pixel algorithmic -> {gbs:.0f} GB/s = {gbs / 8000 * 100:.0f} % of 8 TB/s"
    print(f"{name:45s} {t:7.3f} ms{extra}")
tr.shutdown()
