"""Which pixels of the image-space chain differ from the oracle on the GPU, by how much, and why?  (tests/test_post.py accepted 0.1 % of
the pixels of every plane outside 1e-5 + 1e-4 |ref| without naming them.)  1920x1080 over the 256^3 scene, two frames, device chain fed
with the device's own G-buffer against the oracle chain fed with the same planes."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blok_amd import world as W
from blok_amd.tracer import HipTracer
from tests import oracle_ffi as O
SEED = 0xB10C0001
w, h = 1920, 1080
cm = W.ChunkManager(128, 1.0); cm.generate_scene(256, SEED); cm.rebuild_dirty_chunks()
pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(SEED))
tr = HipTracer(w, h).init(); tr.add_world(pw)
o = O.OracleDenoiser(w, h)
n = w * h
P = {k: torch.zeros((n, 4), dtype=torch.float32, device="cuda") for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
den = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); res = torch.zeros_like(den)
base = W.scene_camera(256, 0, w, h, SEED)
cams = [base, base.copy(), base.copy()]
cams[1]["pos"][0][0] += 0.3; cams[2]["pos"][0][0] += 0.6
def ulps(a, b):
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai); bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)
for k, cam in enumerate(cams):
    tr.trace_paths_device(cam, P["color"].data_ptr(), spp=1, max_bounces=2, frame_index=k, world_pos_ptr=P["world_pos"].data_ptr(),
                          normal_roughness_ptr=P["normal_roughness"].data_ptr(), albedo_metallic_ptr=P["albedo_metallic"].data_ptr())
    prev = W.view_proj_from_camera(cams[max(k - 1, 0)])
    tr.denoise_device(P["color"].data_ptr(), P["world_pos"].data_ptr(), P["normal_roughness"].data_ptr(), prev, k, den.data_ptr())
    tr.taa_device(den.data_ptr(), res.data_ptr(), k)
    torch.cuda.synchronize()
    host = {name: t.cpu().numpy().reshape(h, w, 4) for name, t in P.items()}
    ref = o.denoise(host["color"], host["world_pos"], host["normal_roughness"], prev, k)
    ref_res = o.taa(ref, k)
    hist, mom, hl, var, mot = tr.denoise_state()
    flip = hl != o.prev["hist_len"]
    near = flip.copy()
    for r in range(1, 17):                               # a-trous reaches 16 pixels (5 iterations, steps 1..16)
        near[r:, :] |= flip[:-r, :]; near[:-r, :] |= flip[r:, :]; near[:, r:] |= flip[:, :-r]; near[:, :-r] |= flip[:, r:]
    print(f"frame {k}: history-length decisions that differ: {int(flip.sum())} pixels of {n}")
    for got, want, what in ((var[..., None], o.variance[..., None], "variance"), (mom, o.prev["moments"], "moments"), (hist, o.prev["color"], "history colour"),
                            (den.cpu().numpy().reshape(h, w, 4), ref, "denoised"), (res.cpu().numpy().reshape(h, w, 4), ref_res, "resolved (TAA)")):
        got = np.ascontiguousarray(got, dtype=np.float32); want = np.ascontiguousarray(want, dtype=np.float32)
        err = np.abs(got - want)
        tight = (err <= 1e-5 + 1e-4 * np.abs(want)).reshape(h, w, -1).all(axis=2)
        loose = (err <= 1e-4 + 1e-3 * np.abs(want)).reshape(h, w, -1).all(axis=2)
        u = ulps(got, want).reshape(h, w, -1).max(axis=2)
        bad = ~tight
        print(f"   {what:16s} outside 1e-5+1e-4|ref|: {int(bad.sum()):6d} ({bad.mean():.5%}), of them within 16 px of a differing decision: {int((bad & near).sum()):6d}; "
              f"outside 1e-4+1e-3|ref|: {int((~loose).sum()):6d}; max |err| {err.max():.3e}; ulps: median {np.median(u):.0f}, 99.9% {np.percentile(u, 99.9):.0f}, max {u.max()}")
tr.shutdown()
