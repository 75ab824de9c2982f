"""the image-space chain behind the path tracer — temporal accumulation, variance, a-trous,
TAA, sharpen (reference assets/shaders/temporal_reproject.comp, variance.comp, atrous.comp, taa.comp, sharpen.comp).

The reference holds no fixture for these shaders and they cannot run here, so the oracle (oracle/blok_oracle_post.cpp)
is a literal restatement pinned only by the hand-computed cases below ("parity unpinned" in its header).  The product
(blok_amd/csrc/hip/post_core.h) is checked against it twice: compiled for the CPU it must agree bit for bit over a
multi-frame sequence with camera motion; on the GPU it may differ through expf in variance.comp's depth weight
(a few ulp, which can flip that shader's `weight > 0.01` test on isolated pixels), so the GPU test requires
|delta| <= 1e-5 + 1e-4 |ref| on >= 99.9 % of the pixels of every plane and at most 1 LSB in the final RGBA8 image
on those pixels."""
import ctypes as C

import numpy as np
import pytest

from blok_amd import world as W
from tests import harness_ffi as H
from tests import oracle_ffi as O
from tests.conftest import SEED

Wd, Ht = 96, 64


# ------------------------------------------------------------------------------------------------ helpers
class HarnessPost:
    """The product's per-pixel bodies compiled for the CPU, with the product's ping-pong (tests/host_harness)."""

    def __init__(self, w, h, settings=None):
        self.L, self.w, self.h = H.lib(), w, h
        self.L.hh_post_new.restype = C.c_void_p
        self.L.hh_post_denoise.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_void_p, C.c_void_p]
        self.L.hh_post_state.argtypes = [C.c_void_p] * 6
        self.L.hh_post_taa.argtypes = [C.c_void_p] * 3 + [C.c_float, C.c_float, C.c_uint32, C.c_void_p]
        self.L.hh_post_free.argtypes = [C.c_void_p]
        self.p = C.c_void_p(self.L.hh_post_new(w, h))
        self.S = settings or O.OrcDenoiseSettings.default()

    def __del__(self):
        self.L.hh_post_free(self.p)

    def denoise(self, color, world_pos, normals, prev_view_proj, frame_count, motion=None):
        out = np.zeros((self.h, self.w, 4), np.float32)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (color, world_pos, normals)]
        M = np.ascontiguousarray(prev_view_proj, dtype=np.float32)
        m = None if motion is None else np.ascontiguousarray(motion, dtype=np.float32)
        self.L.hh_post_denoise(self.p, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, None if m is None else m.ctypes.data,
                               M.ctypes.data, frame_count, C.byref(self.S), out.ctypes.data)
        return out

    def state(self):
        h, w = self.h, self.w
        out = (np.zeros((h, w, 4), np.float32), np.zeros((h, w, 2), np.float32), np.zeros((h, w), np.float32),
               np.zeros((h, w), np.float32), np.zeros((h, w, 2), np.float32))
        self.L.hh_post_state(self.p, *[a.ctypes.data for a in out])
        return out

    def taa(self, color, frame_count, fmin=0.93, fmax=0.98):
        out = np.zeros((self.h, self.w, 4), np.float32)
        c = np.ascontiguousarray(color, dtype=np.float32)
        self.L.hh_post_taa(self.p, c.ctypes.data, None, fmin, fmax, frame_count, out.ctypes.data)
        return out


def harness_sharpen(rgba8, strength=0.5):
    L = H.lib()
    L.hh_post_sharpen.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
    rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint32)
    out = np.zeros_like(rgba8)
    L.hh_post_sharpen(rgba8.ctypes.data, rgba8.shape[1], rgba8.shape[0], strength, out.ctypes.data)
    return out


def camera_path(n_frames):
    """A slow dolly + pan around the 64^3 scene: consecutive frames overlap, so history is reused but reprojected."""
    cams = []
    for k in range(n_frames):
        eye = (96.0 + 0.25 * k, 60.0 + 0.05 * k, -30.0 + 0.1 * k)
        cams.append(W.camera_look_at(eye, (32.0 + 0.05 * k, 24.0, 32.0), 60.0, Wd, Ht))
    return cams


@pytest.fixture(scope="module")
def gbuffer_frames():
    """Noisy 1-spp path-traced frames of the 64^3 scene from the oracle's raygen.rgen restatement."""
    cm = W.ChunkManager(128, 1.0)
    cm.generate_scene(64, SEED)
    cm.rebuild_dirty_chunks()
    mats = W.scene_materials(SEED)
    pw = cm.pack_chunks_to_gpu_svo(mats)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    frames = []
    cams = camera_path(5)
    for k, cam in enumerate(cams):
        planes, _ = O.render_paths(lat, mats, cam, Wd, Ht, spp=1, max_bounces=2, frame_index=k, threads=8)
        prev = W.view_proj_from_camera(cams[max(k - 1, 0)])
        frames.append((planes, prev))
    return frames, pw, mats, cams


# ------------------------------------------------------------------------------------------------ binary16
def test_binary16_rounding_matches_numpy():
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.normal(0, 1, 2000), rng.normal(0, 1e-6, 500), rng.uniform(-7e4, 7e4, 500),
                         [0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 1e-8, 2.0 ** -25, 2.0 ** -24, 3 * 2.0 ** -26, 6.1e-5, 6.2e-5,
                          0.1, 1.0009765625, 1.00048828125, 1.00146484375, 64.0, 0.95]]).astype(np.float32)
    ref = xs.astype(np.float16).astype(np.float32)
    L = H.lib()
    L.hh_q16.restype = C.c_float; L.hh_q16.argtypes = [C.c_float]
    L.hh_f2h.restype = C.c_uint16; L.hh_f2h.argtypes = [C.c_float]
    for x, r in zip(xs, ref):
        assert np.float32(O.q16(x)).tobytes() == r.tobytes(), x
        assert np.float32(L.hh_q16(x)).tobytes() == r.tobytes(), x
        assert L.hh_f2h(x) == np.float16(x).view(np.uint16), x


# ------------------------------------------------------------------------------------------------ hand-computed cases
def flat_gbuffer(color, depth=50.0, normal=(0.0, 1.0, 0.0)):
    c = np.zeros((Ht, Wd, 4), np.float32); c[..., :3] = color; c[..., 3] = 1
    wp = np.zeros((Ht, Wd, 4), np.float32)
    ys, xs = np.mgrid[0:Ht, 0:Wd]
    wp[..., 0] = xs * 0.1; wp[..., 1] = 0.0; wp[..., 2] = ys * 0.1; wp[..., 3] = depth
    nr = np.zeros((Ht, Wd, 4), np.float32); nr[..., :3] = normal; nr[..., 3] = 0.5
    return c, wp, nr


@pytest.mark.parametrize("impl", ["oracle", "product_on_cpu"])
def test_known_answers(impl):
    make = (lambda: O.OracleDenoiser(Wd, Ht)) if impl == "oracle" else (lambda: HarnessPost(Wd, Ht))
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    # frame 0 has no history: temporal output = clamp(colour, 0, 100), moments = (lum, lum^2), history length 1;
    # on a flat surface of constant colour every filter is the identity
    col = np.array([0.2, 0.5, 150.0], np.float32)
    c, wp, nr = flat_gbuffer(col)
    d = make()
    out = d.denoise(c, wp, nr, ident, 0)
    clamped = np.minimum(col, np.float32(100.0))
    # (the a-trous normalisation sum(c w) / sum(w) reproduces c to rounding only)
    assert np.allclose(out[..., :3], np.broadcast_to(clamped, (Ht, Wd, 3)), rtol=2e-6, atol=0) and (out[..., 3] == 1).all()
    lum = (np.float32(0.2126) * col[0] + np.float32(0.7152) * col[1]) + np.float32(0.0722) * col[2]
    if impl == "product_on_cpu":
        hist, mom, hl, var, mot = d.state()
    else:
        hist, mom, hl, var, mot = d.prev["color"], d.prev["moments"], d.prev["hist_len"], d.variance, d.motion
    assert np.array_equal(hist[..., :3], np.broadcast_to(clamped, (Ht, Wd, 3)))
    assert (mom[..., 0] == lum).all() and (mom[..., 1] == min(lum * lum, np.float32(10000.0))).all()
    assert (hl == 1).all()
    # variance: temporal weight 0 at history 1, spatial variance 0 on a constant plane, boost 1.5..1 -> floor 1e-4
    assert (var == np.float32(0.0001)).all()
    # sky pixels (depth > 9000) pass through the a-trous filter untouched and are never sampled
    c, wp, nr = flat_gbuffer(np.array([0.3, 0.3, 0.3], np.float32))
    rng = np.random.default_rng(0)
    c[..., :3] = rng.uniform(0, 1, (Ht, Wd, 3)).astype(np.float32)
    wp[: Ht // 2, :, 3] = 10000.0
    out = make().denoise(c, wp, nr, ident, 0)
    assert np.array_equal(out[: Ht // 2, :, :3], c[: Ht // 2, :, :3])
    assert not np.array_equal(out[Ht // 2 + 8:, :, :3], c[Ht // 2 + 8:, :, :3])       # the surface half is filtered
    assert (out[Ht // 2 + 8:, :, :3].std() < c[Ht // 2 + 8:, :, :3].std())
    # surfaces facing different ways do not mix: a vertical normal discontinuity keeps a hard colour edge
    c, wp, nr = flat_gbuffer(np.array([1.0, 0.0, 0.0], np.float32))
    c[:, Wd // 2:, :3] = (0.0, 0.0, 1.0); nr[:, Wd // 2:, :3] = (1.0, 0.0, 0.0)
    out = make().denoise(c, wp, nr, ident, 0)
    assert np.allclose(out[..., :3], c[..., :3], rtol=2e-6, atol=0) and (out[:, : Wd // 2, 2] == 0).all() and (out[:, Wd // 2:, 0] == 0).all()


def test_sharpen_and_taa_formulas():
    # sharpen: constant image is a fixed point; an isolated texel follows e + (e - blur) * 1.5
    img = np.full((Ht, Wd), 0xFF808080, np.uint32)
    assert np.array_equal(O.sharpen(img), img) and np.array_equal(harness_sharpen(img), img)
    img[10, 10] = 0xFFFFFFFF
    out = O.sharpen(img)
    e, nb = np.float32(1.0), np.float32(128 / 255)
    blur = ((nb + nb + nb + nb) * np.float32(1) + (nb + nb + nb + nb) * np.float32(2) + np.float32(4) * e) / np.float32(16)
    centre = min(max(e + (e - blur) * np.float32(1.5), 0), 1)
    assert out[10, 10] & 0xFF == int(centre * np.float32(255) + np.float32(0.5))
    blur_n = (((nb + nb) + nb + nb) + ((nb + nb) + nb + e) * np.float32(2) + np.float32(4) * nb) / np.float32(16)   # a 4-neighbour
    assert out[10, 11] & 0xFF == int(min(max(nb + (nb - blur_n) * np.float32(1.5), 0), 1) * np.float32(255) + np.float32(0.5))
    assert np.array_equal(harness_sharpen(img), out)
    # TAA with frameCount 0: feedback 0 -> output = current + 0.1 (current - neighbourhood mean), history = current
    rng = np.random.default_rng(1)
    c = np.zeros((Ht, Wd, 4), np.float32); c[..., :3] = rng.uniform(0, 2, (Ht, Wd, 3)); c[..., 3] = 0.25
    o = O.OracleDenoiser(Wd, Ht)
    out = o.taa(c, 0, motion=np.zeros((Ht, Wd, 2), np.float32))
    assert np.array_equal(o.taa_hist[..., :3], c[..., :3]) and (out[..., 3] == 0.25).all()
    pad = np.pad(c[..., :3].astype(np.float64), ((1, 1), (1, 1), (0, 0)), mode="edge")
    mean = sum(pad[1 + dy:Ht + 1 + dy, 1 + dx:Wd + 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)) / 9.0
    assert np.abs(out[..., :3] - (c[..., :3] + 0.1 * (c[..., :3] - mean))).max() < 1e-5


# ------------------------------------------------------------------------------------------------ sequences
def test_product_on_cpu_equals_oracle_over_a_sequence(gbuffer_frames):
    frames, _, _, _ = gbuffer_frames
    o, p = O.OracleDenoiser(Wd, Ht), HarnessPost(Wd, Ht)
    reused = []
    for k, (planes, prev_vp) in enumerate(frames):
        a = o.denoise(planes["color"], planes["world_pos"], planes["normal_roughness"], prev_vp, k)
        b = p.denoise(planes["color"], planes["world_pos"], planes["normal_roughness"], prev_vp, k)
        assert a.tobytes() == b.tobytes(), f"denoised frame {k}"
        hist, mom, hl, var, mot = p.state()
        assert hist.tobytes() == o.prev["color"].tobytes() and mom.tobytes() == o.prev["moments"].tobytes()
        assert hl.tobytes() == o.prev["hist_len"].tobytes() and var.tobytes() == o.variance.tobytes()
        q = np.vectorize(O.q16)(o.motion).astype(np.float32)
        assert mot.tobytes() == q.tobytes()
        ta, tb = o.taa(a, k), p.taa(b, k)
        assert ta.tobytes() == tb.tobytes(), f"taa frame {k}"
        ldr = O.tonemap(ta)
        assert np.array_equal(O.sharpen(ldr.reshape(Ht, Wd)), harness_sharpen(ldr.reshape(Ht, Wd)))
        surface = planes["world_pos"][..., 3] < 9000
        reused.append((hl[surface] > 1).mean())
        if k:
            assert (np.abs(mot[surface]).max() > 1e-3)                    # the camera moves: real reprojection
    # history is accepted on most surface pixels and grows by one per frame (the sky plane sits 10000 units away, so
    # the slightest rotation moves it by more than the 2-unit world-position tolerance: it never accumulates)
    assert reused[0] == 0 and reused[-1] > 0.5, reused
    assert hl.max() == len(frames)
    # the chain does what it is for: the denoised sequence is less noisy than its input
    noisy, surface = frames[-1][0]["color"][..., :3], frames[-1][0]["world_pos"][..., 3] < 9000
    lap = lambda im: np.abs(im[1:-1, 1:-1] - 0.25 * (im[:-2, 1:-1] + im[2:, 1:-1] + im[1:-1, :-2] + im[1:-1, 2:]))[surface[1:-1, 1:-1]].mean()
    assert lap(a[..., :3]) < 0.8 * lap(noisy)            # (at 96x64 most of the remaining variation is voxel-face detail)


def test_oracle_chain_matches_committed_digests():
    """Regression pin of the restatement (tests/golden/post_chain.json, made by tests/golden/make_golden.py)."""
    import json
    from pathlib import Path
    from tests.golden.make_golden import post_chain_sequence
    want = json.loads((Path(__file__).parent / "golden" / "post_chain.json").read_text())
    assert (want["width"], want["height"]) == (Wd, Ht)
    assert post_chain_sequence() == want["frames"]


def test_explicit_motion_plane_and_settings(gbuffer_frames):
    """A caller-provided motion plane replaces the computed one; non-default settings and 0 / 5 iterations."""
    frames, _, _, _ = gbuffer_frames
    for iters in (0, 1, 5):
        S = O.OrcDenoiseSettings.default()
        S.atrousIterations = iters; S.temporalAlpha = 0.2; S.phiColor = 1.5; S.minHistoryLength = 8; S.varianceClipGamma = 0.75
        o, p = O.OracleDenoiser(Wd, Ht, S), HarnessPost(Wd, Ht, S)
        rng = np.random.default_rng(iters)
        for k, (planes, prev_vp) in enumerate(frames[:3]):
            motion = rng.normal(0, 0.01, (Ht, Wd, 2)).astype(np.float32)
            a = o.denoise(planes["color"], planes["world_pos"], planes["normal_roughness"], prev_vp, k, motion=motion)
            b = p.denoise(planes["color"], planes["world_pos"], planes["normal_roughness"], prev_vp, k, motion=motion)
            assert a.tobytes() == b.tobytes(), (iters, k)


def test_view_proj_from_camera_inverts_the_primary_ray_mapping():
    cam = camera_path(3)[2]
    M = W.view_proj_from_camera(cam).reshape(4, 4).T.astype(np.float64)          # rows
    rays = O.primary_rays(cam, Wd, Ht)
    for (x, y) in [(0, 0), (Wd - 1, Ht - 1), (17, 40), (Wd // 2, Ht // 2)]:
        r = rays[y * Wd + x]
        p = np.append(np.asarray(r["org"], np.float64) + 37.5 * np.asarray(r["dir"], np.float64), 1.0)
        clip = M @ p
        u, v = clip[0] / clip[3] * 0.5 + 0.5, clip[1] / clip[3] * 0.5 + 0.5
        assert abs(u * Wd - (x + 0.5)) < 1e-3 and abs(v * Ht - (y + 0.5)) < 1e-3


def test_c_abi_view_proj_matches_the_python_helper():
    from blok_amd import _ffi
    L = _ffi.hip_lib()
    for cam in camera_path(4):
        m = (C.c_float * 16)()
        L.blok_camera_view_proj(_ffi.ptr(np.ascontiguousarray(cam)), m)
        assert np.allclose(np.array(m, dtype=np.float32), W.view_proj_from_camera(cam), rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_draw_frame_rt_is_the_composition_of_the_passes(gbuffer_frames):
    """blok_hip_draw_frame_rt == trace_paths -> denoise -> taa -> tonemap -> sharpen with the reference's defaults, frame
    counter and previous camera kept by the context."""
    import torch
    from blok_amd.tracer import HipTracer
    _, pw, mats, cams = gbuffer_frames
    one, many = HipTracer(Wd, Ht).init(), HipTracer(Wd, Ht).init()
    one.add_world(pw); many.add_world(pw)
    n = Wd * Ht
    P = {k: torch.zeros((n, 4), dtype=torch.float32, device="cuda") for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
    den = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); taa = torch.zeros_like(den)
    ldr = torch.zeros(n, dtype=torch.int32, device="cuda"); sharp = torch.zeros_like(ldr)
    for k, cam in enumerate(cams[:4]):
        got, frames = one.draw_frame_rt(cam, spp=2)
        assert frames == k + 1
        many.set_taa_jitter(W.taa_jitter(k))                 # frame k's projection carries jitterSequence[k mod 16] (renderer_draw.cpp:64-81)
        many.trace_paths_device(cam, P["color"].data_ptr(), spp=2, max_bounces=2, frame_index=k, world_pos_ptr=P["world_pos"].data_ptr(),
                                normal_roughness_ptr=P["normal_roughness"].data_ptr(), albedo_metallic_ptr=P["albedo_metallic"].data_ptr())
        many.denoise_device(P["color"].data_ptr(), P["world_pos"].data_ptr(), P["normal_roughness"].data_ptr(),
                            many.camera_view_proj(cams[max(k - 1, 0)]), k, den.data_ptr())
        many.taa_device(den.data_ptr(), taa.data_ptr(), k)
        many.tonemap_device(taa.data_ptr(), ldr.data_ptr())
        many.sharpen_device(ldr.data_ptr(), sharp.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(got.reshape(-1), sharp.cpu().numpy().view(np.uint32))
    one.post_reset()
    _, frames = one.draw_frame_rt(cams[0], spp=1)
    assert frames == 1
    # the jitter is what makes frame k differ from the un-jittered composition (frame 1: (-0.25, +1/6) px), and it can be turned off
    one.post_reset(); many.post_reset()
    one.set_rt_taa_jitter(False); many.set_taa_jitter(None)
    for k, cam in enumerate(cams[:2]):
        got, _ = one.draw_frame_rt(cam, spp=2)
        many.trace_paths_device(cam, P["color"].data_ptr(), spp=2, max_bounces=2, frame_index=k, world_pos_ptr=P["world_pos"].data_ptr(),
                                normal_roughness_ptr=P["normal_roughness"].data_ptr(), albedo_metallic_ptr=P["albedo_metallic"].data_ptr())
        many.denoise_device(P["color"].data_ptr(), P["world_pos"].data_ptr(), P["normal_roughness"].data_ptr(),
                            many.camera_view_proj(cams[max(k - 1, 0)]), k, den.data_ptr())
        many.taa_device(den.data_ptr(), taa.data_ptr(), k)
        many.tonemap_device(taa.data_ptr(), ldr.data_ptr())
        many.sharpen_device(ldr.data_ptr(), sharp.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(got.reshape(-1), sharp.cpu().numpy().view(np.uint32))
    one.shutdown(); many.shutdown()


@pytest.mark.gpu
def test_reference_format_gbuffer_planes(gbuffer_frames):
    """blok_hip_trace_paths_ref_device: the G-buffer in the reference's image formats (A15: RGBA16F normal + roughness, RGBA8
    albedo + metallic, RG16F motion written by the path kernel, raygen.rgen:55-59,392-413) holds exactly the oracle's float
    planes narrowed the way the image stores narrow them, the motion vectors of computeMotionVector (raygen.rgen:150-155) for a
    moved previous camera, and the denoiser fed with them gives the frame it gives for the float4 planes, bit for bit."""
    import torch
    from blok_amd.tracer import HipTracer
    _, pw, mats, cams = gbuffer_frames
    n = Wd * Ht
    tr, tf = HipTracer(Wd, Ht).init(), HipTracer(Wd, Ht).init()
    tr.add_world(pw); tf.add_world(pw)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    col = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); pos = torch.zeros_like(col)
    nr_h = torch.zeros((n, 4), dtype=torch.int16, device="cuda"); am = torch.zeros(n, dtype=torch.int32, device="cuda")
    mo_h = torch.zeros((n, 2), dtype=torch.int16, device="cuda")
    F = {k: torch.zeros((n, 4), dtype=torch.float32, device="cuda") for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
    out_r = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); out_f = torch.zeros_like(out_r)
    for k in range(3):
        cam, prev = cams[k + 1], cams[k]
        vp = tr.camera_view_proj(prev)
        tr.trace_paths_ref_device(cam, col.data_ptr(), pos.data_ptr(), nr_h.data_ptr(), am.data_ptr(), mo_h.data_ptr(), prev_view_proj=vp,
                                  spp=2, max_bounces=2, frame_index=k)
        torch.cuda.synchronize()
        want, _ = O.render_paths(lat, mats, cam, Wd, Ht, spp=2, max_bounces=2, frame_index=k, threads=8)
        assert np.array_equal(pos.cpu().numpy().reshape(Ht, Wd, 4), want["world_pos"])
        assert np.array_equal(nr_h.cpu().numpy().view(np.uint16).reshape(Ht, Wd, 4), want["normal_roughness"].astype(np.float16).view(np.uint16))
        a = np.floor(np.clip(want["albedo_metallic"], 0.0, 1.0).astype(np.float32) * np.float32(255.0) + np.float32(0.5)).astype(np.uint32)
        packed = a[..., 0] | (a[..., 1] << 8) | (a[..., 2] << 16) | (a[..., 3] << 24)
        assert np.array_equal(am.cpu().numpy().view(np.uint32).reshape(Ht, Wd), packed)
        # computeMotionVector in binary32, one rounded operation at a time
        M = np.asarray(vp, dtype=np.float32)
        p = want["world_pos"]
        x, y, z, depth = (p[..., i].astype(np.float32) for i in range(4))
        cx = ((M[0] * x + M[4] * y) + M[8] * z) + M[12]; cy = ((M[1] * x + M[5] * y) + M[9] * z) + M[13]; cw = ((M[3] * x + M[7] * y) + M[11] * z) + M[15]
        yy, xx = np.mgrid[0:Ht, 0:Wd]
        cu = (xx.astype(np.float32) + np.float32(0.5)) / np.float32(Wd); cv = (yy.astype(np.float32) + np.float32(0.5)) / np.float32(Ht)
        with np.errstate(all="ignore"):
            mu = cu - ((cx / cw) * np.float32(0.5) + np.float32(0.5)); mv = cv - ((cy / cw) * np.float32(0.5) + np.float32(0.5))
        sky = ~(depth < np.float32(9999.0))
        mu[sky] = 0; mv[sky] = 0
        got_m = mo_h.cpu().numpy().view(np.uint16).reshape(Ht, Wd, 2)
        assert np.array_equal(got_m[..., 0], mu.astype(np.float16).view(np.uint16)) and np.array_equal(got_m[..., 1], mv.astype(np.float16).view(np.uint16))
        assert (got_m[~sky] != 0).any()
        # the denoiser over these planes == the denoiser over the float4 planes
        tf.trace_paths_device(cam, F["color"].data_ptr(), spp=2, max_bounces=2, frame_index=k, world_pos_ptr=F["world_pos"].data_ptr(),
                              normal_roughness_ptr=F["normal_roughness"].data_ptr(), albedo_metallic_ptr=F["albedo_metallic"].data_ptr())
        tr.denoise_ref_device(col.data_ptr(), pos.data_ptr(), nr_h.data_ptr(), mo_h.data_ptr(), vp, k, out_r.data_ptr())
        tf.denoise_device(F["color"].data_ptr(), F["world_pos"].data_ptr(), F["normal_roughness"].data_ptr(), vp, k, out_f.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(col, F["color"]) and torch.equal(out_r, out_f), k
    tr.shutdown(); tf.shutdown()


@pytest.mark.gpu
def test_taa_jitter_on_the_gpu_matches_the_oracle():
    """A8 on the device: first-hit frames and path-traced G-buffers with the frame's Halton jitter equal the oracle's with
    the same jitter (first hits bit for bit, G-buffer planes exactly, colour within the stated tolerance), for frames 0-3 and
    13; with the beam pre-pass on (its frustum is grown by a whole pixel: jitter <= 0.5 plus the sample jitter <= 0.25)."""
    from blok_amd.tracer import HipTracer
    from tests.conftest import make_scene_world, records_equal, SEED
    cm, pw = make_scene_world(64)
    w, h = 200, 120
    tr = HipTracer(w, h).init()
    tr.add_world(pw)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    cam = W.scene_camera(64, 0, w, h, SEED)
    plain = tr.draw_frame(cam).reshape(-1)
    try:
        for frame in (0, 1, 2, 3, 13):
            j = W.taa_jitter(frame)
            tr.set_taa_jitter(j)
            O.set_jitter_clip(j, w, h)
            got = tr.draw_frame(cam).reshape(-1)
            ref, _ = lat.trace_primary(cam, w, h, threads=8)
            assert records_equal(got, ref).all(), frame
            assert not records_equal(got, plain).all(), frame
            tr.set_beam(0)
            assert records_equal(tr.draw_frame(cam).reshape(-1), ref).all(), frame
            tr.set_beam(32)
            planes = tr.trace_paths(cam, spp=4, max_bounces=2, frame_index=frame)
            want, _ = O.render_paths(lat, pw.materials, cam, w, h, spp=4, max_bounces=2, frame_index=frame, threads=8)
            for k in ("world_pos", "normal_roughness", "albedo_metallic"):
                assert np.array_equal(planes[k], want[k]), (frame, k)
            err = np.abs(planes["color"] - want["color"])
            assert (err <= 1e-4 + 1e-3 * np.abs(want["color"])).all(), (frame, float(err.max()))
        with pytest.raises(Exception):
            tr.set_taa_jitter((0.75, 0.0))
        tr.set_taa_jitter(None)
        O.set_jitter_clip(None)
        assert records_equal(tr.draw_frame(cam).reshape(-1), plain).all()
    finally:
        O.set_jitter_clip(None)
    tr.shutdown()



@pytest.mark.gpu
def test_gpu_chain_matches_oracle(gbuffer_frames):
    """Path kernel -> denoiser -> TAA -> tonemap -> sharpen on the device, against the oracle chain fed with the same
    (device-produced) G-buffer planes."""
    import torch
    from blok_amd.tracer import HipTracer
    _, pw, mats, cams = gbuffer_frames
    tr = HipTracer(Wd, Ht).init()
    tr.add_world(pw)
    o = O.OracleDenoiser(Wd, Ht)
    n = Wd * Ht
    planes = {k: torch.zeros((n, 4), dtype=torch.float32, device="cuda") for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
    den = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); taa = torch.zeros_like(den)
    ldr = torch.zeros(n, dtype=torch.int32, device="cuda"); sharp = torch.zeros_like(ldr)

    def close(got, ref, what):
        # EVERY pixel.  The device differs from the host only in expf (variance.comp's weights) by a few ulp; every stage is compared
        # on the SAME inputs — the TAA resolve gets the device's own denoised plane — because taa.comp's variance clip takes
        # sqrt(max(E[c^2] - mean^2, 0)) of the 3x3 neighbourhood (:95-107): where that neighbourhood is nearly uniform the difference
        # cancels, and a 1e-6 change of the input moves the clip box by 1e-3 (measured end to end: 0.07 % of the pixels off by up to
        # 2e-2, all of them in the TAA plane, none in the denoiser's; profiles/r03_post_chain_outliers.txt).
        ok = np.abs(got - ref) <= 1e-5 + 1e-4 * np.abs(ref)
        ok = ok.reshape(Ht, Wd, -1).all(axis=2)
        assert ok.all(), (what, int((~ok).sum()), float(np.abs(got - ref).max()))
        return ok

    for k, cam in enumerate(cams):
        tr.trace_paths_device(cam, planes["color"].data_ptr(), spp=1, max_bounces=2, frame_index=k,
                              world_pos_ptr=planes["world_pos"].data_ptr(), normal_roughness_ptr=planes["normal_roughness"].data_ptr(),
                              albedo_metallic_ptr=planes["albedo_metallic"].data_ptr())
        prev_vp = W.view_proj_from_camera(cams[max(k - 1, 0)])
        tr.denoise_device(planes["color"].data_ptr(), planes["world_pos"].data_ptr(), planes["normal_roughness"].data_ptr(), prev_vp, k, den.data_ptr())
        tr.taa_device(den.data_ptr(), taa.data_ptr(), k)
        tr.tonemap_device(taa.data_ptr(), ldr.data_ptr())
        tr.sharpen_device(ldr.data_ptr(), sharp.data_ptr())
        torch.cuda.synchronize()
        host = {name: t.cpu().numpy().reshape(Ht, Wd, 4) for name, t in planes.items()}
        ref_den = o.denoise(host["color"], host["world_pos"], host["normal_roughness"], prev_vp, k)
        ok = close(den.cpu().numpy().reshape(Ht, Wd, 4), ref_den, f"denoised {k}")
        hist, mom, hl, var, mot = tr.denoise_state()
        close(hist, o.prev["color"], f"history {k}"); close(mom, o.prev["moments"], f"moments {k}")
        assert np.array_equal(hl, o.prev["hist_len"])
        close(var[..., None], o.variance[..., None], f"variance {k}")
        assert np.array_equal(mot, np.vectorize(O.q16)(o.motion).astype(np.float32))
        got_taa = taa.cpu().numpy().reshape(Ht, Wd, 4)
        ref_taa = o.taa(den.cpu().numpy().reshape(Ht, Wd, 4), k)            # the device's denoised plane in, see close()
        ok &= close(got_taa, ref_taa, f"taa {k}")
        ref_sharp = O.sharpen(O.tonemap(got_taa).reshape(Ht, Wd))           # tonemap + sharpen of the device's resolved plane: <= 1 LSB
        got_sharp = sharp.cpu().numpy().view(np.uint32).reshape(Ht, Wd)
        d = np.abs(got_sharp.view(np.uint8).reshape(Ht, Wd, 4).astype(int) - ref_sharp.view(np.uint8).reshape(Ht, Wd, 4).astype(int))
        inner = ok.copy()                                                   # a differing neighbour reaches into the 3x3 sharpen footprint
        inner[1:-1, 1:-1] = ok[1:-1, 1:-1] & ok[:-2, 1:-1] & ok[2:, 1:-1] & ok[1:-1, :-2] & ok[1:-1, 2:] & ok[:-2, :-2] & ok[2:, 2:] & ok[:-2, 2:] & ok[2:, :-2]
        assert (d[inner] <= 1).all(), d[inner].max()
    assert (hl[host["world_pos"][..., 3] < 9000] > 1).mean() > 0.5
    tr.post_reset()
    tr.shutdown()


@pytest.mark.gpu
def test_gpu_explicit_motion_planes_and_settings():
    """Caller-provided motion planes (denoiser and TAA), non-default settings, 0 and 5 a-trous iterations, an odd frame size:
    device chain vs oracle on synthetic planes (same tolerance as above)."""
    import torch
    from blok_amd.tracer import HipTracer
    from blok_amd import _ffi
    w, h = 101, 57
    rng = np.random.default_rng(11)
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    for iters in (0, 5):
        S = O.OrcDenoiseSettings.default()
        S.atrousIterations = iters; S.temporalAlpha = 0.15; S.phiColor = 2.0; S.phiDepth = 0.3; S.minHistoryLength = 6; S.varianceBoost = 2.5
        ps = _ffi.DenoiseSettings(S.temporalAlpha, S.momentAlpha, S.varianceClipGamma, S.depthThreshold, S.normalThreshold, S.phiColor,
                                  S.phiNormal, S.phiDepth, S.atrousIterations, S.varianceBoost, S.minHistoryLength)
        tr = HipTracer(w, h).init()
        o = O.OracleDenoiser(w, h, S)
        for k in range(3):
            color = rng.uniform(0, 2, (h, w, 4)).astype(np.float32)
            wp = np.zeros((h, w, 4), np.float32)
            ys, xs = np.mgrid[0:h, 0:w]
            wp[..., 0] = xs * 0.05; wp[..., 2] = ys * 0.05; wp[..., 3] = np.where(ys < 6, 10000.0, 30.0 + 0.01 * xs)
            nr = np.zeros((h, w, 4), np.float32); nr[..., 1] = 1.0; nr[:, w // 2:, :3] = (0.6, 0.8, 0.0); nr[..., 3] = 0.37
            motion = rng.normal(0, 0.004, (h, w, 2)).astype(np.float32)
            dev = [torch.from_numpy(a).cuda() for a in (color, wp, nr, motion)]
            den = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda"); res = torch.zeros_like(den)
            tr.denoise_device(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), ident, k, den.data_ptr(), motion_ptr=dev[3].data_ptr(), settings=ps)
            tr.taa_device(den.data_ptr(), res.data_ptr(), k, 0.8, 0.95, motion_ptr=dev[3].data_ptr())
            torch.cuda.synchronize()
            ref = o.denoise(color, wp, nr, ident, k, motion=motion)
            got_den = den.cpu().numpy().reshape(h, w, 4)
            ref_res = o.taa(got_den, k, 0.8, 0.95, motion=motion)           # every stage on the same inputs (see test_gpu_chain_matches_oracle)
            for got, want, what in ((got_den, ref, "denoised"), (res.cpu().numpy().reshape(h, w, 4), ref_res, "resolved")):
                ok = (np.abs(got - want) <= 1e-5 + 1e-4 * np.abs(want)).all(axis=2)
                assert ok.all(), (iters, k, what, int((~ok).sum()), float(np.abs(got - want).max()))
            hist, mom, hl, var, mot = tr.denoise_state()
            assert np.array_equal(mot, np.vectorize(O.q16)(motion).astype(np.float32))
            assert np.array_equal(hl, o.prev["hist_len"])
        assert (hl > 1).mean() > 0.3                       # small motion: most of the surface keeps its history
        tr.shutdown()


@pytest.mark.gpu
def test_gpu_chain_at_1080p_matches_oracle():
    """The same comparison at a real frame size (1920x1080 over the 256^3 scene, two frames): every pixel of every plane within the
    stated tolerance, each stage on the same inputs."""
    import torch
    from blok_amd.tracer import HipTracer
    w, h = 1920, 1080
    cm = W.ChunkManager(128, 1.0)
    cm.generate_scene(256, SEED)
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(SEED))
    tr = HipTracer(w, h).init()
    tr.add_world(pw)
    o = O.OracleDenoiser(w, h)
    n = w * h
    P = {k: torch.zeros((n, 4), dtype=torch.float32, device="cuda") for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
    den = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); res = torch.zeros_like(den)
    base = W.scene_camera(256, 0, w, h, SEED)
    cams = [base, base.copy()]
    cams[1]["pos"][0][0] += 0.3
    for k, cam in enumerate(cams):
        tr.trace_paths_device(cam, P["color"].data_ptr(), spp=1, max_bounces=2, frame_index=k, world_pos_ptr=P["world_pos"].data_ptr(),
                              normal_roughness_ptr=P["normal_roughness"].data_ptr(), albedo_metallic_ptr=P["albedo_metallic"].data_ptr())
        prev = W.view_proj_from_camera(cams[max(k - 1, 0)])
        tr.denoise_device(P["color"].data_ptr(), P["world_pos"].data_ptr(), P["normal_roughness"].data_ptr(), prev, k, den.data_ptr())
        tr.taa_device(den.data_ptr(), res.data_ptr(), k)
        torch.cuda.synchronize()
        host = {name: t.cpu().numpy().reshape(h, w, 4) for name, t in P.items()}
        ref = o.denoise(host["color"], host["world_pos"], host["normal_roughness"], prev, k)
        got_den = den.cpu().numpy().reshape(h, w, 4)
        ref_res = o.taa(got_den, k)
        for got, want, what in ((got_den, ref, "denoised"), (res.cpu().numpy().reshape(h, w, 4), ref_res, "resolved")):
            ok = (np.abs(got - want) <= 1e-5 + 1e-4 * np.abs(want)).all(axis=2)
            assert ok.all(), (k, what, int((~ok).sum()), float(np.abs(got - want).max()))
    hl = tr.denoise_state()[2]
    assert (hl[host["world_pos"][..., 3] < 9000] > 1).mean() > 0.8
    tr.shutdown()
