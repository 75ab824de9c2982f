"""Progressive mode of the compute backend (reference cuda_tracer.cu:372-386, 450-486): the
accumulation buffer, its camera-change reset and the ACES + gamma 2.2 RGBA8 output.

Tolerances: the accumulation buffer is a sum of path-traced frames, so it inherits test_paths' colour tolerance
(|delta| <= 1e-4 + 1e-3*|ref| per channel on >= 99.5 % of pixels); the 8-bit output may differ by 1 LSB where the
device's powf and libm's round differently, and more only on the rare pixels outside the colour tolerance."""
import numpy as np
import pytest

from blok_amd import world as W
from tests import oracle_ffi as O
from tests.conftest import SEED


def test_oracle_accumulate_known_answers():
    accum = np.zeros((1, 4, 4), dtype=np.float32)
    color = np.array([[[0, 0, 0, 1], [1, 1, 1, 1], [0.18, 0.18, 0.18, 1], [100, 0.5, 1e-3, 1]]], dtype=np.float32)
    out = O.accumulate(accum, color)
    assert (accum[..., 3] == 1).all() and np.array_equal(accum[..., :3], color[..., :3])
    aces = lambda x: np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0, 1)
    srgb = lambda x: int(np.clip(x, 0, 1) ** (1 / 2.2) * 255 + 0.5)
    px = out[0]
    assert px[0] == 0xFF000000
    assert px[1] & 0xFF == srgb(aces(1.0)) and (px[1] >> 24) == 0xFF
    assert px[2] & 0xFF == srgb(aces(0.18))
    assert px[3] & 0xFF == 255 and (px[3] >> 8) & 0xFF == srgb(aces(0.5)) and (px[3] >> 16) & 0xFF == srgb(aces(1e-3))
    out2 = O.accumulate(accum, np.zeros_like(color))          # the average halves
    assert (accum[..., 3] == 2).all()
    assert out2[0][1] & 0xFF == srgb(aces(0.5))


@pytest.mark.gpu
def test_progressive_frames_match_oracle_and_reset_on_camera_change(built):
    from blok_amd.tracer import HipTracer
    w, h, spp, bounces = 96, 64, 2, 2
    cm = W.ChunkManager(128, 1.0)
    cm.generate_scene(64, SEED)
    cm.rebuild_dirty_chunks()
    mats = W.scene_materials(SEED)
    pw = cm.pack_chunks_to_gpu_svo(mats)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    tr = HipTracer(w, h).init()
    tr.add_world(pw)
    cam = W.scene_camera(64, 0, w, h, SEED)

    def check(frames_expected, cam, accum_ref):
        ref_color, _ = O.render_paths(lat, mats, cam, w, h, spp=spp, max_bounces=bounces, frame_index=frames_expected - 1, threads=8)
        ref_px = O.accumulate(accum_ref, ref_color["color"])
        px, frames = tr.draw_frame_accumulate(cam, spp, bounces)
        assert frames == frames_expected
        got = tr.accum_download()
        assert (got[..., 3] == frames_expected).all()
        ok = (np.abs(got[..., :3] - accum_ref[..., :3]) <= frames_expected * (1e-4 + 1e-3 * np.abs(accum_ref[..., :3]))).all(axis=2)
        assert ok.all(), (ok.mean(), np.argwhere(~ok)[:8].tolist())
        d = np.abs(px.view(np.uint8).astype(int) - ref_px.view(np.uint8).astype(int)).reshape(h, w, 4)
        assert (d[ok] <= 1).all(), d[ok].max()
        assert (d[..., 3] == 0).all()

    accum_ref = np.zeros((h, w, 4), dtype=np.float32)
    for f in (1, 2, 3):
        check(f, cam, accum_ref)
    # a camera that moved by less than the reference's 1e-5 threshold keeps accumulating ...
    cam2 = cam.copy(); cam2["pos"][0][0] += np.float32(5e-6)
    _, frames = tr.draw_frame_accumulate(cam2, spp, bounces)
    assert frames == 4
    # ... one that moved further starts over (camChanged, cuda_tracer.cu:456-472,485)
    cam3 = W.scene_camera(64, 1, w, h, SEED)
    check(1, cam3, np.zeros((h, w, 4), dtype=np.float32))
    # explicit reset and resize
    tr.reset_accum()
    _, frames = tr.draw_frame_accumulate(cam3, spp, bounces)
    assert frames == 1
    tr.resize(48, 32)
    px, frames = tr.draw_frame_accumulate(W.scene_camera(64, 1, 48, 32, SEED), 1, 1)
    assert frames == 1 and px.shape == (32, 48) and tr.accum_download().shape == (32, 48, 4)
    tr.shutdown()
