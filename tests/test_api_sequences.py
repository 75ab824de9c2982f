"""Odd but legal call orders through the C ABI: nothing may crash, leak a HIP error (conftest guard) or return stale data."""
import numpy as np
import pytest

from blok_amd import world as W
from tests.conftest import SEED, make_scene_world, records_equal


@pytest.mark.gpu
def test_contexts_resizes_and_mode_switches():
    import torch
    from blok_amd.tracer import HipTracer
    from blok_amd._ffi import BlokError
    cm, pw = make_scene_world(64)
    mats = pw.materials
    a, b = HipTracer(160, 90).init(), HipTracer(96, 64).init()           # two contexts side by side
    with pytest.raises(BlokError):
        a.draw_frame(W.scene_camera(64, 0, 160, 90, SEED))                # no world yet
    with pytest.raises(BlokError):
        a.draw_frame_rt(W.scene_camera(64, 0, 160, 90, SEED))
    a.add_world(pw); b.add_world(pw)
    cam_a, cam_b = W.scene_camera(64, 0, 160, 90, SEED), W.scene_camera(64, 1, 96, 64, SEED)
    first = a.draw_frame(cam_a).copy()
    # interleave every mode on both contexts
    for k in range(3):
        pa, na = a.draw_frame_rt(cam_a, spp=2)
        pb, nb = b.draw_frame_accumulate(cam_b, 2, 2)
        assert (na, nb) == (k + 1, k + 1) and pa.shape == (90, 160) and pb.shape == (64, 96)
        assert records_equal(a.draw_frame(cam_a).reshape(-1), first.reshape(-1)).all()        # first-hit frames unaffected
    # resize in the middle of a progressive / denoised sequence: histories restart, sizes follow
    a.resize(128, 72)
    cam_a2 = W.scene_camera(64, 0, 128, 72, SEED)
    px, n = a.draw_frame_rt(cam_a2, spp=1)
    assert px.shape == (72, 128) and n == 1
    px, n = a.draw_frame_accumulate(cam_a2, 1, 1)
    assert px.shape == (72, 128) and n == 1
    assert a.draw_frame(cam_a2).shape == (72, 128)
    a.resize(160, 90)
    assert records_equal(a.draw_frame(cam_a).reshape(-1), first.reshape(-1)).all()
    # a resize that keeps the pixel count (160x90 -> 90x160) must drop the history planes too: the first denoised frame after it
    # equals the first frame of a fresh context of that shape
    a.draw_frame_rt(cam_a, spp=1); a.draw_frame_rt(cam_a, spp=1)
    a.resize(90, 160)
    cam_t = W.scene_camera(64, 0, 90, 160, SEED)
    px_resized, n = a.draw_frame_rt(cam_t, spp=1)
    fresh = HipTracer(90, 160).init(); fresh.add_world(pw)
    px_fresh, _ = fresh.draw_frame_rt(cam_t, spp=1)
    fresh.shutdown()
    assert px_resized.shape == (160, 90) and (px_resized == px_fresh).all()
    a.resize(160, 90)
    # a resident volume replaces the uploaded world; uploading a world again replaces the volume's tree
    ids = W.scene_dense(64, SEED)
    a.volume_create((0, 0, 0), (64, 64, 64), 128, 1.0)
    a.volume_upload((ids != 0).astype(np.float32), ids)
    st = a.volume_rebuild(mats)
    assert st.n_voxels == int((ids != 0).sum())
    assert records_equal(a.draw_frame(cam_a).reshape(-1), first.reshape(-1)).all()            # same voxels, same picture
    for centre, radius in (((float("nan"), 40.0, 32.0), 10.0), ((32.0, float("inf"), 32.0), 10.0), ((32.0, 40.0, 32.0), float("nan")), ((3.0e9, 40.0, 32.0), 1.0)):
        with pytest.raises(BlokError):                                                        # not finite / beyond int32: refused, no launch
            a.volume_apply_brush(centre, radius, 0.0, 1)
    a.volume_apply_brush((32.0, 40.0, 32.0), 10.0, 0.0, 1)
    a.volume_rebuild(mats)
    edited = a.draw_frame(cam_a)
    assert (~records_equal(edited.reshape(-1), first.reshape(-1))).sum() > 10
    px, n = a.draw_frame_rt(cam_a, spp=1)                                                     # post chain over the edited world
    assert n >= 1 and len(np.unique(px)) > 50
    a.add_world(pw)
    assert records_equal(a.draw_frame(cam_a).reshape(-1), first.reshape(-1)).all()
    a.volume_destroy()
    # empty world and back
    a.cleanup_world() if hasattr(a, "cleanup_world") else None
    a.add_world(pw)
    assert records_equal(a.draw_frame(cam_a).reshape(-1), first.reshape(-1)).all()
    # streams: the same frame on two torch streams at once
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h1 = torch.zeros((160 * 90, 4), dtype=torch.int32, device="cuda"); h2 = torch.zeros_like(h1)
    for _ in range(4):
        a.draw_frame_device(cam_a, h1.data_ptr(), 0, stream=s1.cuda_stream)
        a.draw_frame_device(cam_a, h2.data_ptr(), 0, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(h1, h2) and (h1.cpu().numpy().view(np.uint8).reshape(-1, 16) == first.reshape(-1).view(np.uint8).reshape(-1, 16)).all()
    a.shutdown(); b.shutdown()
