"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle — bit-exact first-hit records."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from blok_amd import tiles as T
from blok_amd import world as W
from tests import oracle_ffi as O
from tests.conftest import SEED, edge_case_rays, make_scene_world, random_rays, records_equal

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def tracer_cls():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from blok_amd.tracer import HipTracer
    from blok_amd import _ffi
    assert _ffi.HIP_LIB.exists(), "libblok_hip.so must be built in-tree"
    return HipTracer


def test_config1_golden_first_hits(tracer_cls, scene64):
    """64^3, 256x256: committed golden crops + digests of the full buffers."""
    cm, pw = scene64
    meta = json.loads((GOLDEN / "first_hit_64.json").read_text())
    gold = np.load(GOLDEN / "first_hit_64.npz")
    tr = tracer_cls(256, 256).init()
    st = tr.add_world(pw)
    assert st.n_voxels == 10082 and st.levels == 3
    for pose in (0, 1, 2):
        cam = W.scene_camera(64, pose, 256, 256, SEED)
        hits = tr.draw_frame(cam)
        assert records_equal(hits[96:160, 96:160], gold[f"crop_pose{pose}"]).all()
        assert hashlib.sha256(hits.tobytes()).hexdigest() == meta[f"pose{pose}"]["sha256_full"]
    tr.shutdown()


def test_explicit_rays_edge_cases(tracer_cls, scene64):
    cm, pw = scene64
    tr = tracer_cls(64, 64).init()
    tr.add_world(pw)
    rays = np.concatenate([edge_case_rays(), random_rays(64, 20000, 31)])
    ref, ctr = O.Lattice(pw.nodes, pw.sub_chunks).trace(rays, threads=8)
    got = tr.trace_rays(rays)
    assert ctr["hits"] > 3000
    assert records_equal(got, ref).all()
    assert len(tr.trace_rays(rays[:0])) == 0
    tr.shutdown()


def test_config2_dense_256_1080p(tracer_cls, scene256):
    """BASELINE.json configs[1]: 256^3 dense grid upload, 1920x1080.  Full frame vs oracle."""
    cm, pw = scene256
    ids = W.scene_dense(256, SEED)
    tr = tracer_cls(1920, 1080).init()
    st = tr.add_dense(ids, (0, 0, 0), W.scene_materials(SEED))
    assert st.n_voxels == int((ids != 0).sum())
    cam = W.scene_camera(256, 0, 1920, 1080, SEED)
    dense_hits = tr.draw_frame(cam)
    ref, ctr = O.Lattice(pw.nodes, pw.sub_chunks).trace_primary(cam, 1920, 1080, threads=16)
    assert ctr["hits"] > 100000
    assert records_equal(dense_hits.reshape(-1), ref).all()
    # the same voxels through the reference-format upload give the same frame
    tr.add_world(pw)
    assert records_equal(tr.draw_frame(cam).reshape(-1), ref).all()
    # every hit names a filled voxel with its material; t is positive and below tmax
    h = dense_hits.reshape(-1)
    hit = h[h["hit"] == 1]
    v = hit["voxel"].astype(np.int64)
    assert (ids[v[:, 2], v[:, 1], v[:, 0]] == hit["material_id"]).all()
    assert (hit["t"] >= np.float32(0.001)).all() and (hit["t"] < 10000).all() and (hit["face"] < 6).all()
    miss = h[h["hit"] == 0]
    assert (miss["t"] == -1).all() and (miss["face"] == 0xFF).all() and (miss["material_id"] == 0).all()
    tr.shutdown()


def test_config2_dense_grid_dda_kernel(tracer_cls, scene256):
    """BASELINE.json configs[1] through the dense-grid kernel itself (dense_kernels.hip: the id grid in 8^3-cell tiles, tile
    occupancy bits in LDS, two-level DDA over the canonical plane sequence): full 1080p frames of all three poses and odd
    rectangles equal the oracle, the tree kernel and the RGBA8 shading bit for bit; a ragged grid at a negative origin with the
    camera inside, outside and axis-parallel does too; the other entry points keep working beside it."""
    import time
    cm, pw = scene256
    ids = W.scene_dense(256, SEED)
    mats = W.scene_materials(SEED)
    tr = tracer_cls(1920, 1080).init()
    tr.set_dense_dda(True)
    tr.add_dense(ids, (0, 0, 0), mats)
    tree = tracer_cls(1920, 1080).init()
    tree.add_dense(ids, (0, 0, 0), mats)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    for pose in (0, 1, 2):
        cam = W.scene_camera(256, pose, 1920, 1080, SEED)
        got = tr.draw_frame(cam)
        ref, ctr = lat.trace_primary(cam, 1920, 1080, threads=16)
        assert ctr["hits"] > 100000
        assert records_equal(got.reshape(-1), ref).all(), pose
        assert (tr.shade_rgba8(cam) == tree.shade_rgba8(cam)).all(), pose
        for rect in ((0, 0, 17, 9), (1001, 503, 333, 211), (1920 - 50, 1080 - 30, 50, 30)):
            x0, y0, w, h = rect
            assert records_equal(tr.draw_frame(cam, rect).reshape(-1), got[y0:y0 + h, x0:x0 + w].reshape(-1)).all(), (pose, rect)
    cam = W.scene_camera(256, 0, 1920, 1080, SEED)
    for name, t in (("dense-grid kernel", tr), ("tree kernel behind the pre-pass", tree)):
        t.set_timing(True)
        for _ in range(3):
            t.draw_frame(cam)
        print(f"256^3, 1920x1080, {name}: {t.last_kernel_ms():.3f} ms")
        t.set_timing(False)
    planes = tr.trace_paths(cam, spp=1, max_bounces=1, rect=(800, 500, 64, 32))          # the tree serves the other entries
    assert np.array_equal(planes["world_pos"], tree.trace_paths(cam, spp=1, max_bounces=1, rect=(800, 500, 64, 32))["world_pos"])
    tr.shutdown(); tree.shutdown()
    # ragged grid, negative origin
    rng = np.random.default_rng(12)
    g = np.where(rng.random((37, 22, 51)) < 0.03, rng.integers(1, 300, size=(37, 22, 51)), 0).astype(np.uint32)   # [z][y][x]
    origin = (-20, 5, -9)
    a, b = tracer_cls(203, 117).init(), tracer_cls(203, 117).init()
    a.set_dense_dda(True)
    a.add_dense(g, origin, mats); b.add_dense(g, origin, mats)
    cams = [W.camera_look_at((80.0, 60.0, -70.0), (5.0, 15.0, 9.0), 60.0, 203, 117), W.camera_look_at((3.3, 14.2, 7.7), (40.0, 9.0, 31.0), 110.0, 203, 117),
            W.camera_look_at((5.5, 200.0, 9.5), (5.5, 0.0, 9.5), 30.0, 203, 117), W.camera_look_at((-100.0, 16.0, 9.5), (100.0, 16.0, 9.5), 40.0, 203, 117)]
    for k, c in enumerate(cams):
        assert records_equal(a.draw_frame(c).reshape(-1), b.draw_frame(c).reshape(-1)).all(), k
    a.shutdown(); b.shutdown()


@pytest.fixture(scope="module")
def scene1024():
    return make_scene_world(1024)


def test_config3_1024_svo_4k(tracer_cls, scene1024):
    """BASELINE.json configs[2] at full size: 4K over the 1024^3 SVO.  Oracle on EVERY pixel of every pose;
    rectangle tiling, tile partition + untile, and determinism on the full frame."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    tr = tracer_cls(Wd, Ht).init()
    st = tr.add_world(pw)
    assert st.levels == 5 and st.n_sub_chunks == len(pw.sub_chunks)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    frames = {}
    for pose in (0, 1, 2):
        cam = W.scene_camera(1024, pose, Wd, Ht, SEED)
        full = tr.draw_frame(cam)
        frames[pose] = full
        ref, ctr = lat.trace_primary(cam, Wd, Ht, stride=1, threads=16)              # every pixel of the 4K frame
        assert ctr["rays"] == Wd * Ht and ctr["hits"] > 2000000 and ctr["iter_limit_hits"] == 0 and ctr["stack_limit_hits"] == 0
        assert records_equal(full.reshape(-1), ref).all(), f"pose {pose}"
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    full = frames[0]
    # idempotence / determinism
    assert records_equal(tr.draw_frame(cam).reshape(-1), full.reshape(-1)).all()
    # any rectangle of the frame equals the same pixels of the full frame (tile-able entry point)
    for rect in [(0, 0, 17, 9), (1000, 700, 333, 211), (3840 - 50, 2160 - 30, 50, 30)]:
        x0, y0, w, h = rect
        assert records_equal(tr.draw_frame(cam, rect).reshape(-1), full[y0:y0 + h, x0:x0 + w].reshape(-1)).all()
    # multi-GPU partition rehearsed on one device: n virtual ranks -> gathered buffer -> untile == full frame
    for n_ranks, tile in [(8, 32), (3, 64)]:
        per = tr.tiles_for_rank(tile, 0, n_ranks)
        gathered = torch.zeros((n_ranks * per * tile * tile, 4), dtype=torch.int32, device="cuda")
        for r in range(n_ranks):
            assert tr.tiles_for_rank(tile, r, n_ranks) == T.tiles_for_rank(Wd, Ht, tile, r, n_ranks)
            tr.draw_tiles_device(cam, tile, r, n_ranks, hits_ptr=gathered[r * per * tile * tile:].data_ptr())
        out = torch.empty((Ht * Wd, 4), dtype=torch.int32, device="cuda")
        tr.untile_device(gathered.data_ptr(), 16, tile, n_ranks, per, out.data_ptr())
        torch.cuda.synchronize()
        assert (out.cpu().numpy().view(np.uint8).reshape(-1, 16) == full.reshape(-1).view(np.uint8).reshape(-1, 16)).all()
        host = T.untile(gathered.cpu().numpy().view(O.HIT).reshape(-1), Wd, Ht, tile, n_ranks, per)
        assert records_equal(host.reshape(-1), full.reshape(-1)).all()
    # RGBA8 framebuffer: fused output == shading of the records; tiles -> gather layout -> un-permute == full frame
    rgba_full = tr.shade_rgba8(cam)
    mats = pw.materials
    flat = full.reshape(-1)
    face_k = np.array([0.8, 0.8, 1.0, 0.4, 0.6, 0.6], dtype=np.float32)
    alb = mats["albedo"][np.minimum(flat["material_id"], len(mats) - 1)] * face_k[np.minimum(flat["face"], 5)][:, None]
    q = (np.minimum(alb, np.float32(1.0)) * np.float32(255.0) + np.float32(0.5)).astype(np.uint32)
    expect = np.where(flat["hit"] == 1, 0xFF000000 | (q[:, 2] << 16) | (q[:, 1] << 8) | q[:, 0],
                      0xFF000000 | (230 << 16) | (200 << 8) | 160).astype(np.uint32)
    assert (rgba_full.reshape(-1) == expect).all()
    n_ranks, tile = 8, 32
    per = tr.tiles_for_rank(tile, 0, n_ranks)
    g = torch.zeros((n_ranks, per * tile * tile), dtype=torch.int32, device="cuda")
    for r in range(n_ranks):
        tr.draw_tiles_device(cam, tile, r, n_ranks, rgba_ptr=g[r].data_ptr())
    out = torch.empty(Ht * Wd, dtype=torch.int32, device="cuda")
    tr.untile_device(g.data_ptr(), 4, tile, n_ranks, per, out.data_ptr())
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint32) == rgba_full.reshape(-1)).all()
    # the bench's frame pipeline at N = 1
    from blok_amd.multi_gpu import FramePipeline, HipBackend
    pipe = FramePipeline(HipBackend(tr, cam), Wd, Ht)
    pipe.step(); pipe.flush()
    torch.cuda.synchronize()
    assert (pipe.frame_rgba.cpu().numpy().view(np.uint32) == rgba_full.reshape(-1)).all()
    assert records_equal(pipe.hits.cpu().numpy().view(O.HIT).reshape(-1), flat).all()
    tr.shutdown()


def test_beam_prepass_never_changes_a_record(tracer_cls, scene64, scene1024):
    """The beam pre-pass (blok_amd/csrc/hip/beam.h) only raises the walk's start parameter to a conservative bound and
    writes whole tiles as misses when their frustum meets no voxel: frames with every beam tile size — full frame, odd
    rectangles and rank tiles, camera outside / inside / grazing the world — equal the frames without it, bit for bit."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    tr = tracer_cls(Wd, Ht).init()
    tr.add_world(pw)
    cams = [W.scene_camera(1024, pose, Wd, Ht, SEED) for pose in (0, 1, 2)]
    inside = W.scene_camera(1024, 0, Wd, Ht, SEED).copy()
    inside["pos"][0] = (512.3, 300.7, 512.9)           # inside the terrain shell's bounding volume, looking along the pose
    cams.append(inside)
    for cam in cams:
        tr.set_beam(0)
        plain = tr.draw_frame(cam)
        rect = (1001, 703, 333, 211)
        plain_rect = tr.draw_frame(cam, rect)
        for beam in (8, 16, 32, 64):
            tr.set_beam(beam)
            for fused in (1, 0, 2, 3, 4, 5):            # persistent launch with queues / beam kernel, then trace kernel / joint launch / automatic / list-fed joint / list-fed walk
                tr.set_fused(fused)
                assert records_equal(tr.draw_frame(cam).reshape(-1), plain.reshape(-1)).all(), (beam, fused)
                assert records_equal(tr.draw_frame(cam, rect).reshape(-1), plain_rect.reshape(-1)).all(), (beam, fused)
    cam = cams[1]
    tr.set_beam(0)
    plain = tr.draw_frame(cam)
    for beam, tile, n_ranks, fused in [(32, 32, 8, 1), (32, 48, 3, 1), (16, 64, 2, 1), (64, 64, 5, 1), (32, 32, 8, 0), (16, 64, 2, 0), (32, 32, 8, 2), (16, 64, 3, 2), (32, 32, 8, 4), (32, 48, 3, 4), (16, 64, 3, 5), (32, 32, 8, 3)]:
        tr.set_beam(beam)
        tr.set_fused(fused)
        per = tr.tiles_for_rank(tile, 0, n_ranks)
        gathered = torch.zeros((n_ranks * per * tile * tile, 4), dtype=torch.int32, device="cuda")
        for r in range(n_ranks):
            tr.draw_tiles_device(cam, tile, r, n_ranks, hits_ptr=gathered[r * per * tile * tile:].data_ptr())
        out = torch.empty((Ht * Wd, 4), dtype=torch.int32, device="cuda")
        tr.untile_device(gathered.data_ptr(), 16, tile, n_ranks, per, out.data_ptr())
        torch.cuda.synchronize()
        assert (out.cpu().numpy().view(np.uint8).reshape(-1, 16) == plain.reshape(-1).view(np.uint8).reshape(-1, 16)).all(), (beam, tile)
    tr.shutdown()
    # small world, small odd frame, many camera positions incl. inside filled voxels and far outside
    cm, pw = scene64
    tr = tracer_cls(203, 117).init()
    tr.add_world(pw)
    rng = np.random.default_rng(7)
    for k in range(24):
        cam = W.scene_camera(64, k % 3, 203, 117, SEED).copy()
        if k >= 3:
            cam["pos"][0] = rng.uniform(-40, 104, 3).astype(np.float32)
        tr.set_beam(0)
        plain = tr.draw_frame(cam)
        for beam in (8, 32):
            tr.set_beam(beam)
            for fused in (1, 0, 2, 4, 5):
                tr.set_fused(fused)
                assert records_equal(tr.draw_frame(cam).reshape(-1), plain.reshape(-1)).all(), (k, beam, fused)
    with pytest.raises(Exception):
        tr.set_beam(12)
    tr.shutdown()


def test_beam_prepass_random_cameras_and_odd_worlds(tracer_cls):
    """Stress of the conservative pre-pass: 60 random cameras (anywhere from deep inside the volume to far outside, any
    orientation, 4..150 degree fields of view, non-square odd frames) over a world that straddles the origin with negative
    chunk coordinates, an isolated voxel far from the rest and a one-voxel-thick wall; records with beam tiles of 8 and
    32 pixels equal the records without the pre-pass, and the oracle agrees on a sample of them."""
    rng = np.random.default_rng(2024)
    cm = W.ChunkManager(128, 1.0)
    pts = rng.integers(-40, 40, size=(6000, 3)).astype(np.int32)
    wall = np.array([(x, y, 17) for x in range(-60, 60) for y in range(-30, 30)], dtype=np.int32)
    far = np.array([(-200, 90, -170), (211, -3, 140)], dtype=np.int32)
    xyz = np.concatenate([pts, wall, far])
    cm.set_voxels(xyz, (rng.integers(1, 200, size=len(xyz))).astype(np.uint32))
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(SEED))
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    w, h = 161, 93
    tr = tracer_cls(w, h).init()
    tr.add_world(pw)
    for k in range(60):
        scale = (30.0, 80.0, 400.0)[k % 3]
        eye = rng.normal(0.0, scale, 3)
        target = eye + rng.normal(0.0, 1.0, 3) if k % 5 else rng.normal(0.0, 20.0, 3)
        if np.linalg.norm(target - eye) < 1e-3:
            target = eye + (1.0, 0.0, 0.0)
        cam = W.camera_look_at(tuple(float(v) for v in eye), tuple(float(v) for v in target), float(rng.uniform(4.0, 150.0)), w, h)
        tr.set_beam(0)
        plain = tr.draw_frame(cam).reshape(-1)
        for beam in (8, 32):
            tr.set_beam(beam)
            for fused in (1, 0, 2, 4, 5):
                tr.set_fused(fused)
                assert records_equal(tr.draw_frame(cam).reshape(-1), plain).all(), (k, beam, fused)
        if k % 10 == 0:
            ref, _ = lat.trace(O.primary_rays(cam, w, h), threads=8)
            assert records_equal(plain, ref).all(), k
    tr.shutdown()


def test_one_launch_frame_queue_rearms_itself(tracer_cls, scene1024):
    """The persistent frame kernel (trace_kernels.hip: frame_kernel) leaves its work queue empty and its counters at zero:
    forty frames back to back on one stream, then three streams with frames in flight and a change of frame size in
    between, all equal the two-launch form's frame."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 1920, 1080
    tr = tracer_cls(Wd, Ht).init()
    tr.add_world(pw)
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    tr.set_fused(False)
    want = torch.from_numpy(tr.draw_frame(cam).reshape(-1).view(np.int32).reshape(-1, 4)).cuda()
    tr.set_fused(True)
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.zeros((Ht * Wd, 4), dtype=torch.int32, device="cuda") for _ in range(3)]
    for k in range(40):
        tr.draw_frame_device(cam, outs[0].data_ptr(), 0, stream=streams[0].cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], want)
    small = tr.draw_frame(cam, (64, 32, 200, 100))          # a different launch size on the default stream in between
    for k in range(30):
        outs[k % 3].zero_()
        torch.cuda.synchronize()
        for j in range(3):
            tr.draw_frame_device(cam, outs[(k + j) % 3].data_ptr(), 0, stream=streams[(k + j) % 3].cuda_stream)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, want), k
    tr.set_fused(False)
    assert records_equal(tr.draw_frame(cam, (64, 32, 200, 100)).reshape(-1), small.reshape(-1)).all()
    tr.shutdown()


def test_beam_visit_budget_exhaustion_is_conservative(tracer_cls, scene1024):
    """A beam search that runs out of its visit budget answers with the lower bound over its pending cells (beam.h), never
    "none": with the default budget (256) and budgets of 1, 2, 3, 7, 20 and 64 node visits (searches average 35, so nearly
    every tile runs out at the small ones) the 4K frames of all three poses, a rectangle and a rank's tiles are the frames of
    an unlimited search, bit for bit."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    tr = tracer_cls(Wd, Ht).init()
    tr.add_world(pw)
    for pose in (0, 1, 2):
        cam = W.scene_camera(1024, pose, Wd, Ht, SEED)
        tr.set_beam_budget(1 << 20)
        want = tr.draw_frame(cam).reshape(-1)
        rect = (777, 333, 640, 480)
        want_rect = tr.draw_frame(cam, rect).reshape(-1)
        for budget in (0, 1, 2, 3, 7, 20, 64):
            tr.set_beam_budget(budget)
            assert records_equal(tr.draw_frame(cam).reshape(-1), want).all(), (pose, budget)
            assert records_equal(tr.draw_frame(cam, rect).reshape(-1), want_rect).all(), (pose, budget)
            for fused in (1, 0, 2, 4, 5):
                tr.set_fused(fused)
                assert records_equal(tr.draw_frame(cam).reshape(-1), want).all(), (pose, budget, fused)
    cam = W.scene_camera(1024, 1, Wd, Ht, SEED)
    tr.set_beam_budget(1 << 20)
    per = tr.tiles_for_rank(32, 3, 8)
    a = torch.zeros((per * 1024, 4), dtype=torch.int32, device="cuda"); b = torch.zeros_like(a)
    tr.draw_tiles_device(cam, 32, 3, 8, hits_ptr=a.data_ptr())
    tr.set_beam_budget(2)
    tr.draw_tiles_device(cam, 32, 3, 8, hits_ptr=b.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    tr.shutdown()


def test_joint_launch_prefix_and_search_wave_fallback(tracer_cls, scene1024):
    """Joint launch (blok_hip_set_fused 2 and the automatic default 3): searches and walk waves in one grid; with an order in force
    walk waves exist only for the tiles that walked when the order was made, and a tile that is live now without one is walked by
    its search wave.  4K over 1024^3: a static camera long enough for the order to be adopted; a camera creeping by 0.05 degrees per
    frame (inside the order's 0.25-degree window, so tiles at the silhouettes change sides); a cap on the walk waves that leaves
    most of the frame to the search waves; a jump to another pose and back — every frame equals the two-launch form's, records and
    RGBA8, and no wave ever gave up waiting."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    ref = tracer_cls(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0); ref.set_tile_ordering(0)
    tr = tracer_cls(Wd, Ht).init(); tr.add_world(pw)
    hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
    want_h = torch.zeros_like(hits); want_c = torch.zeros_like(rgba)

    def same(cam, tag):
        hits.fill_(5); rgba.fill_(5)
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr())
        ref.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(hits, want_h) and torch.equal(rgba, want_c), tag

    centre = np.array([512.0, 60.0, 512.0])
    for form in (3, 2):
        tr.set_fused(form)
        for pose in (0, 2):
            cam = W.scene_camera(1024, pose, Wd, Ht, SEED)
            for k in range(24):                      # the order is sorted behind frame 1-2 and adopted a few frames later
                same(cam, (form, pose, "static", k))
            pos = np.array(cam["pos"][0], dtype=np.float64) - centre
            for k in range(1, 13):                   # creep around the world's centre: 0.05 degrees per frame
                a = np.radians(0.05 * k)
                p = centre + np.array([pos[0] * np.cos(a) - pos[2] * np.sin(a), pos[1], pos[0] * np.sin(a) + pos[2] * np.cos(a)])
                same(W.camera_look_at(tuple(p), tuple(centre), 60.0, Wd, Ht), (form, pose, "creep", k))
            for k in range(12):
                same(cam, (form, pose, "back", k))
            for limit in (20000, 1000, 1):           # most of the frame is walked by the search waves
                tr.set_joint_prefix_limit(limit)
                for k in range(3):
                    same(cam, (form, pose, "limit", limit, k))
            tr.set_joint_prefix_limit(0)
            same(W.scene_camera(1024, 1, Wd, Ht, SEED), (form, pose, "jump"))
            same(cam, (form, pose, "return"))
    assert tr.frame_queue_stalls() == 0
    # a rectangle of the frame, and frames in flight on three streams (the automatic form falls back to two launches there)
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    tr.set_fused(3)
    rect = (640, 360, 2560, 1440)
    for k in range(12):
        assert records_equal(tr.draw_frame(cam, rect).reshape(-1), ref.draw_frame(cam, rect).reshape(-1)).all(), ("rect", k)
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [(torch.zeros_like(hits), torch.zeros_like(rgba)) for _ in streams]
    ref.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr())
    for form in (3, 2):
        tr.set_fused(form)
        for k in range(30):
            tr.draw_frame_device(cam, bufs[k % 3][0].data_ptr(), bufs[k % 3][1].data_ptr(), stream=streams[k % 3].cuda_stream)
        torch.cuda.synchronize()
        for b in bufs:
            assert torch.equal(b[0], want_h) and torch.equal(b[1], want_c), ("in flight", form)
        # the automatic form never has two joint launches in flight; FORCED joint launches in flight may wait for each other's
        # searches in a circle until some waves give up and start at the ray origin (same frame, as just checked)
        if form == 3:
            assert tr.frame_queue_stalls() == 0
    tr.shutdown(); ref.shutdown()


def test_tile_ordering_is_pure_scheduling(tracer_cls, scene1024):
    """Longest-first scheduling (tile_order.hip): the walk's workgroups take their tiles in descending order of the clocks the
    tiles' waves spent in the previous frame.  Whatever the history — first frame, repeated camera, a camera that jumps between
    poses (stale costs), rectangles in between (another geometry resets the history), three streams with frames in flight
    sharing the cost buffer while sorts read it — every frame equals the frame of a context with ordering off."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    a, b = tracer_cls(Wd, Ht).init(), tracer_cls(Wd, Ht).init()
    b.set_tile_ordering(False)
    a.add_world(pw); b.add_world(pw)
    cams = [W.scene_camera(1024, p, Wd, Ht, SEED) for p in (0, 0, 0, 1, 2, 0, 1, 1)]
    want = {}
    for k, cam in enumerate(cams):
        key = cam.tobytes()
        if key not in want:
            want[key] = b.draw_frame(cam).reshape(-1)
        assert records_equal(a.draw_frame(cam).reshape(-1), want[key]).all(), k
        if k in (2, 5):
            rect = (512, 256, 2048, 1024)                       # 32768 wave tiles: ordered too, own geometry
            for _ in range(2):
                assert records_equal(a.draw_frame(cam, rect).reshape(-1), b.draw_frame(cam, rect).reshape(-1)).all(), k
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.zeros((Ht * Wd, 4), dtype=torch.int32, device="cuda") for _ in range(3)]
    ref = {p: torch.from_numpy(want[W.scene_camera(1024, p, Wd, Ht, SEED).tobytes()].view(np.int32).reshape(-1, 4)).cuda() for p in (0, 1)}
    for k in range(12):
        pose = (k // 3) % 2
        cam = W.scene_camera(1024, pose, Wd, Ht, SEED)
        for j in range(3):
            a.draw_frame_device(cam, outs[j].data_ptr(), 0, stream=streams[j].cuda_stream)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, ref[pose]), k
    a.shutdown(); b.shutdown()


def test_a_frame_launch_can_be_captured_into_a_graph_and_replayed(tracer_cls):
    """the *_device forms neither allocate nor synchronise, so a caller may capture them into a hipGraph.  A launch that is
    being captured takes the plain two-launch form with none of the per-frame bookkeeping (event queries and records, serial numbers, an
    order adopted between frames would all be frozen into the graph or illegal during capture); replays give the frame of a direct launch."""
    import torch
    n, Wd, Ht = 256, 1920, 1080
    cm = W.ChunkManager(128, 1.0); cm.generate_scene(n, SEED); cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(SEED))
    tr = tracer_cls(Wd, Ht).init(); tr.add_world(pw)
    cam = W.scene_camera(n, 0, Wd, Ht, SEED)
    hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
    want_h = torch.zeros_like(hits); want_c = torch.zeros_like(rgba)
    tr.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr()); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    for _ in range(3):                                   # the stream's buffers exist before the capture (a capture cannot allocate)
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=s.cuda_stream)
    assert tr.last_launch_kind() == 1                    # two launches
    for _ in range(4):
        hits.fill_(9); rgba.fill_(9)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(hits, want_h) and torch.equal(rgba, want_c)
    # and the context goes on as before afterwards
    tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), stream=s.cuda_stream); torch.cuda.synchronize()
    assert torch.equal(hits, want_h) and tr.last_launch_kind() in (1, 3)
    del g
    tr.shutdown()


def test_class_order_kernels_match_their_numpy_reference(tracer_cls):
    """The three launches that follow a moving camera's frames (tile_order.hip: dilate + classify + count per 16x16-tile block; scan the block
    counts class by class; scatter) against numpy: the order is the tiles by descending class of the largest cost within `radius` tiles —
    64 classes, four to the octave from 256 clocks up — within a class by block (row-major) and row-major inside the block; rank_of its
    inverse; live = the tiles of the classes above 0; the depth sums those of 1 / max(t, 1) over the beam tiles that have a start parameter.
    Grids that do and do not fill their last blocks, every radius 0..8, costs with big empty regions, all-empty and all-equal costs."""
    from scipy.ndimage import maximum_filter
    tr = tracer_cls(256, 256).init()
    rng = np.random.default_rng(11)

    def reference(cost, radius):
        ty, tx = cost.shape
        m = maximum_filter(cost.astype(np.int64), size=2 * radius + 1, mode="constant", cval=0).astype(np.uint32) if radius else cost
        q = (m.astype(np.float32).view(np.uint32) >> 21).astype(np.int64)
        cls = np.where(m == 0, 0, np.clip(np.where(q <= 540, 1, q - 539), 1, 63))
        yy, xx = np.mgrid[0:ty, 0:tx]
        blocks_x = (tx + 15) // 16
        block = (yy // 16) * blocks_x + xx // 16
        inside = (yy % 16) * 16 + xx % 16
        key = (63 - cls).astype(np.int64) * (1 << 40) + block.astype(np.int64) * 256 + inside
        order = np.argsort(key.reshape(-1), kind="stable").astype(np.uint32)
        return order, int((cls > 0).sum())

    for (tx, ty) in ((480, 270), (251, 126), (64, 64), (17, 300), (1, 1)):
        for radius in (0, 1, 2, 4, 8):
            cost = np.zeros((ty, tx), dtype=np.uint32)
            n_spots = max(1, tx * ty // 40)
            cost.reshape(-1)[rng.integers(0, tx * ty, n_spots)] = np.maximum(256, (2.0 ** rng.uniform(8, 22, n_spots)).astype(np.uint32))
            if tx > 100:
                cost[: ty // 3, :] = 0                                  # a third of the screen with nothing in it
            beam = np.where(rng.random(777) < 0.4, 3.0e38, rng.uniform(0.25, 5000.0, 777)).astype(np.float32)
            order, rank, live, sums = tr.debug_class_order(cost, radius, beam)
            want, want_live = reference(cost, radius)
            assert np.array_equal(order, want), (tx, ty, radius, int((order != want).sum()))
            assert np.array_equal(rank[order], np.arange(tx * ty, dtype=np.uint32)) and live == want_live, (tx, ty, radius)
            inv = 1.0 / np.maximum(beam[beam < 1e38].astype(np.float64), 1.0)
            assert sums[0] == len(inv) and abs(sums[1] - inv.sum()) <= 1e-4 * inv.sum() and abs(sums[2] - (inv * inv).sum()) <= 1e-4 * (inv * inv).sum(), sums
    for cost in (np.zeros((126, 251), dtype=np.uint32), np.full((126, 251), 5000, dtype=np.uint32)):
        order, rank, live, _ = tr.debug_class_order(cost, 3)
        want, want_live = reference(cost, 3)
        assert np.array_equal(order, want) and live == want_live and np.array_equal(rank[order], np.arange(cost.size, dtype=np.uint32))
    tr.shutdown()


def test_carried_order_of_a_moving_camera_is_pure_scheduling(tracer_cls, scene1024):
    """A camera in motion (blok_hip_set_moving_order): the frame's clocks, dilated, are sorted behind it, and the next launch walks in that
    order carried to its own view by a whole-tile shift of the screen — walk waves only for the order's live prefix, whatever else has
    become live walked by the search waves.  4K over 1024^3, camera orbiting the world by 0.3 to 5 degrees per frame, back and forth,
    with a jump, a stop and a rectangle in between; launch forms 3 (automatic), 2 (joint) and 0 (two launches); a capped prefix: every
    frame equals the frame of a context with all ordering off, records and RGBA8, the carried order was in use on most of them, and no
    wave gave up waiting."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    ref = tracer_cls(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0); ref.set_tile_ordering(False)
    tr = tracer_cls(Wd, Ht).init(); tr.add_world(pw)
    hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
    want_h = torch.zeros_like(hits); want_c = torch.zeros_like(rgba)
    centre = np.array([512.0, 256.0, 512.0]); start = np.array([-358.0, 870.0, -358.0]) - centre

    def orbit(deg, lift=0.0):
        r = np.radians(deg)
        p = centre + np.array([start[0] * np.cos(r) - start[2] * np.sin(r), start[1] + lift, start[0] * np.sin(r) + start[2] * np.cos(r)])
        return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), 60.0, Wd, Ht)

    uses = []

    def same(cam, tag, rect=None):
        hits.fill_(5); rgba.fill_(5); want_h.fill_(5); want_c.fill_(5)
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr(), rect=rect)
        ref.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr(), rect=rect)
        torch.cuda.synchronize()
        uses.append(tr.last_order_use())
        assert torch.equal(hits, want_h) and torch.equal(rgba, want_c), (tag, uses[-1])

    angle, lift = 5.0, 0.0
    for form in (3, 2, 0):
        tr.set_fused(form)
        uses.clear()
        for k, step in enumerate([0.3] * 4 + [1.0] * 6 + [-1.0] * 3 + [2.0] * 4 + [5.0] * 3 + [0.0] * 4 + [1.0] * 3):
            angle += step
            lift = 3.0 * k if step else lift                               # step 0: the camera really rests
            same(orbit(angle, lift=lift), (form, k, step))
        carried = [u for u in uses if u[0] == 2]
        assert len(carried) >= 15, uses                                   # in use on most frames ...
        assert any(u[1] or u[2] for u in carried), uses                    # ... with a real shift on some
        assert any(u[0] == 1 for u in uses[-8:-3]), uses                   # the stop: the view's own order takes over
        same(W.scene_camera(1024, 1, Wd, Ht, SEED), (form, "jump"))         # into the world, grazing: nothing carries over
        assert uses[-1][0] == 0, uses[-1]
        same(orbit(angle), (form, "jump back"))
        for k in range(3):                                                 # a rectangle in between: its own geometry, its own orders
            angle += 1.0
            same(orbit(angle), (form, "rect", k), rect=(512, 256, 2048, 1024))
        assert uses[-1][0] == 2, uses
        for k in range(4):                                                 # an odd rectangle (251 x 126 wave tiles: cut blocks in the counting sort), the camera rising: vertical shifts
            lift += 40.0
            same(orbit(angle, lift=lift), (form, "odd rect", k), rect=(100, 60, 2001, 1003))
        assert uses[-1][0] == 2 and any(u[2] for u in uses[-3:]), uses
    # any shift of any order is a permutation: forced shifts that wrap the whole screen, with the full prefix and with a capped one
    # (tiles without a walk wave are found through the shift taken off again), in the joint and in the two-launch form
    tr.set_fused(3)
    for k, forced in enumerate([(479, 269), (240, 135), (1, 0), (0, 1), (123456, 7777), (37, 200)]):
        tr.debug_force_order_shift(forced)
        tr.set_joint_prefix_limit(0 if k % 2 == 0 else 9000)
        tr.set_fused(3 if k % 3 else 0)
        same(orbit(angle), ("forced shift", forced))
        same(orbit(angle), ("forced shift, at rest", forced))
        same(orbit(angle), ("forced shift, own order", forced), rect=(100, 60, 2001, 1003))
    tr.debug_force_order_shift(None)
    tr.set_fused(3)
    tr.set_joint_prefix_limit(6000)                                        # most of the frame left to the search waves
    for k in range(4):
        angle += 1.0
        same(orbit(angle), ("capped prefix", k))
    tr.set_joint_prefix_limit(0)
    assert tr.frame_queue_stalls() == 0
    tr.set_moving_order(False)
    for k in range(3):
        angle += 1.0
        same(orbit(angle), ("off", k))
        assert uses[-1][0] == 0
    tr.shutdown(); ref.shutdown()


def test_list_launches_equal_the_two_launch_form(tracer_cls, scene1024):
    """List launches (blok_hip_set_fused 4 and 5; the automatic default 3 is NOT one of them — it picks the joint launch 2 for a launch that has the
    device to itself and two launches otherwise — and runs here beside them, like the plain joint form): the walk waves take their wave tiles from
    the list the frame's own searches publish, and the walk grid is sized from the previous launch's list — a hint.  4K over 1024^3: a static
    camera; a camera creeping by 0.05 degrees per frame; jumps between poses; a view of nothing but sky followed by the top-down pose
    (the hint says "empty", every walk wave strides over several entries); the plain joint form 2 — every frame equals the
    two-launch form's, records and RGBA8, and no wave ever gave up waiting.  Then a rectangle, and frames in flight on three streams
    and from two contexts (the automatic form keeps searches and walk apart there)."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    ref = tracer_cls(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0)
    tr = tracer_cls(Wd, Ht).init(); tr.add_world(pw)
    hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
    want_h = torch.zeros_like(hits); want_c = torch.zeros_like(rgba)

    def same(cam, tag):
        hits.fill_(5); rgba.fill_(5)
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr())
        ref.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(hits, want_h) and torch.equal(rgba, want_c), tag

    centre = np.array([512.0, 60.0, 512.0])
    sky = W.camera_look_at((512.0, 900.0, 512.0), (600.0, 2000.0, 700.0), 60.0, Wd, Ht)       # looks up and away: no ray hits anything
    kinds = {3: 3, 4: 4, 5: 5, 2: 3}                 # what each form is launched as when the device is otherwise idle
    for form in (3, 4, 5, 2):
        tr.set_fused(form)
        for pose in (0, 2):
            cam = W.scene_camera(1024, pose, Wd, Ht, SEED)
            for k in range(4):
                same(cam, (form, pose, "static", k))
            assert tr.last_launch_kind() == kinds[form], (form, tr.last_launch_kind())
            pos = np.array(cam["pos"][0], dtype=np.float64) - centre
            for k in range(1, 7):                    # creep around the world's centre: 0.05 degrees per frame
                a = np.radians(0.05 * k)
                p = centre + np.array([pos[0] * np.cos(a) - pos[2] * np.sin(a), pos[1], pos[0] * np.sin(a) + pos[2] * np.cos(a)])
                same(W.camera_look_at(tuple(p), tuple(centre), 60.0, Wd, Ht), (form, pose, "creep", k))
            same(W.scene_camera(1024, 1, Wd, Ht, SEED), (form, pose, "jump"))
            same(cam, (form, pose, "return"))
        for k in range(2):
            same(sky, (form, "sky", k))
            assert int((hits[:, 3] >> 24).sum().item()) == 0
        same(W.scene_camera(1024, 2, Wd, Ht, SEED), (form, "after sky"))      # the smallest walk grid over the longest list
        same(W.scene_camera(1024, 1, Wd, Ht, SEED), (form, "after sky, inside"))
    assert tr.frame_queue_stalls() == 0
    # a rectangle of the frame
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    tr.set_fused(3)
    rect = (640, 360, 2560, 1440)
    for k in range(4):
        assert records_equal(tr.draw_frame(cam, rect).reshape(-1), ref.draw_frame(cam, rect).reshape(-1)).all(), ("rect", k)
    odd = (333, 77, 1001, 515)                       # cut wave tiles and beam tiles on every side
    assert records_equal(tr.draw_frame(cam, odd).reshape(-1), ref.draw_frame(cam, odd).reshape(-1)).all()
    # frames in flight on three streams, and a second context rendering at the same time
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [(torch.zeros_like(hits), torch.zeros_like(rgba)) for _ in streams]
    other = tracer_cls(Wd, Ht).init(); other.add_world(pw)
    o_h = torch.zeros_like(hits); o_c = torch.zeros_like(rgba); o_stream = torch.cuda.Stream()
    ref.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr())
    for form in (3, 4, 2):
        tr.set_fused(form); other.set_fused(form)
        seen = set()
        for k in range(30):
            tr.draw_frame_device(cam, bufs[k % 3][0].data_ptr(), bufs[k % 3][1].data_ptr(), stream=streams[k % 3].cuda_stream)
            seen.add(tr.last_launch_kind())
            if k % 5 == 0:
                other.draw_frame_device(cam, o_h.data_ptr(), o_c.data_ptr(), stream=o_stream.cuda_stream)
        torch.cuda.synchronize()
        for b in bufs + [(o_h, o_c)]:
            assert torch.equal(b[0], want_h) and torch.equal(b[1], want_c), ("in flight", form)
        # the automatic form never has two joint launches in flight on a device; FORCED joint launches in flight may wait for each
        # other's searches until some waves give up (same frame, as just checked)
        if form == 3:
            assert 1 in seen, seen                   # two launches whenever another stream or context had a frame pending
            assert tr.frame_queue_stalls() == 0 and other.frame_queue_stalls() == 0
    tr.shutdown(); ref.shutdown(); other.shutdown()


def test_alternating_views_each_get_their_own_order(tracer_cls, scene1024):
    """Round 4: orders are cached per VIEW (four slots per launch geometry).  A caller that alternates between fixed views in one rectangle —
    stereo eyes, a camera cycle — used to find the order of the other view in the one buffer and walk in row-major order for ever; now each
    view's clocks are sorted into its own slot after its second appearance and every later frame of that view walks in it.  Two views, then
    a cycle of three, then of five (more views than slots: the least recently walked one is evicted), on one stream and on three: every
    frame equals the frame of a context with all ordering off."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    ref = tracer_cls(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0); ref.set_tile_ordering(False)
    tr = tracer_cls(Wd, Ht).init(); tr.add_world(pw)
    hits = torch.zeros((Wd * Ht, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros(Wd * Ht, dtype=torch.int32, device="cuda")
    centre = np.array([512.0, 256.0, 512.0]); start = np.array([-358.0, 870.0, -358.0]) - centre

    def orbit(deg):
        r = np.radians(deg)
        p = centre + np.array([start[0] * np.cos(r) - start[2] * np.sin(r), start[1], start[0] * np.sin(r) + start[2] * np.cos(r)])
        return W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), 60.0, Wd, Ht)

    views = [W.scene_camera(1024, 0, Wd, Ht, SEED), orbit(3.0), W.scene_camera(1024, 2, Wd, Ht, SEED), orbit(40.0), W.scene_camera(1024, 1, Wd, Ht, SEED)]
    want = []
    for cam in views:
        h = torch.zeros_like(hits); c = torch.zeros_like(rgba)
        ref.draw_frame_device(cam, h.data_ptr(), c.data_ptr()); torch.cuda.synchronize()
        want.append((h, c))
    for n_views in (2, 3, 5):
        uses = []
        for k in range(8 * n_views):
            v = k % n_views
            hits.fill_(5); rgba.fill_(5)
            tr.draw_frame_device(views[v], hits.data_ptr(), rgba.data_ptr()); torch.cuda.synchronize()
            uses.append(tr.last_order_use()[0])
            assert torch.equal(hits, want[v][0]) and torch.equal(rgba, want[v][1]), (n_views, k)
        if n_views <= 3:                                 # every view found its own order: the last two rounds walked in them
            assert all(u == 1 for u in uses[-2 * n_views:]), (n_views, uses)
    # a zoom: the lens changes from frame to frame (0.25 degrees of field of view, then 3): the previous frame's order is carried across it
    # (the stretch about the screen's centre is residual to the shift, launch_policy.h) while the zoom is slow, dropped when it is not
    want_h = torch.zeros_like(hits); want_c = torch.zeros_like(rgba)
    uses = []
    fov = 60.0
    for k, step in enumerate([0.25] * 10 + [3.0] * 3 + [-0.25] * 6):
        fov -= step
        cam = W.camera_look_at((-358.0, 870.0, -358.0), (512.0, 256.0, 512.0), fov, Wd, Ht)
        hits.fill_(5); rgba.fill_(5)
        tr.draw_frame_device(cam, hits.data_ptr(), rgba.data_ptr())
        ref.draw_frame_device(cam, want_h.data_ptr(), want_c.data_ptr()); torch.cuda.synchronize()
        uses.append(tr.last_order_use()[0])
        assert torch.equal(hits, want_h) and torch.equal(rgba, want_c), ("zoom", k, fov)
    # (the first 3-degree step leaves more over than twice the dilation the slow zoom was sorted with; the sorts that follow dilate by more)
    assert sum(1 for u in uses[2:10] if u == 2) >= 5 and uses[10] == 0, uses
    # three streams in flight, two views alternating
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [(torch.zeros_like(hits), torch.zeros_like(rgba)) for _ in streams]
    torch.cuda.synchronize()                             # (torch fills on its own stream; the side streams do not wait for it)
    for k in range(24):
        j, v = k % 3, (k // 2) % 2
        if k >= 3 and j == 0:
            torch.cuda.synchronize()
            for jj in range(3):
                vv = ((k - 3 + jj) // 2) % 2
                assert torch.equal(outs[jj][0], want[vv][0]) and torch.equal(outs[jj][1], want[vv][1]), (k, jj)
        tr.draw_frame_device(views[v], outs[j][0].data_ptr(), outs[j][1].data_ptr(), stream=streams[j].cuda_stream)
    torch.cuda.synchronize()
    assert tr.frame_queue_stalls() == 0
    tr.shutdown(); ref.shutdown()


def test_alternating_rectangles_with_three_streams_in_flight(tracer_cls, scene1024):
    """a change of launch geometry used to leave the held markers of the LAST ADOPTION in place, which cover the readers of the
    other order buffer — the first sort of the new rectangle could rewrite the buffer frames in flight on other streams were still
    walking in.  Now the markers are held at the change too.  Two rectangles alternate every few frames, three streams in flight, the
    sort due after every frame (interval 1): every frame equals the frame of a context with ordering off; then a stream is released
    (blok_hip_release_stream) and the context goes on."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    a, b = tracer_cls(Wd, Ht).init(), tracer_cls(Wd, Ht).init()
    a.add_world(pw); b.add_world(pw)
    a.set_tile_ordering(1); b.set_tile_ordering(False)
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    rects = [(0, 0, 3840, 2160), (256, 128, 3072, 1728)]
    want = []
    for r in rects:
        h = torch.zeros((r[2] * r[3], 4), dtype=torch.int32, device="cuda")
        b.draw_frame_device(cam, h.data_ptr(), 0, rect=r); torch.cuda.synchronize()
        want.append(h)
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [[torch.zeros_like(w) for w in want] for _ in streams]
    torch.cuda.synchronize()                             # (torch fills on its own stream; the side streams do not wait for it)
    issued = []
    for k in range(36):
        which = (k // 4) % 2                             # four frames of one rectangle, then four of the other, no synchronisation in between
        j = k % 3
        if k >= 3 and k % 3 == 0:                        # the three frames issued a round ago have to be checked before their buffers are reused
            for s_ in streams: s_.synchronize()
            for jj, ww in issued[-3:]:
                assert torch.equal(outs[jj][ww], want[ww]), (k, jj, ww)
        with torch.cuda.stream(streams[j]):                # (torch's side streams do not wait for its default stream: the fill goes where the frame goes)
            outs[j][which].fill_(7)
        a.draw_frame_device(cam, outs[j][which].data_ptr(), 0, rect=rects[which], stream=streams[j].cuda_stream)
        issued.append((j, which))
    torch.cuda.synchronize()
    for jj, ww in issued[-3:]:
        assert torch.equal(outs[jj][ww], want[ww])
    # a caller that destroys a stream tells the context first; the others go on, and so does a new stream
    a.release_stream(streams[2].cuda_stream)
    streams[2] = torch.cuda.Stream()
    torch.cuda.synchronize()
    for k in range(6):
        a.draw_frame_device(cam, outs[k % 3][0].data_ptr(), 0, rect=rects[0], stream=streams[k % 3].cuda_stream)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o[0], want[0])
    a.shutdown(); b.shutdown()


def test_list_launches_on_rectangles_smaller_than_their_segments(tracer_cls, scene1024):
    """list forms 4 and 5 over fewer than 8 beam tiles — the segments beyond the last search have no search to finalise them;
    their walk workgroups used to spin out the whole poll budget (~1 ms) and trip the stall counter.  Rectangles of 1, 2, 3 and 7 beam
    tiles (and a cut one), forms 4 and 5: frames equal the two-launch form's, no wave ever gave up, and a launch is not a millisecond."""
    import torch, time
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    ref = tracer_cls(Wd, Ht).init(); ref.add_world(pw); ref.set_fused(0)
    tr = tracer_cls(Wd, Ht).init(); tr.add_world(pw)
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    for form in (4, 5):
        tr.set_fused(form)
        for rect in ((1800, 1100, 32, 32), (1800, 1100, 64, 32), (1800, 1100, 96, 32), (1700, 1100, 224, 32), (1811, 1103, 50, 40)):
            for k in range(3):
                got = tr.draw_frame(cam, rect).reshape(-1)
                assert records_equal(got, ref.draw_frame(cam, rect).reshape(-1)).all(), (form, rect, k)
            assert got["hit"].any(), rect
        assert tr.frame_queue_stalls() == 0, form
    # and it does not take a poll budget: 20 launches of the smallest rectangle
    hits = torch.zeros((32 * 32, 4), dtype=torch.int32, device="cuda")
    tr.set_fused(4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        tr.draw_frame_device(cam, hits.data_ptr(), 0, rect=(1800, 1100, 32, 32))
    torch.cuda.synchronize()
    per_launch = (time.perf_counter() - t0) / 20
    assert per_launch < 0.5e-3, per_launch
    tr.shutdown(); ref.shutdown()



def test_degenerate_cameras_are_refused(tracer_cls, scene64):
    """A camera with a non-finite component or a zero field of view is an argument error (every ray would be NaN and the
    beam pre-pass could cull nothing), not a launch."""
    from blok_amd._ffi import BlokError
    cm, pw = scene64
    tr = tracer_cls(64, 48).init()
    tr.add_world(pw)
    good = W.scene_camera(64, 0, 64, 48, SEED)
    assert tr.draw_frame(good)["hit"].any()
    for field, value in (("tan_half_fov", 0.0), ("tan_half_fov", float("nan")), ("aspect", -1.0), ("pos", float("inf")), ("fwd", float("nan"))):
        bad = good.copy()
        if field in ("pos", "fwd"):
            bad[field][0][1] = value
        else:
            bad[field] = value
        for call in (lambda: tr.draw_frame(bad), lambda: tr.trace_paths(bad, spp=1), lambda: tr.draw_frame_accumulate(bad, 1, 1)):
            with pytest.raises(BlokError) as e:
                call()
            assert e.value.status == -1
    tr.shutdown()


def test_world_edge_cases(tracer_cls):
    """Empty world, single voxel, negative coordinates, material ids above 65535, replace-world, errors."""
    from blok_amd._ffi import BlokError
    tr = tracer_cls(32, 32).init()
    cam = W.camera_look_at((5.5, 20, 7.5), (5.5, 0, 7.5), 60.0, 32, 32)
    with pytest.raises(BlokError) as e:
        tr.draw_frame(cam)
    assert e.value.status == -4                                    # BLOK_ERR_NO_WORLD
    empty = W.ChunkManager(128, 1.0).pack_chunks_to_gpu_svo()
    tr.add_world(empty)
    assert (tr.draw_frame(cam)["hit"] == 0).all()
    cm = W.ChunkManager(128, 1.0)
    rng = np.random.default_rng(4)
    xyz = rng.integers(-150, 150, size=(6000, 3)).astype(np.int32)
    cm.set_voxels(xyz, rng.integers(1, 70000, size=len(xyz)).astype(np.uint32))
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo()
    tr.add_world(pw)
    rays = random_rays(300, 5000, 12)
    rays["org"] -= 150
    ref, ctr = O.Lattice(pw.nodes, pw.sub_chunks).trace(rays, threads=4)
    got = tr.trace_rays(rays)
    assert ctr["hits"] > 200 and records_equal(got, ref).all()
    assert got["material_id"].max() > 65535
    bad = pw.sub_chunks.copy()
    bad["world_min"][0, 1] += 0.25
    with pytest.raises(BlokError) as e:
        tr.add_world(W.PackedWorld(pw.nodes, bad, pw.materials))
    assert e.value.status == -5 and "lattice" in str(e.value)     # BLOK_ERR_UNSUPPORTED
    with pytest.raises(BlokError):
        tr.draw_frame(cam, (0, 0, 64, 64))                          # rectangle outside the frame
    # x0 + w wraps in 32 bits: still INVALID_ARG (-1), and before anything is sized by w * h (straight through the C ABI: the
    # Python mirror would try to allocate the output first)
    from blok_amd import _ffi
    camc = np.ascontiguousarray(cam, dtype=_ffi.CAMERA)
    scratch = np.zeros(64 * 64, dtype=_ffi.HIT)
    for rect in ((0xFFFFFFF0, 0, 32, 32), (0, 0xFFFFFFF0, 32, 32), (16, 16, 0xFFFFFFF8, 8), (0, 0, 0, 8), (32, 0, 1, 1)):
        assert tr._lib.blok_hip_trace_primary(tr._ctx, _ffi.ptr(camc), *rect, _ffi.ptr(scratch)) == -1, rect
        assert tr._lib.blok_hip_shade_rgba8(tr._ctx, _ffi.ptr(camc), *rect, _ffi.ptr(scratch)) == -1, rect
        assert tr._lib.blok_hip_trace_primary_device(tr._ctx, _ffi.ptr(camc), *rect, _ffi.ptr(scratch), None, None) == -1, rect
    tr.resize(64, 48)
    assert tr.draw_frame(W.camera_look_at((0, 200, 0), (0, 0, 0), 60.0, 64, 48)).shape == (48, 64)
    tr.shutdown()


def test_shaded_debug_view_and_timing(tracer_cls, scene64):
    cm, pw = scene64
    tr = tracer_cls(128, 128).init()
    tr.add_world(pw)
    cam = W.scene_camera(64, 0, 128, 128, SEED)
    hits = tr.draw_frame(cam)
    rgba = tr.shade_rgba8(cam)
    assert rgba.shape == (128, 128)
    assert ((rgba >> 24) == 0xFF).all()
    assert (rgba[hits["hit"] == 0] == rgba[hits["hit"] == 0][0]).all()
    tr.set_timing(True)
    tr.draw_frame(cam)
    assert 0.0 < tr.last_kernel_ms() < 1000.0
    tr.shutdown()


def test_device_build_equals_host_build(tracer_cls, scene64, scene256, scene1024):
    """blok_hip_upload_world builds the 64-tree on the device (gpu_build.hip); the general host builder
    (tree_build.cpp) must produce the same structure and the same frames."""
    import time
    for n, (cm, pw) in ((64, scene64), (256, scene256), (1024, scene1024)):
        tr = tracer_cls(640, 360).init()
        t0 = time.perf_counter(); st_dev = tr.add_world(pw); t_dev = time.perf_counter() - t0
        assert tr.built_on_device()
        nodes_dev, mats_dev = tr.download_tree()
        cam = W.scene_camera(n, 0, 640, 360, SEED)
        frame_dev = tr.draw_frame(cam)
        tr.set_host_build(True)
        t0 = time.perf_counter(); st_host = tr.add_world(pw); t_host = time.perf_counter() - t0
        assert not tr.built_on_device()
        nodes_host, mats_host = tr.download_tree()
        frame_host = tr.draw_frame(cam)
        assert (st_dev.n_voxels, st_dev.levels, list(st_dev.origin)) == (st_host.n_voxels, st_host.levels, list(st_host.origin))
        assert np.array_equal(nodes_dev, nodes_host) and np.array_equal(mats_dev, mats_host)
        assert records_equal(frame_dev.reshape(-1), frame_host.reshape(-1)).all()
        print(f"upload {n}^3: device build {t_dev * 1e3:.1f} ms, host build {t_host * 1e3:.1f} ms")
        tr.shutdown()


def test_device_build_odd_worlds(tracer_cls):
    """Negative coordinates / several chunks (tree origins may differ between the builders: compare traced rays),
    unsupported worlds are rejected by the device path with the same messages, empty worlds take the host path."""
    from blok_amd._ffi import BlokError
    rng = np.random.default_rng(8)
    cm = W.ChunkManager(128, 1.0)
    xyz = rng.integers(-200, 180, size=(9000, 3)).astype(np.int32)
    cm.set_voxels(xyz, rng.integers(1, 500, size=len(xyz)).astype(np.uint32))
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo()
    rays = random_rays(380, 8000, 3)
    rays["org"] -= 200
    ref, ctr = O.Lattice(pw.nodes, pw.sub_chunks).trace(rays, threads=8)
    tr = tracer_cls(64, 64).init()
    st = tr.add_world(pw)
    assert tr.built_on_device() and st.n_voxels == len(np.unique(xyz, axis=0)) and ctr["hits"] > 300
    assert records_equal(tr.trace_rays(rays), ref).all()
    tr.set_host_build(True)
    tr.add_world(pw)
    assert records_equal(tr.trace_rays(rays), ref).all()
    tr.set_host_build(False)
    bad = pw.nodes.copy()
    root = int(pw.sub_chunks[0]["node_offset"] + pw.sub_chunks[0]["root_node_index"])
    bad[root]["child_mask"] = 0
    bad[root]["occupancy"] = 1.0
    with pytest.raises(BlokError, match="leaf above voxel level"):
        tr.add_world(W.PackedWorld(bad, pw.sub_chunks, pw.materials))
    tr.add_world(W.ChunkManager(128, 1.0).pack_chunks_to_gpu_svo())
    assert not tr.built_on_device()
    small = W.ChunkManager(8, 1.0)                      # 8^3 chunks -> sub-chunks of one voxel: general host path
    small.set_voxels(np.array([[1, 2, 3], [9, 2, 3]], dtype=np.int32), np.array([4, 5], dtype=np.uint32))
    small.rebuild_dirty_chunks()
    spw = small.pack_chunks_to_gpu_svo()
    assert tr.add_world(spw).n_voxels == 2 and not tr.built_on_device()
    r = np.zeros(1, dtype=O.RAY); r["org"] = (1.5, 10, 3.5); r["dir"] = (0, -1, 0); r["tmin"] = 0.001; r["tmax"] = 1e4
    got = tr.trace_rays(r)
    assert got[0]["hit"] == 1 and got[0]["material_id"] == 4 and tuple(got[0]["voxel"]) == (1, 2, 3)
    tr.shutdown()


def test_config4_2048_svo_4k_tiles(tracer_cls):
    """BASELINE.json configs[3] geometry: 2048^3 SVO (6 tree levels, 633 MB of reference nodes) at 4K, the
    8-GPU tile partition rehearsed with virtual ranks on one device, oracle on every pixel; wide-angle and grazing cameras."""
    import torch
    cm, pw = make_scene_world(2048)
    Wd, Ht = 3840, 2160
    tr = tracer_cls(Wd, Ht).init()
    st = tr.add_world(pw)
    assert st.levels == 6 and tr.built_on_device() and st.n_ref_nodes == len(pw.nodes)
    cam = W.scene_camera(2048, 0, Wd, Ht, SEED)
    full = tr.draw_frame(cam)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    ref, ctr = lat.trace_primary(cam, Wd, Ht, stride=1, threads=16)                   # every pixel of the 4K frame
    assert ctr["rays"] == Wd * Ht and ctr["hits"] > 1500000 and ctr["iter_limit_hits"] == 0
    assert records_equal(full.reshape(-1), ref).all()
    # the pre-pass at its limits in the big world: a 150-degree field of view from inside the volume, and a camera grazing
    # the terrain from a corner (long rays through half-empty cells), each against the oracle on every 2nd pixel
    for cam_x in (W.camera_look_at((1024.3, 700.2, 1024.9), (1500.0, 300.0, 1700.0), 150.0, Wd, Ht),
                  W.camera_look_at((-40.0, 1040.0, -35.0), (2048.0, 980.0, 2048.0), 60.0, Wd, Ht)):
        got = tr.draw_frame(cam_x)
        refx, cx = lat.trace_primary(cam_x, Wd, Ht, stride=2, threads=16)
        assert cx["hits"] > 100000
        assert records_equal(got[::2, ::2].reshape(-1), refx).all()
        tr.set_beam(0)
        assert records_equal(tr.draw_frame(cam_x).reshape(-1), got.reshape(-1)).all()
        tr.set_beam(32)
        # every launch form, a few frames each (the list forms size their second launch from the first one's list)
        for form in (0, 2, 3, 4, 5):
            tr.set_fused(form)
            for k in range(3):
                assert records_equal(tr.draw_frame(cam_x).reshape(-1), got.reshape(-1)).all(), (form, k)
    assert tr.frame_queue_stalls() == 0
    n_ranks, tile = 8, 32
    per = tr.tiles_for_rank(tile, 0, n_ranks)
    gathered = torch.zeros((n_ranks * per * tile * tile, 4), dtype=torch.int32, device="cuda")
    for r in range(n_ranks):
        tr.draw_tiles_device(cam, tile, r, n_ranks, hits_ptr=gathered[r * per * tile * tile:].data_ptr())
    out = torch.empty((Ht * Wd, 4), dtype=torch.int32, device="cuda")
    tr.untile_device(gathered.data_ptr(), 16, tile, n_ranks, per, out.data_ptr())
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint8).reshape(-1, 16) == full.reshape(-1).view(np.uint8).reshape(-1, 16)).all()
    tr.shutdown()


def test_sparse_tile_exchange_kernels(tracer_cls, scene1024):
    """blok_hip_compact_tiles_device / blok_hip_scatter_tiles_device (the sparse framebuffer exchange of the tile partition):
    every virtual rank's RGBA8 tiles compacted on the device, the prefixes laid side by side as a gather would, scattered
    over a sky-filled frame == the full frame's RGBA8; the record counts are the ranks' non-sky tile counts (far below the
    tile count for this 73 %-sky pose) and the records equal the numpy reference's."""
    import torch
    from blok_amd import tiles as T
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    tr = tracer_cls(Wd, Ht).init()
    tr.add_world(pw)
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    want = tr.shade_rgba8(cam)
    for n_ranks, tile in ((8, 32), (3, 64)):
        per = tr.tiles_for_rank(tile, 0, n_ranks)
        words = T.compact_words(tile, per)
        gathered = torch.zeros((n_ranks, words), dtype=torch.int32, device="cuda")
        counts = []
        for r in range(n_ranks):
            mine = tr.tiles_for_rank(tile, r, n_ranks)
            dense = torch.zeros(per * tile * tile, dtype=torch.int32, device="cuda")
            tr.draw_tiles_device(cam, tile, r, n_ranks, rgba_ptr=dense.data_ptr())
            tr.compact_tiles_device(dense.data_ptr(), tile, mine, gathered[r].data_ptr())
            torch.cuda.synchronize()
            got = gathered[r].cpu().numpy().view(np.uint32)
            ref = T.compact_tiles(dense.cpu().numpy(), tile, mine)
            counts.append(int(got[0]))
            assert got[0] == ref[0]
            px = tile * tile
            rec_got = got[1:1 + counts[-1] * (1 + px)].reshape(-1, 1 + px)
            rec_ref = ref[1:1 + counts[-1] * (1 + px)].reshape(-1, 1 + px)
            assert (rec_got[np.argsort(rec_got[:, 0])] == rec_ref).all()        # same records, the kernel's order is arbitrary
        most = max(counts)
        assert 0 < most < 0.5 * per
        # what a gather of the first `most` records of every rank delivers: the rest of each row is never transferred
        prefix = 1 + most * (1 + tile * tile)
        gathered[:, prefix:] = -1
        frame = torch.zeros(Ht * Wd, dtype=torch.int32, device="cuda")
        tr.scatter_tiles_device(gathered.data_ptr(), n_ranks, words, tile, most, frame.data_ptr())
        torch.cuda.synchronize()
        assert (frame.cpu().numpy().view(np.uint32).reshape(Ht, Wd) == want).all(), (n_ranks, tile)
    tr.shutdown()


def test_rank_tile_launches_in_their_own_order_equal_the_plain_ones(tracer_cls, scene1024):
    """Round 4 (extends the several-frames test to the ORDERED form): a rank's tile launches — one frame or several per launch — walk their
    wave tiles longest first and, in the automatic form, over the order's live prefix, once the view rests.  4K over 1024^3, ranks 1 of 2, 3 of
    4 and 0 and 7 of 8, 1 / 4 / 8 frames per launch: launch after launch of the same view (the order is measured, sorted for, adopted and in use
    by the end), then a jump to another view and back, then launches whose frames differ (no order applies): every launch equals the launch of
    a context with the ordering off, hit records and RGBA8, padding tiles and the half row of tiles below the frame included.  Then three
    streams in flight."""
    import torch
    cm, pw = scene1024
    Wd, Ht = 3840, 2160
    ref = tracer_cls(Wd, Ht).init(); ref.add_world(pw); ref.set_tile_ordering(False)
    tr = tracer_cls(Wd, Ht).init(); tr.add_world(pw)
    camA, camB, camC = (W.scene_camera(1024, pose, Wd, Ht, SEED) for pose in (0, 2, 1))
    tile = 32
    for n_ranks, rank, n_frames in ((2, 1, 1), (4, 3, 4), (8, 0, 8), (8, 7, 1), (7, 6, 2)):
        per = tr.tiles_for_rank(tile, 0, n_ranks)          # the most any rank has: ranks beyond the last tile hold a padding tile
        px = tile * tile
        hits = torch.zeros((n_frames, per * px, 4), dtype=torch.int32, device="cuda"); rgba = torch.zeros((n_frames, per * px), dtype=torch.int32, device="cuda")
        want_h = torch.zeros_like(hits); want_c = torch.zeros_like(rgba)

        def same(cams, tag):
            hits.fill_(3); rgba.fill_(3); want_h.fill_(3); want_c.fill_(3)
            batch = np.concatenate(cams)
            if n_frames == 1:
                tr.draw_tiles_device(cams[0], tile, rank, n_ranks, hits_ptr=hits.data_ptr(), rgba_ptr=rgba.data_ptr())
                ref.draw_tiles_device(cams[0], tile, rank, n_ranks, hits_ptr=want_h.data_ptr(), rgba_ptr=want_c.data_ptr())
            else:
                tr.draw_tile_frames_device(batch, tile, rank, n_ranks, per, hits_ptr=hits.data_ptr(), rgba_ptr=rgba.data_ptr())
                ref.draw_tile_frames_device(batch, tile, rank, n_ranks, per, hits_ptr=want_h.data_ptr(), rgba_ptr=want_c.data_ptr())
            torch.cuda.synchronize()
            use = tr.last_order_use()[0]
            assert torch.equal(hits, want_h) and torch.equal(rgba, want_c), (n_ranks, rank, n_frames, tag, use)
            return use

        uses = [same([camA] * n_frames, ("rest", k)) for k in range(14)]
        assert all(u == 1 for u in uses[-4:]), (n_ranks, rank, n_frames, uses)      # the view's own order is in use
        assert same([camB] * n_frames, "jump") == 0
        assert same([camA] * n_frames, "back") == 1                                # still cached
        if n_frames > 1:
            mixed = [camA, camB, camC, camA][:n_frames] + [camB] * max(0, n_frames - 4)
            for k in range(3):
                assert same(mixed, ("mixed", k)) == 0
        for k in range(3):
            same([camA] * n_frames, ("after", k))
    # three launches in flight on three streams (4 frames per launch, rank 2 of 4)
    n_ranks, rank, n_frames = 4, 2, 4
    per = tr.tiles_for_rank(tile, 0, n_ranks); px = tile * tile
    want_h = torch.zeros((n_frames, per * px, 4), dtype=torch.int32, device="cuda")
    ref.draw_tile_frames_device(np.concatenate([camA] * n_frames), tile, rank, n_ranks, per, hits_ptr=want_h.data_ptr()); torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.zeros_like(want_h) for _ in streams]
    torch.cuda.synchronize()                             # (torch fills on its own stream; the side streams do not wait for it)
    for k in range(18):
        tr.draw_tile_frames_device(np.concatenate([camA] * n_frames), tile, rank, n_ranks, per, hits_ptr=outs[k % 3].data_ptr(), stream=streams[k % 3].cuda_stream)
        if k % 3 == 2:
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, want_h), k
    assert tr.frame_queue_stalls() == 0
    tr.shutdown(); ref.shutdown()


def test_several_frames_per_launch_equal_frame_by_frame(tracer_cls, scene1024):
    """blok_hip_trace_tile_frames_device and the *_frames_device forms of compact / scatter / un-permute (one launch for up to 8
    frames of a rank's tile share, each frame with its own camera): every frame bit-identical to what the one-frame entries
    give for its camera — hit records and RGBA8, for rank counts that do and do not divide the tiles, with and without the
    pre-pass — and the assembled frames equal the single-GPU frames."""
    import torch
    from blok_amd import tiles as T
    cm, pw = scene1024
    Wd, Ht = 1920, 1080
    tr = tracer_cls(Wd, Ht).init()
    tr.add_world(pw)
    cams = [W.scene_camera(1024, pose, Wd, Ht, SEED) for pose in (0, 1, 2)]
    centre = 512.0
    cams += [W.camera_look_at((centre + 900.0 * np.cos(a), 300.0 + 40.0 * k, centre + 900.0 * np.sin(a)), (centre, 60.0, centre), 60.0, Wd, Ht)
             for k, a in enumerate(np.linspace(0.3, 5.1, 5))]
    assert len(cams) == 8
    want = [tr.shade_rgba8(c).reshape(-1) for c in cams]
    for beam in (32, 0):
        tr.set_beam(beam)
        for n_ranks, tile, n_frames in ((8, 32, 8), (3, 64, 5), (2, 32, 2)):
            per = tr.tiles_for_rank(tile, 0, n_ranks)
            px = tile * tile
            words = T.compact_words(tile, per)
            dense_g = torch.zeros((n_ranks, n_frames, per * px), dtype=torch.int32, device="cuda")
            sparse_g = torch.zeros((n_ranks, n_frames * words), dtype=torch.int32, device="cuda")
            cwords = T.compact_code_words(tile, per)
            code_g = torch.zeros((n_ranks, n_frames * cwords), dtype=torch.int32, device="cuda")
            assert tr.exchange_code_bits() == 16
            most = 0
            for r in range(n_ranks):
                mine = tr.tiles_for_rank(tile, r, n_ranks)
                hits = torch.zeros((n_frames, per * px, 4), dtype=torch.int32, device="cuda")
                tr.draw_tile_frames_device(np.concatenate(cams[:n_frames]), tile, r, n_ranks, per, hits_ptr=hits.data_ptr(), rgba_ptr=dense_g[r].data_ptr())
                tr.compact_tile_frames_device(dense_g[r].data_ptr(), tile, mine, n_frames, per, sparse_g[r].data_ptr())
                tr.compact_hit_tile_frames_device(hits.data_ptr(), tile, mine, n_frames, per, code_g[r].data_ptr())
                one_h = torch.zeros((per * px, 4), dtype=torch.int32, device="cuda")
                one_c = torch.zeros(per * px, dtype=torch.int32, device="cuda")
                one_s = torch.zeros(words, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                a = sparse_g[r].cpu().numpy().view(np.uint32)
                ref = T.compact_tile_frames(dense_g[r].cpu().numpy(), tile, mine)
                assert (a[:n_frames] == ref[:n_frames]).all()
                # 16-bit codes from the first-hit tiles: same counts, and per frame the same records as the numpy reference
                c = code_g[r].cpu().numpy().view(np.uint32)
                cref = T.compact_hit_tile_frames(hits.cpu().numpy().view(O.HIT).reshape(n_frames, -1), tile, mine, len(pw.materials))
                assert (c[:n_frames] == a[:n_frames]).all() and (cref[:n_frames] == a[:n_frames]).all()
                top, rw = int(a[:n_frames].max()), 1 + px // 2
                cg = c[n_frames:n_frames + top * n_frames * rw].reshape(-1, n_frames, rw)
                cr = cref[n_frames:n_frames + top * n_frames * rw].reshape(-1, n_frames, rw)
                for f in range(n_frames):
                    x, y = cg[:a[f], f], cr[:a[f], f]
                    assert (x[np.argsort(x[:, 0])] == y[np.argsort(y[:, 0])]).all(), ("codes", beam, n_ranks, r, f)
                for f in range(n_frames):
                    tr.draw_tiles_device(cams[f], tile, r, n_ranks, hits_ptr=one_h.data_ptr(), rgba_ptr=one_c.data_ptr())
                    tr.compact_tiles_device(one_c.data_ptr(), tile, mine, one_s.data_ptr())
                    torch.cuda.synchronize()
                    assert torch.equal(hits[f][:mine * px], one_h[:mine * px]), (beam, n_ranks, r, f)
                    assert torch.equal(dense_g[r, f][:mine * px], one_c[:mine * px]), (beam, n_ranks, r, f)
                    b = one_s.cpu().numpy().view(np.uint32)
                    assert a[f] == b[0]
                    # frame f's records sit at slots j * n_frames + f of the interleaved buffer
                    ra = a[n_frames:n_frames + int(a[:n_frames].max()) * n_frames * (1 + px)].reshape(-1, n_frames, 1 + px)[:a[f], f]
                    rb = b[1:1 + b[0] * (1 + px)].reshape(-1, 1 + px)
                    assert (ra[np.argsort(ra[:, 0])] == rb[np.argsort(rb[:, 0])]).all()
                    most = max(most, int(a[f]))
            frames = torch.zeros((n_frames, Ht * Wd), dtype=torch.int32, device="cuda")
            tr.untile_frames_device(dense_g.data_ptr(), 4, tile, n_ranks, n_frames * per, n_frames, per, frames.data_ptr())
            torch.cuda.synchronize()
            for f in range(n_frames):
                assert (frames[f].cpu().numpy().view(np.uint32) == want[f]).all(), ("dense", beam, n_ranks, f)
            # the sparse exchange: the first `most` record slots of every frame of every rank = ONE prefix per rank, as the gather delivers it
            n = n_frames * (1 + most * (1 + px))
            packed = torch.full((n_ranks, n_frames * words), -1, dtype=torch.int32, device="cuda")
            packed[:, :n].copy_(sparse_g[:, :n])
            frames.fill_(12345)                          # without a tile state every pixel is written
            tr.scatter_tile_frames_device(packed.data_ptr(), n_ranks, n_frames * words, tile, most, n_frames, frames.data_ptr())
            torch.cuda.synchronize()
            for f in range(n_frames):
                assert (frames[f].cpu().numpy().view(np.uint32) == want[f]).all(), ("sparse", beam, n_ranks, f)
                assert (T.scatter_tile_frames(packed.cpu().numpy(), n_ranks, n_frames * words, tile, most, n_frames, Wd, Ht)[f].reshape(-1) == want[f]).all()
            # the same through 16-bit codes: half the words travel, the root expands them with the material table
            cn = n_frames * (1 + most * (1 + px // 2))
            cpacked = torch.full((n_ranks, n_frames * cwords), -1, dtype=torch.int32, device="cuda")
            cpacked[:, :cn].copy_(code_g[:, :cn])
            frames.fill_(54321)
            tr.scatter_code_tile_frames_device(cpacked.data_ptr(), n_ranks, n_frames * cwords, tile, most, n_frames, frames.data_ptr())
            torch.cuda.synchronize()
            cpu_frames = T.scatter_code_tile_frames(cpacked.cpu().numpy(), n_ranks, n_frames * cwords, tile, most, n_frames, Wd, Ht, pw.materials["albedo"])
            for f in range(n_frames):
                assert (frames[f].cpu().numpy().view(np.uint32) == want[f]).all(), ("codes", beam, n_ranks, f)
                assert (cpu_frames[f].reshape(-1) == want[f]).all()
            assert cn * 2 < n * 1.01 + 2 * n_frames
            # with a tile state (buffer all sky, state all zero to begin with): the frames, then no records at all (every live tile
            # goes back to sky), then the frames again, then the frames in another order (live <-> sky per tile)
            sky = int(T.SKY_RGBA) - (1 << 32)
            frames.fill_(sky)
            state = torch.zeros((n_frames, T.tiles_total(Wd, Ht, tile)), dtype=torch.uint8, device="cuda")
            empty = torch.zeros_like(packed)
            back = torch.full_like(packed, -1)           # frame f <- frame n_frames - 1 - f
            bw = back[:, :n]; sw = sparse_g[:, :n]
            bw[:, :n_frames] = sw[:, :n_frames].flip(1)
            bw[:, n_frames:].view(n_ranks, most, n_frames, 1 + px).copy_(sw[:, n_frames:].view(n_ranks, most, n_frames, 1 + px).flip(2))
            for step, (buf, order) in enumerate(((packed, 1), (empty, 0), (packed, 1), (back, -1), (packed, 1))):
                tr.scatter_tile_frames_device(buf.data_ptr(), n_ranks, n_frames * words, tile, most, n_frames, frames.data_ptr(), state.data_ptr())
                torch.cuda.synchronize()
                for f in range(n_frames):
                    got = frames[f].cpu().numpy().view(np.uint32)
                    expect = want[f] if order == 1 else want[n_frames - 1 - f] if order == -1 else np.uint32(T.SKY_RGBA)
                    assert (got == expect).all(), ("state", step, beam, n_ranks, f)
            # no record slots at all (every rank's count is zero: an all-sky batch): the buffer goes back to sky
            tr.scatter_tile_frames_device(packed.data_ptr(), n_ranks, n_frames * words, tile, 0, n_frames, frames.data_ptr(), state.data_ptr())
            torch.cuda.synchronize()
            assert (frames.cpu().numpy().view(np.uint32) == np.uint32(T.SKY_RGBA)).all() and not state.any().item()
    tr.set_beam(32)
    # refused: more frames than one launch takes, a frame stride smaller than the rank's tiles
    buf = torch.zeros(9 * tr.tiles_for_rank(32, 0, 2) * 1024, dtype=torch.int32, device="cuda")
    with pytest.raises(Exception):
        tr.draw_tile_frames_device(np.concatenate(cams + cams[:1]), 32, 0, 2, tr.tiles_for_rank(32, 0, 2), rgba_ptr=buf.data_ptr())
    with pytest.raises(Exception):
        tr.draw_tile_frames_device(np.concatenate(cams[:2]), 32, 0, 2, 5, rgba_ptr=buf.data_ptr())
    tr.shutdown()


@pytest.mark.parametrize("vs", [0.5, 2.0, 0.125])
def test_power_of_two_voxel_sizes(tracer_cls, vs):
    """ChunkManager(chunkSize, voxelSize) with voxelSize != 1 (reference chunk_manager.cpp:19-25): for a power of two every box
    plane stays exactly representable, the walk scales its integer planes, and first hits equal the oracle's (which walks the
    scaled SubChunkGpu records literally) bit for bit — frames from outside and inside, explicit rays, with and without the
    pre-pass; the record's voxel is the lattice coordinate.  Other sizes, and a world whose size was not announced, are refused."""
    from blok_amd._ffi import BlokError
    rng = np.random.default_rng(int(vs * 1000))
    cm = W.ChunkManager(128, vs)
    pts = rng.integers(-70, 90, size=(5000, 3)).astype(np.int32)
    wall = np.array([(x, 21, z) for x in range(-50, 60) for z in range(-40, 50)], dtype=np.int32)
    xyz = np.concatenate([pts, wall, np.array([(-190, 60, 140)], dtype=np.int32)])
    cm.set_voxels(xyz, rng.integers(1, 200, size=len(xyz)).astype(np.uint32))
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(SEED))
    assert float(pw.sub_chunks["sub_chunk_size"][0]) == 16 * vs
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    w, h = 320, 180
    tr = tracer_cls(w, h).init()
    with pytest.raises(BlokError) as e:
        tr.add_world(pw)                                    # voxel size not announced: the leaves are not at unit level
    assert e.value.status == -5
    for bad in (0.3, 3.0, 0.0, -2.0, 1024.0):
        with pytest.raises(BlokError) as e:
            tr.set_voxel_size(bad)
        assert e.value.status == -5
    tr.set_voxel_size(vs)
    st = tr.add_world(pw)
    assert st.n_voxels == len(np.unique(xyz, axis=0))
    cams = [W.camera_look_at((150.0 * vs, 120.0 * vs, -160.0 * vs), (0.0, 10.0 * vs, 0.0), 60.0, w, h),
            W.camera_look_at((3.3 * vs, 40.2 * vs, 7.7 * vs), (60.0 * vs, 0.0, 31.0 * vs), 95.0, w, h),
            W.camera_look_at((-400.0 * vs, 90.0 * vs, 300.0 * vs), (-190.0 * vs, 60.5 * vs, 140.5 * vs), 8.0, w, h)]
    for k, cam in enumerate(cams):
        ref, ctr = lat.trace(O.primary_rays(cam, w, h), threads=8)
        assert ctr["hits"] > 50
        for beam in (32, 0, 8):
            tr.set_beam(beam)
            for fused in (0, 1, 2, 4, 5):
                tr.set_fused(fused)
                assert records_equal(tr.draw_frame(cam).reshape(-1), ref).all(), (k, beam, fused)
        tr.set_beam(32); tr.set_fused(3)
        hit = ref[ref["hit"] == 1]
        assert np.isin(hit["voxel"].astype(np.int32).view([("", np.int32)] * 3), np.ascontiguousarray(xyz).view([("", np.int32)] * 3)).all()   # lattice coordinates
    rays = random_rays(int(160 * vs) + 8, 6000, 3)
    rays["org"] -= np.float32(40 * vs)
    ref, ctr = lat.trace(rays, threads=8)
    assert ctr["hits"] > 20 and records_equal(tr.trace_rays(rays), ref).all()
    planes = tr.trace_paths(cams[0], spp=2, max_bounces=2, frame_index=1)
    want, _ = O.render_paths(lat, pw.materials, cams[0], w, h, spp=2, max_bounces=2, frame_index=1, threads=8)
    for key in ("world_pos", "normal_roughness", "albedo_metallic"):
        assert np.array_equal(planes[key], want[key]), key
    assert (np.abs(planes["color"] - want["color"]) <= 1e-4 + 1e-3 * np.abs(want["color"])).all()
    tr.set_voxel_size(1.0)                                  # and back: the unit-voxel world again
    cm1, pw1 = make_scene_world(64)
    tr.add_world(pw1)
    cam1 = W.scene_camera(64, 0, w, h, SEED)
    ref1, _ = O.Lattice(pw1.nodes, pw1.sub_chunks).trace(O.primary_rays(cam1, w, h), threads=8)
    assert records_equal(tr.draw_frame(cam1).reshape(-1), ref1).all()
    tr.shutdown()


def test_multi_device_entry_one_process(tracer_cls, scene1024):
    """blok_hip_multi_* (C ABI; one process, one context and stream per rank): 1, 3 and 8 ranks on device 0 — transport "none" /
    "peer-copy" — give the single-device RGBA8 frame with either exchange (sparse-pull / dense), one and several frames per call,
    and each rank's first-hit records are its tiles of the single-device records; creation with RCCL allowed but a repeated
    device falls back to peer copies; bad arguments are refused."""
    from blok_amd.multi_gpu import HipMultiTracer
    from blok_amd._ffi import BlokError
    from blok_amd import tiles as T
    cm, pw = scene1024
    Wd, Ht = 1920, 1080
    tr = tracer_cls(Wd, Ht).init()
    tr.add_world(pw)
    cam = W.scene_camera(1024, 0, Wd, Ht, SEED)
    want = tr.shade_rgba8(cam)
    hits = tr.draw_frame(cam)
    for devices, tile, transport in (([0], 32, "none"), ([0, 0, 0], 64, "peer-copy"), ([0] * 8, 32, "peer-copy")):
        mt = HipMultiTracer(devices, Wd, Ht, tile=tile, allow_rccl=True)
        assert mt.transport == transport
        mt.add_world(pw)
        n = len(devices)
        cams = [cam] + [W.scene_camera(1024, p, Wd, Ht, SEED) for p in (1, 2)]
        wants = [want] + [tr.shade_rgba8(c) for c in cams[1:]]
        # both exchanges — the default "sparse-pull" (the root reads the ranks' 16-bit code records where they lie) and "dense"
        # (whole RGBA8 tiles travel, then an un-permute) — one frame and three frames per call, back and forth
        for mode, name in ((-1, "sparse-pull"), (0, "dense"), (1, "sparse-pull"), (0, "dense")):
            mt.set_exchange(mode)
            assert mt.exchange == name
            for _ in range(2):
                assert (mt.draw_frame(cam) == want).all(), (devices, name)
            got3 = mt.draw_frames(cams)
            for f in range(3):
                assert (got3[f] == wants[f]).all(), (devices, name, f)
            assert (mt.draw_frame(cams[2]) == wants[2]).all(), (devices, name)       # live <-> sky per tile between calls
            assert (mt.draw_frame(cam) == want).all(), (devices, name)
            for r in (0, n - 1):
                got = mt.rank_hits(r)
                for k, (x0, y0) in enumerate(T.rank_tile_origins(Wd, Ht, tile, r, n)):
                    h, w = min(tile, Ht - y0), min(tile, Wd - x0)
                    block = got[k * tile * tile:(k + 1) * tile * tile].reshape(tile, tile)[:h, :w]
                    assert records_equal(block.reshape(-1), hits[y0:y0 + h, x0:x0 + w].reshape(-1)).all(), (devices, name, r, k)
        # as on a node without peer access: sparse-pull is refused, the default falls back to the dense exchange, same frames
        if n > 1:
            mt.set_exchange(1)
            mt.deny_peer_access(True)
            assert mt.exchange == "dense"
            with pytest.raises(BlokError):
                mt.set_exchange(1)
            for mode in (-1, 0):
                mt.set_exchange(mode)
                assert mt.exchange == "dense" and (mt.draw_frame(cam) == want).all(), (devices, "denied", mode)
            mt.deny_peer_access(False)
            mt.set_exchange(-1)
            assert mt.exchange == "sparse-pull" and (mt.draw_frame(cams[1]) == wants[1]).all()
            # calls are asynchronous: back-to-back calls without a synchronize in between must not tear the first one's frames (a peer's
            # next trace / compact waits for the root's assembly of the previous call)
            for mode in (-1, 0):
                mt.set_exchange(mode)
                for k in range(6):
                    mt.draw_frames_async([cams[k % 3]])
                mt.synchronize()
                assert (mt.draw_frame(cams[2]) == wants[2]).all(), (devices, "async", mode)
        with pytest.raises(BlokError):
            mt.draw_frames([cam] * 9)
        mt.shutdown()
    with pytest.raises(BlokError):
        HipMultiTracer([0, 99], Wd, Ht)
    with pytest.raises(BlokError):
        HipMultiTracer([0], Wd, Ht, tile=20)
    tr.shutdown()


def test_dense_upload_device_build(tracer_cls):
    """blok_hip_upload_dense builds on the device straight from the id grid: ragged extents, negative origin,
    same frames as the general host path; 256^3 timing printed."""
    import time
    rng = np.random.default_rng(12)
    ids = np.where(rng.random((37, 22, 51)) < 0.03, rng.integers(1, 300, size=(37, 22, 51)), 0).astype(np.uint32)   # [z][y][x]
    origin = (-20, 5, -9)
    z, y, x = np.nonzero(ids)
    cm = W.ChunkManager(128, 1.0)
    cm.set_voxels(np.stack([x + origin[0], y + origin[1], z + origin[2]], 1), ids[z, y, x])
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo()
    rays = random_rays(60, 6000, 2)
    rays["org"] += np.array(origin, dtype=np.float32)
    ref, ctr = O.Lattice(pw.nodes, pw.sub_chunks).trace(rays, threads=4)
    tr = tracer_cls(64, 64).init()
    st = tr.add_dense(ids, origin)
    assert tr.built_on_device() and st.n_voxels == len(x) and list(st.origin) == list(origin) and ctr["hits"] > 300
    assert records_equal(tr.trace_rays(rays), ref).all()
    tr.set_host_build(True)
    tr.add_dense(ids, origin)
    assert not tr.built_on_device() and records_equal(tr.trace_rays(rays), ref).all()
    tr.set_host_build(False)
    big = W.scene_dense(256, SEED)
    t0 = time.perf_counter(); tr.add_dense(big); t_dev = time.perf_counter() - t0
    tr.set_host_build(True)
    t0 = time.perf_counter(); tr.add_dense(big); t_host = time.perf_counter() - t0
    print(f"dense 256^3 upload: device build {t_dev * 1e3:.1f} ms, host build {t_host * 1e3:.1f} ms")
    tr.shutdown()
