"""SURVEY.md §8(f) N3 (host side): sphere brush edits -> dirty chunks -> rebuild -> re-upload
(reference blok/src/brush.cpp:13-63).  Product vs oracle, byte for byte, then the edited world on the GPU."""
import numpy as np
import pytest

from blok_amd import world as W
from tests import oracle_ffi as O
from tests.conftest import SEED, records_equal


def edited_worlds():
    cm, ow = W.ChunkManager(128, 1.0), O.OracleWorld(128, 1.0)
    ids = W.scene_dense(64, SEED)
    z, y, x = np.nonzero(ids)
    xyz = np.stack([x, y, z], 1).astype(np.int32)
    cm.set_voxels(xyz, ids[z, y, x]); ow.set_voxels(xyz, ids[z, y, x])
    edits = [((32.3, 20.7, 31.9), 9.5, 0.0, "subtract"),      # dig a crater
             ((20.0, 40.0, 20.0), 6.25, 1.0, "add"),           # float a ball of material-0 voxels
             ((126.5, 30.2, 10.1), 5.0, 0.75, "add"),          # straddles the chunk border at x = 128
             ((-3.0, 5.0, 8.0), 4.0, 1.0, "add"),              # negative coordinates
             ((20.0, 40.0, 20.0), 3.0, -1.0, "subtract"),      # hollow the ball (negative density = empty)
             ((32.3, 20.7, 31.9), 3.0, 0.5, "add")]            # refill part of the crater: old material ids return
    for c, r, v, m in edits:
        cm.apply_brush(c, r, v, m); ow.apply_brush(c, r, v, m)
    return cm, ow, edits


def test_brush_edits_match_oracle():
    cm, ow, edits = edited_worlds()
    assert cm.rebuild_dirty_chunks() == ow.rebuild()
    pw = cm.pack_chunks_to_gpu_svo()
    on, osub = ow.pack()
    assert pw.nodes.tobytes() == on.tobytes() and pw.sub_chunks.tobytes() == osub.tobytes()
    assert cm.chunk_count() == ow.n_chunks() >= 3
    rng = np.random.default_rng(0)
    for c, r, _, _ in edits:
        for p in np.asarray(c) + rng.uniform(-r - 1, r + 1, size=(200, 3)):
            assert cm.get_voxel_material(p) == ow.get_voxel_material(p)
    assert cm.get_voxel_material((32.3, 20.7, 27.0)) == 0                   # inside the crater, outside the refill
    assert cm.get_voxel_material((20.0, 44.5, 20.0)) == 0                   # ball shell: density 1 but material id 0
    home = [i for i in range(cm.chunk_count()) if cm.chunk(i)[0] == (0, 0, 0)][0]
    assert cm.find_leaf(home, 20, 44, 20) >= 0 and cm.find_leaf(home, 20, 40, 20) < 0


@pytest.mark.gpu
def test_edited_world_on_gpu():
    """edit -> rebuildDirtyChunks -> packChunksToGpuSvo -> updateWorld -> frame (todo.txt:16's direction)."""
    from blok_amd.tracer import HipTracer
    cm, ow, _ = edited_worlds()
    w, h = 320, 240
    tr = HipTracer(w, h).init()
    cam = W.camera_look_at((70.0, 60.0, -20.0), (30.0, 20.0, 30.0), 60.0, w, h)
    before = None
    for step in range(2):
        cm.rebuild_dirty_chunks()
        pw = cm.pack_chunks_to_gpu_svo(W.scene_materials(SEED))
        tr.update_world(pw)
        ref, ctr = O.Lattice(pw.nodes, pw.sub_chunks).trace(O.primary_rays(cam, w, h), threads=8)
        got = tr.draw_frame(cam).reshape(-1)
        assert ctr["hits"] > 5000 and records_equal(got, ref).all()
        if before is not None:
            assert (~records_equal(got, before)).sum() > 100                # the new edit is visible
        before = got
        cm.apply_brush((40.0, 25.0, 30.0), 8.0, 0.0, "subtract")
    tr.shutdown()


@pytest.mark.gpu
def test_device_resident_volume_edits_and_rebuild():
    """N3 on the GPU: the dense store lives in HBM (blok_hip_volume_*), setVoxelMaterial / applyBrush are kernels with
    the reference's arithmetic and the rebuild derives the traversal structure on the device.  After the same edits
    as the oracle's ChunkManager: the dense arrays are bit-identical to the oracle's chunks, and frames of the rebuilt
    world equal frames of the oracle's packed records uploaded the ordinary way."""
    from blok_amd.tracer import HipTracer
    from blok_amd._ffi import BlokError
    _, ow, edits = edited_worlds()
    C_ = 128
    w, h = 320, 240
    mats = W.scene_materials(SEED)
    vol = HipTracer(w, h).init()
    box_origin, box_shape = (-C_, 0, 0), (3 * C_, C_, C_)               # chunks (-1..1, 0, 0): everything the edits touch
    vol.volume_create(box_origin, box_shape, C_, 1.0)
    ids = W.scene_dense(64, SEED)
    z, y, x = np.nonzero(ids)
    xyz = np.stack([x, y, z], 1).astype(np.int32)
    vol.volume_set_voxels(xyz, ids[z, y, x], np.ones(len(xyz), dtype=np.float32))
    # later entries of the same voxel win, like sequential setVoxelMaterial calls
    vol.volume_set_voxels([[5, 5, 5], [5, 5, 5]], [7, 9], [1.0, 0.25])
    d, m = vol.volume_download()
    assert d[5, 5, 5 + C_] == np.float32(0.25) and m[5, 5, 5 + C_] == 9      # arrays are [z][y][x - origin x]
    vol.volume_set_voxels([[5, 5, 5]], [int(ids[5, 5, 5])], [1.0 if ids[5, 5, 5] else 0.0])
    for c, r, v, mode in edits:
        vol.volume_apply_brush(c, r, v, {"add": 0, "subtract": 1}[mode])
    # 1. dense arrays == the oracle's chunks, bit for bit (signs of zero and negative densities included)
    d, m = vol.volume_download()
    seen = 0
    for i in range(ow.n_chunks()):
        (cx, cy, cz), _ = ow.chunk(i)
        od, om = ow.chunk_dense(i, C_)
        x0 = cx * C_ - box_origin[0]
        assert cy == 0 and cz == 0 and 0 <= x0 < box_shape[0]
        assert d[:, :, x0:x0 + C_].tobytes() == od.tobytes(), (cx, cy, cz)
        filled = od > 0
        assert np.array_equal(m[:, :, x0:x0 + C_][filled], om[filled])
        seen += 1
    assert seen == ow.n_chunks() >= 3
    # 2. frames of the device-rebuilt world == frames of the oracle's records through the ordinary upload
    st = vol.volume_rebuild(mats)
    assert vol.built_on_device()
    ow.rebuild()
    on, osub = ow.pack()
    ref_tr = HipTracer(w, h).init()
    ref_st = ref_tr.add_world(W.PackedWorld(on, osub, mats))
    assert st.n_voxels == ref_st.n_voxels == int((np.concatenate([ow.chunk_dense(i, C_)[0].ravel() for i in range(ow.n_chunks())]) > 0).sum())
    lat = O.Lattice(on, osub)
    for eye, at in [((70.0, 60.0, -20.0), (30.0, 20.0, 30.0)), ((-30.0, 30.0, 40.0), (20.0, 20.0, 20.0)), ((150.0, 50.0, 60.0), (120.0, 30.0, 10.0))]:
        cam = W.camera_look_at(eye, at, 60.0, w, h)
        got = vol.draw_frame(cam).reshape(-1)
        assert records_equal(got, ref_tr.draw_frame(cam).reshape(-1)).all()
        ref, ctr = lat.trace(O.primary_rays(cam, w, h), threads=8)
        assert ctr["hits"] > 1000 and records_equal(got, ref).all()
    # 3. a further edit + rebuild changes the picture; edits leaving the box are refused and write nothing
    cam = W.camera_look_at((70.0, 60.0, -20.0), (30.0, 20.0, 30.0), 60.0, w, h)
    before = vol.draw_frame(cam).reshape(-1)
    vol.volume_apply_brush((40.0, 25.0, 30.0), 8.0, 0.0, 1)
    vol.volume_rebuild(mats)
    ow.apply_brush((40.0, 25.0, 30.0), 8.0, 0.0, "subtract"); ow.rebuild()
    on, osub = ow.pack()
    ref, _ = O.Lattice(on, osub).trace(O.primary_rays(cam, w, h), threads=8)
    after = vol.draw_frame(cam).reshape(-1)
    assert records_equal(after, ref).all() and (~records_equal(after, before)).sum() > 100
    d0, _ = vol.volume_download()
    with pytest.raises(BlokError):
        vol.volume_apply_brush((60.0, 125.0, 30.0), 6.0, 1.0, 0)        # reaches y = 131 > 128
    with pytest.raises(BlokError):
        vol.volume_set_voxels([[0, -1, 0]])
    assert vol.volume_download()[0].tobytes() == d0.tobytes()
    # 4. an emptied volume is an empty world
    vol.volume_upload(None, None)
    assert vol.volume_rebuild(mats).n_voxels == 0 and (vol.draw_frame(cam)["hit"] == 0).all()
    vol.shutdown(); ref_tr.shutdown()


@pytest.mark.gpu
def test_keyed_volume_rebuild_equals_the_general_rebuild():
    """The rebuild of a keyed volume (bricks indexed by their tree key under a pyramid of occupancy words: scans over the pyramid, material
    ids of untouched bricks taken over from the previous build) against the general one (all bricks scanned, keyed, radix-sorted, upper
    levels grouped on the host) over the same edit history: node and material arrays byte for byte after every rebuild — bricks and
    whole 16^3 / 64^3 cells appearing and vanishing, a material id changing under an unchanged mask, density without a material,
    several edits per rebuild, a rebuild without an edit, the box emptied and refilled — and the path kernel's frames (the shadow
    rays' last-occluder map is patched for the edited region only)."""
    from blok_amd.tracer import HipTracer
    C_ = 128
    w, h = 160, 120
    mats = W.scene_materials(SEED)
    rng = np.random.default_rng(5)
    a, b = HipTracer(w, h).init(), HipTracer(w, h).init()
    b.set_volume_layout(False)
    origin, shape = (-64, -16, 0), (200, 120, 136)                      # ragged: not a multiple of 4 bricks per level-2 cell, levels = 4
    for t in (a, b):
        t.volume_create(origin, shape, C_, 1.0)
    ids = W.scene_dense(64, SEED)
    z, y, x = np.nonzero(ids)
    xyz = np.stack([x, y, z], 1).astype(np.int32)
    cam = W.camera_look_at((90.0, 80.0, -40.0), (30.0, 20.0, 30.0), 60.0, w, h)

    def both(fn):
        for t in (a, b):
            fn(t)

    def check(tag):
        sa, sb = a.volume_rebuild(mats), b.volume_rebuild(mats)
        assert (sa.n_voxels, sa.n_tree_nodes, sa.levels, tuple(sa.origin)) == (sb.n_voxels, sb.n_tree_nodes, sb.levels, tuple(sb.origin)), tag
        if sa.n_voxels == 0:
            return sa
        na, ma = a.download_tree(); nb, mb = b.download_tree()
        assert na.tobytes() == nb.tobytes(), tag
        assert ma.tobytes() == mb.tobytes(), tag
        pa, pb = a.trace_paths(cam, spp=2, max_bounces=2, frame_index=3), b.trace_paths(cam, spp=2, max_bounces=2, frame_index=3)
        for k in pa:
            assert pa[k].tobytes() == pb[k].tobytes(), (tag, k)
        return sa

    both(lambda t: t.volume_set_voxels(xyz, ids[z, y, x], np.ones(len(xyz), dtype=np.float32)))
    first = check("scene")
    assert first.n_voxels == len(xyz) and first.levels == 4
    check("no edit")                                                      # nothing dirty: every material id comes from the previous build
    both(lambda t: t.volume_apply_brush((30.0, 25.0, 30.0), 8.0, 0.0, 1))
    check("dig")
    both(lambda t: t.volume_apply_brush((100.0, 90.0, 100.0), 9.5, 1.0, 0))            # a ball in empty space: new bricks, new 16^3 and 64^3 cells, no material ids
    check("ball")
    both(lambda t: t.volume_set_voxels([[10, 3, 12], [10, 3, 12], [11, 3, 12]], [77, 78, 5], [1.0, 1.0, 1.0]))   # ids change, masks may not
    check("ids")
    for k in range(4):                                                    # several edits, one rebuild
        c = tuple(float(v) for v in rng.uniform((-40, 0, 20), (110, 80, 110)))
        both(lambda t: t.volume_apply_brush(c, float(rng.integers(2, 12)), float(k % 2), k % 2))
    check("several")
    both(lambda t: t.volume_apply_brush((100.0, 90.0, 100.0), 12.0, 0.0, 1))            # the ball goes: its bricks and cells vanish
    check("ball gone")
    both(lambda t: t.volume_upload(None, None))
    assert check("empty").n_voxels == 0
    both(lambda t: t.volume_set_voxels(xyz[::3], ids[z, y, x][::3], np.ones(len(xyz[::3]), dtype=np.float32)))
    assert check("refilled").n_voxels == len(xyz[::3])
    a.shutdown(); b.shutdown()


@pytest.mark.gpu
def test_sun_map_stays_valid_across_volume_edits():
    """Round 4: the shadow rays' last-occluder map is no longer searched again behind every edit (30-60 us of a 0.4 ms brush -> tree latency): an edit that
    can only have emptied voxels leaves it alone, one that may have filled some raises the texels under its box to the box's far corner along the sun,
    and the search runs over the union of the boxes every 32 edits.  The map is an upper bound, so every path-traced plane must stay bit-identical with
    the map on and off — after balls added above the terrain (new shadows), after digging, across the 32-edit tightening, after setVoxel writes."""
    from blok_amd.tracer import HipTracer
    w, h = 160, 120
    mats = W.scene_materials(SEED)
    rng = np.random.default_rng(11)
    tr = HipTracer(w, h).init()
    tr.volume_create((0, 0, 0), (64, 96, 64), 128, 1.0)
    ids = W.scene_dense(64, SEED)
    z, y, x = np.nonzero(ids)
    tr.volume_set_voxels(np.stack([x, y, z], 1).astype(np.int32), ids[z, y, x], np.ones(len(x), dtype=np.float32))
    tr.volume_rebuild(mats)
    cams = [W.scene_camera(64, 0, w, h, SEED), W.camera_look_at((5.0, 30.0, 5.0), (40.0, 12.0, 40.0), 70.0, w, h)]

    def same(tag):
        for cam in cams:
            tr.set_sun_map(False)
            plain = tr.trace_paths(cam, spp=3, max_bounces=2, frame_index=4)
            tr.set_sun_map(True)
            got = tr.trace_paths(cam, spp=3, max_bounces=2, frame_index=4)
            for k in plain:
                assert got[k].tobytes() == plain[k].tobytes(), (tag, k)

    same("scene")
    for k in range(70):
        c = tuple(float(v) for v in rng.uniform((9, 22, 9), (54, 86, 54)))
        if k % 3 == 2:
            tr.volume_apply_brush(c, float(rng.integers(2, 7)), 0.0, 1)                       # dig
        elif k % 7 == 3:
            p = np.array([[int(c[0]), int(c[1]), int(c[2])], [int(c[0]) + 1, int(c[1]), int(c[2])]], dtype=np.int32)
            tr.volume_set_voxels(p, [9, 10], [1.0, 1.0])                                      # setVoxel: may fill
        else:
            tr.volume_apply_brush(c, float(rng.integers(2, 6)), 1.0, 0)                       # a ball: a new shadow
        tr.volume_rebuild(mats)
        if k % 6 == 0 or k in (30, 31, 32, 33, 63, 64, 65):
            same(k)
    tr.shutdown()


@pytest.mark.gpu
def test_sun_map_tightens_in_bands_on_a_wide_world():
    """Round 4: with texels of one voxel the union of 32 scattered edits' boxes can span the map, so it is made tight a band of rows per edit (api.hip:
    update_sun_map, at most 32 768 texels each).  A world wide enough for several bands (256 x 128 x 256: a map of ~450 x 450 texels), edits scattered
    over all of it: planes bit-identical with the map on and off before the tightening starts, while bands are pending, after a second union has joined
    the pending rectangle, and at the end."""
    from blok_amd.tracer import HipTracer
    w, h = 160, 120
    mats = W.scene_materials(SEED)
    rng = np.random.default_rng(23)
    tr = HipTracer(w, h).init()
    tr.volume_create((0, 0, 0), (256, 128, 256), 128, 1.0)
    ids = W.scene_dense(256, SEED)[:, :128, :]
    z, y, x = np.nonzero(ids)
    tr.volume_set_voxels(np.stack([x, y, z], 1).astype(np.int32), ids[z, y, x], np.ones(len(x), dtype=np.float32))
    tr.volume_rebuild(mats)
    cams = [W.camera_look_at((-60.0, 150.0, -60.0), (128.0, 30.0, 128.0), 60.0, w, h), W.camera_look_at((20.0, 70.0, 20.0), (200.0, 20.0, 180.0), 70.0, w, h)]

    def same(tag):
        for cam in cams:
            tr.set_sun_map(False)
            plain = tr.trace_paths(cam, spp=2, max_bounces=2, frame_index=4)
            tr.set_sun_map(True)
            got = tr.trace_paths(cam, spp=2, max_bounces=2, frame_index=4)
            for k in plain:
                assert got[k].tobytes() == plain[k].tobytes(), (tag, k)

    same("scene")
    for k in range(72):
        c = tuple(float(v) for v in rng.uniform((12, 20, 12), (244, 110, 244)))
        if k % 3 == 2:
            tr.volume_apply_brush(c, float(rng.integers(2, 7)), 0.0, 1)                       # dig
        else:
            tr.volume_apply_brush(c, float(rng.integers(2, 6)), 1.0, 0)                       # a ball: a new shadow
        tr.volume_rebuild(mats)
        if k in (5, 30, 31, 32, 33, 36, 50, 63, 64, 65, 71):
            same(k)
    tr.shutdown()
