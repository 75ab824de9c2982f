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
