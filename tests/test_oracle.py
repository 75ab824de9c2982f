"""CPU: the oracle against the reference's own vectors and against itself (three formulations)."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from tests import oracle_ffi as O
from tests.conftest import SEED, edge_case_rays, random_rays, records_equal

GOLDEN = Path(__file__).resolve().parent / "golden"


def test_morton_matches_reference_vectors():
    """Golden vectors were produced by the reference's morton.hpp compiled in place."""
    vec = json.loads((GOLDEN / "morton_reference.json").read_text())["vectors"]
    assert len(vec) > 200
    L = O.lib()
    for v in vec:
        x, y, z = v["xyz"]
        code = int(v["code"], 16)
        assert L.orc_morton_encode(x, y, z) == code
        import ctypes as C
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        L.orc_morton_decode(code, C.byref(a), C.byref(b), C.byref(c))
        assert (a.value, b.value, c.value) == (x, y, z)
        assert [L.orc_morton_octant(code, 7, lvl) for lvl in range(7)] == v["octants_depth7"]


def test_morton_against_live_reference_build():
    ref = O.ref_morton()
    if ref is None:
        pytest.skip("oracle/_ref/libref_morton.so not present (reference tree absent and no prebuilt copy)")
    rng = np.random.default_rng(11)
    L = O.lib()
    for x, y, z in rng.integers(-(1 << 20), 1 << 20, size=(2000, 3)):
        x, y, z = int(x), int(y), int(z)
        code = int(ref.ref_morton_encode(x, y, z))
        assert L.orc_morton_encode(x, y, z) == code
        for lvl in range(7):
            assert L.orc_morton_octant(code, 7, lvl) == ref.ref_morton_octant(code, 7, lvl)


def test_svo_builder_digests():
    cases = json.loads((GOLDEN / "svo_builder.json").read_text())["cases"]
    for c in cases:
        rng = np.random.default_rng(c["seed"])
        span = c["span"]
        xyz = rng.integers(-span if c["seed"] == 3 else 0, span, size=(c["count"], 3)).astype(np.int32)
        mats = rng.integers(1, 1 << 16, size=c["count"]).astype(np.uint32)
        ow = O.OracleWorld(128, 1.0)
        ow.set_voxels(xyz, mats)
        ow.rebuild()
        nodes, subs = ow.pack()
        assert len(nodes) == c["n_nodes"] and len(subs) == c["n_sub_chunks"]
        assert hashlib.sha256(nodes.tobytes()).hexdigest() == c["nodes_sha256"]
        assert hashlib.sha256(subs.tobytes()).hexdigest() == c["sub_chunks_sha256"]


def test_find_leaf_agrees_with_dense_store():
    """SvoTree::findLeaf (svo.cpp:103-130) vs ChunkManager::getVoxelMaterial (chunk_manager.cpp:330-348)."""
    rng = np.random.default_rng(5)
    xyz = rng.integers(0, 128, size=(3000, 3)).astype(np.int32)
    mats = rng.integers(1, 1000, size=3000).astype(np.uint32)
    ow = O.OracleWorld(128, 1.0)
    ow.set_voxels(xyz, mats)
    ow.set_voxel((3.5, 4.5, 5.5), 77, 1.0)
    ow.set_voxel((3.5, 4.5, 5.5), 78, 0.0)      # cleared again: density 0 is never inserted (svo.cpp:60-61)
    ow.rebuild()
    _, nodes = ow.chunk(0)
    probe = rng.integers(0, 128, size=(4000, 3))
    probe = np.concatenate([probe, xyz[:500], [[3, 4, 5]]])
    for x, y, z in probe:
        leaf = ow.find_leaf(0, int(x), int(y), int(z))
        mat = ow.get_voxel_material((x + 0.5, y + 0.5, z + 0.5))
        if mat == 0:
            assert leaf < 0
        else:
            assert leaf >= 0 and nodes[leaf]["material_id"] == mat and nodes[leaf]["occupancy"] > 0
    assert ow.find_leaf(0, 3, 4, 5) < 0


def _voxels_of(cm):
    from blok_amd import world as W
    # the dense generator gives the same voxel set as the chunked one
    ids = W.scene_dense(64, SEED)
    z, y, x = np.nonzero(ids)
    return np.stack([x, y, z], 1).astype(np.int32), ids[z, y, x]


@pytest.mark.parametrize("pose", [0, 1, 2])
def test_three_formulations_agree(scene64, pose):
    """Literal shader over every sub-chunk in array order == literal shader behind the front-to-back slot
    lattice == minimum over all filled voxels of the leaf-level slab test; and no ray has a tied minimum."""
    from blok_amd import world as W
    cm, pw = scene64
    xyz, mats = _voxels_of(cm)
    cam = W.scene_camera(64, pose, 80, 80, SEED)
    rays = O.primary_rays(cam, 80, 80)
    hb, cb = O.trace_bruteforce(pw.nodes, pw.sub_chunks, rays)
    hl, cl = O.Lattice(pw.nodes, pw.sub_chunks).trace(rays)
    hv, cv = O.trace_voxels_bruteforce(xyz, mats, rays)
    assert records_equal(hb, hl).all()
    assert records_equal(hb, hv).all()
    assert cv["ties"] == 0
    assert cb["iter_limit_hits"] == 0 and cb["stack_limit_hits"] == 0      # MAX_ITER / MAX_STACK never bind
    assert cl["sub_chunks_entered"] <= cb["sub_chunks_entered"]
    assert cb["hits"] > 0


def test_edge_case_rays_formulations_agree(scene64):
    cm, pw = scene64
    xyz, mats = _voxels_of(cm)
    rays = np.concatenate([edge_case_rays(), random_rays(64, 600, 3)])
    hb, cb = O.trace_bruteforce(pw.nodes, pw.sub_chunks, rays)
    hl, _ = O.Lattice(pw.nodes, pw.sub_chunks).trace(rays)
    hv, cv = O.trace_voxels_bruteforce(xyz, mats, rays)
    assert records_equal(hb, hl).all()
    assert records_equal(hb, hv).all()
    assert cv["ties"] == 0 and cb["hits"] > 100


def test_first_hit_golden(scene64):
    """BASELINE.json configs[0]: 64^3 grid, 256x256, CPU reference path."""
    from blok_amd import world as W
    cm, pw = scene64
    meta = json.loads((GOLDEN / "first_hit_64.json").read_text())
    gold = np.load(GOLDEN / "first_hit_64.npz")
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    for pose in (0, 1, 2):
        cam = W.scene_camera(64, pose, 256, 256, SEED)
        assert cam.tobytes() == gold[f"cam_pose{pose}"].tobytes()
        hits, ctr = lat.trace(O.primary_rays(cam, 256, 256), threads=4)
        assert hashlib.sha256(hits.tobytes()).hexdigest() == meta[f"pose{pose}"]["sha256_full"]
        assert records_equal(hits.reshape(256, 256)[96:160, 96:160], gold[f"crop_pose{pose}"]).all()
        assert int(ctr["hits"]) == meta[f"pose{pose}"]["hits"]


def test_hit_surface_unpack():
    """hit.rchit:58-76: face LUT, material clamp, flag unpack, roughness floor."""
    from blok_amd import world as W
    mats = W.scene_materials(SEED)
    hit = np.zeros(1, dtype=O.HIT)
    hit["t"], hit["material_id"], hit["face"], hit["hit"] = 12.5, 3, 3, 1
    s = O.shade_surface(hit, mats)
    assert list(s[:3]) == [0, -1, 0]
    assert np.allclose(s[3:6], mats[3]["albedo"])
    assert s[6] == np.float32(127 / 255.0) and s[7] == 0.0 and s[8] == 12.5
    mats2 = mats.copy()
    mats2[3]["flags"] = (255 << 24) | (2 << 16)
    s = O.shade_surface(hit, mats2)
    assert s[6] == np.float32(0.04) and s[7] == 1.0
