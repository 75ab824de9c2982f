"""CPU: the product's host data model (libblok_host.so) against the oracle, bit for bit."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from blok_amd import world as W
from tests import oracle_ffi as O
from tests.conftest import SEED

GOLDEN = Path(__file__).resolve().parent / "golden"


def test_morton_reference_vectors():
    for v in json.loads((GOLDEN / "morton_reference.json").read_text())["vectors"]:
        x, y, z = v["xyz"]
        code = int(v["code"], 16)
        assert W.morton_encode(x, y, z) == code
        assert W.morton_decode(code) == (x, y, z)
        assert [W.morton_octant(code, 7, lvl) for lvl in range(7)] == v["octants_depth7"]


@pytest.mark.parametrize("case", json.loads((GOLDEN / "svo_builder.json").read_text())["cases"], ids=lambda c: f"seed{c['seed']}")
def test_builder_matches_oracle_digests(case):
    rng = np.random.default_rng(case["seed"])
    span = case["span"]
    xyz = rng.integers(-span if case["seed"] == 3 else 0, span, size=(case["count"], 3)).astype(np.int32)
    mats = rng.integers(1, 1 << 16, size=case["count"]).astype(np.uint32)
    cm = W.ChunkManager(128, 1.0)
    cm.set_voxels(xyz, mats)
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo()
    assert hashlib.sha256(pw.nodes.tobytes()).hexdigest() == case["nodes_sha256"]
    assert hashlib.sha256(pw.sub_chunks.tobytes()).hexdigest() == case["sub_chunks_sha256"]
    assert cm.chunk_count() == len(case["chunks"])
    for i, c in enumerate(case["chunks"]):
        coord, nodes = cm.chunk(i)
        assert list(coord) == c["coord"] and hashlib.sha256(nodes.tobytes()).hexdigest() == c["sha256"]


def test_float_writes_overwrites_and_clears_match_oracle():
    """setVoxelMaterial semantics: floor mapping incl. negatives, last write wins, density <= 0 clears."""
    rng = np.random.default_rng(21)
    cm, ow = W.ChunkManager(128, 1.0), O.OracleWorld(128, 1.0)
    pts = rng.uniform(-140, 140, size=(1500, 3)).astype(np.float32)
    pts[:300] = np.floor(pts[:300])                 # exactly on lattice planes
    pts[300:500] = pts[:200]                        # overwrites
    mats = rng.integers(1, 500, size=len(pts))
    dens = np.where(rng.random(len(pts)) < 0.15, 0.0, rng.uniform(0.1, 2.0, len(pts))).astype(np.float32)
    for p, m, d in zip(pts, mats, dens):
        cm.set_voxel_material(p, int(m), float(d))
        ow.set_voxel(p, int(m), float(d))
    cm.rebuild_dirty_chunks()
    ow.rebuild()
    pw = cm.pack_chunks_to_gpu_svo()
    on, osub = ow.pack()
    assert pw.nodes.tobytes() == on.tobytes() and pw.sub_chunks.tobytes() == osub.tobytes()
    for p in pts[::7]:
        assert cm.get_voxel_material(p) == ow.get_voxel_material(p)
    assert cm.get_voxel_material((1e4, 0, 0)) == 0


def test_rebuild_budget_and_find_leaf():
    cm = W.ChunkManager(128, 1.0)
    cm.set_voxels(np.array([[1, 2, 3], [200, 2, 3], [1, 300, 3]], dtype=np.int32), np.array([5, 6, 7], dtype=np.uint32))
    assert cm.rebuild_dirty_chunks(2) == 2          # maxPerFrame (chunk_manager.cpp:125-126)
    assert cm.rebuild_dirty_chunks(16) == 1
    assert cm.rebuild_dirty_chunks(16) == 0
    assert cm.chunk_count() == 3
    leaf = cm.find_leaf(0, 1, 2, 3)
    coord, nodes = cm.chunk(0)
    assert coord == (0, 0, 0) and leaf >= 0 and nodes[leaf]["material_id"] == 5
    assert cm.find_leaf(0, 1, 2, 4) == -1
    assert len(nodes) == 1 + 8 * 7                  # one path: 8 siblings per level (svo.cpp:36-57)


def test_scene_generator_forms_agree():
    cm = W.ChunkManager(128, 1.0)
    cm.generate_scene(64, SEED)
    cm.rebuild_dirty_chunks()
    ids = W.scene_dense(64, SEED)
    z, y, x = np.nonzero(ids)
    for i in range(0, len(x), 97):
        assert cm.get_voxel_material((x[i] + 0.5, y[i] + 0.5, z[i] + 0.5)) == ids[z[i], y[i], x[i]]
    ow = O.OracleWorld(128, 1.0)
    ow.set_voxels(np.stack([x, y, z], 1), ids[z, y, x])
    ow.rebuild()
    on, osub = ow.pack()
    pw = cm.pack_chunks_to_gpu_svo()
    assert pw.nodes.tobytes() == on.tobytes() and pw.sub_chunks.tobytes() == osub.tobytes()
    fill = len(x) / 64 ** 3
    assert 0.01 < fill < 0.08


def test_camera_basis_is_orthonormal_and_matches_reference_convention():
    cam = W.camera_from_yaw_pitch((0, 10, -5), 0.0, 0.0, 60.0, 1280, 720)[0]      # Camera defaults, camera.hpp:17-20
    assert np.allclose(cam["fwd"], (1, 0, 0), atol=1e-7)
    assert np.allclose(cam["right"], (0, 0, 1), atol=1e-7)     # cross(forward, +Y)
    assert np.allclose(cam["up"], (0, 1, 0), atol=1e-7)
    assert np.isclose(cam["aspect"], 1280 / 720)
    assert np.isclose(cam["tan_half_fov"], np.tan(0.5 * 60 * 3.14159 / 180), rtol=1e-6)   # cuda_tracer.cu:406
    cam = W.scene_camera(1024, 0, 3840, 2160)[0]
    for a in ("fwd", "right", "up"):
        assert np.isclose(np.linalg.norm(cam[a]), 1, atol=1e-6)
    assert abs(np.dot(cam["fwd"], cam["right"])) < 1e-6 and abs(np.dot(cam["fwd"], cam["up"])) < 1e-6


def test_taa_jitter_sequence_hand_computed():
    """A8: frame k uses entry k mod 16 of Halton(2,3) - 0.5 (reference renderer_postprocess.cpp:208-228, 660-663).
    Radical inverses by hand: base 2 -> 1/2, 1/4, 3/4, 1/8; base 3 -> 1/3, 2/3, 1/9, 4/9."""
    want = [(0.0, 1 / 3 - 0.5), (0.25 - 0.5, 2 / 3 - 0.5), (0.75 - 0.5, 1 / 9 - 0.5), (0.125 - 0.5, 4 / 9 - 0.5)]
    for k, (jx, jy) in enumerate(want):
        got = W.taa_jitter(k)
        assert abs(float(got[0]) - jx) < 1e-7 and abs(float(got[1]) - jy) < 1e-6, (k, got)
    assert float(W.taa_jitter(0)[0]) == 0.0 and abs(float(W.taa_jitter(0)[1]) + 1 / 6) < 1e-7        # frame 0 = (0, -1/6) px
    for k in (0, 5, 15):
        assert (W.taa_jitter(k) == W.taa_jitter(k + 16)).all() and (W.taa_jitter(k) == W.taa_jitter(k + 160)).all()
    seq = np.array([W.taa_jitter(k) for k in range(16)])
    assert (np.abs(seq) <= 0.5).all() and len({tuple(v) for v in seq}) == 16


def test_matrix_form_ray_equals_basis_form_with_pixel_offset():
    """A8/A9: the ray raygen.rgen:201-205 forms from invProj / invView of the (jittered) matrices is the basis-form ray of the
    kernels through the pixel centre moved by the jitter.  Matrices in float, unprojection in double: agreement to ~1e-6."""
    w, h = 640, 360
    for cam in (W.scene_camera(1024, 0, w, h), W.scene_camera(1024, 1, w, h), W.camera_look_at((3.0, 40.0, -7.0), (60.0, 2.0, 31.0), 75.0, w, h)):
        c = cam[0]
        view = W.camera_view(cam).astype(np.float64).reshape(4, 4).T
        proj = W.camera_projection(cam)
        for frame in (0, 1, 2, 7):
            j = W.taa_jitter(frame) if frame else np.zeros(2, dtype=np.float32)
            pj = W.jittered_projection(proj, j, w, h)
            assert pj[8] == proj[8] + np.float32(2 * j[0]) / np.float32(w) and pj[9] == proj[9] + np.float32(2 * j[1]) / np.float32(h)
            inv_proj = W.mat4_inverse(pj).astype(np.float64).reshape(4, 4).T
            inv_view = W.mat4_inverse(W.camera_view(cam)).astype(np.float64).reshape(4, 4).T
            assert np.allclose(inv_view @ view, np.eye(4), atol=1e-4)
            for (px, py) in ((0, 0), (w - 1, h - 1), (w // 2, h // 3), (17, 301)):
                d = np.array([2 * (px + 0.5) / w - 1, 2 * (py + 0.5) / h - 1, 1.0, 1.0])
                target = inv_proj @ d
                t3 = target[:3] / np.linalg.norm(target[:3])
                ray = (inv_view @ np.array([*t3, 0.0]))[:3]
                ray /= np.linalg.norm(ray)
                u = (2 * (px + 0.5 + j[0]) / w - 1) * float(c["tan_half_fov"]) * float(c["aspect"])
                v = (1 - 2 * (py + 0.5 + j[1]) / h) * float(c["tan_half_fov"])
                b = c["fwd"].astype(np.float64) + c["right"].astype(np.float64) * u + c["up"].astype(np.float64) * v
                b /= np.linalg.norm(b)
                assert np.abs(ray - b).max() < 2e-6, (frame, px, py, ray, b)


def test_kernel_body_and_oracle_agree_with_jitter():
    """The jittered primary ray in the kernel body (host harness) and in the oracle: bit-identical first hits, different from
    the un-jittered frame, and jitter 0 is bit-identical to no jitter."""
    from tests import harness_ffi as H, oracle_ffi as O
    import ctypes as C
    cm = W.ChunkManager(128, 1.0); cm.generate_scene(64); cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo()
    w, h = 96, 64
    cam = W.scene_camera(64, 0, w, h)
    hk = H.HostKernel(pw.nodes, pw.sub_chunks)
    L = H.lib()
    L.hh_set_jitter_clip.argtypes = [C.c_float, C.c_float]
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    plain = hk.trace_primary(cam, w, h)
    try:
        for frame in (0, 3, 10):
            j = W.taa_jitter(frame)
            jc = (float(np.float32(2) * j[0] / np.float32(w)), float(np.float32(2) * j[1] / np.float32(h)))
            L.hh_set_jitter_clip(*jc)
            O.set_jitter_clip(j, w, h)
            got = hk.trace_primary(cam, w, h)
            ref, _ = lat.trace_primary(cam, w, h)
            assert got.tobytes() == ref.tobytes(), frame
            assert got.tobytes() != plain.tobytes(), frame
        L.hh_set_jitter_clip(0.0, 0.0)
        O.set_jitter_clip(np.zeros(2, dtype=np.float32), w, h)
        assert hk.trace_primary(cam, w, h).tobytes() == plain.tobytes()
    finally:
        L.hh_set_jitter_clip(0.0, 0.0)
        O.set_jitter_clip(None)
