import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session", autouse=True)
def built():
    """Everything under test is compiled in-tree once per session (no-op when fresh)."""
    from blok_amd import build as b
    b.build_host()
    b.build_hip()
    b.build_oracle()
    return True


SEED = 0xB10C0001


def make_scene_world(n: int, seed: int = SEED):
    """Synthetic scene G(n, seed) as (packed world, voxel xyz, voxel material ids)."""
    from blok_amd import world as W
    cm = W.ChunkManager(128, 1.0)
    cm.generate_scene(n, seed)
    cm.rebuild_dirty_chunks()
    return cm, cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))


@pytest.fixture(scope="session")
def scene64():
    return make_scene_world(64)


@pytest.fixture(scope="session")
def scene256():
    return make_scene_world(256)


def records_equal(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Per-record bitwise equality of two 16-byte hit record arrays."""
    ab = np.ascontiguousarray(a).view(np.uint8).reshape(-1, 16)
    bb = np.ascontiguousarray(b).view(np.uint8).reshape(-1, 16)
    return (ab == bb).all(axis=1)


def edge_case_rays():
    """Rays the reference's traversal treats specially: axis-aligned (|d| < 1e-6 -> +1e-6, intersect.rint:79),
    negative directions (octant mask, :94-97), origins inside the grid / inside a voxel / on lattice planes,
    grazing edges and corners, short tmax (shadow-ray interval raygen.rgen:294-296)."""
    from tests.oracle_ffi import RAY
    rays = []

    def add(o, d, tmin=0.001, tmax=10000.0):
        d = np.asarray(d, dtype=np.float64)
        nrm = np.linalg.norm(d)
        d = (d / nrm).astype(np.float32) if nrm > 0 else d.astype(np.float32)
        rays.append((tuple(np.float32(o)), np.float32(tmin), tuple(d), np.float32(tmax)))

    for o in [(-10.5, 20.25, 31.5), (70.0, 40.0, 32.0), (32.5, 80.0, 32.5), (32.5, 20.5, -5.0), (32.0, 20.0, 32.0),
              (10.5, 60.5, 10.5), (0.0, 0.0, 0.0), (63.999, 30.0, 63.999), (32.5, 9.5, 32.5)]:
        for d in [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (1, -1, 0),
                  (-1, -1, -1), (1, 1, 1), (1, -1, 1), (1e-7, -1, 1e-7), (-1e-7, -1, 0), (0.5, -0.5, 1e-6),
                  (3, -2, 1), (-3, -2, -1), (1, -0.001, 0), (0, -0.001, 1), (1, -1e-3, 1)]:
            add(o, d)
            add(o, d, 0.001, 5.0)
            add(o, d, 2.0, 1000.0)
    return np.array(rays, dtype=RAY)


def random_rays(n: int, count: int, seed: int):
    from tests.oracle_ffi import RAY
    rng = np.random.default_rng(seed)
    rays = np.zeros(count, dtype=RAY)
    org = rng.uniform(-0.5 * n, 1.5 * n, size=(count, 3))
    inside = rng.random(count) < 0.3
    org[inside] = rng.uniform(0, n, size=(int(inside.sum()), 3))
    tgt = rng.uniform(0, n, size=(count, 3)) * np.array([1, 0.5, 1])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    axis = rng.random(count) < 0.05
    d[axis, rng.integers(0, 3)] = 0.0
    d[axis] /= np.maximum(np.linalg.norm(d[axis], axis=1, keepdims=True), 1e-9)
    rays["org"] = org.astype(np.float32)
    rays["dir"] = d.astype(np.float32)
    rays["tmin"] = 0.001
    rays["tmax"] = 10000.0
    return rays


@pytest.fixture(autouse=True)
def no_hip_error_left_behind(request):
    """After every GPU test: no HIP runtime call may have failed silently (a stale error would surface in an unrelated
    later launch check).  Found a double free this way."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import ctypes
    libs = [line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line]
    if not libs:
        return
    hip = ctypes.CDLL(libs[0])
    hip.hipGetErrorString.restype = ctypes.c_char_p
    err = hip.hipGetLastError()
    assert err == 0, f"a HIP call failed silently during this test: {hip.hipGetErrorString(err).decode()}"
