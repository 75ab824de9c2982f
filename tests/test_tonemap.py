"""The tonemap pass of the reference's post-process chain (assets/shaders/tonemap.comp): kernel body on the CPU and
HIP kernel on the GPU against the oracle.  The Khronos operator is pure + - * / sqrt, so RGBA8 output is bit-exact;
the soft-clip operator uses exp() and may differ by one code value on the GPU."""
import ctypes as C

import numpy as np
import pytest

from tests import harness_ffi as H
from tests import oracle_ffi as O


def hdr_samples():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 1.2, size=(20000, 3)), rng.uniform(0, 30, size=(5000, 3)),
                        rng.uniform(0, 0.1, size=(3000, 3)), np.zeros((10, 3)), np.full((5, 3), 0.76),
                        np.array([[5, 0.01, 0.01], [0.079, 0.5, 0.9], [0.08, 0.08, 0.08], [100, 100, 100]])]).astype(np.float32)
    return np.concatenate([x, np.ones((len(x), 1), np.float32)], axis=1)


@pytest.mark.parametrize("exposure,boost,op", [(1.0, 1.15, 1), (2.5, 1.0, 1), (0.7, 0.5, 1), (1.0, 2.0, 1), (1.0, 1.15, 0)])
def test_kernel_body_matches_oracle_on_cpu(exposure, boost, op):
    hdr = hdr_samples()
    want = O.tonemap(hdr, exposure, boost, op)
    got = np.zeros(len(hdr), dtype=np.uint32)
    L = H.lib()
    L.hh_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.hh_tonemap(C.c_void_p(hdr.ctypes.data), len(hdr), exposure, boost, op, C.c_void_p(got.ctypes.data))
    assert np.array_equal(got, want)
    assert (want >> 24 == 0xFF).all()


def test_khronos_operator_reference_points():
    """Below the compression start the operator only subtracts the toe offset; greys stay grey; output is clamped."""
    hdr = np.array([[0.5, 0.5, 0.5, 1], [0.0, 0.0, 0.0, 1], [50, 50, 50, 1], [0.2, 0.4, 0.6, 1]], dtype=np.float32)
    out = O.tonemap(hdr, 1.0, 1.0, 1)
    px = np.stack([out & 255, (out >> 8) & 255, (out >> 16) & 255], axis=1)
    assert tuple(px[0]) == (117, 117, 117)          # 0.5 - 0.04 = 0.46 -> round(0.46 * 255)
    assert tuple(px[1]) == (0, 0, 0)
    assert px[2][0] == px[2][1] == px[2][2] >= 250
    assert tuple(px[3]) == tuple(int((v - 0.04) * 255 + 0.5) for v in (0.2, 0.4, 0.6))


@pytest.mark.gpu
def test_gpu_tonemap_matches_oracle():
    from blok_amd.tracer import HipTracer
    tr = HipTracer(16, 16).init()
    hdr = hdr_samples()
    for exposure, boost, op in [(1.0, 1.15, 1), (2.5, 1.0, 1), (0.7, 0.5, 1), (1.0, 2.0, 1)]:
        assert np.array_equal(tr.tonemap(hdr, exposure, boost, op), O.tonemap(hdr, exposure, boost, op))
    got, want = tr.tonemap(hdr, 1.0, 1.15, 0), O.tonemap(hdr, 1.0, 1.15, 0)
    diff = np.abs(np.stack([(got >> s) & 255 for s in (0, 8, 16)]).astype(int) - np.stack([(want >> s) & 255 for s in (0, 8, 16)]).astype(int))
    assert diff.max() <= 1
    tr.shutdown()
