"""CPU: the C-ABI libraries load, export every symbol the headers declare, and fail loudly without a GPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from blok_amd import _ffi

ROOT = Path(__file__).resolve().parent.parent


def declared_functions(header: Path):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(blok_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    """Both headers: blok_hip.h is the drop-in surface (what a reference-side binding includes), blok_hip_debug.h the diagnostics,
    test hooks and tuning knobs beside it.  Every declared name is exported, every binding is declared, nothing is declared twice."""
    names = declared_functions(ROOT / "include" / "blok_hip.h")
    debug = declared_functions(ROOT / "include" / "blok_hip_debug.h")
    assert len(names) >= 18 and len(debug) >= 6
    assert not set(names) & set(debug), "declared in both headers"
    lib = C.CDLL(str(_ffi.HIP_LIB))
    for n in names + debug:
        assert hasattr(lib, n), f"libblok_hip.so does not export {n}"
    assert set(names) | set(debug) == set(_ffi.HIP_SYMBOLS), "python bindings out of sync with include/blok_hip.h + blok_hip_debug.h"


def test_drop_in_header_carries_no_hooks_or_launch_knobs():
    """the surface a maintainer links against holds lifecycle, world, trace, paths, post, volume, multi — no test hook, no
    experiment switch.  (reference surface: blok/include/cuda_tracer.hpp:23-58, blok/include/renderer.hpp:29-77)"""
    names = declared_functions(ROOT / "include" / "blok_hip.h")
    for n in names:
        assert "debug" not in n, n
    for knob in ("blok_hip_set_fused", "blok_hip_set_beam_budget", "blok_hip_set_miss_writer", "blok_hip_set_list_classes", "blok_hip_set_joint_prefix_limit",
                 "blok_hip_beam_prepass", "blok_hip_trace_wave_tiles_device", "blok_hip_set_tile_ordering", "blok_hip_set_moving_order"):
        assert knob not in names, knob
    mirror = (ROOT / "include" / "blok" / "hip_tracer.hpp").read_text()
    assert "blok_hip_debug.h" not in mirror, "the C++ mirror must build against the drop-in header alone"


def test_host_library_exports_every_declared_symbol():
    names = declared_functions(ROOT / "include" / "blok_world.h")
    lib = C.CDLL(str(_ffi.HOST_LIB))
    for n in names:
        assert hasattr(lib, n), f"libblok_host.so does not export {n}"
    assert set(names) == set(_ffi.HOST_SYMBOLS)


def test_record_sizes_match_reference_layouts():
    assert _ffi.SVO_NODE.itemsize == 16      # svo.hpp:23-28
    assert _ffi.SUB_CHUNK.itemsize == 48     # resources.hpp:184 static_assert
    assert _ffi.MATERIAL.itemsize == 32      # material.hpp:114 static_assert
    assert _ffi.HIT.itemsize == 16
    assert _ffi.hip_lib().blok_hip_abi_version() >> 16 == 1


def test_no_device_is_a_loud_error_not_a_fallback():
    """In the GPU-less container create() must fail with BLOK_ERR_NO_DEVICE; on a GPU box it succeeds."""
    import torch
    from blok_amd.tracer import HipTracer
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_ffi.BlokError) as e:
        HipTracer(64, 64).init()
    assert e.value.status == -2


def test_argument_validation_without_device():
    lib = _ffi.hip_lib()
    assert lib.blok_hip_create(None, 0, 16, 16) == -1
    ctx = C.c_void_p()
    assert lib.blok_hip_create(C.byref(ctx), 0, 0, 16) == -1
    assert b"zero" in lib.blok_hip_last_error(None)
    assert lib.blok_hip_tiles_for_rank(3840, 2160, 32, 0, 8) == 1020
    assert sum(lib.blok_hip_tiles_for_rank(3840, 2160, 32, r, 8) for r in range(8)) == 120 * 68
    assert sum(lib.blok_hip_tiles_for_rank(100, 70, 32, r, 3) for r in range(3)) == 4 * 3
    assert lib.blok_hip_tiles_for_rank(100, 70, 32, 3, 3) == 0


def test_product_does_not_link_or_import_the_oracle():
    """The oracle is test infrastructure: nothing under blok_amd/ or bench's product path may reference it."""
    for path in (ROOT / "blok_amd").rglob("*"):
        if path.suffix in {".py", ".cpp", ".hip", ".h"}:
            text = path.read_text()
            assert "oracle" not in text.lower() or path.name == "build.py", f"{path} mentions the oracle"
    import subprocess
    for so in (_ffi.HIP_LIB, _ffi.HOST_LIB):
        out = subprocess.run(["ldd", str(so)], capture_output=True, text=True).stdout
        assert "oracle" not in out


def test_headers_compile_as_c11_and_the_mirror_as_cxx20(tmp_path):
    """The boundary is a C ABI: both headers must be consumable from plain C (the reference-side binding could be C, cgo, JNI ...)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "blok_hip.h"\n#include "blok_hip_debug.h"\n#include "blok_world.h"\nint main(void) { return (int)sizeof(blok_hit) - 16; }\n')
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT / 'include'}", "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cxx = tmp_path / "mirror.cpp"
    cxx.write_text("#include <blok/hip_tracer.hpp>\nint main() { return 0; }\n")
    r = subprocess.run(["g++", "-std=c++20", "-Wall", "-Wextra", f"-I{ROOT / 'include'}", "-fsyntax-only", str(cxx)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
