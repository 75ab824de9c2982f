"""ctypes bindings for tests/host_harness/libhost_harness.so (the kernel body compiled for the CPU) — TESTS ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "host_harness"
LIB = SRC / "libhost_harness.so"
HIT = np.dtype([("t", "<f4"), ("material_id", "<u4"), ("voxel", "<i2", 3), ("face", "u1"), ("hit", "u1")])


def build(sanitize: bool = False, libm_pow: bool = False) -> Path:
    out = SRC / ("libhost_harness_asan.so" if sanitize else ("libhost_harness_libm_pow.so" if libm_pow else "libhost_harness.so"))
    deps = [SRC / "harness.cpp", SRC / "host_harness_shims.h", ROOT / "blok_amd/csrc/hip/trace_core.h",
            ROOT / "blok_amd/csrc/hip/path_core.h", ROOT / "blok_amd/csrc/hip/post_core.h",
            ROOT / "blok_amd/csrc/hip/trace_kernels.h", ROOT / "blok_amd/csrc/hip/tree_build.cpp",
            ROOT / "blok_amd/csrc/hip/tree.h"]
    if out.exists() and all(d.stat().st_mtime <= out.stat().st_mtime for d in deps):
        return out
    cmd = ["g++", "-O1", "-g", "-std=c++20", "-fPIC", "-ffp-contract=off", "-Wall", f"-I{ROOT / 'include'}",
           f"-I{ROOT / 'blok_amd/csrc/hip'}", f"-I{SRC}", "-shared", "-o", os.fspath(out),
           os.fspath(SRC / "harness.cpp"), os.fspath(ROOT / "blok_amd/csrc/hip/tree_build.cpp")]
    if sanitize:
        cmd[1:1] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    if libm_pow:
        cmd[1:1] = ["-DBLOK_PATH_LIBM_POW"]              # pow(x, 5 | 8 | 128) through libm, as the oracle writes them (path_core.h)
    subprocess.run(cmd, check=True)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.fspath(build()))
        L.hh_build.restype = C.c_void_p
        L.hh_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p)]
        L.hh_free.argtypes = [C.c_void_p]
        L.hh_levels.restype = C.c_uint32
        L.hh_levels.argtypes = [C.c_void_p]
        L.hh_voxels.restype = C.c_uint64
        L.hh_voxels.argtypes = [C.c_void_p]
        L.hh_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.hh_trace_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.hh_render_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_uint32] * 6 + [C.c_void_p] * 4
        _lib = L
    return _lib


def render_paths_libm_pow(nodes, subs, cam, materials, width, height, spp=8, max_bounces=2, frame_index=0):
    """The path loop of the kernel body with the shader's three pow() calls through libm (the only operations in which the shipped body
    departs from the oracle's text): on one libm that build must equal the oracle bit for bit, colour included."""
    L = C.CDLL(os.fspath(build(libm_pow=True)))
    L.hh_build.restype = C.c_void_p
    L.hh_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p)]
    L.hh_free.argtypes = [C.c_void_p]
    L.hh_render_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_uint32] * 6 + [C.c_void_p] * 4
    why = C.c_char_p()
    h = L.hh_build(C.c_void_p(nodes.ctypes.data), len(nodes), C.c_void_p(subs.ctypes.data), len(subs), C.byref(why))
    if not h:
        raise RuntimeError(why.value.decode())
    materials = np.ascontiguousarray(materials)
    planes = {k: np.zeros((height, width, 4), dtype=np.float32) for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
    L.hh_render_paths(C.c_void_p(h), C.c_void_p(cam.ctypes.data), C.c_void_p(materials.ctypes.data), len(materials), width, height, spp, max_bounces, frame_index,
                      *[C.c_void_p(planes[k].ctypes.data) for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")])
    L.hh_free(C.c_void_p(h))
    return planes


class HostKernel:
    def __init__(self, nodes: np.ndarray, subs: np.ndarray):
        why = C.c_char_p()
        self.h = lib().hh_build(C.c_void_p(nodes.ctypes.data), len(nodes), C.c_void_p(subs.ctypes.data), len(subs),
                                C.byref(why))
        if not self.h:
            raise RuntimeError(why.value.decode())
        self.h = C.c_void_p(self.h)
        self.levels = lib().hh_levels(self.h)
        self.n_voxels = lib().hh_voxels(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().hh_free(self.h)
            self.h = None

    def trace_rays(self, rays: np.ndarray) -> np.ndarray:
        out = np.zeros(len(rays), dtype=HIT)
        lib().hh_trace_rays(self.h, C.c_void_p(rays.ctypes.data), len(rays), C.c_void_p(out.ctypes.data))
        return out

    def trace_primary(self, cam: np.ndarray, width: int, height: int) -> np.ndarray:
        out = np.zeros(width * height, dtype=HIT)
        lib().hh_trace_primary(self.h, C.c_void_p(cam.ctypes.data), width, height, C.c_void_p(out.ctypes.data))
        return out


    def render_paths(self, cam, materials, width, height, spp=8, max_bounces=2, frame_index=0):
        materials = np.ascontiguousarray(materials)
        planes = {k: np.zeros((height, width, 4), dtype=np.float32)
                  for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
        lib().hh_render_paths(self.h, C.c_void_p(cam.ctypes.data), C.c_void_p(materials.ctypes.data), len(materials),
                              width, height, spp, max_bounces, frame_index,
                              *[C.c_void_p(planes[k].ctypes.data) for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")])
        return planes
