"""ctypes bindings for oracle/liboracle.so and oracle/_ref/libref_morton.so — TESTS ONLY."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_LIB = ROOT / "oracle" / "liboracle.so"
REF_MORTON_LIB = ROOT / "oracle" / "_ref" / "libref_morton.so"

HIT = np.dtype([("t", "<f4"), ("material_id", "<u4"), ("voxel", "<i2", 3), ("face", "u1"), ("hit", "u1")])
RAY = np.dtype([("org", "<f4", 3), ("tmin", "<f4"), ("dir", "<f4", 3), ("tmax", "<f4")])
COUNTERS = np.dtype([(k, "<u8") for k in ("rays", "hits", "sub_chunks_entered", "nodes_fetched",
                                           "iter_limit_hits", "stack_limit_hits", "max_stack", "max_iter", "ties")])
SVO_NODE = np.dtype([("child_mask", "<u4"), ("first_child", "<u4"), ("material_id", "<u4"), ("occupancy", "<f4")])
SUB_CHUNK = np.dtype([("node_offset", "<u4"), ("root_node_index", "<u4"), ("node_count", "<u4"),
                      ("start_depth", "<u4"), ("world_min", "<f4", 3), ("sub_chunk_size", "<f4"),
                      ("world_max", "<f4", 3), ("pad0", "<f4")])

_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not ORACLE_LIB.exists():
            raise RuntimeError(f"{ORACLE_LIB} not built: run `make -C oracle`")
        L = C.CDLL(os.fspath(ORACLE_LIB))
        L.orc_morton_encode.restype = C.c_uint64
        L.orc_morton_encode.argtypes = [C.c_int32] * 3
        L.orc_morton_decode.restype = None
        L.orc_morton_decode.argtypes = [C.c_uint64] + [C.POINTER(C.c_int32)] * 3
        L.orc_morton_octant.restype = C.c_uint32
        L.orc_morton_octant.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_world_new.restype = C.c_void_p
        L.orc_world_new.argtypes = [C.c_uint32, C.c_float]
        L.orc_world_free.argtypes = [C.c_void_p]
        L.orc_world_set_voxel.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_float]
        L.orc_world_set_voxels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_world_get_voxel_material.restype = C.c_uint32
        L.orc_world_get_voxel_material.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_world_apply_brush.argtypes = [C.c_void_p] + [C.c_float] * 5 + [C.c_int]
        L.orc_world_rebuild.restype = C.c_int
        L.orc_world_rebuild.argtypes = [C.c_void_p, C.c_int]
        L.orc_world_pack.argtypes = [C.c_void_p]
        for name in ("orc_world_n_nodes", "orc_world_n_subs", "orc_world_n_chunks"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("orc_world_nodes", "orc_world_subs"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_world_chunk_info.restype = C.c_int
        L.orc_world_chunk_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.orc_world_chunk_nodes.restype = C.c_void_p
        L.orc_world_chunk_nodes.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_world_find_leaf.restype = C.c_int64
        L.orc_world_find_leaf.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_primary_rays.argtypes = [C.c_void_p] + [C.c_uint32] * 7 + [C.c_void_p]
        L.orc_trace_bruteforce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p]
        L.orc_lattice_build.restype = C.c_void_p
        L.orc_lattice_build.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_lattice_free.argtypes = [C.c_void_p]
        L.orc_trace_lattice.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trace_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + \
                                       [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trace_voxels_bruteforce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                  C.c_void_p, C.c_void_p]
        L.orc_render_paths.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 10 + [C.c_void_p] * 5 + [C.c_int]
        L.orc_vox_load.restype = C.c_void_p
        L.orc_vox_load.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_vox_free.argtypes = [C.c_void_p]
        L.orc_vox_n_models.restype = C.c_uint32
        L.orc_vox_n_models.argtypes = [C.c_void_p]
        L.orc_vox_model_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_vox_model_voxels.restype = C.c_void_p
        L.orc_vox_model_voxels.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_vox_palette.restype = C.c_void_p
        L.orc_vox_palette.argtypes = [C.c_void_p]
        L.orc_default_palette.argtypes = [C.c_void_p]
        L.orc_matlib_new.restype = C.c_void_p
        L.orc_matlib_free.argtypes = [C.c_void_p]
        L.orc_matlib_size.restype = C.c_uint32
        L.orc_matlib_size.argtypes = [C.c_void_p]
        L.orc_matlib_from_color.restype = C.c_uint32
        L.orc_matlib_from_color.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint8]
        L.orc_matlib_pack.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_vox_import_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_vox_import_to_world.restype = C.c_uint32
        L.orc_vox_import_to_world.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint32]
        L.orc_shade_surface.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.orc_sizeof_counters.restype = C.c_uint32
        assert L.orc_sizeof_counters() == COUNTERS.itemsize
        _lib = L
    return _lib


def ref_morton():
    """The reference's own morton.hpp, compiled in place (None when oracle/_ref is absent)."""
    global _ref
    if _ref is None and REF_MORTON_LIB.exists():
        R = C.CDLL(os.fspath(REF_MORTON_LIB))
        R.ref_morton_encode.restype = C.c_uint64
        R.ref_morton_encode.argtypes = [C.c_int] * 3
        R.ref_morton_decode.restype = None
        R.ref_morton_decode.argtypes = [C.c_uint64] + [C.POINTER(C.c_int)] * 3
        R.ref_morton_octant.restype = C.c_uint
        R.ref_morton_octant.argtypes = [C.c_uint64, C.c_uint, C.c_uint]
        _ref = R
    return _ref


def _p(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def _copy(address, count, dtype):
    if not count:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(address)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class OracleWorld:
    """Literal ChunkManager restatement (dense per-chunk store)."""

    def __init__(self, chunk_size=128, voxel_size=1.0):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_world_new(chunk_size, voxel_size))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_world_free(self.h)
            self.h = None

    def set_voxel(self, p, mat, density=1.0):
        self.L.orc_world_set_voxel(self.h, p[0], p[1], p[2], mat, density)

    def set_voxels(self, xyz, mats):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32)
        mats = np.ascontiguousarray(mats, dtype=np.uint32)
        self.L.orc_world_set_voxels(self.h, _p(xyz), _p(mats), len(mats))

    def get_voxel_material(self, p):
        return int(self.L.orc_world_get_voxel_material(self.h, p[0], p[1], p[2]))

    def apply_brush(self, center, radius, value, mode="add"):
        self.L.orc_world_apply_brush(self.h, center[0], center[1], center[2], radius, value, {"add": 0, "subtract": 1}[mode])

    def rebuild(self, max_per_frame=1 << 30):
        return self.L.orc_world_rebuild(self.h, max_per_frame)

    def pack(self):
        self.L.orc_world_pack(self.h)
        nodes = _copy(self.L.orc_world_nodes(self.h), self.L.orc_world_n_nodes(self.h), SVO_NODE)
        subs = _copy(self.L.orc_world_subs(self.h), self.L.orc_world_n_subs(self.h), SUB_CHUNK)
        return nodes, subs

    def n_chunks(self):
        return self.L.orc_world_n_chunks(self.h)

    def chunk(self, i):
        coord = (C.c_int32 * 3)()
        n = C.c_uint64()
        assert self.L.orc_world_chunk_info(self.h, i, coord, C.byref(n)) == 0
        return tuple(coord), _copy(self.L.orc_world_chunk_nodes(self.h, i), n.value, SVO_NODE)

    def find_leaf(self, i, x, y, z):
        return int(self.L.orc_world_find_leaf(self.h, i, x, y, z))

    def chunk_dense(self, i, chunk_size=128):
        """(density, material ids) of chunk i as [z][y][x] arrays (copies)."""
        d, m = C.c_void_p(), C.c_void_p()
        self.L.orc_world_chunk_dense.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        assert self.L.orc_world_chunk_dense(self.h, i, C.byref(d), C.byref(m)) == 0
        n = chunk_size ** 3
        shape = (chunk_size,) * 3
        return (_copy(d.value, n, np.dtype("<f4")).reshape(shape), _copy(m.value, n, np.dtype("<u4")).reshape(shape))


def primary_rays(cam: np.ndarray, width, height, x0=0, y0=0, w=None, h=None, stride=1) -> np.ndarray:
    w = width if w is None else w
    h = height if h is None else h
    n = ((w + stride - 1) // stride) * ((h + stride - 1) // stride)
    rays = np.zeros(n, dtype=RAY)
    lib().orc_primary_rays(_p(cam), width, height, x0, y0, w, h, stride, _p(rays))
    return rays


def trace_bruteforce(nodes, subs, rays):
    hits = np.zeros(len(rays), dtype=HIT)
    ctr = np.zeros(1, dtype=COUNTERS)
    lib().orc_trace_bruteforce(_p(nodes), _p(subs), len(subs), _p(rays), len(rays), _p(hits), _p(ctr))
    return hits, ctr[0]


def set_jitter_clip(jitter_px=None, width=1, height=1):
    """TAA jitter of the oracle's primary rays (pixels -> clip space as getJitterClipSpace does); None = off."""
    L = lib()
    L.orc_set_jitter_clip.argtypes = [C.c_float, C.c_float]
    if jitter_px is None:
        L.orc_set_jitter_clip(0.0, 0.0)
    else:
        L.orc_set_jitter_clip(float(np.float32(2.0) * np.float32(jitter_px[0]) / np.float32(width)),
                              float(np.float32(2.0) * np.float32(jitter_px[1]) / np.float32(height)))


class Lattice:
    def __init__(self, nodes, subs):
        self.nodes = np.ascontiguousarray(nodes)
        self.subs = np.ascontiguousarray(subs)
        self.h = C.c_void_p(lib().orc_lattice_build(_p(self.subs), len(self.subs)))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_lattice_free(self.h)
            self.h = None

    def trace(self, rays, threads=1):
        hits = np.zeros(len(rays), dtype=HIT)
        ctr = np.zeros(1, dtype=COUNTERS)
        lib().orc_trace_lattice(self.h, _p(self.nodes), _p(self.subs), _p(rays), len(rays), _p(hits), _p(ctr), threads)
        return hits, ctr[0]

    def trace_primary(self, cam, width, height, x0=0, y0=0, w=None, h=None, stride=1, threads=1, want_hits=True):
        w = width if w is None else w
        h = height if h is None else h
        n = ((w + stride - 1) // stride) * ((h + stride - 1) // stride)
        hits = np.zeros(n, dtype=HIT) if want_hits else None
        ctr = np.zeros(1, dtype=COUNTERS)
        lib().orc_trace_primary(self.h, _p(self.nodes), _p(self.subs), _p(cam), width, height, x0, y0, w, h,
                                stride, _p(hits) if want_hits else None, _p(ctr), threads)
        return hits, ctr[0]


class NativeLattice:
    """The timing build of the traversal (`make -C oracle native`: counters compiled out, -O3 -march=native, built on the
    machine that runs it, into a scratch directory) — bench.py's cpu_baseline only.  Same entry points as Lattice."""

    def __init__(self, nodes, subs, out_dir=None):
        import subprocess
        import tempfile
        self.dir = out_dir or tempfile.mkdtemp(prefix="blok_oracle_native_")
        subprocess.run(["make", "-s", "-C", os.fspath(ROOT / "oracle"), "native", f"NATIVE_OUT={self.dir}"], check=True,
                       capture_output=True)
        self.L = C.CDLL(os.path.join(self.dir, "liboracle_native.so"))
        self.L.orc_lattice_build.restype = C.c_void_p
        self.L.orc_lattice_build.argtypes = [C.c_void_p, C.c_size_t]
        self.L.orc_lattice_free.argtypes = [C.c_void_p]
        self.L.orc_trace_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + \
                                            [C.c_void_p, C.c_void_p, C.c_int]
        self.nodes = np.ascontiguousarray(nodes)
        self.subs = np.ascontiguousarray(subs)
        self.h = C.c_void_p(self.L.orc_lattice_build(_p(self.subs), len(self.subs)))

    def trace_primary(self, cam, width, height, x0=0, y0=0, w=None, h=None, stride=1, threads=1, want_hits=True):
        w = width if w is None else w
        h = height if h is None else h
        n = ((w + stride - 1) // stride) * ((h + stride - 1) // stride)
        hits = np.zeros(n, dtype=HIT) if want_hits else None
        ctr = np.zeros(1, dtype=COUNTERS)
        self.L.orc_trace_primary(self.h, _p(self.nodes), _p(self.subs), _p(cam), width, height, x0, y0, w, h,
                                 stride, _p(hits) if want_hits else None, _p(ctr), threads)
        return hits, ctr[0]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_lattice_free(self.h)
            self.h = None


def render_paths(lattice: "Lattice", materials, cam, width, height, spp=8, max_bounces=2, frame_index=0, rect=None,
                 stride=1, threads=1):
    """raygen.rgen restatement: dict of (rows, cols, 4) float32 planes + counters."""
    x0, y0, w, h = rect if rect is not None else (0, 0, width, height)
    cols, rows = (w + stride - 1) // stride, (h + stride - 1) // stride
    materials = np.ascontiguousarray(materials)
    planes = {k: np.zeros((rows, cols, 4), dtype=np.float32) for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
    ctr = np.zeros(1, dtype=COUNTERS)
    lib().orc_render_paths(lattice.h, _p(lattice.nodes), _p(lattice.subs), _p(materials), _p(cam), width, height,
                           x0, y0, w, h, stride, spp, max_bounces, frame_index, _p(planes["color"]),
                           _p(planes["world_pos"]), _p(planes["normal_roughness"]), _p(planes["albedo_metallic"]),
                           _p(ctr), threads)
    return planes, ctr[0]


def accumulate(accum, color):
    """In place: accum += color (w += 1); returns the ACES + gamma RGBA8 image of the running average."""
    n = accum.size // 4
    out = np.zeros(accum.shape[:-1], dtype=np.uint32)
    L = lib()
    L.orc_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_accumulate(_p(accum), _p(np.ascontiguousarray(color, dtype=np.float32)), n, _p(out))
    return out


def tonemap(hdr, exposure=1.0, saturation_boost=1.15, operator=1):
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    out = np.zeros(hdr.shape[:-1], dtype=np.uint32)
    L = lib()
    L.orc_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.orc_tonemap(_p(hdr), out.size, exposure, saturation_boost, operator, _p(out))
    return out


def trace_voxels_bruteforce(xyz, mats, rays):
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    mats = np.ascontiguousarray(mats, dtype=np.uint32)
    hits = np.zeros(len(rays), dtype=HIT)
    ctr = np.zeros(1, dtype=COUNTERS)
    lib().orc_trace_voxels_bruteforce(_p(xyz), _p(mats), len(mats), _p(rays), len(rays), _p(hits), _p(ctr))
    return hits, ctr[0]


def shade_surface(hit: np.ndarray, materials: np.ndarray) -> np.ndarray:
    out = (C.c_float * 12)()
    lib().orc_shade_surface(_p(hit), _p(materials), out)
    return np.array(out[:], dtype=np.float32)


MATERIAL = np.dtype([("albedo", "<f4", 3), ("flags", "<u4"), ("emission", "<f4", 3), ("ior", "<f4")])


class OracleVox:
    def __init__(self, data: bytes):
        buf = np.frombuffer(data, dtype=np.uint8)
        err = C.create_string_buffer(256)
        h = lib().orc_vox_load(_p(buf) if len(buf) else None, len(buf), err, len(err))
        self.error = err.value.decode()
        self.h = C.c_void_p(h) if h else None

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_vox_free(self.h)
            self.h = None

    def n_models(self):
        return lib().orc_vox_n_models(self.h)

    def model(self, i):
        size = (C.c_uint32 * 3)()
        n = C.c_uint32()
        lib().orc_vox_model_info(self.h, i, size, C.byref(n))
        return tuple(size), _copy(lib().orc_vox_model_voxels(self.h, i), n.value * 4, np.dtype("u1")).reshape(-1, 4)

    def palette(self):
        return _copy(lib().orc_vox_palette(self.h), 256, np.dtype("<u4"))


class OracleMaterialLibrary:
    def __init__(self):
        self.h = C.c_void_p(lib().orc_matlib_new())

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_matlib_free(self.h)
            self.h = None

    def __len__(self):
        return lib().orc_matlib_size(self.h)

    def from_color(self, r, g, b):
        return lib().orc_matlib_from_color(self.h, r, g, b)

    def pack(self):
        out = np.zeros(len(self), dtype=MATERIAL)
        lib().orc_matlib_pack(self.h, _p(out))
        return out

    def import_vox(self, vox: OracleVox):
        mapping = np.zeros(256, dtype=np.uint32)
        lib().orc_vox_import_materials(vox.h, self.h, _p(mapping))
        return mapping


def vox_import_to_world(vox: OracleVox, world: "OracleWorld", matlib, offset=(0.0, 0.0, 0.0), model=0):
    return lib().orc_vox_import_to_world(vox.h, world.h, matlib.h if matlib is not None else None,
                                         offset[0], offset[1], offset[2], model)


def default_palette():
    out = np.zeros(256, dtype=np.uint32)
    lib().orc_default_palette(_p(out))
    return out


# ---------------------------------------------------------------- image-space chain (oracle/blok_oracle_post.cpp)
class OrcDenoiseSettings(C.Structure):
    _fields_ = [("temporalAlpha", C.c_float), ("momentAlpha", C.c_float), ("varianceClipGamma", C.c_float),
                ("depthThreshold", C.c_float), ("normalThreshold", C.c_float), ("phiColor", C.c_float), ("phiNormal", C.c_float),
                ("phiDepth", C.c_float), ("atrousIterations", C.c_int), ("varianceBoost", C.c_float), ("minHistoryLength", C.c_int)]

    @classmethod
    def default(cls):                                         # renderer_denoising.hpp:49-66
        return cls(0.05, 0.2, 1.5, 0.1, 0.95, 0.5, 128.0, 0.1, 4, 1.5, 4)


def q16(x):
    L = lib()
    L.orc_q16.restype = C.c_float; L.orc_q16.argtypes = [C.c_float]
    return L.orc_q16(float(x))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleDenoiser:
    """Denoiser::denoise + copyCurrentGeometryToHistory + swapHistoryBuffers over the oracle's pass functions
    (renderer_denoising.cpp:714-776, 833-866, 690-697), and PostProcess's TAA history (renderer_postprocess.cpp:526-534)."""

    def __init__(self, w, h, settings=None):
        self.w, self.h = w, h
        self.S = settings or OrcDenoiseSettings.default()
        z = lambda c: np.zeros((h, w, c), np.float32) if c > 1 else np.zeros((h, w), np.float32)
        self.prev = dict(color=z(4), moments=z(2), hist_len=z(1), world_pos=z(4), normals=z(4))
        self.taa_hist = z(4)
        self.variance = z(1); self.motion = z(2)

    def denoise(self, color, world_pos, normal_roughness, prev_view_proj, frame_count, motion=None):
        L, w, h = lib(), self.w, self.h
        color, world_pos, normal_roughness = _f32(color), _f32(world_pos), _f32(normal_roughness)
        M = _f32(prev_view_proj).reshape(-1)
        if motion is None:
            motion = np.zeros((h, w, 2), np.float32)
            L.orc_motion_vectors(_p(world_pos), w, h, _p(M), _p(motion))
        self.motion = _f32(motion)
        out_c, out_m, out_l = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 2), np.float32), np.zeros((h, w), np.float32)
        P = self.prev
        L.orc_temporal(_p(color), _p(world_pos), _p(normal_roughness), _p(self.motion), _p(P["color"]), _p(P["moments"]),
                       _p(P["hist_len"]), _p(P["world_pos"]), _p(P["normals"]), w, h, C.c_uint32(frame_count), _p(M),
                       C.byref(self.S), _p(out_c), _p(out_m), _p(out_l))
        var = np.zeros((h, w), np.float32)
        L.orc_variance(_p(out_c), _p(out_m), _p(out_l), _p(world_pos), _p(normal_roughness), w, h, C.byref(self.S), _p(var))
        cur = out_c
        for it in range(self.S.atrousIterations):
            nxt = np.zeros((h, w, 4), np.float32)
            L.orc_atrous(_p(cur), _p(var), _p(world_pos), _p(normal_roughness), w, h, 1 << it, C.c_float(self.S.phiColor),
                         C.c_float(self.S.phiNormal), C.c_float(self.S.phiDepth), _p(nxt))
            cur = nxt
        self.prev = dict(color=out_c, moments=out_m, hist_len=out_l, world_pos=world_pos.copy(), normals=normal_roughness.copy())
        self.variance = var
        return cur

    def taa(self, color, frame_count, feedback_min=0.93, feedback_max=0.98, motion=None):
        L, w, h = lib(), self.w, self.h
        color = _f32(color)
        motion = self.motion if motion is None else _f32(motion)
        out, hist = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
        L.orc_taa(_p(color), _p(self.taa_hist), _p(motion), w, h, C.c_uint32(frame_count), C.c_float(feedback_min),
                  C.c_float(feedback_max), _p(out), _p(hist))
        self.taa_hist = hist
        return out


def sharpen(rgba8, strength=0.5):
    rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint32)
    h, w = rgba8.shape
    out = np.zeros((h, w), np.uint32)
    lib().orc_sharpen(_p(rgba8), w, h, C.c_float(strength), _p(out))
    return out
