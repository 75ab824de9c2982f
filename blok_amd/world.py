"""Host voxel data model — thin mirror of blok::ChunkManager / packChunksToGpuSvo over libblok_host.

Method names follow the reference (blok/include/chunk_manager.hpp:26-52): ``set_voxel_material``,
``get_voxel_material``, ``rebuild_dirty_chunks``, ``pack_chunks_to_gpu_svo``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import BlokError, CAMERA, MATERIAL, SUB_CHUNK, SVO_NODE


def morton_encode(x: int, y: int, z: int) -> int:
    return int(_ffi.host_lib().blok_morton_encode(x, y, z))


def morton_decode(code: int):
    x, y, z = C.c_int32(), C.c_int32(), C.c_int32()
    _ffi.host_lib().blok_morton_decode(code, C.byref(x), C.byref(y), C.byref(z))
    return x.value, y.value, z.value


def morton_octant(code: int, max_depth: int, level: int) -> int:
    return int(_ffi.host_lib().blok_morton_octant(code, max_depth, level))


class PackedWorld:
    """The three host arrays of WorldSvoGpu (reference blok/include/resources.hpp:195-203)."""

    def __init__(self, nodes: np.ndarray, sub_chunks: np.ndarray, materials: np.ndarray):
        self.nodes = np.ascontiguousarray(nodes, dtype=SVO_NODE)
        self.sub_chunks = np.ascontiguousarray(sub_chunks, dtype=SUB_CHUNK)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL)


class ChunkManager:
    def __init__(self, chunk_size: int = 128, voxel_size: float = 1.0):
        self._lib = _ffi.host_lib()
        h = C.c_void_p()
        rc = self._lib.blok_world_create(C.byref(h), chunk_size, voxel_size)
        if rc != 0:
            raise BlokError(rc, f"blok_world_create({chunk_size}, {voxel_size})")
        self._h = h
        self.C = chunk_size
        self.voxel_size = voxel_size

    def close(self):
        if getattr(self, "_h", None):
            self._lib.blok_world_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc: int):
        if rc < 0:
            raise BlokError(rc, self._lib.blok_world_last_error(self._h).decode())
        return rc

    def set_voxel_material(self, world_pos, material_id: int, density: float = 1.0):
        p = (C.c_float * 3)(*world_pos)
        self._check(self._lib.blok_world_set_voxel(self._h, p, material_id, density))

    def set_voxels(self, xyz: np.ndarray, material_ids: np.ndarray):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32).reshape(-1, 3)
        mats = np.ascontiguousarray(material_ids, dtype=np.uint32)
        assert len(xyz) == len(mats)
        self._check(self._lib.blok_world_set_voxels(self._h, _ffi.ptr(xyz), _ffi.ptr(mats), len(mats)))

    def get_voxel_material(self, world_pos) -> int:
        p = (C.c_float * 3)(*world_pos)
        return int(self._lib.blok_world_get_voxel_material(self._h, p))

    def apply_brush(self, center, radius: float, value: float, mode: str = "add"):
        """= applyBrush (reference blok/src/brush.cpp:13-63); mode "add" | "subtract"."""
        c = (C.c_float * 3)(*center)
        self._check(self._lib.blok_world_apply_brush(self._h, c, radius, value, {"add": 0, "subtract": 1}[mode]))

    def rebuild_dirty_chunks(self, max_per_frame: int = 1 << 30) -> int:
        return self._check(self._lib.blok_world_rebuild_dirty(self._h, max_per_frame))

    def pack_chunks_to_gpu_svo(self, materials: np.ndarray | None = None) -> PackedWorld:
        self._check(self._lib.blok_world_pack(self._h))
        nodes = _ffi.as_array(self._lib.blok_world_nodes(self._h), self._lib.blok_world_node_count(self._h), SVO_NODE)
        subs = _ffi.as_array(self._lib.blok_world_sub_chunks(self._h),
                             self._lib.blok_world_sub_chunk_count(self._h), SUB_CHUNK)
        if materials is None:
            materials = np.zeros(1, dtype=MATERIAL)
        return PackedWorld(nodes, subs, materials)

    # per-chunk views for parity tests
    def chunk_count(self) -> int:
        return int(self._lib.blok_world_chunk_count(self._h))

    def chunk(self, i: int):
        coord = (C.c_int32 * 3)()
        n = C.c_uint64()
        self._check(self._lib.blok_world_chunk_info(self._h, i, coord, C.byref(n)))
        nodes = _ffi.as_array(self._lib.blok_world_chunk_nodes(self._h, i), n.value, SVO_NODE)
        return tuple(coord), nodes

    def find_leaf(self, i: int, x: int, y: int, z: int) -> int:
        return int(self._lib.blok_world_find_leaf(self._h, i, x, y, z))

    # synthetic benchmark scene
    def generate_scene(self, n: int, seed: int = 0xB10C0001) -> int:
        count = C.c_uint64()
        self._check(self._lib.blok_scene_generate(self._h, n, seed, C.byref(count)))
        return count.value


def scene_dense(n: int, seed: int = 0xB10C0001) -> np.ndarray:
    ids = np.zeros(n * n * n, dtype=np.uint32)
    rc = _ffi.host_lib().blok_scene_generate_dense(n, seed, _ffi.ptr(ids), None)
    if rc != 0:
        raise BlokError(rc, "blok_scene_generate_dense")
    return ids.reshape(n, n, n)  # [z][y][x]


def scene_materials(seed: int = 0xB10C0001) -> np.ndarray:
    mats = np.zeros(256, dtype=MATERIAL)
    rc = _ffi.host_lib().blok_scene_materials(seed, _ffi.ptr(mats))
    if rc != 0:
        raise BlokError(rc, "blok_scene_materials")
    return mats


def scene_camera(n: int, pose: int, width: int, height: int, seed: int = 0xB10C0001) -> np.ndarray:
    cam = np.zeros(1, dtype=CAMERA)
    rc = _ffi.host_lib().blok_scene_camera(n, seed, pose, width, height, _ffi.ptr(cam))
    if rc != 0:
        raise BlokError(rc, "blok_scene_camera")
    return cam


def camera_from_yaw_pitch(pos, yaw_deg, pitch_deg, fov_deg, width, height) -> np.ndarray:
    cam = np.zeros(1, dtype=CAMERA)
    p = (C.c_float * 3)(*pos)
    rc = _ffi.host_lib().blok_camera_from_yaw_pitch(p, yaw_deg, pitch_deg, fov_deg, width, height, _ffi.ptr(cam))
    if rc != 0:
        raise BlokError(rc, "blok_camera_from_yaw_pitch")
    return cam


def camera_look_at(pos, target, fov_deg, width, height) -> np.ndarray:
    cam = np.zeros(1, dtype=CAMERA)
    p = (C.c_float * 3)(*pos)
    t = (C.c_float * 3)(*target)
    rc = _ffi.host_lib().blok_camera_look_at(p, t, fov_deg, width, height, _ffi.ptr(cam))
    if rc != 0:
        raise BlokError(rc, "blok_camera_look_at")
    return cam


def view_proj_from_camera(cam: np.ndarray) -> np.ndarray:
    """Column-major 4x4 (GLM layout) M with uv = ndc.xy * 0.5 + 0.5, ndc = (M @ [p, 1]).xyz / w, mapping a world point to
    the screen position of the camera basis' own primary-ray mapping (pixel x + 0.5 = u * width, y + 0.5 = v * height):
    what FrameUBO::prevViewProj is to the reference's shaders (temporal_reproject.comp:108-113), for callers that hold a
    camera basis instead of view / projection matrices."""
    c = np.asarray(cam).reshape(-1)[0]
    pos, fwd, right, up = (np.asarray(c[k], dtype=np.float64) for k in ("pos", "fwd", "right", "up"))
    t, a = float(c["tan_half_fov"]), float(c["aspect"])
    rows = np.zeros((4, 4), dtype=np.float64)
    rows[0, :3] = right / (t * a); rows[0, 3] = -np.dot(right, pos) / (t * a)
    rows[1, :3] = -up / t;         rows[1, 3] = np.dot(up, pos) / t
    rows[2, :3] = fwd;             rows[2, 3] = -np.dot(fwd, pos)
    rows[3, :3] = fwd;             rows[3, 3] = -np.dot(fwd, pos)
    return np.ascontiguousarray(rows.T.astype(np.float32)).reshape(-1)       # column-major: M[c * 4 + r]


def camera_view(cam: np.ndarray) -> np.ndarray:
    """FrameUBO::view: glm::lookAt(pos, pos + forward, up) (reference camera.hpp:49-52), column-major (4, 4).T flattened."""
    m = (C.c_float * 16)()
    _ffi.host_lib().blok_camera_view(_ffi.ptr(np.ascontiguousarray(cam, dtype=CAMERA)), m)
    return np.array(m, dtype=np.float32)


def camera_projection(cam: np.ndarray, z_near: float = 0.1, z_far: float = 10000.0) -> np.ndarray:
    """FrameUBO::proj before jitter: glm::perspective (depth 0..1) with p[1][1] *= -1 (camera.hpp:54-59); near / far as
    renderer_draw.cpp:55-56."""
    m = (C.c_float * 16)()
    _ffi.host_lib().blok_camera_projection(_ffi.ptr(np.ascontiguousarray(cam, dtype=CAMERA)), z_near, z_far, m)
    return np.array(m, dtype=np.float32)


def mat4_inverse(m: np.ndarray) -> np.ndarray:
    a = (C.c_float * 16)(*[float(v) for v in np.asarray(m).reshape(-1)])
    out = (C.c_float * 16)()
    if _ffi.host_lib().blok_mat4_inverse(a, out) != 0:
        raise BlokError(-1, "blok_mat4_inverse: singular matrix")
    return np.array(out, dtype=np.float32)


def taa_jitter(frame_index: int) -> np.ndarray:
    """Entry frame_index mod 16 of the reference's Halton(2,3) - 0.5 sequence, in pixels."""
    j = (C.c_float * 2)()
    _ffi.host_lib().blok_taa_jitter(frame_index, j)
    return np.array(j, dtype=np.float32)


def jittered_projection(proj: np.ndarray, jitter_px, width: int, height: int) -> np.ndarray:
    a = (C.c_float * 16)(*[float(v) for v in np.asarray(proj).reshape(-1)])
    j = (C.c_float * 2)(float(jitter_px[0]), float(jitter_px[1]))
    out = (C.c_float * 16)()
    _ffi.host_lib().blok_jittered_projection(a, j, width, height, out)
    return np.array(out, dtype=np.float32)
