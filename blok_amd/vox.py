"""MagicaVoxel .vox import and the material library — mirror of the reference's loadVoxFile / importVoxMaterials /
importVoxToChunks / loadAndImportVox (blok/src/vox_loader.cpp) and MaterialLibrary (blok/src/material.cpp) over
libblok_host."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import BlokError, MATERIAL

MATERIAL_DESC = np.dtype([("albedo", "<f4", 3), ("alpha", "<f4"), ("metallic", "<f4"), ("roughness", "<f4"),
                          ("ior", "<f4"), ("specular", "<f4"), ("emission", "<f4", 3), ("emission_power", "<f4"),
                          ("type", "u1"), ("_pad", "u1"), ("vox_palette_index", "<i2"), ("name", "S32")])
assert MATERIAL_DESC.itemsize == 84


class MaterialLibrary:
    def __init__(self):
        self._lib = _ffi.host_lib()
        h = C.c_void_p()
        rc = self._lib.blok_material_library_create(C.byref(h))
        if rc != 0:
            raise BlokError(rc, "blok_material_library_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.blok_material_library_destroy(self._h)
            self._h = None

    __del__ = close

    def __len__(self):
        return int(self._lib.blok_material_library_size(self._h))

    @staticmethod
    def new_desc() -> np.ndarray:
        d = np.zeros(1, dtype=MATERIAL_DESC)
        _ffi.host_lib().blok_material_desc_init(_ffi.ptr(d))
        return d

    def add_material(self, desc: np.ndarray) -> int:
        return int(self._lib.blok_material_library_add(self._h, _ffi.ptr(np.ascontiguousarray(desc, dtype=MATERIAL_DESC))))

    def add_or_find_material(self, desc: np.ndarray) -> int:
        return int(self._lib.blok_material_library_add_or_find(self._h, _ffi.ptr(np.ascontiguousarray(desc, dtype=MATERIAL_DESC))))

    def get_material(self, material_id: int) -> np.ndarray:
        d = np.zeros(1, dtype=MATERIAL_DESC)
        self._lib.blok_material_library_get(self._h, material_id, _ffi.ptr(d))
        return d

    def get_material_id_by_name(self, name: str) -> int:
        return int(self._lib.blok_material_library_id_by_name(self._h, name.encode()))

    def get_or_create_from_color(self, r: int, g: int, b: int) -> int:
        return int(self._lib.blok_material_library_from_color(self._h, r, g, b))

    def set_vox_palette_mapping(self, palette_index: int, material_id: int):
        self._lib.blok_material_library_set_vox_palette(self._h, palette_index, material_id)

    def get_material_from_vox_palette(self, palette_index: int) -> int:
        return int(self._lib.blok_material_library_from_vox_palette(self._h, palette_index))

    def pack_for_gpu(self) -> np.ndarray:
        out = np.zeros(len(self), dtype=MATERIAL)
        rc = self._lib.blok_material_library_pack(self._h, _ffi.ptr(out), len(out))
        if rc != 0:
            raise BlokError(rc, "blok_material_library_pack")
        return out

    def clear(self):
        self._lib.blok_material_library_clear(self._h)


class VoxFile:
    def __init__(self, handle):
        self._lib = _ffi.host_lib()
        self._h = handle

    @classmethod
    def _load(cls, fn, *args):
        lib = _ffi.host_lib()
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = fn(lib, h, err)
        if rc != 0:
            raise BlokError(rc, err.value.decode() or "vox load failed")
        return cls(h)

    @classmethod
    def load_file(cls, path: str) -> "VoxFile":
        return cls._load(lambda lib, h, err: lib.blok_vox_load_file(str(path).encode(), C.byref(h), err, len(err)))

    @classmethod
    def load_memory(cls, data: bytes) -> "VoxFile":
        buf = np.frombuffer(data, dtype=np.uint8)
        return cls._load(lambda lib, h, err: lib.blok_vox_load_memory(_ffi.ptr(buf) if len(buf) else None, len(buf),
                                                                      C.byref(h), err, len(err)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.blok_vox_free(self._h)
            self._h = None

    __del__ = close

    def model_count(self) -> int:
        return int(self._lib.blok_vox_model_count(self._h))

    def model(self, i: int):
        size = (C.c_uint32 * 3)()
        n = C.c_uint32()
        rc = self._lib.blok_vox_model_info(self._h, i, size, C.byref(n))
        if rc != 0:
            raise BlokError(rc, f"no model {i}")
        vox = _ffi.as_array(self._lib.blok_vox_model_voxels(self._h, i), n.value * 4, np.dtype("u1")).reshape(-1, 4)
        return tuple(size), vox

    def palette(self) -> np.ndarray:
        return _ffi.as_array(self._lib.blok_vox_palette(self._h), 256, np.dtype("<u4"))

    def get_material(self, palette_index: int) -> np.ndarray:
        d = np.zeros(1, dtype=MATERIAL_DESC)
        self._lib.blok_vox_get_material(self._h, palette_index, _ffi.ptr(d))
        return d

    def import_materials(self, library: MaterialLibrary) -> np.ndarray:
        mapping = np.zeros(256, dtype=np.uint32)
        self._lib.blok_vox_import_materials(self._h, library._h, _ffi.ptr(mapping))
        return mapping

    def import_to_chunks(self, chunk_manager, world_offset=(0.0, 0.0, 0.0), model_index: int = 0) -> int:
        off = (C.c_float * 3)(*world_offset)
        return int(self._lib.blok_vox_import_to_world(self._h, chunk_manager._h, off, model_index))


def load_and_import_vox(path, chunk_manager, material_library: MaterialLibrary | None = None,
                        world_offset=(0.0, 0.0, 0.0), model_index: int = 0) -> None:
    lib = _ffi.host_lib()
    err = C.create_string_buffer(256)
    off = (C.c_float * 3)(*world_offset)
    rc = lib.blok_load_and_import_vox(str(path).encode(), chunk_manager._h,
                                      material_library._h if material_library is not None else None, off, model_index,
                                      err, len(err))
    if rc != 0:
        raise BlokError(rc, err.value.decode())
    if material_library is not None:
        chunk_manager._material_library = material_library     # keep it alive: the world holds a raw pointer
