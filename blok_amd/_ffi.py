"""ctypes bindings for the two product libraries.

* ``libblok_host.so`` — host data model (include/blok_world.h), g++ only.
* ``libblok_hip.so``  — gfx950 trace backend (include/blok_hip.h), hipcc.

The bindings are the same stub a maintainer would write for any C consumer; see INTEGRATION.md
for the C++ one.  There is no fallback: a missing library raises ``BlokLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
HOST_LIB = PKG_DIR / "libblok_host.so"
HIP_LIB = Path(os.environ.get("BLOK_HIP_LIB", PKG_DIR / "libblok_hip.so"))   # override: A/B builds of the kernels


class BlokLibraryError(RuntimeError):
    pass


class BlokError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"blok status {status}: {message}")
        self.status = status


# ---------------------------------------------------------------- records (include/blok_hip.h)
SVO_NODE = np.dtype([("child_mask", "<u4"), ("first_child", "<u4"), ("material_id", "<u4"),
                     ("occupancy", "<f4")])
SUB_CHUNK = np.dtype([("node_offset", "<u4"), ("root_node_index", "<u4"), ("node_count", "<u4"),
                      ("start_depth", "<u4"), ("world_min", "<f4", 3), ("sub_chunk_size", "<f4"),
                      ("world_max", "<f4", 3), ("pad0", "<f4")])
MATERIAL = np.dtype([("albedo", "<f4", 3), ("flags", "<u4"), ("emission", "<f4", 3), ("ior", "<f4")])
CAMERA = np.dtype([("pos", "<f4", 3), ("fwd", "<f4", 3), ("right", "<f4", 3), ("up", "<f4", 3),
                   ("tan_half_fov", "<f4"), ("aspect", "<f4")])
HIT = np.dtype([("t", "<f4"), ("material_id", "<u4"), ("voxel", "<i2", 3), ("face", "u1"), ("hit", "u1")])
RAY = np.dtype([("org", "<f4", 3), ("tmin", "<f4"), ("dir", "<f4", 3), ("tmax", "<f4")])
assert SVO_NODE.itemsize == 16 and SUB_CHUNK.itemsize == 48 and MATERIAL.itemsize == 32
assert CAMERA.itemsize == 56 and HIT.itemsize == 16 and RAY.itemsize == 32


class GBuffer(C.Structure):
    _fields_ = [("color", C.c_void_p), ("world_pos", C.c_void_p), ("normal_roughness", C.c_void_p),
                ("albedo_metallic", C.c_void_p)]


class GBufferRef(C.Structure):
    """= blok_gbuffer_ref: the reference's image formats (RGBA32F x 2, RGBA16F, RGBA8, RG16F)."""
    _fields_ = [("color", C.c_void_p), ("world_pos", C.c_void_p), ("normal_roughness", C.c_void_p),
                ("albedo_metallic", C.c_void_p), ("motion", C.c_void_p)]


class DenoiseSettings(C.Structure):
    """= blok_denoise_settings (Denoiser::Settings, reference blok/include/renderer_denoising.hpp:49-66)."""
    _fields_ = [("temporal_alpha", C.c_float), ("moment_alpha", C.c_float), ("variance_clip_gamma", C.c_float),
                ("depth_threshold", C.c_float), ("normal_threshold", C.c_float), ("phi_color", C.c_float),
                ("phi_normal", C.c_float), ("phi_depth", C.c_float), ("atrous_iterations", C.c_int32),
                ("variance_boost", C.c_float), ("min_history_length", C.c_int32)]


class WorldStats(C.Structure):
    _fields_ = [("n_voxels", C.c_uint64), ("n_ref_nodes", C.c_uint64), ("n_sub_chunks", C.c_uint64),
                ("n_tree_nodes", C.c_uint64), ("tree_bytes", C.c_uint64), ("levels", C.c_uint32),
                ("origin", C.c_int32 * 3)]


HOST_SYMBOLS = {
    "blok_morton_encode": (C.c_uint64, [C.c_int32, C.c_int32, C.c_int32]),
    "blok_morton_decode": (None, [C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "blok_morton_octant": (C.c_uint32, [C.c_uint64, C.c_uint32, C.c_uint32]),
    "blok_world_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_float]),
    "blok_world_destroy": (None, [C.c_void_p]),
    "blok_world_last_error": (C.c_char_p, [C.c_void_p]),
    "blok_world_set_voxel": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_float]),
    "blok_world_set_voxels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "blok_world_get_voxel_material": (C.c_uint32, [C.c_void_p, C.POINTER(C.c_float)]),
    "blok_world_apply_brush": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int]),
    "blok_world_rebuild_dirty": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_world_pack": (C.c_int, [C.c_void_p]),
    "blok_world_node_count": (C.c_size_t, [C.c_void_p]),
    "blok_world_sub_chunk_count": (C.c_size_t, [C.c_void_p]),
    "blok_world_nodes": (C.c_void_p, [C.c_void_p]),
    "blok_world_sub_chunks": (C.c_void_p, [C.c_void_p]),
    "blok_world_chunk_count": (C.c_size_t, [C.c_void_p]),
    "blok_world_chunk_info": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "blok_world_chunk_nodes": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "blok_world_find_leaf": (C.c_int64, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32]),
    "blok_material_desc_init": (None, [C.c_void_p]),
    "blok_material_pack": (None, [C.c_void_p, C.c_void_p]),
    "blok_material_library_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "blok_material_library_destroy": (None, [C.c_void_p]),
    "blok_material_library_size": (C.c_uint32, [C.c_void_p]),
    "blok_material_library_add": (C.c_uint32, [C.c_void_p, C.c_void_p]),
    "blok_material_library_add_or_find": (C.c_uint32, [C.c_void_p, C.c_void_p]),
    "blok_material_library_get": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "blok_material_library_id_by_name": (C.c_uint32, [C.c_void_p, C.c_char_p]),
    "blok_material_library_from_color": (C.c_uint32, [C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint8]),
    "blok_material_library_set_vox_palette": (None, [C.c_void_p, C.c_uint8, C.c_uint32]),
    "blok_material_library_from_vox_palette": (C.c_uint32, [C.c_void_p, C.c_uint8]),
    "blok_material_library_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "blok_material_library_clear": (None, [C.c_void_p]),
    "blok_world_set_material_library": (None, [C.c_void_p, C.c_void_p]),
    "blok_world_get_material_library": (C.c_void_p, [C.c_void_p]),
    "blok_world_set_voxel_rgb": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint8, C.c_uint8, C.c_uint8, C.c_float]),
    "blok_vox_load_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "blok_vox_load_memory": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "blok_vox_free": (None, [C.c_void_p]),
    "blok_vox_model_count": (C.c_uint32, [C.c_void_p]),
    "blok_vox_model_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "blok_vox_model_voxels": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "blok_vox_palette": (C.c_void_p, [C.c_void_p]),
    "blok_vox_get_material": (C.c_int, [C.c_void_p, C.c_uint8, C.c_void_p]),
    "blok_vox_import_materials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_vox_import_to_world": (C.c_uint32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint32]),
    "blok_load_and_import_vox": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint32,
                                           C.c_char_p, C.c_size_t]),
    "blok_camera_from_yaw_pitch": (C.c_int, [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float,
                                             C.c_uint32, C.c_uint32, C.c_void_p]),
    "blok_camera_look_at": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                      C.c_uint32, C.c_uint32, C.c_void_p]),
    "blok_camera_view": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "blok_camera_projection": (None, [C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "blok_mat4_inverse": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "blok_taa_jitter": (None, [C.c_uint32, C.POINTER(C.c_float)]),
    "blok_taa_jitter_clip": (None, [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "blok_jittered_projection": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "blok_scene_generate": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "blok_scene_generate_dense": (C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]),
    "blok_scene_materials": (C.c_int, [C.c_uint32, C.c_void_p]),
    "blok_scene_camera": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]),
}

HIP_SYMBOLS = {
    "blok_hip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32]),
    "blok_hip_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "blok_hip_destroy": (None, [C.c_void_p]),
    "blok_hip_last_error": (C.c_char_p, [C.c_void_p]),
    "blok_hip_release_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "blok_hip_upload_world": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t]),
    "blok_hip_upload_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_int32), C.c_void_p, C.c_size_t]),
    "blok_hip_set_host_build": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_world_built_on_device": (C.c_int, [C.c_void_p]),
    "blok_hip_download_tree": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "blok_hip_world_stats": (C.c_int, [C.c_void_p, C.POINTER(WorldStats)]),
    "blok_hip_trace_primary": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_void_p]),
    "blok_hip_trace_primary_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_tiles_for_rank": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "blok_hip_trace_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_untile_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p]),
    "blok_hip_trace_paths_device": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + [C.POINTER(GBuffer), C.c_void_p]),
    "blok_hip_trace_paths": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + [C.POINTER(GBuffer)]),
    "blok_hip_trace_paths_ref_device": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + [C.POINTER(C.c_float), C.POINTER(GBufferRef), C.c_void_p]),
    "blok_hip_denoise_ref_device": (C.c_int, [C.c_void_p, C.POINTER(GBufferRef), C.POINTER(C.c_float), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_tonemap_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "blok_hip_tonemap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "blok_hip_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "blok_hip_shade_rgba8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_void_p]),
    "blok_hip_draw_frame_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]),
    "blok_hip_accum_download": (C.c_int, [C.c_void_p, C.c_void_p]),
    "blok_hip_reset_accum": (C.c_int, [C.c_void_p]),
    "blok_hip_set_beam": (C.c_int, [C.c_void_p, C.c_uint32]),
    "blok_hip_set_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_dense_dda": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_voxel_size": (C.c_int, [C.c_void_p, C.c_float]),
    "blok_hip_multi_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "blok_hip_multi_destroy": (None, [C.c_void_p]),
    "blok_hip_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "blok_hip_multi_device_count": (C.c_uint32, [C.c_void_p]),
    "blok_hip_multi_transport": (C.c_char_p, [C.c_void_p]),
    "blok_hip_multi_context": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "blok_hip_multi_upload_world": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "blok_hip_multi_draw_frame_device": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "blok_hip_multi_synchronize": (C.c_int, [C.c_void_p]),
    "blok_hip_multi_draw_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_multi_set_exchange": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_multi_exchange": (C.c_char_p, [C.c_void_p]),
    "blok_hip_multi_debug_deny_peer_access": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_multi_draw_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "blok_hip_multi_draw_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "blok_hip_multi_download_hits": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "blok_hip_set_beam_budget": (C.c_int, [C.c_void_p, C.c_uint32]),
    "blok_hip_set_miss_writer": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_last_launch_kind": (C.c_int, [C.c_void_p]),
    "blok_hip_set_volume_layout": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_list_classes": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_joint_prefix_limit": (C.c_int, [C.c_void_p, C.c_uint32]),
    "blok_hip_set_tile_ordering": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_rank_tile_ordering": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_moving_order": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_last_fallback_tiles": (C.c_int64, [C.c_void_p]),
    "blok_hip_debug_force_order_shift": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]),
    "blok_hip_debug_class_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]),
    "blok_hip_last_order_use": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "blok_hip_beam_prepass": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p, C.c_size_t]),
    "blok_hip_trace_wave_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_set_debug_wave_clocks": (C.c_int, [C.c_void_p, C.c_void_p]),
    "blok_hip_set_taa_jitter": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "blok_hip_set_rt_taa_jitter": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_compact_words": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "blok_hip_trace_tile_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_untile_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_void_p, C.c_void_p]),
    "blok_hip_compact_tile_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                                      C.c_void_p]),
    "blok_hip_scatter_tile_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_exchange_code_bits": (C.c_uint32, [C.c_void_p]),
    "blok_hip_compact_code_words": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "blok_hip_compact_hit_tile_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                                          C.c_void_p]),
    "blok_hip_scatter_code_tile_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_compact_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "blok_hip_scatter_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "blok_hip_frame_queue_stalls": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "blok_hip_set_sun_map": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_ray_batching": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_set_path_start": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "blok_denoise_settings_default": (None, [C.c_void_p]),
    "blok_hip_denoise_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "blok_hip_denoise_state": (C.c_int, [C.c_void_p] * 6),
    "blok_hip_taa_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]),
    "blok_hip_sharpen_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "blok_hip_post_reset": (C.c_int, [C.c_void_p]),
    "blok_camera_view_proj": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "blok_hip_draw_frame_rt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "blok_hip_volume_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]),
    "blok_hip_volume_destroy": (C.c_int, [C.c_void_p]),
    "blok_hip_volume_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_volume_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "blok_hip_volume_set_voxels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "blok_hip_volume_apply_brush": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int]),
    "blok_hip_volume_rebuild": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "blok_hip_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "blok_hip_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "blok_hip_abi_version": (C.c_uint32, []),
}


def _load(path: Path, symbols: dict) -> C.CDLL:
    if not path.exists():
        raise BlokLibraryError(
            f"{path.name} is not built (expected at {path}); run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        lib = C.CDLL(os.fspath(path))
    except OSError as e:  # e.g. libamdhip64 missing
        raise BlokLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in symbols.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if "BLOK_HIP_LIB" in os.environ and path == HIP_LIB:      # an A/B build of another revision (scripts/build_variant.sh): bind what it has
                continue
            raise BlokLibraryError(f"{path.name} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


_host = None
_hip = None


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        _host = _load(HOST_LIB, HOST_SYMBOLS)
    return _host


def hip_lib() -> C.CDLL:
    global _hip
    if _hip is None:
        # PyTorch ships its own libamdhip64; if this library pulled in /opt/rocm's copy first, torch would later
        # bind to that one and report no device.  Loading torch first makes both share torch's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip = _load(HIP_LIB, HIP_SYMBOLS)
    return _hip


def ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def as_array(address: int, count: int, dtype: np.dtype) -> np.ndarray:
    """Copy `count` records of `dtype` from a C pointer into a fresh numpy array."""
    if count == 0 or not address:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(address)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()
