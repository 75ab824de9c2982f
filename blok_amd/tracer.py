"""HipTracer — host-side mirror of the reference's compute backend class (blok::CudaTracer,
reference blok/include/cuda_tracer.hpp:23-58) over the C ABI in include/blok_hip.h.

Lifecycle follows the reference: ``HipTracer(w, h)`` → ``init()`` → ``add_world()``
(= Renderer::addWorld, reference blok/include/renderer.hpp:40-54) → ``draw_frame(cam)`` per frame →
``resize`` / ``shutdown``.  ``begin_frame`` / ``end_frame`` are no-ops as in the reference
(cuda_tracer.cu:450-456).  Errors raise ``BlokError`` (the reference throws std::runtime_error).
There is no CPU path: constructing or tracing without libblok_hip.so and a gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import BlokError, CAMERA, GBuffer, HIT, MATERIAL, RAY, SUB_CHUNK, SVO_NODE, WorldStats


class HipTracer:
    def __init__(self, width: int, height: int, device: int = 0):
        self.width, self.height, self.device = int(width), int(height), int(device)
        self._lib = None
        self._ctx = None
        self._beam_tile = 32

    # -- lifecycle -------------------------------------------------------------------------
    def init(self) -> "HipTracer":
        self._lib = _ffi.hip_lib()
        ctx = C.c_void_p()
        rc = self._lib.blok_hip_create(C.byref(ctx), self.device, self.width, self.height)
        if rc != 0:
            raise BlokError(rc, self._lib.blok_hip_last_error(None).decode())
        self._ctx = ctx
        return self

    def shutdown(self):
        if self._ctx:
            self._lib.blok_hip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    def begin_frame(self):
        pass

    def end_frame(self):
        pass

    def resize(self, width: int, height: int):
        self._check(self._lib.blok_hip_resize(self._ctx, width, height))
        self.width, self.height = int(width), int(height)

    # -- device-resident dense store: edits and rebuilds without leaving HBM ----------------
    def volume_create(self, origin, shape_xyz, chunk_size: int = 128, voxel_size: float = 1.0):
        o = (C.c_int32 * 3)(*[int(v) for v in origin])
        self._check(self._lib.blok_hip_volume_create(self._ctx, o, int(shape_xyz[0]), int(shape_xyz[1]), int(shape_xyz[2]),
                                                     int(chunk_size), float(voxel_size)))
        self._volume_shape = (int(shape_xyz[2]), int(shape_xyz[1]), int(shape_xyz[0]))        # arrays are [z][y][x]

    def set_volume_layout(self, keyed: bool):
        """Diagnostic (blok_hip.h): whether the next volume_create may use the keyed brick layout (default) or the row-major one."""
        self._check(self._lib.blok_hip_set_volume_layout(self._ctx, 1 if keyed else 0))

    def volume_destroy(self):
        self._check(self._lib.blok_hip_volume_destroy(self._ctx))

    def volume_upload(self, density=None, material_ids=None):
        d = None if density is None else np.ascontiguousarray(density, dtype=np.float32)
        m = None if material_ids is None else np.ascontiguousarray(material_ids, dtype=np.uint32)
        for a in (d, m):
            assert a is None or a.shape == self._volume_shape, "arrays are [z][y][x] over the whole box"
        self._check(self._lib.blok_hip_volume_upload(self._ctx, None if d is None else _ffi.ptr(d), None if m is None else _ffi.ptr(m)))

    def volume_download(self):
        d = np.zeros(self._volume_shape, dtype=np.float32)
        m = np.zeros(self._volume_shape, dtype=np.uint32)
        self._check(self._lib.blok_hip_volume_download(self._ctx, _ffi.ptr(d), _ffi.ptr(m)))
        return d, m

    def volume_set_voxels(self, xyz, material_ids=None, density=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32).reshape(-1, 3)
        m = None if material_ids is None else np.ascontiguousarray(material_ids, dtype=np.uint32)
        d = None if density is None else np.ascontiguousarray(density, dtype=np.float32)
        self._check(self._lib.blok_hip_volume_set_voxels(self._ctx, _ffi.ptr(xyz), None if m is None else _ffi.ptr(m),
                                                         None if d is None else _ffi.ptr(d), len(xyz)))

    def volume_apply_brush(self, center, radius: float, value: float, mode: int):
        c = (C.c_float * 3)(*[float(v) for v in center])
        self._check(self._lib.blok_hip_volume_apply_brush(self._ctx, c, float(radius), float(value), int(mode)))

    def volume_rebuild(self, materials=None) -> WorldStats:
        mats = np.zeros(0, dtype=MATERIAL) if materials is None else np.ascontiguousarray(materials, dtype=MATERIAL)
        self._check(self._lib.blok_hip_volume_rebuild(self._ctx, _ffi.ptr(mats) if len(mats) else None, len(mats)))
        return self.world_stats()

    # -- image-space chain: denoiser, TAA, sharpen on device planes ----------------------------
    def denoise_settings(self) -> "_ffi.DenoiseSettings":
        s = _ffi.DenoiseSettings()
        self._lib.blok_denoise_settings_default(C.byref(s))
        return s

    def denoise_device(self, color_ptr: int, world_pos_ptr: int, normal_roughness_ptr: int, prev_view_proj, frame_count: int,
                       out_color_ptr: int, motion_ptr: int = 0, settings=None, stream: int = 0):
        """Denoiser::denoise for one frame over float4 device planes (as trace_paths_device writes them)."""
        planes = _ffi.GBuffer(color_ptr, world_pos_ptr, normal_roughness_ptr, 0)
        m = (C.c_float * 16)(*[float(v) for v in np.asarray(prev_view_proj, dtype=np.float32).reshape(-1)])
        self._check(self._lib.blok_hip_denoise_device(self._ctx, C.byref(planes), motion_ptr or None, m, int(frame_count),
                                                      C.byref(settings) if settings is not None else None, out_color_ptr,
                                                      stream or None))

    def denoise_state(self):
        """Host copies of (history colour, moments, history length, variance, motion vectors) after the last frame."""
        h, w = self.height, self.width
        out = (np.zeros((h, w, 4), np.float32), np.zeros((h, w, 2), np.float32), np.zeros((h, w), np.float32),
               np.zeros((h, w), np.float32), np.zeros((h, w, 2), np.float32))
        self._check(self._lib.blok_hip_denoise_state(self._ctx, *[_ffi.ptr(a) for a in out]))
        return out

    def taa_device(self, color_ptr: int, out_color_ptr: int, frame_count: int, feedback_min: float = 0.93, feedback_max: float = 0.98,
                   motion_ptr: int = 0, stream: int = 0):
        self._check(self._lib.blok_hip_taa_device(self._ctx, color_ptr, motion_ptr or None, feedback_min, feedback_max,
                                                  int(frame_count), out_color_ptr, stream or None))

    def sharpen_device(self, rgba8_ptr: int, out_rgba8_ptr: int, strength: float = 0.5, stream: int = 0):
        self._check(self._lib.blok_hip_sharpen_device(self._ctx, rgba8_ptr, strength, out_rgba8_ptr, stream or None))

    def draw_frame_rt(self, cam: np.ndarray, spp: int = 8, max_bounces: int = 2, settings=None):
        """Renderer::drawFrame's ray-tracing path in one call: path trace -> denoise -> TAA -> tonemap -> sharpen.
        Returns (RGBA8 (h, w) uint32, frames rendered so far)."""
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        out = np.zeros((self.height, self.width), dtype=np.uint32)
        frames = C.c_uint32()
        self._check(self._lib.blok_hip_draw_frame_rt(self._ctx, _ffi.ptr(cam), spp, max_bounces,
                                                     C.byref(settings) if settings is not None else None, _ffi.ptr(out), C.byref(frames)))
        return out, frames.value

    def camera_view_proj(self, cam: np.ndarray) -> np.ndarray:
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        m = (C.c_float * 16)()
        self._lib.blok_camera_view_proj(_ffi.ptr(cam), m)
        return np.array(m, dtype=np.float32)

    def post_reset(self):
        self._check(self._lib.blok_hip_post_reset(self._ctx))

    def set_path_start(self, resume_from_anchor=False, wave_tile_beam=True):
        """Where the path kernel's walks start (blok_hip_set_path_start): from the pixel's latest hit's ancestors, behind the wave tile's own
        beam; frames are identical in every combination."""
        self._check(self._lib.blok_hip_set_path_start(self._ctx, int(bool(resume_from_anchor)), int(bool(wave_tile_beam))))

    def set_ray_batching(self, mode):
        """Path kernel scheduling (never changes a result): 0 / False = off, 1 = one kind of ray at a time, 2 / True (default) = one
        kind at a time and the oldest sample first."""
        self._check(self._lib.blok_hip_set_ray_batching(self._ctx, 2 if mode is True else int(mode)))

    def set_sun_map(self, enabled: bool):
        """Shadow rays stop at the last occluder of their sun-direction column (never changes a result); default on."""
        self._check(self._lib.blok_hip_set_sun_map(self._ctx, int(bool(enabled))))

    def set_beam(self, beam_tile_pixels: int):
        """Beam pre-pass granularity of the frame kernels in pixels (0 = off, default 32); never changes a result."""
        self._check(self._lib.blok_hip_set_beam(self._ctx, beam_tile_pixels))
        self._beam_tile = int(beam_tile_pixels)

    def set_taa_jitter(self, jitter_px=None):
        """Sub-pixel TAA jitter (pixels, each within +-0.5) of the primary rays of all following frames; None = off."""
        if jitter_px is None:
            self._check(self._lib.blok_hip_set_taa_jitter(self._ctx, None))
        else:
            j = (C.c_float * 2)(float(jitter_px[0]), float(jitter_px[1]))
            self._check(self._lib.blok_hip_set_taa_jitter(self._ctx, j))

    def set_rt_taa_jitter(self, enabled: bool):
        """draw_frame_rt applies jitter entry (frame mod 16) by itself (default on = PostProcess::Settings::enableTAA)."""
        self._check(self._lib.blok_hip_set_rt_taa_jitter(self._ctx, 1 if enabled else 0))

    def beam_prepass(self, cam, rect=None, want_visits=False):
        """The pre-pass alone: (t0, visits) per beam tile of the rectangle, row-major (blok_hip.h: blok_hip_beam_prepass)."""
        x0, y0, w, h = rect or (0, 0, self.width, self.height)
        tile = self._beam_tile
        n = ((w + tile - 1) // tile) * ((h + tile - 1) // tile)
        t0 = np.zeros(n, dtype=np.float32)
        visits = np.zeros(n, dtype=np.uint32) if want_visits else None
        cam = np.ascontiguousarray(cam)
        self._check(self._lib.blok_hip_beam_prepass(self._ctx, C.c_void_p(cam.ctypes.data), x0, y0, w, h, C.c_void_p(t0.ctypes.data),
                                                    C.c_void_p(visits.ctypes.data) if want_visits else None, n))
        return t0, visits

    def trace_wave_tiles_device(self, cam, tiles, t0, hits_ptr=0, rgba_ptr=0, rect=None, stream=0):
        """Walks the listed 8x8-pixel wave tiles of the rectangle in list order (blok_hip.h: blok_hip_trace_wave_tiles_device)."""
        x0, y0, w, h = rect or (0, 0, self.width, self.height)
        tiles = np.ascontiguousarray(tiles, dtype=np.uint32)
        t0 = None if t0 is None else np.ascontiguousarray(t0, dtype=np.float32)
        cam = np.ascontiguousarray(cam)
        self._check(self._lib.blok_hip_trace_wave_tiles_device(self._ctx, C.c_void_p(cam.ctypes.data), x0, y0, w, h, C.c_void_p(tiles.ctypes.data),
                                                               C.c_void_p(t0.ctypes.data) if t0 is not None else None, len(tiles),
                                                               C.c_void_p(hits_ptr) if hits_ptr else None, C.c_void_p(rgba_ptr) if rgba_ptr else None,
                                                               C.c_void_p(stream) if stream else None))

    def set_debug_wave_clocks(self, dev_ptr):
        self._check(self._lib.blok_hip_set_debug_wave_clocks(self._ctx, C.c_void_p(dev_ptr) if dev_ptr else None))

    def set_tile_ordering(self, resort_every_n_frames):
        """Longest-first scheduling of the walk from earlier frames' per-wave clocks, re-sorted asynchronously every N frames
        (default 8; 0 / False = off; True = 8); applied only to launches that have the chip to themselves; never changes a result."""
        n = 8 if resort_every_n_frames is True else int(resort_every_n_frames or 0)
        self._check(self._lib.blok_hip_set_tile_ordering(self._ctx, n))

    def set_rank_tile_ordering(self, enabled: bool):
        """Longest-first order and live prefix for a rank's tile launches too (a view at rest); never changes a frame."""
        self._check(self._lib.blok_hip_set_rank_tile_ordering(self._ctx, int(bool(enabled))))

    def set_moving_order(self, enabled: bool):
        """Longest-first scheduling for a camera in motion: the previous frame's clocks, dilated, carried to this view by a whole-tile
        shift (default on; launches alone on the device only); never changes a result."""
        self._check(self._lib.blok_hip_set_moving_order(self._ctx, 1 if enabled else 0))

    def debug_class_order(self, cost, radius, beam=None):
        """Test hook: (order, rank_of, live, depth sums) of blok_hip_debug_class_order for a 2-D array of per-wave-tile costs."""
        cost = np.ascontiguousarray(cost, dtype=np.uint32)
        ty, tx = cost.shape
        order = np.zeros(tx * ty, dtype=np.uint32); rank = np.zeros(tx * ty, dtype=np.uint32)
        live = C.c_uint32(0); sums = np.zeros(3, dtype=np.float32)
        b = None if beam is None else np.ascontiguousarray(beam, dtype=np.float32)
        self._check(self._lib.blok_hip_debug_class_order(self._ctx, C.c_void_p(cost.ctypes.data), tx, ty, int(radius), C.c_void_p(b.ctypes.data) if b is not None else None,
                                                         0 if b is None else len(b), C.c_void_p(order.ctypes.data), C.c_void_p(rank.ctypes.data), C.byref(live), C.c_void_p(sums.ctypes.data)))
        return order, rank, live.value, sums

    def debug_force_order_shift(self, shift=None):
        """Test hook: every ordered launch applies this (x, y) whole-tile shift to its order; None = off."""
        self._check(self._lib.blok_hip_debug_force_order_shift(self._ctx, 0 if shift is None else 1, *(shift or (0, 0))))

    def last_fallback_tiles(self) -> int:
        """Diagnostic: wave tiles the search waves of the latest prefix launch walked themselves (after a synchronise)."""
        return int(self._lib.blok_hip_last_fallback_tiles(self._ctx))

    def last_order_use(self):
        """Diagnostic: (0 row-major | 1 order of this view | 2 order carried from another view, shift_x, shift_y) of the latest rectangle launch."""
        sx, sy = C.c_int32(0), C.c_int32(0)
        return self._lib.blok_hip_last_order_use(self._ctx, C.byref(sx), C.byref(sy)), sx.value, sy.value

    def set_joint_prefix_limit(self, max_walk_waves: int):
        """Diagnostic: cap on the walk waves of a joint launch; the rest is walked by the search waves (same frame)."""
        self._check(self._lib.blok_hip_set_joint_prefix_limit(self._ctx, max_walk_waves))

    def set_list_classes(self, enabled: bool):
        """List launches: order the walk by the previous frame's measured cost, in four classes (default on); never changes a result."""
        self._check(self._lib.blok_hip_set_list_classes(self._ctx, 1 if enabled else 0))

    def last_launch_kind(self) -> int:
        """Which kernels the latest rectangle / tile launch was issued as (blok_hip.h: blok_hip_last_launch_kind)."""
        return int(self._lib.blok_hip_last_launch_kind(self._ctx))

    def set_miss_writer(self, in_walk: bool):
        """Empty tiles' miss pixels: written by the walk launch's waves (True, default) or by the pre-pass (blok_hip.h)."""
        self._check(self._lib.blok_hip_set_miss_writer(self._ctx, 1 if in_walk else 0))

    def set_beam_budget(self, max_node_visits: int):
        """Node visits a beam search may spend (0 = default); running out is answered conservatively, never changes a result."""
        self._check(self._lib.blok_hip_set_beam_budget(self._ctx, max_node_visits))

    def set_voxel_size(self, voxel_size: float):
        """ChunkManager's voxelSize for the next add_world: a power of two in [1/256, 256] (default 1)."""
        self._check(self._lib.blok_hip_set_voxel_size(self._ctx, float(voxel_size)))

    def set_dense_dda(self, enabled: bool):
        """Dense-grid path: when on at add_dense, rectangle traces walk the uploaded grid itself (tiles + LDS occupancy bits)
        with a two-level DDA instead of the derived tree; records are identical."""
        self._check(self._lib.blok_hip_set_dense_dda(self._ctx, 1 if enabled else 0))

    def set_fused(self, enabled: bool):
        """Launch form (blok_hip.h): 0 two launches, 1 persistent grid with queues, 2 joint launch, 3 automatic (default); never changes a result."""
        self._check(self._lib.blok_hip_set_fused(self._ctx, int(enabled)))      # False/0, True/1 or 2 (joint launch, blok_hip.h)

    def frame_queue_stalls(self) -> int:
        """Waves of one-launch frames that ever gave up waiting for a queue entry (0 in a working system); synchronises."""
        n = C.c_uint32(0)
        self._check(self._lib.blok_hip_frame_queue_stalls(self._ctx, C.byref(n)))
        return int(n.value)

    def reset_accum(self):
        self._check(self._lib.blok_hip_reset_accum(self._ctx))

    def _check(self, rc: int):
        if rc != 0:
            raise BlokError(rc, self._lib.blok_hip_last_error(self._ctx).decode())

    # -- world -----------------------------------------------------------------------------
    def add_world(self, world) -> WorldStats:
        """world: blok_amd.world.PackedWorld (nodes, sub_chunks, materials)."""
        nodes = np.ascontiguousarray(world.nodes, dtype=SVO_NODE)
        subs = np.ascontiguousarray(world.sub_chunks, dtype=SUB_CHUNK)
        mats = np.ascontiguousarray(world.materials, dtype=MATERIAL)
        self._check(self._lib.blok_hip_upload_world(self._ctx, _ffi.ptr(nodes), len(nodes), _ffi.ptr(subs), len(subs),
                                                    _ffi.ptr(mats), len(mats)))
        return self.world_stats()

    update_world = add_world

    def add_dense(self, material_ids: np.ndarray, origin=(0, 0, 0), materials: np.ndarray | None = None) -> WorldStats:
        """material_ids[z][y][x], 0 = empty."""
        ids = np.ascontiguousarray(material_ids, dtype=np.uint32)
        nz, ny, nx = ids.shape
        mats = np.zeros(1, dtype=MATERIAL) if materials is None else np.ascontiguousarray(materials, dtype=MATERIAL)
        o = (C.c_int32 * 3)(*origin)
        self._check(self._lib.blok_hip_upload_dense(self._ctx, _ffi.ptr(ids), nx, ny, nz, o, _ffi.ptr(mats), len(mats)))
        return self.world_stats()

    def set_host_build(self, enabled: bool):
        self._check(self._lib.blok_hip_set_host_build(self._ctx, int(enabled)))

    def built_on_device(self) -> bool:
        return bool(self._lib.blok_hip_world_built_on_device(self._ctx))

    def download_tree(self):
        """(nodes as (n, 4) uint32, material ids) of the device-resident structure."""
        st = self.world_stats()
        nodes = np.zeros((st.n_tree_nodes, 4), dtype=np.uint32)
        mats = np.zeros(max(st.n_voxels, 1), dtype=np.uint32)
        self._check(self._lib.blok_hip_download_tree(self._ctx, _ffi.ptr(nodes), len(nodes), _ffi.ptr(mats), len(mats)))
        return nodes, mats[:st.n_voxels]

    def world_stats(self) -> WorldStats:
        s = WorldStats()
        self._check(self._lib.blok_hip_world_stats(self._ctx, C.byref(s)))
        return s

    # -- trace -----------------------------------------------------------------------------
    def draw_frame(self, cam: np.ndarray, rect=None) -> np.ndarray:
        """Primary first-hit records of the frame (or of rect = (x0, y0, w, h)), shape (h, w)."""
        x0, y0, w, h = rect if rect is not None else (0, 0, self.width, self.height)
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        hits = np.zeros(w * h, dtype=HIT)
        self._check(self._lib.blok_hip_trace_primary(self._ctx, _ffi.ptr(cam), x0, y0, w, h, _ffi.ptr(hits)))
        return hits.reshape(h, w)

    def draw_frame_device(self, cam: np.ndarray, hits_ptr: int = 0, rgba_ptr: int = 0, rect=None, stream: int = 0):
        """Asynchronous, device-resident outputs (either pointer may be 0, not both)."""
        x0, y0, w, h = rect if rect is not None else (0, 0, self.width, self.height)
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        self._check(self._lib.blok_hip_trace_primary_device(self._ctx, _ffi.ptr(cam), x0, y0, w, h,
                                                            C.c_void_p(hits_ptr), C.c_void_p(rgba_ptr),
                                                            C.c_void_p(stream)))

    def tiles_for_rank(self, tile: int, rank: int, n_ranks: int) -> int:
        return int(_ffi.hip_lib().blok_hip_tiles_for_rank(self.width, self.height, tile, rank, n_ranks))

    def draw_tiles_device(self, cam: np.ndarray, tile: int, rank: int, n_ranks: int, hits_ptr: int = 0,
                          rgba_ptr: int = 0, stream: int = 0):
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        self._check(self._lib.blok_hip_trace_tiles_device(self._ctx, _ffi.ptr(cam), tile, rank, n_ranks,
                                                          C.c_void_p(hits_ptr), C.c_void_p(rgba_ptr),
                                                          C.c_void_p(stream)))

    def untile_device(self, gathered_ptr: int, elem_bytes: int, tile: int, n_ranks: int, tiles_per_rank_max: int,
                      out_ptr: int, stream: int = 0):
        self._check(self._lib.blok_hip_untile_device(self._ctx, C.c_void_p(gathered_ptr), elem_bytes, tile, n_ranks,
                                                     tiles_per_rank_max, C.c_void_p(out_ptr), C.c_void_p(stream)))

    def compact_tiles_device(self, rgba_tiles_ptr: int, tile: int, n_tiles: int, out_ptr: int, stream: int = 0):
        """Dense RGBA8 tiles of this rank -> {count, {local tile index, pixels} per tile with a non-sky pixel} (blok_hip.h)."""
        self._check(self._lib.blok_hip_compact_tiles_device(self._ctx, C.c_void_p(rgba_tiles_ptr), tile, n_tiles, C.c_void_p(out_ptr), C.c_void_p(stream)))

    def scatter_tiles_device(self, gathered_ptr: int, n_ranks: int, rank_stride_words: int, tile: int, max_records: int, out_ptr: int, stream: int = 0):
        """Root: sky-filled frame + the compacted records of every rank (blok_hip.h)."""
        self._check(self._lib.blok_hip_scatter_tiles_device(self._ctx, C.c_void_p(gathered_ptr), n_ranks, rank_stride_words, tile, max_records,
                                                            C.c_void_p(out_ptr), C.c_void_p(stream)))

    # several frames per call (blok_hip.h: BLOK_MAX_TILE_FRAMES): cams is an array of 1..8 cameras
    def draw_tile_frames_device(self, cams: np.ndarray, tile: int, rank: int, n_ranks: int, frame_stride_tiles: int, hits_ptr: int = 0,
                                rgba_ptr: int = 0, stream: int = 0):
        cams = np.ascontiguousarray(cams, dtype=CAMERA).reshape(-1)
        self._check(self._lib.blok_hip_trace_tile_frames_device(self._ctx, _ffi.ptr(cams), len(cams), tile, rank, n_ranks, frame_stride_tiles,
                                                                C.c_void_p(hits_ptr), C.c_void_p(rgba_ptr), C.c_void_p(stream)))

    def untile_frames_device(self, gathered_ptr: int, elem_bytes: int, tile: int, n_ranks: int, tiles_per_rank_max: int, n_frames: int,
                             frame_stride_tiles: int, out_ptr: int, stream: int = 0):
        self._check(self._lib.blok_hip_untile_frames_device(self._ctx, C.c_void_p(gathered_ptr), elem_bytes, tile, n_ranks, tiles_per_rank_max,
                                                            n_frames, frame_stride_tiles, C.c_void_p(out_ptr), C.c_void_p(stream)))

    def compact_tile_frames_device(self, rgba_tiles_ptr: int, tile: int, n_tiles: int, n_frames: int, frame_stride_tiles: int, out_ptr: int,
                                   stream: int = 0):
        """n_frames frames of dense RGBA8 tiles -> counts + records interleaved by frame (blok_hip.h)."""
        self._check(self._lib.blok_hip_compact_tile_frames_device(self._ctx, C.c_void_p(rgba_tiles_ptr), tile, n_tiles, n_frames, frame_stride_tiles,
                                                                  C.c_void_p(out_ptr), C.c_void_p(stream)))

    def scatter_tile_frames_device(self, gathered_ptr: int, n_ranks: int, rank_stride_words: int, tile: int, max_records: int, n_frames: int,
                                   out_ptr: int, tile_state_ptr: int = 0, stream: int = 0):
        self._check(self._lib.blok_hip_scatter_tile_frames_device(self._ctx, C.c_void_p(gathered_ptr), n_ranks, rank_stride_words, tile, max_records,
                                                                  n_frames, C.c_void_p(out_ptr), C.c_void_p(tile_state_ptr), C.c_void_p(stream)))

    def exchange_code_bits(self) -> int:
        """16 if pixels can travel as 16-bit (material, face) codes with this world's material table, else 0 (blok_hip.h)."""
        return int(self._lib.blok_hip_exchange_code_bits(self._ctx))

    def compact_hit_tile_frames_device(self, hit_tiles_ptr: int, tile: int, n_tiles: int, n_frames: int, frame_stride_tiles: int, out_ptr: int,
                                       stream: int = 0):
        self._check(self._lib.blok_hip_compact_hit_tile_frames_device(self._ctx, C.c_void_p(hit_tiles_ptr), tile, n_tiles, n_frames, frame_stride_tiles,
                                                                      C.c_void_p(out_ptr), C.c_void_p(stream)))

    def scatter_code_tile_frames_device(self, gathered_ptr: int, n_ranks: int, rank_stride_words: int, tile: int, max_records: int, n_frames: int,
                                        out_ptr: int, tile_state_ptr: int = 0, stream: int = 0):
        self._check(self._lib.blok_hip_scatter_code_tile_frames_device(self._ctx, C.c_void_p(gathered_ptr), n_ranks, rank_stride_words, tile, max_records,
                                                                       n_frames, C.c_void_p(out_ptr), C.c_void_p(tile_state_ptr), C.c_void_p(stream)))

    def trace_rays(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=RAY)
        hits = np.zeros(len(rays), dtype=HIT)
        self._check(self._lib.blok_hip_trace_rays(self._ctx, _ffi.ptr(rays), len(rays), _ffi.ptr(hits)))
        return hits

    def trace_paths(self, cam: np.ndarray, spp: int = 8, max_bounces: int = 2, frame_index: int = 0, rect=None):
        """raygen.rgen's sample/bounce loop: dict of (h, w, 4) float32 planes
        color, world_pos, normal_roughness, albedo_metallic."""
        x0, y0, w, h = rect if rect is not None else (0, 0, self.width, self.height)
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        planes = {k: np.zeros((h, w, 4), dtype=np.float32) for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")}
        g = GBuffer(*[planes[k].ctypes.data for k in ("color", "world_pos", "normal_roughness", "albedo_metallic")])
        self._check(self._lib.blok_hip_trace_paths(self._ctx, _ffi.ptr(cam), x0, y0, w, h, spp, max_bounces, frame_index,
                                                   C.byref(g)))
        return planes

    def trace_paths_device(self, cam: np.ndarray, color_ptr: int, spp: int = 8, max_bounces: int = 2,
                           frame_index: int = 0, rect=None, world_pos_ptr: int = 0, normal_roughness_ptr: int = 0,
                           albedo_metallic_ptr: int = 0, stream: int = 0):
        x0, y0, w, h = rect if rect is not None else (0, 0, self.width, self.height)
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        g = GBuffer(color_ptr, world_pos_ptr, normal_roughness_ptr, albedo_metallic_ptr)
        self._check(self._lib.blok_hip_trace_paths_device(self._ctx, _ffi.ptr(cam), x0, y0, w, h, spp, max_bounces,
                                                          frame_index, C.byref(g), C.c_void_p(stream)))

    def trace_paths_ref_device(self, cam: np.ndarray, color_ptr: int = 0, world_pos_ptr: int = 0, normal_roughness_h_ptr: int = 0,
                               albedo_metallic_u8_ptr: int = 0, motion_h_ptr: int = 0, prev_view_proj=None, spp: int = 8,
                               max_bounces: int = 2, frame_index: int = 0, rect=None, stream: int = 0):
        """Path-traced frame with the G-buffer in the reference's image formats (RGBA32F x 2, RGBA16F, RGBA8, RG16F motion)."""
        x0, y0, w, h = rect if rect is not None else (0, 0, self.width, self.height)
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        g = _ffi.GBufferRef(color_ptr, world_pos_ptr, normal_roughness_h_ptr, albedo_metallic_u8_ptr, motion_h_ptr)
        m = None if prev_view_proj is None else (C.c_float * 16)(*[float(v) for v in np.asarray(prev_view_proj, dtype=np.float32).reshape(-1)])
        self._check(self._lib.blok_hip_trace_paths_ref_device(self._ctx, _ffi.ptr(cam), x0, y0, w, h, spp, max_bounces, frame_index,
                                                              m, C.byref(g), C.c_void_p(stream)))

    def denoise_ref_device(self, color_ptr: int, world_pos_ptr: int, normal_roughness_h_ptr: int, motion_h_ptr: int, prev_view_proj,
                           frame_count: int, out_color_ptr: int, settings=None, stream: int = 0):
        """Denoiser::denoise for one frame over reference-format planes (as trace_paths_ref_device writes them)."""
        planes = _ffi.GBufferRef(color_ptr, world_pos_ptr, normal_roughness_h_ptr, 0, motion_h_ptr)
        m = (C.c_float * 16)(*[float(v) for v in np.asarray(prev_view_proj, dtype=np.float32).reshape(-1)])
        self._check(self._lib.blok_hip_denoise_ref_device(self._ctx, C.byref(planes), m, int(frame_count),
                                                          C.byref(settings) if settings is not None else None, out_color_ptr, stream or None))

    def draw_frame_accumulate(self, cam: np.ndarray, spp_per_frame: int = 1, max_bounces: int = 2):
        """Progressive frame of the compute backend (CudaTracer::drawFrame): returns (RGBA8 (h, w) uint32, frames accumulated)."""
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        out = np.zeros((self.height, self.width), dtype=np.uint32)
        frames = C.c_uint32()
        self._check(self._lib.blok_hip_draw_frame_accumulate(self._ctx, _ffi.ptr(cam), spp_per_frame, max_bounces,
                                                             _ffi.ptr(out), C.byref(frames)))
        return out, frames.value

    def accum_download(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._check(self._lib.blok_hip_accum_download(self._ctx, _ffi.ptr(out)))
        return out

    def tonemap(self, hdr: np.ndarray, exposure: float = 1.0, saturation_boost: float = 1.15, operator: int = 1) -> np.ndarray:
        """tonemap.comp: (..., 4) float32 HDR -> (...) uint32 RGBA8 (reference defaults)."""
        hdr = np.ascontiguousarray(hdr, dtype=np.float32)
        out = np.zeros(hdr.shape[:-1], dtype=np.uint32)
        self._check(self._lib.blok_hip_tonemap(self._ctx, _ffi.ptr(hdr), out.size, exposure, saturation_boost, operator,
                                               _ffi.ptr(out)))
        return out

    def tonemap_device(self, hdr_ptr: int, out_rgba8_ptr: int, n_pixels: int = 0, exposure: float = 1.0, saturation_boost: float = 1.15,
                       operator: int = 1, stream: int = 0):
        self._check(self._lib.blok_hip_tonemap_device(self._ctx, hdr_ptr, n_pixels or self.width * self.height, exposure,
                                                      saturation_boost, operator, out_rgba8_ptr, stream or None))

    def shade_rgba8(self, cam: np.ndarray, rect=None) -> np.ndarray:
        x0, y0, w, h = rect if rect is not None else (0, 0, self.width, self.height)
        cam = np.ascontiguousarray(cam, dtype=CAMERA)
        out = np.zeros(w * h, dtype=np.uint32)
        self._check(self._lib.blok_hip_shade_rgba8(self._ctx, _ffi.ptr(cam), x0, y0, w, h, _ffi.ptr(out)))
        return out.reshape(h, w)

    # -- timing ----------------------------------------------------------------------------
    def release_stream(self, stream: int):
        """Before destroying a HIP stream that was passed to *_device entries: the context drops its per-stream scratch and markers."""
        self._check(self._lib.blok_hip_release_stream(self._ctx, C.c_void_p(stream)))

    def set_timing(self, enabled: bool):
        self._check(self._lib.blok_hip_set_timing(self._ctx, int(enabled)))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self._lib.blok_hip_last_kernel_ms(self._ctx, C.byref(ms)))
        return ms.value
