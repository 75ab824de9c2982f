"""Frame pipeline: one frame per step on 1..N GPUs of one node (one process per GPU).

No reference counterpart for the multi-GPU part — blok is single-GPU; the frames-in-flight
part mirrors the reference's MAX_FRAMES_IN_FLIGHT = 2 frame loop (reference blok/src/renderer_draw.cpp:93-95,335).

Multi-GPU: the frame is cut into screen tiles dealt round-robin to the ranks (blok_amd/tiles.py); the world is
replicated; every rank traces its own tiles into first-hit records (kept on the rank) and RGBA8 pixels; ONE
collective per frame — a gather of the RGBA8 tile buffers to rank 0 over RCCL/xGMI — assembles the framebuffer,
which rank 0 un-permutes.  Why RGBA8 and a gather-to-root: xGMI is point to point (7 links per GPU, each far below
HBM speed), so the frame time at N = 8 is bounded by bytes per link, not by the kernel: 16-byte hit records would
put 16.6 MB on every link per 4K frame (~0.2 ms, longer than half the whole 1-GPU frame), RGBA8 4.1 MB; a gather
uses each peer's own link to the root in parallel where an all-gather would move 7x the bytes.

Sparse exchange (`sparse=True`): most tiles of a frame can be sky, and the root's seven links are the bound at 8 GPUs
(29 MB of RGBA8 per 4K frame), so a rank compacts the tiles with at least one non-sky pixel into {count, {local tile index,
pixels}...} on the device (blok_hip_compact_tiles_device) and only a prefix of that buffer is gathered: S records, S = the
largest count over the ranks.  S has to be a host-side number (collective sizes are), so it is reduced (a one-word MAX
all-reduce) when the frame is TRACED and read back, without waiting, when the frame is RETIRED `depth` frames later — by then
the value has long arrived; the gather itself is issued at retirement.  The root fills the frame with sky and scatters the
records (blok_hip_scatter_tiles_device).  Exact for any camera: the size exchange belongs to the same frame as the data.

Frames in flight: `depth` slots, each with its own HIP stream and buffers.  A trace kernel ends with a long tail
(waves holding grazing rays run up to ~80 us while the rest of the chip drains — a 480x270 frame costs 85 us
against 360 us for 64x the rays), so consecutive frames go to alternating streams and the next frame's waves fill
the slots the previous frame's tail leaves idle; with N ranks the gather of frame k (RCCL's stream) also runs
under the trace of frame k+1.  A slot is reused only after its previous frame was retired.
"""
from __future__ import annotations

import contextlib
from typing import List


class HipMultiTracer:
    """Several devices, ONE process: mirror of blok::HipMultiTracer (include/blok/hip_tracer.hpp) over blok_hip_multi_*
    (include/blok_hip.h) — the C++20 host's way to drive the tile partition; the one-process-per-GPU FramePipeline below is what
    bench.py runs under torch.distributed.  `devices` may repeat an ordinal (ranks rehearsed on one GPU, peer-copy transport)."""

    def __init__(self, devices, width: int, height: int, tile: int = 32, allow_rccl: bool = True):
        import ctypes as C
        from . import _ffi
        self._C, self._ffi, self._lib = C, _ffi, _ffi.hip_lib()
        self.width, self.height, self.tile, self.n = width, height, tile, len(devices)
        self._m = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self._lib.blok_hip_multi_create(C.byref(self._m), arr, len(devices), width, height, tile, 1 if allow_rccl else 0)
        if rc != 0:
            raise _ffi.BlokError(rc, (self._lib.blok_hip_multi_last_error(None) or b"").decode())

    def _check(self, rc):
        if rc != 0:
            raise self._ffi.BlokError(rc, (self._lib.blok_hip_multi_last_error(self._m) or b"").decode())

    @property
    def transport(self) -> str:
        return (self._lib.blok_hip_multi_transport(self._m) or b"").decode()

    def add_world(self, packed):
        import numpy as np
        nodes, subs, mats = (np.ascontiguousarray(a) for a in (packed.nodes, packed.sub_chunks, packed.materials))
        self._check(self._lib.blok_hip_multi_upload_world(self._m, self._ffi.ptr(nodes), len(nodes), self._ffi.ptr(subs), len(subs),
                                                          self._ffi.ptr(mats), len(mats)))

    def set_beam(self, beam_tile_pixels: int):
        for r in range(self.n):
            rc = self._lib.blok_hip_set_beam(self._C.c_void_p(self._lib.blok_hip_multi_context(self._m, r)), beam_tile_pixels)
            if rc != 0:
                raise self._ffi.BlokError(rc, "blok_hip_set_beam")

    def draw_frame(self, cam):
        import numpy as np
        cam = np.ascontiguousarray(cam, dtype=self._ffi.CAMERA)
        out = np.zeros((self.height, self.width), dtype=np.uint32)
        self._check(self._lib.blok_hip_multi_draw_frame(self._m, self._ffi.ptr(cam), self._ffi.ptr(out)))
        return out

    def draw_frames(self, cams):
        """1..8 cameras, one launch pair per device: (n, height, width) RGBA8 frames."""
        import numpy as np
        cams = np.ascontiguousarray(np.concatenate([np.asarray(c).reshape(-1) for c in cams]), dtype=self._ffi.CAMERA)
        out = np.zeros((len(cams), self.height, self.width), dtype=np.uint32)
        self._check(self._lib.blok_hip_multi_draw_frames(self._m, self._ffi.ptr(cams), len(cams), self._ffi.ptr(out)))
        return out

    def set_exchange(self, mode: int):
        """-1 = sparse-pull when possible (default), 0 = dense, 1 = sparse-pull or an error (blok_hip.h)."""
        self._check(self._lib.blok_hip_multi_set_exchange(self._m, mode))

    @property
    def exchange(self) -> str:
        return (self._lib.blok_hip_multi_exchange(self._m) or b"").decode()

    def deny_peer_access(self, deny: bool):
        """Diagnostic: as if the root could not read the other devices' memory (blok_hip.h)."""
        self._check(self._lib.blok_hip_multi_debug_deny_peer_access(self._m, 1 if deny else 0))

    def draw_frames_async(self, cams):
        """Enqueues one call (1..8 cameras) and returns at once; the root's device frames are valid after synchronize()."""
        import numpy as np
        cams = np.ascontiguousarray(np.concatenate([np.asarray(c).reshape(-1) for c in cams]), dtype=self._ffi.CAMERA)
        ptr = self._C.c_void_p()
        self._check(self._lib.blok_hip_multi_draw_frames_device(self._m, self._ffi.ptr(cams), len(cams), self._C.byref(ptr)))
        return ptr.value

    def synchronize(self):
        self._check(self._lib.blok_hip_multi_synchronize(self._m))

    def rank_hits(self, rank: int):
        import numpy as np
        n = int(self._lib.blok_hip_tiles_for_rank(self.width, self.height, self.tile, rank, self.n)) * self.tile * self.tile
        out = np.zeros(n, dtype=self._ffi.HIT)
        self._check(self._lib.blok_hip_multi_download_hits(self._m, rank, self._ffi.ptr(out), n))
        return out

    def shutdown(self):
        if self._m:
            self._lib.blok_hip_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass


class HipBackend:
    """Adapter from torch tensors to the C-ABI entry points of a HipTracer."""

    def __init__(self, tracer, cam):
        self.tracer, self.cam = tracer, cam

    def tiles_for_rank(self, tile, rank, n_ranks):
        return self.tracer.tiles_for_rank(tile, rank, n_ranks)

    def trace_full(self, hits, rgba, stream):
        self.tracer.draw_frame_device(self.cam, hits.data_ptr() if hits is not None else 0,
                                      rgba.data_ptr() if rgba is not None else 0, stream=stream)

    def trace_tiles(self, tile, rank, n_ranks, hits, rgba, stream):
        self.tracer.draw_tiles_device(self.cam, tile, rank, n_ranks, hits.data_ptr() if hits is not None else 0,
                                      rgba.data_ptr() if rgba is not None else 0, stream=stream)

    def untile(self, gathered, elem_bytes, tile, n_ranks, per_rank, out, stream):
        self.tracer.untile_device(gathered.data_ptr(), elem_bytes, tile, n_ranks, per_rank, out.data_ptr(), stream=stream)

    # several frames per call (blok_hip.h: BLOK_MAX_TILE_FRAMES); a view is what one frame is traced from: here, its camera
    max_frames = 8

    def view(self):
        return self.cam

    def trace_tile_frames(self, views, tile, rank, n_ranks, stride_tiles, hits, rgba, stream):
        import numpy as np
        cams = views[0] if len(views) == 1 else np.concatenate([np.asarray(v).reshape(-1) for v in views])
        self.tracer.draw_tile_frames_device(cams, tile, rank, n_ranks, stride_tiles, hits.data_ptr() if hits is not None else 0,
                                            rgba.data_ptr() if rgba is not None else 0, stream=stream)

    def untile_frames(self, gathered, elem_bytes, tile, n_ranks, per_rank, n_frames, stride_tiles, out, stream):
        self.tracer.untile_frames_device(gathered.data_ptr(), elem_bytes, tile, n_ranks, per_rank, n_frames, stride_tiles, out.data_ptr(), stream=stream)

    def compact_frames(self, rgba, tile, n_tiles, n_frames, stride_tiles, out, stream):
        self.tracer.compact_tile_frames_device(rgba.data_ptr(), tile, n_tiles, n_frames, stride_tiles, out.data_ptr(), stream=stream)

    def scatter_frames(self, gathered, n_ranks, rank_stride, tile, max_records, n_frames, out, tile_state, stream):
        self.tracer.scatter_tile_frames_device(gathered.data_ptr(), n_ranks, rank_stride, tile, max_records, n_frames, out.data_ptr(),
                                               tile_state.data_ptr() if tile_state is not None else 0, stream=stream)

    # the same exchange with 16-bit (material, face) codes made from the first-hit tiles (blok_hip.h): half the bytes on the wire
    def code_bits(self):
        return self.tracer.exchange_code_bits()

    def compact_hit_frames(self, hits, tile, n_tiles, n_frames, stride_tiles, out, stream):
        self.tracer.compact_hit_tile_frames_device(hits.data_ptr(), tile, n_tiles, n_frames, stride_tiles, out.data_ptr(), stream=stream)

    def scatter_code_frames(self, gathered, n_ranks, rank_stride, tile, max_records, n_frames, out, tile_state, stream):
        self.tracer.scatter_code_tile_frames_device(gathered.data_ptr(), n_ranks, rank_stride, tile, max_records, n_frames, out.data_ptr(),
                                                    tile_state.data_ptr() if tile_state is not None else 0, stream=stream)

    def compact(self, rgba, tile, n_tiles, out, stream):
        self.tracer.compact_tiles_device(rgba.data_ptr(), tile, n_tiles, out.data_ptr(), stream=stream)

    def scatter(self, gathered, n_ranks, rank_stride, tile, max_records, out, stream):
        self.tracer.scatter_tiles_device(gathered.data_ptr(), n_ranks, rank_stride, tile, max_records, out.data_ptr(), stream=stream)


class FramePipeline:
    """`depth` slots in flight, each a BATCH of `batch` consecutive frames that are traced by ONE launch pair and share ONE exchange.
    At N = 8 a rank's share of a frame is 1/8 of a launch pair whose duration is mostly latency (~130 us alone for 26 us of work),
    and every call and collective costs host time (scripts/host_cost_probe.py on one GPU with a one-rank RCCL group: 76 us of host
    time per frame for trace + dense gather + un-permute, 145 us for the sparse exchange), so frame by frame the host is the bound.
    A batch of 8 frames of a rank's eighth is a launch of the single-GPU size.  A step is still one frame (its view — the camera —
    is taken when step() is called); the launches and the exchange of a batch are issued with its last frame."""

    def __init__(self, backend, width: int, height: int, rank: int = 0, world_size: int = 1, dist=None,
                 tile: int = 32, device="cuda", depth: int = 3, sparse=False, partition=None, batch: int = 1):
        import torch
        self.torch = torch
        self.backend, self.width, self.height = backend, width, height
        self.rank, self.world_size, self.dist, self.tile, self.depth = rank, world_size, dist, tile, depth
        # partition=True with world_size 1 and a process group: the whole distributed path (tiles, collectives, assembly) on one rank —
        # how the RCCL calls get exercised on a one-GPU box (scripts/rccl_smoke.py)
        self.partitioned = world_size > 1 if partition is None else bool(partition) and dist is not None
        self.sparse = bool(sparse) and self.partitioned
        # sparse = 2: pixels travel as 16-bit (material, face) codes when the backend's material table allows it (else as RGBA8)
        self.codes = self.sparse and int(sparse) == 2 and getattr(backend, "code_bits", lambda: 0)() == 16
        self.batch = min(max(1, int(batch)), getattr(backend, "max_frames", 1 << 30)) if self.partitioned else 1
        self.on_gpu = str(device).startswith("cuda")
        self.streams = [torch.cuda.Stream() for _ in range(depth)] if self.on_gpu else [None] * depth
        self.frames_submitted = 0
        self.frames_done = 0
        self.in_flight: List[tuple] = []          # (slot, frames in the batch, work or None)
        self.filling = 0                          # frames traced into the batch that is being filled
        self.records_gathered = 0                 # sparse: tile records per rank asked for so far (sum of S over the frames)
        n_px = width * height
        # the newest completed frame: frame_rgba (rank 0) and hits (this rank's pixels); last_frames: all frames of the batch retired last
        self.frame_rgba = None
        self.hits = None
        self.last_frames = []
        F = self.batch
        if not self.partitioned:
            self._hits = [torch.zeros((1, n_px, 4), dtype=torch.int32, device=device) for _ in range(depth)]
            self._frame = [torch.zeros((1, n_px), dtype=torch.int32, device=device) for _ in range(depth)]
            return
        self.per_rank = backend.tiles_for_rank(tile, 0, world_size)          # rank 0 owns the most tiles
        self.mine = backend.tiles_for_rank(tile, rank, world_size)
        self.n_tile_px = n_tile_px = self.per_rank * tile * tile
        # zero-initialised: the tile slots a rank does not own (rank counts that do not divide the tiles) are never written
        self._hits = [torch.zeros((F, n_tile_px, 4), dtype=torch.int32, device=device) for _ in range(depth)]
        self.rgba = [torch.zeros((F, n_tile_px), dtype=torch.int32, device=device) for _ in range(depth)]
        if self.sparse:
            self.record_words = 1 + (tile * tile // 2 if self.codes else tile * tile)
            self.words = 1 + self.per_rank * self.record_words               # blok_hip_compact_words / blok_hip_compact_code_words
            # a batch's counts, then its records interleaved by frame, so that what travels is one prefix (blok_hip.h)
            self.compacted = [torch.zeros(F * self.words, dtype=torch.int32, device=device) for _ in range(depth)]
            self.smax = [torch.zeros(F, dtype=torch.int32, device=device) for _ in range(depth)]
            self.smax_host = [torch.zeros(F, dtype=torch.int32).pin_memory() if self.on_gpu else torch.zeros(F, dtype=torch.int32)
                              for _ in range(depth)]
            self.smax_event = [torch.cuda.Event() if self.on_gpu else None for _ in range(depth)]
        if rank == 0:
            width_words = F * (self.words if self.sparse else n_tile_px)
            self.gathered = [torch.zeros((world_size, width_words), dtype=torch.int32, device=device) for _ in range(depth)]
            if self.sparse:
                # frames start as sky, with an all-zero tile state: the root then writes only tiles that hold, or held, something else
                from .tiles import SKY_RGBA
                self._frame = [torch.full((F, n_px), int(SKY_RGBA) - (1 << 32), dtype=torch.int32, device=device) for _ in range(depth)]
                self.tile_state = [torch.zeros((F, backend.tiles_for_rank(tile, 0, 1)), dtype=torch.uint8, device=device) for _ in range(depth)]
            else:
                self._frame = [torch.zeros((F, n_px), dtype=torch.int32, device=device) for _ in range(depth)]

    def _on(self, slot):
        return self.torch.cuda.stream(self.streams[slot]) if self.on_gpu else contextlib.nullcontext()

    def _handle(self, slot):
        return self.streams[slot].cuda_stream if self.on_gpu else 0

    # one frame: its view joins the batch that is being filled; the last frame of a batch issues the batch
    def step(self):
        if self.filling == 0:
            while len(self.in_flight) >= self.depth:
                self._retire()
            self._cur = (self.frames_submitted // self.batch) % self.depth
            self._views = []
        if not self.partitioned:
            slot = self._cur
            with self._on(slot):
                self.backend.trace_full(self._hits[slot][0], self._frame[slot][0], self._handle(slot))
        else:
            self._views.append(self.backend.view())
        self.filling += 1
        self.frames_submitted += 1
        if self.filling == self.batch:
            self._issue()

    # trace the batch that is being filled (all `filling` frames of it) and start its exchange
    def _issue(self):
        slot, n_frames = self._cur, self.filling
        work = None
        if self.partitioned:
            with self._on(slot):
                h = self._handle(slot)
                self.backend.trace_tile_frames(self._views, self.tile, self.rank, self.world_size, self.per_rank, self._hits[slot],
                                               self.rgba[slot], h)
                if self.sparse:
                    # compact on the device; the largest record count over the ranks, per frame of the batch, starts its way to every host now
                    if self.codes:
                        self.backend.compact_hit_frames(self._hits[slot], self.tile, self.mine, n_frames, self.per_rank, self.compacted[slot], h)
                    else:
                        self.backend.compact_frames(self.rgba[slot], self.tile, self.mine, n_frames, self.per_rank, self.compacted[slot], h)
                    self.smax[slot][:n_frames].copy_(self.compacted[slot][:n_frames])
                    w = self.dist.all_reduce(self.smax[slot], op=self.dist.ReduceOp.MAX, async_op=True)
                    w.wait()                       # the slot's stream (or the host, on gloo) waits for that reduction only
                    self.smax_host[slot].copy_(self.smax[slot], non_blocking=True)
                    if self.on_gpu:
                        self.smax_event[slot].record()
                else:
                    n = n_frames * self.n_tile_px
                    gather_list = [self.gathered[slot][r][:n] for r in range(self.world_size)] if self.rank == 0 else None
                    work = self.dist.gather(self.rgba[slot].view(-1)[:n], gather_list=gather_list, dst=0, async_op=True)
        self.in_flight.append((slot, n_frames, work))
        self.filling = 0

    def _retire(self):
        slot, n_frames, work = self.in_flight.pop(0)
        with self._on(slot):
            if self.sparse:
                if self.on_gpu:
                    self.smax_event[slot].synchronize()        # enqueued `depth` batches ago
                records = int(self.smax_host[slot][:n_frames].max())    # one prefix length for the batch: the exchange is one block
                n = n_frames * (1 + records * self.record_words)        # the counts and the first `records` record slots of every frame
                self.records_gathered += records * n_frames
                gather_list = [self.gathered[slot][r][:n] for r in range(self.world_size)] if self.rank == 0 else None
                self.dist.gather(self.compacted[slot][:n], gather_list=gather_list, dst=0)
                if self.rank == 0:
                    scatter = self.backend.scatter_code_frames if self.codes else self.backend.scatter_frames
                    scatter(self.gathered[slot], self.world_size, self.gathered[slot].shape[1], self.tile, records,
                            n_frames, self._frame[slot], self.tile_state[slot], self._handle(slot))
            elif work is not None:
                work.wait()                        # the slot's stream (or the host, on gloo) waits for that gather only
                if self.rank == 0:
                    stride_tiles = self.gathered[slot].shape[1] // (self.tile * self.tile)
                    self.backend.untile_frames(self.gathered[slot], 4, self.tile, self.world_size, stride_tiles, n_frames, self.per_rank,
                                               self._frame[slot], self._handle(slot))
        self.hits = self._hits[slot][n_frames - 1]
        if self.rank == 0:
            self.frame_rgba = self._frame[slot][n_frames - 1]
            self.last_frames = [self._frame[slot][f] for f in range(n_frames)]
        self.frames_done += n_frames

    def flush(self):
        if self.filling:
            self._issue()                       # a partial batch at the end
        while self.in_flight:
            self._retire()
        if self.on_gpu:
            for s in self.streams:
                s.synchronize()
