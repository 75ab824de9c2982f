"""Frame pipeline: one frame per step on 1..N GPUs of one node (one process per GPU).

No reference counterpart — blok is single-GPU.  The frame is cut into screen tiles dealt
round-robin to the ranks (blok_amd/tiles.py); the world is replicated; every rank traces its own tiles into
first-hit records (kept on the rank) and RGBA8 pixels; ONE collective per frame — a gather of the RGBA8 tile
buffers to rank 0 over RCCL/xGMI — assembles the framebuffer, which rank 0 un-permutes.

Why RGBA8 and a gather-to-root: xGMI is point to point (7 links per GPU, each far below HBM speed), so the
frame time at N = 8 is bounded by bytes per link, not by the kernel: 16-byte hit records would put 16.6 MB on
every link per 4K frame (~0.2 ms, longer than the whole 1-GPU frame), RGBA8 4.1 MB; a gather uses each peer's own
link to the root in parallel where an all-gather would move 7x the bytes.

The pipeline is software-pipelined two deep: the gather of frame k (RCCL's own stream) runs under the trace of
frame k+1 (compute stream); a slot is rewritten only after its gather has been waited for.
"""
from __future__ import annotations

from typing import List, Optional


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


class FramePipeline:
    def __init__(self, backend, width: int, height: int, rank: int = 0, world_size: int = 1, dist=None,
                 tile: int = 32, device="cuda", depth: int = 2, stream_handle=lambda: 0):
        import torch
        self.torch = torch
        self.backend, self.width, self.height = backend, width, height
        self.rank, self.world_size, self.dist, self.tile, self.depth = rank, world_size, dist, tile, depth
        self.stream_handle = stream_handle
        self.frames_submitted = 0
        self.frames_done = 0
        self.in_flight: List[tuple] = []          # (slot, work)
        n_px = width * height
        self.frame_rgba = torch.empty(n_px, dtype=torch.int32, device=device) if rank == 0 else None
        if world_size == 1:
            self.hits = torch.empty((n_px, 4), dtype=torch.int32, device=device)
            return
        self.per_rank = backend.tiles_for_rank(tile, 0, world_size)          # rank 0 owns the most tiles
        n_tile_px = self.per_rank * tile * tile
        self.hits = torch.empty((n_tile_px, 4), dtype=torch.int32, device=device)     # this rank's tiles only
        self.rgba = [torch.empty(n_tile_px, dtype=torch.int32, device=device) for _ in range(depth)]
        self.gathered = ([torch.empty((world_size, n_tile_px), dtype=torch.int32, device=device) for _ in range(depth)]
                         if rank == 0 else None)

    # one frame: enqueue the trace, start its gather, retire the previous frame
    def step(self):
        stream = self.stream_handle()
        if self.world_size == 1:
            self.backend.trace_full(self.hits, self.frame_rgba, stream)
            self.frames_submitted += 1
            self.frames_done += 1
            return
        slot = self.frames_submitted % self.depth
        self.backend.trace_tiles(self.tile, self.rank, self.world_size, self.hits, self.rgba[slot], stream)
        gather_list = [self.gathered[slot][r] for r in range(self.world_size)] if self.rank == 0 else None
        work = self.dist.gather(self.rgba[slot], gather_list=gather_list, dst=0, async_op=True)
        self.in_flight.append((slot, work))
        self.frames_submitted += 1
        while len(self.in_flight) >= self.depth:
            self._retire()

    def _retire(self):
        slot, work = self.in_flight.pop(0)
        work.wait()                                # compute stream (or host, on gloo) waits for that gather only
        if self.rank == 0:
            self.backend.untile(self.gathered[slot], 4, self.tile, self.world_size, self.per_rank,
                                self.frame_rgba, self.stream_handle())
        self.frames_done += 1

    def flush(self):
        while self.in_flight:
            self._retire()
