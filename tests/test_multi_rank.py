"""CPU, world_size 2 over gloo: the screen-tile partition, the gather and the un-permute reproduce the
single-rank frame.  Ranks produce their tiles with the kernel body compiled for the host (test harness)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_dir):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    from blok_amd import tiles as T
    from blok_amd import world as W
    from tests import harness_ffi as H
    from tests import oracle_ffi as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    n, width, height, tile = 64, 200, 136, 32            # ragged: 7x5 tiles, edge tiles padded
    cm = W.ChunkManager(128, 1.0)
    cm.generate_scene(n)
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo()                      # world replicated on every rank
    cam = W.scene_camera(n, 0, width, height)
    hk = H.HostKernel(pw.nodes, pw.sub_chunks)
    full = hk.trace_primary(cam, width, height).reshape(height, width)
    per = T.tiles_for_rank(width, height, tile, 0, world_size)
    mine = np.zeros(per * tile * tile, dtype=O.HIT)
    mine["t"], mine["face"] = -1.0, 0xFF                  # padding = miss records
    for k, (x0, y0) in enumerate(T.rank_tile_origins(width, height, tile, rank, world_size)):
        block = mine[k * tile * tile:(k + 1) * tile * tile].reshape(tile, tile)
        h, w = min(tile, height - y0), min(tile, width - x0)
        block[:h, :w] = full[y0:y0 + h, x0:x0 + w]        # this rank only keeps its own tiles
    send = torch.from_numpy(mine.view(np.int32).reshape(-1, 4).copy())
    gathered = torch.empty((world_size * len(send), 4), dtype=torch.int32)
    dist.all_gather_into_tensor(gathered, send)
    frame = T.untile(gathered.numpy().view(O.HIT).reshape(-1), width, height, tile, world_size, per)
    ok = frame.tobytes() == full.tobytes()
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # the bench's max-over-ranks timing reduction
    (Path(out_dir) / f"rank{rank}.txt").write_text(f"{int(ok)} {float(t[0])} {T.tiles_for_rank(width, height, tile, rank, world_size)}")
    dist.destroy_process_group()


def test_tile_partition_gather_untile_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "rank0.txt").read_text().split()
    r1 = (tmp_path / "rank1.txt").read_text().split()
    assert r0[0] == "1" and r1[0] == "1"
    assert float(r0[1]) == 2.0 and float(r1[1]) == 2.0
    assert int(r0[2]) + int(r1[2]) == 35 and int(r0[2]) == 18


def test_tile_math_matches_c_abi():
    from blok_amd import _ffi
    from blok_amd import tiles as T
    lib = _ffi.hip_lib()
    for (w, h, tile, n) in [(3840, 2160, 32, 8), (3840, 2160, 32, 1), (200, 136, 32, 2), (100, 70, 64, 3), (16, 16, 16, 4)]:
        for r in range(n):
            assert lib.blok_hip_tiles_for_rank(w, h, tile, r, n) == T.tiles_for_rank(w, h, tile, r, n)
            assert len(T.rank_tile_origins(w, h, tile, r, n)) == T.tiles_for_rank(w, h, tile, r, n)


class _FakeBackend:
    """CPU stand-in for HipBackend: 'traces' by copying from a precomputed frame, so that the scheduling code of
    blok_amd.multi_gpu.FramePipeline (slots, async gather, retire order, un-permute) runs unchanged over gloo."""

    def __init__(self, width, height, frames):
        self.width, self.height, self.frames, self.k = width, height, frames, 0

    def tiles_for_rank(self, tile, rank, n):
        from blok_amd import tiles as T
        return T.tiles_for_rank(self.width, self.height, tile, rank, n)

    def view(self):                      # what a frame is traced from: here, its index
        self.k += 1
        return self.k - 1

    def trace_tile_frames(self, views, tile, rank, n, stride_tiles, hits, rgba, stream):
        for f, v in enumerate(views):
            self._trace_tiles(v, tile, rank, n, rgba[f])

    def untile_frames(self, gathered, elem_bytes, tile, n, per, n_frames, stride_tiles, out, stream):
        flat = gathered.view(-1)
        for f in range(n_frames):
            self.untile(flat[f * stride_tiles * tile * tile:], elem_bytes, tile, n, per, out[f], stream)

    def compact_frames(self, rgba, tile, n_tiles, n_frames, stride_tiles, out, stream):
        import torch
        from blok_amd import tiles as T
        words = T.compact_tile_frames(rgba[:n_frames].numpy(), tile, n_tiles)
        out[:len(words)].copy_(torch.from_numpy(words.view(np.int32)))

    def scatter_frames(self, gathered, n, rank_stride, tile, max_records, n_frames, out, tile_state, stream):
        import torch
        from blok_amd import tiles as T
        frames = T.scatter_tile_frames(gathered.numpy().reshape(-1), n, rank_stride, tile, max_records, n_frames, self.width, self.height)
        out[:n_frames].copy_(torch.from_numpy(frames.view(np.int32).reshape(n_frames, -1)))

    def trace_tiles(self, tile, rank, n, hits, rgba, stream):
        self._trace_tiles(self.view(), tile, rank, n, rgba)

    def _trace_tiles(self, k, tile, rank, n, rgba):
        import torch
        from blok_amd import tiles as T
        frame = self.frames[k % len(self.frames)]
        out = np.full(len(rgba), -1, dtype=np.int32)
        for j, (x0, y0) in enumerate(T.rank_tile_origins(self.width, self.height, tile, rank, n)):
            block = out[j * tile * tile:(j + 1) * tile * tile].reshape(tile, tile)
            h, w = min(tile, self.height - y0), min(tile, self.width - x0)
            block[:h, :w] = frame[y0:y0 + h, x0:x0 + w]
        rgba.copy_(torch.from_numpy(out))

    def untile(self, gathered, elem_bytes, tile, n, per, out, stream):
        import torch
        from blok_amd import tiles as T
        assert elem_bytes == 4
        out.copy_(torch.from_numpy(T.untile(gathered.numpy().reshape(-1), self.width, self.height, tile, n, per).reshape(-1)))

    def compact(self, rgba, tile, n_tiles, out, stream):
        import torch
        from blok_amd import tiles as T
        words = T.compact_tiles(rgba.numpy(), tile, n_tiles)
        out[:len(words)].copy_(torch.from_numpy(words.view(np.int32)))

    def scatter(self, gathered, n, rank_stride, tile, max_records, out, stream):
        import torch
        from blok_amd import tiles as T
        frame = T.scatter_tiles(gathered.numpy().reshape(-1), n, rank_stride, tile, max_records, self.width, self.height)
        out.copy_(torch.from_numpy(frame.view(np.int32).reshape(-1)))


def _pipeline_worker(rank, world_size, port, out_dir, sparse=False, batch=1):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    from blok_amd.multi_gpu import FramePipeline
    from blok_amd import tiles as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    width, height, tile = 200, 136, 32
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 1 << 30, size=(height, width)).astype(np.int32) for _ in range(5)]
    if sparse:
        # a different, uneven set of all-sky tiles in every frame (one frame all sky, one with a single live pixel): the two
        # ranks' record counts differ, so the gathered prefix is sized by the larger one
        sky = np.int32(T.SKY_RGBA.view(np.int32) if hasattr(T.SKY_RGBA, "view") else np.uint32(T.SKY_RGBA).astype(np.int32))
        for k, f in enumerate(frames):
            if k == 1:
                f[:] = sky
            elif k == 3:
                f[:] = sky
                f[70, 131] = 12345
            else:
                for ty in range(0, height, tile):
                    for tx in range(0, width, tile):
                        if rng.random() < 0.6:
                            f[ty:ty + tile, tx:tx + tile] = sky
    pipe = FramePipeline(_FakeBackend(width, height, frames), width, height, rank, world_size, dist, tile=tile,
                         device="cpu", depth=2, sparse=sparse, batch=batch)
    seen = []

    def collect():
        if rank == 0 and pipe.frames_done > len(seen):        # the frames of the batch retired last are complete on the root
            seen.extend(f.numpy().reshape(height, width).copy() for f in pipe.last_frames)
    for k in range(5):
        pipe.step()
        collect()
    if pipe.filling:
        pipe._issue()                                       # 5 frames in batches of 2: the last batch is partial
    while pipe.in_flight:
        pipe._retire()
        collect()
    pipe.flush()
    if rank == 0:
        ok = len(seen) == 5 and all((a == b).all() for a, b in zip(seen, frames))
    else:
        ok = pipe.frames_done == 5
    if sparse:      # fewer records travelled than a dense gather's 18 tiles per rank and frame, and none for the all-sky frame
        ok = ok and 0 < pipe.records_gathered < 5 * 18 * 0.7
    (Path(out_dir) / f"pipe{rank}.txt").write_text(str(int(ok)))
    dist.destroy_process_group()


def test_frame_pipeline_two_ranks_in_order(tmp_path):
    """The 2-deep pipeline delivers every frame, in order, bit-identical, on rank 0."""
    import torch.multiprocessing as mp
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "pipe0.txt").read_text() == "1" and (tmp_path / "pipe1.txt").read_text() == "1"


def test_frame_pipeline_sparse_gather_two_ranks(tmp_path):
    """Sparse exchange (compact -> size reduction at trace time -> prefix gather at retirement -> sky fill + scatter): every
    frame, in order, bit-identical on rank 0, with sky tiles that never travel."""
    import torch.multiprocessing as mp
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    assert (tmp_path / "pipe0.txt").read_text() == "1" and (tmp_path / "pipe1.txt").read_text() == "1"


@pytest.mark.parametrize("sparse", [False, True])
def test_frame_pipeline_batched_exchange_two_ranks(tmp_path, sparse):
    """Two frames per exchange (and a partial last batch): the same frames, in order, bit-identical on rank 0."""
    import torch.multiprocessing as mp
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), sparse, 2), nprocs=2, join=True)
    assert (tmp_path / "pipe0.txt").read_text() == "1" and (tmp_path / "pipe1.txt").read_text() == "1"


def test_compact_scatter_reference_round_trip():
    """tiles.compact_tiles / scatter_tiles (the references of the two kernels): a frame cut into rank tiles, compacted per rank
    and scattered back equals the frame, for ragged frames and rank counts that do not divide the tiles."""
    from blok_amd import tiles as T
    rng = np.random.default_rng(5)
    for (w, h, tile, n) in [(200, 136, 32, 2), (100, 70, 16, 3), (64, 64, 32, 8)]:
        frame = np.full((h, w), T.SKY_RGBA, dtype=np.uint32)
        for ty in range(0, h, tile):
            for tx in range(0, w, tile):
                if rng.random() < 0.5:
                    frame[ty:ty + tile, tx:tx + tile] = rng.integers(0, 1 << 31, size=frame[ty:ty + tile, tx:tx + tile].shape)
        per = T.tiles_for_rank(w, h, tile, 0, n)
        stride = T.compact_words(tile, per)
        gathered = np.zeros(n * stride, dtype=np.uint32)
        most = 0
        for r in range(n):
            mine = T.tiles_for_rank(w, h, tile, r, n)
            dense = np.full(per * tile * tile, T.SKY_RGBA, dtype=np.uint32)
            for k, (x0, y0) in enumerate(T.rank_tile_origins(w, h, tile, r, n)):
                block = dense[k * tile * tile:(k + 1) * tile * tile].reshape(tile, tile)
                hh, ww = min(tile, h - y0), min(tile, w - x0)
                block[:hh, :ww] = frame[y0:y0 + hh, x0:x0 + ww]
            c = T.compact_tiles(dense, tile, mine)
            gathered[r * stride:r * stride + len(c)] = c
            most = max(most, int(c[0]))
        assert (T.scatter_tiles(gathered, n, stride, tile, most, w, h) == frame).all()


def test_coded_exchange_reference_round_trip():
    """tiles.compact_hit_tile_frames / scatter_code_tile_frames (the references of the 16-bit exchange kernels): first-hit records
    cut into rank tiles, coded per rank (min(material, n) * 8 + face, 0xFFFF = sky), interleaved by frame, expanded on the root ==
    the shading of the records (material ids beyond the table included), for ragged frames and uneven rank counts."""
    from blok_amd import tiles as T
    from tests import oracle_ffi as O
    rng = np.random.default_rng(11)
    n_mat = 37
    albedo = rng.uniform(0, 1.2, (n_mat, 3)).astype(np.float32)
    for (w, h, tile, n, F) in [(200, 136, 32, 2, 3), (100, 70, 16, 3, 1), (64, 64, 32, 8, 2)]:
        hits = np.zeros((F, h, w), dtype=O.HIT)
        hits["face"] = 0xFF
        for f in range(F):
            for ty in range(0, h, tile):
                for tx in range(0, w, tile):
                    if rng.random() < 0.5:
                        blk = hits[f, ty:ty + tile, tx:tx + tile]
                        blk["hit"] = rng.random(blk.shape) < 0.7
                        blk["material_id"] = rng.integers(0, n_mat + 5, size=blk.shape)       # some beyond the table
                        blk["face"] = rng.integers(0, 6, size=blk.shape)
        want = np.where(hits["hit"] != 0, T.shade_rgba8(albedo, hits["material_id"], hits["face"]), T.SKY_RGBA).astype(np.uint32)
        per = T.tiles_for_rank(w, h, tile, 0, n)
        stride = F * T.compact_code_words(tile, per)
        gathered = np.zeros(n * stride, dtype=np.uint32)
        most = 0
        for r in range(n):
            mine = T.tiles_for_rank(w, h, tile, r, n)
            dense = np.zeros((F, per * tile * tile), dtype=O.HIT)
            dense["face"] = 0xFF
            for f in range(F):
                for k, (x0, y0) in enumerate(T.rank_tile_origins(w, h, tile, r, n)):
                    block = dense[f, k * tile * tile:(k + 1) * tile * tile].reshape(tile, tile)
                    hh, ww = min(tile, h - y0), min(tile, w - x0)
                    block[:hh, :ww] = hits[f, y0:y0 + hh, x0:x0 + ww]
            c = T.compact_hit_tile_frames(dense, tile, mine, n_mat)
            gathered[r * stride:r * stride + len(c)] = c
            most = max(most, int(c[:F].max()))
        got = T.scatter_code_tile_frames(gathered, n, stride, tile, most, F, w, h, albedo)
        assert (got == want).all()


def _share_worker(rank, world_size, port, out_dir):
    sys.path.insert(0, str(ROOT))
    import hashlib
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    built = []
    real = bench.build_world
    bench.build_world = lambda n, seed: (built.append(rank), real(n, seed))[1]       # who actually builds
    pw = bench.build_world_shared(64, 0xB10C0001, dist, rank)
    digest = hashlib.sha256(pw.nodes.tobytes() + pw.sub_chunks.tobytes() + pw.materials.tobytes()).hexdigest()
    (Path(out_dir) / f"share{rank}.txt").write_text(f"{digest} {len(built)} {len(pw.nodes)} {len(pw.sub_chunks)} {len(pw.materials)}")
    dist.barrier()                                            # bench.py's exit: everybody leaves together
    dist.destroy_process_group()


def test_world_is_built_once_and_broadcast_world_size_2(tmp_path):
    """bench.py --gpus N: rank 0 builds the world, the others receive the three arrays of WorldSvoGpu over the process group."""
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT))
    import hashlib
    import bench
    port = _free_port()
    mp.spawn(_share_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = bench.build_world(64, 0xB10C0001)
    digest = hashlib.sha256(want.nodes.tobytes() + want.sub_chunks.tobytes() + want.materials.tobytes()).hexdigest()
    rows = [(tmp_path / f"share{r}.txt").read_text().split() for r in range(2)]
    assert rows[0][0] == rows[1][0] == digest and len(want.nodes) > 1000
    assert (rows[0][1], rows[1][1]) == ("1", "0")            # one build, on rank 0
