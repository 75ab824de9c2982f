"""Screen-tile partition of a frame over ranks (no reference counterpart — blok is
single-GPU).  Tile i (row-major over ceil(W/T) x ceil(H/T) tiles) belongs to rank i % n; a rank stores its
tiles densely in ascending i, each tile row-major, edge tiles padded with miss records.  The same index
arithmetic as the HIP kernels (trace_kernel<Tiles>, untile_kernel)."""
from __future__ import annotations

import numpy as np


def tiles_total(width: int, height: int, tile: int) -> int:
    return ((width + tile - 1) // tile) * ((height + tile - 1) // tile)


def tiles_for_rank(width: int, height: int, tile: int, rank: int, n_ranks: int) -> int:
    total = tiles_total(width, height, tile)
    return (total - rank + n_ranks - 1) // n_ranks if total > rank else 0


def rank_tile_origins(width: int, height: int, tile: int, rank: int, n_ranks: int):
    """Pixel origin (x, y) of each tile this rank owns, in storage order."""
    tiles_x = (width + tile - 1) // tile
    return [((g % tiles_x) * tile, (g // tiles_x) * tile)
            for g in range(rank, tiles_total(width, height, tile), n_ranks)]


def untile(gathered: np.ndarray, width: int, height: int, tile: int, n_ranks: int, tiles_per_rank_max: int) -> np.ndarray:
    """gathered: (n_ranks * tiles_per_rank_max * tile * tile,) records, rank-major.  Returns (height, width)."""
    y, x = np.mgrid[0:height, 0:width]
    tiles_x = (width + tile - 1) // tile
    g = (y // tile) * tiles_x + x // tile
    src = ((g % n_ranks) * tiles_per_rank_max + g // n_ranks) * tile * tile + (y % tile) * tile + (x % tile)
    return gathered[src]


SKY_RGBA = np.uint32(0xFF000000 | (230 << 16) | (200 << 8) | 160)      # trace_core.h: kSkyRgba


def compact_words(tile: int, n_tiles: int) -> int:
    return 1 + n_tiles * (1 + tile * tile)


def compact_tiles(rgba_tiles: np.ndarray, tile: int, n_tiles: int) -> np.ndarray:
    """Reference of blok_hip_compact_tiles_device: word 0 = count, then {local tile index, tile*tile pixels} per tile with a
    non-sky pixel (ascending here; the kernel's order is arbitrary)."""
    px = tile * tile
    t = np.asarray(rgba_tiles).view(np.uint32).reshape(-1)[:n_tiles * px].reshape(n_tiles, px)
    out = np.zeros(compact_words(tile, n_tiles), dtype=np.uint32)
    live = np.flatnonzero((t != SKY_RGBA).any(axis=1))
    out[0] = len(live)
    rec = out[1:1 + len(live) * (1 + px)].reshape(len(live), 1 + px)
    rec[:, 0] = live
    rec[:, 1:] = t[live]
    return out


def scatter_tiles(gathered: np.ndarray, n_ranks: int, rank_stride: int, tile: int, max_records: int, width: int, height: int) -> np.ndarray:
    """Reference of blok_hip_scatter_tiles_device: (height, width) uint32 frame."""
    px = tile * tile
    g = np.asarray(gathered).view(np.uint32).reshape(-1)
    frame = np.full((height, width), SKY_RGBA, dtype=np.uint32)
    tiles_x = (width + tile - 1) // tile
    for r in range(n_ranks):
        base = g[r * rank_stride:]
        for j in range(min(int(base[0]), max_records)):
            rec = base[1 + j * (1 + px):1 + (j + 1) * (1 + px)]
            gt = r + int(rec[0]) * n_ranks
            x0, y0 = (gt % tiles_x) * tile, (gt // tiles_x) * tile
            h, w = min(tile, height - y0), min(tile, width - x0)
            frame[y0:y0 + h, x0:x0 + w] = rec[1:].reshape(tile, tile)[:h, :w]
    return frame


def compact_tile_frames(rgba_frames: np.ndarray, tile: int, n_tiles: int) -> np.ndarray:
    """Reference of blok_hip_compact_tile_frames_device: rgba_frames (F, >= n_tiles * tile*tile).  F count words, then the records
    interleaved by frame: slot j of frame f at word F + (j*F + f) * (1 + tile*tile) (ascending tiles here; the kernel's order is arbitrary)."""
    frames = np.asarray(rgba_frames).view(np.uint32)
    F, px = frames.shape[0], tile * tile
    out = np.zeros(F * compact_words(tile, n_tiles), dtype=np.uint32)
    for f in range(F):
        one = compact_tiles(frames[f], tile, n_tiles)
        out[f] = one[0]
        for j in range(int(one[0])):
            at = F + (j * F + f) * (1 + px)
            out[at:at + 1 + px] = one[1 + j * (1 + px):1 + (j + 1) * (1 + px)]
    return out


def scatter_tile_frames(gathered: np.ndarray, n_ranks: int, rank_stride: int, tile: int, max_records: int, n_frames: int, width: int,
                        height: int) -> np.ndarray:
    """Reference of blok_hip_scatter_tile_frames_device: (n_frames, height, width) uint32 frames."""
    px = tile * tile
    g = np.asarray(gathered).view(np.uint32).reshape(-1)
    frames = np.full((n_frames, height, width), SKY_RGBA, dtype=np.uint32)
    tiles_x = (width + tile - 1) // tile
    for r in range(n_ranks):
        base = g[r * rank_stride:]
        for f in range(n_frames):
            for j in range(min(int(base[f]), max_records)):
                at = n_frames + (j * n_frames + f) * (1 + px)
                rec = base[at:at + 1 + px]
                gt = r + int(rec[0]) * n_ranks
                x0, y0 = (gt % tiles_x) * tile, (gt // tiles_x) * tile
                h, w = min(tile, height - y0), min(tile, width - x0)
                frames[f, y0:y0 + h, x0:x0 + w] = rec[1:].reshape(tile, tile)[:h, :w]
    return frames


# ---- 16-bit coded exchange (blok_hip.h: blok_hip_compact_hit_tile_frames_device / blok_hip_scatter_code_tile_frames_device) ----------
CODE_SKY = 0xFFFF


def compact_code_words(tile: int, n_tiles: int) -> int:
    return 1 + n_tiles * (1 + tile * tile // 2)


def shade_rgba8(albedo: np.ndarray, material_id: np.ndarray, face: np.ndarray) -> np.ndarray:
    """trace_core.h: shade_rgba — albedo (n_materials, 3) float32; ids beyond the table show magenta."""
    albedo = np.asarray(albedo, dtype=np.float32)
    mid = np.asarray(material_id).astype(np.int64)
    rgb = np.where((mid < len(albedo))[..., None], albedo[np.minimum(mid, len(albedo) - 1)], np.array([1.0, 0.0, 1.0], dtype=np.float32))
    k = np.array([0.8, 0.8, 1.0, 0.4, 0.6, 0.6, 0.6, 0.6], dtype=np.float32)[np.asarray(face) & 7]
    q = (np.minimum(rgb * k[..., None], np.float32(1.0)) * np.float32(255.0) + np.float32(0.5)).astype(np.uint32)
    return (0xFF000000 | (q[..., 2] << 16) | (q[..., 1] << 8) | q[..., 0]).astype(np.uint32)


def pixel_codes(hits: np.ndarray, n_materials: int) -> np.ndarray:
    """hits: structured first-hit records (fields material_id, face, hit) -> uint16 codes."""
    code = np.minimum(hits["material_id"].astype(np.int64), n_materials) * 8 + (hits["face"] & 7)
    return np.where(hits["hit"] != 0, code, CODE_SKY).astype(np.uint16)


def compact_hit_tile_frames(hit_frames: np.ndarray, tile: int, n_tiles: int, n_materials: int) -> np.ndarray:
    """Reference of blok_hip_compact_hit_tile_frames_device: hit_frames (F, >= n_tiles * tile*tile) first-hit records."""
    F, px = hit_frames.shape[0], tile * tile
    rw = 1 + px // 2
    out = np.zeros(F * compact_code_words(tile, n_tiles), dtype=np.uint32)
    for f in range(F):
        codes = pixel_codes(hit_frames[f][:n_tiles * px], n_materials).reshape(n_tiles, px)
        live = np.flatnonzero((codes != CODE_SKY).any(axis=1))
        out[f] = len(live)
        for j, t in enumerate(live):
            at = F + (j * F + f) * rw
            out[at] = t
            out[at + 1:at + rw] = codes[t].astype(np.uint32)[0::2] | (codes[t].astype(np.uint32)[1::2] << 16)
    return out


def scatter_code_tile_frames(gathered: np.ndarray, n_ranks: int, rank_stride: int, tile: int, max_records: int, n_frames: int, width: int,
                             height: int, albedo: np.ndarray) -> np.ndarray:
    """Reference of blok_hip_scatter_code_tile_frames_device: (n_frames, height, width) uint32 frames."""
    px = tile * tile
    rw = 1 + px // 2
    g = np.asarray(gathered).view(np.uint32).reshape(-1)
    frames = np.full((n_frames, height, width), SKY_RGBA, dtype=np.uint32)
    tiles_x = (width + tile - 1) // tile
    for r in range(n_ranks):
        base = g[r * rank_stride:]
        for f in range(n_frames):
            for j in range(min(int(base[f]), max_records)):
                at = n_frames + (j * n_frames + f) * rw
                words = base[at + 1:at + rw]
                codes = np.empty(px, dtype=np.uint32)
                codes[0::2] = words & 0xFFFF; codes[1::2] = words >> 16
                rgba = np.where(codes == CODE_SKY, SKY_RGBA, shade_rgba8(albedo, codes >> 3, codes & 7)).astype(np.uint32)
                gt = r + int(base[at]) * n_ranks
                x0, y0 = (gt % tiles_x) * tile, (gt // tiles_x) * tile
                h, w = min(tile, height - y0), min(tile, width - x0)
                frames[f, y0:y0 + h, x0:x0 + w] = rgba.reshape(tile, tile)[:h, :w]
    return frames
