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
