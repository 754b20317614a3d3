"""Round-aware chunking of the vision-tower and prefill passes (host logic, no device work).

The bf16 tile GEMM (csrc/gemm_pp.hip) is ONE persistent workgroup per CU walking 256 x 256 output tiles, so a launch
costs ceil(m_tiles * n_tiles / CUs) rounds of K / 64 K-tiles each, and the last round runs at whatever occupancy the
remainder leaves.  With the vision tower over whole 16-region chunks (272 tiles of 1025 tokens: 1090 m-tiles) the
N = 1024 GEMMs (proj, fc2) need 17.03 rounds -> 18: 5.4 % of their time is an almost empty round; the prefill's
o / down GEMMs (295 m-tiles x 8 n-tiles = 9.2 rounds -> 10) lose 7.8 %.  Image tiles are independent in the vision
tower and sequences are independent in the prefill, so the two passes need not use the same chunks, nor whole regions
on the vision side: `plan_chunks` picks the chunk sizes (in image tiles / in sequences) that minimise the K-weighted
number of rounds, by dynamic programming over the item count, under a row cap (activation memory, and the 4-GiB
operand limit of the GEMM's buffer descriptors).
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Sequence, Tuple

TILE = 256            # output tile side of gemm_bf16_pp_kernel
# what one more launch of every GEMM costs, in rounds: ramp-up, the drained last round and the launch gap of the
# kernels in between — keeps the plan from shaving a round with an extra tiny chunk
LAUNCH_OVERHEAD_ROUNDS = 2.0


def gemm_rounds(rows: int, n: int, cus: int) -> int:
    """rounds of the persistent grid for an [rows, n] output."""
    return math.ceil(math.ceil(rows / TILE) * math.ceil(n / TILE) / cus)


def chunk_cost(items: int, rows_per_item: int, gemms: Sequence[Tuple[int, int]], cus: int,
               overhead: float = LAUNCH_OVERHEAD_ROUNDS) -> float:
    """K-weighted rounds of one pass over `items` items; gemms = [(N, K), ...] of one layer."""
    rows = items * rows_per_item
    return float(sum(k * (gemm_rounds(rows, n, cus) + overhead) for n, k in gemms))


@lru_cache(maxsize=256)
def _plan(n_items: int, rows_per_item: int, gemms: Tuple[Tuple[int, int], ...], max_items: int, cus: int) -> Tuple[int, ...]:
    best = [0.0] + [math.inf] * n_items
    prev = [0] * (n_items + 1)
    costs = [0.0] + [chunk_cost(c, rows_per_item, gemms, cus) for c in range(1, max_items + 1)]
    for i in range(1, n_items + 1):
        for c in range(1, min(max_items, i) + 1):
            v = best[i - c] + costs[c]
            if v < best[i] - 1e-9:
                best[i], prev[i] = v, c
    out, i = [], n_items
    while i > 0:
        out.append(prev[i])
        i -= prev[i]
    return tuple(sorted(out, reverse=True))          # largest first: the small remainder runs last


def plan_chunks(n_items: int, rows_per_item: int, gemms: Sequence[Tuple[int, int]], max_rows: int, cus: int = 256,
                max_items: int = 0) -> List[int]:
    """Chunk sizes (sum == n_items, each <= the caps) minimising the K-weighted rounds of `gemms` over all chunks.

    n_items        image tiles (vision tower) or sequences (prefill)
    rows_per_item  GEMM rows an item contributes (tokens per tile / prompt length)
    gemms          (N, K) of every tile GEMM of one layer of the pass
    max_rows       row cap of a chunk (at least one item is always taken)
    """
    if n_items <= 0:
        return []
    cap = max(1, max_rows // max(1, rows_per_item))
    if max_items:
        cap = min(cap, max_items)
    cap = min(cap, n_items)
    return list(_plan(int(n_items), int(rows_per_item), tuple((int(n), int(k)) for n, k in gemms), int(cap), int(cus)))


def waste(chunks: Sequence[int], rows_per_item: int, gemms: Sequence[Tuple[int, int]], cus: int = 256) -> float:
    """K-weighted rounds of the plan / the same work at perfect occupancy (>= 1; reporting only)."""
    ideal = sum(k * (sum(chunks) * rows_per_item / TILE) * math.ceil(n / TILE) / cus for n, k in gemms)
    return sum(chunk_cost(c, rows_per_item, gemms, cus, 0.0) for c in chunks) / ideal
