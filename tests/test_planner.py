"""Round-aware pass planner (gar_amd/planner.py): host logic only."""
import math

import pytest

from gar_amd.planner import chunk_cost, gemm_rounds, plan_chunks, waste

VIT = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]            # GAR-1B PE-L/14 layer: (N, K)
LLM = [(3072, 2048), (2048, 2048), (16384, 2048), (2048, 8192)]           # Llama-3.2-1B layer


def test_rounds_known_values():
    # 272 tiles x 1025 tokens = 1090 m-tiles; N = 1024 -> 4360 output tiles = 17.03 rounds of 256 CUs
    assert gemm_rounds(272 * 1025, 1024, 256) == 18
    assert gemm_rounds(271 * 1025, 1024, 256) == 17
    assert gemm_rounds(16 * 4718, 2048, 256) == 10 and gemm_rounds(26 * 4718, 2048, 256) == 15


@pytest.mark.parametrize("n,rows,gemms,cap", [(1088, 1025, VIT, 400_000), (64, 4718, LLM, 131_072),
                                               (17, 1025, VIT, 400_000), (1, 4718, LLM, 131_072), (5, 300_000, LLM, 131_072)])
def test_plan_is_a_partition_under_the_cap(n, rows, gemms, cap):
    p = plan_chunks(n, rows, gemms, cap)
    assert sum(p) == n and all(c >= 1 for c in p)
    assert all(c * rows <= cap or c == 1 for c in p)
    assert p == sorted(p, reverse=True)


def test_plan_beats_region_chunks_on_the_bench_shapes():
    pv = plan_chunks(64 * 17, 1025, VIT, 400_000)
    assert waste(pv, 1025, VIT) < 1.015 < waste([272] * 4, 1025, VIT)
    pl = plan_chunks(64, 4718, LLM, 131_072)
    assert waste(pl, 4718, LLM) < 1.015 < waste([16] * 4, 4718, LLM)
    assert len(pv) <= 4 and len(pl) <= 4           # the launch overhead term keeps tiny remainder chunks out


def test_plan_is_optimal_against_brute_force():
    def brute(n, rows, gemms, cap):
        best = {0: 0.0}
        for i in range(1, n + 1):
            best[i] = min(best[i - c] + chunk_cost(c, rows, gemms, 256) for c in range(1, min(cap, i) + 1))
        return best[n]
    for n, rows, gemms, cap in [(40, 1025, VIT, 16), (23, 4718, LLM, 9)]:
        p = plan_chunks(n, rows, gemms, cap * rows)
        assert math.isclose(sum(chunk_cost(c, rows, gemms, 256) for c in p), brute(n, rows, gemms, cap), rel_tol=1e-12)


def test_empty_and_single():
    assert plan_chunks(0, 1025, VIT, 1000) == []
    assert plan_chunks(1, 1025, VIT, 10) == [1]
