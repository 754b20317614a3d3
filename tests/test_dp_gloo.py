"""CPU, world_size 2 over gloo: the data-parallel runner's collectives (one weight broadcast per dtype arena, caption gather,
barrier, max-over-ranks) and the region sharding — the N > 1 path of bench.py without a GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from gar_amd import dp
    r, l, w = dp.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(123)
    ref = [torch.randn(1000, generator=g), torch.randn(7, 13, generator=g), torch.randn(50000, generator=g),
           torch.randint(0, 100, (33,), generator=g), torch.randn(3, generator=g).to(torch.bfloat16)]
    # ---- the replica's weights as ONE arena per dtype: one in-place collective per arena, bit-equal replicas
    import torch.distributed as dist
    from gar_amd.weights import ARENA_ALIGN_BYTES, pack_arenas
    src_w = [t.clone() if rank == 0 else torch.full_like(t, 7) for t in ref]
    arenas, views = pack_arenas(src_w)
    assert set(arenas) == {torch.float32, torch.int64, torch.bfloat16}
    for v, t in zip(views, src_w):               # views of the arena, aligned, same values
        a = arenas[v.dtype]
        off = v.data_ptr() - a.data_ptr()
        assert 0 <= off < a.numel() * a.element_size() and off % ARENA_ALIGN_BYTES == 0 and torch.equal(v, t)
    calls = []
    real = dist.broadcast
    dist.broadcast = lambda t, src=0, **kw: (calls.append(t.data_ptr()), real(t, src=src, **kw))[1]
    try:
        n = dp.broadcast_arenas(arenas, src=0)
    finally:
        dist.broadcast = real
    assert n == len(arenas) == len(calls) and sorted(calls) == sorted(a.data_ptr() for a in arenas.values())
    assert all(torch.equal(v, t) for v, t in zip(views, ref))
    # ---- region sharding + caption gather
    n_regions, n_new = 7, 5
    idx = dp.shard_indices(n_regions, rank, world)
    local = torch.tensor([[1000 * i + j for j in range(n_new)] for i in idx[:3]], dtype=torch.int64)  # equal-size shards
    got = dp.gather_captions(local, dst=0)
    if rank == 0:
        assert got is not None and len(got) == world
        flat = torch.cat(got)
        assert sorted(flat[:, 0].tolist()) == sorted(1000 * i for r_ in range(world) for i in dp.shard_indices(n_regions, r_, world)[:3])
    else:
        assert got is None
    dp.barrier()
    assert dp.max_over_ranks(float(rank + 1), device="cpu") == float(world)
    # ---- benchmark loops: per-rank (index, record) lists -> index-ordered list on rank 0
    from gar_amd.bench_loops import _gather
    items = [(i, {"id": i, "text": f"caption {i}"}) for i in dp.shard_indices(5, rank, world)]
    merged = _gather(items, rank, world)
    if rank == 0:
        assert [m["id"] for m in merged] == [0, 1, 2, 3, 4]
    else:
        assert merged is None
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("ok")
    torch.distributed.destroy_process_group()


def test_dp_collectives_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_shard_indices_partition():
    from gar_amd import dp
    for n in (0, 1, 8, 13):
        for world in (1, 2, 8):
            parts = [dp.shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_paths_are_noops():
    from gar_amd import dp
    t = torch.arange(6).view(2, 3)
    assert dp.gather_captions(t)[0] is t
    assert dp.broadcast_arenas({t.dtype: t}) == 0
    dp.barrier()
    assert dp.max_over_ranks(3.5) == 3.5
