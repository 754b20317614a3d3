"""Data-parallel serving of regions over the GPUs of one node: one process per GPU, full model replica per rank.

The reference runs one process / one GPU / batch 1 and issues no collective. Regions are independent,
so the only exchanges are:
  * one RCCL broadcast of the prepared weight tensors from rank 0 at start-up (3.1 GB bf16 for GAR-1B),
  * one RCCL gather of the [n_local, n_new] caption ids to rank 0 per step,
  * barriers around timed regions.
No activation ever crosses xGMI; there is no all-reduce on the path."""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin partition: rank r takes items i with i % world == r."""
    return list(range(rank, n_items, world))


def broadcast_arenas(arenas, src: int = 0) -> int:
    """The weight exchange of the data-parallel runner: ``arenas`` = {dtype: flat tensor} holding every
    prepared weight of a replica (gar_amd.weights.pack_arenas) — ONE in-place broadcast per arena from rank ``src``, no staging
    copy. Returns the number of collectives issued (0 without a process group)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    n = 0
    for dt in sorted(arenas, key=str):          # the same order on every rank
        dist.broadcast(arenas[dt], src=src)
        n += 1
    return n


def gather_captions(local_ids: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """[n_local, n_new] int64 from every rank -> list on rank `dst` (None elsewhere)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_ids]
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        out = [torch.empty_like(local_ids) for _ in range(world)]
        dist.gather(local_ids, out, dst=dst)
        return out
    dist.gather(local_ids, None, dst=dst)
    return None


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_floats(x: float, device=None) -> List[float]:
    """every rank's value, in rank order, on every rank ([x] without a process group)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [x]
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def describe() -> dict:
    """what the process group actually is — printed into bench.py's JSON line so that a scaling run proves its ranks."""
    if not dist.is_initialized():
        return {"ranks_seen": 1, "backend": None}
    return {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend()}
