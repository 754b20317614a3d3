"""CPU stand-in for bench.GpuRuntime (gloo): a stub model whose `generate` is arithmetic on the ids, so that bench.py's OWN
control flow for N > 1 — self-launch, rank-0 weight build + broadcast to a shapes-only replica, per-rank batches, warm-up, the
barrier / synchronize bracket around exactly K timed steps, max-over-ranks, the caption gather, the rank proof and the single
JSON line on rank 0 — runs on a box without GPUs. Selected with `bench.py --runtime tests.bench_stub:StubRuntime` (tests only:
nothing here computes what the product computes, and nothing here is imported by the product)."""
import os
import types

import torch

LOG = {"generate_calls": 0, "syncs": 0}


class StubModel:
    def __init__(self, rank):
        # rank 0 holds the "weights", the other ranks an uninitialised replica (GARModel.from_shapes)
        from gar_amd.weights import pack_arenas
        w = [torch.full((3000,), 3.0) if rank == 0 else torch.full((3000,), float("nan")),
             torch.arange(17, dtype=torch.int64) if rank == 0 else torch.zeros(17, dtype=torch.int64)]
        self.arenas, self.w = pack_arenas(w)         # like GARModel._pack_weights: one allocation per dtype, the tensors are views

    def weight_tensors(self):
        return self.w

    def weight_arena_bytes(self):
        return sum(a.numel() * a.element_size() for a in self.arenas.values())

    def broadcast_weights(self, src=0):
        from gar_amd import dp
        return dp.broadcast_arenas(self.arenas, src)

    def generate(self, input_ids=None, max_new_tokens=64, **kw):
        assert kw.get("validate") is False and kw.get("eos_token_id", 0) is None
        assert float(self.w[0][0]) == 3.0 and int(self.w[1][16]) == 16, "the broadcast did not reach this rank"
        LOG["generate_calls"] += 1
        seq = input_ids[:, :1] * int(self.w[0][0]) + torch.arange(max_new_tokens, dtype=torch.int64)[None]
        return types.SimpleNamespace(sequences=seq, input_flags=torch.zeros(1, dtype=torch.int32))

    def _plan_passes(self, B, tiles, S):
        return [B * tiles], [B]


class StubRuntime:
    backend = "gloo"

    def device_of(self, local):
        return "cpu"

    def sync(self):
        LOG["syncs"] += 1

    def peak_mem_gib(self, device):
        return 0.0

    def build_model(self, args, cfg, rank, device):
        assert rank == int(os.environ.get("RANK", "0"))
        return StubModel(rank), None

    def build_batches(self, args, cfg, rank, world, device):
        B, tiles, S = args.batch, 2, 11
        batches = []
        for pidx in range(args.pool):
            ids = torch.full((B, S), 7, dtype=torch.int64)
            import bench
            ids[:, 0] = torch.tensor([bench.region_index(rank, world, pidx * B + k) for k in range(B)])     # the region index itself
            batches.append(dict(input_ids=ids, pixel_values=torch.zeros(B * tiles, 3, 4, 4),
                                global_mask_values=torch.zeros(B * tiles, 3, 4, 4), bboxes=[{}] * B,
                                aspect_ratios=torch.ones(B, 2, dtype=torch.int64)))
        return batches, None, args.pool * B
