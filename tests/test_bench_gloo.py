"""CPU, world_size 2 over gloo: bench.py's OWN control flow for N > 1 — rank-0 weight build + broadcast to a shapes-only
replica, per-rank batches, warm-up, the barrier / synchronize bracket around exactly K timed steps, max-over-ranks, the
caption gather and the single JSON line on rank 0 — with a stub model on the CPU in place of the GPU runtime
(bench.GpuRuntime). The collectives themselves are covered by tests/test_dp_gloo.py."""
import io
import json
import os
import socket
import sys
import types

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
    import bench
    from gar_amd import dp
    log = {"generate_calls": 0, "syncs": 0}

    class StubModel:
        def __init__(self):
            # rank 0 holds the "weights", the other ranks an uninitialised replica (GARModel.from_shapes)
            self.w = [torch.full((3000,), 3.0) if rank == 0 else torch.full((3000,), float("nan")),
                      torch.arange(17, dtype=torch.int64) if rank == 0 else torch.zeros(17, dtype=torch.int64)]

        def broadcast_weights(self, src=0):
            dp.broadcast_tensors(self.w, src)

        def generate(self, input_ids=None, max_new_tokens=64, **kw):
            assert kw.get("validate") is False and kw.get("eos_token_id", 0) is None
            log["generate_calls"] += 1
            seq = input_ids[:, :1] * int(self.w[0][0]) + torch.arange(max_new_tokens, dtype=torch.int64)[None]
            return types.SimpleNamespace(sequences=seq, input_flags=torch.zeros(1, dtype=torch.int32))

        def _plan_passes(self, B, tiles, S):
            return [B * tiles], [B]

    class StubRuntime:
        backend = "gloo"

        def device_of(self, local):
            return "cpu"

        def sync(self):
            log["syncs"] += 1

        def peak_mem_gib(self, device):
            return 0.0

        def build_model(self, args, cfg, rank_, device):
            assert rank_ == rank
            return StubModel(), None

        def build_batches(self, args, cfg, rank_, world_, device):
            B, tiles, S = args.batch, 2, 11
            batches = []
            for pidx in range(args.pool):
                ids = torch.full((B, S), 7, dtype=torch.int64)
                ids[:, 0] = 1000 * rank_ + 10 * pidx + torch.arange(B)
                batches.append(dict(input_ids=ids, pixel_values=torch.zeros(B * tiles, 3, 4, 4),
                                    global_mask_values=torch.zeros(B * tiles, 3, 4, 4), bboxes=[{}] * B,
                                    aspect_ratios=torch.ones(B, 2, dtype=torch.int64)))
            return batches, None, args.pool * B

    buf = io.StringIO()
    old, sys.stdout = sys.stdout, buf
    try:
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "2", "--batch", "4", "--new-tokens", "5", "--model",
                    "tiny", "--no-cpu-baseline", "--max-num-tiles", "4"], runtime=StubRuntime())
    finally:
        sys.stdout = old
    assert log["generate_calls"] == 5 and log["syncs"] == 4
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"stdout": buf.getvalue(), "threads": torch.get_num_threads()}, f)
    torch.distributed.destroy_process_group()


def test_bench_control_flow_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r1["stdout"].strip() == ""                             # ONE JSON line, on rank 0
    lines = [ln for ln in r0["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"]
    assert d["unit"] == "regions/s" and d["dtype"] == "bf16" and d["vs_baseline"] is None
    # value = whole-job regions / max-over-ranks time of the K timed steps
    assert abs(d["value"] - 2 * 3 * 4 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]
    assert d["config"]["regions_per_step_per_gpu"] == 4 and d["config"]["parallelism"].startswith("dp2")
    assert "cpu_baseline" not in d                                 # reported at N = 1 only
    cores = os.cpu_count() or 2
    assert r0["threads"] == r1["threads"] == max(1, cores // 2)    # the ranks share the host cores
