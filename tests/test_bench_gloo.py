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
    for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    import bench_stub
    from gar_amd import dp
    log = bench_stub.LOG
    StubRuntime = bench_stub.StubRuntime
    gathered = []
    real_gather = dp.gather_captions
    dp.gather_captions = lambda ids, dst=0: (gathered.append(real_gather(ids, dst=dst)), gathered[-1])[1]

    buf = io.StringIO()
    old, sys.stdout = sys.stdout, buf
    try:
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "2", "--batch", "4", "--new-tokens", "5", "--model",
                    "tiny", "--no-cpu-baseline", "--max-num-tiles", "4"], runtime=StubRuntime())
    finally:
        sys.stdout = old
    assert log["generate_calls"] == 5 and log["syncs"] == 6          # 2 around the weight broadcast + 4 around the timed region
    last = gathered[-1]
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"stdout": buf.getvalue(), "threads": torch.get_num_threads(), "gathers": len(gathered),
                   # the stub's first token of a caption is 3 x the region index it was asked for
                   "last_gather_regions": None if last is None else [[int(t) // 3 for t in part[:, 0]] for part in last]}, f)
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
    # the rank proof: what the process group was, every rank's own clock, the broadcast
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo" and len(d["per_rank_ms_per_step"]) == 2
    assert abs(max(d["per_rank_ms_per_step"]) - d["ms_per_step"]) < 1e-6 * d["ms_per_step"]
    # one collective per dtype arena (float32 + int64), bytes = the arenas (3000 * 4 and 17 * 8: each a single tensor at offset 0)
    assert d["weight_broadcast"]["bytes"] == 3000 * 4 + 17 * 8 and d["weight_broadcast"]["seconds"] >= 0
    assert d["weight_broadcast"]["collectives"] == 2
    cores = os.cpu_count() or 2
    assert r0["threads"] == r1["threads"] == max(1, cores // 2)    # the ranks share the host cores


def test_plain_python_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain interpreter (no torchrun environment — the shape of the driver's N = 1 command):
    bench.py starts its two ranks under torch.distributed.run itself; the JSON line proves them."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch",
                          "3", "--new-tokens", "4", "--model", "tiny", "--no-cpu-baseline", "--max-num-tiles", "4", "--runtime",
                          "bench_stub:StubRuntime"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["steps"] == 2
    assert len(d["per_rank_ms_per_step"]) == 2 and d["config"]["parallelism"].startswith("dp2")
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    # a launcher whose world size disagrees with --gpus is refused, not silently re-interpreted
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--runtime",
                          "bench_stub:StubRuntime", "--no-cpu-baseline"], env=env2, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_bench_control_flow_world8(tmp_path):
    """the shape of the driver's SCALE run at N = 8 (CPU stub, gloo): eight ranks seen, ONE broadcast per weight arena, every step's
    [8 x B, new_tokens] ids on rank 0, and region i served by rank i % 8. No hardware is involved: the
    1 -> 8 GPU curve itself stays unmeasured from this container."""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert all(o["stdout"].strip() == "" for o in outs[1:])
    lines = [ln for ln in outs[0]["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["backend"] == "gloo" and len(d["per_rank_ms_per_step"]) == 8
    assert d["weight_broadcast"]["collectives"] == 2 and d["weight_broadcast"]["bytes"] == 3000 * 4 + 17 * 8
    assert d["caption_gather"]["rank0_ids_shape"] == [8 * 4, 5] and d["config"]["parallelism"].startswith("dp8")
    assert abs(d["value"] - 8 * 3 * 4 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]
    assert all(o["gathers"] == 5 for o in outs)                    # one gather per step (2 warm-up + 3 timed), on every rank
    assert all(o["last_gather_regions"] is None for o in outs[1:])
    parts = outs[0]["last_gather_regions"]
    assert len(parts) == 8
    seen = set()
    for r, regions in enumerate(parts):
        assert len(regions) == 4 and all(i % 8 == r for i in regions), (r, regions)
        seen |= set(regions)
    assert len(seen) == 32


def test_dry_run_prints_the_launch_plan():
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["dry_run"] and d["n_gpus"] == 8 and len(d["ranks"]) == 8 and "--nproc-per-node=8" in d["launch"]
    assert [r["device"] for r in d["ranks"]] == [f"cuda:{i}" for i in range(8)]
    assert d["ranks"][5]["regions"].startswith("i % 8 == 5: 5, 13, 21")
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
