"""N > 1 path on CPU: world_size-2 gloo processes exercise the sequence sharding and the one
per-step collective (all_gather of token ids)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quest_amd.parallel import gather_tokens, shard_by_cost, shard_sequences


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_seqs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = shard_sequences(n_seqs, world, rank)
        res = []
        for step in range(3):
            local = torch.arange(b, e, dtype=torch.int64) * 1000 + step  # "sampled token" of each local sequence
            res.append(gather_tokens(local, dist).tolist())
        q.put((rank, b, e, res))
    finally:
        dist.destroy_process_group()


def test_sequence_sharding_and_token_gather_world2():
    world, n_seqs = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_seqs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert [(g[1], g[2]) for g in got] == [(0, 4), (4, 8)]
    for step in range(3):
        expect = [s * 1000 + step for s in range(n_seqs)]
        for g in got:
            assert g[3][step] == expect  # every rank sees all tokens in sequence order


def test_shard_partitions():
    for n, w in [(64, 8), (7, 3), (2, 4), (0, 2)]:
        spans = [shard_sequences(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    parts = shard_by_cost([5, 1, 1, 1, 4, 4], 2)
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([5, 1, 1, 1, 4, 4][i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 1


def test_bench_launcher_starts_n_ranks_and_reports_world_size():
    """`python bench.py --gpus 2` (no WORLD_SIZE in the environment) must start 2 rank processes by itself and
    print ONE JSON line carrying n_gpus == the world size the ranks actually saw.  QUEST_BENCH_STUB=1 swaps the
    GPU workload for a trivial CPU step (gloo), so the launcher, rendezvous, per-step token gather,
    max-over-ranks timing and JSON contract run on a CPU-only box."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["QUEST_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--config", "5"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    # ... and nothing else: whatever libraries print (RCCL writes a version banner to the process's stdout from C when
    # its first communicator is created) is sent to stderr, the driver parses stdout
    assert r.stdout.strip() == lines[0], r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size_seen"] == 2 and out["config"]["gpus_requested"] == 2
    assert out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak" and out["data"] == "stub"
    assert out["config"]["sequences_per_gpu"] == 8 and out["config"]["num_kv_heads"] == 8
    assert "configs[4]" in out["config"]["workload"]
    assert out["value"] > 0 and out["ms_per_step_without_token_gather"] is not None
    # value = tokens of ALL ranks / max-over-ranks time
    assert abs(out["value"] - 2 * 8 * 4 / (out["ms_per_step"] * 4e-3)) / out["value"] < 1e-6


def test_bench_default_headline_reports_configs4_side_measurement_under_n_ranks():
    """under N > 1 the default (cfg 3) headline run must ALSO measure BASELINE configs[4]'s shape -- 8 GQA
    sequences per GPU on every rank, each step ending with the all_gather of the token ids -- and report it as a side
    object (aggregate tokens/s over all ranks); the headline itself stays one cfg-3 sequence per GPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["QUEST_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--side-steps", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["survey_cfg"] == 3 and out["config"]["sequences_per_gpu"] == 1
    assert "batched_8seq" not in out and "side_error" not in out
    side = out["cfg5_8seq_gqa"]
    assert side["n_gpus"] == 2 and side["sequences_per_gpu"] == 8 and side["steps"] == 3 and "configs[4]" in side["workload"]
    assert side["ms_per_step_without_token_gather"] is not None
    assert abs(side["tokens_per_s"] - 2 * 8 * 3 / (side["ms_per_step"] * 3e-3)) / side["tokens_per_s"] < 1e-6
    assert "aggregate over all ranks" in side["note"]


def test_bench_launcher_at_the_width_the_driver_uses_world_size_8():
    """the launcher stub at world size 8 -- port handling, the poll loop over 8 children, the headline's
    rendezvous and the side measurement's (every rank decodes configs[4]'s 8 sequences, 64 token ids gathered per step) at
    the width of the driver's SCALE run.  CPU stub workload (gloo): no scaling figure, only the control flow."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["QUEST_BENCH_STUB"] = "1"
    env["OMP_NUM_THREADS"] = "1"  # 8 ranks on the 8 cores of the build container
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--side-steps", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["world_size_seen"] == 8 and out["config"]["gpus_requested"] == 8
    assert out["config"]["sequences_per_gpu"] == 1 and "side_error" not in out
    assert abs(out["value"] - 8 * 1 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-6
    side = out["cfg5_8seq_gqa"]
    assert side["n_gpus"] == 8 and side["sequences_per_gpu"] == 8 and side["steps"] == 2
    assert abs(side["tokens_per_s"] - 8 * 8 * 2 / (side["ms_per_step"] * 2e-3)) / side["tokens_per_s"] < 1e-6


def test_bench_side_measurement_that_one_rank_cannot_build_leaves_every_rank_together():
    """a rank that cannot build the side configuration's workload (memory) must not leave its peers inside the
    side measurement's collectives: every rank learns it before the first collective, all skip the side object, the
    headline line is printed with the error, exit code 0 -- promptly."""
    import json
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["QUEST_BENCH_STUB"] = "1"
    env["QUEST_BENCH_STUB_BUILD_FAIL"] = "1:1"  # rank 1, in the side measurement
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--side-steps", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 60
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "error" in out["cfg5_8seq_gqa"] and "failed to build its workload" in out["cfg5_8seq_gqa"]["error"]


def test_bench_launcher_returns_promptly_when_a_peer_dies_before_rendezvous():
    """Rank 1 exits (code 3) before init_process_group: the launcher must notice, terminate rank 0 (which would
    otherwise sit in the rendezvous until the process-group timeout), print one JSON error line and return non-zero
    well inside 30 s -- an 8-GPU SCALE point must never hang on a bad LOCAL_RANK / missing GPU / OOM of one rank."""
    import json
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["QUEST_BENCH_STUB"] = "1"
    env["QUEST_BENCH_STUB_FAIL_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    took = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert took < 30, took
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    err = json.loads(lines[0])
    assert err["failed_rank"] == 1 and err["exit_code"] == 3 and err["n_gpus"] == 2 and "error" in err


def test_bench_launcher_reports_a_failing_rank0():
    """Rank 0 itself dies: still one JSON error line and a non-zero exit code, and rank 1 does not linger."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["QUEST_BENCH_STUB"] = "1"
    env["QUEST_BENCH_STUB_FAIL_RANK"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["failed_rank"] == 0


def test_bench_default_single_gpu_line_names_its_pool_layout_and_carries_the_nhd_figures_beside_it():
    """Round 6: the default pool layout of bench.py is this build's row-rotated NHD.  The line says so (config.kv_layout) and
    the default N = 1 run measures the headline and both 8-sequence side configurations on the REFERENCE's NHD pool as
    well (a headline on another layout carries the NHD figure beside it); `--layout NHD` runs the reference's
    layout and needs no such objects.  CPU stub: the control flow and the JSON contract, not the numbers."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["QUEST_BENCH_STUB"] = "1"

    def run(*extra):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--side-steps", "3",
                            "--no-cpu-baseline", *extra], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and r.stdout.strip() == lines[0], r.stdout
        return json.loads(lines[0])

    out = run()
    assert out["n_gpus"] == 1 and out["config"]["kv_layout"] == "NHD_ROT" and "side_error" not in out
    for key, layout, seqs in (("batched_8seq", "NHD_ROT", 8), ("cfg5_8seq_gqa", "NHD_ROT", 8), ("reference_layout_nhd", "NHD", 1),
                              ("batched_8seq_nhd", "NHD", 8), ("cfg5_8seq_gqa_nhd", "NHD", 8)):
        side = out[key]
        assert "error" not in side, (key, side)
        assert side["kv_layout"] == layout and side["sequences_per_gpu"] == seqs and side["steps"] == 3, (key, side)
    assert "configs[4]" in out["cfg5_8seq_gqa_nhd"]["workload"] and "configs[2]" in out["reference_layout_nhd"]["workload"]
    out = run("--layout", "NHD")
    assert out["config"]["kv_layout"] == "NHD" and "batched_8seq" in out and "cfg5_8seq_gqa" in out
    assert not any(k.endswith("_nhd") for k in out)
    out = run("--no-side")
    assert not any(k in out for k in ("batched_8seq", "cfg5_8seq_gqa", "reference_layout_nhd"))


def test_bench_config_labels_follow_the_arguments():
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse([])
    assert (a.seqlen, a.token_budget, a.heads, a.kv_heads, a.seqs_per_gpu) == (32768, 2048, 32, 32, 1)
    assert a.workload_label.startswith("BASELINE configs[2]")
    a = bench.parse(["--config", "4"])
    assert (a.seqlen, a.token_budget, a.kv_heads) == (131072, 4096, 8) and "configs[3]" in a.workload_label
    a = bench.parse(["--seqlen", "4096"])
    assert a.workload_label.startswith("custom") and "seqlen=4096" in a.workload_label
    b = bench.bytes_per_layer(bench.parse(["--config", "3"]))
    assert b["attn"] == 128 * 16 * 2 * 32 * 128 * 2 + 32 * 128 * 2 + 32 * 127 * 4 + 32 * 128 * 2 and b["sparse"]
    assert not bench.bytes_per_layer(bench.parse(["--config", "2"]))["sparse"]
