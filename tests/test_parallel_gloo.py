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
