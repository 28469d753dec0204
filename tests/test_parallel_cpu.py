"""CPU, world_size 2 over gloo: frames shard across ranks with no overlap and the
single all-reduce of the batch energy reproduces the single-process total."""
import os
import socket

import numpy as np
import pytest


def test_shard_range_partitions_exactly():
    from tensoralloy_amd.parallel import shard_range
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[k][1] == spans[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tensoralloy_amd.parallel import shard_range, allreduce_sum_, world_from_env
    from tests.helpers import fcc, make_nn, oracle_eval
    assert world_from_env() == (rank, rank, world)
    nn = make_nn(["Ni"], 4.0, True, [8])
    lo, hi = shard_range(n_frames, rank, world)
    # per-rank batch energy: on a GPU box this is the engine's batch_energy; here the oracle
    # stands in so the sharding + collective logic is what is under test
    local = sum(oracle_eval(nn, fcc(rep=(1, 1, 2), seed=100 + f), want_forces=False)["energy"]
                for f in range(lo, hi))
    t = torch.tensor([local], dtype=torch.float64)
    allreduce_sum_(t)
    q.put((rank, lo, hi, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_energy_allreduce():
    import torch.multiprocessing as mp
    from tests.helpers import fcc, make_nn, oracle_eval
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_frames = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nn = make_nn(["Ni"], 4.0, True, [8])
    total = sum(oracle_eval(nn, fcc(rep=(1, 1, 2), seed=100 + f), want_forces=False)["energy"]
                for f in range(n_frames))
    spans = sorted((lo, hi) for _, lo, hi, _ in out)
    assert spans == [(0, 3), (3, 5)]
    for _, _, _, e in out:
        assert abs(e - total) < 1e-9


def test_bench_with_more_ranks_than_gpus_fails_fast():
    """`python bench.py --gpus 2` (nccl) where fewer than two GPUs are visible — none here — must
    exit non-zero within seconds and say why, not sit in the rendezvous."""
    import subprocess
    import sys
    import time
    from tensoralloy_amd import _lib
    _lib.build()  # the spawning process would build a stale library first: not what is timed here
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TA_BENCH_BACKEND"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--rep", "4", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0
    assert time.time() - t0 < 60
    assert "GPU(s) visible" in p.stderr and "rank exit codes" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
