"""GPU: the multi-rank path with real engines. Two ranks share the one GPU of the test box (gloo
carries the 8-byte all-reduce; on a multi-GPU node the same code runs one rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
import numpy as np
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from tensoralloy_amd import Engine, _lib
from tensoralloy_amd.parallel import shard_range, allreduce_sum_
from tests.helpers import fcc, make_nn
nn = make_nn(["Ni"], 6.5, True, [16, 16])
n_frames = 5
lo, hi = shard_range(n_frames, rank, world)
frames = [fcc(rep=(2, 2, 2), seed=100 + f) for f in range(lo, hi)]
with Engine(nn, device=0) as eng:
    res = eng.evaluate(frames)
    local = sum(r["energy"] for r in res)
    # the device-resident batch energy (what bench.py hands to the collective) agrees with it
    t = torch.tensor([local], dtype=torch.float64)
    allreduce_sum_(t)
    fsum = np.sum([np.abs(r["forces"]).sum() for r in res])
print(json.dumps({{"rank": rank, "lo": lo, "hi": hi, "local": local, "total": float(t.item()),
                  "fsum": float(fsum)}}))
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_engine_ranks_share_the_batch(lib, tmp_path):
    """2 ranks x Engine on the one GPU: frames shard without overlap, the all-reduced energy equals
    the single-process sum over all frames."""
    from tensoralloy_amd import Engine
    from tests.helpers import fcc, make_nn
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    nn = make_nn(["Ni"], 6.5, True, [16, 16])
    with Engine(nn) as eng:
        ref = eng.evaluate([fcc(rep=(2, 2, 2), seed=100 + f) for f in range(5)])
    total = sum(r["energy"] for r in ref)
    assert sorted((o["lo"], o["hi"]) for o in outs) == [(0, 3), (3, 5)]
    for o in outs:
        assert abs(o["total"] - total) < 1e-9
        assert abs(o["local"] - sum(r["energy"] for r in ref[o["lo"]:o["hi"]])) < 1e-9


def test_bench_spawns_its_own_ranks(lib):
    """`python bench.py --gpus 2` without a launcher starts two ranks and reports n_gpus = 2."""
    env = dict(os.environ, TA_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--rep", "4", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["atoms_per_gpu"] == 256
    c5 = out["config5"]
    assert c5["frames_per_gpu"] == 64 and c5["frames_total"] == 128
    assert c5["batch_energy_check"] < 1e-6 * abs(c5["batch_energy_sum_eV"])
    assert out["value"] > 0 and c5["value"] > 0


def test_bench_with_two_ranks_on_one_gpu_fails_fast(lib):
    """nccl with `--gpus 2` on a one-GPU box: rank 1 has no device and exits; rank 0, which is
    waiting for it in the rendezvous, is terminated by the spawning process: non-zero exit with
    the reason in well under 30 s instead of c10d's 10-minute timeout."""
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two GPUs")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TA_BENCH_BACKEND"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--rep", "4", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=300)
    took = time.time() - t0
    assert p.returncode != 0
    assert took < 30, took
    assert "local rank 1 but 1 GPU(s) visible" in p.stderr, p.stderr[-2000:]
    assert "were terminated" in p.stderr


def test_update_positions_reuses_and_rebuilds_the_list(lib):
    """ta_update_positions: small moves keep the skin-padded list, results equal those of an exact
    list at the new positions; a large move rebuilds; skin = 0 always rebuilds."""
    from tensoralloy_amd import Engine, _lib
    from tests.helpers import fcc, make_nn, make_eam
    want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
    rng = np.random.RandomState(3)
    for nn in (make_nn(["Ni"], 6.5, True, [16, 16]), make_eam(["Ni"], 6.0), make_eam(["Ni"], 6.0, adp=True),
               make_eam(["Ni"], 5.0, potential=None)):
        atoms = fcc(rep=(4, 4, 4))
        with Engine(nn) as eng, Engine(nn) as exact:
            eng.set_skin(0.6)
            info = eng.set_frames([atoms])
            exact_info = exact.set_frames([atoms])
            assert info.n_pairs > exact_info.n_pairs
            pos = atoms.positions.copy()
            for step in range(4):
                pos = pos + rng.normal(0, 0.03, pos.shape)
                rebuilt = eng.update_positions(pos)
                eng.compute(want)
                got = eng.fetch(want)
                moved = atoms.copy()
                moved.positions = pos
                ref = exact.evaluate([moved])[0]
                assert abs(got["energy"][0] - ref["energy"]) < 1e-9
                assert np.abs(got["forces"] - ref["forces"]).max() < 1e-10
                assert np.abs(got["virial"][0] - ref["virial"]).max() < 1e-8
                assert np.abs(got["atomic"] - ref["atomic"]).max() < 1e-10
            builds, reuses = eng.list_stats()
            assert builds >= 1 and reuses >= 1 and builds + reuses == 5
            # one atom jumps by more than skin / 2: the list must be rebuilt
            pos[0] += 0.5
            assert eng.update_positions(pos) is True
            eng.compute(want)
            moved = atoms.copy()
            moved.positions = pos
            ref = exact.evaluate([moved])[0]
            assert abs(eng.fetch(want)["energy"][0] - ref["energy"]) < 1e-9
            # a changed cell rebuilds too
            cell = np.asarray(atoms.get_cell(complete=True)) * 1.01
            assert eng.update_positions(pos, cells=cell[None]) is True
            # skin = 0: exact list, every call rebuilds
            eng.set_skin(0.0)
            assert eng.update_positions(pos) is True
            assert eng.update_positions(pos) is True
            assert int(eng.info.n_atoms) == len(atoms)


def test_step_and_step_view_match_the_three_calls(lib):
    """`ta_step` (results copied into caller arrays) and `ta_step_view` (results left in the library's
    page-locked buffer, wrapped as arrays) against update_positions + compute + fetch, over list reuses
    and rebuilds; the view of an energy-only step hands back no force arrays."""
    from tensoralloy_amd import Engine, _lib
    from tests.helpers import fcc, make_nn
    want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
    nn = make_nn(["Ni"], 6.0, True, [16])
    atoms = fcc(rep=(4, 4, 4))
    rng = np.random.RandomState(4)
    with Engine(nn) as a, Engine(nn) as b, Engine(nn) as c:
        for eng in (a, b, c):
            eng.set_skin(0.5)
            eng.set_frames([atoms])
        pos = atoms.positions.copy()
        for step in range(6):
            pos = pos + rng.normal(0, 0.05, pos.shape)
            a.update_positions(pos)
            a.compute(want)
            ref = a.fetch(want)
            got = b.step(pos, want)
            view = c.step(pos, want, view=True)
            for res in (got, view):
                # (the angular kernels' LDS adds arrive in any order: last-bit noise between engines)
                assert abs(res["energy"][0] - ref["energy"][0]) < 1e-9
                assert np.abs(res["forces"] - ref["forces"]).max() < 1e-12
                assert np.abs(res["virial"] - ref["virial"]).max() < 1e-10
                assert np.abs(res["atomic"] - ref["atomic"]).max() < 1e-12
        assert a.list_stats() == b.list_stats() == c.list_stats()
        assert a.list_stats()[0] > 1 and a.list_stats()[1] > 0   # both reuses and rebuilds happened
        only_e = c.step(pos, _lib.TA_WANT_ENERGY, view=True)
        assert "forces" not in only_e and abs(only_e["energy"][0] - ref["energy"][0]) < 1e-9
    # a batch of uneven frames (groups of 16 atoms straddle the frame boundaries) and an EAM model
    from tests.helpers import make_eam
    frames = [fcc(rep=(3, 3, 3), seed=1), fcc(rep=(2, 3, 4), a=3.6, seed=2), fcc(rep=(3, 2, 2), seed=3)]
    for model in (nn, make_eam(["Ni"], 6.0)):
        with Engine(model) as a, Engine(model) as c:
            for eng in (a, c):
                eng.set_skin(0.5)
                eng.set_frames(frames)
            pos = np.concatenate([f.positions for f in frames])
            for step in range(3):
                pos = pos + rng.normal(0, 0.04, pos.shape)
                a.update_positions(pos)
                a.compute(want)
                ref = a.fetch(want)
                view = c.step(pos, want, view=True)
                assert view["energy"].shape == (3,) and view["virial"].shape == (3, 3, 3)
                assert np.abs(view["energy"] - ref["energy"]).max() < 1e-9
                assert np.abs(view["forces"] - ref["forces"]).max() < 1e-12
                assert np.abs(view["virial"] - ref["virial"]).max() < 1e-10
                assert np.abs(view["atomic"] - ref["atomic"]).max() < 1e-12


TRAIN_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
import numpy as np
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from tensoralloy_amd.train import Trainer
from tests.helpers import fcc, make_nn
data = np.load({data!r}, allow_pickle=True)
frames = [fcc(rep=(2, 2, 2), a=3.4 + 0.05 * k, seed=k, jitter=0.1) for k in range(4)]
student = make_nn(["Ni"], 5.0, True, [12, 12], seed=77)
tr = Trainer(student, frames, data["e"], list(data["f"]), data["s"], device=0, learning_rate=0.01)
total, terms, grad = tr.loss_and_gradient()
from tensoralloy_amd.train import allreduce_mean
mean = allreduce_mean(grad, None)
l0 = tr.step()[0]
for _ in range(3):
    l1 = tr.step()[0]
print(json.dumps({{"rank": rank, "n_local": len(tr.frames), "grad": grad.tolist(), "mean": mean.tolist(),
                  "theta": tr.theta.tolist(), "l0": l0, "l1": l1}}))
tr.close()
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_training_step_with_engines(lib, tmp_path):
    """The reference's only collective is the gradient all-reduce of its replicas
    (train/distribute_utils.py:56-81). Two ranks, each an Engine on the one GPU with its shard of the
    frames: the mean of their (analytic, energy + forces + stress) gradients equals the mean of the
    two shard gradients computed one after the other in this process, and both ranks hold the same
    weights after four Adam steps."""
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import Trainer
    from tests.helpers import fcc, make_nn
    teacher = make_nn(["Ni"], 5.0, True, [12, 12], seed=5)
    frames = [fcc(rep=(2, 2, 2), a=3.4 + 0.05 * k, seed=k, jitter=0.1) for k in range(4)]
    with Engine(teacher) as eng:
        ref = eng.evaluate(frames)
    data = tmp_path / "labels.npz"
    np.savez(data, e=np.array([r["energy"] for r in ref]), f=np.array([r["forces"] for r in ref]),
             s=np.array([r["stress"] for r in ref]))
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER.format(root=ROOT, data=str(data)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda o: o["rank"])
    assert [o["n_local"] for o in outs] == [2, 2]
    # the same two shard gradients, one after the other, in this process
    shard = []
    for lo, hi in ((0, 2), (2, 4)):
        student = make_nn(["Ni"], 5.0, True, [12, 12], seed=77)
        tr = Trainer(student, frames[lo:hi], [r["energy"] for r in ref[lo:hi]], [r["forces"] for r in ref[lo:hi]],
                     [r["stress"] for r in ref[lo:hi]], device=0)
        shard.append(tr.loss_and_gradient()[2])
        tr.close()
    mean = 0.5 * (shard[0] + shard[1])
    for o, g in zip(outs, shard):
        assert np.abs(np.array(o["grad"]) - g).max() < 1e-10 * max(1.0, np.abs(g).max())
        assert np.abs(np.array(o["mean"]) - mean).max() < 1e-10 * max(1.0, np.abs(mean).max())
    assert np.abs(np.array(outs[0]["theta"]) - np.array(outs[1]["theta"])).max() < 1e-12
    assert all(np.isfinite(o["l1"]) for o in outs)
