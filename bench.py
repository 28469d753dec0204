#!/usr/bin/env python3
"""
bench.py — atom-steps/s of the hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (pair geometry -> G2/G4 descriptors ->
per-atom MLP -> energy -> analytic forces + virial) over this rank's resident
batch of frames. Workload at N = 1 = BASELINE.json configs[1]: 4000-atom Ni fcc
supercell (a = 3.524 A, 10x10x10 cells, N(0, 0.05 A) jitter, RandomState(611)),
rcut = acut = 6.5 A, G2 eta {0.05,4,20,80} x omega {0}, G4 beta {0.005} x
gamma {1,-1} x zeta {1,4} (D = 8), MLP 8-64-64-1 softplus, fp64,
energy + forces + virial. `value`: inputs (positions, cells, neighbour list)
resident in HBM before the timed region. Reported beside it, never instead:
`transfer_inclusive` (H2D of the positions and D2H of energy / forces / virial
inside every step; SURVEY 8(d) protocol) and the neighbour-list build.

N > 1: one process per GPU. Started by a launcher (torchrun sets WORLD_SIZE /
RANK / LOCAL_RANK / MASTER_*) the process is one rank; started plainly as
`python bench.py --gpus N` it spawns the N ranks itself BEFORE it touches the
GPU and relays rank 0's line. Every rank owns `--frames-per-gpu` independent
frames (weak scaling); the only collective is one RCCL all-reduce of the 8-byte
batch energy per step. `config5` in the same line: BASELINE.json configs[4],
64 frames per GPU (512 over 8), same protocol; it runs BEFORE the headline
region (about 100 ms of work: the GPU is at its working clocks when the short
single-frame region starts; W warm-up and K timed steps as asked for, both).

Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md §Chip-level parameters
# dense fp64 peak, vector and matrix alike: AMD's MI355X figure, 78.6 TFLOP/s = 32 flop/clk/SIMD x
# 1024 SIMDs x 2.4 GHz (v_mfma_f64_16x16x4_f64 = 2048 flop in 64 cycles; v_fma_f64 = 128 flop per
# wavefront instruction, 4 cycles; the guide's table has no fp64 row)
FP64_PEAK_TFLOPS = 78.6
FLOP_PER_TRIPLE = 150.0  # forward + backward of one contributing triple
E_TOL, F_TOL = 1e-6, 1e-5  # north_star parity tolerances (eV, eV/A)
CONFIG5_FRAMES_PER_GPU = 64


def host_cores():
    """Threads for the CPU baseline: the cores this process may actually use
    (affinity mask, capped by the cgroup CPU quota when there is one)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fp:
            quota, period = fp.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = int(env)
    return max(1, min(n, 64))


def ni_frame(seed, rep=10, a=3.524, jitter=0.05):
    from tensoralloy_amd import Atoms
    base = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]]) * a
    g = np.arange(rep)
    cells = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 1, 3) * a
    pts = (cells + base[None]).reshape(-1, 3)
    pts = pts + np.random.RandomState(seed).normal(0.0, jitter, pts.shape)
    return Atoms(symbols=["Ni"] * len(pts), positions=pts, cell=np.eye(3) * a * rep, pbc=True)


def ni_model():
    from tensoralloy_amd import AtomicNN, SymmetryFunction, UniversalTransformer
    clf = UniversalTransformer(["Ni"], rcut=6.5, acut=6.5, angular=True)
    nn = AtomicNN(["Ni"], SymmetryFunction(["Ni"]), hidden_sizes=[64, 64], activation="softplus",
                  minmax_scale=False, use_resnet_dt=False, use_atomic_static_energy=True,
                  export_properties=("energy", "forces", "stress"))
    nn.attach_transformer(clf)
    nn.initialize(seed=611)
    return nn


def oracle_sfmodel(nn):
    """Same model for the CPU oracle (checker / cpu_baseline only)."""
    from oracle.sf import SFModel
    d = nn.descriptor.as_dict()
    clf = nn.transformer
    return SFModel(nn.elements, clf.rcut, acut=clf.acut, angular=clf.angular, eta=d["eta"],
                   omega=d["omega"], beta=d["beta"], gamma=d["gamma"], zeta=d["zeta"],
                   cutoff_function=d["cutoff_function"], hidden_sizes=nn.hidden_sizes,
                   activation=nn._activation, weights=nn.weights,
                   use_resnet_dt=nn._use_resnet_dt, minmax=None)


def source_stamp():
    """sha256 over the kernel sources + this file: ties committed PMC numbers to the code they were
    measured on (scripts/summarize_profile.py writes the same stamp into profiles/pmc_traffic.json)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "tensoralloy_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp")))
    files += [os.path.join(ROOT, "include", "tensoralloy_amd.h"), os.path.join(ROOT, "bench.py")]
    for f in files:
        with open(f, "rb") as fp:
            h.update(os.path.basename(f).encode() + b"\0" + fp.read())
    return h.hexdigest()[:16]


RENDEZVOUS_TIMEOUT_S = 120  # init_process_group / store timeout: a missing rank fails the job in
                            # minutes, not after c10d's default 10 (past the driver's 600 s limit)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this
    one (which never touches the GPU, and never re-executes itself) and relay rank 0's JSON line.
    All children are polled: the first one that exits non-zero ends the job within seconds — the
    others are terminated (they may be sitting in the rendezvous waiting for it) and this process
    exits non-zero with the codes on stderr."""
    import tempfile
    import threading
    from tensoralloy_amd import _lib
    _lib.build()  # once, here: the children then find the library up to date
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # Several processes share device memory handles through RCCL: this image's host driver only
        # supports dmabuf IPC, and with the legacy mode (the runtime's default) RCCL's set-up fails in
        # `hipIpcGetMemHandle: invalid argument`. The image exports the variable already; it is set
        # here only if the caller's environment lost it.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    # rank 0's stdout is drained by a thread so that polling never blocks on a full pipe
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    codes = [p.returncode for p in procs]
    out0 = b"".join(c for c in chunks if c)
    lines = [l for l in out0.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    if failed is not None or any(codes) or not lines:
        if failed is not None:
            sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}; the other ranks "
                             f"were terminated\n")
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        raise SystemExit(max([c for c in codes if c and c > 0] + [1]))
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()


class Collective:
    """The one exchange step of the path: all-reduce of the 8-byte batch energy, double-buffered,
    ordered against the kernels by stream (no host synchronisation per step)."""

    def __init__(self, eng, use_dist, torch, dist, local_rank):
        self.eng, self.use, self.torch, self.dist = eng, use_dist, torch, dist
        self.inflight = [None, None]
        if use_dist:
            self.buf = torch.zeros(2, dtype=torch.float64, device=f"cuda:{local_rank}")
            self.slots = [self.buf[0:1], self.buf[1:2]]
            self.ptrs = [s.data_ptr() for s in self.slots]

    def step(self, k, want):
        if not self.use:
            self.eng.compute(want)
            return None
        if self.inflight[k % 2] is not None:
            self.inflight[k % 2].wait()  # stream-level wait before the slot is overwritten
        # frame_reduce writes the batch energy straight into this step's slot of the torch buffer,
        # then ONE all-reduce of 8 bytes reduces it in place
        self.eng.set_batch_energy_target(self.ptrs[k % 2])
        self.eng.compute(want)
        self.inflight[k % 2] = self.dist.all_reduce(self.slots[k % 2], op=self.dist.ReduceOp.SUM, async_op=True)
        return self.inflight[k % 2]

    def sync(self):
        self.eng.synchronize()
        if self.use:
            self.torch.cuda.synchronize()
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def run(self, steps, warmup, want):
        """`warmup` untimed then `steps` timed steps, barrier + synchronize on both sides;
        returns the MAX over ranks of the elapsed seconds and the last reduced batch energy."""
        for w in [self.step(k, want) for k in range(warmup)]:
            if w is not None:
                w.wait()
        self.sync()
        t0 = time.perf_counter()
        for w in [self.step(k, want) for k in range(steps)]:
            if w is not None:
                w.wait()
        self.sync()
        elapsed = time.perf_counter() - t0
        esum = None
        if self.use:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.buf.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
            esum = float(self.slots[(steps - 1) % 2].item()) if steps else None
            self.inflight = [None, None]
        return elapsed, esum

    def total(self, x):
        if not self.use:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.buf.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames-per-gpu", type=int, default=1)
    ap.add_argument("--rep", type=int, default=10, help="fcc cells per edge (10 -> 4000 atoms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--cpu-evals", type=int, default=40)  # ~10 s of 16-thread CPU work
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)

    # the driver reads ONE JSON line from stdout: native libraries print there too (RCCL writes a
    # version banner on initialisation, gloo its connection report), so everything else that lands
    # on fd 1 during the run is sent to stderr and the line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from tensoralloy_amd.parallel import world_from_env
    rank, local_rank, world = world_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(torchrun --nproc-per-node {args.gpus}), or no launcher at all")
    dist = torch = None
    # TA_BENCH_FORCE_DIST=1 rehearses the multi-rank code path with a single rank
    use_dist = world > 1 or os.environ.get("TA_BENCH_FORCE_DIST") == "1"
    backend = None
    if use_dist:
        # torch first: its bundled HIP runtime must be the one the process shares
        import torch
        import torch.distributed as dist
        # TA_BENCH_BACKEND=gloo rehearses several ranks on a box with fewer GPUs than ranks (ranks
        # share devices; the all-reduce then goes through gloo): the driver's runs use nccl = RCCL
        backend = os.environ.get("TA_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        elif local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local rank {local_rank} but {torch.cuda.device_count()} GPU(s) "
                             f"visible (TA_BENCH_BACKEND=gloo lets ranks share a device)")
        torch.cuda.set_device(local_rank)
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("TA_BENCH_RENDEZVOUS_TIMEOUT", RENDEZVOUS_TIMEOUT_S)))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        world = dist.get_world_size()  # as observed, not as asked for

    from tensoralloy_amd import Engine, _lib
    _lib.build()
    nn = ni_model()
    eng = Engine(nn, device=local_rank)
    fpg = args.frames_per_gpu
    frames = [ni_frame(611 + rank * fpg + k, rep=args.rep) for k in range(fpg)]
    t0 = time.perf_counter()
    info = eng.set_frames(frames)
    t_nl = time.perf_counter() - t0
    # second call: buffers exist, this is the per-MD-step cost of a new neighbour list
    t0 = time.perf_counter()
    info = eng.set_frames(frames)
    t_set = time.perf_counter() - t0
    want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
    n_atoms, P, T = int(info.n_atoms), int(info.n_pairs), int(info.n_triples)
    D = int(info.descriptor_dim)
    nl_info = {"first_call_s": t_nl, "steady_call_s": t_set,
               "neighbor_list_on_device": bool(info.nl_on_device),
               "neighbor_list_ms": info.nl_ms, "c_abi_ms": info.set_frames_ms}

    if use_dist:
        # kernels go onto torch's current stream so the collective is ordered behind them by stream
        # semantics. A dedicated non-blocking stream measured the same with nccl and makes gloo's
        # CUDA path block for milliseconds per step (TA_BENCH_SIDE_STREAM=1 selects it)
        if os.environ.get("TA_BENCH_SIDE_STREAM") == "1":
            torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    coll = Collective(eng, use_dist, torch, dist, local_rank)

    # (config 5 runs first: 64 frames per step keep the GPU busy for ~100 ms, so the short single-frame
    # region behind it, 20 steps are 2.6 ms, does not start on a GPU that is still raising its clocks)
    # ---- BASELINE.json configs[4]: 64 independent frames per GPU, sum of the energies all-reduced ---
    config5 = None
    preheat = None
    if not args.no_config5:
        t_pre = time.perf_counter()
        f5 = CONFIG5_FRAMES_PER_GPU
        frames5 = [ni_frame(611 + rank * f5 + k, rep=args.rep) for k in range(f5)]
        info5 = eng.set_frames(frames5)
        # six warm-up steps (29 ms): this block is the first sustained work of the process, and two were
        # not enough for the GPU to reach its working clocks (54.7 against 55.3 M, same session)
        k5, w5 = max(5, min(args.steps, 20)), 6
        el5, esum = coll.run(k5, w5, want)
        atoms5 = coll.total(int(info5.n_atoms))
        local_sum = float(eng.fetch(_lib.TA_WANT_ENERGY)["energy"].sum())
        check = coll.total(local_sum)
        if os.environ.get("TA_BENCH_DEBUG"):
            sys.stderr.write(f"[rank {rank}] config5 esum={esum} local_sum={local_sum} check={check} "
                             f"atoms5={atoms5}\n")
        config5 = {"workload": f"{f5} independent {int(info5.n_atoms) // f5}-atom Ni frames per GPU "
                               f"(seeds 611+rank*{f5}+k), {f5 * world} frames in all, one all-reduce "
                               f"of the batch energy per step",
                   "frames_per_gpu": f5, "frames_total": f5 * world, "value": atoms5 * k5 / el5,
                   "unit": "atom-steps/s", "steps": k5, "warmup": w5, "ms_per_step": el5 / k5 * 1e3,
                   "us_per_frame": el5 / k5 / f5 * 1e6, "pairs_per_gpu": int(info5.n_pairs),
                   "batch_energy_sum_eV": esum if esum is not None else local_sum,
                   "batch_energy_check": abs((esum if esum is not None else local_sum) - check)}
        del frames5
        preheat = {"what": f"config5 block ({w5} + {k5} steps of {f5} frames) runs before the headline region",
                   "gpu_busy_ms": el5 / k5 * (k5 + w5) * 1e3, "wall_ms": (time.perf_counter() - t_pre) * 1e3}
        info = eng.set_frames(frames)  # back to the headline batch
        if use_dist:
            eng.set_batch_energy_target(None)

    # ---- the headline figure: `warmup` untimed, EXACTLY `steps` timed -------------------------------
    elapsed, _ = coll.run(args.steps, args.warmup, want)
    total_atoms = coll.total(n_atoms)
    value = total_atoms * args.steps / elapsed
    res = eng.fetch(want) if rank == 0 else None

    out = None
    if rank == 0:
        # ---- per-kernel durations with HIP events on the engine's stream ----
        n_ev = max(5, min(args.steps, 20))
        ev_total_ms, slots_raw = eng.time_compute(want, 2, n_ev)
        # the slot pass records an event pair around every launch, which stretches the step by ~9 %
        # (sum of the slots 131 us against 120 us for the same launches without the events): the
        # per-kernel durations reported and used below are the slots scaled so that they sum to the
        # un-instrumented step of the same call; the raw slots stay in `kernel_ms_instrumented`
        slot_sum = sum(slots_raw.values())
        scale = (ev_total_ms / n_ev) / slot_sum if slot_sum > 0 else 1.0
        slots = {k: v * scale for k, v in slots_raw.items()}
        fwd_ms, bwd_ms = slots["g4_forward"], slots["backward"]
        # SURVEY §8(d) prices one pass over the packed records at 32 B / pair + 60 B / triple
        dom_name = "backward_v2_kernel<1,2,2,12,true>"
        bwd_bytes = 32.0 * P + 60.0 * T
        achieved = bwd_bytes / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0
        eval_bytes = 2.0 * (32.0 * P + 60.0 * T) + n_atoms * 8.0 * (3 * D + 4) + 72.0 * fpg
        onthefly_bytes = 2.0 * 32.0 * P + n_atoms * 8.0 * (3 * D + 4)  # SURVEY 8(d) secondary formula
        # HBM bytes and VALU instructions per launch: PMC counters cannot be read from inside this
        # process, they come from the committed rocprofv3 passes of this same command
        # (scripts/profile_bench.sh; FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled
        # as MI355X_MICROARCH.md prescribes for 16 B/lane reads on gfx950) and are used only while
        # the stamp in that file matches the sources this run was built from
        traffic = valu = step_pmc_bytes = None
        pmc_note = "profiles/pmc_traffic.json absent"
        stamp = source_stamp()
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and fpg == 1 and args.rep == 10:
            try:
                with open(tpath) as fp:
                    pmc = json.load(fp)
                meta = pmc.get("_stamp", {})
                if meta.get("source_sha") == stamp:
                    step_kernels = ("g4_forward_v2_kernel", "mlp_all_kernel", "mlp_kernel", "backward_v2_kernel",
                                    "force_gather_kernel", "frame_reduce_kernel")
                    step_pmc_bytes = 0.0
                    for name, rec in pmc.items():
                        if name.startswith("backward_v2_kernel"):
                            traffic = rec["hbm_bytes_per_launch"]
                            valu = rec.get("valu_wave_insts_per_launch")
                        if name.startswith(step_kernels):
                            step_pmc_bytes += rec["hbm_bytes_per_launch"]
                    pmc_note = f"profiles/pmc_traffic.json, commit {meta.get('commit')}, sources {stamp}"
                else:
                    pmc_note = (f"profiles/pmc_traffic.json was measured on sources {meta.get('source_sha')}, "
                                f"this run is {stamp}: counters dropped")
            except Exception as exc:  # noqa: BLE001
                pmc_note = f"profiles/pmc_traffic.json unreadable: {exc}"
        # What bounds these kernels: FP64 VALU issue. One wavefront instruction occupies a SIMD for
        # 4 cycles (16 lanes/cycle for fp64); 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock.
        valu_obj = None
        if valu and bwd_ms > 0:
            peak_issue = 256 * 4 * 2.4e9 / 4.0            # wavefront instructions per second
            valu_obj = {"wave_insts_per_launch": valu, "achieved_Ginst_per_s": valu / (bwd_ms * 1e-3) / 1e9,
                        "peak_Ginst_per_s": peak_issue / 1e9,
                        "frac": valu / (bwd_ms * 1e-3) / peak_issue, "source": "SQ_INSTS_VALU, " + pmc_note}
        # useful arithmetic: triples whose three sides are all inside acut, 150 flop each for
        # forward + backward, over the two angular kernels' time
        n_contrib = eng.count_contributing_triples()
        useful_flop = FLOP_PER_TRIPLE * n_contrib
        ang_ms = fwd_ms + bwd_ms
        useful_tf = useful_flop / (ang_ms * 1e-3) / 1e12 if ang_ms > 0 else 0.0
        copy_gbs = eng.measure_hbm_copy(1 << 30, 10)  # achievable copy rate on this box, SURVEY 8(d)
        # MFMA utilisation of the batched per-atom MLP (north_star): v_mfma_f64_16x16x4_f64 count of
        # one launch (forward + backward-to-inputs, padded tiles) x 2048 flop, over the kernel's
        # HIP-event duration, against the dense fp64 matrix peak
        # One-element models of the usual shapes run the transposed kernels (mlp_quad_kernel below 1024
        # tiles, mlp_wave_kernel from there): per 16-atom tile the first layer takes ceil(D / 4)
        # k-steps per output tile, hidden-to-hidden layers their full K, the scalar output layer is
        # VALU work, and the backward sweep mirrors the hidden layers plus one tile row of dE/dG.
        hidden = list(nn.hidden_sizes[nn.elements[0]])
        pad = lambda v: (v + 15) // 16 * 16
        tiles = lambda v: pad(v) // 16
        per_tile = tiles(hidden[0]) * ((D + 3) // 4)                                   # forward, first layer
        per_tile += sum(tiles(hidden[l]) * (pad(hidden[l - 1]) // 4) for l in range(1, len(hidden)))
        per_tile += sum(tiles(hidden[l - 1]) * (pad(hidden[l]) // 4) for l in range(1, len(hidden)))  # backward
        per_tile += tiles(D) * (pad(hidden[0]) // 4)                                   # dE/dG
        n_tiles = ((n_atoms // fpg + 15) // 16) * fpg
        n_mfma = per_tile * n_tiles
        mlp_ms = slots.get("mlp", 0.0)
        mlp_mfma = None
        if mlp_ms > 0:
            tf = n_mfma * 2048.0 / (mlp_ms * 1e-3) / 1e12
            transposed = n_tiles > 256       # below: the generic 16-row tile kernel (padded K and N)
            if not transposed:
                sizes = [D] + hidden + [1]
                n_mfma = n_tiles * 2 * sum((pad(sizes[l]) // 4) * tiles(sizes[l + 1]) for l in range(len(sizes) - 1))
                tf = n_mfma * 2048.0 / (mlp_ms * 1e-3) / 1e12
            mlp_mfma = {"kernel": "mlp_kernel (16-row tiles)" if not transposed else
                        ("mlp_quad_kernel" if n_tiles < 1024 else "mlp_wave_kernel"),
                        "mfma_insts_per_launch": n_mfma,
                        "flop_per_launch": n_mfma * 2048.0, "kernel_ms": mlp_ms, "achieved": tf,
                        "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                        "note": "16-atom tiles; one frame is 250 workgroups on 256 CUs: bound by launch and "
                                "the latency of the dependent GEMM phases of a tile, not by the matrix pipe (a "
                                "v_mfma_f64_16x16x4 is 64 cycles); launches of more than 256 tiles run the "
                                "transposed kernels (no padding of K = D and N = 1)"}
        hbm_formula = {"bound": "hbm (SURVEY 8(d) primary formula on packed triple records; NOT a bound "
                                "for these kernels: the triples are generated in LDS and never read from HBM)",
                       "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "peak_measured_copy": copy_gbs,
                       "frac_of_measured_copy": achieved / copy_gbs if copy_gbs > 0 else None,
                       "algorithmic_bytes_per_launch": bwd_bytes, "kernel_ms": bwd_ms,
                       "whole_eval_bytes": eval_bytes,
                       "whole_eval_frac": eval_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                       "on_the_fly_bytes_per_step": onthefly_bytes}
        roofline = {"bound": "fp64_valu", "kernel": "g4_forward_v2_kernel + backward_v2_kernel (the two "
                                                     "angular kernels; dominant: " + dom_name + ")",
                    "achieved": useful_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": useful_tf / FP64_PEAK_TFLOPS,
                    "traffic": traffic,
                    "contributing_triples": n_contrib, "all_triples": T, "flop_per_triple": FLOP_PER_TRIPLE,
                    "useful_flop_per_step": useful_flop, "angular_kernels_ms": ang_ms,
                    "kernel_ms": {"g4_forward": fwd_ms, "backward": bwd_ms},
                    "pmc_bytes_per_step": step_pmc_bytes,
                    "pmc_source": pmc_note,
                    "valu_issue": valu_obj,
                    "hbm_packed_formula": hbm_formula,
                    "mlp_mfma": mlp_mfma,
                    "note": "useful arithmetic = triples with r_ij, r_ik, r_jk < acut x 150 flop (forward + "
                            "backward, SURVEY 8(d)) over the two angular kernels' time, against the fp64 "
                            "vector peak at 2.4 GHz; `traffic` = PMC HBM bytes per launch of the dominant "
                            "kernel; `hbm_packed_formula` keeps SURVEY's primary figure for continuity "
                            "(frac > 1: not a bound)"}

        # ---- SURVEY 8(d) protocol figure: coordinates in, results out, inside every step -------------
        # (a) list reused under a Verlet skin of 0.5 A (same atoms, MD-sized moves); (b) a new exact
        # list every step, which is what the reference does per `calculate`
        inclusive = None
        if world == 1 and fpg == 1:
            pos = np.ascontiguousarray(frames[0].positions)
            n_inc = max(10, min(args.steps, 50))

            def loop(n, view=True):
                eng.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    # ta_step_view = ta_update_positions + ta_compute + results in the library's page-locked
                    # host buffer, wrapped as arrays (view=False: ta_step, copied on into caller arrays)
                    eng.step(pos, want, view=view)
                return (time.perf_counter() - t) / n
            loop(3)
            t_rebuild = loop(n_inc)
            eng.set_skin(0.5)
            info_s = eng.set_frames(frames)
            loop(3)
            t_reuse = loop(n_inc)
            t_reuse_copy = loop(n_inc, view=False)
            builds, reuses = eng.list_stats()
            eng.set_skin(0.0)
            eng.set_frames(frames)
            inclusive = {"value": n_atoms / t_reuse, "unit": "atom-steps/s", "ms_per_step": t_reuse * 1e3,
                         "steps": n_inc, "skin_A": 0.5, "pairs_in_list": int(info_s.n_pairs),
                         "h2d_bytes_per_step": 24 * n_atoms, "d2h_bytes_per_step": 8 * (4 * n_atoms + 10),
                         "lists_reused": reuses,
                         "new_list_every_step": {"value": n_atoms / t_rebuild, "ms_per_step": t_rebuild * 1e3},
                         "copied_into_caller_arrays": {"value": n_atoms / t_reuse_copy,
                                                       "ms_per_step": t_reuse_copy * 1e3},
                         "note": "H2D positions + ta_compute + D2H energy/forces/virial/atomic, host-timed, "
                                 "one synchronisation per step; results left in the library's page-locked host "
                                 "buffer (ta_step_view); `copied_into_caller_arrays` = ta_step"}

        # ---- parity gate + CPU baseline (oracle = checker, timed on host cores) ----
        cpu = None
        if not args.no_cpu_baseline:
            # the baseline is TIMED at N = 1 only; at N > 1 one oracle evaluation still gates parity
            time_cpu = world == 1
            from oracle import csf
            m = oracle_sfmodel(nn)
            a = frames[0]
            prep = csf.prepare(m, a.get_chemical_symbols(), a.positions,
                               np.asarray(a.get_cell(complete=True)), a.pbc)
            cm = csf.make_cmodel(m)
            cores = host_cores()
            ref = csf.run(m, prep, True, cores, cm)  # warm-up + parity reference
            n0 = len(a)
            dE = abs(ref["energy"] - float(res["energy"][0]))
            dF = float(np.abs(ref["forces"] - res["forces"][:n0]).max())
            dW = float(np.abs(ref["virial"] - res["virial"][0]).max())
            if not (dE < E_TOL and dF < F_TOL):
                raise SystemExit(f"PARITY FAILURE vs CPU oracle: dE={dE:.3e} eV dF={dF:.3e} eV/A")
            t0 = time.perf_counter()
            for _ in range(args.cpu_evals if time_cpu else 0):
                csf.run(m, prep, True, cores, cm)
            tc = (time.perf_counter() - t0) / max(1, args.cpu_evals)
            serial = dense = None
            if time_cpu:
                # the same port on ONE thread (the reference's `serial_mode=True` analogue)
                t1 = time.perf_counter()
                for _ in range(2):
                    csf.run(m, prep, True, 1, cm)
                ts = (time.perf_counter() - t1) / 2
                serial = {"value": n0 / ts, "cores": 1, "sample": "2 evaluations of the same frame",
                          "ms_per_eval": ts * 1e3}
                dense = dense_algorithm_row(nn)
            cpu = None if not time_cpu else {
                "value": n0 / tc, "unit": "atom-steps/s", "cores": cores, "kind": "port",
                "sample": f"{args.cpu_evals} evaluations of frame 0 ({n0} atoms, {len(prep['i'])} "
                          f"pairs) by oracle/c/sf_oracle.c (OpenMP, {cores} threads), same "
                          f"model; neighbour list excluded as for the GPU",
                "ms_per_eval": tc * 1e3, "serial": serial, "dense_algorithm": dense,
                "parity": {"dE_eV": dE, "dF_max_eV_per_A": dF, "dW_max_eV": dW}}
        extra = None
        if world == 1 and fpg == 1 and args.rep == 10 and not args.no_extra_configs:
            extra = extra_config_rows(max(10, min(args.steps, 50)), host_cores())
        out = {
            "metric": "atom-steps/sec (energy+forces), 4000-atom Ni rcut=6.5 A",
            "value": value, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n_atoms // fpg}-atom Ni fcc supercell (a=3.524, "
                                   f"{args.rep}^3 cells, jitter 0.05 A, seed 611+frame), G2+G4 "
                                   f"rcut=acut=6.5, D={D}, MLP {D}-64-64-1 softplus, "
                                   f"energy+forces+virial",
                       "frames_per_gpu": fpg, "atoms_per_gpu": n_atoms, "pairs_per_gpu": P,
                       "triples_per_gpu": T, "parallelism": f"frames sharded over {world} GPU(s)",
                       "backend": backend},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "config5": config5,
            "transfer_inclusive": inclusive,
            "kernel_ms": slots,
            "kernel_ms_instrumented": slots_raw,
            "event_ms_per_step": ev_total_ms / n_ev,
            "preheat": preheat,
            "extra_configs": extra,
            "set_frames": nl_info,
            "source_stamp": stamp,
        }
        if world > 1 and config5 is not None:
            out["note"] = (f"N = {world}: read `config5` first — {config5['value']:.4g} atom-steps/s at "
                           f"{CONFIG5_FRAMES_PER_GPU} frames per GPU (BASELINE configs[4]; the 8-byte all-reduce "
                           f"is <1 % of such a step). `value` keeps the N = 1 workload per GPU (ONE frame per "
                           f"rank, weak scaling), where the ~20 us collective hand-off is ~16 % of the step")
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    sys.stdout.flush()
    if out is not None:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


def extra_config_rows(steps, cores):
    """BASELINE.json configs[2] and configs[3] at full size on this GPU, each with its own parity gate
    against the CPU oracle (checker only), so that these figures are driver-visible too. A few ms of
    GPU time each; the oracle evaluations are the slow part (~1 s each)."""
    from tensoralloy_amd import Engine, _lib
    from tests.helpers import make_eam, make_nn, nimo_supercell, oracle_eam_eval, oracle_model
    want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
    rows = {}

    def sf_ref(nn, atoms):
        from oracle import csf
        m = oracle_model(nn)
        return csf.run(m, csf.prepare(m, atoms.get_chemical_symbols(), atoms.positions,
                                      np.asarray(atoms.get_cell(complete=True)), atoms.pbc), True, cores)

    cases = [
        ("C3_NiMo", "Ni4Mo (mp-11507) 7x7x8 = 3920 atoms, jitter 0.05 A, per-species G2/G4 with "
                    "cross-element channels (D = 20), MLP 20-128-128-1, rc 6.5, E+F+virial",
         lambda: make_nn(["Ni", "Mo"], 6.5, True, [128, 128]), nimo_supercell, sf_ref),
        ("C4_EAM", "4000-atom Ni, EamAlloyNN zjw04, rc 6.5, E+F+virial",
         lambda: make_eam(["Ni"], 6.5), lambda: ni_frame(611), oracle_eam_eval),
        ("C4_ADP", "4000-atom Ni, AdpNN zjw04 + mishinh dipole / quadrupole, rc 6.5, E+F+virial",
         lambda: make_eam(["Ni"], 6.5, adp=True), lambda: ni_frame(611), oracle_eam_eval),
        # the reference's DEFAULT EAM / ADP potentials: every function a 1x1 CNN (alloy.py:110-112,
        # adp.py:120-124); inference through the device-built Hermite tables, checked against the oracle's
        # exact networks
        ("C4_nnEAM", "4000-atom Ni, EamAlloyNN with nn rho / phi / embedding (the reference's default "
                     "potential), rc 6.5, E+F+virial",
         lambda: make_eam(["Ni"], 6.5, potential=None), lambda: ni_frame(611), oracle_eam_eval),
        ("C4_nnADP", "4000-atom Ni, AdpNN with nn rho / phi / embedding / dipole / quadrupole, rc 6.5, "
                     "E+F+virial",
         lambda: make_eam(["Ni"], 6.5, adp=True, potential=None), lambda: ni_frame(611), oracle_eam_eval),
    ]
    for key, what, mk_nn, mk_atoms, ref_fn in cases:
        try:
            nn, atoms = mk_nn(), mk_atoms()
            with Engine(nn) as eng:
                info = eng.set_frames([atoms])
                ms, slots = eng.time_compute(want, 5, steps)
                res = eng.fetch(want)
            ref = ref_fn(nn, atoms)
            n = len(atoms)
            dE = abs(ref["energy"] - float(res["energy"][0]))
            dF = float(np.abs(ref["forces"] - res["forces"][:n]).max())
            dW = float(np.abs(ref["virial"] - res["virial"][0]).max())
            if not (dE < E_TOL and dF < F_TOL):
                raise SystemExit(f"PARITY FAILURE ({key}) vs CPU oracle: dE={dE:.3e} eV dF={dF:.3e} eV/A")
            rows[key] = {"workload": what, "value": n / (ms / steps) * 1e3, "unit": "atom-steps/s",
                         "ms_per_step": ms / steps, "steps": steps, "atoms": n, "pairs": int(info.n_pairs),
                         "kernel_ms": {k: v for k, v in slots.items() if v > 0},
                         "parity": {"dE_eV": dE, "dF_max_eV_per_A": dF, "dW_max_eV": dW}}
        except SystemExit:
            raise
        except Exception as exc:  # noqa: BLE001  (a missing helper must not take the bench line down)
            rows[key] = {"workload": what, "error": f"{type(exc).__name__}: {exc}"}
    return rows


def dense_algorithm_row(nn):
    """The reference's own algorithm — dense padded `[n_terms, n_vap, nnl_max(, ij2k_max)]` tensors
    filled by scatter, elementwise G2 / G4 over them (universal.py:583-694, sf.py:79-182) — restated in
    NumPy (oracle/dense.py) and timed once on a frame small enough for its memory (the 4000-atom frame
    would need 2.8 GB per angular tensor). Descriptors only, one thread."""
    try:
        from oracle.dense import descriptors_from_dense
        a = ni_frame(611, rep=4)  # 256 atoms, 14.1 A box
        clf = nn.transformer
        t0 = time.perf_counter()
        feed = clf.get_np_feed_dict(a)          # universal.py:851-893 (vectorised mirror)
        t_feed = time.perf_counter() - t0
        d = nn.descriptor.as_dict()
        t0 = time.perf_counter()
        uni = clf.get_descriptors(feed)         # the scatters of universal.py:583-694
        descriptors_from_dense(uni, clf.elements, clf.rcut, clf.acut, d["eta"], d["omega"], d["beta"],
                               d["gamma"], d["zeta"])
        t_desc = time.perf_counter() - t0
        return {"value": len(a) / t_desc, "unit": "atom-steps/s (descriptors only, no MLP / forces)",
                "cores": 1,
                "sample": f"1 evaluation, {len(a)}-atom Ni frame (4^3 cells), {len(feed['g4.v2g_map'])} "
                          f"triples, NumPy on the dense padded tensors",
                "ms_per_eval": t_desc * 1e3, "feed_dict_ms": t_feed * 1e3}
    except Exception as exc:  # noqa: BLE001  (a missing helper must not take the bench line down)
        return {"error": str(exc)}


if __name__ == "__main__":
    main()
