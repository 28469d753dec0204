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
energy + forces + virial. Inputs (positions, cells, neighbour list) are
resident in HBM before the timed region; the neighbour-list build is reported
separately. For N > 1 every rank owns `--frames-per-gpu` independent frames
(weak scaling) and the only collective is one RCCL all-reduce of the batch
energy per step.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md §Chip-level parameters
# dense fp64 matrix peak: AMD's MI355X figure, 78.6 TFLOP/s = 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz
# (v_mfma_f64_16x16x4_f64 = 2048 flop in 64 cycles; the guide's table has no fp64 row)
FP64_MFMA_PEAK_TFLOPS = 78.6
E_TOL, F_TOL = 1e-6, 1e-5  # north_star parity tolerances (eV, eV/A)


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
    pts = np.array([base + np.array([x, y, z]) * a
                    for x in range(rep) for y in range(rep) for z in range(rep)]).reshape(-1, 3)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames-per-gpu", type=int, default=1)
    ap.add_argument("--rep", type=int, default=10, help="fcc cells per edge (10 -> 4000 atoms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-evals", type=int, default=40)  # ~10 s of 16-thread CPU work
    args = ap.parse_args()

    # the driver reads ONE JSON line from stdout: native libraries print there too (RCCL writes a
    # version banner on initialisation, gloo its connection report), so everything else that lands
    # on fd 1 during the run is sent to stderr and the line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from tensoralloy_amd.parallel import world_from_env
    rank, local_rank, world = world_from_env()
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = torch = None
    # TA_BENCH_FORCE_DIST=1 rehearses the multi-rank code path with a single rank
    use_dist = world > 1 or os.environ.get("TA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        # torch first: its bundled HIP runtime must be the one the process shares
        import torch
        import torch.distributed as dist
        # TA_BENCH_BACKEND=gloo rehearses several ranks on a box with fewer GPUs than ranks (ranks
        # share devices; the all-reduce then goes through gloo): the driver's runs use nccl = RCCL
        backend = os.environ.get("TA_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

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

    ebuf = None
    inflight = [None, None]
    if use_dist:
        ebuf = torch.zeros(2, dtype=torch.float64, device=f"cuda:{local_rank}")
        # kernels go onto torch's current stream so the collective is ordered
        # behind them by stream semantics: no host synchronisation per step
        # torch's current (default) stream: a dedicated non-blocking stream measured the same with
        # nccl (173.9 against 174.8 us/step, one rank) and makes gloo's CUDA path block for
        # milliseconds per step (TA_BENCH_SIDE_STREAM=1 selects it for experiments)
        if os.environ.get("TA_BENCH_SIDE_STREAM") == "1":
            side = torch.cuda.Stream(device=local_rank)
            torch.cuda.set_stream(side)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        slots_ = [ebuf[0:1], ebuf[1:2]]
        slot_ptrs = [s_.data_ptr() for s_ in slots_]

    def step(k):
        if use_dist and inflight[k % 2] is not None:
            inflight[k % 2].wait()  # stream-level wait before the slot is overwritten
        if use_dist:
            # the frame-reduce kernel writes the batch energy straight into this step's slot of
            # the torch buffer, then ONE RCCL all-reduce of 8 bytes reduces it in place
            slot = slots_[k % 2]
            eng.set_batch_energy_target(slot_ptrs[k % 2])
            eng.compute(want)
            inflight[k % 2] = dist.all_reduce(slot, op=dist.ReduceOp.SUM, async_op=True)
            return inflight[k % 2]
        eng.compute(want)
        return None

    def sync_all():
        eng.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    pending = [step(k) for k in range(args.warmup)]
    for w in pending:
        if w is not None:
            w.wait()
    sync_all()
    t0 = time.perf_counter()
    pending = [step(k) for k in range(args.steps)]
    for w in pending:
        if w is not None:
            w.wait()
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(n_atoms)], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_atoms = float(tot.item())
    else:
        total_atoms = float(n_atoms)
    value = total_atoms * args.steps / elapsed

    out = None
    if rank == 0:
        # ---- per-kernel durations with HIP events on the engine's stream ----
        ev_total_ms, slots = eng.time_compute(want, 2, max(5, min(args.steps, 20)))
        # dominant kernel. SURVEY §8(d) prices one pass over the packed records at 32 B / pair +
        # 60 B / triple; the fused kernel makes both passes (descriptors, then dE/dD) in one
        # launch, the separate backward kernel one.
        if slots.get("fused", 0.0) > 0:
            dom_name, dom_key, passes = "sf_fused_kernel<1,2,2,16,true>", "sf_fused_kernel", 2.0
            bwd_ms = slots["fused"]
        else:
            dom_name, dom_key, passes = "backward_v2_kernel<1,2,2,12,true>", "backward_v2_kernel", 1.0
            bwd_ms = slots["backward"]
        bwd_bytes = passes * (32.0 * P + 60.0 * T)
        achieved = bwd_bytes / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0
        eval_bytes = 2.0 * (32.0 * P + 60.0 * T) + n_atoms * 8.0 * (3 * D + 4) + 72.0 * fpg
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes of this
        # same command (scripts/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate passes,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane reads on gfx950)
        traffic = valu = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and fpg == 1 and args.rep == 10:
            try:
                with open(tpath) as fp:
                    for name, rec in json.load(fp).items():
                        if name.startswith(dom_key):
                            traffic = rec["hbm_bytes_per_launch"]
                            valu = rec.get("valu_wave_insts_per_launch")
            except Exception:
                traffic = valu = None
        # what actually bounds this kernel: FP64 VALU issue. One wavefront instruction occupies a
        # SIMD for 4 cycles (16 lanes/cycle); 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock.
        valu_obj = None
        if valu and bwd_ms > 0:
            peak_issue = 256 * 4 * 2.4e9 / 4.0            # wavefront instructions per second
            valu_obj = {"wave_insts_per_launch": valu, "achieved_Ginst_per_s": valu / (bwd_ms * 1e-3) / 1e9,
                        "peak_Ginst_per_s": peak_issue / 1e9,
                        "frac": valu / (bwd_ms * 1e-3) / peak_issue,
                        "source": "SQ_INSTS_VALU, profiles/pmc_traffic.json"}
        copy_gbs = eng.measure_hbm_copy(1 << 30, 10)  # achievable copy rate on this box, SURVEY 8(d)
        # MFMA utilisation of the batched per-atom MLP (north_star): v_mfma_f64_16x16x4_f64 count of
        # one launch (forward + backward-to-inputs, padded tiles; equals SQ_INSTS_VALU_MFMA_F64 in
        # profiles/r01_rocprofv3_summary.txt) x 2048 flop, over the kernel's HIP-event duration,
        # against the dense fp64 matrix peak of MI355X_MICROARCH.md
        sizes = [D] + list(nn.hidden_sizes[nn.elements[0]]) + [1]
        pad = lambda v: (v + 15) // 16 * 16
        per_tile = 2 * sum((pad(sizes[l]) // 4) * (pad(sizes[l + 1]) // 16) for l in range(len(sizes) - 1))
        n_mfma = per_tile * ((n_atoms // fpg + 15) // 16) * fpg
        mlp_ms = slots.get("mlp", 0.0)
        mlp_mfma = None
        if mlp_ms > 0:
            tf = n_mfma * 2048.0 / (mlp_ms * 1e-3) / 1e12
            mlp_mfma = {"kernel": "mlp_kernel<256>", "mfma_insts_per_launch": n_mfma,
                        "flop_per_launch": n_mfma * 2048.0, "kernel_ms": mlp_ms, "achieved": tf,
                        "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                        "note": "16-row tiles: one frame is 250 workgroups on 256 CUs, the kernel is "
                                "bound by the latency of its 6 dependent GEMM phases, not by the matrix pipe"}
        roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "peak_measured_copy": copy_gbs,
                    "frac_of_measured_copy": achieved / copy_gbs if copy_gbs > 0 else None,
                    "traffic": traffic,
                    "algorithmic_bytes_per_launch": bwd_bytes,
                    "kernel_ms": bwd_ms,
                    "whole_eval_bytes": eval_bytes,
                    "whole_eval_frac": eval_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                    "valu_issue": valu_obj,
                    "mlp_mfma": mlp_mfma,
                    "note": "triples are generated on the fly from LDS-staged pair records, so "
                            "HBM traffic is far below the packed-record bytes the formula prices "
                            "(frac > 1); the kernel is bounded by FP64 VALU issue, see valu_issue"}

        # ---- parity gate + CPU baseline (oracle = checker, timed on host cores) ----
        res = eng.fetch(want)
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
            # the same port on ONE thread (the reference's `serial_mode=True` analogue)
            serial = None
            if time_cpu:
                t1 = time.perf_counter()
                for _ in range(2):
                    csf.run(m, prep, True, 1, cm)
                ts = (time.perf_counter() - t1) / 2
                serial = {"value": n0 / ts, "cores": 1, "sample": "2 evaluations of the same frame", "ms_per_eval": ts * 1e3}
            cpu = None if not time_cpu else {"value": n0 / tc, "unit": "atom-steps/s", "cores": cores, "kind": "port",
                   "sample": f"{args.cpu_evals} evaluations of frame 0 ({n0} atoms, {len(prep['i'])} "
                             f"pairs) by oracle/c/sf_oracle.c (OpenMP, {cores} threads), same "
                             f"model; neighbour list excluded as for the GPU",
                   "ms_per_eval": tc * 1e3, "serial": serial,
                   "parity": {"dE_eV": dE, "dF_max_eV_per_A": dF, "dW_max_eV": dW}}
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
                       "triples_per_gpu": T, "parallelism": f"frames sharded over {world} GPU(s)"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernel_ms": slots,
            "event_ms_per_step": ev_total_ms / max(5, min(args.steps, 20)),
            "set_frames": {"first_call_s": t_nl, "steady_call_s": t_set,
                           "neighbor_list_on_device": bool(info.nl_on_device),
                           "neighbor_list_ms": info.nl_ms, "c_abi_ms": info.set_frames_ms},
        }
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    sys.stdout.flush()
    if out is not None:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
