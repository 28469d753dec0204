"""One-rank probe of the per-step cost of the RCCL hand-off (no collective / async all_reduce /
blocking all_reduce): host enqueue time and wall time per step. Run on a GPU box:
    HSA_ENABLE_IPC_MODE_LEGACY=0 python scripts/dist_probe.py
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
from bench import ni_frame, ni_model
from tensoralloy_amd import Engine, _lib
eng = Engine(ni_model(), device=0)
eng.set_frames([ni_frame(611)])
want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
RING = 64
ebuf = torch.zeros(RING, dtype=torch.float64, device="cuda:0")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
slots=[ebuf[k:k+1] for k in range(RING)]; ptrs=[s.data_ptr() for s in slots]
def run(K, mode):
    if mode == "ring":
        infl=[None]*RING
        torch.cuda.synchronize()
        t0=time.perf_counter(); tc=0.0; ta=0.0
        for k in range(K):
            s_ = k % RING
            if infl[s_] is not None: infl[s_].wait()
            t1=time.perf_counter()
            eng.set_batch_energy_target(ptrs[s_]); eng.compute(want)
            t2=time.perf_counter()
            infl[s_]=dist.all_reduce(slots[s_], async_op=True)
            t3=time.perf_counter(); tc+=t2-t1; ta+=t3-t2
        t_enq=time.perf_counter()-t0
        for w in infl:
            if w is not None: w.wait()
        torch.cuda.synchronize(); eng.synchronize()
        t_all=time.perf_counter()-t0
        print(mode, "per step: total %.1f us, host enqueue %.1f us (compute %.1f, allreduce %.1f)"%(t_all/K*1e6, t_enq/K*1e6, tc/K*1e6, ta/K*1e6))
        return
    infl=[None,None]
    torch.cuda.synchronize()
    t0=time.perf_counter(); tc=0.0; ta=0.0
    for k in range(K):
        if mode!="none" and infl[k%2] is not None: infl[k%2].wait()
        t1=time.perf_counter()
        if mode!="none": eng.set_batch_energy_target(ptrs[k%2])
        eng.compute(want)
        t2=time.perf_counter()
        if mode=="async": infl[k%2]=dist.all_reduce(slots[k%2], async_op=True)
        elif mode=="sync": dist.all_reduce(slots[k%2])
        t3=time.perf_counter()
        tc+=t2-t1; ta+=t3-t2
    t_enq=time.perf_counter()-t0
    torch.cuda.synchronize(); eng.synchronize()
    t_all=time.perf_counter()-t0
    print(mode, "per step: total %.1f us, host enqueue %.1f us (compute %.1f, allreduce %.1f)"%(t_all/K*1e6, t_enq/K*1e6, tc/K*1e6, ta/K*1e6))
for mode in ("none","async","ring","sync","ring","none"):
    run(20, mode); run(200, mode)
dist.destroy_process_group()
