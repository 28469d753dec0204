python -m pytest tests -x -q -m gpu > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log
bash scripts/ab.sh sf "1 64" 2 2>/dev/null > gpurun_out/ab2.log; cat gpurun_out/ab2.log
for cfg in eam adp; do for nf in 1 64; do
  echo -n "$cfg $nf new: "; python scripts/run_config.py $cfg $nf 30 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['us_per_frame'],2), round(d['algorithmic_GBps']), {k: round(v*1e3/d['frames'],1) for k,v in d['kernel_ms'].items()})"
  echo -n "$cfg $nf rec: "; TA_EAM_RECORDS=1 python scripts/run_config.py $cfg $nf 30 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['us_per_frame'],2), round(d['algorithmic_GBps']), {k: round(v*1e3/d['frames'],1) for k,v in d['kernel_ms'].items()})"
done; done > gpurun_out/eam_ab.log 2>&1; cat gpurun_out/eam_ab.log
for w in 4 8 16 32; do echo -n "copy wg/cu $w: "; TA_COPY_WG_PER_CU=$w python -c "
from bench import ni_model, ni_frame
from tensoralloy_amd import Engine
with Engine(ni_model()) as e:
    e.set_frames([ni_frame(611, rep=4)])
    print(round(e.measure_hbm_copy(1<<30, 10)), round(e.measure_hbm_copy(1<<30, 10)))
"; done > gpurun_out/copy.log 2>&1; cat gpurun_out/copy.log
