#!/bin/bash
# debug: 2-rank bench over gloo on one GPU
cd "$(dirname "$0")/.."
TA_BENCH_BACKEND=gloo TA_BENCH_DEBUG=1 python bench.py --gpus 2 --steps 3 --warmup 1 --rep 4 --no-cpu-baseline > gpurun_out/dbg_c5.json 2> gpurun_out/dbg_c5.err
echo "rc=$?"
