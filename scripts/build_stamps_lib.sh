#!/bin/bash
# Diagnostic build: every translation unit with -DTA_PHASE_STAMPS (phase boundary stamps in the angular
# kernels, see ta_kernels_v2.hip::TA_STAMP) and -DTA_V2_FEW (benchmark shape only) into
# tensoralloy_amd/libtensoralloy_amd_stamps.so. Run with
#   TA_LIB_AB=tensoralloy_amd/libtensoralloy_amd_stamps.so TA_PHASE_STAMPS_OUT=gpurun_out/stamps.txt \
#     python scripts/run_config.py sf 1 20 && python scripts/phase_stamps.py gpurun_out/stamps.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/tensoralloy_amd/csrc
OUT=$CS/build/stamps
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -I$ROOT/include -I$CS -DTA_PHASE_STAMPS -DTA_V2_FEW"
pids=""
for f in ta_api.hip ta_kernels.hip ta_kernels_v2.hip ta_mlp.hip ta_eam.hip ta_nlist.hip ta_grap.hip ta_train.hip ta_hvp.hip ta_neighbor.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -c $CS/$f -o $OUT/${f%.*}.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OUT/*.o -o $ROOT/tensoralloy_amd/libtensoralloy_amd_stamps.so
echo built $ROOT/tensoralloy_amd/libtensoralloy_amd_stamps.so
