#!/bin/bash
# Build tensoralloy_amd/libtensoralloy_amd_base.so: the current objects, except that the named
# translation units come from a git revision (default HEAD). For same-session kernel A/B runs:
#   TA_LIB_AB=tensoralloy_amd/libtensoralloy_amd_base.so python scripts/run_config.py sf 1 50
# Usage: bash scripts/build_ab_lib.sh [rev] [tu ...]      (default: HEAD ta_kernels_v2)
set -e
REV=${1:-HEAD}; shift || true
TUS=${@:-ta_kernels_v2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/tensoralloy_amd/csrc
AB=$CS/build/ab
mkdir -p $AB
python -c "from tensoralloy_amd import _lib; _lib.build()"
OBJS=""
for o in $CS/build/*.o; do
  b=$(basename $o .o); keep=1
  for t in $TUS; do [ "$b" = "$t" ] && keep=0; done
  [ $keep = 1 ] && OBJS="$OBJS $o"
done
for t in $TUS; do
  ext=hip; [ -f $CS/$t.cpp ] && ext=cpp
  git -C $ROOT show $REV:tensoralloy_amd/csrc/$t.$ext > $AB/$t.$ext
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -I$ROOT/include -I$CS -c $AB/$t.$ext -o $AB/$t.o
  OBJS="$OBJS $AB/$t.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS -o $ROOT/tensoralloy_amd/libtensoralloy_amd_base.so
echo built $ROOT/tensoralloy_amd/libtensoralloy_amd_base.so from $REV: $TUS
