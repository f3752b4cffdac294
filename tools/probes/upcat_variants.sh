#!/bin/bash
# Build alternative libraries whose upcat object is the round-3 reproducer (tools/probes/upcat_r03.hip) under several flag sets,
# then (on a GPU box) run tools/probes/upcat_check.py against each through FZ_LIB_PATH.
#   bash tools/probes/upcat_variants.sh build      (here: cross-compiles)
#   bash tools/probes/upcat_variants.sh run        (GPU box)
set -u
cd "$(dirname "$0")/../.."
OBJ=factorizer_amd/csrc/build
OUT=tools/probes/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-comment -ffp-contract=fast -Ifactorizer_amd/csrc"
declare -A V
V[w4]="-DUPCAT_WAVES=4"
V[w8]="-DUPCAT_WAVES=8"
V[w8nop]="-DUPCAT_WAVES=8 -DFZ_UPCAT_NOP"
V[w8pad]="-DUPCAT_WAVES=8 -mllvm -amdgpu-mfma-padding-ratio=100"
V[w8O1]="-DUPCAT_WAVES=8 -O1"
V[w8noslp]="-DUPCAT_WAVES=8 -fno-slp-vectorize"
if [ "$1" = build ]; then
  python -m factorizer_amd.build > /dev/null
  mkdir -p $OUT
  for k in "${!V[@]}"; do
    hipcc $FLAGS ${V[$k]} -c tools/probes/upcat_r03.hip -o $OUT/upcat_r03_$k.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v "/upcat.o")
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libfz_upcat_r03_$k.so $objs $OUT/upcat_r03_$k.o || exit 1
    echo built $k
  done
else
  for k in w4 w8 w8nop w8pad w8O1 w8noslp; do
    for rep in 1 2; do
      echo "== $k run $rep"; FZ_LIB_PATH=$OUT/libfz_upcat_r03_$k.so python tools/probes/upcat_check.py 2>&1 | grep -v amdgpu.ids
    done
  done
fi
