#!/bin/bash
# abl_build.sh <source.hip> [compiler flags / defines ...]
# Builds vadc_amd/csrc/build/abl/libvadc_amd_abl.so = the product's objects with <source>'s object recompiled with the given flags (profiling stamps, timing-only
# ablations whose results are WRONG), and prints its path on the last line.  The product library vadc_amd/libvadc_amd.so and its objects are never touched: a script that
# dies half way leaves nothing behind that a later bench or parity run could pick up.  Python loads the variant through VADC_AMD_LIB (vadc_amd/_lib.py):
#     export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_lstm.hip -DVADC_LSTM_ABL_NOMFMA | tail -1)
set -e
cd "$(dirname "$0")/../vadc_amd/csrc"
make -j8 > /dev/null
src=$1; shift
obj=${src%.hip}.o
mkdir -p build/abl
extra=""
[ "$src" = kernels_frontend.hip ] && extra="-ffp-contract=off -fno-slp-vectorize"
[ "$src" = kernels_encoder_fused.hip ] && extra="-fno-slp-vectorize"          # as the Makefile (tools/check_pk_opsel.py)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c "$src" -o "build/abl/$obj"
objs=""
for o in build/*.o; do
   if [ "$(basename "$o")" = "$obj" ]; then objs="$objs build/abl/$obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/abl/libvadc_amd_abl.so $objs
echo "$(pwd)/build/abl/libvadc_amd_abl.so"
