#!/bin/sh
# Compile the reference's hot-path sources IN PLACE from $1 (default /root/reference) into
# oracle/_ref/libvadc_ref.so.  Flags mirror build_msvc.bat:68 (/O2 /arch:AVX2 /DNDEBUG) with MSVC's
# default /fp:precise semantics reproduced by -ffp-contract=off.  No reference source is copied.
set -e
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ ! -f "$REF/silero_v3.c" ]; then
   echo "build_ref.sh: $REF not present -- skipping (GPU box uses the prebuilt file if any)" >&2
   exit 0
fi
mkdir -p "$HERE/_ref"
gcc -O2 -mavx2 -mfma -ffp-contract=off -DNDEBUG -std=gnu11 -w -fPIC -shared \
    -I"$REF" -I"$REF/tracy" \
    -o "$HERE/_ref/libvadc_ref.so" "$HERE/ref_harness.c" -lm
echo "built $HERE/_ref/libvadc_ref.so"
