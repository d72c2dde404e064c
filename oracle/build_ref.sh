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

# The reference's segmenter functions (vadc.c:165-299: feed_probability, emit_speech_segment, combine_or_emit_speech_segment) between the two halves of
# ref_segmenter_harness.c, straight from the reference tree into the compiler: no excerpt is written anywhere.  vadc.c as a whole needs <windows.h>.
H="$HERE/ref_segmenter_harness.c"
{ sed -n '/---8<--- PROLOGUE/,/---8<--- DRIVER/p' "$H"; sed -n '165,299p' "$REF/vadc.c"; sed -n '/---8<--- DRIVER/,$p' "$H"; } | \
   gcc -O1 -std=gnu11 -w -include stddef.h -DONNX_INFERENCE_ENABLED=0 -I"$REF" -x c - -o "$HERE/_ref/ref_segmenter"
echo "built $HERE/_ref/ref_segmenter"
