#!/bin/bash
# a variant of libvadc_amd.so with extra defines for kernels_encoder_mfma.hip (e.g. -DVADC_PHASE_PROF), built BESIDE the product (tools/abl_build.sh);
# usage: export VADC_AMD_LIB=$(tools/build_prof.sh [defines...] | tail -1)
exec bash "$(dirname "$0")/abl_build.sh" kernels_encoder_mfma.hip "$@"
