#!/bin/bash
# Build the current csrc/ into rtrec_amd/lib/ab_<name>.so (select it with RTREC_AMD_LIB=<path>).
set -e
cd "$(dirname "$0")/.."
mkdir -p rtrec_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $AB_FLAGS \
  -o rtrec_amd/lib/ab_$1.so rtrec_amd/csrc/score.hip rtrec_amd/csrc/fit.hip rtrec_amd/csrc/store_host.hip rtrec_amd/csrc/store_device.hip rtrec_amd/csrc/seg_build.hip rtrec_amd/csrc/score_refine.hip rtrec_amd/csrc/score_cands.hip
echo rtrec_amd/lib/ab_$1.so
