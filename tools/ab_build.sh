#!/bin/bash
# Build the current csrc/ into ab/ab_<name>.so (select it with RTREC_AMD_LIB=<path>). ab/ is scratch: it ships with a gpurun
# snapshot (gpurun_out/ does not), so delete it when the A/B is over (rm -rf ab).
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $AB_FLAGS \
  -o ab/ab_$1.so rtrec_amd/csrc/score.hip rtrec_amd/csrc/fit.hip rtrec_amd/csrc/store_host.hip rtrec_amd/csrc/store_device.hip rtrec_amd/csrc/seg_build.hip rtrec_amd/csrc/score_refine.hip rtrec_amd/csrc/score_cands.hip rtrec_amd/csrc/fit_sgd.hip rtrec_amd/csrc/score_dense_fill.hip rtrec_amd/csrc/score_first_touch.hip rtrec_amd/csrc/ordered_fold.hip
echo ab/ab_$1.so
