#!/bin/bash
# Per-phase clocks of score_sparse_kernel (diagnostic build -DSCORE_PROFILE, tools/ab_build.sh prof):
#   AB_FLAGS=-DSCORE_PROFILE bash tools/ab_build.sh prof   (here)
#   bash tools/score_phase_profile.sh "c3 c2"              (GPU box)
cd $GRAFT_REPO_ROOT
for W in ${1:-c3 c2}; do
  RTREC_AMD_LIB=$GRAFT_REPO_ROOT/ab/ab_prof.so python3 bench.py --workload $W --no-cpu-baseline --no-fast-fit --stream-batches 0 --steps 3 2>&1 >/dev/null | grep "score profile" | sed "s/^/$W /"
done
