#!/bin/bash
# usage: tools/ab_run.sh "<workloads>" "<extra bench args>" <variant> <variant> ...   (GPU box; alternates variants twice)
cd $GRAFT_REPO_ROOT
WLS=$1; shift
ARGS=$1; shift
for rep in 1 2; do for V in "$@"; do for W in $WLS; do
RTREC_AMD_LIB=$GRAFT_REPO_ROOT/ab/ab_$V.so python bench.py --workload $W --no-cpu-baseline --steps 5 $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V $W', round(d['roofline']['kernel_ms_avg'],3), round(d['ms_per_step'],3), d['topk_ids_crc32'], round(d['fit']['seconds'],3))"
done; done; done
