#!/bin/bash
# usage: tools/ab_stream.sh <workload> <variant> <variant> ...   (GPU box; alternates the variants twice)
# Each line: variant, then per mini-batch (first one dropped) total / to_csc / kernel ms and the heaviest target's phases.
cd $GRAFT_REPO_ROOT
W=$1; shift
for rep in 1 2; do for V in "$@"; do
RTREC_AMD_LIB=$GRAFT_REPO_ROOT/ab/ab_$V.so python tools/stream_profile.py --workload $W --batches 3 2>/dev/null | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')][1:]
for d in rows:
    t = d['top'][0]
    print('$V', 'total', round(d['total_ms']), 'kernel', round(d['fit_kernel_ms']), 'sum_target_s', round(d['sum_target_s'], 1), 'top: dur', round(t['dur_ms']), 'fold', round(t.get('fold_ms', 0)), 'upd', round(t.get('upd_ms', 0)), 'gap', round(t.get('gap_ms', 0)))
"
done; done
