#!/bin/bash
# Bulk fit: how many of the longest targets go to the multi-wave kernel, on how many work-queue slots.
# usage (GPU box): tools/heavy_sweep.sh <workload> "<heavy:slots> ..."
cd $GRAFT_REPO_ROOT
W=$1; shift
for rep in 1 2; do for HS in $@; do
H=${HS%%:*}; S=${HS##*:}
RTREC_AMD_FIT_HEAVY=$H RTREC_AMD_FIT_HEAVY_SLOTS=$S python bench.py --workload $W --no-cpu-baseline --no-api --no-structured --no-c4 --stream-batches 0 --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('heavy $H slots $S  fit', round(d['fit']['seconds'],3), 's  crc', d['topk_ids_crc32'])"
done; done
