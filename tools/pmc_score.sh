#!/bin/bash
# SQ counter passes for the score kernel: bash tools/pmc_score.sh <workload>
WL=${1:-c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_score_$WL
mkdir -p $O
cd $R
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d $O -o p$i --output-format csv -- python3 bench.py --workload $WL --steps 3 --no-cpu-baseline > $O/p$i.log 2>&1
done
python3 tools/pmc_summary.py $O/*_counter_collection.csv --match "score_sparse_kernel<float, false>" > $O/summary.json
cat $O/summary.json | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    n=max(x['dispatches'] for x in v.values())
    print(k, 'dispatches', n)
    for c,x in sorted(v.items()): print('   %-26s %.4g per launch' % (c, x['sum']/x['dispatches']))
"
