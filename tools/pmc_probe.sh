#!/bin/bash
# SQ / TCP / TCC counter passes for the segment score kernels on a structured workload (tools/c3s_probe.py):
#   bash tools/pmc_probe.sh <workload> <tag>      (each pass is its own rocprofv3 run: --pmc with --kernel-trace only)
WL=${1:-c3s}
TAG=${2:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_probe_${TAG}_$WL
mkdir -p $O
cd $R
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O -o p$i --output-format csv -- python3 tools/c3s_probe.py --workload $WL --steps 2 --only-all > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 tools/pmc_summary.py $O/*_counter_collection.csv --match "score_seg" > $O/summary.json
python3 - $O/summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    n=max(x['dispatches'] for x in v.values())
    print(k, 'dispatches', n)
    for c,x in sorted(v.items()): print('   %-34s %.5g per launch' % (c, x['sum']/x['dispatches']))
PY
rm -f $O/*_kernel_trace.csv $O/*_counter_collection.csv $O/*_agent_info.csv
