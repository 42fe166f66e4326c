#!/bin/bash
# FETCH / WRITE / L2-hit passes only, then the traffic summary: bash tools/pmc_traffic_only.sh r03 c3s
TAG=${1:-r03}; WL=${2:-c3s}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; rm -f $O/pmc*_${WL}_counter_collection.csv; cd $R
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O -o pmc${i}_$WL --output-format csv -- python3 bench.py --workload $WL --steps 3 --no-cpu-baseline --no-fast-fit --stream-batches 0 --no-api --no-structured > /dev/null 2> $O/rocprof_pmc$i.log
  echo "pmc pass $i rc=$?"
done
head -1 $O/pmc1_${WL}_counter_collection.csv
python3 tools/pmc_round_summary.py $O $TAG $WL
cat $O/${TAG}_${WL}_pmc_traffic.json
rm -f $O/*_kernel_trace.csv $O/*_counter_collection.csv $O/*_agent_info.csv
