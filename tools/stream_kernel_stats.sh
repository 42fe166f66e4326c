#!/bin/bash
# rocprofv3 kernel statistics of a few exact-mode mini-batches (which kernels a partial_fit spends its time in).
# usage (GPU box): bash tools/stream_kernel_stats.sh [workload] [batches]
WL=${1:-c3}; NB=${2:-3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stream_stats; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O -o sp --output-format csv -- python3 tools/stream_profile.py --workload $WL --batches $NB > $O/out.txt 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/sp_kernel_stats.csv")))
for r in rows[:25]:
    print(r["Name"][:90], r["Calls"], "total_ms", round(int(r["TotalDurationNs"]) / 1e6, 2), "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
rm -f $O/*_kernel_trace.csv $O/*_agent_info.csv
