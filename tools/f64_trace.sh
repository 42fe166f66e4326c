#!/bin/bash
# kernel times of the float64 / DENSE passes: bash tools/f64_trace.sh c3s
WL=${1:-c3s}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_f64; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 tools/dense_bench.py --workload $WL > $O/out.txt 2> $O/log.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/s_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms  calls {r["Calls"]:>6}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:100]}')
PY
rm -f $O/*trace.csv
