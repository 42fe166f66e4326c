#!/bin/bash
# Per-kernel timeline of one scoring step (GPU box): bash tools/step_timeline.sh c3
WL=${1:-c3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_step; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 bench.py --workload $WL --no-cpu-baseline --no-fast-fit --stream-batches 0 --steps 8 > /dev/null 2> $O/log.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/s_kernel_trace.csv")))
ks=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows)
idx=[i for i,k in enumerate(ks) if "score_frows" in k[2]]
i=idx[5]; t0=ks[i][0]
for k in ks[i-3:i+8]:
    print(f"{(k[0]-t0)/1e3:9.1f} us  dur {(k[1]-k[0])/1e3:8.1f} us  {k[2][:70]}")
PY
rm -f $O/*trace.csv
