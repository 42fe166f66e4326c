#!/bin/bash
# GPU-side timeline of single-user recommend calls (kernels + copies per request): bash tools/serve_trace.sh c3
WL=${1:-c3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_serve; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --memory-copy-trace -d $O -o s --output-format csv -- python3 tools/serve_latency.py --workload $WL --requests 400 --single-only > $O/out.txt 2> $O/log.txt
tail -1 $O/out.txt
python3 - <<PY
import csv
ev=[]
for r in csv.DictReader(open("$O/s_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("(anonymous namespace)::","")[:80]))
for r in csv.DictReader(open("$O/s_memory_copy_trace.csv")):
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Size","")))
ev.sort()
tail=ev[-120:-60]
t0=tail[0][0]
for a,b,n in tail:
    print(f"{(a-t0)/1e3:9.1f} us  dur {(b-a)/1e3:7.1f} us  {n}")
PY
rm -f $O/*trace.csv
