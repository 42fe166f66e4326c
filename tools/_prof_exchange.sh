cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ex; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 tools/exchange_overhead.py --workload c3 --chunk-rows 140000 --steps 10 > /dev/null 2> $O/log.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/s_kernel_trace.csv")))
ks=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows)
idx=[i for i,k in enumerate(ks) if "score_frows" in k[2]]
i=idx[-3]; t0=ks[i][0]
for k in ks[i-2:i+40]:
    print(f"{(k[0]-t0)/1e3:9.1f} us  dur {(k[1]-k[0])/1e3:8.1f} us  {k[2][:90]}")
    if k is not ks[i] and "score_frows" in k[2]: break
PY
rm -f $O/*trace.csv
