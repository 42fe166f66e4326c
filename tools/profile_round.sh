#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
# usage: bash tools/profile_round.sh <round tag, e.g. r01b> [workload]
TAG=${1:-r01}
WL=${2:-c3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
python3 bench.py --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.log
rocprofv3 --kernel-trace --stats -d $O -o stats_$WL --output-format csv -- python3 bench.py --workload $WL --no-cpu-baseline > $O/bench_${WL}_under_rocprof.json 2> $O/rocprof_stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch_$WL --output-format csv -- python3 bench.py --workload $WL --steps 2 --no-cpu-baseline > /dev/null 2> $O/rocprof_fetch.log
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O -o write_$WL --output-format csv -- python3 bench.py --workload $WL --steps 2 --no-cpu-baseline > /dev/null 2> $O/rocprof_write.log
python3 tools/pmc_summary.py $O/fetch_${WL}_counter_collection.csv $O/write_${WL}_counter_collection.csv > $O/pmc_$WL.json
ls $O
cat $O/bench_$WL.json
