#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes for the score and
# fit kernels (separate rocprofv3 runs: --pmc with --kernel-trace only).  Summaries land in gpurun_out/prof_<tag>/ and
# are turned into the profiles/<tag>_* files by tools/pmc_round_summary.py.
# usage: bash tools/profile_round.sh <round tag, e.g. r02> [workload]
TAG=${1:-r03}
WL=${2:-c3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
python3 bench.py --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.log
echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $O -o stats_$WL --output-format csv -- python3 bench.py --workload $WL --no-cpu-baseline --no-api --no-structured --no-c4 --stream-batches 0 > $O/bench_${WL}_under_rocprof.json 2> $O/rocprof_stats.log
echo "stats rc=$?"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O -o pmc${i}_$WL --output-format csv -- python3 bench.py --workload $WL --steps 3 --no-cpu-baseline --no-fast-fit --stream-batches 0 --no-api --no-structured --no-c4 > /dev/null 2> $O/rocprof_pmc$i.log
  echo "pmc pass $i rc=$?"
done
python3 tools/pmc_round_summary.py $O $TAG $WL
# the raw traces are large (gpurun copies back at most 64 MiB): the summaries above are what is kept
rm -f $O/*_kernel_trace.csv $O/*_counter_collection.csv $O/*_agent_info.csv
ls $O | head -40
