cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fit_${1:-c3}
mkdir -p $O
cd $R
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O -o p$i --output-format csv -- python3 tools/fit_sweep.py --workload ${1:-c3} --configs sw:4096 > $O/p$i.log 2>&1
done
python3 tools/pmc_summary.py $O/*_counter_collection.csv --match fit_columns > $O/summary.json
cat $O/summary.json
