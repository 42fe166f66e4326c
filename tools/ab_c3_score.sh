# usage (GPU box): bash tools/ab_c3_score.sh <variant> <variant> ...   -- alternates ab/ab_<variant>.so twice on the C3 score pass
cd $GRAFT_REPO_ROOT
F="--workload c3 --no-cpu-baseline --no-api --no-structured --no-c4 --stream-batches 0 --no-fast-fit --steps 20 --warmup 3"
for rep in 1 2; do
for V in "$@"; do
  RTREC_AMD_LIB=$GRAFT_REPO_ROOT/ab/ab_$V.so python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'ms_per_step', round(d['ms_per_step'],4), 'crc', d['topk_ids_crc32'])"
done; done
