# Functional run of the sharded bench paths on ONE GPU (RTREC_BENCH_SAME_GPU=1: every rank drives cuda:0, collectives over gloo):
# the ids CRC of every N must equal the single-rank one, for the main workload and for the c4 leg.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_sharded; mkdir -p $O
F="--no-cpu-baseline --no-fast-fit --stream-batches 0 --no-api --no-structured --steps 3"
python bench.py --workload small $F --no-c4 > $O/small_n1.json 2> $O/small_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 --workload small $F --no-c4 > $O/small_n2.json 2> $O/small_n2.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 4 --workload small $F --no-c4 --exchange-scores > $O/small_n4_scores.json 2> $O/small_n4_scores.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 4 --workload small $F --no-c4 --score-shard columns > $O/small_n4_cols.json 2> $O/small_n4_cols.log
python bench.py --workload c3 $F > $O/c3_n1.json 2> $O/c3_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 900 python bench.py --gpus 2 --workload c3 $F > $O/c3_n2.json 2> $O/c3_n2.log
RTREC_BENCH_SAME_GPU=1 timeout 900 python bench.py --gpus 4 --workload c3 $F > $O/c3_n4.json 2> $O/c3_n4.log
tail -2 $O/c3_n4.log
cat $O/small_n1.json $O/small_n2.json $O/small_n4_scores.json $O/small_n4_cols.json $O/c3_n1.json $O/c3_n2.json $O/c3_n4.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); a=d.get('alt_sharding') or {}; c=d.get('score_shard_choice') or {}; c4=d.get('c4') or {}
        print(d['n_gpus'], d['config']['workload'][:6], d['config']['parallelism'], '|', d['config'].get('exchange'), '|', round(d['value']), round(d['ms_per_step'],3), 'crc', d['topk_ids_crc32'], 'fit', round(d['fit']['seconds'],3), d['backend'], c.get('mode'), '| alt', a.get('score_shard'), a.get('same_topk_ids'), '| c4', c4.get('topk_ids_crc32'), c4.get('ms_per_step'), c4.get('fit_seconds'), c4.get('score_shard'), c4.get('error'))
" | tee $O/summary.txt
