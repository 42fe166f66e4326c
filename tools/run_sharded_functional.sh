# Same-GPU functional run of the sharded engine path (every rank drives cuda:0, collectives over gloo): the ids CRC of
# every N must equal the single-rank CRC, for the division of record (user rows) and the other one (item columns:
# `alt_sharding`, or --score-shard columns).  bench.py launches its ranks itself (torch.distributed.run child).
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-fast-fit --stream-batches 0"
python bench.py --workload small $F > gpurun_out/bench_small_n1.json 2> gpurun_out/bench_small_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --workload small $F > gpurun_out/bench_small_n2.json 2> gpurun_out/bench_small_n2.log
python bench.py --workload c2 $F > gpurun_out/bench_c2_n1.json 2> gpurun_out/bench_c2_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 4 --workload c2 $F > gpurun_out/bench_c2_n4.json 2> gpurun_out/bench_c2_n4.log
python bench.py --workload c3 $F > gpurun_out/bench_c3_n1.json 2> gpurun_out/bench_c3_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --workload c3 $F > gpurun_out/bench_c3_n2.json 2> gpurun_out/bench_c3_n2.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --workload c3 $F --score-shard columns > gpurun_out/bench_c3_n2_cols.json 2> gpurun_out/bench_c3_n2_cols.log
tail -2 gpurun_out/bench_small_n2.log gpurun_out/bench_c2_n4.log gpurun_out/bench_c3_n2.log gpurun_out/bench_c3_n2_cols.log
cat gpurun_out/bench_small_n1.json gpurun_out/bench_small_n2.json gpurun_out/bench_c2_n1.json gpurun_out/bench_c2_n4.json gpurun_out/bench_c3_n1.json gpurun_out/bench_c3_n2.json gpurun_out/bench_c3_n2_cols.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); a=d.get('alt_sharding') or {}
        print(d['n_gpus'], d['config']['workload'][:6], d['config']['parallelism'], round(d['value']), round(d['ms_per_step'],3), d['topk_ids_crc32'], round(d['fit']['seconds'],3), d['ranks_seen'], d['backend'], '| alt', a.get('score_shard'), a.get('ms_per_step') and round(a['ms_per_step'],3), a.get('same_topk_ids'))
"
