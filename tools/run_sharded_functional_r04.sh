cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-fast-fit --stream-batches 0 --no-api --no-structured --steps 5"
python bench.py --workload small $F > gpurun_out/f_small_n1.json 2> gpurun_out/f_small_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 --workload small $F > gpurun_out/f_small_n2.json 2> gpurun_out/f_small_n2.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 --workload small $F --shard-w > gpurun_out/f_small_n2_sw.json 2> gpurun_out/f_small_n2_sw.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 4 --workload small $F --score-shard columns > gpurun_out/f_small_n4_cols.json 2> gpurun_out/f_small_n4_cols.log
python bench.py --workload smalls $F > gpurun_out/f_smalls_n1.json 2> gpurun_out/f_smalls_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 --workload smalls $F > gpurun_out/f_smalls_n2.json 2> gpurun_out/f_smalls_n2.log
RTREC_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 --workload smalls $F --shard-w > gpurun_out/f_smalls_n2_sw.json 2> gpurun_out/f_smalls_n2_sw.log
RTREC_BENCH_SAME_GPU=1 timeout 400 python bench.py --gpus 2 --workload c3 $F > gpurun_out/f_c3_n2.json 2> gpurun_out/f_c3_n2.log
tail -3 gpurun_out/f_small_n2_sw.log gpurun_out/f_smalls_n2_sw.log gpurun_out/f_c3_n2.log
cat gpurun_out/f_small_n1.json gpurun_out/f_small_n2.json gpurun_out/f_small_n2_sw.json gpurun_out/f_small_n4_cols.json gpurun_out/f_smalls_n1.json gpurun_out/f_smalls_n2.json gpurun_out/f_smalls_n2_sw.json gpurun_out/f_c3_n2.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); a=d.get('alt_sharding') or {}; c=d.get('score_shard_choice') or {}
        print(d['n_gpus'], d['config']['workload'][:6], d['config']['parallelism'], round(d['value']), round(d['ms_per_step'],3), d['topk_ids_crc32'], round(d['fit']['seconds'],3), d['ranks_seen'], d['backend'], c.get('mode'), c.get('chosen_by'), c.get('w_column_sharded'), '| alt', a.get('score_shard'), a.get('same_topk_ids'))
"
