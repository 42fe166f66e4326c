# Same-GPU functional run of the sharded engine path (every rank drives cuda:0, collectives over gloo): the ids CRC of
# every N must equal the single-rank CRC.  bench.py launches its ranks itself (torch.distributed.run child).
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-fast-fit --stream-batches 0"
python bench.py --workload small $F > gpurun_out/bench_small_n1.json 2> gpurun_out/bench_small_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --workload small $F > gpurun_out/bench_small_n2.json 2> gpurun_out/bench_small_n2.log
python bench.py --workload c2 $F > gpurun_out/bench_c2_n1.json 2> gpurun_out/bench_c2_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 4 --workload c2 $F > gpurun_out/bench_c2_n4.json 2> gpurun_out/bench_c2_n4.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 4 --workload c2 $F --score-shard rows > gpurun_out/bench_c2_n4_rows.json 2> gpurun_out/bench_c2_n4_rows.log
python bench.py --workload c3 $F > gpurun_out/bench_c3_n1.json 2> gpurun_out/bench_c3_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --workload c3 $F > gpurun_out/bench_c3_n2.json 2> gpurun_out/bench_c3_n2.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --workload c3 $F --score-shard rows > gpurun_out/bench_c3_n2_rows.json 2> gpurun_out/bench_c3_n2_rows.log
tail -2 gpurun_out/bench_small_n2.log gpurun_out/bench_c2_n4.log gpurun_out/bench_c3_n2.log
cat gpurun_out/bench_small_n1.json gpurun_out/bench_small_n2.json gpurun_out/bench_c2_n1.json gpurun_out/bench_c2_n4.json gpurun_out/bench_c2_n4_rows.json gpurun_out/bench_c3_n1.json gpurun_out/bench_c3_n2.json gpurun_out/bench_c3_n2_rows.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['n_gpus'], d['config']['workload'][:6], d['config']['parallelism'], d['value'], d['ms_per_step'], d['topk_ids_crc32'], d['fit']['seconds'], d['ranks_seen'], d['backend'])
"
