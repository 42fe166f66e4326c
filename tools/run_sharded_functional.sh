cd $GRAFT_REPO_ROOT
python bench.py --workload small --no-cpu-baseline > gpurun_out/bench_small_n1.json 2> gpurun_out/bench_small_n1.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload small --no-cpu-baseline > gpurun_out/bench_small_n2.json 2> gpurun_out/bench_small_n2.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --workload c2 --no-cpu-baseline > gpurun_out/bench_c2_n4.json 2> gpurun_out/bench_c2_n4.log
RTREC_BENCH_SAME_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --workload c2 --no-cpu-baseline --score-shard rows > gpurun_out/bench_c2_n4_rows.json 2> gpurun_out/bench_c2_n4_rows.log
python bench.py --workload c2 --no-cpu-baseline > gpurun_out/bench_c2_n1.json 2> gpurun_out/bench_c2_n1.log
tail -2 gpurun_out/bench_small_n2.log gpurun_out/bench_c2_n4.log
cat gpurun_out/bench_small_n1.json gpurun_out/bench_small_n2.json gpurun_out/bench_c2_n1.json gpurun_out/bench_c2_n4.json gpurun_out/bench_c2_n4_rows.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['n_gpus'], d['config']['workload'][:6], d['value'], d['ms_per_step'], d['topk_ids_crc32'], d['fit']['seconds'])
"
