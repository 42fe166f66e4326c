cd $GRAFT_REPO_ROOT
for W in c3 c2; do for T in 8192 4096 2048 1024; do
python bench.py --workload $W --no-cpu-baseline --steps 5 --tile-cols $T 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W tile=$T', round(d['roofline']['kernel_ms_avg'],3), round(d['ms_per_step'],3), d['topk_ids_crc32'], d['config']['n_tiles'])"
done; done
