cd $GRAFT_REPO_ROOT
for rep in 1 2; do for V in 1 ""; do for W in c2 c3 c4; do
RTREC_AMD_NO_ROWHDR=$V python bench.py --workload $W --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('norowhdr=$V $W', round(d['roofline']['kernel_ms_avg'],3), round(d['ms_per_step'],3), d['topk_ids_crc32'], round(d['roofline']['frac'],3))"
done; done; done
