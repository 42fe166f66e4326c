#!/bin/bash
# GPU box: segment-kernel parity tests, then the c3s pass for several thresholds of the workgroup-per-user kernel
# (RTREC_AMD_SG_HEAVY_MIN=v: users of more than v - 1 items get a workgroup; 0 = the library's default).  usage: bash tools/sweep_heavy_min.sh "0 385 257"
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_seg.py -m gpu -x -q > gpurun_out/r04_seg_tests.log 2>&1; tail -2 gpurun_out/r04_seg_tests.log
for HM in ${1:-0 385 257 193}; do
RTREC_AMD_SG_HEAVY_MIN=$HM python bench.py --workload c3s --no-cpu-baseline --steps 5 --no-api --no-structured --stream-batches 0 --no-fast-fit 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hm=$HM c3s', round(d['roofline']['kernel_ms_avg'],3), round(d['ms_per_step'],3), d['topk_ids_crc32'])"
done
