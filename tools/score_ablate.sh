cd $GRAFT_REPO_ROOT
# NOTE: needs a diagnostic build of the library (AB_FLAGS=-DRTREC_DIAGNOSTICS bash tools/ab_build.sh diag; RTREC_AMD_LIB=ab/ab_diag.so):
# the release library ignores rtrec_score_opts.diagnostics.
for W in c3 c2; do
for A in 0 1 2 4 8 15; do
RTREC_AMD_ABLATE=$A python bench.py --workload $W --no-cpu-baseline --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W ablate=$A', round(d['roofline']['kernel_ms_avg'],3))"
done; done
