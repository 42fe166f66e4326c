import sys, os, time, numpy as np, torch
sys.path.insert(0,'.')
from rtrec_amd.engine import SlimEngine
from rtrec_amd.synth import interaction_matrix
U,I=138493,26744
X = interaction_matrix(U, I, 26_000_000, seed=20251003, float_ratings=True)
Xc = X.tocsc(); Xc.sort_indices()
eng = SlimEngine(device="cuda:0"); eng.set_interactions(Xc, X)
nnz = np.diff(Xc.indptr); order = np.argsort(-nnz)
def run(cols, label, mode):
    os.environ["RTREC_AMD_FIT_MODE"]=mode
    torch.cuda.synchronize(); t=time.time()
    tg, items, coef, count, n_iter = eng.fit_columns(cols, nn_feature_selection=50)
    torch.cuda.synchronize(); dt=time.time()-t
    print(mode, label, "n=%d time=%.3fs sweeps max=%d" % (len(cols), dt, n_iter.max()), flush=True)
for mode in ("sw","mw","sw","mw"):
    run(order[:1], "top1", mode)
for mode in ("sw","mw"):
    run(order[:64], "top64", mode)
    run(order[:512], "top512", mode)
    run(order[:2048], "top2048", mode)
