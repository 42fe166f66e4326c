import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
t0=time.time()
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t1=time.time()
from rtrec_amd.engine import SlimEngine
e=SlimEngine(device="cuda:0"); torch.cuda.synchronize()
t2=time.time()
e2=SlimEngine(device="cuda:0"); torch.cuda.synchronize()
t3=time.time()
print(f"torch+context {t1-t0:.2f}s, first engine (with warm-up) {t2-t1:.2f}s, second engine {t3-t2:.3f}s")
