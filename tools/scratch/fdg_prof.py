import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rbdreference_amd import RBDReference, atlas_like
r = RBDReference(atlas_like(), build=False)
B = 16384; n = 30
rng = np.random.default_rng(0)
q, qd, u = (torch.tensor(x, dtype=torch.float32, device="cuda") for x in (rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-5, 5, (B, n))))
for _ in range(20): r.forward_dynamics_grad(q, qd, u)
torch.cuda.synchronize()
