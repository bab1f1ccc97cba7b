import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from rbdreference_amd import RBDReference, iiwa_like
r = RBDReference(iiwa_like(), build=False)
B = 1 << 20
rng = np.random.default_rng(0)
q, qd, u = (torch.tensor(x, dtype=torch.float32, device="cuda") for x in (rng.uniform(-3, 3, (B, 7)), rng.uniform(-1, 1, (B, 7)), rng.uniform(-5, 5, (B, 7))))
for _ in range(20): r.forward_dynamics_grad(q, qd, u)
torch.cuda.synchronize()
