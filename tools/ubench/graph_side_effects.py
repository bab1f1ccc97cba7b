#!/usr/bin/env python3
"""Does capturing / replaying a HIP graph of the bench launches change the eager launches that follow (bench.py --graph)?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from rbdreference_amd import RBDReference, iiwa_like
import numpy as np
B = int(sys.argv[1]); dev = torch.device("cuda:0")
rbd = RBDReference(iiwa_like(), build=False)
rng = np.random.default_rng(0)
inputs = [tuple(torch.tensor(x, dtype=torch.float32, device=dev) for x in (rng.uniform(-3, 3, (B, 7)), rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7)))) for _ in range(4)]
step = bench.GradStep(rbd, inputs)
def t(n=200):
    for _ in range(20): step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("eager before graph      %.2f us" % t(), "%.2f" % t())
gs = bench.GraphSteps(step, 50, dev)
print("eager after capture     %.2f us" % t(), "%.2f" % t())
gs(); torch.cuda.synchronize()
print("eager after replay      %.2f us" % t(), "%.2f" % t())
g0 = torch.cuda.Event(enable_timing=True); g1 = torch.cuda.Event(enable_timing=True)
g0.record(); gs(); g1.record(); torch.cuda.synchronize()
print("graph replay            %.2f us per launch" % (g0.elapsed_time(g1) / 50 * 1e3))
print("eager after replay 2    %.2f us" % t(), "%.2f" % t())
del gs
import gc; gc.collect(); torch.cuda.synchronize()
print("eager after del         %.2f us" % t(), "%.2f" % t())
torch.cuda.empty_cache()
print("eager after empty_cache %.2f us" % t(), "%.2f" % t())
