#!/usr/bin/env python3
"""K independent steps (rotating buffer sets) of the per-rank shard of configs[3] (131 072 rows): one stream vs two streams
used alternately (a step's row stores overlap the next step's arithmetic).  python tools/ubench/two_stream_steps.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from rbdreference_amd import RBDReference, iiwa_like
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = torch.device("cuda:0")
rbd = RBDReference(iiwa_like(), build=False)
rng = np.random.default_rng(0)
inputs = [tuple(torch.tensor(x, dtype=torch.float32, device=dev) for x in (rng.uniform(-3, 3, (B, 7)), rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7)))) for _ in range(4)]
step = bench.GradStep(rbd, inputs)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
K = 200
def one():
    for _ in range(K): step()
def two():
    for k in range(K): step.on_stream((s1 if k & 1 else s2).cuda_stream)
def t(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K * 1e6)
    return best
for _ in range(3):
    print(f"B = {B}: one stream {t(one):7.2f} us per step   two streams alternately {t(two):7.2f} us per step")
