#!/usr/bin/env python3
"""What a fork / join of a second stream costs around a ~20 us kernel (events + stream waits), back to back:
   A  minv(atlas, B = 16384) on one stream
   B  the same + a small kernel on a second stream between an event fork and an event join
Decides whether the small root subtrees of a big robot's minv could run as a second, one-lane kernel beside the torso's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, atlas_like, quadruped_like
r = RBDReference(atlas_like(), build=False)
r2 = RBDReference(quadruped_like(), build=False)
B = 16384
q = torch.tensor(np.random.default_rng(0).uniform(-3, 3, (B, 30)), dtype=torch.float32, device="cuda")
q2 = torch.tensor(np.random.default_rng(0).uniform(-3, 3, (B, 12)), dtype=torch.float32, device="cuda")
M = torch.empty((B, 30, 30), dtype=torch.float32, device="cuda")
M2 = torch.empty((B, 12, 12), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream(); s2 = torch.cuda.Stream()
def A():
    r.minv(q, out=M)
def Bf():
    e1 = torch.cuda.Event(); e1.record(s)
    s2.wait_event(e1)
    r.minv(q, out=M)
    with torch.cuda.stream(s2):
        r2.minv(q2, out=M2)
        e2 = torch.cuda.Event(); e2.record(s2)
    s.wait_event(e2)
def C():     # both on one stream
    r.minv(q, out=M); r2.minv(q2, out=M2)
def t(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rnd in range(4):
    print(f"one stream {t(A):7.2f} us   + small kernel same stream {t(C):7.2f} us   fork/join second stream {t(Bf):7.2f} us   small alone {t(lambda: r2.minv(q2, out=M2)):7.2f} us")
