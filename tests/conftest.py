"""pytest configuration: `gpu` marker + shared fixtures.

`-m "not gpu"` runs on the GPU-less build container (oracle vs golden vectors, host logic, C-ABI
symbol checks, gloo sharding); `-m gpu` are the parity tests proper and call the HIP kernels
through the C-ABI on a real MI355X.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _golden_names():
    return sorted(f[len("golden_"):-len(".npz")] for f in os.listdir(GOLDEN_DIR)
                  if f.startswith("golden_") and f.endswith(".npz"))


def all_golden_names():
    """Fixed-base fixtures (every pass of the path)."""
    return [n for n in _golden_names() if not n.startswith("fb_")]


def fb_golden_names():
    """Floating-base fixtures (rnea, minv, forward_dynamics: what the reference can do with such a robot)."""
    return [n for n in _golden_names() if n.startswith("fb_")]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"golden_{name}.npz")))


def make_robot(name):
    """Rebuild the robot a golden fixture was generated with (see oracle/gen_golden.py)."""
    from rbdreference_amd.robot import BUILTIN_ROBOTS, FloatingBaseRobot, floating_quadruped_like, random_tree
    if name in BUILTIN_ROBOTS:
        return BUILTIN_ROBOTS[name]()
    if name == "fb_quadruped_like":
        return floating_quadruped_like()
    if name == "fb_random_tree_n6":
        return FloatingBaseRobot(random_tree([-1, 0, 1, 0, 3, 3], seed=5, name="t6"), name)
    if name == "fb_random_tree_n4":
        return FloatingBaseRobot(random_tree([-1, 0, 1, 1], seed=9, name="t4"), name)
    if name == "random_tree_n9":
        return random_tree([-1, 0, 1, 1, 3, -1, 5, 5, 7], seed=7, name=name)
    if name == "random_chain_n7":
        return random_tree([-1, 0, 1, 2, 3, 4, 5], seed=21, name=name)
    if name == "random_prismatic_n6":
        return random_tree([-1, 0, 1, 2, 2, 4], seed=11, prismatic_every=3, name=name)
    if name == "random_limbs_n14":      # one root, three 4-body limbs under body 1 (segment waves)
        return random_tree([-1, 0, 1, 2, 3, 4, 1, 6, 7, 8, 1, 10, 11, 12], seed=41, name=name)
    if name == "random_forest_n8":      # interleaved root subtrees + a branch (no per-root groups)
        return random_tree([-1, -1, 0, 1, 2, 0, 3, 5], seed=33, name=name)
    if name == "random_twochains_n18":  # two independent nine-body chains (fp64 workspace tree kernel, one block per root)
        return random_tree([-1, 0, 1, 2, 3, 4, 5, 6, 7, -1, 9, 10, 11, 12, 13, 14, 15, 16], seed=57, name=name)
    raise KeyError(name)


def rel_err(x, ref):
    """Per-tensor normwise relative error  max|x - ref| / max|ref|  (SURVEY.md §8d)."""
    x = np.asarray(x, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    d = np.max(np.abs(x - ref)) if x.size else 0.0
    s = np.max(np.abs(ref)) if ref.size else 0.0
    return d / s if s > 0 else d


def rel_err_rows(x, ref):
    """Per-row (per-configuration) normwise relative error; returns the worst row."""
    x = np.asarray(x, dtype=np.float64).reshape(x.shape[0], -1)
    ref = np.asarray(ref, dtype=np.float64).reshape(ref.shape[0], -1)
    d = np.max(np.abs(x - ref), axis=1)
    s = np.max(np.abs(ref), axis=1)
    return float(np.max(d / np.where(s > 0, s, 1.0)))


@pytest.fixture(params=all_golden_names())
def golden_case(request):
    name = request.param
    return name, make_robot(name), load_golden(name)
