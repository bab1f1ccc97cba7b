"""The C restatement of the oracle (oracle/rbd_oracle.c) against the golden vectors from the real
reference and against the numpy oracle on random inputs (CPU only; gcc)."""
import shutil

import numpy as np
import pytest

from conftest import rel_err
from oracle import rbd_oracle as orc

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")


def test_c_oracle_vs_golden(golden_case):
    from oracle.c_oracle import COracle
    name, robot, g = golden_case
    co = COracle(robot)
    c, dc = co.rnea_grad(g["q"], g["qd"], g["qdd"])
    assert rel_err(c, g["c"]) < 1e-12 and rel_err(dc, g["dc_du"]) < 1e-12
    _, dcd = co.rnea_grad(g["q"], g["qd"], g["qdd"], USE_VELOCITY_DAMPING=True)
    assert rel_err(dcd, g["dc_du_damped"]) < 1e-12
    _, dcn = co.rnea_grad(g["q"], g["qd"])
    assert rel_err(dcn, g["dc_du_noqdd"]) < 1e-12
    c2, v, a, f = co.rnea(g["q"], g["qd"], g["qdd"])
    assert rel_err(v, g["fpass_v"]) < 1e-12 and rel_err(a, g["fpass_a"]) < 1e-12
    assert rel_err(f, g["f_acc"]) < 1e-12 and rel_err(c2, g["c"]) < 1e-12
    assert rel_err(co.minv(g["q"]), g["Minv_dense"]) < 1e-11
    assert rel_err(co.minv(g["q"], output_dense=False), g["Minv_upper"]) < 1e-11   # incl. lower-triangle by-products


def test_c_oracle_vs_numpy_oracle_threads():
    from oracle.c_oracle import COracle
    from rbdreference_amd import atlas_like
    robot = atlas_like()
    co = COracle(robot); om = orc.model_from_robot(robot)
    rng = np.random.default_rng(2)
    q = rng.uniform(-3, 3, (37, 30)); qd = rng.uniform(-1, 1, (37, 30)); qdd = rng.uniform(-1, 1, (37, 30))
    c1, d1 = co.rnea_grad(q, qd, qdd, GRAVITY=-3.7, threads=1)
    c4, d4 = co.rnea_grad(q, qd, qdd, GRAVITY=-3.7, threads=4)
    assert np.array_equal(d1, d4) and np.array_equal(c1, c4)
    cr, dr = orc.rnea_grad(om, q, qd, qdd, GRAVITY=-3.7, return_c=True)
    assert rel_err(d1, dr) < 1e-12 and rel_err(c1, cr) < 1e-12
    assert co.max_threads >= 1
