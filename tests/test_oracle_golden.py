"""Pin the CPU oracle (oracle/rbd_oracle.py) to golden vectors produced by the REAL reference
(oracle/gen_golden.py, run in the build container).  fp64, rtol 1e-12, every pass separately --
the per-pass surface the reference's README designates for accelerator testing (README.md:19)."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import rbd_oracle as orc

TOL = 1e-12


def _chk(name, got, want, tol=TOL):
    assert got.shape == want.shape, f"{name}: shape {got.shape} != {want.shape}"
    e = rel_err(got, want)
    assert e <= tol, f"{name}: rel err {e:.3e} > {tol}"


def test_model_extraction(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    assert m.parent == list(g["parent"])
    X = orc.Xmats(m, g["q"])
    for s in (0, len(g["q"]) - 1):
        for i in range(m.n):
            assert np.allclose(X[s, i], robot.get_Xmat_Func_by_id(i)(g["q"][s, i]), rtol=0, atol=1e-13)


def test_rnea_passes(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    v, a, f = orc.rnea_fpass(m, g["q"], g["qd"], g["qdd"])
    _chk("v", v, g["fpass_v"]); _chk("a", a, g["fpass_a"]); _chk("f", f, g["fpass_f"])
    c, f_acc = orc.rnea_bpass(m, g["q"], g["fpass_f"].copy())
    _chk("c", c, g["c"]); _chk("f_acc", f_acc, g["f_acc"])
    c2, v2, a2, f2 = orc.rnea(m, g["q"], g["qd"], g["qdd"])
    _chk("rnea.c", c2, g["c"]); _chk("rnea.f", f2, g["f_acc"])   # accumulated f is returned
    _chk("c_noqdd", orc.rnea(m, g["q"], g["qd"])[0], g["c_noqdd"])


def test_rnea_bpass_is_in_place(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    f = g["fpass_f"].copy()
    _, f_ret = orc.rnea_bpass(m, g["q"], f)
    assert f_ret is f                                     # RBDReference.py:619 mutates its argument


def test_rnea_unbatched_shapes(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    c, v, a, f = orc.rnea(m, g["q"][3], g["qd"][3], g["qdd"][3])
    assert c.shape == (m.n,) and v.shape == (6, m.n) and f.shape == (6, m.n)
    _chk("c[3]", c, g["c"][3]); _chk("f[3]", f, g["f_acc"][3])
    d = orc.rnea_grad(m, list(g["q"][3]), list(g["qd"][3]), list(g["qdd"][3]))   # lists work too
    _chk("dc_du[3]", d, g["dc_du"][3])
    _chk("minv[3]", orc.minv(m, g["q"][3]), g["Minv_dense"][3])


def test_rnea_grad_passes(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    dv, da, df = orc.rnea_grad_fpass_dq(m, g["q"], g["qd"], g["fpass_v"], g["fpass_a"])
    _chk("dq_dv", dv, g["dq_dv"]); _chk("dq_da", da, g["dq_da"]); _chk("dq_df", df, g["dq_df"])
    dv2, da2, df2 = orc.rnea_grad_fpass_dqd(m, g["q"], g["qd"], g["fpass_v"])
    _chk("dqd_dv", dv2, g["dqd_dv"]); _chk("dqd_da", da2, g["dqd_da"]); _chk("dqd_df", df2, g["dqd_df"])
    _chk("dc_dq", orc.rnea_grad_bpass_dq(m, g["q"], g["f_acc"], g["dq_df"].copy()), g["dc_dq"])
    _chk("dc_dqd", orc.rnea_grad_bpass_dqd(m, g["q"], g["dqd_df"].copy()), g["dc_dqd"])
    _chk("dc_dqd_damped", orc.rnea_grad_bpass_dqd(m, g["q"], g["dqd_df"].copy(), True),
         g["dc_dqd_damped"])


def test_rnea_grad(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    _chk("dc_du", orc.rnea_grad(m, g["q"], g["qd"], g["qdd"]), g["dc_du"])
    _chk("dc_du_damped", orc.rnea_grad(m, g["q"], g["qd"], g["qdd"], USE_VELOCITY_DAMPING=True),
         g["dc_du_damped"])
    _chk("dc_du_noqdd", orc.rnea_grad(m, g["q"], g["qd"]), g["dc_du_noqdd"])
    # damping adds damping[i] on dc_dqd[i, i] only (RBDReference.py:1341)
    n = m.n
    diff = g["dc_du_damped"] - g["dc_du"]
    want = np.zeros((n, 2 * n)); want[np.arange(n), n + np.arange(n)] = m.damping
    assert np.allclose(diff, want[None], rtol=0, atol=1e-12)


def test_minv_passes(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    Mb, F, U, D = orc.minv_bpass(m, g["q"])
    _chk("mb_Minv", Mb, g["mb_Minv"]); _chk("mb_F", F, g["mb_F"])
    _chk("mb_U", U, g["mb_U"]); _chk("mb_Dinv", D, g["mb_Dinv"])
    Mf = orc.minv_fpass(m, g["q"], g["mb_Minv"].copy(), g["mb_F"].copy(), g["mb_U"], g["mb_Dinv"])
    _chk("Minv_upper(full, incl. by-products)", Mf, g["Minv_upper"])
    _chk("Minv_dense", orc.minv(m, g["q"], True), g["Minv_dense"])
    iu = np.triu_indices(m.n)
    _chk("triu", orc.minv(m, g["q"], False)[:, iu[0], iu[1]], g["Minv_upper"][:, iu[0], iu[1]])


def test_crba_witness_and_forward_dynamics(golden_case):
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    H = orc.crba(m, g["q"])
    _chk("H", H, g["H"])
    eye = np.einsum("bij,bjk->bik", orc.minv(m, g["q"]), H)
    assert np.max(np.abs(eye - np.eye(m.n)[None])) < 1e-9
    _chk("fd_qdd", orc.forward_dynamics(m, g["q"], g["qd"], g["qdd"]), g["fd_qdd"], 1e-11)
    a, b = orc.forward_dynamics_grad(m, g["q"], g["qd"], g["qdd"])
    _chk("fd_dq", a, g["fd_dq"], 1e-10); _chk("fd_dqd", b, g["fd_dqd"], 1e-10)


def test_aba(golden_case):
    """aba (RBDReference.py:940-1024) against the reference's own aba output and its forward_dynamics."""
    name, robot, g = golden_case
    m = orc.model_from_robot(robot)
    qdd = orc.aba(m, g["q"], g["qd"], g["qdd"])
    _chk("aba_qdd", qdd, g["aba_qdd"], 1e-10)
    _chk("aba vs fd", qdd, g["fd_qdd"], 1e-9)
    assert orc.aba(m, g["q"][0], g["qd"][0], g["qdd"][0]).shape == (m.n,)
    for grav in (0.0, 3.7):
        a = orc.aba(m, g["q"], g["qd"], g["qdd"], GRAVITY=grav)
        c = orc.rnea(m, g["q"], g["qd"], a, GRAVITY=grav)[0]       # rnea(aba(tau)) == tau
        assert np.max(np.abs(c - g["qdd"])) < 1e-9 * max(1.0, np.max(np.abs(a)))


@pytest.mark.parametrize("name", ["iiwa_like", "quadruped_like", "atlas_like", "random_tree_n9"])
def test_invariants_without_reference(name):
    """Maths-only checks (no golden file): finite-difference gradients, Minv H = I, zero gravity."""
    from conftest import make_robot
    m = orc.model_from_robot(make_robot(name))
    rng = np.random.default_rng(5)
    n = m.n
    q = rng.uniform(-np.pi, np.pi, (3, n)); qd = rng.uniform(-1, 1, (3, n)); qdd = rng.uniform(-1, 1, (3, n))
    dc = orc.rnea_grad(m, q, qd, qdd)
    h = 1e-6
    for k in range(n):
        e = np.zeros(n); e[k] = h
        fd_q = (orc.rnea(m, q + e, qd, qdd)[0] - orc.rnea(m, q - e, qd, qdd)[0]) / (2 * h)
        fd_qd = (orc.rnea(m, q, qd + e, qdd)[0] - orc.rnea(m, q, qd - e, qdd)[0]) / (2 * h)
        scale = max(1.0, np.max(np.abs(dc)))
        assert np.max(np.abs(dc[:, :, k] - fd_q)) < 2e-6 * scale
        assert np.max(np.abs(dc[:, :, n + k] - fd_qd)) < 2e-6 * scale
    c0 = orc.rnea(m, q, np.zeros_like(q), np.zeros_like(q), GRAVITY=0.0)[0]
    assert np.all(c0 == 0.0)
    # rnea(q, qd, qdd) = H qdd + rnea(q, qd, 0)
    H = orc.crba(m, q)
    lhs = orc.rnea(m, q, qd, qdd)[0]
    rhs = np.einsum("bij,bj->bi", H, qdd) + orc.rnea(m, q, qd, np.zeros_like(qdd))[0]
    assert np.max(np.abs(lhs - rhs)) < 1e-10 * max(1.0, np.max(np.abs(lhs)))
