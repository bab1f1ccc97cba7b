"""GPU tests of the sin / cos paths no ordinary input reaches (run with ``-m gpu``): huge angles, NaN and +-Inf in q.

Why this file exists (docs/NOTEBOOK.md 3.1 a' (xv)): round 3 hit an aperture fault in
`rnea_grad_idsva_kernel<double>` whose cause sat inside the device library's `sincos(double)` -- a lane-masked
if / else in whose EXEC = 0 window the register allocator had placed a copy of a caller value.  Round 3 routed
|q| <= 1e6 around the library; round 4 removed the library routines from the device code altogether
(rbdreference_amd/csrc/rbd_sincos.h: own fast paths + an own BRANCH-FREE Payne-Hanek path for every other q, NaN for
non-finite q).  These tests take exactly the path that used to be the library's, in fp64 and fp32, through every
gradient kernel family with AGPR-resident state, `rnea`, `minv`, and the model-handle library:

  * rows with |q| up to 3e6 agree with the oracle (which evaluates numpy's sin / cos of the same angles,
    RBDReference.py:562-564, :574: fp64 arithmetic, Xmat(q) for any q) at 1e-9 (fp64) / 1e-5 (fp32, oracle fed the
    float32-rounded angles),
  * a row that holds a NaN or an Inf is non-finite IN ITS OWN ROW ONLY: every other row of the batch -- same wave or
    not -- still agrees with the oracle, and the process survives (a diverged MPC rollout hands over exactly such rows).
"""
import warnings

import numpy as np
import pytest

from conftest import make_robot, rel_err_rows

pytestmark = pytest.mark.gpu

B = 64 * 4 + 37          # four full waves / tiles and a ragged tail


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch


_RBD = {}


def rbd_for(name, generic="never"):
    key = (name, generic)
    if key not in _RBD:
        from rbdreference_amd import RBDReference
        _RBD[key] = RBDReference(make_robot(name), build=False, generic=generic)
    return _RBD[key]


def oracle_for(name):
    robot = make_robot(name)
    if getattr(robot, "floating_base", False):
        from oracle import rbd_oracle_fb as o
    else:
        from oracle import rbd_oracle as o
    return o, o.model_from_robot(robot)


def poisoned_batch(n, seed, floating):
    """q with: wave 0 -- every row huge (|q| up to 3e6: the whole wave on the wide path, all finite); wave 1 -- ordinary
    rows with one NaN row and one huge row among them; wave 2 -- ordinary rows with a +Inf and a -Inf row; wave 3 and
    the tail -- ordinary rows (fast path), one NaN in the tail.  Returns q, qd, qdd and the set of poisoned rows."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (B, n)); qd = rng.uniform(-1, 1, (B, n)); qdd = rng.uniform(-1, 1, (B, n))
    j0 = 3 if floating else 0                      # floating base: q[0:3] is a position (no sin / cos), q[3:6] the base angles
    q[:64, j0:] = rng.uniform(-3e6, 3e6, (64, n - j0))
    q[5, j0] = 3.0e6; q[6, n - 1] = -3.0e6
    q[64 + 9, j0:] = rng.uniform(1e6, 3e6, n - j0)
    bad = {64 + 17: np.nan, 128 + 3: np.inf, 128 + 40: -np.inf, 256 + 20: np.nan}
    cols = {64 + 17: n - 1, 128 + 3: j0, 128 + 40: n // 2 if n // 2 >= j0 else j0, 256 + 20: j0 + 1 if j0 + 1 < n else j0}
    for r, val in bad.items():
        q[r, cols[r]] = val
    return q, qd, qdd, sorted(bad)


def check_rows(what, got, ref, bad, tol):
    got = got.double().cpu().numpy().reshape(B, -1); ref = np.asarray(ref).reshape(B, -1)
    good = np.array([r for r in range(B) if r not in bad])
    assert np.all(np.isfinite(got[good])), f"{what}: a poisoned row leaked into rows {good[~np.isfinite(got[good]).all(1)][:8]}"
    e = rel_err_rows(got[good], ref[good])
    assert e <= tol, f"{what}: finite rows differ from the oracle: {e:.3e} > {tol}"
    for r in bad:
        # what a NaN / Inf angle does to a row is the ORACLE's to say: an output that does not depend on the poisoned
        # joint stays finite in the reference too (Minv of a fixed-base robot never reads the root joint's transform,
        # RBDReference.py:711-733, :762-781)
        want_bad = not np.all(np.isfinite(ref[r]))
        got_bad = not np.all(np.isfinite(got[r]))
        # ... and a kernel may poison MORE of the row than the reference does, never less: the world-frame kernels use
        # every joint's rotation in every entry, while the reference's velocity derivatives never multiply by the root's
        # transform (v_parent = 0 is not multiplied at :576-581), so e.g. fd_dqd of a row whose ROOT angle is Inf is
        # finite there and NaN here.  What must hold: non-finite where the reference is, and right wherever finite.
        assert got_bad or not want_bad, f"{what}: row {r} holds a NaN / Inf angle and the oracle's row is non-finite, the kernel's is finite"
        if not got_bad:
            assert rel_err_rows(got[r:r + 1], ref[r:r + 1]) <= tol, f"{what}: row {r}"
    return e


# (robot, gradient-kernel options to force; the name each must report)
CASES = [
    ("iiwa_like", {"BATCH": "idsva", "TREE": "tree_ws", "COLS": None}),
    ("random_chain_n7", {"BATCH": "idsva", "TREE": "tree_ws"}),
    ("atlas_like", {"AUTO": "tree_ws"}),
    ("random_twochains_n18", {"AUTO": "tree_ws"}),
    ("quadruped_like", {"BATCH": "rnea_grad_kernel", "TREE": None, "COLS": None}),
    ("random_tree_n9", {"BATCH": "rnea_grad_kernel", "TREE": None, "COLS": None}),
    ("fb_quadruped_like", {"AUTO": None, "COLS": None}),
]


@pytest.mark.parametrize("prec", ["float64", "float32"])
@pytest.mark.parametrize("name,kernels", CASES, ids=[c[0] for c in CASES])
def test_huge_nan_and_inf_angles_every_kernel_family(name, kernels, prec):
    torch = _torch()
    from rbdreference_amd import _lib as L
    dt = torch.float64 if prec == "float64" else torch.float32
    tol = 1e-9 if prec == "float64" else 1e-5
    rbd = rbd_for(name); o, om = oracle_for(name)
    floating = bool(getattr(make_robot(name), "floating_base", False))
    q, qd, qdd, bad = poisoned_batch(rbd.nv, 1234, floating)
    if prec == "float32":                          # the oracle sees the angles the kernel sees
        q = q.astype(np.float32).astype(np.float64); qd = qd.astype(np.float32).astype(np.float64); qdd = qdd.astype(np.float32).astype(np.float64)
    tq, tqd, tqdd = (torch.tensor(x, device="cuda:0", dtype=dt) for x in (q, qd, qdd))
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        c_ref, dc_ref = o.rnea_grad(om, q, qd, qdd, return_c=True) if not floating else (o.rnea(om, q, qd, qdd)[0], o.rnea_grad(om, q, qd, qdd))
        _, v_ref, a_ref, f_ref = o.rnea(om, q, qd, qdd)
        Mi_ref = o.minv(om, q)
    # rnea (c, v, a, f) and minv
    c, v, a, f = rbd.rnea(tq, tqd, tqdd)
    torch.cuda.synchronize()
    check_rows("rnea c", c, c_ref, bad, tol); check_rows("rnea f", f, f_ref, bad, tol)
    Mi = rbd.minv(tq)
    torch.cuda.synchronize()
    check_rows("minv", Mi, Mi_ref, bad, tol * (10 if prec == "float64" else 1))
    # rnea_grad through every kernel family the robot has
    opt = {"AUTO": L.RBD_GRAD_KERNEL_AUTO, "BATCH": L.RBD_GRAD_KERNEL_BATCH, "TREE": L.RBD_GRAD_KERNEL_TREE, "COLS": L.RBD_GRAD_KERNEL_COLS}
    seen = set()
    try:
        for k, want in kernels.items():
            rbd._lib.set_option(L.RBD_OPT_GRAD_KERNEL, opt[k])
            kn = rbd._lib.kernel_name(L.RBD_OP_RNEA_GRAD, 8 if prec == "float64" else 4, B)
            if want is not None and prec == "float64":
                assert want in kn, (name, k, kn)
            if kn in seen:
                continue
            seen.add(kn)
            cg, dc = rbd.rnea_grad(tq, tqd, tqdd, return_c=True)
            torch.cuda.synchronize()
            check_rows(f"rnea_grad dc_du [{kn}]", dc, dc_ref, bad, tol)
            check_rows(f"rnea_grad c [{kn}]", cg, c_ref, bad, tol)
    finally:
        rbd._lib.set_option(L.RBD_OPT_GRAD_KERNEL, L.RBD_GRAD_KERNEL_AUTO)


@pytest.mark.parametrize("prec", ["float64", "float32"])
@pytest.mark.parametrize("name", ["iiwa_like", "atlas_like", "random_tree_n9", "fb_quadruped_like"])
def test_huge_nan_and_inf_angles_model_handle_library(name, prec):
    """The model-handle library (csrc_generic/rbd_generic.hip) called `sincos(double)` unconditionally; it uses
    rbd_sincos.h now.  Same batch, same demands."""
    torch = _torch()
    dt = torch.float64 if prec == "float64" else torch.float32
    tol = 1e-9 if prec == "float64" else 1e-5
    gen = rbd_for(name, generic="only"); o, om = oracle_for(name)
    floating = bool(getattr(make_robot(name), "floating_base", False))
    q, qd, qdd, bad = poisoned_batch(gen.nv, 4321, floating)
    if prec == "float32":
        q = q.astype(np.float32).astype(np.float64); qd = qd.astype(np.float32).astype(np.float64); qdd = qdd.astype(np.float32).astype(np.float64)
    tq, tqd, tqdd = (torch.tensor(x, device="cuda:0", dtype=dt) for x in (q, qd, qdd))
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        c_ref = o.rnea(om, q, qd, qdd)[0]
        dc_ref = o.rnea_grad(om, q, qd, qdd)
        Mi_ref = o.minv(om, q)
    c = gen.rnea(tq, tqd, tqdd)[0]
    dc = gen.rnea_grad(tq, tqd, tqdd)
    Mi = gen.minv(tq)
    torch.cuda.synchronize()
    check_rows("generic rnea c", c, c_ref, bad, tol)
    check_rows("generic dc_du", dc, dc_ref, bad, tol)
    check_rows("generic minv", Mi, Mi_ref, bad, tol * (10 if prec == "float64" else 1))


def test_forward_dynamics_grad_with_poisoned_rows():
    """forward_dynamics_grad composes rnea, minv and rnea_grad (fused epilogues): same demands on the composition."""
    torch = _torch()
    for name in ("iiwa_like", "quadruped_like"):
        rbd = rbd_for(name); o, om = oracle_for(name)
        q, qd, u, bad = poisoned_batch(rbd.nv, 99, False)
        tq, tqd, tu = (torch.tensor(x, device="cuda:0", dtype=torch.float64) for x in (q, qd, u))
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            a1, a2 = o.forward_dynamics_grad(om, q, qd, u)
        g1, g2 = rbd.forward_dynamics_grad(tq, tqd, tu)
        torch.cuda.synchronize()
        check_rows(f"{name} fd_dq", g1, a1, bad, 1e-7); check_rows(f"{name} fd_dqd", g2, a2, bad, 1e-7)
