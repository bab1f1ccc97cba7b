"""Floating base: the reference's floating-base branches of rnea / minv / forward_dynamics
(RBDReference.py:585-593, :652-691, :761-779).

CPU part: the numpy restatement (oracle/rbd_oracle_fb.py) against golden vectors produced by the REAL reference
on duck-typed floating-base robots (oracle/gen_golden.py), the packer's validation, and reference-free
invariants.  GPU part (-m gpu): the HIP kernels of rbd_fb.h through the C-ABI against the same vectors."""
import numpy as np
import pytest

from conftest import fb_golden_names, load_golden, make_robot, rel_err, rel_err_rows
from oracle import rbd_oracle_fb as fbo

TOL = 1e-12


@pytest.mark.parametrize("name", fb_golden_names())
def test_fb_oracle_vs_reference_golden(name):
    g = load_golden(name); robot = make_robot(name); m = fbo.model_from_robot(robot)
    assert m.n == g["q"].shape[1] == m.nb + 5
    v, a, f = fbo.rnea_fpass(m, g["q"], g["qd"], g["qdd"])
    for k, x in (("fpass_v", v), ("fpass_a", a), ("fpass_f", f)):
        assert rel_err(x, g[k]) <= TOL, k
    c, f_acc = fbo.rnea_bpass(m, g["q"], g["fpass_f"])
    assert rel_err(c, g["c"]) <= TOL and rel_err(f_acc, g["f_acc"]) <= TOL
    assert rel_err(fbo.rnea(m, g["q"], g["qd"])[0], g["c_noqdd"]) <= TOL
    Mb, F, U, D = fbo.minv_bpass(m, g["q"])
    for k, x in (("mb_Minv", Mb), ("mb_F", F), ("mb_U", U), ("mb_Dinv", D)):
        assert rel_err(x, g[k]) <= TOL, k
    assert rel_err(fbo.minv(m, g["q"]), g["Minv_dense"]) <= TOL
    assert rel_err(fbo.minv(m, g["q"], output_dense=False), g["Minv_upper"]) <= TOL
    assert rel_err(fbo.forward_dynamics(m, g["q"], g["qd"], g["qdd"]), g["fd_qdd"]) <= 1e-10
    # unbatched call
    c1, v1, _, _ = fbo.rnea(m, g["q"][2], g["qd"][2], g["qdd"][2])
    assert c1.shape == (m.n,) and v1.shape == (6, m.nb) and rel_err(c1, g["c"][2]) <= TOL


@pytest.mark.parametrize("name", fb_golden_names())
def test_fb_invariants(name):
    """Reference-free: Minv H = I with H from rnea columns, and the matrix is the full symmetric inverse."""
    robot = make_robot(name); m = fbo.model_from_robot(robot)
    rng = np.random.default_rng(3)
    q = rng.uniform(-np.pi, np.pi, (4, m.n))
    H = fbo.joint_space_inertia(m, q)
    Mi = fbo.minv(m, q)
    assert np.abs(H - np.swapaxes(H, 1, 2)).max() < 1e-12
    assert np.abs(Mi @ H - np.eye(m.n)).max() < 1e-9
    qd = rng.uniform(-1, 1, (4, m.n)); u = rng.uniform(-1, 1, (4, m.n))
    qdd = fbo.forward_dynamics(m, q, qd, u)
    assert np.abs(fbo.rnea(m, q, qd, qdd)[0] - u).max() < 1e-9


def test_what_the_reference_cannot_do_is_on_record():
    """The fixtures record how the reference's other entry points behave on a floating base (generated with
    the fixture, RBDReference.py line numbers): crba and aba raise; rnea_grad only runs because NB >= 6."""
    for name in fb_golden_names():
        r = [str(x) for x in load_golden(name)["reference_raises"]]
        assert any(x.startswith("crba: IndexError") and x.endswith(":1063") for x in r), r
        assert any(x.startswith("aba:") and x.endswith(":900") for x in r), r


def test_fb_packer_validation():
    from rbdreference_amd import pack_robot
    from rbdreference_amd.packer import emit_header
    robot = make_robot("fb_quadruped_like")
    m = pack_robot(robot)
    assert m.floating and m.n == 13 and m.nv == 18 and m.jtype[0] == 2 and m.parent[0] == -1
    hdr = emit_header(m)
    assert "FLOATING_BASE = true" in hdr and "NV = 18" in hdr

    class BadIdx(type(robot)):
        def get_joint_index_q(self, i):
            return [0, 1, 2, 3, 4, 5] if i == 0 else i       # missing the +5 shift
    bad = BadIdx.__new__(BadIdx); bad.__dict__.update(robot.__dict__)
    with pytest.raises(ValueError):
        pack_robot(bad)

    class BadX(type(robot)):
        def get_Xmat_Func_by_id(self, i):
            f = super().get_Xmat_Func_by_id(i)
            return (lambda q: f(q[::-1])) if i == 0 else f    # another base parametrisation
    bad = BadX.__new__(BadX); bad.__dict__.update(robot.__dict__)
    with pytest.raises(ValueError):
        pack_robot(bad)
    from rbdreference_amd import iiwa_like
    assert not pack_robot(iiwa_like()).floating


# ---- GPU ----------------------------------------------------------------------------------------------
def _rbd(name, _cache={}):
    if name not in _cache:
        from rbdreference_amd import RBDReference
        _cache[name] = RBDReference(make_robot(name), build=False)
    return _cache[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", fb_golden_names())
@pytest.mark.parametrize("precision", ["float32", "float64"])
def test_fb_kernels_vs_golden(name, precision):
    import torch
    dt, tol = (torch.float32, 1e-5) if precision == "float32" else (torch.float64, 1e-11)
    g = load_golden(name); rbd = _rbd(name)
    assert rbd.model.floating and rbd.nv == rbd.n + 5
    q, qd, qdd = (torch.tensor(g[k], device="cuda:0", dtype=dt) for k in ("q", "qd", "qdd"))

    def chk(nm, got, want, t=tol):
        e = rel_err_rows(got.double().cpu().numpy(), want)
        assert e <= t, f"{nm}: {e:.3e} > {t}"
    c, v, a, f = rbd.rnea(q, qd, qdd)
    chk("c", c, g["c"]); chk("v", v, g["fpass_v"]); chk("a", a, g["fpass_a"]); chk("f", f, g["f_acc"])
    chk("c_noqdd", rbd.rnea(q, qd)[0], g["c_noqdd"])
    chk("c only", rbd.rnea(q, qd, qdd, outputs="c")[0], g["c"])
    Mi = rbd.minv(q)
    chk("Minv_dense", Mi, g["Minv_dense"])
    assert torch.equal(Mi, Mi.transpose(1, 2))
    up = rbd.minv(q, output_dense=False)
    chk("Minv_upper", up, np.triu(g["Minv_upper"]))
    # fp32 forward dynamics: bound from cond(H) of each row, H = inverse of the golden Minv
    if dt == torch.float32:
        cond = np.array([np.linalg.cond(M) for M in g["Minv_dense"]])
        got = rbd.forward_dynamics(q, qd, qdd).double().cpu().numpy()
        err = np.max(np.abs(got - g["fd_qdd"]), 1) / np.max(np.abs(g["fd_qdd"]), 1)
        assert np.all(err <= 8.0 * 2.0 ** -24 * cond), (err, cond)
    else:
        chk("fd_qdd", rbd.forward_dynamics(q, qd, qdd), g["fd_qdd"], 1e-9)
    # unbatched + numpy in / numpy out
    out = rbd.minv(g["q"][1])
    assert isinstance(out, np.ndarray) and out.shape == (rbd.nv, rbd.nv) and rel_err(out, g["Minv_dense"][1]) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 65, 1000])
def test_fb_ragged_batches_vs_oracle(B):
    import torch
    name = "fb_quadruped_like"
    rbd = _rbd(name); m = fbo.model_from_robot(make_robot(name))
    rng = np.random.default_rng(B)
    q = rng.uniform(-np.pi, np.pi, (B, m.n)); qd = rng.uniform(-1, 1, (B, m.n)); qdd = rng.uniform(-1, 1, (B, m.n))
    tq, tqd, tqdd = (torch.tensor(x, device="cuda:0", dtype=torch.float64) for x in (q, qd, qdd))
    c, v, a, f = rbd.rnea(tq, tqd, tqdd)
    cr, vr, ar, fr = fbo.rnea(m, q, qd, qdd)
    for got, want in ((c, cr), (v, vr), (a, ar), (f, fr)):
        assert rel_err_rows(got.cpu().numpy(), want) <= 1e-11
    assert rel_err_rows(rbd.minv(tq).cpu().numpy(), fbo.minv(m, q)) <= 1e-11
    assert rel_err_rows(rbd.forward_dynamics(tq, tqd, tqdd).cpu().numpy(), fbo.forward_dynamics(m, q, qd, qdd)) <= 1e-9
    assert rel_err_rows(rbd.minv(tq.float()).double().cpu().numpy(), fbo.minv(m, q)) <= 1e-5


@pytest.mark.gpu
def test_fb_unsupported_entry_points_say_so():
    import torch
    from rbdreference_amd._lib import RBD_ERR_UNSUPPORTED, RbdError
    rbd = _rbd("fb_quadruped_like")
    q = torch.zeros((4, rbd.nv), device="cuda:0", dtype=torch.float32)
    for call in (lambda: rbd.rnea_grad(q, q, q), lambda: rbd.aba(q, q, q), lambda: rbd.forward_dynamics_grad(q, q, q)):
        with pytest.raises(RbdError) as ei:
            call()
        assert ei.value.code == RBD_ERR_UNSUPPORTED and "floating-base" in str(ei.value)
