"""Floating base: the reference's floating-base branches of rnea / rnea_grad / minv /
forward_dynamics (RBDReference.py:585-593, :652-691, :761-779, :1141-1341).

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
    if m.nb >= 6:
        assert rel_err(fbo.rnea_grad(m, g["q"], g["qd"], g["qdd"]), g["dc_du"]) <= TOL
        assert rel_err(fbo.rnea_grad(m, g["q"], g["qd"]), g["dc_du_noqdd"]) <= TOL
        assert rel_err(fbo.rnea_grad(m, g["q"], g["qd"], g["qdd"], USE_VELOCITY_DAMPING=True), g["dc_du_damped"]) <= TOL
        assert fbo.rnea_grad(m, g["q"][1], g["qd"][1], g["qdd"][1]).shape == (m.n, 2 * m.n)
        # the four gradient passes and forward_dynamics_grad (the reference runs them on floating bases: README.md:19)
        dv, da, df = fbo.rnea_grad_fpass_dq(m, g["q"], g["qd"], g["fpass_v"], g["fpass_a"])
        for k, x in (("dq_dv", dv), ("dq_da", da), ("dq_df", df)):
            assert rel_err(x, g[k]) <= TOL, k
        dv, da, df = fbo.rnea_grad_fpass_dqd(m, g["q"], g["qd"], g["fpass_v"])
        for k, x in (("dqd_dv", dv), ("dqd_da", da), ("dqd_df", df)):
            assert rel_err(x, g[k]) <= TOL, k
        dc, dfa = fbo.rnea_grad_bpass_dq(m, g["q"], g["f_acc"], g["dq_df"])
        assert rel_err(dc, g["dc_dq"]) <= TOL and rel_err(dfa, g["dq_df_after"]) <= TOL
        dc, dfa = fbo.rnea_grad_bpass_dqd(m, g["q"], g["dqd_df"])
        assert rel_err(dc, g["dc_dqd"]) <= TOL and rel_err(dfa, g["dqd_df_after"]) <= TOL
        assert rel_err(fbo.rnea_grad_bpass_dqd(m, g["q"], g["dqd_df"], True)[0], g["dc_dqd_damped"]) <= TOL
        a1, a2 = fbo.forward_dynamics_grad(m, g["q"], g["qd"], g["qdd"])
        assert rel_err(a1, g["fd_dq"]) <= 1e-10 and rel_err(a2, g["fd_dqd"]) <= 1e-10
        assert np.array_equal(np.hstack([g["dc_dq"][0], g["dc_dqd"][0]]), g["dc_du"][0])      # :1367
    else:
        assert "dc_du" not in g
        with pytest.raises(AssertionError):
            fbo.rnea_grad(m, g["q"], g["qd"], g["qdd"])


@pytest.mark.parametrize("name", [n for n in fb_golden_names() if n != "fb_random_tree_n4"])
def test_fb_rnea_grad_is_the_derivative_of_rnea(name):
    """Reference-free: central differences of rnea.  Joint columns (>= 6) of dc_dq and every column of dc_dqd
    are coordinate derivatives; the base's six dc_dq columns are derivatives along a base-frame twist
    (X_0 <- (1 - crm(e_k) h) X_0, :1168-1175), which moves gravity only: a_0 <- a_0 + h crm(X_0 a_grav) e_k."""
    robot = make_robot(name); m = fbo.model_from_robot(robot)
    rng = np.random.default_rng(5)
    q = rng.uniform(-np.pi, np.pi, (2, m.n)); qd = rng.uniform(-1, 1, (2, m.n)); qdd = rng.uniform(-1, 1, (2, m.n))
    dc = fbo.rnea_grad(m, q, qd, qdd)
    h = 1e-6
    for j in range(m.n):
        e = np.zeros(m.n); e[j] = h
        fd_qd = (fbo.rnea(m, q, qd + e, qdd)[0] - fbo.rnea(m, q, qd - e, qdd)[0]) / (2 * h)
        assert np.abs(fd_qd - dc[:, :, m.n + j]).max() <= 1e-6 * max(1.0, np.abs(dc[:, :, m.n + j]).max()), ("dqd", j)
        if j >= 6:
            fd_q = (fbo.rnea(m, q + e, qd, qdd)[0] - fbo.rnea(m, q - e, qd, qdd)[0]) / (2 * h)
            assert np.abs(fd_q - dc[:, :, j]).max() <= 1e-6 * max(1.0, np.abs(dc[:, :, j]).max()), ("dq", j)
    # base twist columns: c is affine in a_0 through qdd[0:6], so the perturbed gravity term can ride on qdd
    X0 = fbo.Xmats(m, q)[:, 0]
    g0 = np.zeros(6); g0[5] = 9.81
    from oracle import rbd_oracle as fx
    for k in range(6):
        ek = np.zeros(6); ek[k] = 1.0
        da0 = np.einsum("bij,j->bi", fx._crm(X0 @ g0), ek)
        qp = qdd.copy(); qp[:, 0:6] += h * da0
        qm = qdd.copy(); qm[:, 0:6] -= h * da0
        fd = (fbo.rnea(m, q, qd, qp)[0] - fbo.rnea(m, q, qd, qm)[0]) / (2 * h)
        assert np.abs(fd - dc[:, :, k]).max() <= 1e-6 * max(1.0, np.abs(dc[:, :, k]).max()), ("base", k)


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
    the fixture, RBDReference.py line numbers): crba and aba raise; rnea_grad runs only when NB >= 6."""
    for name in fb_golden_names():
        g = load_golden(name)
        r = [str(x) for x in g["reference_raises"]]
        assert any(x.startswith("crba: IndexError") and x.endswith(":1063") for x in r), r
        assert any(x.startswith("aba:") and x.endswith(":900") for x in r), r
        if g["parent"].shape[0] >= 6:
            assert "rnea_grad: ran" in r, r
        else:
            assert any(x.startswith("rnea_grad: IndexError") and x.endswith(":1168") for x in r), r


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
    if rbd.n >= 6:
        tg = 1e-5 if dt == torch.float32 else 1e-11
        chk("dc_du", rbd.rnea_grad(q, qd, qdd), g["dc_du"], tg)
        chk("dc_du_noqdd", rbd.rnea_grad(q, qd), g["dc_du_noqdd"], tg)
        chk("dc_du_damped", rbd.rnea_grad(q, qd, qdd, USE_VELOCITY_DAMPING=True), g["dc_du_damped"], tg)
        c1, dc1 = rbd.rnea_grad(q, qd, qdd, return_c=True)
        chk("c of rnea_grad", c1, g["c"]); chk("dc_du (with c)", dc1, g["dc_du"], tg)
        c2, v2, a2, f2, dc2 = rbd.rnea_and_grad(q, qd, qdd)
        assert torch.equal(dc2, dc1) and torch.equal(c2, c) and torch.equal(v2, v) and torch.equal(a2, a) and torch.equal(f2, f)
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
    dcr = fbo.rnea_grad(m, q, qd, qdd)
    assert rel_err_rows(rbd.rnea_grad(tq, tqd, tqdd).cpu().numpy(), dcr) <= 1e-11
    assert rel_err_rows(rbd.rnea_grad(tq.float(), tqd.float(), tqdd.float()).double().cpu().numpy(), dcr) <= 1e-5


@pytest.mark.gpu
def test_fb_unsupported_entry_points_say_so():
    import torch
    from rbdreference_amd._lib import RBD_ERR_UNSUPPORTED, RbdError
    rbd = _rbd("fb_quadruped_like")
    q = torch.zeros((4, rbd.nv), device="cuda:0", dtype=torch.float32)
    for call in (lambda: rbd.crba(q), lambda: rbd.aba(q, q, q)):      # the reference's own crba / aba raise (:1063, :900)
        with pytest.raises(RbdError) as ei:
            call()
        assert ei.value.code == RBD_ERR_UNSUPPORTED and "floating-base" in str(ei.value)
    # fewer than six bodies: the reference's own rnea_grad raises IndexError (:1168, on record in the fixture)
    small = _rbd("fb_random_tree_n4")
    q = torch.zeros((4, small.nv), device="cuda:0", dtype=torch.float64)
    v = torch.zeros((4, 6, small.n), device="cuda:0", dtype=torch.float64)
    for call in (lambda: small.rnea_grad(q, q, q), lambda: small.rnea_and_grad(q, q, q), lambda: small.forward_dynamics_grad(q, q, q),
                 lambda: small.rnea_grad_fpass_dq(q, q, v, v), lambda: small.rnea_grad_fpass_dqd(q, q, v)):
        with pytest.raises(RbdError) as ei:
            call()
        assert ei.value.code == RBD_ERR_UNSUPPORTED and "NB = 4" in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("name", fb_golden_names())
@pytest.mark.parametrize("dtname", ["f64", "f32"])
def test_fb_passes_vs_golden(name, dtname):
    """The per-pass surface (README.md:19) for floating bases against the real reference's outputs: rnea_fpass /
    rnea_bpass (in-place f), the four gradient passes (in-place df), minv_bpass / minv_fpass, forward_dynamics_grad."""
    import torch
    dt = torch.float64 if dtname == "f64" else torch.float32
    tol = 1e-11 if dt == torch.float64 else 1e-5
    g = load_golden(name); rbd = _rbd(name)
    T = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda:0", dtype=dt)
    q, qd, qdd = T(g["q"]), T(g["qd"]), T(g["qdd"])

    def chk(nm, got, want, t=tol):
        e = rel_err_rows(got.double().cpu().numpy(), want)
        assert e <= t, f"{nm}: {e:.3e} > {t}"
    v, a, f = rbd.rnea_fpass(q, qd, qdd)
    chk("fpass_v", v, g["fpass_v"]); chk("fpass_a", a, g["fpass_a"]); chk("fpass_f", f, g["fpass_f"])
    f_in = T(g["fpass_f"])
    c, f_ret = rbd.rnea_bpass(q, f_in)
    assert f_ret is f_in                                                   # accumulated IN PLACE (:619)
    chk("c", c, g["c"]); chk("f_acc", f_in, g["f_acc"])
    Mb, F, U, D = rbd.minv_bpass(q)
    chk("mb_Minv", Mb, g["mb_Minv"]); chk("mb_F", F, g["mb_F"]); chk("mb_U", U, g["mb_U"]); chk("mb_Dinv", D, g["mb_Dinv"])
    Min = T(g["mb_Minv"]); Fin = torch.full_like(T(g["mb_F"]), 7.0)        # F is rebuilt: its incoming values are not read (:774-781)
    Mo = rbd.minv_fpass(q, Min, Fin, T(g["mb_U"]), T(g["mb_Dinv"]))
    chk("minv_fpass", Mo, g["Minv_upper"])                                  # a floating base completes every row (:779)
    if rbd.n < 6:
        return
    dv, da, df = rbd.rnea_grad_fpass_dq(q, qd, T(g["fpass_v"]), T(g["fpass_a"]))
    chk("dq_dv", dv, g["dq_dv"]); chk("dq_da", da, g["dq_da"]); chk("dq_df", df, g["dq_df"])
    dv, da, df = rbd.rnea_grad_fpass_dqd(q, qd, T(g["fpass_v"]))
    chk("dqd_dv", dv, g["dqd_dv"]); chk("dqd_da", da, g["dqd_da"]); chk("dqd_df", df, g["dqd_df"])
    dfin = T(g["dq_df"])
    chk("dc_dq", rbd.rnea_grad_bpass_dq(q, T(g["f_acc"]), dfin), g["dc_dq"])
    chk("dq_df after the pass", dfin, g["dq_df_after"])                    # (:1291-1294)
    dfin = T(g["dqd_df"])
    chk("dc_dqd", rbd.rnea_grad_bpass_dqd(q, dfin), g["dc_dqd"])
    chk("dqd_df after the pass", dfin, g["dqd_df_after"])                  # (:1331)
    chk("dc_dqd_damped", rbd.rnea_grad_bpass_dqd(q, T(g["dqd_df"]), True), g["dc_dqd_damped"])
    # forward_dynamics_grad: fp32 is multiplied by Minv (cond 1e2..1e4): bound from cond(H) per row
    a1, a2 = rbd.forward_dynamics_grad(q, qd, qdd)
    if dt == torch.float64:
        chk("fd_dq", a1, g["fd_dq"], 1e-9); chk("fd_dqd", a2, g["fd_dqd"], 1e-9)
    else:
        cond = np.array([np.linalg.cond(M) for M in g["Minv_dense"]])
        for nm, got in (("fd_dq", a1), ("fd_dqd", a2)):
            gn = got.double().cpu().numpy()
            err = np.abs(gn - g[nm]).reshape(len(cond), -1).max(1) / np.abs(g[nm]).reshape(len(cond), -1).max(1)
            assert np.all(err <= 16.0 * 2.0 ** -24 * cond), (nm, err, cond)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in fb_golden_names() if n != "fb_random_tree_n4"])
def test_fb_both_gradient_kernels_vs_golden(name):
    """rnea_grad on the world-frame kernel (AUTO) and on the column recursion (COLS), damped and not."""
    import torch
    from rbdreference_amd._lib import RBD_GRAD_KERNEL_AUTO, RBD_GRAD_KERNEL_COLS, RBD_OPT_GRAD_KERNEL
    g = load_golden(name); rbd = _rbd(name)
    try:
        for opt, want_kernel in ((RBD_GRAD_KERNEL_AUTO, "rnea_grad_fbw_kernel"), (RBD_GRAD_KERNEL_COLS, "rnea_grad_fb_kernel")):
            rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, opt)
            for dt, tol in ((torch.float64, 1e-11), (torch.float32, 1e-5)):
                assert rbd._lib.kernel_name(1, 8 if dt == torch.float64 else 4, 8).startswith(want_kernel)
                q, qd, qdd = (torch.tensor(g[k], device="cuda:0", dtype=dt) for k in ("q", "qd", "qdd"))
                for kw, key in (({}, "dc_du"), ({"USE_VELOCITY_DAMPING": True}, "dc_du_damped")):
                    e = rel_err_rows(rbd.rnea_grad(q, qd, qdd, **kw).double().cpu().numpy(), g[key])
                    assert e <= tol, (want_kernel, dt, key, e)
                e = rel_err_rows(rbd.rnea_grad(q, qd).double().cpu().numpy(), g["dc_du_noqdd"])
                assert e <= tol, (want_kernel, dt, "noqdd", e)
    finally:
        rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_AUTO)


@pytest.mark.gpu
@pytest.mark.parametrize("dtname", ["f32", "f64"])
def test_fb_full_size_sampled_rows(dtname):
    """B = 65 536 (the size the floating-base timings are quoted at): 256 sampled rows of every product against the
    oracle, exact symmetry of Minv, and Minv (c(qdd) - c(0)) = qdd on every row."""
    import torch
    dt = torch.float64 if dtname == "f64" else torch.float32
    tol = 1e-11 if dt == torch.float64 else 1e-5
    name = "fb_quadruped_like"; B = 65536
    rbd = _rbd(name); m = fbo.model_from_robot(make_robot(name))
    rng = np.random.default_rng(77)
    q = rng.uniform(-np.pi, np.pi, (B, m.n)); qd = rng.uniform(-1, 1, (B, m.n)); qdd = rng.uniform(-1, 1, (B, m.n))
    tq, tqd, tqdd = (torch.tensor(x, device="cuda:0", dtype=dt) for x in (q, qd, qdd))
    rows = rng.choice(B, 256, replace=False)
    c, v, a, f = rbd.rnea(tq, tqd, tqdd)
    dc = rbd.rnea_grad(tq, tqd, tqdd)
    Mi = rbd.minv(tq)
    cr, vr, ar, fr = fbo.rnea(m, q[rows], qd[rows], qdd[rows])
    for nm, got, want in (("c", c, cr), ("v", v, vr), ("a", a, ar), ("f", f, fr),
                          ("dc_du", dc, fbo.rnea_grad(m, q[rows], qd[rows], qdd[rows])), ("Minv", Mi, fbo.minv(m, q[rows]))):
        e = rel_err_rows(got[rows].double().cpu().numpy(), want)
        assert e <= tol, (nm, e)
    assert torch.equal(Mi, Mi.transpose(1, 2))
    c0 = rbd.rnea(tq, tqd, torch.zeros_like(tqdd), outputs="c")[0]
    back = torch.einsum("bij,bj->bi", Mi.double(), (c - c0).double())
    err = (back - tqdd.double()).abs().amax(1) / tqdd.double().abs().amax(1)
    assert float(err.max()) <= (1e-9 if dt == torch.float64 else 2e-2), float(err.max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", fb_golden_names())
def test_fb_both_minv_kernels_vs_golden(name):
    """minv on the wave-per-subtree kernel (AUTO, rbd_fb_minv.h) and on the four-lanes kernel (RBD_MINV_PHASE_A_LANE),
    dense and upper, both precisions, and a ragged batch that ends inside a 16-configuration output image."""
    import torch
    from rbdreference_amd._lib import RBD_MINV_PHASE_A_AUTO, RBD_MINV_PHASE_A_LANE, RBD_OPT_MINV_PHASE_A
    g = load_golden(name); rbd = _rbd(name); m = fbo.model_from_robot(make_robot(name))
    rng = np.random.default_rng(3)
    qr = rng.uniform(-np.pi, np.pi, (83, m.n))
    Mr = fbo.minv(m, qr)
    try:
        for opt, want in ((RBD_MINV_PHASE_A_AUTO, "minv_fbm_kernel"), (RBD_MINV_PHASE_A_LANE, "minv_fb_kernel")):
            rbd._lib.set_option(RBD_OPT_MINV_PHASE_A, opt)
            for dt, tol in ((torch.float64, 1e-11), (torch.float32, 1e-5)):
                assert rbd._lib.kernel_name(2, 8 if dt == torch.float64 else 4, 8).startswith(want)
                q = torch.tensor(g["q"], device="cuda:0", dtype=dt)
                Mi = rbd.minv(q)
                assert rel_err_rows(Mi.double().cpu().numpy(), g["Minv_dense"]) <= tol, (want, dt)
                assert torch.equal(Mi, Mi.transpose(1, 2))
                assert rel_err_rows(rbd.minv(q, output_dense=False).double().cpu().numpy(), np.triu(g["Minv_upper"])) <= tol
                e = rel_err_rows(rbd.minv(torch.tensor(qr, device="cuda:0", dtype=dt)).double().cpu().numpy(), Mr)
                assert e <= tol, (want, dt, "ragged", e)
    finally:
        rbd._lib.set_option(RBD_OPT_MINV_PHASE_A, RBD_MINV_PHASE_A_AUTO)


@pytest.mark.gpu
@pytest.mark.parametrize("name", fb_golden_names())
def test_fb_forward_dynamics_one_launch_and_three_launch_paths(name):
    """forward_dynamics(_grad) of a floating-base robot on both paths: AUTO = ONE launch (the wave-per-subtree minv kernel
    computes the bias force from qd and qdd = Minv (u - c) itself, rbd_fb_minv.h) and RBD_MINV_PHASE_A_LANE = the c-only rnea
    launch + the four-lanes minv + the product kernel; both against the oracle (RBDReference.py:1371-1384), both precisions, a
    ragged batch, non-default gravity."""
    import torch
    from rbdreference_amd._lib import RBD_MINV_PHASE_A_AUTO, RBD_MINV_PHASE_A_LANE, RBD_OPT_MINV_PHASE_A
    rbd = _rbd(name); m = fbo.model_from_robot(make_robot(name))
    rng = np.random.default_rng(11)
    B = 83
    q, qd, u = rng.uniform(-np.pi, np.pi, (B, m.n)), rng.uniform(-1, 1, (B, m.n)), rng.uniform(-2, 2, (B, m.n))
    for grav in (-9.81, -3.7):
        qdd_ref = fbo.forward_dynamics(m, q, qd, u, grav)
        has_grad = m.nb >= 6                    # (the reference's floating-base rnea_grad needs NB >= 6, :1168)
        if has_grad:
            dq_ref, dqd_ref = fbo.forward_dynamics_grad(m, q, qd, u, grav)
        try:
            for opt in (RBD_MINV_PHASE_A_AUTO, RBD_MINV_PHASE_A_LANE):
                rbd._lib.set_option(RBD_OPT_MINV_PHASE_A, opt)
                for dt, tol in ((torch.float64, 1e-9), (torch.float32, 2e-4)):
                    tq, tqd, tu = (torch.tensor(x, device="cuda:0", dtype=dt) for x in (q, qd, u))
                    qdd = rbd.forward_dynamics(tq, tqd, tu, GRAVITY=grav)
                    assert rel_err_rows(qdd.double().cpu().numpy(), qdd_ref) <= tol, (opt, dt, grav)
                    if not has_grad:
                        continue
                    dq, dqd = rbd.forward_dynamics_grad(tq, tqd, tu, GRAVITY=grav)
                    assert rel_err_rows(dq.double().cpu().numpy(), dq_ref) <= tol * 10, (opt, dt, grav)
                    assert rel_err_rows(dqd.double().cpu().numpy(), dqd_ref) <= tol * 10, (opt, dt, grav)
        finally:
            rbd._lib.set_option(RBD_OPT_MINV_PHASE_A, RBD_MINV_PHASE_A_AUTO)
