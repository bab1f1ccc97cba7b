"""GPU parity tests (run with ``-m gpu`` on a real MI355X).  Every call goes through the ctypes
C-ABI into the per-robot HIP library; the CPU oracle and the golden vectors generated from the real
reference are the checkers.  Tolerances (BASELINE.json north_star): per-row,
per-tensor normwise relative error  max|x - ref| / max|ref|  <= 1e-5 in fp32, <= 1e-11 in fp64
(element-wise relative error is meaningless in fp32: input rounding alone moves small entries by
1e-3)."""
import numpy as np
import pytest

from conftest import all_golden_names, load_golden, make_robot, rel_err_rows

pytestmark = pytest.mark.gpu

TOL32 = 1e-5
TOL64 = 1e-11


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch


_RBD = {}


def rbd_for(name):
    if name not in _RBD:
        from rbdreference_amd import RBDReference
        _RBD[name] = RBDReference(make_robot(name), build=False)   # prebuilt by __graft_entry__.build()
    return _RBD[name]


def dev_tensors(dtype, *arrs):
    torch = _torch()
    return [None if a is None else torch.tensor(a, device="cuda:0", dtype=dtype) for a in arrs]


EPS32 = 2.0 ** -24          # unit round-off of float32
COND_SLACK = 8.0            # rounding errors of an O(10)-operation chain per entry, all assumed to add up


def cond_rows(H):
    """2-norm condition number of every H[b]."""
    return np.array([np.linalg.cond(h) for h in np.asarray(H, dtype=np.float64)])


def check_conditioned(name, got, want, H, slack=COND_SLACK):
    """fp32 results that solve H x = rhs (forward dynamics, aba): a backward-stable solve with
    float32-rounded data has a forward error of at most ~ cond(H) * eps per row (normwise).  The bound is
    DERIVED per row from the row's own H, not a fixed allowance; returns the worst err / bound."""
    got = got.double().cpu().numpy().reshape(len(want), -1); want = np.asarray(want).reshape(len(want), -1)
    err = np.max(np.abs(got - want), axis=1) / np.max(np.abs(want), axis=1)
    bound = slack * EPS32 * cond_rows(H)
    worst = float(np.max(err / bound))
    assert worst <= 1.0, f"{name}: error / (slack * eps32 * cond(H)) = {worst:.3f} (err {err.max():.2e}, cond {cond_rows(H).max():.1e})"
    return worst


def check(name, got, want, tol):
    got = got.double().cpu().numpy()
    assert got.shape == want.shape, f"{name}: shape {got.shape} != {want.shape}"
    assert np.all(np.isfinite(got)), f"{name}: non-finite values"
    e = rel_err_rows(got, want)
    assert e <= tol, f"{name}: worst-row normwise rel err {e:.3e} > {tol}"
    return e


@pytest.fixture(params=["float32", "float64"])
def prec(request):
    torch = _torch()
    return (torch.float32, TOL32) if request.param == "float32" else (torch.float64, TOL64)


@pytest.mark.parametrize("name", all_golden_names())
def test_library_loaded_is_the_hip_one(name):
    rbd = rbd_for(name)
    assert rbd._lib.path.endswith(".so") and "_build" in rbd._lib.path
    assert rbd._lib.info.n == rbd.n


@pytest.mark.parametrize("name", all_golden_names())
@pytest.mark.parametrize("rnea_kernel", ["batch", "groups", "auto"])
def test_rnea_vs_golden(name, prec, rnea_kernel):
    """Every rnea kernel: one lane per configuration, one wave per independent root subtree, and AUTO's choice
    (Atlas: one wave per stem / limb / root subtree; a robot without groups or limbs ignores the option)."""
    from rbdreference_amd._lib import (RBD_OPT_RNEA_KERNEL, RBD_RNEA_KERNEL_AUTO, RBD_RNEA_KERNEL_BATCH,
                                       RBD_RNEA_KERNEL_GROUPS)
    dt, tol = prec
    g = load_golden(name); rbd = rbd_for(name)
    rbd._lib.set_option(RBD_OPT_RNEA_KERNEL, {"batch": RBD_RNEA_KERNEL_BATCH, "groups": RBD_RNEA_KERNEL_GROUPS,
                                              "auto": RBD_RNEA_KERNEL_AUTO}[rnea_kernel])
    try:
        _rnea_vs_golden(rbd, g, dt, tol)
    finally:
        rbd._lib.set_option(RBD_OPT_RNEA_KERNEL, RBD_RNEA_KERNEL_AUTO)


@pytest.mark.parametrize("name", ["atlas_like", "random_limbs_n14"])
@pytest.mark.parametrize("B", [1, 63, 200])
def test_rnea_segment_waves_ragged(name, B):
    """The segment-wave kernel (AUTO on Atlas: stem + two arms + two limb-less legs; on the 14-body tree: one
    root, a two-body stem and three limbs; fp32) on ragged batches against the oracle, and against the
    one-lane-per-configuration kernel: same recursion, other summation order of the limb forces at most."""
    from rbdreference_amd._lib import RBD_OPT_RNEA_KERNEL, RBD_RNEA_KERNEL_AUTO, RBD_RNEA_KERNEL_BATCH
    from oracle import rbd_oracle as orc
    torch = _torch()
    rbd = rbd_for(name); om = orc.model_from_robot(make_robot(name))
    rng = np.random.default_rng(77 + B)
    q = rng.uniform(-np.pi, np.pi, (B, rbd.n)); qd = rng.uniform(-1, 1, (B, rbd.n)); qdd = rng.uniform(-1, 1, (B, rbd.n))
    tq, tqd, tqdd = dev_tensors(torch.float32, q, qd, qdd)
    assert "rnea_segments_kernel" in rbd._lib.kernel_name(0, 4, B)
    got = rbd.rnea(tq, tqd, tqdd)
    rbd._lib.set_option(RBD_OPT_RNEA_KERNEL, RBD_RNEA_KERNEL_BATCH)
    try:
        base = rbd.rnea(tq, tqd, tqdd)
    finally:
        rbd._lib.set_option(RBD_OPT_RNEA_KERNEL, RBD_RNEA_KERNEL_AUTO)
    want = orc.rnea(om, q, qd, qdd)
    for nm, x, y, w in zip("cvaf", got, base, want):
        check(nm, x, w, TOL32)
        check(nm + " vs batch kernel", x, y.double().cpu().numpy(), 1e-6)
    got0 = rbd.rnea(tq, tqd)
    check("c_noqdd", got0[0], orc.rnea(om, q, qd)[0], TOL32)


def _rnea_vs_golden(rbd, g, dt, tol):
    q, qd, qdd = dev_tensors(dt, g["q"], g["qd"], g["qdd"])
    c, v, a, f = rbd.rnea(q, qd, qdd)
    check("c", c, g["c"], tol); check("v", v, g["fpass_v"], tol); check("a", a, g["fpass_a"], tol)
    check("f (accumulated)", f, g["f_acc"], tol)
    c0, _, _, _ = rbd.rnea(q, qd)                       # qdd=None
    check("c_noqdd", c0, g["c_noqdd"], tol)
    c1, v1, a1, f1 = rbd.rnea(q, qd, qdd, outputs="c")
    assert v1 is None and f1 is None
    check("c (c-only kernel)", c1, g["c"], tol)


@pytest.fixture(params=["batch", "cols"])
def grad_kernel(request):
    """Every rnea_grad test runs on the robot's batch-parallel kernel AND on the column kernel (which
    AUTO would pick for these small batches): rbd_set_option, restored afterwards."""
    from rbdreference_amd._lib import (RBD_GRAD_KERNEL_AUTO, RBD_GRAD_KERNEL_BATCH, RBD_GRAD_KERNEL_COLS,
                                       RBD_OPT_GRAD_KERNEL)
    want = RBD_GRAD_KERNEL_BATCH if request.param == "batch" else RBD_GRAD_KERNEL_COLS
    touched = []

    def use(rbd):
        rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, want)
        touched.append(rbd)
        return request.param
    yield use
    for rbd in touched:
        rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_AUTO)


@pytest.mark.parametrize("name", all_golden_names())
def test_rnea_grad_vs_golden(name, prec, grad_kernel):
    from rbdreference_amd._lib import RBD_OP_RNEA_GRAD
    dt, tol = prec
    g = load_golden(name); rbd = rbd_for(name)
    which = grad_kernel(rbd)
    if which == "cols" and "cols" not in rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, 4 if dt == _torch().float32 else 8, 8):
        pytest.skip("the column kernel is not built for this robot size / precision")
    assert ("cols" in rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, 4 if dt == _torch().float32 else 8, 8)) == (which == "cols")
    q, qd, qdd = dev_tensors(dt, g["q"], g["qd"], g["qdd"])
    c, dc = rbd.rnea_grad(q, qd, qdd, return_c=True)
    check("dc_du", dc, g["dc_du"], tol); check("c", c, g["c"], tol)
    check("dc_du_damped", rbd.rnea_grad(q, qd, qdd, USE_VELOCITY_DAMPING=True), g["dc_du_damped"], tol)
    check("dc_du_noqdd", rbd.rnea_grad(q, qd), g["dc_du_noqdd"], tol)
    # rnea + rnea_grad from one call (one launch on the column kernel)
    c2, v, a, f, dc2 = rbd.rnea_and_grad(q, qd, qdd)
    check("c", c2, g["c"], tol); check("v", v, g["fpass_v"], tol); check("a", a, g["fpass_a"], tol)
    check("f (accumulated)", f, g["f_acc"], tol); check("dc_du", dc2, g["dc_du"], tol)
    c3, _, _, _, dc3 = rbd.rnea_and_grad(q, qd, USE_VELOCITY_DAMPING=True)
    check("c_noqdd", c3, g["c_noqdd"], tol)
    want = np.concatenate((g["dc_du_noqdd"][..., :rbd.n], g["dc_du_noqdd"][..., rbd.n:] +
                           (g["dc_du_damped"] - g["dc_du"])[..., rbd.n:]), axis=-1)
    check("dc_du damped, qdd=None", dc3, want, tol)


@pytest.mark.parametrize("name", all_golden_names())
def test_minv_vs_golden(name, prec):
    dt, tol = prec
    g = load_golden(name); rbd = rbd_for(name)
    (q,) = dev_tensors(dt, g["q"])
    tol_m = tol      # measured (tools/minv_error.py, profiles/r02_minv_error.json): Atlas fp32 9.7e-8 golden, 2.5e-7 random rows
    check("Minv_dense", rbd.minv(q), g["Minv_dense"], tol_m)
    up = rbd.minv(q, output_dense=False)
    want = np.triu(g["Minv_upper"])                    # reference leaves by-products below the diagonal
    check("Minv_upper", up, want, tol_m)
    assert float(up.double().cpu().numpy()[:, np.tril_indices(rbd.n, -1)[0], np.tril_indices(rbd.n, -1)[1]].__abs__().max(initial=0.0)) == 0.0


@pytest.mark.parametrize("name", ["iiwa_like", "quadruped_like", "atlas_like"])
@pytest.mark.parametrize("B", [1, 31, 33, 257, 1000])
def test_batch_sizes_vs_oracle(name, B, grad_kernel):
    """Ragged batches (not multiples of the 32/64-configuration blocks) against the oracle."""
    from oracle import rbd_oracle as orc
    torch = _torch()
    robot = make_robot(name); rbd = rbd_for(name); om = orc.model_from_robot(robot)
    grad_kernel(rbd)
    rng = np.random.default_rng(1000 + B)
    n = rbd.n
    q = rng.uniform(-np.pi, np.pi, (B, n)); qd = rng.uniform(-1, 1, (B, n)); qdd = rng.uniform(-1, 1, (B, n))
    tq, tqd, tqdd = dev_tensors(torch.float64, q, qd, qdd)
    c_ref, dc_ref = orc.rnea_grad(om, q, qd, qdd, return_c=True)
    c, dc = rbd.rnea_grad(tq, tqd, tqdd, return_c=True)
    check("dc_du", dc, dc_ref, TOL64); check("c", c, c_ref, TOL64)
    cr, vr, ar, fr = orc.rnea(om, q, qd, qdd)
    c2, v, a, f = rbd.rnea(tq, tqd, tqdd)
    check("c", c2, cr, TOL64); check("v", v, vr, TOL64); check("a", a, ar, TOL64); check("f", f, fr, TOL64)
    check("minv", rbd.minv(tq), orc.minv(om, q), TOL64)
    # fp32 on the same inputs
    sq, sqd, sqdd = dev_tensors(torch.float32, q, qd, qdd)
    check("dc_du f32", rbd.rnea_grad(sq, sqd, sqdd), dc_ref, TOL32)
    check("minv f32", rbd.minv(sq), orc.minv(om, q), TOL32)


def test_rows_are_independent_of_the_batch():
    """out[b] == the single-configuration call on row b (bit-exact: same lane arithmetic)."""
    torch = _torch()
    rbd = rbd_for("iiwa_like")
    rng = np.random.default_rng(3)
    q, qd, qdd = dev_tensors(torch.float32, rng.uniform(-3, 3, (200, 7)), rng.uniform(-1, 1, (200, 7)),
                             rng.uniform(-1, 1, (200, 7)))
    dc = rbd.rnea_grad(q, qd, qdd)
    Mi = rbd.minv(q)
    for b in (0, 1, 63, 64, 199):
        assert torch.equal(rbd.rnea_grad(q[b], qd[b], qdd[b]), dc[b])
        assert torch.equal(rbd.minv(q[b]), Mi[b])
        assert rbd.rnea_grad(q[b], qd[b], qdd[b]).shape == (7, 14)


def test_numpy_in_numpy_out_and_lists():
    rbd = rbd_for("iiwa_like")
    g = load_golden("iiwa_like")
    dc = rbd.rnea_grad(g["q"][2], list(g["qd"][2]), g["qdd"][2])
    assert isinstance(dc, np.ndarray) and dc.dtype == np.float64 and dc.shape == (7, 14)
    assert rel_err_rows(dc[None], g["dc_du"][2:3]) < TOL64
    c, v, a, f = rbd.rnea(g["q"], g["qd"], g["qdd"])
    assert isinstance(f, np.ndarray) and rel_err_rows(f, g["f_acc"]) < TOL64
    assert rel_err_rows(rbd.minv(g["q"]), g["Minv_dense"]) < TOL64


def test_argument_errors():
    torch = _torch()
    rbd = rbd_for("iiwa_like")
    q = torch.zeros((4, 7), device="cuda:0")
    with pytest.raises(ValueError):
        rbd.rnea(q, q[:, :6])
    with pytest.raises(ValueError):
        rbd.rnea(q, q[:3])
    with pytest.raises(TypeError):
        rbd.rnea(q, q.double())
    with pytest.raises(RuntimeError):
        rbd.rnea(q.cpu(), q.cpu())
    with pytest.raises(TypeError):
        rbd.rnea(q.half(), q.half())
    # C-ABI argument validation: null output pointer
    from rbdreference_amd._lib import RBD_ERR_ARG, RbdError
    with pytest.raises(RbdError) as ei:
        rbd._lib.check(rbd._lib.lib.rbd_rnea_grad_f32(q.data_ptr(), q.data_ptr(), None, -9.81, 0, 4, None, None, None))
    assert ei.value.code == RBD_ERR_ARG
    # ... and output buffers that are not 16-byte aligned (include/rbd_hip.h: the kernels store 16-byte pieces)
    buf = torch.zeros(4 * 7 * 14 + 4 * 7 * 7 + 8, device="cuda:0")
    off = buf.data_ptr() + 4                                   # 4 bytes into an allocation
    for call in (lambda: rbd._lib.lib.rbd_rnea_grad_f32(q.data_ptr(), q.data_ptr(), None, -9.81, 0, 4, None, off, None),
                 lambda: rbd._lib.lib.rbd_rnea_f32(q.data_ptr(), q.data_ptr(), None, -9.81, 4, off, None, None, None, None),
                 lambda: rbd._lib.lib.rbd_minv_f32(q.data_ptr(), 4, 1, off, None, 0, None)):
        with pytest.raises(RbdError) as ei:
            rbd._lib.check(call())
        assert ei.value.code == RBD_ERR_ARG and "aligned" in str(ei.value)
    # B = 0 is a no-op
    e = torch.zeros((0, 7), device="cuda:0")
    assert rbd.rnea_grad(e, e, e).shape == (0, 7, 14)
    assert rbd.minv(e).shape == (0, 7, 7)


def test_large_joint_angles_take_the_library_sincos():
    """fp32 sin/cos: own reduction up to |q| = 8192, sincosf beyond (rbd_spatial.h) -- both sides of the switch, with
    the oracle fed the float32-rounded angles (at |q| ~ 1e4 one fp32 ulp of q is 1e-3 rad)."""
    torch = _torch()
    from oracle import rbd_oracle as orc
    rbd = rbd_for("iiwa_like"); om = orc.model_from_robot(make_robot("iiwa_like"))
    rng = np.random.default_rng(123)
    q = rng.uniform(-20000.0, 20000.0, (512, 7)).astype(np.float32)
    q[:8] = np.array([8191.9, 8192.0, 8192.1, -8192.0, 5000.0, -7000.0, 1e6])[None, :7] * np.ones((8, 1), dtype=np.float32)
    qd = rng.uniform(-1, 1, (512, 7)).astype(np.float32); qdd = rng.uniform(-1, 1, (512, 7)).astype(np.float32)
    tq, tqd, tqdd = (torch.tensor(x, device="cuda:0") for x in (q, qd, qdd))
    c, v, a, f = rbd.rnea(tq, tqd, tqdd)
    cr, vr, ar, fr = orc.rnea(om, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))
    for nm, x, w in (("c", c, cr), ("v", v, vr), ("a", a, ar), ("f", f, fr)):
        check(nm, x, w, TOL32)
    check("dc_du", rbd.rnea_grad(tq, tqd, tqdd), orc.rnea_grad(om, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64)), TOL32)
    check("minv", rbd.minv(tq), orc.minv(om, q.astype(np.float64)), TOL32)


def test_preallocated_outputs():
    """`out=` / `workspace=`: the Python class writes into caller-owned device tensors (same kernels, no allocation)."""
    torch = _torch()
    for name in ("iiwa_like", "atlas_like"):
        rbd = rbd_for(name)
        g = load_golden(name)
        q, qd, qdd = dev_tensors(torch.float32, g["q"], g["qd"], g["qdd"])
        B, n = q.shape
        ref = rbd.rnea(q, qd, qdd)
        bufs = (torch.empty((B, n), device="cuda:0"), torch.empty((B, 6, n), device="cuda:0"),
                torch.empty((B, 6, n), device="cuda:0"), torch.empty((B, 6, n), device="cuda:0"))
        got = rbd.rnea(q, qd, qdd, out=bufs)
        assert all(x is y for x, y in zip(got, bufs)) and all(torch.equal(x, y) for x, y in zip(got, ref))
        dc = torch.empty((B, n, 2 * n), device="cuda:0"); c = torch.empty((B, n), device="cuda:0")
        assert rbd.rnea_grad(q, qd, qdd, out=dc) is dc and torch.equal(dc, rbd.rnea_grad(q, qd, qdd))
        c2, dc2 = rbd.rnea_grad(q, qd, qdd, return_c=True, out=(c, dc))
        assert c2 is c and dc2 is dc and torch.allclose(c, ref[0], rtol=1e-4, atol=1e-4)   # (the gradient kernel's own c)
        M = torch.empty((B, n, n), device="cuda:0")
        ws = torch.empty((max(rbd.minv_workspace_bytes(B), 16),), dtype=torch.uint8, device="cuda:0")
        assert rbd.minv(q, out=M, workspace=ws) is M and torch.equal(M, rbd.minv(q))
        with pytest.raises(ValueError):
            rbd.rnea(q, qd, qdd, out=(bufs[0], bufs[1], bufs[2]))              # three tensors
        with pytest.raises(ValueError):
            rbd.rnea_grad(q, qd, qdd, out=dc.double())                          # wrong dtype
        with pytest.raises(ValueError):
            rbd.minv(q, out=M[:, :, : n - 1])                                   # wrong shape
        with pytest.raises(ValueError):
            rbd.minv(g["q"], out=M)                                              # numpy input


def test_noncontiguous_inputs_are_handled():
    torch = _torch()
    rbd = rbd_for("iiwa_like")
    g = load_golden("iiwa_like")
    big = torch.tensor(np.concatenate([g["q"], g["qd"], g["qdd"]], axis=1), device="cuda:0")   # [S, 21]
    q, qd, qdd = big[:, 0:7], big[:, 7:14], big[:, 14:21]           # strided views
    assert not q.is_contiguous()
    check("dc_du", rbd.rnea_grad(q, qd, qdd), g["dc_du"], TOL64)


@pytest.mark.parametrize("name,B", [("iiwa_like", 1 << 20), ("atlas_like", 16384), ("quadruped_like", 65536)])
@pytest.mark.parametrize("precision", ["float32", "float64"])
def test_full_size_properties(name, B, precision):
    """BASELINE.json sizes in BOTH precisions (configs[4] is fp64 at B = 65 536; configs[1..3] fp32):
    size-independent properties + sampled rows against the oracle.
       * rnea is affine in qdd:  c(q,qd,qdd) - c(q,qd,0) = H(q) qdd  and  Minv H = I
       * dc_dqd of rnea_grad does not depend on qdd; dc_dq is affine in qdd
       * sampled rows equal the oracle (fp32 <= 1e-5, fp64 <= 1e-11 normwise per row)."""
    from oracle import rbd_oracle as orc
    torch = _torch()
    robot = make_robot(name); rbd = rbd_for(name); om = orc.model_from_robot(robot)
    n = rbd.n
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(B)
    f32 = precision == "float32"
    dt = torch.float32 if f32 else torch.float64
    tol = TOL32 if f32 else TOL64
    q = (torch.rand((B, n), device="cuda:0", generator=gen, dtype=dt) * 2 - 1) * np.pi
    qd = torch.rand((B, n), device="cuda:0", generator=gen, dtype=dt) * 2 - 1
    qdd = torch.rand((B, n), device="cuda:0", generator=gen, dtype=dt) * 2 - 1
    c, dc = rbd.rnea_grad(q, qd, qdd, return_c=True)
    c0, dc0 = rbd.rnea_grad(q, qd, torch.zeros_like(qdd), return_c=True)
    Mi = rbd.minv(q)
    assert torch.isfinite(dc).all() and torch.isfinite(Mi).all()
    # Minv (c - c0) = Minv H qdd = qdd
    back = torch.einsum("bij,bj->bi", Mi.double(), (c - c0).double())
    scale = qdd.abs().max().item()
    # fp32: c carries ~1e-6 relative rounding on gravity-sized torques; (c - c0) = H qdd is O(1), so the
    # difference has ~5e-5 relative error, amplified by cond(Minv) ~ 1e2 (n = 7, 12) .. 1e3 (n = 30);
    # fp64: the same chain starts from 1e-15
    lim = (3e-3 if n < 30 else 1e-2) if f32 else 1e-9
    assert (back - qdd.double()).abs().max().item() < lim * scale
    # dc_dqd independent of qdd
    d = (dc[:, :, n:] - dc0[:, :, n:]).abs().amax(dim=(1, 2)) / dc0[:, :, n:].abs().amax(dim=(1, 2)).clamp_min(1e-30)
    assert d.max().item() < tol
    # symmetric dense Minv
    assert torch.equal(Mi, Mi.transpose(1, 2))
    # sampled rows vs oracle
    idx = np.unique(np.concatenate([[0, 1, 31, 32, 63, 64, B - 1], np.random.default_rng(0).integers(0, B, 249)]))
    tidx = torch.tensor(idx, device="cuda:0")
    qs, qds, qdds = (x[tidx].double().cpu().numpy() for x in (q, qd, qdd))
    c_ref, dc_ref = orc.rnea_grad(om, qs, qds, qdds, return_c=True)
    check("dc_du sample", dc[tidx], dc_ref, tol)
    check("c sample", c[tidx], c_ref, tol)
    check("minv sample", Mi[tidx], orc.minv(om, qs), tol)
    cr, vr, ar, fr = orc.rnea(om, qs, qds, qdds)
    c2, v, a, f = rbd.rnea(q, qd, qdd)
    check("rnea c sample", c2[tidx], cr, tol); check("rnea v sample", v[tidx], vr, tol)
    check("rnea a sample", a[tidx], ar, tol); check("rnea f sample", f[tidx], fr, tol)


# ---- next row: forward dynamics compositions ------------------------------------
@pytest.mark.parametrize("name", all_golden_names())
def test_forward_dynamics_vs_golden(name, prec):
    """forward_dynamics / forward_dynamics_grad (RBDReference.py:1371-1384) against the reference's
    own outputs.  fp32 tolerance is looser than for rnea_grad: the result is multiplied by Minv,
    whose condition number (1e2..1e4 for these robots) amplifies input rounding."""
    dt, tol = prec
    torch = _torch()
    g = load_golden(name); rbd = rbd_for(name)
    q, qd, u = dev_tensors(dt, g["q"], g["qd"], g["qdd"])       # gen_golden used u = qdd
    if dt == torch.float32:
        # the bound follows from cond(H) of each sampled row (the golden file carries H = crba(q))
        check_conditioned("fd_qdd", rbd.forward_dynamics(q, qd, u), g["fd_qdd"], g["H"])
        a, b = rbd.forward_dynamics_grad(q, qd, u)
        check_conditioned("fd_dq", a.contiguous(), g["fd_dq"], g["H"])
        check_conditioned("fd_dqd", b.contiguous(), g["fd_dqd"], g["H"])
        return
    tol_fd = 1e-9
    check("fd_qdd", rbd.forward_dynamics(q, qd, u), g["fd_qdd"], tol_fd)
    a, b = rbd.forward_dynamics_grad(q, qd, u)
    check("fd_dq", a.contiguous(), g["fd_dq"], tol_fd)
    check("fd_dqd", b.contiguous(), g["fd_dqd"], tol_fd)


@pytest.mark.parametrize("name,B", [("iiwa_like", (1 << 20) + 37), ("random_chain_n7", 65536 + 5), ("quadruped_like", 65536), ("atlas_like", 16384)])
@pytest.mark.parametrize("precision", ["float32", "float64"])
def test_forward_dynamics_grad_full_size(name, B, precision):
    """forward_dynamics_grad at BASELINE sizes (plus a ragged tail for the chains: the two-launch path of rbd_fd_chain.h keeps
    Minv in whole 64-row tiles), both precisions.  Size-independent property: it IS -minv(q) rnea_grad(q, qd, qdd) with
    qdd = forward_dynamics(q, qd, u) (RBDReference.py:1376-1384) -- recomputed from the three separate entry points on
    the device for EVERY row -- and sampled rows (tile boundaries, the last rows) equal the oracle."""
    from oracle import rbd_oracle as orc
    torch = _torch()
    rbd = rbd_for(name); om = orc.model_from_robot(make_robot(name)); n = rbd.n
    f32 = precision == "float32"
    dt = torch.float32 if f32 else torch.float64
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(B)
    q = (torch.rand((B, n), device="cuda:0", generator=gen, dtype=dt) * 2 - 1) * np.pi
    qd = torch.rand((B, n), device="cuda:0", generator=gen, dtype=dt) * 2 - 1
    u = (torch.rand((B, n), device="cuda:0", generator=gen, dtype=dt) * 2 - 1) * 5
    a, b = rbd.forward_dynamics_grad(q, qd, u)
    assert a.shape == (B, n, n) and b.shape == (B, n, n) and torch.isfinite(a).all() and torch.isfinite(b).all()
    qdd = rbd.forward_dynamics(q, qd, u)
    dc = rbd.rnea_grad(q, qd, qdd)
    Mi = rbd.minv(q)
    want = -torch.einsum("bij,bjk->bik", Mi.double(), dc.double())
    got = torch.cat([a, b], dim=2).double()
    err = ((got - want).abs().amax(dim=(1, 2)) / want.abs().amax(dim=(1, 2)).clamp_min(1e-30)).max().item()
    # fp32: qdd of the two routes differs by rounding amplified by cond(H) (<= ~1e4 here), and dc_dq is affine in qdd
    assert err < (2e-2 if f32 else 1e-8), err
    idx = np.unique(np.concatenate([[0, 1, 63, 64, 65, B - 65, B - 64, B - 2, B - 1], np.random.default_rng(1).integers(0, B, 120)]))
    tidx = torch.tensor(idx, device="cuda:0")
    qs, qds, us = (x[tidx].double().cpu().numpy() for x in (q, qd, u))
    r1, r2 = orc.forward_dynamics_grad(om, qs, qds, us)
    if f32:
        H = orc.crba(om, qs)
        check_conditioned("fd_dq sample", a[tidx].contiguous(), r1, H, slack=2 * COND_SLACK)
        check_conditioned("fd_dqd sample", b[tidx].contiguous(), r2, H, slack=2 * COND_SLACK)
    else:
        check("fd_dq sample", a[tidx].contiguous(), r1, 1e-8); check("fd_dqd sample", b[tidx].contiguous(), r2, 1e-8)


def test_forward_dynamics_round_trip_full_size():
    """rnea(q, qd, forward_dynamics(q, qd, u)) == u at B = 1M (fp32), ragged B, numpy path."""
    torch = _torch()
    rbd = rbd_for("iiwa_like")
    for B in (1 << 20, 1000):
        gen = torch.Generator(device="cuda:0"); gen.manual_seed(B)
        q = (torch.rand((B, 7), device="cuda:0", generator=gen) * 2 - 1) * np.pi
        qd = torch.rand((B, 7), device="cuda:0", generator=gen) * 2 - 1
        u = (torch.rand((B, 7), device="cuda:0", generator=gen) * 2 - 1) * 10
        qdd = rbd.forward_dynamics(q, qd, u)
        c, _, _, _ = rbd.rnea(q, qd, qdd, outputs="c")
        err = ((c - u).abs().amax(dim=1) / u.abs().amax(dim=1)).max().item()
        assert err < 2e-3, err
    g = load_golden("iiwa_like")
    out = rbd.forward_dynamics(g["q"][1], g["qd"][1], g["qdd"][1])
    assert isinstance(out, np.ndarray) and rel_err_rows(out[None], g["fd_qdd"][1:2]) < 1e-9
    a, b = rbd.forward_dynamics_grad(g["q"][1], g["qd"][1], g["qdd"][1])
    assert a.shape == (7, 7) and rel_err_rows(np.ascontiguousarray(b)[None], g["fd_dqd"][1:2]) < 1e-9


@pytest.mark.parametrize("name", all_golden_names())
def test_rnea_per_pass_surface(name, prec):
    """rnea_fpass / rnea_bpass (README.md:19's accelerator-testing surface) against the reference's
    per-pass outputs; rnea_bpass mutates its f argument in place and returns it (RBDReference.py:619)."""
    dt, tol = prec
    g = load_golden(name); rbd = rbd_for(name)
    q, qd, qdd = dev_tensors(dt, g["q"], g["qd"], g["qdd"])
    v, a, f = rbd.rnea_fpass(q, qd, qdd)
    check("fpass v", v, g["fpass_v"], tol); check("fpass a", a, g["fpass_a"], tol)
    check("fpass f (local)", f, g["fpass_f"], tol)
    (f_in,) = dev_tensors(dt, g["fpass_f"])
    c, f_ret = rbd.rnea_bpass(q, f_in)
    assert f_ret is f_in
    check("bpass c", c, g["c"], tol); check("bpass f (accumulated, in place)", f_in, g["f_acc"], tol)
    c1, f1 = rbd.rnea_bpass(g["q"][0], g["fpass_f"][0])                 # numpy, unbatched
    assert isinstance(f1, np.ndarray) and f1.shape == (6, rbd.n)
    assert rel_err_rows(f1[None], g["f_acc"][:1]) < TOL64 and rel_err_rows(c1[None], g["c"][:1]) < TOL64


@pytest.mark.parametrize("name", all_golden_names())
def test_crba_vs_golden(name, prec):
    """crba (RBDReference.py:1091-1124) against the reference's H, and Minv H = I on the device."""
    dt, tol = prec
    torch = _torch()
    g = load_golden(name); rbd = rbd_for(name)
    (q,) = dev_tensors(dt, g["q"])
    H = rbd.crba(q)
    check("H", H, g["H"], tol)
    assert torch.equal(H, H.transpose(1, 2))
    eye = torch.einsum("bij,bjk->bik", rbd.minv(q).double(), H.double())
    err = (eye - torch.eye(rbd.n, device="cuda:0", dtype=torch.float64)).abs().max().item()
    assert err < (1e-9 if dt == torch.float64 else 5e-3)
    rng = np.random.default_rng(9)
    qb = torch.tensor(rng.uniform(-3, 3, (130, rbd.n)), device="cuda:0", dtype=dt)      # ragged batch
    from oracle import rbd_oracle as orc
    check("H ragged", rbd.crba(qb), orc.crba(orc.model_from_robot(make_robot(name)), qb.double().cpu().numpy()), tol)


@pytest.mark.parametrize("name", ["iiwa_like", "quadruped_like", "random_prismatic_n6"])
@pytest.mark.parametrize("grav", [0.0, -3.71, 9.81])
def test_gravity_argument(name, grav):
    """GRAVITY is a runtime argument of every entry point (reference default -9.81, :559,:623,:1345)."""
    from oracle import rbd_oracle as orc
    torch = _torch()
    robot = make_robot(name); rbd = rbd_for(name); om = orc.model_from_robot(robot)
    rng = np.random.default_rng(77)
    n = rbd.n
    q = rng.uniform(-np.pi, np.pi, (70, n)); qd = rng.uniform(-1, 1, (70, n)); qdd = rng.uniform(-1, 1, (70, n))
    tq, tqd, tqdd = dev_tensors(torch.float64, q, qd, qdd)
    c_ref, dc_ref = orc.rnea_grad(om, q, qd, qdd, GRAVITY=grav, return_c=True)
    c, dc = rbd.rnea_grad(tq, tqd, tqdd, GRAVITY=grav, return_c=True)
    check("dc_du", dc, dc_ref, TOL64)
    if np.abs(c_ref).max() > 0:
        check("c", c, c_ref, TOL64)
    cr, vr, ar, fr = orc.rnea(om, q, qd, qdd, GRAVITY=grav)
    c2, v, a, f = rbd.rnea(tq, tqd, tqdd, GRAVITY=grav)
    check("a", a, ar, TOL64); check("f", f, fr, TOL64)
    if grav == 0.0:      # zero gravity, zero motion => exactly zero torques (SURVEY.md §4 invariant)
        z = torch.zeros_like(tq)
        c0, dc0 = rbd.rnea_grad(tq, z, z, GRAVITY=0.0, return_c=True)
        assert float(c0.abs().max()) == 0.0


def test_large_angles_and_nan_propagation():
    """q far outside [-pi, pi] takes sincosf's slow range reduction; NaN inputs give NaN outputs
    for that row only."""
    from oracle import rbd_oracle as orc
    torch = _torch()
    robot = make_robot("iiwa_like"); rbd = rbd_for("iiwa_like"); om = orc.model_from_robot(robot)
    rng = np.random.default_rng(5)
    q = rng.uniform(-1e4, 1e4, (64, 7)); qd = rng.uniform(-1, 1, (64, 7)); qdd = rng.uniform(-1, 1, (64, 7))
    tq, tqd, tqdd = dev_tensors(torch.float64, q, qd, qdd)
    check("dc_du large q", rbd.rnea_grad(tq, tqd, tqdd), orc.rnea_grad(om, q, qd, qdd), 1e-9)
    check("minv large q", rbd.minv(tq), orc.minv(om, q), 1e-9)
    sq = tq.float(); sq[3, 2] = float("nan")
    dc = rbd.rnea_grad(sq, tqd.float(), tqdd.float())
    assert torch.isnan(dc[3]).any() and torch.isfinite(dc[:3]).all() and torch.isfinite(dc[4:]).all()


# ---- per-pass gradient / Minv surface (README.md:19 of the reference) -----------------------------------
@pytest.mark.parametrize("name", all_golden_names())
def test_rnea_grad_passes_vs_golden(name, prec):
    """rnea_grad_fpass_dq/dqd and rnea_grad_bpass_dq/dqd fed exactly as RBDReference.rnea_grad feeds
    them (RBDReference.py:1353-1365), against the reference's own per-pass outputs."""
    dt, tol = prec
    g = load_golden(name); rbd = rbd_for(name)
    q, qd, v, a, f = dev_tensors(dt, g["q"], g["qd"], g["fpass_v"], g["fpass_a"], g["f_acc"])
    dv, da, df = rbd.rnea_grad_fpass_dq(q, qd, v, a)
    check("dv_dq", dv, g["dq_dv"], tol); check("da_dq", da, g["dq_da"], tol); check("df_dq", df, g["dq_df"], tol)
    dv2, da2, df2 = rbd.rnea_grad_fpass_dqd(q, qd, v)
    check("dv_dqd", dv2, g["dqd_dv"], tol); check("da_dqd", da2, g["dqd_da"], tol)
    check("df_dqd", df2, g["dqd_df"], tol)
    # backward passes on the reference's df (the kernels' own df differs by rounding only)
    (gdf, gdf2) = dev_tensors(dt, g["dq_df"], g["dqd_df"])
    n = rbd.n
    check("dc_dq", rbd.rnea_grad_bpass_dq(q, f, gdf), g["dc_dq"], tol)
    check("dc_dqd", rbd.rnea_grad_bpass_dqd(q, gdf2.clone()), g["dc_dqd"], tol)
    check("dc_dqd damped", rbd.rnea_grad_bpass_dqd(q, gdf2.clone(), USE_VELOCITY_DAMPING=True),
          g["dc_dqd_damped"], tol)
    # chained: our fpass -> our bpass reproduces rnea_grad's halves
    check("dc_dq (chained)", rbd.rnea_grad_bpass_dq(q, f, df), g["dc_du"][:, :, :n], tol)
    check("dc_dqd (chained)", rbd.rnea_grad_bpass_dqd(q, df2), g["dc_du"][:, :, n:], tol)


@pytest.mark.parametrize("name", ["iiwa_like", "random_tree_n9", "random_prismatic_n6", "atlas_like"])
def test_rnea_grad_bpass_mutates_df_like_the_reference(name):
    """The backward passes accumulate df child -> parent in place (RBDReference.py:1291-1294, :1331);
    checked against the oracle on DENSE random df (the passes accept any input)."""
    from oracle import rbd_oracle as orc
    torch = _torch()
    robot = make_robot(name); rbd = rbd_for(name); om = orc.model_from_robot(robot)
    n = rbd.n; B = 37
    rng = np.random.default_rng(77)
    q = rng.uniform(-np.pi, np.pi, (B, n)); f = rng.normal(size=(B, 6, n))
    df = rng.normal(size=(B, 6, n, n))
    df_ref = df.copy(); dc_ref = orc.rnea_grad_bpass_dq(om, q, f, df_ref)
    tq, tf, tdf = dev_tensors(torch.float64, q, f, df)
    dc = rbd.rnea_grad_bpass_dq(tq, tf, tdf)
    check("dc_dq", dc, dc_ref, TOL64); check("df_dq after", tdf, df_ref, TOL64)
    df2_ref = df.copy(); dc2_ref = orc.rnea_grad_bpass_dqd(om, q, df2_ref, True)
    (tdf2,) = dev_tensors(torch.float64, df)
    check("dc_dqd", rbd.rnea_grad_bpass_dqd(tq, tdf2, True), dc2_ref, TOL64)
    check("df_dqd after", tdf2, df2_ref, TOL64)
    # numpy in -> the caller's ndarray is updated too, unbatched shapes like the reference's
    dfn = df[0].copy()
    dcn = rbd.rnea_grad_bpass_dq(q[0], f[0], dfn)
    assert dcn.shape == (n, n) and np.abs(dfn - df_ref[0]).max() <= 1e-9 * np.abs(df_ref[0]).max()


@pytest.mark.parametrize("name", all_golden_names())
def test_minv_passes_vs_golden(name, prec):
    """minv_bpass -> (Minv, F, U, Dinv) and minv_fpass against the reference's per-pass outputs
    (RBDReference.py:630-783), including the by-products minv_fpass leaves below the diagonal."""
    dt, tol = prec
    g = load_golden(name); rbd = rbd_for(name)
    tol_m = tol      # measured (tools/minv_error.py, profiles/r02_minv_error.json): Atlas fp32 9.7e-8 golden, 2.5e-7 random rows
    (q,) = dev_tensors(dt, g["q"])
    Mb, F, U, D = rbd.minv_bpass(q)
    check("minv_bpass Minv", Mb, g["mb_Minv"], tol_m); check("minv_bpass F", F, g["mb_F"], tol_m)
    check("minv_bpass U", U, g["mb_U"], tol_m); check("minv_bpass Dinv (= D)", D, g["mb_Dinv"], tol_m)
    M = rbd.minv_fpass(q, Mb, F, U, D)
    assert M is Mb                                                  # updated in place, like the reference
    check("minv_fpass Minv (whole matrix, junk included)", M, g["Minv_upper"], tol_m)
    # the reference's own bpass outputs as input
    gM, gF, gU, gD = dev_tensors(dt, g["mb_Minv"], g["mb_F"], g["mb_U"], g["mb_Dinv"])
    check("minv_fpass on golden inputs", rbd.minv_fpass(q, gM, gF, gU, gD), g["Minv_upper"], tol_m)


def test_per_pass_shapes_unbatched_numpy():
    """(n,) numpy inputs give the reference's unbatched shapes (RBDReference.py:1132-1134, :656-660)."""
    g = load_golden("iiwa_like"); rbd = rbd_for("iiwa_like"); n = rbd.n
    q, qd = g["q"][0], g["qd"][0]
    dv, da, df = rbd.rnea_grad_fpass_dq(q, qd, g["fpass_v"][0], g["fpass_a"][0])
    assert dv.shape == da.shape == df.shape == (6, n, n) and isinstance(df, np.ndarray)
    assert np.abs(df - g["dq_df"][0]).max() <= 1e-11 * np.abs(g["dq_df"][0]).max()
    Mb, F, U, D = rbd.minv_bpass(q)
    assert Mb.shape == (n, n) and F.shape == (n, 6, n) and U.shape == (n, 6) and D.shape == (n,)
    M = rbd.minv_fpass(q, Mb, F, U, D)
    assert np.abs(M - g["Minv_upper"][0]).max() <= 1e-11 * np.abs(g["Minv_upper"][0]).max()
    with pytest.raises(ValueError):
        rbd.rnea_grad_fpass_dqd(q, qd, np.zeros((6, n + 1)))
    with pytest.raises(ValueError):
        rbd.minv_fpass(q, Mb, F, U[:, :5], D)


@pytest.mark.parametrize("name", all_golden_names())
def test_aba_vs_golden(name, prec):
    """aba (RBDReference.py:940-1024) against the reference's aba and forward_dynamics outputs."""
    dt, tol = prec
    g = load_golden(name); rbd = rbd_for(name)
    q, qd, tau = dev_tensors(dt, g["q"], g["qd"], g["qdd"])
    qdd = rbd.aba(q, qd, tau)
    if dt == _torch().float32:      # bound derived from cond(H) of each row, as for forward_dynamics
        check_conditioned("aba_qdd", qdd, g["aba_qdd"], g["H"]); check_conditioned("aba vs forward_dynamics", qdd, g["fd_qdd"], g["H"])
    else:
        check("aba_qdd", qdd, g["aba_qdd"], 1e-9); check("aba vs forward_dynamics", qdd, g["fd_qdd"], 1e-9)
    assert rbd.aba(q[0], qd[0], tau[0]).shape == (rbd.n,)
    assert rbd.aba(g["q"][0], g["qd"][0], g["qdd"][0], f_ext=[]).shape == (rbd.n,)


@pytest.mark.parametrize("name,B", [("iiwa_like", 100003), ("atlas_like", 16384), ("quadruped_like", 65536)])
def test_aba_round_trip_full_size(name, B):
    """rnea(q, qd, aba(q, qd, tau)) == tau at BASELINE sizes, ragged last block included (fp64)."""
    torch = _torch()
    rbd = rbd_for(name); n = rbd.n
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    q = (torch.rand((B, n), device="cuda:0", dtype=torch.float64, generator=gen) * 2 - 1) * np.pi
    qd = torch.rand((B, n), device="cuda:0", dtype=torch.float64, generator=gen) * 2 - 1
    tau = torch.rand((B, n), device="cuda:0", dtype=torch.float64, generator=gen) * 2 - 1
    for grav in (-9.81, 0.0):
        qdd = rbd.aba(q, qd, tau, GRAVITY=grav)
        c, _, _, _ = rbd.rnea(q, qd, qdd, GRAVITY=grav, outputs="c")
        scale = max(1.0, float(qdd.abs().max()))
        assert float((c - tau).abs().max()) <= 1e-9 * scale
    # fp32 against fp64 on 512 sampled rows: bound derived from cond(H) of each row (H from the fp64 crba)
    idx = torch.tensor(np.random.default_rng(1).integers(0, B, 512), device="cuda:0")
    qs, qds, taus = q[idx], qd[idx], tau[idx]
    H = rbd.crba(qs).cpu().numpy()
    check_conditioned("aba fp32 vs fp64", rbd.aba(qs.float(), qds.float(), taus.float()), rbd.aba(qs, qds, taus).cpu().numpy(), H)


def test_tree_gradient_kernel_forced_on_every_robot():
    """rbd_set_option(RBD_OPT_GRAD_KERNEL, TREE) routes rnea_grad of every eligible robot (all
    revolute, rigid inertias) through the chain-by-chain world-frame kernel (rbd_idsva_tree.h), which is
    the default only for Atlas-size fp32 trees; rbd_kernel_name reports what runs."""
    torch = _torch()
    from rbdreference_amd._lib import RBD_GRAD_KERNEL_AUTO, RBD_GRAD_KERNEL_TREE, RBD_OP_RNEA_GRAD, RBD_OPT_GRAD_KERNEL
    worst = {}
    ran_tree = 0
    for name in all_golden_names():
        g = load_golden(name); rbd = rbd_for(name)
        rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_TREE)
        try:
            assert rbd._lib.get_option(RBD_OPT_GRAD_KERNEL) == RBD_GRAD_KERNEL_TREE
            for dt, tol in ((torch.float32, TOL32), (torch.float64, TOL64)):
                ran_tree += "tree" in rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, 4 if dt == torch.float32 else 8, 8)
                q, qd, qdd = dev_tensors(dt, g["q"], g["qd"], g["qdd"])
                c, dc = rbd.rnea_grad(q, qd, qdd, return_c=True)
                e = max(check("dc_du", dc, g["dc_du"], tol), check("c", c, g["c"], tol))
                want = np.concatenate((g["dc_du_noqdd"][..., :rbd.n], g["dc_du_noqdd"][..., rbd.n:] +
                                       (g["dc_du_damped"] - g["dc_du"])[..., rbd.n:]), axis=-1)
                e2 = check("dc_du damped, qdd=None", rbd.rnea_grad(q, qd, USE_VELOCITY_DAMPING=True), want, tol)
                worst[(name, str(dt))] = max(e, e2)
            # ragged batch, fp32, against the single-configuration call
            rng = np.random.default_rng(5); n = rbd.n
            q, qd, qdd = dev_tensors(torch.float32, *(rng.uniform(-2, 2, (130, n)) for _ in range(3)))
            dc = rbd.rnea_grad(q, qd, qdd)
            for b in (0, 63, 64, 129):
                assert torch.equal(rbd.rnea_grad(q[b], qd[b], qdd[b]), dc[b]), (name, b)
        finally:
            rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_AUTO)
    assert ran_tree >= 8, ran_tree      # iiwa, quadruped, chain, tree in both precisions, Atlas in fp32, ...


@pytest.mark.parametrize("name,forced", [("atlas_like", False), ("random_limbs_n14", False), ("iiwa_like", True),
                                         ("random_chain_n7", True), ("random_twochains_n18", False)])
def test_fp64_workspace_tree_kernel(name, forced):
    """fp64 rnea_grad of trees too big for registers + LDS (rbd_idsva_tree_ws.h): path vectors, pending entries and
    parked composites in a library-owned global workspace.  Default for the robots whose fp32 default is the tree
    kernel (Atlas, the 14-body limbs robot), on request for the chains.  Checked: the golden vectors of the real
    reference (qdd given / None, damping), a ragged batch row by row, a batch that walks the workspace in several
    chunks (rows on both sides of every chunk boundary are bit-identical to a small call and agree with the oracle),
    and two streams at once (one workspace per stream).  Multi-root robots on the single-wave layout (one block per
    root: the two nine-body chains, whose default kernel this is) are the ADVICE r3 case: the blocks of different
    roots run concurrently and each needs its own workspace region."""
    from oracle import rbd_oracle as orc
    torch = _torch()
    from rbdreference_amd._lib import RBD_GRAD_KERNEL_AUTO, RBD_GRAD_KERNEL_TREE, RBD_OP_RNEA_GRAD, RBD_OPT_GRAD_KERNEL
    g = load_golden(name); rbd = rbd_for(name); n = rbd.n
    if forced:
        rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_TREE)
    try:
        assert "rnea_grad_tree_ws_kernel<double" in rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, 8, 16384)
        q, qd, qdd = dev_tensors(torch.float64, g["q"], g["qd"], g["qdd"])
        c, dc = rbd.rnea_grad(q, qd, qdd, return_c=True)
        check("dc_du", dc, g["dc_du"], TOL64); check("c", c, g["c"], TOL64)
        check("dc_du qdd=None", rbd.rnea_grad(q, qd), g["dc_du_noqdd"], TOL64)
        check("dc_du damped", rbd.rnea_grad(q, qd, qdd, USE_VELOCITY_DAMPING=True), g["dc_du_damped"], TOL64)
        rng = np.random.default_rng(11)
        B = 3 * 16384 + 71          # at least three chunks on a 256-CU device (one 64-row block per CU and launch)
        Q, QD, QDD = (rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n)))
        tq, tqd, tqdd = dev_tensors(torch.float64, Q, QD, QDD)
        cb, dcb = rbd.rnea_grad(tq, tqd, tqdd, return_c=True)
        om = orc.model_from_robot(make_robot(name))
        for lo in (0, 16384 - 70, 2 * 16384 - 70, 3 * 16384 - 70, B - 130):
            hi = min(lo + 130, B)
            c1, dc1 = rbd.rnea_grad(tq[lo:hi], tqd[lo:hi], tqdd[lo:hi], return_c=True)       # ragged: 130 or fewer rows
            assert torch.equal(dc1, dcb[lo:hi]) and torch.equal(c1, cb[lo:hi]), (name, lo)
            c_ref, dc_ref = orc.rnea_grad(om, Q[lo:hi], QD[lo:hi], QDD[lo:hi], return_c=True)
            check("dc_du rows", dcb[lo:hi], dc_ref, TOL64); check("c rows", cb[lo:hi], c_ref, TOL64)
        for b in (0, 63, 64, 129):
            assert torch.equal(rbd.rnea_grad(tq[b], tqd[b], tqdd[b]), dcb[b]), (name, b)
        # two streams, interleaved launches: every stream owns its workspace
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        outs = []
        for rep in range(3):
            with torch.cuda.stream(s1):
                outs.append(rbd.rnea_grad(tq[:20000], tqd[:20000], tqdd[:20000]))
            with torch.cuda.stream(s2):
                outs.append(rbd.rnea_grad(tq[20000:40000], tqd[20000:40000], tqdd[20000:40000]))
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            lo = 0 if k % 2 == 0 else 20000
            assert torch.equal(o, dcb[lo:lo + 20000]), (name, "streams", k)
    finally:
        rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_AUTO)


@pytest.mark.parametrize("name", ["atlas_like", "random_limbs_n14", "random_tree_n9", "random_forest_n8"])
def test_minv_both_phase_a_kernels(name):
    """Robots too big for the one-lane minv kernel: phase A with one lane per configuration (the default from
    B = 262 144) and with eight lanes per configuration feed the column kernel; robots whose big groups have
    limbs (Atlas, the 14-body tree) also have the one-launch kernel (their default; a request
    for it on another robot falls back to the two launches).  All are held to the golden vectors and to the oracle
    on a ragged batch, forward dynamics included."""
    torch = _torch()
    from rbdreference_amd._lib import (RBD_MINV_PHASE_A_AUTO, RBD_MINV_PHASE_A_FUSED, RBD_MINV_PHASE_A_IA8,
                                       RBD_MINV_PHASE_A_LANE, RBD_OPT_MINV_PHASE_A)
    from oracle import rbd_oracle as orc
    g = load_golden(name); rbd = rbd_for(name)
    if int(rbd._lib.lib.rbd_minv_workspace_bytes(64, 4)) == 0:
        pytest.skip("robot uses the fused one-lane minv kernel (no phase A / B)")
    om = orc.model_from_robot(make_robot(name))
    rng = np.random.default_rng(9)
    qr = rng.uniform(-np.pi, np.pi, (777, rbd.n))
    ref = orc.minv(om, qr)
    qdr = rng.uniform(-1, 1, (41, rbd.n)); ur = rng.uniform(-1, 1, (41, rbd.n))
    fdg_ref = orc.forward_dynamics_grad(om, qr[:41], qdr, ur)
    if name in ("atlas_like", "random_limbs_n14"):
        assert "minv_fused_kernel" in rbd._lib.kernel_name(2, 4, 777)
    try:
        for mode in (RBD_MINV_PHASE_A_LANE, RBD_MINV_PHASE_A_IA8, RBD_MINV_PHASE_A_FUSED):
            rbd._lib.set_option(RBD_OPT_MINV_PHASE_A, mode)
            for dt, tol in ((torch.float32, TOL32), (torch.float64, TOL64)):
                (q,) = dev_tensors(dt, g["q"])
                check("Minv_dense", rbd.minv(q), g["Minv_dense"], tol)
                check("Minv_upper", rbd.minv(q, output_dense=False), np.triu(g["Minv_upper"]), tol)
                (q2,) = dev_tensors(dt, qr)
                check("Minv ragged", rbd.minv(q2), ref, tol)
            # the kernels' qdd = Minv (u - c) epilogue, through forward_dynamics_grad (rnea -> minv + product -> rnea_grad)
            q2, qd2, u2 = dev_tensors(torch.float64, qr[:41], qdr, ur)
            got_dq, got_dqd = rbd.forward_dynamics_grad(q2, qd2, u2)
            check("forward_dynamics_grad dq ragged", got_dq, fdg_ref[0], 1e-8)
            check("forward_dynamics_grad dqd ragged", got_dqd, fdg_ref[1], 1e-8)
    finally:
        rbd._lib.set_option(RBD_OPT_MINV_PHASE_A, RBD_MINV_PHASE_A_AUTO)


def test_sharded_rbd_over_nccl():
    """ShardedRBD (rbdreference_amd/dist.py) on the real backend: `world` fresh processes (started before
    they touch the GPU), one per device -- world = 1 on a one-GPU box, up to 4 where more devices exist --
    nccl (= RCCL) process group, HIP kernels per rank; the gathered outputs must equal the unsharded call
    bit for bit, scatter_rows must deliver the shard, and the scattered path must agree."""
    import os
    import socket
    import subprocess
    import sys
    torch = _torch()
    world = max(1, min(4, torch.cuda.device_count()))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "dist_nccl_worker.py")], cwd=root, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    import time
    deadline = time.time() + 600                  # ONE deadline for all ranks; nobody is left running on a failure
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=max(1.0, deadline - time.time()))[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.communicate()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{o[-3000:]}"
        assert "sharded ok" in o, o[-2000:]


def test_shard_equals_rows_of_the_unsharded_call_bit_for_bit():
    """a shard must equal the same rows of the unsharded call EXACTLY, whatever the number of ranks.
    Kernel selection depends on the batch size (4 096-row shards of a 65 536-row batch would run the small-batch
    column kernel, the unsharded call the batch-parallel one): `shard_of(global_rows)` -- what ShardedRBD wraps every
    shard in -- pins the selection to the global size.  Checked here on one GPU by slicing."""
    torch = _torch()
    from rbdreference_amd import RBDReference
    from rbdreference_amd.dist import ShardedRBD, shard_bounds
    rbd = RBDReference(make_robot("iiwa_like"), build=False)
    B = 65536
    rng = np.random.default_rng(12)
    q, qd, qdd = (torch.tensor(rng.uniform(-1, 1, (B, rbd.n)), device="cuda:0", dtype=torch.float32) for _ in range(3))
    full = rbd.rnea_grad(q, qd, qdd)
    full_minv = rbd.minv(q)
    assert rbd._lib.kernel_name(1, 4, B) != rbd._lib.kernel_name(1, 4, 4096)     # the hazard is real for this robot
    for world in (8, 16):
        for rank in (0, world // 2, world - 1):
            a, b = shard_bounds(B, world, rank)
            with rbd.shard_of(B):
                assert rbd._lib.kernel_name(1, 4, b - a) == rbd._lib.kernel_name(1, 4, B)
                part = rbd.rnea_grad(q[a:b], qd[a:b], qdd[a:b])
                part_minv = rbd.minv(q[a:b])
            assert torch.equal(part, full[a:b]), (world, rank)
            assert torch.equal(part_minv, full_minv[a:b]), (world, rank)
    # ShardedRBD itself (world = 1 here) goes through the same context and leaves the option as it found it
    sh = ShardedRBD(rbd)
    assert torch.equal(sh.rnea_grad(q[:4096], qd[:4096], qdd[:4096]), rbd.rnea_grad(q[:4096], qd[:4096], qdd[:4096]))
    from rbdreference_amd._lib import RBD_OPT_SELECT_BATCH
    assert rbd._lib.get_option(RBD_OPT_SELECT_BATCH) == 0


@pytest.mark.parametrize("B", [4096, 8192, 8193])
def test_iiwa_auto_path_at_configs1_and_at_the_selection_boundary(B):
    """BASELINE configs[1] is iiwa at exactly B = 4 096: the AUTO path (the column kernel there) for rnea, rnea_grad
    and the one-launch rnea_and_grad against the oracle on 256 sampled rows; and both sides of the batch size at
    which AUTO switches kernels (8 192 | 8 193 rows for n = 7)."""
    torch = _torch()
    from rbdreference_amd import RBDReference
    from oracle import rbd_oracle as fxo
    rbd = RBDReference(make_robot("iiwa_like"), build=False)
    m = fxo.model_from_robot(make_robot("iiwa_like"))
    rng = np.random.default_rng(B)
    q = rng.uniform(-np.pi, np.pi, (B, rbd.n)); qd = rng.uniform(-1, 1, (B, rbd.n)); qdd = rng.uniform(-1, 1, (B, rbd.n))
    tq, tqd, tqdd = (torch.tensor(x, device="cuda:0", dtype=torch.float32) for x in (q, qd, qdd))
    name = rbd._lib.kernel_name(1, 4, B)
    assert name.startswith("rnea_grad_cols_kernel") == (B <= 8192), name
    rows = rng.choice(B, 256, replace=False)
    c, v, a, f = rbd.rnea(tq, tqd, tqdd)
    dc = rbd.rnea_grad(tq, tqd, tqdd)
    c2, v2, a2, f2, dc2 = rbd.rnea_and_grad(tq, tqd, tqdd)
    cr, vr, ar, fr = fxo.rnea(m, q[rows], qd[rows], qdd[rows])
    dcr = fxo.rnea_grad(m, q[rows], qd[rows], qdd[rows])
    for nm, got, want in (("c", c, cr), ("v", v, vr), ("a", a, ar), ("f", f, fr), ("dc_du", dc, dcr),
                          ("c (one call)", c2, cr), ("v (one call)", v2, vr), ("a (one call)", a2, ar), ("f (one call)", f2, fr),
                          ("dc_du (one call)", dc2, dcr)):
        e = rel_err_rows(got[rows].double().cpu().numpy(), want)
        assert e <= 1e-5, (nm, e)


def test_bound_launches_equal_the_methods_and_replay_in_a_graph():
    """RBDReference.bind: one ctypes call per launch on pre-allocated buffers -- bit-identical to the methods, reads
    the caller's buffers in place (new inputs, same launch), capturable in a torch.cuda.graph; the per-call cost is
    printed next to the method's (configs[1]'s shape: the 7-DoF arm at B = 4096)."""
    import time
    torch = _torch()
    rbd = rbd_for("iiwa_like")
    B = 4096
    rng = np.random.default_rng(12)
    q, qd, qdd = dev_tensors(torch.float32, rng.uniform(-3, 3, (B, 7)), rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7)))
    for op, ref in (("rnea", lambda: rbd.rnea(q, qd, qdd)), ("rnea_grad", lambda: rbd.rnea_grad(q, qd, qdd, return_c=True)),
                    ("rnea_and_grad", lambda: rbd.rnea_and_grad(q, qd, qdd)), ("minv", lambda: (rbd.minv(q),))):
        launch = rbd.bind(op, q, qd, qdd) if op != "minv" else rbd.bind(op, q)
        outs = launch()
        for a, b in zip(outs, ref()):
            assert torch.equal(a, b), op
    launch = rbd.bind("rnea_and_grad", q, qd, qdd)
    q2 = torch.tensor(rng.uniform(-3, 3, (B, 7)), device="cuda:0", dtype=torch.float32)
    want = rbd.rnea_and_grad(q2, qd, qdd)
    q.copy_(q2)                                           # the launch reads the caller's buffer in place
    for a, b in zip(launch(), want):
        assert torch.equal(a, b)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launch(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            launch()
    launch.outputs[4].zero_(); g.replay(); torch.cuda.synchronize()
    assert torch.equal(launch.outputs[4], want[4])
    with pytest.raises(ValueError):
        rbd.bind("rnea", q[::2], qd[::2], qdd[::2])         # not contiguous: the launch could not read it in place
    for nm, fn in (("method", lambda: rbd.rnea_and_grad(q, qd, qdd)), ("bound", launch)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300):
            fn()
        torch.cuda.synchronize()
        print(f"\nrnea_and_grad B={B}: {nm} {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call", end="")
