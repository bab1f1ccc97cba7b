"""The one-shot methods without per-call allocations (rbdreference_amd/api.py): a repeated call with
the same signature takes its outputs from a small pool.  The contract that must survive is the reference's: every call
returns FRESH outputs (RBDReference.py:623, :785, :1345) -- a tensor the caller still holds, or any view / alias of it,
is never overwritten by a later call."""
import numpy as np
import pytest

from conftest import make_robot

pytestmark = pytest.mark.gpu


def _setup(name="atlas_like", B=512, dtype=None):
    import torch
    from rbdreference_amd import RBDReference
    assert torch.cuda.is_available()
    rbd = RBDReference(make_robot(name), build=False, generic="never")
    rng = np.random.default_rng(3)
    mk = lambda: torch.tensor(rng.uniform(-1, 1, (B, rbd.nv)), device="cuda:0", dtype=dtype or torch.float32)   # noqa: E731
    return torch, rbd, mk(), mk(), mk()


def test_pooled_outputs_are_fresh_while_held_and_reused_once_dropped():
    torch, rbd, q, qd, qdd = _setup()
    assert rbd._pool_on and rbd._lib.stable
    c1, v1, a1, f1 = rbd.rnea(q, qd, qdd)
    ref = [t.clone() for t in (c1, v1, a1, f1)]
    p1 = c1.data_ptr()
    q2 = q * 0.5
    c2, v2, a2, f2 = rbd.rnea(q2, qd, qdd)             # first set still held -> a second set
    torch.cuda.synchronize()
    assert c2.data_ptr() != p1
    assert all(torch.equal(t, r) for t, r in zip((c1, v1, a1, f1), ref))      # untouched by the second call
    assert not torch.equal(c2, c1)
    del c1, v1, a1, f1
    c3, v3, a3, f3 = rbd.rnea(q, qd, qdd)              # first set was dropped -> handed out again, same bytes as before
    torch.cuda.synchronize()
    assert c3.data_ptr() == p1
    assert all(torch.equal(t, r) for t, r in zip((c3, v3, a3, f3), ref))
    # a VIEW keeps its set busy although the tensor object itself is gone
    keep = f3[:, 2, :5]
    keep_ref = keep.clone(); p3 = f3.data_ptr()
    del c3, v3, a3, f3
    outs = [rbd.rnea(q2, qd, qdd) for _ in range(4)]
    torch.cuda.synchronize()
    assert torch.equal(keep, keep_ref)
    assert all(o[3].data_ptr() != p3 for o in outs)
    # holding only the returned TUPLE counts as holding
    res = rbd.rnea(q, qd, qdd); pr = res[0].data_ptr()
    other = rbd.rnea(q2, qd, qdd)
    assert other[0].data_ptr() != pr
    torch.cuda.synchronize()
    assert torch.equal(res[0], ref[0])


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_pooled_calls_equal_the_general_path_bit_for_bit(dtype, monkeypatch):
    import torch
    dt = getattr(torch, dtype)
    torch, rbd, q, qd, qdd = _setup("iiwa_like", 1000, dt)
    from rbdreference_amd import RBDReference
    monkeypatch.setenv("RBD_OUTPUT_POOL", "0")
    slow = RBDReference(make_robot("iiwa_like"), build=False, generic="never")
    assert not slow._pool_on
    for rep in range(3):                                # repeated: pooled sets are reused from the second round on
        for name, call in (("rnea", lambda r: r.rnea(q, qd, qdd)), ("rnea noqdd", lambda r: r.rnea(q, qd)),
                           ("grad", lambda r: (r.rnea_grad(q, qd, qdd),)), ("grad c", lambda r: r.rnea_grad(q, qd, qdd, return_c=True)),
                           ("grad noqdd damped", lambda r: (r.rnea_grad(q, qd, USE_VELOCITY_DAMPING=True),)),
                           ("both", lambda r: r.rnea_and_grad(q, qd, qdd)),
                           ("minv", lambda r: (r.minv(q),)), ("minv upper", lambda r: (r.minv(q, output_dense=False),)),
                           ("fd", lambda r: (r.forward_dynamics(q, qd, qdd),)), ("fdg", lambda r: r.forward_dynamics_grad(q, qd, qdd))):
            a = call(rbd); b = call(slow)
            torch.cuda.synchronize()
            for x, y in zip(a, b):
                assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y), (name, rep)


def test_pool_follows_options_streams_and_odd_inputs():
    torch, rbd, q, qd, qdd = _setup("atlas_like", 300)
    from rbdreference_amd import _lib as L
    m0 = rbd.minv(q).clone()
    # an option that changes the kernel (and the workspace it needs) invalidates the cached plan
    rbd._lib.set_option(L.RBD_OPT_MINV_PHASE_A, L.RBD_MINV_PHASE_A_IA8)
    try:
        m1 = rbd.minv(q)
        torch.cuda.synchronize()
        assert float((m1 - m0).abs().max()) <= 1e-5 * float(m0.abs().max())
    finally:
        rbd._lib.set_option(L.RBD_OPT_MINV_PHASE_A, L.RBD_MINV_PHASE_A_AUTO)
    # another stream: its own plan; results identical
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m2 = rbd.minv(q)
    s.synchronize()
    assert torch.equal(m2, m0)
    # inputs the fast path does not take (strided view, unbatched, numpy) still work and agree
    qs = torch.empty((300, 2 * rbd.nv), device="cuda:0")[:, ::2]
    qs.copy_(q)
    assert not qs.is_contiguous() and torch.equal(rbd.minv(qs), m0)
    assert torch.equal(rbd.minv(q[7]), m0[7])
    assert np.allclose(rbd.minv(q[:3].double().cpu().numpy()), m0[:3].double().cpu().numpy(), rtol=0, atol=2e-5 * float(m0.abs().max()))
    rbd.release_pools()
    assert torch.equal(rbd.minv(q), m0)
