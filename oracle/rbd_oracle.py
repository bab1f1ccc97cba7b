"""CPU oracle for the rnea / rnea_grad / minv path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch numpy restatement (float64, vectorised over a leading batch axis B) of the
reference's per-pass algorithms in ``/root/reference/RBDReference.py``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package ``rbdreference_amd`` never does (its HIP path fails loudly instead of falling back).

Pinning: ``oracle/gen_golden.py`` runs the *real* reference (imported from /root/reference in the
build container) on the package's ``Robot`` objects and writes per-pass inputs/outputs to
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` holds this file to those vectors at
rtol 1e-12.  The reference itself owns no tests or golden vectors.

Every function cites the reference lines it restates.  Shapes: the reference's ``(6, NB)`` becomes
``[B, 6, NB]``, ``(6, n, NB)`` becomes ``[B, 6, n, NB]``, ``(n, n)`` becomes ``[B, n, n]``; a 1-D
``q`` is treated as B = 1 and the batch axis is dropped again on return.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

__all__ = ["OracleModel", "model_from_robot", "Xmats", "rnea_fpass", "rnea_bpass", "rnea",
           "rnea_grad_fpass_dq", "rnea_grad_fpass_dqd", "rnea_grad_bpass_dq",
           "rnea_grad_bpass_dqd", "rnea_grad", "minv_bpass", "minv_fpass", "minv", "crba",
           "forward_dynamics", "forward_dynamics_grad"]


@dataclass
class OracleModel:
    n: int
    parent: List[int]
    subtree: List[List[int]]
    S: np.ndarray          # [n, 6]
    I: np.ndarray          # [n, 6, 6]
    damping: np.ndarray    # [n]
    prismatic: np.ndarray  # [n] bool
    X0: np.ndarray         # [n, 6, 6]   X(q) = X0 + Xs sin q + Xc cos q   (revolute)
    Xs: np.ndarray         # [n, 6, 6]        = X0 + Xs q                  (prismatic)
    Xc: np.ndarray         # [n, 6, 6]


def model_from_robot(robot) -> OracleModel:
    """Black-box sampling of the getters the reference's hot path reads:
    ``X_i(q)`` is affine in (sin q, cos q) for a revolute joint and in q for a prismatic one, so
    three (two) evaluations of ``get_Xmat_Func_by_id(i)`` recover it exactly; a fourth evaluation
    checks the fit."""
    if getattr(robot, "floating_base", False):
        raise NotImplementedError("floating-base robots are out of scope (SURVEY.md §2 row 16)")
    n = int(robot.get_num_bodies())
    assert int(robot.get_num_vel()) == n
    parent = [int(robot.get_parent_id(i)) for i in range(n)]
    subtree = [[int(j) for j in robot.get_subtree_by_id(i)] for i in range(n)]
    S = np.stack([np.asarray(robot.get_S_by_id(i), dtype=np.float64).reshape(6) for i in range(n)])
    I = np.stack([np.asarray(robot.get_Imat_by_id(i), dtype=np.float64).reshape(6, 6)
                  for i in range(n)])
    damping = np.array([float(robot.get_damping_by_id(i)) for i in range(n)])
    X0 = np.zeros((n, 6, 6)); Xs = np.zeros((n, 6, 6)); Xc = np.zeros((n, 6, 6))
    prismatic = np.zeros(n, dtype=bool)
    for i in range(n):
        assert int(robot.get_joint_index_q(i)) == i and int(robot.get_joint_index_v(i)) == i
        f = robot.get_Xmat_Func_by_id(i)
        prismatic[i] = bool(np.any(S[i, 3:] != 0) and not np.any(S[i, :3] != 0))
        A = np.asarray(f(0.0), dtype=np.float64)
        if prismatic[i]:
            X0[i] = A
            Xs[i] = np.asarray(f(1.0), dtype=np.float64) - A
            Xfit = X0[i] + Xs[i] * 0.37
        else:
            Bm = np.asarray(f(math.pi), dtype=np.float64)
            Cm = np.asarray(f(math.pi / 2), dtype=np.float64)
            X0[i] = 0.5 * (A + Bm)
            Xc[i] = 0.5 * (A - Bm)
            Xs[i] = Cm - X0[i]
            Xfit = X0[i] + Xs[i] * math.sin(0.37) + Xc[i] * math.cos(0.37)
        if not np.allclose(Xfit, np.asarray(f(0.37), dtype=np.float64), rtol=0, atol=1e-12):
            raise ValueError(f"joint {i}: Xmat(q) is not affine in (sin q, cos q) / q")
    return OracleModel(n, parent, subtree, S, I, damping, prismatic, X0, Xs, Xc)


# ---------------------------------------------------------------------------------------------
# helpers (RBDReference.py:9-182)
# ---------------------------------------------------------------------------------------------
def _as_batch(x, n) -> Tuple[np.ndarray, bool]:
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        assert x.shape[0] == n
        return x[None, :], True
    assert x.ndim == 2 and x.shape[1] == n
    return x, False


def Xmats(m: OracleModel, q: np.ndarray) -> np.ndarray:
    """``get_Xmat_Func_by_id(i)(q_i)`` for every body: [B, n, 6, 6]."""
    s = np.where(m.prismatic[None, :], q, np.sin(q))[:, :, None, None]
    c = np.where(m.prismatic[None, :], 0.0, np.cos(q))[:, :, None, None]
    return m.X0[None] + m.Xs[None] * s + m.Xc[None] * c


def _crm(v: np.ndarray) -> np.ndarray:
    """cross_operator, RBDReference.py:9-21; v [..., 6] -> [..., 6, 6]."""
    z = np.zeros_like(v[..., 0])
    r0 = np.stack([z, -v[..., 2], v[..., 1], z, z, z], -1)
    r1 = np.stack([v[..., 2], z, -v[..., 0], z, z, z], -1)
    r2 = np.stack([-v[..., 1], v[..., 0], z, z, z, z], -1)
    r3 = np.stack([z, -v[..., 5], v[..., 4], z, -v[..., 2], v[..., 1]], -1)
    r4 = np.stack([v[..., 5], z, -v[..., 3], v[..., 2], z, -v[..., 0]], -1)
    r5 = np.stack([-v[..., 4], v[..., 3], z, -v[..., 1], v[..., 0], z], -1)
    return np.stack([r0, r1, r2, r3, r4, r5], -2)


def _mxS(S: np.ndarray, vec: np.ndarray, alpha=1.0) -> np.ndarray:
    """_mxS / mxS, RBDReference.py:56-75: alpha * crm(vec) @ S; S [6], vec [..., 6]."""
    out = np.einsum("...ij,j->...i", _crm(vec), S)
    return out * alpha


def _fxv(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """fxv, RBDReference.py:149-164: crf(a) @ b with crf = -crm^T (``:23-25``)."""
    return -np.einsum("...ji,...j->...i", _crm(a), b)


def _vxIv(v: np.ndarray, I: np.ndarray) -> np.ndarray:
    """vxIv, RBDReference.py:170-182: crf(v) @ (I v)."""
    return _fxv(v, np.einsum("ij,...j->...i", I, v))


# ---------------------------------------------------------------------------------------------
# RNEA (RBDReference.py:559-628)
# ---------------------------------------------------------------------------------------------
def rnea_fpass(m: OracleModel, q, qd, qdd=None, GRAVITY=-9.81):
    """RBDReference.py:559-598."""
    q, un = _as_batch(q, m.n)
    qd, _ = _as_batch(qd, m.n)
    if qdd is not None:
        qdd, _ = _as_batch(qdd, m.n)
    B, n = q.shape
    X = Xmats(m, q)
    v = np.zeros((B, 6, n)); a = np.zeros((B, 6, n)); f = np.zeros((B, 6, n))
    gravity_vec = np.zeros(6)
    gravity_vec[5] = -GRAVITY                                             # :565-566
    for i in range(n):
        p = m.parent[i]
        S = m.S[i]
        if p == -1:
            a[:, :, i] = np.einsum("bij,j->bi", X[:, i], gravity_vec)      # :578
        else:
            v[:, :, i] = np.einsum("bij,bj->bi", X[:, i], v[:, :, p])      # :580
            a[:, :, i] = np.einsum("bij,bj->bi", X[:, i], a[:, :, p])      # :581
        vJ = S[None, :] * qd[:, i:i + 1]                                  # :586
        v[:, :, i] += vJ                                                  # :587
        a[:, :, i] += np.einsum("bij,bj->bi", _crm(v[:, :, i]), vJ)       # :588  mxS(vJ, v_i)
        if qdd is not None:
            a[:, :, i] += S[None, :] * qdd[:, i:i + 1]                    # :589-593
        I = m.I[i]
        f[:, :, i] = np.einsum("ij,bj->bi", I, a[:, :, i]) + _vxIv(v[:, :, i], I)   # :595-596
    if un:
        return v[0], a[0], f[0]
    return v, a, f


def rnea_bpass(m: OracleModel, q, f):
    """RBDReference.py:600-621.  Like the reference it accumulates into ``f`` IN PLACE and returns
    the same array (``:619``)."""
    q, un = _as_batch(q, m.n)
    fb = f[None] if un else f
    B, n = q.shape
    X = Xmats(m, q)
    c = np.zeros((B, n))
    for i in range(n - 1, -1, -1):
        p = m.parent[i]
        c[:, i] = np.einsum("j,bj->b", m.S[i], fb[:, :, i])               # :612
        if p != -1:
            fb[:, :, p] = fb[:, :, p] + np.einsum("bji,bj->bi", X[:, i], fb[:, :, i])  # :618-619
    if un:
        return c[0], f
    return c, f


def rnea(m: OracleModel, q, qd, qdd=None, GRAVITY=-9.81, f_ext=None):
    """RBDReference.py:623-628 (``f_ext`` accepted and ignored, as there)."""
    v, a, f = rnea_fpass(m, q, qd, qdd, GRAVITY)
    c, f = rnea_bpass(m, q, f)
    return c, v, a, f


# ---------------------------------------------------------------------------------------------
# RNEA gradient (RBDReference.py:1127-1368)
# ---------------------------------------------------------------------------------------------
def _grad_df(m: OracleModel, i: int, v_i, dv_i, da_i):
    """df = I da + crf(dv) (I v) + crf(v) (I dv) on every column (``:1179-1185`` / ``:1247-1252``).
    v_i [B,6]; dv_i, da_i [B,6,n] -> [B,6,n]."""
    I = m.I[i]
    df = np.einsum("ij,bjc->bic", I, da_i)
    Iv = np.einsum("ij,bj->bi", I, v_i)
    dvT = np.swapaxes(dv_i, 1, 2)                                         # [B,n,6]
    t1 = _fxv(dvT, Iv[:, None, :])                                        # crf(dv_c) Iv
    t2 = _fxv(v_i[:, None, :], np.einsum("ij,bcj->bci", I, dvT))          # crf(v) I dv_c
    return df + np.swapaxes(t1 + t2, 1, 2)


def rnea_grad_fpass_dq(m: OracleModel, q, qd, v, a, GRAVITY=-9.81):
    """RBDReference.py:1127-1187 (fixed-base branch, ``idx = ind``)."""
    q, un = _as_batch(q, m.n)
    qd, _ = _as_batch(qd, m.n)
    if un:
        v = v[None]; a = a[None]
    B, n = q.shape
    X = Xmats(m, q)
    dv = np.zeros((B, 6, n, n)); da = np.zeros((B, 6, n, n)); df = np.zeros((B, 6, n, n))
    gravity_vec = np.zeros(6)
    gravity_vec[5] = -GRAVITY
    for i in range(n):
        p = m.parent[i]
        S = m.S[i]
        Xi = X[:, i]
        if p != -1:
            dv[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, dv[:, :, :, p])              # :1158
            dv[:, :, i, i] += _mxS(S, np.einsum("bij,bj->bi", Xi, v[:, :, p]))           # :1159
            da[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, da[:, :, :, p])              # :1163
        # :1164-1170  da[:,c,i] += qd_i * crm(dv[:,c,i]) S   for every column c
        dvT = np.swapaxes(dv[:, :, :, i], 1, 2)                                         # [B,n,6]
        da[:, :, :, i] += np.swapaxes(_mxS(S, dvT), 1, 2) * qd[:, None, i:i + 1]
        if p != -1:
            da[:, :, i, i] += _mxS(S, np.einsum("bij,bj->bi", Xi, a[:, :, p]))           # :1173
        else:
            da[:, :, i, i] += _mxS(S, np.einsum("bij,j->bi", Xi, gravity_vec))           # :1175
        df[:, :, :, i] = _grad_df(m, i, v[:, :, i], dv[:, :, :, i], da[:, :, :, i])      # :1177-1185
    if un:
        return dv[0], da[0], df[0]
    return dv, da, df


def rnea_grad_fpass_dqd(m: OracleModel, q, qd, v):
    """RBDReference.py:1189-1255 (fixed-base branch)."""
    q, un = _as_batch(q, m.n)
    qd, _ = _as_batch(qd, m.n)
    if un:
        v = v[None]
    B, n = q.shape
    X = Xmats(m, q)
    dv = np.zeros((B, 6, n, n)); da = np.zeros((B, 6, n, n)); df = np.zeros((B, 6, n, n))
    for i in range(n):
        p = m.parent[i]
        S = m.S[i]
        Xi = X[:, i]
        if p != -1:
            dv[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, dv[:, :, :, p])              # :1230
        dv[:, :, i, i] += S[None, :]                                                    # :1231
        if p != -1:
            da[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, da[:, :, :, p])              # :1234
        dvT = np.swapaxes(dv[:, :, :, i], 1, 2)
        da[:, :, :, i] += np.swapaxes(_mxS(S, dvT), 1, 2) * qd[:, None, i:i + 1]         # :1235-1240
        da[:, :, i, i] += _mxS(S, v[:, :, i])                                           # :1243
        df[:, :, :, i] = _grad_df(m, i, v[:, :, i], dv[:, :, :, i], da[:, :, :, i])      # :1245-1252
    if un:
        return dv[0], da[0], df[0]
    return dv, da, df


def rnea_grad_bpass_dq(m: OracleModel, q, f, df_dq):
    """RBDReference.py:1257-1297.  ``f`` is the ACCUMULATED RNEA force (``:1353,:1362``).  Mutates
    ``df_dq`` in place like the reference (``:1291``).  ``fxS(S, f) = -crm(f) S`` is kept literally
    (``:166-168,:1292``); it equals the true ``crf(S) f`` only for revolute S."""
    q, un = _as_batch(q, m.n)
    if un:
        f = f[None]; df_dq = df_dq[None]
    B, n = q.shape
    X = Xmats(m, q)
    dc = np.zeros((B, n, n))
    for i in range(n - 1, -1, -1):
        p = m.parent[i]
        S = m.S[i]
        dc[:, i, :] = np.einsum("j,bjc->bc", S, df_dq[:, :, :, i])                      # :1284
        if p != -1:
            Xi = X[:, i]
            df_dq[:, :, :, p] += np.einsum("bji,bjc->bic", Xi, df_dq[:, :, :, i])       # :1291
            delta = np.einsum("bji,bj->bi", Xi, -_mxS(S, f[:, :, i]))                   # :1292
            df_dq[:, :, i, p] += delta                                                  # :1293-1294
    return dc[0] if un else dc


def rnea_grad_bpass_dqd(m: OracleModel, q, df_dqd, USE_VELOCITY_DAMPING=False):
    """RBDReference.py:1299-1343."""
    q, un = _as_batch(q, m.n)
    if un:
        df_dqd = df_dqd[None]
    B, n = q.shape
    X = Xmats(m, q)
    dc = np.zeros((B, n, n))
    for i in range(n - 1, -1, -1):
        p = m.parent[i]
        dc[:, i, :] = np.einsum("j,bjc->bc", m.S[i], df_dqd[:, :, :, i])                # :1325
        if p != -1:
            df_dqd[:, :, :, p] += np.einsum("bji,bjc->bic", X[:, i], df_dqd[:, :, :, i])  # :1331
    if USE_VELOCITY_DAMPING:
        for i in range(n):
            dc[:, i, i] += m.damping[i]                                                 # :1336-1341
    return dc[0] if un else dc


def rnea_grad(m: OracleModel, q, qd, qdd=None, GRAVITY=-9.81, USE_VELOCITY_DAMPING=False,
              return_c=False):
    """RBDReference.py:1345-1368 -> dc_du = hstack(dc_dq, dc_dqd), [B, n, 2n]."""
    c, v, a, f = rnea(m, q, qd, qdd, GRAVITY)
    _, _, df_dq = rnea_grad_fpass_dq(m, q, qd, v, a, GRAVITY)
    _, _, df_dqd = rnea_grad_fpass_dqd(m, q, qd, v)
    dc_dq = rnea_grad_bpass_dq(m, q, f, df_dq)
    dc_dqd = rnea_grad_bpass_dqd(m, q, df_dqd, USE_VELOCITY_DAMPING)
    dc_du = np.concatenate((dc_dq, dc_dqd), axis=-1)                                    # :1367
    return (c, dc_du) if return_c else dc_du


# ---------------------------------------------------------------------------------------------
# Minv (RBDReference.py:630-806), fixed-base branches
# ---------------------------------------------------------------------------------------------
def minv_bpass(m: OracleModel, q):
    """RBDReference.py:630-735.  Returns (Minv, F, U, Dinv) with ``Dinv`` holding D, not 1/D, as the
    reference does (``:698``)."""
    q, un = _as_batch(q, m.n)
    B, n = q.shape
    X = Xmats(m, q)
    Minv = np.zeros((B, n, n)); F = np.zeros((B, n, 6, n)); U = np.zeros((B, n, 6))
    Dinv = np.zeros((B, n))
    IA = np.broadcast_to(m.I[None], (B, n, 6, 6)).copy()                                # :662
    for i in range(n - 1, -1, -1):
        st = m.subtree[i]
        p = m.parent[i]
        S = m.S[i]
        U[:, i] = np.einsum("bij,j->bi", IA[:, i], S)                                   # :697
        Dinv[:, i] = np.einsum("j,bj->b", S, U[:, i])                                   # :698
        Minv[:, i, i] = 1.0 / Dinv[:, i]                                                # :700
        Minv[:, i, st] -= (1.0 / Dinv[:, i])[:, None] * np.einsum("j,bjs->bs", S, F[:, i][:, :, st])  # :702-708
        if p != -1:
            Xi = X[:, i]
            for s in st:                                                                # :720-726
                F[:, i, :, s] += U[:, i] * Minv[:, i, s:s + 1]
                F[:, p, :, s] += np.einsum("bji,bj->bi", Xi, F[:, i, :, s])
            Ia = IA[:, i] - np.einsum("bi,bj->bij", U[:, i], U[:, i] / Dinv[:, i:i + 1])  # :728-731
            IA[:, p] += np.einsum("bji,bjk,bkl->bil", Xi, Ia, Xi)                       # :732-733
    if un:
        return Minv[0], F[0], U[0], Dinv[0]
    return Minv, F, U, Dinv


def minv_fpass(m: OracleModel, q, Minv, F, U, Dinv):
    """RBDReference.py:737-783.  Updates ``Minv`` and ``F`` in place; whole rows are updated
    (``:771``), so the strict lower triangle ends up holding by-products, exactly as there."""
    q, un = _as_batch(q, m.n)
    if un:
        Minv = Minv[None]; F = F[None]; U = U[None]; Dinv = Dinv[None]
    B, n = q.shape
    X = Xmats(m, q)
    for i in range(n):
        p = m.parent[i]
        S = m.S[i]
        Xi = X[:, i]
        if p != -1:
            UX = np.einsum("bj,bjk->bk", U[:, i], Xi)
            Minv[:, i, :] -= (1.0 / Dinv[:, i])[:, None] * np.einsum("bk,bkc->bc", UX, F[:, p])  # :771-773
            F[:, i] = np.einsum("bij,bjc->bic", Xi, F[:, p]) + S[None, :, None] * Minv[:, i, None, :]  # :774-776
        else:
            F[:, i] = S[None, :, None] * Minv[:, i, None, :]                            # :781
    return Minv[0] if un else Minv


def minv(m: OracleModel, q, output_dense=True):
    """RBDReference.py:785-806."""
    Minv, F, U, Dinv = minv_bpass(m, q)
    Minv = minv_fpass(m, q, Minv, F, U, Dinv)
    if output_dense:                                                                    # :799-804
        iu = np.triu_indices(m.n, 1)
        Minv[..., iu[1], iu[0]] = Minv[..., iu[0], iu[1]]
    return Minv


# ---------------------------------------------------------------------------------------------
# Witness / next-row helpers
# ---------------------------------------------------------------------------------------------
def crba(m: OracleModel, q):
    """RBDReference.py:1091-1124 (fixed-base CRBA); used only as the ``minv @ H = I`` witness."""
    q, un = _as_batch(q, m.n)
    B, n = q.shape
    X = Xmats(m, q)
    IC = np.broadcast_to(m.I[None], (B, n, 6, 6)).copy()
    for i in range(n - 1, -1, -1):
        p = m.parent[i]
        if p != -1:
            IC[:, p] = IC[:, p] + np.einsum("bji,bjk,bkl->bil", X[:, i], IC[:, i], X[:, i])  # :1101
    H = np.zeros((B, n, n))
    for i in range(n):
        fh = np.einsum("bij,j->bi", IC[:, i], m.S[i])                                   # :1109
        H[:, i, i] = np.einsum("j,bj->b", m.S[i], fh)                                   # :1110
        j = i
        while m.parent[j] > -1:                                                         # :1113
            fh = np.einsum("bji,bj->bi", X[:, j], fh)                                   # :1116
            j = m.parent[j]
            H[:, i, j] = np.einsum("j,bj->b", m.S[j], fh)                               # :1121
            H[:, j, i] = H[:, i, j]
    return H[0] if un else H


def forward_dynamics(m: OracleModel, q, qd, u):
    """RBDReference.py:1371-1374."""
    c, _, _, _ = rnea(m, q, qd)
    Mi = minv(m, q)
    u = np.asarray(u, dtype=np.float64)
    return np.einsum("...ij,...j->...i", Mi, u - c)


def forward_dynamics_grad(m: OracleModel, q, qd, u):
    """RBDReference.py:1376-1384."""
    qdd = forward_dynamics(m, q, qd, u)
    dc_du = rnea_grad(m, q, qd, qdd)
    n = m.n
    Mi = minv(m, q)
    return (np.einsum("...ij,...jk->...ik", -Mi, dc_du[..., :n]),
            np.einsum("...ij,...jk->...ik", -Mi, dc_du[..., n:]))


def aba(m: OracleModel, q, qd, tau, f_ext=None, GRAVITY=-9.81):
    """RBDReference.py:940-1024 (fixed-base branch of ``aba``) -> qdd.

    The reference writes ``pA[:,ind] = np.matmul(temp, v[:,ind])[0]`` (``:984``): with an ``np.matrix``
    inertia (what URDFParser hands out) that is the whole 6-vector ``crf(v) I v``; with a plain
    ndarray inertia the ``[0]`` picks one scalar and the result is no longer the ABA.  The
    restatement follows the np.matrix reading, which is the one that agrees with
    ``forward_dynamics`` (probe: 1e-14)."""
    q, un = _as_batch(q, m.n)
    qd, _ = _as_batch(qd, m.n)
    tau, _ = _as_batch(tau, m.n)
    B, n = q.shape
    X = Xmats(m, q)
    v = np.zeros((B, n, 6)); c = np.zeros((B, n, 6)); a = np.zeros((B, n, 6))
    pA = np.zeros((B, n, 6)); IA = np.broadcast_to(m.I[None], (B, n, 6, 6)).copy()      # :972
    U = np.zeros((B, n, 6)); d = np.zeros((B, n)); u = np.zeros((B, n)); qdd = np.zeros((B, n))
    a0 = np.zeros(6); a0[5] = -GRAVITY                                                  # :954-955
    for i in range(n):
        p = m.parent[i]
        vJ = m.S[i][None, :] * qd[:, i:i + 1]
        if p == -1:
            v[:, i] = vJ                                                                # :963
        else:
            v[:, i] = np.einsum("bij,bj->bi", X[:, i], v[:, p]) + vJ                    # :966-967
            c[:, i] = _mxS(m.S[i], v[:, i], qd[:, i:i + 1])                             # :968
        pA[:, i] = _vxIv(v[:, i], m.I[i])                                               # :974-984
    for i in range(n - 1, -1, -1):
        p = m.parent[i]
        S = m.S[i]
        U[:, i] = np.einsum("bij,j->bi", IA[:, i], S)                                   # :990
        d[:, i] = U[:, i] @ S                                                           # :991
        u[:, i] = tau[:, i] - pA[:, i] @ S                                              # :992
        if p != -1:
            Ia = IA[:, i] - np.einsum("bi,bj->bij", U[:, i], U[:, i]) / d[:, i, None, None]   # :996-997
            pa = pA[:, i] + np.einsum("bij,bj->bi", Ia, c[:, i]) + U[:, i] * (u[:, i] / d[:, i])[:, None]  # :999
            Xi = X[:, i]
            IA[:, p] += np.einsum("bji,bjk,bkl->bil", Xi, Ia, Xi)                       # :1001-1004
            pA[:, p] += np.einsum("bji,bj->bi", Xi, pa)                                 # :1006-1007
    for i in range(n):
        p = m.parent[i]
        Xi = X[:, i]
        if p == -1:
            a[:, i] = np.einsum("bij,j->bi", Xi, a0) + c[:, i]                          # :1015
        else:
            a[:, i] = np.einsum("bij,bj->bi", Xi, a[:, p]) + c[:, i]                    # :1017
        qdd[:, i] = (u[:, i] - np.einsum("bj,bj->b", U[:, i], a[:, i])) / d[:, i]       # :1020-1021
        a[:, i] += m.S[i][None, :] * qdd[:, i:i + 1]                                    # :1022
    return qdd[0] if un else qdd
