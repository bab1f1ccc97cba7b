"""CPU oracle for the FLOATING-BASE branches of rnea / minv / forward_dynamics -- TEST INFRASTRUCTURE.

Same role and rules as ``oracle/rbd_oracle.py`` (only tests, ``smoke()`` and the bench's CPU-baseline
leg may import it).  A from-scratch numpy restatement (float64, batch axis first) of what
``/root/reference/RBDReference.py`` does when ``robot.floating_base`` is set: body 0 is attached by a
6-DoF joint with ``S = eye(6)`` and owns indices 0..5, body ``i >= 1`` owns index ``i + 5``
(``:585-593, :652-691, :761-779``).

Pinning: ``oracle/gen_golden.py`` runs the real reference on ``FloatingBaseRobot`` objects
(``tests/golden/golden_fb_*.npz``); ``tests/test_oracle_golden.py`` holds this file to them at 1e-12.
Only ``rnea``, ``minv`` and ``forward_dynamics`` exist for floating bases: the reference itself raises in
``rnea_grad`` (``:1168``: indexes body ``ii`` of a ``(6, n, NB)`` array with ``ii`` up to 5), ``crba``
(``:1063``) and ``aba`` (``:900``) on such a robot -- recorded in DESIGN.md, nothing to restate.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from . import rbd_oracle as fx

__all__ = ["FbModel", "model_from_robot", "Xmats", "rnea_fpass", "rnea_bpass", "rnea", "minv_bpass",
           "minv_fpass", "minv", "forward_dynamics", "joint_space_inertia", "rnea_grad",
           "rnea_grad_fpass_dq", "rnea_grad_fpass_dqd", "rnea_grad_bpass_dq", "rnea_grad_bpass_dqd",
           "forward_dynamics_grad"]


@dataclass
class FbModel:
    nb: int                 # bodies
    n: int                  # velocities = nb + 5
    parent: List[int]
    subtree: List[List[int]]
    S: np.ndarray           # [nb, 6]   (row 0 unused: the base has S = eye(6))
    I: np.ndarray           # [nb, 6, 6]
    joints: fx.OracleModel  # X(q) fits of bodies 1.. (index 0 is a dummy identity joint)
    base_X: object          # callable q6 -> 6x6 (get_Xmat_Func_by_id(0))
    damping: np.ndarray = None   # [nb] get_damping_by_id


def model_from_robot(robot) -> FbModel:
    assert getattr(robot, "floating_base", False)
    nb = int(robot.get_num_bodies())
    assert int(robot.get_num_vel()) == nb + 5
    assert np.array_equal(np.asarray(robot.get_S_by_id(0)), np.eye(6))
    assert list(robot.get_joint_index_q(0)) == [0, 1, 2, 3, 4, 5] and list(robot.get_joint_index_v(0)) == [0, 1, 2, 3, 4, 5]

    class _View:       # bodies 1.. seen as a fixed-base robot so that the X(q) sampling of rbd_oracle applies
        floating_base = False

        def __getattr__(self, k):
            return getattr(robot, k)

        def get_num_vel(self):
            return nb

        def get_S_by_id(self, i):
            return np.array([0, 0, 1.0, 0, 0, 0]) if i == 0 else robot.get_S_by_id(i)

        def get_Xmat_Func_by_id(self, i):
            if i == 0:
                from math import cos, sin
                return lambda q: np.block([[np.array([[cos(q), sin(q), 0], [-sin(q), cos(q), 0], [0, 0, 1.0]]), np.zeros((3, 3))],
                                           [np.zeros((3, 3)), np.array([[cos(q), sin(q), 0], [-sin(q), cos(q), 0], [0, 0, 1.0]])]])
            return robot.get_Xmat_Func_by_id(i)

        def get_joint_index_q(self, i):
            return i

        get_joint_index_v = get_joint_index_q
    for i in range(1, nb):
        assert int(robot.get_joint_index_q(i)) == i + 5 and int(robot.get_joint_index_v(i)) == i + 5
    jm = fx.model_from_robot(_View())
    damp = np.array([float(robot.get_damping_by_id(i)) for i in range(nb)])
    return FbModel(nb, nb + 5, jm.parent, jm.subtree, jm.S, jm.I, jm, robot.get_Xmat_Func_by_id(0), damp)


def Xmats(m: FbModel, q: np.ndarray) -> np.ndarray:
    """X_i(q) for every body: [B, nb, 6, 6]; body 0 from the base's six coordinates."""
    B = q.shape[0]
    qj = np.concatenate([np.zeros((B, 1)), q[:, 6:]], axis=1)
    X = fx.Xmats(m.joints, qj)
    X[:, 0] = np.stack([np.asarray(m.base_X(q[b, :6]), dtype=np.float64) for b in range(B)])
    return X


def _batch(x, n):
    x = np.asarray(x, dtype=np.float64)
    return (x[None], True) if x.ndim == 1 else (x, False)


def rnea_fpass(m: FbModel, q, qd, qdd=None, GRAVITY=-9.81):
    """RBDReference.py:559-598 with the floating-base lines :585, :591 (vJ = S qd[0:6] = the base twist)."""
    q, unb = _batch(q, m.n); qd, _ = _batch(qd, m.n)
    qdd = None if qdd is None else _batch(qdd, m.n)[0]
    B = q.shape[0]
    X = Xmats(m, q)
    v = np.zeros((B, 6, m.nb)); a = np.zeros((B, 6, m.nb)); f = np.zeros((B, 6, m.nb))
    g = np.zeros(6); g[5] = -GRAVITY
    for i in range(m.nb):
        p = m.parent[i]
        if p == -1:
            a[:, :, i] = X[:, i] @ g                                                    # :578
        else:
            v[:, :, i] = np.einsum("bij,bj->bi", X[:, i], v[:, :, p])                 # :580
            a[:, :, i] = np.einsum("bij,bj->bi", X[:, i], a[:, :, p])                 # :581
        if i == 0:
            vJ = qd[:, 0:6]                                                             # :585  (S = eye(6))
        else:
            vJ = m.S[i][None, :] * qd[:, i + 5][:, None]                                # :586
        v[:, :, i] += vJ                                                                # :587
        a[:, :, i] += np.einsum("bij,bj->bi", fx._crm(v[:, :, i]), vJ)                 # :588  mxS(vJ, v) = crm(v) vJ
        if qdd is not None:
            a[:, :, i] += qdd[:, 0:6] if i == 0 else m.S[i][None, :] * qdd[:, i + 5][:, None]   # :589-593
        Iv = np.einsum("ij,bj->bi", m.I[i], v[:, :, i])
        f[:, :, i] = np.einsum("ij,bj->bi", m.I[i], a[:, :, i]) + fx._fxv(v[:, :, i], Iv)       # :595-596
    return (v[0], a[0], f[0]) if unb else (v, a, f)


def rnea_bpass(m: FbModel, q, f):
    """RBDReference.py:600-621: c[inds_f] = S^T f (the base's six entries are f_0 itself), f accumulated."""
    q, unb = _batch(q, m.n)
    f = np.array(f, dtype=np.float64, copy=True)
    if unb:
        f = f[None]
    B = q.shape[0]
    X = Xmats(m, q)
    c = np.zeros((B, m.n))
    for i in range(m.nb - 1, -1, -1):
        if i == 0:
            c[:, 0:6] = f[:, :, 0]                                                      # :612 with S = eye(6)
        else:
            c[:, i + 5] = np.einsum("j,bj->b", m.S[i], f[:, :, i])
        p = m.parent[i]
        if p != -1:
            f[:, :, p] += np.einsum("bji,bj->bi", X[:, i], f[:, :, i])                 # :618-619
    return (c[0], f[0]) if unb else (c, f)


def rnea(m: FbModel, q, qd, qdd=None, GRAVITY=-9.81):
    v, a, f = rnea_fpass(m, q, qd, qdd, GRAVITY)
    c, f = rnea_bpass(m, q, f)
    return c, v, a, f


def minv_bpass(m: FbModel, q):
    """RBDReference.py:630-735, floating-base branches (:652-691): matrix index = body + 5, the base
    handled as one 6 x 6 block (fb_Dinv = inv(S^T IA_0 S), :679-683)."""
    q, unb = _batch(q, m.n)
    B = q.shape[0]; n = m.n
    X = Xmats(m, q)
    Minv = np.zeros((B, n, n)); F = np.zeros((B, n, 6, n)); U = np.zeros((B, n, 6)); Dinv = np.zeros((B, n))
    IA = np.broadcast_to(m.I[None], (B, m.nb, 6, 6)).copy()                             # :662
    for i in range(m.nb - 1, -1, -1):
        sub = [j + 5 for j in m.subtree[i]]                                             # :668-671
        p = m.parent[i]
        if p == -1:                                                                     # :676-691 (base)
            U[:, 0:6, :] = IA[:, 0]                                                     # :680  IA S, S = eye(6)
            fb = np.linalg.inv(IA[:, 0])                                                # :681-683
            Minv[:, 0:6, 0:6] = Minv[:, 0:1, 0:1] + fb                                  # :685  (Minv[0, 0] is 0 here)
            Minv[:, 0:6, sub] -= fb @ F[:, 5][:, :, sub]                                # :686-691: the `[-1]` selects F[5], the base's F slot
        else:
            mi = i + 5; mp = p + 5
            U[:, mi] = np.einsum("bij,j->bi", IA[:, i], m.S[i])                         # :697
            Dinv[:, mi] = np.einsum("j,bj->b", m.S[i], U[:, mi])                        # :698 (holds D)
            Minv[:, mi, mi] = 1.0 / Dinv[:, mi]                                         # :700
            Minv[:, mi, sub] -= (1.0 / Dinv[:, mi])[:, None] * np.einsum("j,bjs->bs", m.S[i], F[:, mi][:, :, sub])   # :702-708
            for s in sub:                                                               # :720-726
                F[:, mi, :, s] += U[:, mi] * Minv[:, mi, s][:, None]
                F[:, mp, :, s] += np.einsum("bji,bj->bi", X[:, i], F[:, mi, :, s])
            Ia = IA[:, i] - np.einsum("bi,bj->bij", U[:, mi], U[:, mi]) / Dinv[:, mi][:, None, None]   # :728-731
            IA[:, p] += np.einsum("bji,bjk,bkl->bil", X[:, i], Ia, X[:, i])             # :732-733
    return (Minv[0], F[0], U[0], Dinv[0]) if unb else (Minv, F, U, Dinv)


def minv_fpass(m: FbModel, q, Minv, F, U, Dinv):
    """RBDReference.py:737-783: F is indexed by BODY id here (:772, :774, :779), Minv rows by body + 5."""
    q, unb = _batch(q, m.n)
    Minv = np.array(Minv, dtype=np.float64, copy=True); F = np.array(F, dtype=np.float64, copy=True)
    U = np.asarray(U, dtype=np.float64); Dinv = np.asarray(Dinv, dtype=np.float64)
    if unb:
        Minv, F, U, Dinv = Minv[None], F[None], U[None], Dinv[None]
    X = Xmats(m, q)
    for i in range(m.nb):
        p = m.parent[i]
        if p == -1:
            F[:, 0] = Minv[:, 0:6, :]                                                   # :779  S @ Minv[0:6, 0:]
        else:
            mi = i + 5
            UX = np.einsum("bj,bjk->bk", U[:, mi], X[:, i])
            Minv[:, mi, :] -= (1.0 / Dinv[:, mi])[:, None] * np.einsum("bk,bkc->bc", UX, F[:, p])       # :771-773
            F[:, i] = X[:, i] @ F[:, p] + np.einsum("j,bc->bjc", m.S[i], Minv[:, mi, :])                 # :774-776
    return Minv[0] if unb else Minv


def minv(m: FbModel, q, output_dense=True):
    """RBDReference.py:785-806.  With a floating base the forward pass completes EVERY row (the base rows
    are dense from the start), so the matrix is the full inverse either way; the reference's mirror loop
    (:799-804, over range(NB) only) touches the already symmetric top-left block."""
    Mb, F, U, D = minv_bpass(m, q)
    Mi = minv_fpass(m, q, Mb, F, U, D)
    if output_dense:
        nb = m.nb
        Mi = np.array(Mi, copy=True)
        M2 = Mi if Mi.ndim == 3 else Mi[None]
        for col in range(nb):
            for row in range(nb):
                if col < row:
                    M2[:, row, col] = M2[:, col, row]
    return Mi


def forward_dynamics(m: FbModel, q, qd, u, GRAVITY=-9.81):
    """RBDReference.py:1371-1374: minv(q) @ (u - rnea(q, qd)[0])."""
    c = rnea(m, q, qd, None, GRAVITY)[0]
    Mi = minv(m, q)
    u = np.asarray(u, dtype=np.float64)
    return np.einsum("...ij,...j->...i", Mi, u - c)


def joint_space_inertia(m: FbModel, q):
    """H from rnea columns (the reference's crba raises for floating bases): H[:, k] = rnea(q, 0, e_k) - rnea(q, 0, 0),
    gravity off.  Witness for Minv H = I."""
    q, unb = _batch(q, m.n)
    B = q.shape[0]
    z = np.zeros((B, m.n))
    c0 = rnea(m, q, z, z, GRAVITY=0.0)[0]
    H = np.zeros((B, m.n, m.n))
    for k in range(m.n):
        e = np.zeros((B, m.n)); e[:, k] = 1.0
        H[:, :, k] = rnea(m, q, z, e, GRAVITY=0.0)[0] - c0
    return H[0] if unb else H


def rnea_grad(m: FbModel, q, qd, qdd=None, GRAVITY=-9.81, USE_VELOCITY_DAMPING=False):
    """RBDReference.rnea_grad (:1345-1368) with the floating-base branches of its four passes (:1141-1147,
    :1165-1175, :1212-1243, :1267-1294, :1309-1341), restated column by column.  The base's six "position"
    columns are derivatives along a base-frame twist (the reference perturbs X_0 with crm(.) S, S = eye(6)), not
    along the six coordinates of q[0:6].  Only defined for NB >= 6: the reference's dq forward pass indexes
    bodies 0..5 for the base (:1168) and raises IndexError on smaller robots; on larger ones those updates
    add zeros, and its result agrees with central differences.  Literal quirks kept: velocity damping is
    added at matrix index `ind` (the BODY id, :1341) and, for the base, to a whole 5 x 5 block (:1339)."""
    assert m.nb >= 6, "the reference's floating-base rnea_grad raises IndexError for NB < 6 (:1168)"
    q, unb = _batch(q, m.n); qd, _ = _batch(qd, m.n)
    qdd_b = None if qdd is None else _batch(qdd, m.n)[0]
    B = q.shape[0]; n = m.n; nb = m.nb
    c, v, a, f = rnea(m, q, qd, qdd_b, GRAVITY)
    X = Xmats(m, q)
    g = np.zeros(6); g[5] = -GRAVITY
    dc = np.zeros((B, n, 2 * n))
    crm = fx._crm
    for col in range(n):
        for isqd in (False, True):
            dv = np.zeros((B, nb, 6)); da = np.zeros((B, nb, 6)); df = np.zeros((B, nb, 6))
            for i in range(nb):
                p = m.parent[i]
                if i == 0:
                    if col < 6:
                        e = np.zeros(6); e[col] = 1.0
                        if isqd:
                            dv[:, 0] = e                                                        # :1231  dv[:, inds_v, 0] += S
                        else:
                            da[:, 0] = np.einsum("bij,j->bi", crm(X[:, 0] @ g), e)            # :1175  crm(X a_grav) S
                    if isqd:
                        da[:, 0] += np.einsum("bij,bj->bi", crm(dv[:, 0]), qd[:, 0:6])        # :1236-1238  sum_ii qd_ii crm(dv) S[ii]
                        if col < 6:
                            da[:, 0, :] += crm(v[:, :, 0])[:, :, col]                          # :1243  crm(v) S
                else:
                    idx = i + 5
                    Xi = X[:, i]
                    dv[:, i] = np.einsum("bij,bj->bi", Xi, dv[:, p])                          # :1158 / :1230
                    da[:, i] = np.einsum("bij,bj->bi", Xi, da[:, p])                          # :1163 / :1234
                    if col == idx:
                        if isqd:
                            dv[:, i] += m.S[i]                                                  # :1231
                        else:
                            dv[:, i] += np.einsum("bij,j->bi", crm(np.einsum("bij,bj->bi", Xi, v[:, :, p])), m.S[i])   # :1159
                    da[:, i] += qd[:, idx][:, None] * np.einsum("bij,j->bi", crm(dv[:, i]), m.S[i])   # :1170 / :1240
                    if col == idx:
                        if isqd:
                            da[:, i] += np.einsum("bij,j->bi", crm(v[:, :, i]), m.S[i])       # :1243
                        else:
                            da[:, i] += np.einsum("bij,j->bi", crm(np.einsum("bij,bj->bi", Xi, a[:, :, p])), m.S[i])   # :1173
                Iv = np.einsum("ij,bj->bi", m.I[i], v[:, :, i])
                df[:, i] = (np.einsum("ij,bj->bi", m.I[i], da[:, i]) + fx._fxv(dv[:, i], Iv)
                            + fx._fxv(v[:, :, i], np.einsum("ij,bj->bi", m.I[i], dv[:, i])))   # :1179-1185 / :1247-1252
            out = n + col if isqd else col
            for i in range(nb - 1, -1, -1):
                p = m.parent[i]
                if i == 0:
                    dc[:, 0:6, out] = df[:, 0]                                                  # :1282 / :1325 with S = eye(6)
                else:
                    idx = i + 5
                    dc[:, idx, out] = np.einsum("j,bj->b", m.S[i], df[:, i])                  # :1284 / :1325
                    add = df[:, i].copy()
                    if (not isqd) and col == idx:
                        add += -np.einsum("bij,j->bi", crm(f[:, :, i]), m.S[i])               # :1292-1294  fxS(S, f) = -crm(f) S
                    df[:, p] += np.einsum("bji,bj->bi", X[:, i], add)                         # :1291 / :1331
    if USE_VELOCITY_DAMPING:                                                                     # :1336-1341, literally
        for ind in range(nb):
            if m.parent[ind] == -1:
                dc[:, ind:ind + 5, n + ind:n + ind + 5] += m.damping[ind]
            else:
                dc[:, ind, n + ind] += m.damping[ind]
    return dc[0] if unb else dc


# ---- the four gradient passes as the reference exposes them (README.md:19: the accelerator-testing surface) ----
# Layouts follow the reference: dv / da / df are (6, n, NB) per configuration -> [B, 6, n, NB] here; the backward
# passes mutate df in place (:1291, :1331): the mutated copy is returned next to dc.

def _cols_of(i):
    return list(range(6)) if i == 0 else [i + 5]


def rnea_grad_fpass_dq(m: FbModel, q, qd, v, a, GRAVITY=-9.81):
    """RBDReference.py:1127-1187 with the floating-base branches (:1141-1147, :1165-1169, :1173-1175).  NB >= 6 only
    (the reference indexes bodies 0..5 at :1168); those updates add zeros (dv_dq of the base is zero)."""
    assert m.nb >= 6
    q, unb = _batch(q, m.n); qd, _ = _batch(qd, m.n)
    v = np.asarray(v, dtype=np.float64); a = np.asarray(a, dtype=np.float64)
    if unb:
        v, a = v[None], a[None]
    B = q.shape[0]; n = m.n; nb = m.nb
    X = Xmats(m, q)
    g = np.zeros(6); g[5] = -GRAVITY
    crm = fx._crm
    dv = np.zeros((B, 6, n, nb)); da = np.zeros((B, 6, n, nb)); df = np.zeros((B, 6, n, nb))
    for i in range(nb):
        p = m.parent[i]
        if p != -1:
            idx = i + 5
            Xi = X[:, i]
            dv[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, dv[:, :, :, p])                                  # :1158
            dv[:, :, idx, i] += np.einsum("bij,j->bi", crm(np.einsum("bij,bj->bi", Xi, v[:, :, p])), m.S[i])   # :1159
            da[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, da[:, :, :, p])                                  # :1163
            da[:, :, :, i] += qd[:, idx][:, None, None] * np.einsum("bcij,j->bic", crm(np.moveaxis(dv[:, :, :, i], 1, 2)), m.S[i])   # :1170
            da[:, :, idx, i] += np.einsum("bij,j->bi", crm(np.einsum("bij,bj->bi", Xi, a[:, :, p])), m.S[i])   # :1173
        else:
            # :1165-1168 add crm(dv) S[ii] qd[ii] with dv = 0 -> nothing;  :1175: da[:, idx, 0] += crm(X a_grav) S
            da[:, :, 0:6, 0] += crm(X[:, 0] @ g)
        Iv = np.einsum("ij,bj->bi", m.I[i], v[:, :, i])
        df[:, :, :, i] = np.einsum("ij,bjc->bic", m.I[i], da[:, :, :, i])                                   # :1180
        dvi = np.moveaxis(dv[:, :, :, i], 1, 2)                                                             # [B, n, 6]
        df[:, :, :, i] += np.moveaxis(fx._fxv(dvi.reshape(B * n, 6), np.repeat(Iv, n, axis=0)).reshape(B, n, 6), 1, 2)   # :1184
        Idv = np.einsum("ij,bcj->bci", m.I[i], dvi)
        df[:, :, :, i] += np.moveaxis(fx._fxv(np.repeat(v[:, :, i], n, axis=0), Idv.reshape(B * n, 6)).reshape(B, n, 6), 1, 2)   # :1185
    return (dv[0], da[0], df[0]) if unb else (dv, da, df)


def rnea_grad_fpass_dqd(m: FbModel, q, qd, v):
    """RBDReference.py:1189-1255 with the floating-base branches (:1212-1218, :1231, :1235-1243)."""
    q, unb = _batch(q, m.n); qd, _ = _batch(qd, m.n)
    v = np.asarray(v, dtype=np.float64)
    if unb:
        v = v[None]
    B = q.shape[0]; n = m.n; nb = m.nb
    X = Xmats(m, q)
    crm = fx._crm
    dv = np.zeros((B, 6, n, nb)); da = np.zeros((B, 6, n, nb)); df = np.zeros((B, 6, n, nb))
    for i in range(nb):
        p = m.parent[i]
        if p != -1:
            idx = i + 5
            Xi = X[:, i]
            dv[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, dv[:, :, :, p])                                  # :1230
            dv[:, :, idx, i] += m.S[i]                                                                      # :1231
            da[:, :, :, i] = np.einsum("bij,bjc->bic", Xi, da[:, :, :, p])                                  # :1234
            da[:, :, :, i] += qd[:, idx][:, None, None] * np.einsum("bcij,j->bic", crm(np.moveaxis(dv[:, :, :, i], 1, 2)), m.S[i])   # :1240
            da[:, :, idx, i] += np.einsum("bij,j->bi", crm(v[:, :, i]), m.S[i])                              # :1243
        else:
            dv[:, :, 0:6, 0] += np.eye(6)                                                                   # :1231 with S = eye(6)
            # :1236-1238: sum_ii qd_ii crm(dv[:, c]) S[ii] = crm(dv[:, c]) qd[0:6]
            da[:, :, :, 0] += np.einsum("bcij,bj->bic", crm(np.moveaxis(dv[:, :, :, 0], 1, 2)), qd[:, 0:6])
            da[:, :, 0:6, 0] += crm(v[:, :, 0])                                                             # :1243  crm(v) S
        Iv = np.einsum("ij,bj->bi", m.I[i], v[:, :, i])
        df[:, :, :, i] = np.einsum("ij,bjc->bic", m.I[i], da[:, :, :, i])                                   # :1247
        dvi = np.moveaxis(dv[:, :, :, i], 1, 2)
        df[:, :, :, i] += np.moveaxis(fx._fxv(dvi.reshape(B * n, 6), np.repeat(Iv, n, axis=0)).reshape(B, n, 6), 1, 2)   # :1251
        Idv = np.einsum("ij,bcj->bci", m.I[i], dvi)
        df[:, :, :, i] += np.moveaxis(fx._fxv(np.repeat(v[:, :, i], n, axis=0), Idv.reshape(B * n, 6)).reshape(B, n, 6), 1, 2)   # :1252
    return (dv[0], da[0], df[0]) if unb else (dv, da, df)


def rnea_grad_bpass_dq(m: FbModel, q, f, df_dq):
    """RBDReference.py:1257-1297: dc_dq[0:6] = df[:, :, 0] (:1282), dc_dq[i + 5] = S^T df[:, :, i]; X^T df and the
    fxS term go to the parent IN PLACE.  Returns (dc_dq, df_dq after the pass)."""
    q, unb = _batch(q, m.n)
    f = np.asarray(f, dtype=np.float64)
    df = np.array(df_dq, dtype=np.float64, copy=True)
    if unb:
        f, df = f[None], df[None]
    B = q.shape[0]; n = m.n
    X = Xmats(m, q)
    dc = np.zeros((B, n, n))
    for i in range(m.nb - 1, -1, -1):
        p = m.parent[i]
        if p == -1:
            dc[:, 0:6, :] = df[:, :, :, 0]
        else:
            idx = i + 5
            dc[:, idx, :] = np.einsum("j,bjc->bc", m.S[i], df[:, :, :, i])
            df[:, :, :, p] += np.einsum("bji,bjc->bic", X[:, i], df[:, :, :, i])                            # :1291
            fxS = -np.einsum("bij,j->bi", fx._crm(f[:, :, i]), m.S[i])                                      # :166-168
            df[:, :, idx, p] += np.einsum("bji,bj->bi", X[:, i], fxS)                                       # :1292-1294
    return (dc[0], df[0]) if unb else (dc, df)


def rnea_grad_bpass_dqd(m: FbModel, q, df_dqd, USE_VELOCITY_DAMPING=False):
    """RBDReference.py:1299-1343; the damping lines (:1336-1341) literally: matrix index = BODY id, a 5 x 5 block for
    the base.  Returns (dc_dqd, df_dqd after the pass)."""
    q, unb = _batch(q, m.n)
    df = np.array(df_dqd, dtype=np.float64, copy=True)
    if unb:
        df = df[None]
    B = q.shape[0]; n = m.n
    X = Xmats(m, q)
    dc = np.zeros((B, n, n))
    for i in range(m.nb - 1, -1, -1):
        p = m.parent[i]
        if p == -1:
            dc[:, 0:6, :] = df[:, :, :, 0]                                                                  # :1325 with S = eye(6)
        else:
            dc[:, i + 5, :] = np.einsum("j,bjc->bc", m.S[i], df[:, :, :, i])
            df[:, :, :, p] += np.einsum("bji,bjc->bic", X[:, i], df[:, :, :, i])                            # :1331
    if USE_VELOCITY_DAMPING:
        for ind in range(m.nb):
            if m.parent[ind] == -1:
                dc[:, ind:ind + 5, ind:ind + 5] += m.damping[ind]
            else:
                dc[:, ind, ind] += m.damping[ind]
    return (dc[0], df[0]) if unb else (dc, df)


def forward_dynamics_grad(m: FbModel, q, qd, u, GRAVITY=-9.81):
    """RBDReference.py:1376-1384: qdd = forward_dynamics; [qdd_dq | qdd_dqd] = -Minv dc_du(q, qd, qdd)."""
    q2, unb = _batch(q, m.n)
    qdd = forward_dynamics(m, q, qd, u, GRAVITY)
    dc = rnea_grad(m, q, qd, qdd, GRAVITY)
    Mi = minv(m, q)
    d = -np.einsum("...ij,...jk->...ik", Mi, dc)
    return d[..., :, :m.n], d[..., :, m.n:]
