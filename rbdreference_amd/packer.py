"""Robot -> packed model -> generated C++ header for the per-robot HIP specialisation.

The reference reads the robot through getters on every call and re-evaluates ``Xmat(q)`` per pass
(``/root/reference/RBDReference.py:570-574, 617, 718, 769, 1154, 1226, 1290, 1330``).  Here the
robot is read ONCE, validated, and frozen into compile-time constants: topology, joint axes, the
constant tree transform ``X_tree = X(0)`` and the spatial inertias.  The HIP kernels
(``csrc/rbd_kernels.hip``) are compiled against the generated header, so every structural zero of
``X_tree`` / ``I`` and the whole kinematic tree are resolved by the compiler (SURVEY.md §7.3
"Topology genericity", "Uniform model constants").

Supported joints: 1-DoF revolute about / prismatic along a body-frame coordinate axis
(``S = e_k`` or ``e_{3+k}``), ``X(q) = X_J(q) @ X(0)`` with the Featherstone joint transform
(SURVEY.md Appendix A).  Anything else raises -- there is no fallback path.
"""
from __future__ import annotations

import hashlib
import math
import struct
from dataclasses import dataclass
from typing import List

import numpy as np

__all__ = ["PackedModel", "pack_robot", "emit_header", "safe_name", "ABI_VERSION"]

ABI_VERSION = 2       # 2: rbd_model_info_t gained floating_base / nv; floating-base libraries


@dataclass
class PackedModel:
    name: str
    n: int
    parent: List[int]
    jtype: List[int]          # 0 revolute, 1 prismatic
    axis: List[int]           # 0/1/2
    Xtree: np.ndarray         # [n, 6, 6] float64, = X_i(0)
    I: np.ndarray             # [n, 6, 6] float64
    damping: np.ndarray       # [n] float64
    depth: List[int]          # 0 for children of the base
    hash: str                 # 16 hex digits over everything above (not the name)
    floating: bool = False    # body 0 is attached by a 6-DoF joint (jtype[0] == 2), S = eye(6)

    @property
    def nv(self) -> int:
        """Velocities (= columns of q, qd, qdd): n for fixed-base robots, n + 5 with a floating base."""
        return self.n + 5 if self.floating else self.n

    @property
    def max_depth(self) -> int:
        return max(self.depth) + 1

    def ancestors(self, i: int) -> List[int]:
        """Strict ancestors of body i, root first."""
        out = []
        p = self.parent[i]
        while p != -1:
            out.append(p)
            p = self.parent[p]
        return out[::-1]


def _joint_X(jtype: int, k: int, q: float) -> np.ndarray:
    a, b = (k + 1) % 3, (k + 2) % 3
    X = np.eye(6)
    if jtype == 0:
        c, s = math.cos(q), math.sin(q)
        E = np.eye(3)
        E[a, a] = c; E[a, b] = s; E[b, a] = -s; E[b, b] = c
        X[:3, :3] = E; X[3:, 3:] = E
    else:
        # plux(1, q e_k): lower-left block = -(q e_k)^x
        X[3 + a, b] = q
        X[3 + b, a] = -q
    return X


def _pack_floating_base_joint(robot, n):
    """Validate body 0 of a floating-base robot: S = eye(6), indices 0..5, and the world -> base
    transform of robot.floating_base_X (Px, Py, Pz, Rx, Ry, Rz; RBDReference.py:634-637)."""
    from .robot import floating_base_X
    if int(robot.get_parent_id(0)) != -1:
        raise ValueError("floating base: body 0 must be the base")
    if not np.array_equal(np.asarray(robot.get_S_by_id(0), dtype=np.float64), np.eye(6)):
        raise ValueError("floating base: S of body 0 must be eye(6) (RBDReference.py:679)")
    for g in ("get_joint_index_q", "get_joint_index_v", "get_joint_index_f"):
        if list(getattr(robot, g)(0)) != [0, 1, 2, 3, 4, 5]:
            raise ValueError(f"floating base: {g}(0) must be [0..5]")
    f = robot.get_Xmat_Func_by_id(0)
    for qq in (np.zeros(6), np.array([0.3, -0.2, 0.5, 0.4, -1.1, 2.0]), np.array([-1.0, 0.7, 0.1, -2.5, 0.9, -0.3])):
        if not np.allclose(np.asarray(f(qq), dtype=np.float64), floating_base_X(qq), rtol=0, atol=1e-12):
            raise ValueError("floating base: Xmat of body 0 is not plux(Rz Ry Rx, p) of (px,py,pz,rx,ry,rz) "
                             "(unsupported base parametrisation)")


def pack_robot(robot, name: str | None = None) -> PackedModel:
    floating = bool(getattr(robot, "floating_base", False))
    n = int(robot.get_num_bodies())
    if int(robot.get_num_vel()) != (n + 5 if floating else n):
        raise ValueError("only 1-DoF joints (and one 6-DoF floating base) are supported (num_vel mismatch)")
    if not (1 <= n <= 64):
        raise ValueError(f"n = {n}: supported range is 1..64")
    if floating:
        _pack_floating_base_joint(robot, n)
    off = 5 if floating else 0
    parent, jtype, axis = [], [], []
    Xt = np.zeros((n, 6, 6)); Im = np.zeros((n, 6, 6)); damp = np.zeros(n)
    for i in range(n):
        p = int(robot.get_parent_id(i))
        if not (-1 <= p < i):
            raise ValueError(f"body {i}: parent {p} must precede it")
        if floating and i > 0 and p == -1:
            raise ValueError(f"body {i}: with a floating base every other body must descend from body 0")
        if floating and i == 0:
            parent.append(-1); jtype.append(2); axis.append(0)
            Xt[0] = np.eye(6)
            I = np.asarray(robot.get_Imat_by_id(0), dtype=np.float64).reshape(6, 6)
            Im[0] = 0.5 * (I + I.T)
            damp[0] = float(robot.get_damping_by_id(0))     # the base's 5x5 damping block (:1336-1339)
            continue
        for g in ("get_joint_index_q", "get_joint_index_v", "get_joint_index_f"):
            if int(getattr(robot, g)(i)) != i + off:
                raise ValueError(f"body {i}: {g} != body id{' + 5' if floating else ''} (1-DoF layout expected)")
        parent.append(p)
        S = np.asarray(robot.get_S_by_id(i), dtype=np.float64).reshape(-1)
        nz = np.flatnonzero(S)
        if S.shape != (6,) or len(nz) != 1 or S[nz[0]] != 1.0:
            raise ValueError(f"body {i}: S = {S} is not a unit coordinate axis")
        jtype.append(0 if nz[0] < 3 else 1)
        axis.append(int(nz[0] % 3))
        f = robot.get_Xmat_Func_by_id(i)
        X0 = np.asarray(f(0.0), dtype=np.float64).reshape(6, 6)
        if np.any(X0[:3, 3:] != 0.0) or not np.allclose(X0[:3, :3], X0[3:, 3:], rtol=0, atol=1e-13):
            raise ValueError(f"body {i}: X(0) is not a Plucker motion transform [[E,0],[-E r^x,E]]")
        for qq in (0.37, -2.1, 3.0):
            want = np.asarray(f(qq), dtype=np.float64)
            got = _joint_X(jtype[-1], axis[-1], qq) @ X0
            if not np.allclose(got, want, rtol=0, atol=1e-12):
                raise ValueError(f"body {i}: Xmat(q) != X_J(q) @ X(0) (unsupported joint convention)")
        Xt[i] = X0
        I = np.asarray(robot.get_Imat_by_id(i), dtype=np.float64).reshape(6, 6)
        if not np.allclose(I, I.T, rtol=0, atol=1e-12 * max(1.0, np.abs(I).max())):
            raise ValueError(f"body {i}: spatial inertia is not symmetric")
        Im[i] = 0.5 * (I + I.T)
        damp[i] = float(robot.get_damping_by_id(i))
        st = sorted(int(j) for j in robot.get_subtree_by_id(i))
        if i not in st:
            raise ValueError(f"body {i}: get_subtree_by_id must include the body itself")
    depth = []
    for i in range(n):
        depth.append(0 if parent[i] == -1 else depth[parent[i]] + 1)
    for i in range(n):  # subtree getter must agree with parent[]
        st = sorted(int(j) for j in robot.get_subtree_by_id(i))
        mine = [j for j in range(n) if _is_anc_or_self(parent, i, j)]
        if st != mine:
            raise ValueError(f"body {i}: get_subtree_by_id disagrees with get_parent_id")
    h = hashlib.sha256()
    h.update(struct.pack("<iii", ABI_VERSION, n, int(floating)))
    h.update(np.asarray(parent, dtype=np.int32).tobytes())
    h.update(np.asarray(jtype, dtype=np.int32).tobytes())
    h.update(np.asarray(axis, dtype=np.int32).tobytes())
    h.update(Xt.tobytes()); h.update(Im.tobytes()); h.update(damp.tobytes())
    return PackedModel(name or getattr(robot, "name", "robot"), n, parent, jtype, axis, Xt, Im,
                       damp, depth, h.hexdigest()[:16], floating)


def _is_anc_or_self(parent, i, j) -> bool:
    while j != -1:
        if j == i:
            return True
        j = parent[j]
    return False


def _carr(vals, fmt=lambda v: str(int(v))) -> str:
    return "{" + ", ".join(fmt(v) for v in vals) + "}"


def _hexf(v: float) -> str:
    v = float(v)
    if v == 0.0:
        return "0.0"
    return v.hex()


def safe_name(name: str) -> str:
    """The robot name as it may appear in generated C++ and in file names: ``[A-Za-z0-9_]`` only, at
    most 63 characters (``rbd_model_info_t.name`` holds 64 bytes).  The name comes from the caller or
    from a URDF's ``<robot name=...>`` -- i.e. from untrusted text -- and the generated header is
    compiled and dlopen'ed in-process, so nothing else may pass."""
    s = "".join(ch if (ch.isascii() and (ch.isalnum() or ch == "_")) else "_" for ch in str(name))[:63]
    return s or "robot"


def emit_header(m: PackedModel) -> str:
    """C++ header text consumed by csrc/rbd_kernels.hip (``-include`` on the hipcc command line).
    Floating-point constants are written as hex-float literals, i.e. bit-exact."""
    n = m.n
    L = []
    L.append(f"// GENERATED by rbdreference_amd/packer.py -- model {safe_name(m.name)}, hash {m.hash}. Do not edit.")
    L.append("#pragma once")
    L.append(f'#define RBD_MODEL_NAME "{safe_name(m.name)}"')
    L.append(f"#define RBD_MODEL_HASH 0x{m.hash}ULL")
    L.append(f"#define RBD_ABI_VERSION {ABI_VERSION}")
    L.append("namespace rbdm {")
    L.append(f"constexpr int N = {n};")
    L.append(f"constexpr bool FLOATING_BASE = {'true' if m.floating else 'false'};   // body 0: 6-DoF joint, S = eye(6)")
    L.append(f"constexpr int NV = {m.nv};                 // velocities: N, or N + 5 with a floating base")
    L.append(f"constexpr int MAXDEPTH = {m.max_depth};")
    L.append(f"constexpr int PARENT[N] = {_carr(m.parent)};")
    L.append(f"constexpr int DEPTH[N] = {_carr(m.depth)};")
    L.append(f"constexpr int JTYPE[N] = {_carr(m.jtype)};   // 0 revolute, 1 prismatic, 2 floating base")
    L.append(f"constexpr int AXIS[N] = {_carr(m.axis)};")
    L.append("constexpr double XT[N][36] = {")
    for i in range(n):
        L.append("  " + _carr(m.Xtree[i].reshape(-1), _hexf) + ",")
    L.append("};")
    L.append("constexpr double IM[N][36] = {")
    for i in range(n):
        L.append("  " + _carr(m.I[i].reshape(-1), _hexf) + ",")
    L.append("};")
    L.append(f"constexpr double DAMPING[N] = {_carr(m.damping, _hexf)};")
    L.append("}  // namespace rbdm")
    return "\n".join(L) + "\n"
