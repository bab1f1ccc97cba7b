"""Robot model with the URDFParser getter surface the rnea / rnea_grad / minv path uses.

The reference never builds a robot itself: ``RBDReference(robotObj)`` receives an object made by
the external URDFParser package (``/root/reference/RBDReference.py:6-7``, ``README.md:8``) and reads
it only through getters.  The hot path touches 12 getters + 1 attribute (SURVEY.md §8a row a13;
call sites ``RBDReference.py:561-610, 651-718, 758-769, 1130-1177, 1203-1245, 1260-1290,
1302-1341``).  This class implements exactly that surface, so the *same object* can be handed to
the real reference (to generate golden vectors) and to this package (to pack a device model).

Conventions (the reference is self-consistent under them):

* spatial vectors are ``[angular(3); linear(3)]`` (``RBDReference.py:11-20``);
* ``X_i(q) = X_J(q) @ X_tree_i`` is the parent->child *motion* transform,
  ``plux(E, r) = [[E, 0], [-E r^x, E]]``; forces go child->parent with ``X^T``
  (``RBDReference.py:618``);
* revolute joint about axis k: ``S = e_k``, ``X_J = plux(R_k(q), 0)`` with the coordinate-transform
  rotation (``R_z = [[c, s, 0], [-s, c, 0], [0, 0, 1]]``); prismatic along k: ``S = e_{3+k}``,
  ``X_J = plux(1, q e_k)``;
* ``I = [[Ic + m c^x c^x^T, m c^x], [m c^x^T, m 1]]`` in the joint frame (``RBDReference.py:596``);
* bodies are numbered parents-first; ``parent = -1`` marks a child of the fixed base, and several
  bodies may have it (``RBDReference.py:576``).

The three built-in robots match the BASELINE.json topologies.  Their link constants are authored
here (the reference ships no robot and no URDF): kinematic structure and magnitudes follow the
public iiwa14 / HyQ / Atlas descriptions, but parity never depends on them being URDF-exact because
the same object feeds both sides.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Sequence

import numpy as np

__all__ = ["Link", "Robot", "FloatingBaseRobot", "floating_base_X", "iiwa_like", "quadruped_like",
           "atlas_like", "random_tree", "floating_quadruped_like", "BUILTIN_ROBOTS", "builtin_robot"]


def _skew(r: Sequence[float]) -> np.ndarray:
    x, y, z = (float(t) for t in r)
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def _snap(M: np.ndarray, tol: float = 1e-14) -> np.ndarray:
    """Snap entries within `tol` of 0 / +-1 so that rpy multiples of pi/2 give exact matrices."""
    M = np.array(M, dtype=np.float64)
    M[np.abs(M) < tol] = 0.0
    M[np.abs(M - 1.0) < tol] = 1.0
    M[np.abs(M + 1.0) < tol] = -1.0
    return M


def _rot_axis(k: int, q: float) -> np.ndarray:
    """Coordinate-transform rotation about axis k (transpose of the active rotation)."""
    # (math.cos raises for an infinite angle; numpy's answer for it -- NaN -- is what an `Xmat(q)` closure built on numpy
    # hands the reference, and what the kernels produce; finite angles are untouched)
    c, s = (math.cos(q), math.sin(q)) if math.isfinite(q) else (float("nan"), float("nan"))
    if k == 0:
        return np.array([[1.0, 0, 0], [0, c, s], [0, -s, c]])
    if k == 1:
        return np.array([[c, 0, -s], [0, 1.0, 0], [s, 0, c]])
    return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1.0]])


def _rpy_E(rpy: Sequence[float]) -> np.ndarray:
    """E for a URDF ``rpy`` origin: child-frame coordinates of parent-frame vectors."""
    r, p, y = (float(t) for t in rpy)
    # active rotation R = Rz(y) Ry(p) Rx(r); E = R^T = Rx(r)^T Ry(p)^T Rz(y)^T
    return _snap(_rot_axis(0, r) @ _rot_axis(1, p) @ _rot_axis(2, y))


def plux(E: np.ndarray, r: Sequence[float]) -> np.ndarray:
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ _skew(r)
    return X


def spatial_inertia(mass: float, com: Sequence[float], Ic: np.ndarray) -> np.ndarray:
    C = _skew(com)
    I = np.zeros((6, 6))
    I[:3, :3] = np.asarray(Ic, dtype=np.float64) + mass * (C @ C.T)
    I[:3, 3:] = mass * C
    I[3:, :3] = mass * C.T
    I[3:, 3:] = mass * np.eye(3)
    return I


@dataclass
class Link:
    """One body + the joint that connects it to its parent."""
    name: str
    parent: int                       # body id of the parent, -1 for the fixed base
    axis: int                         # 0/1/2 = x/y/z
    xyz: Sequence[float]              # joint origin in the parent frame
    rpy: Sequence[float] = (0.0, 0.0, 0.0)
    mass: float = 1.0
    com: Sequence[float] = (0.0, 0.0, 0.0)
    inertia: Sequence[float] = (0.01, 0.01, 0.01, 0.0, 0.0, 0.0)  # ixx iyy izz ixy ixz iyz @ com
    damping: float = 0.0
    jtype: str = "revolute"           # or "prismatic"
    rot: object = None                # optional 3x3 E (child-frame coords of parent-frame vectors);
                                      # overrides rpy (used by the URDF loader for arbitrary frames)


class Robot:
    """Getter-compatible stand-in for a URDFParser robot (fixed base, 1-DoF joints)."""

    floating_base = False

    def __init__(self, name: str, links: List[Link]):
        self.name = name
        self.links = list(links)
        n = len(self.links)
        self._n = n
        self._parent = [int(l.parent) for l in self.links]
        for i, p in enumerate(self._parent):
            if not (-1 <= p < i):
                raise ValueError(f"body {i}: parent {p} must precede it (parents-first numbering)")
        self._S: List[np.ndarray] = []
        self._Xtree: List[np.ndarray] = []
        self._I: List[np.ndarray] = []
        self._Xfunc: List[Callable[[float], np.ndarray]] = []
        for l in self.links:
            S = np.zeros(6)
            if l.jtype == "revolute":
                S[l.axis] = 1.0
            elif l.jtype == "prismatic":
                S[3 + l.axis] = 1.0
            else:
                raise ValueError(f"unsupported joint type {l.jtype!r}")
            self._S.append(S)
            E = _rpy_E(l.rpy) if l.rot is None else _snap(np.asarray(l.rot, dtype=np.float64).reshape(3, 3))
            Xt = plux(E, l.xyz)
            self._Xtree.append(Xt)
            ixx, iyy, izz, ixy, ixz, iyz = (float(t) for t in l.inertia)
            Ic = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
            self._I.append(spatial_inertia(float(l.mass), l.com, Ic))
            self._Xfunc.append(self._make_xfunc(l.jtype, l.axis, Xt))
        self._subtree: List[List[int]] = [[i] for i in range(n)]
        for i in range(n - 1, -1, -1):
            p = self._parent[i]
            if p >= 0:
                self._subtree[p] = sorted(self._subtree[p] + self._subtree[i])

    @staticmethod
    def _make_xfunc(jtype: str, axis: int, Xt: np.ndarray) -> Callable[[float], np.ndarray]:
        if jtype == "revolute":
            def xf(q, _k=axis, _Xt=Xt):
                return plux(_rot_axis(_k, float(q)), (0.0, 0.0, 0.0)) @ _Xt
        else:
            def xf(q, _k=axis, _Xt=Xt):
                r = [0.0, 0.0, 0.0]
                r[_k] = float(q)
                return plux(np.eye(3), r) @ _Xt
        return xf

    # ---- URDFParser getter surface used by the hot path -----------------
    def get_num_bodies(self) -> int:
        return self._n

    def get_num_vel(self) -> int:
        return self._n

    def get_num_pos(self) -> int:
        return self._n

    def get_parent_id(self, i: int) -> int:
        return self._parent[i]

    def get_S_by_id(self, i: int) -> np.ndarray:
        return self._S[i].copy()

    def get_Xmat_Func_by_id(self, i: int) -> Callable[[float], np.ndarray]:
        return self._Xfunc[i]

    def get_Imat_by_id(self, i: int) -> np.ndarray:
        return self._I[i].copy()

    def get_Imats_dict_by_id(self) -> Dict[int, np.ndarray]:
        return {i: self._I[i].copy() for i in range(self._n)}

    def get_subtree_by_id(self, i: int) -> List[int]:
        return list(self._subtree[i])

    def get_joint_index_q(self, i: int) -> int:
        return i

    def get_joint_index_v(self, i: int) -> int:
        return i

    def get_joint_index_f(self, i: int) -> int:
        return i

    def get_damping_by_id(self, i: int) -> float:
        return float(self.links[i].damping)

    def __repr__(self) -> str:
        return f"Robot({self.name!r}, n={self._n})"


# ---------------------------------------------------------------------------------------------
# Built-in robots (topologies of SURVEY.md Appendix B)
# ---------------------------------------------------------------------------------------------
_PI = math.pi


def iiwa_like() -> Robot:
    """7-DoF serial arm with the kinematic structure of the KUKA LBR iiwa 14 (all revolute-z,
    joint frames related by multiples of pi/2)."""
    h = _PI / 2
    L = [
        Link("iiwa_link_1", -1, 2, (0, 0, 0.1575), (0, 0, 0), 4.0, (0, -0.03, 0.12),
             (0.1, 0.09, 0.02, 0, 0, 0), 0.5),
        Link("iiwa_link_2", 0, 2, (0, 0, 0.2025), (h, 0, _PI), 4.0, (0.0003, 0.059, 0.042),
             (0.05, 0.018, 0.044, 0, 0, 0), 0.5),
        Link("iiwa_link_3", 1, 2, (0, 0.2045, 0), (h, 0, _PI), 3.0, (0, 0.03, 0.13),
             (0.08, 0.075, 0.01, 0, 0, 0), 0.5),
        Link("iiwa_link_4", 2, 2, (0, 0, 0.2155), (h, 0, 0), 2.7, (0, 0.067, 0.034),
             (0.03, 0.01, 0.029, 0, 0, 0), 0.5),
        Link("iiwa_link_5", 3, 2, (0, 0.1845, 0), (-h, _PI, 0), 1.7, (0.0001, 0.021, 0.076),
             (0.02, 0.018, 0.005, 0, 0, 0), 0.5),
        Link("iiwa_link_6", 4, 2, (0, 0, 0.2155), (h, 0, 0), 1.8, (0, 0.0006, 0.0004),
             (0.005, 0.0036, 0.0047, 0, 0, 0), 0.5),
        Link("iiwa_link_7", 5, 2, (0, 0.081, 0), (-h, _PI, 0), 0.3, (0, 0, 0.02),
             (0.001, 0.001, 0.001, 0, 0, 0), 0.5),
    ]
    return Robot("iiwa_like", L)


def quadruped_like() -> Robot:
    """12-DoF: four independent 3-joint legs (hip abduction x, hip flexion y, knee y) on a fixed
    trunk; HyQ-like magnitudes.  Four bodies have parent -1 (independent roots)."""
    L: List[Link] = []
    legs = [("lf", 0.3735, 0.207), ("rf", 0.3735, -0.207), ("lh", -0.3735, 0.207),
            ("rh", -0.3735, -0.207)]
    for k, (nm, x, y) in enumerate(legs):
        sgn = 1.0 if y > 0 else -1.0
        b = 3 * k
        L.append(Link(f"{nm}_hipassembly", -1, 0, (x, y, 0.0), (0, 0, 0), 2.93 + 0.01 * k,
                      (0.0435, sgn * 0.004, -0.002), (0.0056, 0.0126, 0.0143, 0.0001 * sgn, 0.0003, 0.0),
                      0.1))
        L.append(Link(f"{nm}_upperleg", b, 1, (0.08, sgn * 0.02, 0.0), (0, 0, 0), 2.638 + 0.01 * k,
                      (0.0263, sgn * 0.001, -0.151), (0.0402, 0.0413, 0.0032, 0.0, 0.0009, 0.0001 * sgn),
                      0.1))
        L.append(Link(f"{nm}_lowerleg", b + 1, 1, (0.0, 0.0, -0.35), (0, 0, 0), 0.881 + 0.005 * k,
                      (0.012, 0.0, -0.125), (0.0107, 0.0108, 0.0005, 0.0, 0.0004, 0.0), 0.1))
    return Robot("quadruped_like", L)


def atlas_like() -> Robot:
    """30-DoF humanoid tree on a fixed pelvis: 3-joint back -> {neck; 7-joint left arm; 7-joint
    right arm}; pelvis -> two 6-joint legs.  parent[] per SURVEY.md Appendix B."""
    parent = [-1, 0, 1, 2, 2, 4, 5, 6, 7, 8, 9, 2, 11, 12, 13, 14, 15, 16,
              -1, 18, 19, 20, 21, 22, -1, 24, 25, 26, 27, 28]
    h = _PI / 2
    # (name, axis, xyz, rpy, mass, com, inertia)
    spec = [
        ("back_bkz", 2, (-0.0125, 0, 0), (0, 0, 0), 2.27, (-0.011, 0, 0.075), (0.0039, 0.0034, 0.0017, 0, -0.0001, 0)),
        ("back_bky", 1, (0, 0, 0.162), (0, 0, 0), 0.799, (-0.007, 0.0004, 0.0215), (0.0005, 0.0004, 0.0006, 0, 0, 0)),
        ("back_bkx", 0, (0, 0, 0.05), (0, 0, 0), 63.73, (-0.0582, 0, 0.1731), (1.577, 1.602, 0.565, 0.012, 0.0507, -0.007)),
        ("neck_ry", 1, (0.2546, 0, 0.5215), (0, 0, 0), 1.42, (-0.075, 0, 0.034), (0.0039, 0.0041, 0.0035, 0, 0.0009, 0)),
        ("l_arm_shz", 2, (0.1406, 0.2256, 0.4776), (0, 0, 0), 3.45, (-0.003, -0.099, -0.014), (0.002, 0.002, 0.003, 0, 0, 0.0005)),
        ("l_arm_shx", 0, (0, -0.11, -0.245), (h, 0, 0), 3.012, (0.0002, -0.007, -0.1), (0.0125, 0.0136, 0.0029, 0, 0.0001, -0.0003)),
        ("l_arm_ely", 1, (0, -0.187, -0.016), (0, 0, 0), 3.388, (-0.006, 0.04, -0.016), (0.0033, 0.0067, 0.0054, 0.0003, 0, 0.0002)),
        ("l_arm_elx", 0, (0, -0.119, 0.0092), (0, h, 0), 2.509, (0.0027, -0.14, 0.0153), (0.0077, 0.0024, 0.0074, -0.0004, 0, 0.0007)),
        ("l_arm_wry", 1, (0, -0.29955, -0.00921), (0, 0, 0), 0.35, (0.0, 0.015, -0.003), (0.0004, 0.0005, 0.0004, 0, 0, 0)),
        ("l_arm_wrx", 0, (0, 0, 0), (0, 0, 0), 0.35, (0.0, -0.015, 0.002), (0.0004, 0.0005, 0.0004, 0, 0, 0)),
        ("l_arm_wry2", 1, (0, -0.051, 0), (0, 0, 0), 0.6, (0.0, -0.09, 0.0), (0.0011, 0.0009, 0.0012, 0, 0, 0)),
        ("r_arm_shz", 2, (0.1406, -0.2256, 0.4776), (0, 0, 0), 3.45, (-0.003, 0.099, -0.014), (0.002, 0.002, 0.003, 0, 0, -0.0005)),
        ("r_arm_shx", 0, (0, 0.11, -0.245), (-h, 0, 0), 3.012, (0.0002, 0.007, -0.1), (0.0125, 0.0136, 0.0029, 0, 0.0001, 0.0003)),
        ("r_arm_ely", 1, (0, 0.187, -0.016), (0, 0, 0), 3.388, (-0.006, -0.04, -0.016), (0.0033, 0.0067, 0.0054, -0.0003, 0, -0.0002)),
        ("r_arm_elx", 0, (0, 0.119, 0.0092), (0, h, 0), 2.509, (0.0027, 0.14, 0.0153), (0.0077, 0.0024, 0.0074, 0.0004, 0, -0.0007)),
        ("r_arm_wry", 1, (0, 0.29955, -0.00921), (0, 0, 0), 0.35, (0.0, -0.015, -0.003), (0.0004, 0.0005, 0.0004, 0, 0, 0)),
        ("r_arm_wrx", 0, (0, 0, 0), (0, 0, 0), 0.35, (0.0, 0.015, 0.002), (0.0004, 0.0005, 0.0004, 0, 0, 0)),
        ("r_arm_wry2", 1, (0, 0.051, 0), (0, 0, 0), 0.6, (0.0, 0.09, 0.0), (0.0011, 0.0009, 0.0012, 0, 0, 0)),
        ("l_leg_hpz", 2, (0, 0.089, 0), (0, 0, 0), 2.409, (0.0005, -0.003, 0.032), (0.0039, 0.0046, 0.004, 0, 0, 0)),
        ("l_leg_hpx", 0, (0, 0, 0), (0, 0, 0), 2.39, (0.0045, 0.0035, 0.0377), (0.0029, 0.0031, 0.0039, 0, 0.0001, 0)),
        ("l_leg_hpy", 1, (0.05, 0.0225, -0.066), (0, 0, 0), 12.211, (0.0376, 0.0052, -0.1687), (0.331, 0.332, 0.044, -0.0003, -0.0187, 0.0019)),
        ("l_leg_kny", 1, (-0.05, 0, -0.374), (0, 0, 0), 6.5, (0.001, 0, -0.187), (0.077, 0.076, 0.01, 0, -0.0007, 0)),
        ("l_leg_aky", 1, (0, 0, -0.422), (0, 0, 0), 0.125, (-0.01, 0, 0.01), (0.0001, 0.0001, 0.0001, 0, 0, 0)),
        ("l_leg_akx", 0, (0, 0, 0), (0, 0, 0), 2.41, (0.027, 0, -0.067), (0.002, 0.007, 0.008, 0, 0.0003, 0)),
        ("r_leg_hpz", 2, (0, -0.089, 0), (0, 0, 0), 2.409, (0.0005, 0.003, 0.032), (0.0039, 0.0046, 0.004, 0, 0, 0)),
        ("r_leg_hpx", 0, (0, 0, 0), (0, 0, 0), 2.39, (0.0045, -0.0035, 0.0377), (0.0029, 0.0031, 0.0039, 0, 0.0001, 0)),
        ("r_leg_hpy", 1, (0.05, -0.0225, -0.066), (0, 0, 0), 12.211, (0.0376, -0.0052, -0.1687), (0.331, 0.332, 0.044, 0.0003, -0.0187, -0.0019)),
        ("r_leg_kny", 1, (-0.05, 0, -0.374), (0, 0, 0), 6.5, (0.001, 0, -0.187), (0.077, 0.076, 0.01, 0, -0.0007, 0)),
        ("r_leg_aky", 1, (0, 0, -0.422), (0, 0, 0), 0.125, (-0.01, 0, 0.01), (0.0001, 0.0001, 0.0001, 0, 0, 0)),
        ("r_leg_akx", 0, (0, 0, 0), (0, 0, 0), 2.41, (0.027, 0, -0.067), (0.002, 0.007, 0.008, 0, 0.0003, 0)),
    ]
    assert len(spec) == len(parent) == 30
    L = [Link(nm, parent[i], ax, xyz, rpy, m, com, inr, 0.1)
         for i, (nm, ax, xyz, rpy, m, com, inr) in enumerate(spec)]
    return Robot("atlas_like", L)


def random_tree(parent: Sequence[int], seed: int = 0, prismatic_every: int = 0,
                name: str | None = None) -> Robot:
    """Random robot on a given topology: random axes, generic (non axis-aligned) joint frames,
    random SPD inertias.  Used by tests to exercise fully dense X_tree / I patterns."""
    rng = np.random.default_rng(seed)
    L = []
    for i, p in enumerate(parent):
        A = rng.normal(size=(3, 3))
        Ic = A @ A.T * 0.01 + np.eye(3) * 0.005
        jt = "prismatic" if (prismatic_every and (i % prismatic_every) == prismatic_every - 1) \
            else "revolute"
        L.append(Link(f"b{i}", int(p), int(rng.integers(0, 3)), tuple(rng.uniform(-0.3, 0.3, 3)),
                      tuple(rng.uniform(-_PI, _PI, 3)), float(rng.uniform(0.3, 4.0)),
                      tuple(rng.uniform(-0.1, 0.1, 3)),
                      (Ic[0, 0], Ic[1, 1], Ic[2, 2], Ic[0, 1], Ic[0, 2], Ic[1, 2]),
                      float(rng.uniform(0.0, 1.0)), jt))
    return Robot(name or f"random_tree_n{len(L)}_s{seed}", L)


def floating_base_X(q6: Sequence[float]) -> np.ndarray:
    """World -> base motion transform of the 6-DoF base joint, ``q6 = (px, py, pz, rx, ry, rz)``: the
    six-joint chain Px, Py, Pz, Rx, Ry, Rz the reference describes (``RBDReference.py:634-637``), i.e.
    ``plux(Rz(rz) Ry(ry) Rx(rx), p)`` with the coordinate-transform rotations used for every joint."""
    q6 = np.asarray(q6, dtype=np.float64).reshape(6)
    E = _rot_axis(2, q6[5]) @ _rot_axis(1, q6[4]) @ _rot_axis(0, q6[3])
    return plux(E, q6[:3])


class FloatingBaseRobot:
    """Floating-base robot with the getter surface the reference's floating-base branches read
    (``RBDReference.py:585-593, 652-691, 761-779``): body 0 is the base, attached to the world by a
    6-DoF joint with ``S = eye(6)``; bodies 1.. hang off it with 1-DoF joints.

    Layout (what those branches assume): ``NB`` bodies, ``n = NB + 5`` velocities; body 0 owns
    ``q[0:6]``, ``qd[0:6]`` (``qd[0:6]`` is the base TWIST in base coordinates, because ``S`` is the
    identity), body ``i >= 1`` owns index ``i + 5``.  ``get_Xmat_Func_by_id(0)`` maps the base's six
    coordinates to the world -> base transform (`floating_base_X`).  What URDFParser does for its
    floating bases cannot be checked offline: this class pins the REFERENCE's
    floating-base arithmetic, not URDFParser's parametrisation.

    Built from a fixed-base `Robot` whose body 0 is the only root: that body becomes the base (its own
    1-DoF joint is dropped, its inertia kept)."""

    floating_base = True

    def __init__(self, inner: Robot, name: str | None = None):
        n = inner.get_num_bodies()
        if inner.get_parent_id(0) != -1 or any(inner.get_parent_id(i) == -1 for i in range(1, n)):
            raise ValueError("FloatingBaseRobot: body 0 must be the only root of the fixed-base robot")
        self.inner = inner
        self.name = name or f"fb_{inner.name}"
        self._n = n

    def get_num_bodies(self) -> int:
        return self._n

    def get_num_vel(self) -> int:
        return self._n + 5

    def get_num_pos(self) -> int:
        return self._n + 5

    def get_parent_id(self, i: int) -> int:
        return self.inner.get_parent_id(i)

    def get_S_by_id(self, i: int) -> np.ndarray:
        return np.eye(6) if i == 0 else self.inner.get_S_by_id(i)

    def _idx(self, i: int):
        return [0, 1, 2, 3, 4, 5] if i == 0 else i + 5

    get_joint_index_q = _idx
    get_joint_index_v = _idx
    get_joint_index_f = _idx

    def get_Xmat_Func_by_id(self, i: int):
        return floating_base_X if i == 0 else self.inner.get_Xmat_Func_by_id(i)

    def get_Imat_by_id(self, i: int) -> np.ndarray:
        return self.inner.get_Imat_by_id(i)

    def get_Imats_dict_by_id(self) -> Dict[int, np.ndarray]:
        return self.inner.get_Imats_dict_by_id()

    def get_subtree_by_id(self, i: int) -> List[int]:
        return self.inner.get_subtree_by_id(i)

    def get_damping_by_id(self, i: int) -> float:
        # the base's value feeds the reference's 5 x 5 damping block (RBDReference.py:1336-1339)
        return self.inner.get_damping_by_id(i)

    def __repr__(self) -> str:
        return f"FloatingBaseRobot({self.name!r}, NB={self._n}, n={self._n + 5})"


def floating_quadruped_like() -> FloatingBaseRobot:
    """13 bodies / 18 velocities: a free-floating trunk (HyQ-like mass and inertia) carrying the four
    3-joint legs of `quadruped_like`."""
    legs = quadruped_like().links
    L = [Link("trunk", -1, 2, (0, 0, 0), (0, 0, 0), 53.4, (0.03, 0.0, 0.043), (1.3, 7.0, 7.9, 0.02, -0.2, 0.0), 0.0)]
    for l in legs:
        L.append(Link(l.name, 0 if l.parent == -1 else l.parent + 1, l.axis, l.xyz, l.rpy, l.mass, l.com, l.inertia, l.damping))
    return FloatingBaseRobot(Robot("quadruped_trunk", L), "fb_quadruped_like")


BUILTIN_ROBOTS = {"iiwa_like": iiwa_like, "quadruped_like": quadruped_like,
                  "atlas_like": atlas_like}


def builtin_robot(name: str) -> Robot:
    return BUILTIN_ROBOTS[name]()
