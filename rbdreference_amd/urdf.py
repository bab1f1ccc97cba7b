"""URDF subset -> ``Robot`` (model ingestion).

The reference takes its robot from the external URDFParser package (``/root/reference/README.md:8``),
which is neither vendored nor installed here, and no URDF file ships with the reference; this loader is
therefore written against the URDF specification itself and its parity with URDFParser is UNPINNED
(DESIGN.md §2).  What it produces is a ``rbdreference_amd.robot.Robot`` -- the getter surface the hot
path reads -- so the result can be handed to the real reference and to this
package alike.  A robot that already comes from URDFParser needs none of this: ``pack_robot``
accepts any getter-compatible object.

Supported: ``<link>`` with ``<inertial>`` (origin xyz/rpy, mass, inertia), ``<joint>`` of type
revolute / continuous / prismatic / fixed with ``<origin>``, ``<axis>`` (any direction),
``<dynamics damping>``; fixed base (the root link is welded to the world).  Fixed joints are
folded: the child link's inertia is added to the body that carries it and its children are re-hung
with the composed transform.  A joint axis that is not a positive coordinate axis is handled by
rotating the body frame so that it becomes one (the hot path assumes ``S = e_k``, Appendix A).
Not supported (rejected with a ValueError): floating / planar / mimic joints, several root links,
kinematic loops, movable leaf bodies without mass.
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Tuple

import numpy as np

from .robot import Link, Robot, _rpy_E, _skew, _snap, spatial_inertia

__all__ = ["load_urdf", "loads_urdf", "to_urdf"]


def _vec(s: Optional[str], default=(0.0, 0.0, 0.0)) -> np.ndarray:
    if s is None:
        return np.array(default, dtype=np.float64)
    v = np.array([float(t) for t in s.split()], dtype=np.float64)
    if v.shape != (3,):
        raise ValueError(f"expected three numbers, got {s!r}")
    return v


def _origin(el) -> Tuple[np.ndarray, np.ndarray]:
    """(E, r) of an <origin>: E = child-frame coordinates of parent-frame vectors, r = position."""
    o = None if el is None else el.find("origin")
    if o is None:
        return np.eye(3), np.zeros(3)
    return _rpy_E(_vec(o.get("rpy"))), _vec(o.get("xyz"))


def _axis_frame(a: np.ndarray) -> Tuple[np.ndarray, int]:
    """Coordinate transform A (proper rotation) and axis index k with A a = e_k."""
    a = a / np.linalg.norm(a)
    for k in range(3):
        e = np.zeros(3); e[k] = 1.0
        if np.allclose(a, e, atol=1e-12):
            return np.eye(3), k
        if np.allclose(a, -e, atol=1e-12):          # half turn about the next axis flips k
            A = -np.eye(3); A[(k + 1) % 3, (k + 1) % 3] = 1.0
            return A, k
    # general direction: rotate a onto e_z about a x e_z (Rodrigues)
    z = np.array([0.0, 0.0, 1.0])
    w = np.cross(a, z); s = np.linalg.norm(w); c = float(a @ z)
    W = _skew(w / s)
    R = np.eye(3) + s * W + (1.0 - c) * (W @ W)     # active rotation taking a to z
    return _snap(R), 2


def loads_urdf(text: str, name: Optional[str] = None) -> Robot:
    """Parse URDF text into a fixed-base ``Robot`` (bodies numbered parents-first, depth-first in
    document order)."""
    root = ET.fromstring(text)
    if root.tag != "robot":
        raise ValueError("not a URDF: root element is not <robot>")
    rname = name or root.get("name", "urdf_robot")
    links: Dict[str, ET.Element] = {}
    for l in root.findall("link"):
        links[l.get("name")] = l
    children: Dict[str, List[ET.Element]] = {k: [] for k in links}
    child_names = set()
    for j in root.findall("joint"):
        p, c = j.find("parent").get("link"), j.find("child").get("link")
        if p not in links or c not in links:
            raise ValueError(f"joint {j.get('name')!r} refers to an unknown link")
        if c in child_names:
            raise ValueError(f"link {c!r} has two parent joints (kinematic loops are not supported)")
        if j.find("mimic") is not None:
            raise ValueError(f"joint {j.get('name')!r}: mimic joints are not supported")
        child_names.add(c)
        children[p].append(j)
    roots = [k for k in links if k not in child_names]
    if len(roots) != 1:
        raise ValueError(f"expected exactly one root link, found {roots}")

    bodies: List[dict] = []          # parent, jtype, axis, E, r, damping, name, I (6x6 accumulated)

    def add_inertia(link_el, body: int, E_cl: np.ndarray, r_cl: np.ndarray):
        """Add the link's inertia to body `body`; link coords = E_cl (x_body - r_cl)."""
        ine = link_el.find("inertial")
        if ine is None or body < 0:
            return
        m = float(ine.find("mass").get("value"))
        I = ine.find("inertia")
        Ic = np.array([[float(I.get("ixx")), float(I.get("ixy", 0)), float(I.get("ixz", 0))],
                       [float(I.get("ixy", 0)), float(I.get("iyy")), float(I.get("iyz", 0))],
                       [float(I.get("ixz", 0)), float(I.get("iyz", 0)), float(I.get("izz"))]])
        E_i, c_l = _origin(ine)                    # inertial frame in the link frame
        R_lb = E_cl.T                              # link -> body rotation of coordinates
        Ic_body = R_lb @ (E_i.T @ Ic @ E_i) @ R_lb.T
        com_body = r_cl + R_lb @ c_l
        bodies[body]["I"] += spatial_inertia(m, com_body, Ic_body)

    def walk(link_name: str, body: int, E_cl: np.ndarray, r_cl: np.ndarray):
        add_inertia(links[link_name], body, E_cl, r_cl)
        for j in children[link_name]:
            jt = j.get("type")
            E_o, r_o = _origin(j)
            # joint (= child link at q = 0) frame relative to the carrying body
            E = E_o @ E_cl
            r = r_cl + E_cl.T @ r_o
            child = j.find("child").get("link")
            if jt == "fixed":
                walk(child, body, E, r)
                continue
            if jt in ("revolute", "continuous"):
                kind = "revolute"
            elif jt == "prismatic":
                kind = "prismatic"
            else:
                raise ValueError(f"joint {j.get('name')!r}: type {jt!r} is not supported (fixed base, 1-DoF joints)")
            ax = j.find("axis")
            a = _vec(None if ax is None else ax.get("xyz"), (1.0, 0.0, 0.0))
            if np.linalg.norm(a) == 0.0:
                raise ValueError(f"joint {j.get('name')!r}: zero axis")
            A, k = _axis_frame(a)
            dyn = j.find("dynamics")
            bodies.append(dict(parent=body, jtype=kind, axis=k, E=_snap(A @ E), r=r, name=j.get("name"),
                               damping=float(dyn.get("damping", 0.0)) if dyn is not None else 0.0,
                               I=np.zeros((6, 6))))
            walk(child, len(bodies) - 1, A.T, np.zeros(3))       # child-link coords = A^T body coords

    walk(roots[0], -1, np.eye(3), np.zeros(3))
    if not bodies:
        raise ValueError("the URDF has no movable joint")
    out: List[Link] = []
    has_child = [False] * len(bodies)
    for b in bodies:
        if b["parent"] >= 0:
            has_child[b["parent"]] = True
    for i, b in enumerate(bodies):
        I = b["I"]
        m = float(I[3, 3])
        if m <= 0.0:
            if not has_child[i]:
                raise ValueError(f"joint {b['name']!r} moves a leaf body without mass (the mass matrix would be singular)")
            com = np.zeros(3); Ic = np.zeros((3, 3))
        else:
            mc = I[:3, 3:]                                        # m c^x
            com = np.array([mc[2, 1], mc[0, 2], mc[1, 0]]) / m
            C = _skew(com)
            Ic = I[:3, :3] - m * (C @ C.T)
        out.append(Link(b["name"], b["parent"], b["axis"], tuple(b["r"]), (0.0, 0.0, 0.0), m, tuple(com),
                        (Ic[0, 0], Ic[1, 1], Ic[2, 2], Ic[0, 1], Ic[0, 2], Ic[1, 2]), b["damping"], b["jtype"],
                        rot=b["E"]))
    return Robot(rname, out)


def load_urdf(path: str, name: Optional[str] = None) -> Robot:
    with open(path, "r", encoding="utf-8") as f:
        return loads_urdf(f.read(), name)


def _rpy_of_E(E: np.ndarray) -> Tuple[float, float, float]:
    """rpy with _rpy_E(rpy) == E (E = R^T, R = Rz(y) Ry(p) Rx(r))."""
    R = np.asarray(E, dtype=np.float64).T
    p = math.asin(max(-1.0, min(1.0, -R[2, 0])))
    if abs(math.cos(p)) > 1e-12:
        r = math.atan2(R[2, 1], R[2, 2]); y = math.atan2(R[1, 0], R[0, 0])
    else:                                            # gimbal lock: put everything into yaw
        r = 0.0; y = math.atan2(-R[0, 1], R[1, 1])
    return r, p, y


def to_urdf(robot: Robot) -> str:
    """Write a ``Robot`` built from ``Link`` records as URDF text (base link + one link per body)."""
    ax = ["1 0 0", "0 1 0", "0 0 1"]
    L = [f'<robot name="{robot.name}">', '  <link name="base"/>']
    for i, l in enumerate(robot.links):
        ixx, iyy, izz, ixy, ixz, iyz = (repr(float(t)) for t in l.inertia)
        L.append(f'  <link name="{l.name}"><inertial><origin xyz="{" ".join(repr(float(t)) for t in l.com)}" rpy="0 0 0"/>'
                 f'<mass value="{float(l.mass)!r}"/><inertia ixx="{ixx}" iyy="{iyy}" izz="{izz}" ixy="{ixy}" ixz="{ixz}" iyz="{iyz}"/>'
                 f'</inertial></link>')
        parent = "base" if l.parent < 0 else robot.links[l.parent].name
        rpy = l.rpy if l.rot is None else _rpy_of_E(np.asarray(l.rot))
        L.append(f'  <joint name="joint_{i}_{l.name}" type="{l.jtype}"><parent link="{parent}"/><child link="{l.name}"/>'
                 f'<origin xyz="{" ".join(repr(float(t)) for t in l.xyz)}" rpy="{" ".join(repr(float(t)) for t in rpy)}"/>'
                 f'<axis xyz="{ax[l.axis]}"/><dynamics damping="{float(l.damping)!r}"/>'
                 f'<limit lower="-3.14" upper="3.14" effort="100" velocity="10"/></joint>')
    L.append("</robot>")
    return "\n".join(L) + "\n"
