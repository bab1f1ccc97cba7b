"""URDF subset loader (rbdreference_amd/urdf.py).  No URDFParser and no URDF file
exist in the reference tree, so the loader is checked (a) by a write -> read round trip of every
built-in robot, and (b) against an INDEPENDENT forward-kinematics / energy model written here from the
URDF specification (homogeneous transforms, Rodrigues rotations): gravity torques must equal dV/dq
and the kinetic energy must equal 1/2 qd^T H qd, on a URDF with skewed axes, negative axes, fixed
joints, rotated inertial frames and a branch."""
import math
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from oracle import rbd_oracle as orc
from rbdreference_amd.packer import pack_robot
from rbdreference_amd.robot import BUILTIN_ROBOTS, random_tree
from rbdreference_amd.urdf import loads_urdf, to_urdf

URDF = """
<robot name="skewed">
  <link name="world_link"><inertial><mass value="5"/><inertia ixx="1" iyy="1" izz="1" ixy="0" ixz="0" iyz="0"/></inertial></link>
  <link name="pedestal"><inertial><origin xyz="0 0 0.2" rpy="0 0 0"/><mass value="7"/><inertia ixx="1" iyy="1" izz="1" ixy="0" ixz="0" iyz="0"/></inertial></link>
  <link name="l1"><inertial><origin xyz="0.01 0.02 0.1" rpy="0.3 -0.2 0.5"/><mass value="3"/><inertia ixx="0.05" iyy="0.04" izz="0.02" ixy="0.003" ixz="-0.002" iyz="0.001"/></inertial></link>
  <link name="flange"><inertial><origin xyz="0 0.05 0" rpy="0 0.7 0"/><mass value="0.5"/><inertia ixx="0.002" iyy="0.003" izz="0.001" ixy="0" ixz="0" iyz="0.0004"/></inertial></link>
  <link name="l2"><inertial><origin xyz="0 0 0.15" rpy="0 0 0"/><mass value="2"/><inertia ixx="0.03" iyy="0.03" izz="0.004" ixy="0" ixz="0" iyz="0"/></inertial></link>
  <link name="l3"><inertial><origin xyz="0.05 0 0" rpy="1.0 0 0"/><mass value="1.5"/><inertia ixx="0.004" iyy="0.02" izz="0.02" ixy="0.001" ixz="0" iyz="0"/></inertial></link>
  <link name="slider"><inertial><origin xyz="0 0 0.02" rpy="0 0 0"/><mass value="0.8"/><inertia ixx="0.001" iyy="0.001" izz="0.001" ixy="0" ixz="0" iyz="0"/></inertial></link>
  <link name="tool"><inertial><origin xyz="0.03 0 0.04" rpy="0.2 0.1 0"/><mass value="0.4"/><inertia ixx="0.0005" iyy="0.0006" izz="0.0003" ixy="0" ixz="0.0001" iyz="0"/></inertial></link>
  <joint name="weld" type="fixed"><parent link="world_link"/><child link="pedestal"/><origin xyz="0.1 0 0.3" rpy="0 0 0.4"/></joint>
  <joint name="j1" type="revolute"><parent link="pedestal"/><child link="l1"/><origin xyz="0 0 0.4" rpy="0.1 0.2 0.3"/><axis xyz="0 0 -1"/><dynamics damping="0.25"/></joint>
  <joint name="mount" type="fixed"><parent link="l1"/><child link="flange"/><origin xyz="0 0.1 0.2" rpy="0.5 0 0"/></joint>
  <joint name="j2" type="continuous"><parent link="flange"/><child link="l2"/><origin xyz="0 0 0.05" rpy="0 0.3 0"/><axis xyz="0.6 0 0.8"/></joint>
  <joint name="j3" type="revolute"><parent link="l2"/><child link="l3"/><origin xyz="0 0 0.3" rpy="0 0 0"/><axis xyz="0 -1 0"/></joint>
  <joint name="j4" type="prismatic"><parent link="l1"/><child link="slider"/><origin xyz="0.2 0 0.1" rpy="0 -0.4 0.2"/><axis xyz="1 1 0"/></joint>
  <joint name="tcp" type="fixed"><parent link="l3"/><child link="tool"/><origin xyz="0.1 0 0" rpy="0 0 1.2"/></joint>
</robot>
"""


def _rot(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]]); Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _T(xyz, rpy):
    T = np.eye(4); T[:3, :3] = _rot(rpy); T[:3, 3] = xyz
    return T


def _v(s, d="0 0 0"):
    return [float(t) for t in (s if s is not None else d).split()]


class IndependentModel:
    """Straight-from-the-spec URDF forward kinematics: link poses as 4x4 active transforms."""

    def __init__(self, text):
        self.root = ET.fromstring(text)
        self.joints = self.root.findall("joint")
        self.movable = [j.get("name") for j in self._dfs() if j.get("type") != "fixed"]

    def _dfs(self):
        kids = {}
        childs = set()
        for j in self.joints:
            kids.setdefault(j.find("parent").get("link"), []).append(j)
            childs.add(j.find("child").get("link"))
        root = [l.get("name") for l in self.root.findall("link") if l.get("name") not in childs][0]
        out = []

        def walk(link):
            for j in kids.get(link, []):
                out.append(j)
                walk(j.find("child").get("link"))
        walk(root)
        self.root_link = root
        return out

    def poses(self, q):
        qmap = dict(zip(self.movable, q))
        pose = {self.root_link: np.eye(4)}
        for j in self._dfs():
            o = j.find("origin")
            T = pose[j.find("parent").get("link")] @ _T(_v(o.get("xyz")), _v(o.get("rpy")))
            t = j.get("type")
            if t != "fixed":
                a = np.array(_v(j.find("axis").get("xyz") if j.find("axis") is not None else None, "1 0 0"))
                a = a / np.linalg.norm(a)
                M = np.eye(4)
                if t == "prismatic":
                    M[:3, 3] = a * qmap[j.get("name")]
                else:
                    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                    th = qmap[j.get("name")]
                    M[:3, :3] = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)
                T = T @ M
            pose[j.find("child").get("link")] = T
        return pose

    def inertials(self):
        for l in self.root.findall("link"):
            ine = l.find("inertial")
            if ine is None:
                continue
            o = ine.find("origin")
            xyz, rpy = (_v(o.get("xyz")), _v(o.get("rpy"))) if o is not None else ([0, 0, 0], [0, 0, 0])
            I = ine.find("inertia")
            Ic = np.array([[float(I.get("ixx")), float(I.get("ixy")), float(I.get("ixz"))],
                           [float(I.get("ixy")), float(I.get("iyy")), float(I.get("iyz"))],
                           [float(I.get("ixz")), float(I.get("iyz")), float(I.get("izz"))]])
            yield l.get("name"), float(ine.find("mass").get("value")), _T(xyz, rpy), Ic

    def potential(self, q, g=9.81):
        P = self.poses(q)
        return sum(m * g * (P[name] @ Ti)[2, 3] for name, m, Ti, _ in self.inertials())

    def kinetic(self, q, qd, h=1e-6):
        """1/2 sum(m |v_com|^2 + w^T Iw w) with v, w from central differences of the link poses."""
        Pp, Pm, P0 = self.poses(q + h * qd), self.poses(q - h * qd), self.poses(q)
        T = 0.0
        for name, m, Ti, Ic in self.inertials():
            Ap, Am, A0 = Pp[name] @ Ti, Pm[name] @ Ti, P0[name] @ Ti
            v = (Ap[:3, 3] - Am[:3, 3]) / (2 * h)
            Rd = (Ap[:3, :3] - Am[:3, :3]) / (2 * h)
            W = Rd @ A0[:3, :3].T
            w = np.array([W[2, 1], W[0, 2], W[1, 0]])
            Iw = A0[:3, :3] @ Ic @ A0[:3, :3].T
            T += 0.5 * (m * v @ v + w @ Iw @ w)
        return T


@pytest.mark.parametrize("name", list(BUILTIN_ROBOTS) + ["random_tree_n9", "random_prismatic_n6"])
def test_write_read_round_trip_packs_to_the_same_model(name):
    if name in BUILTIN_ROBOTS:
        robot = BUILTIN_ROBOTS[name]()
    elif name == "random_tree_n9":
        robot = random_tree([-1, 0, 1, 1, 3, -1, 5, 5, 7], seed=7, name=name)
    else:
        robot = random_tree([-1, 0, 1, 2, 2, 4], seed=11, prismatic_every=3, name=name)
    back = loads_urdf(to_urdf(robot))
    m1, m2 = pack_robot(robot), pack_robot(back)
    assert m1.parent == m2.parent and m1.jtype == m2.jtype and m1.axis == m2.axis
    assert np.abs(np.array(m1.Xtree) - np.array(m2.Xtree)).max() < 1e-12
    assert np.abs(np.array(m1.I) - np.array(m2.I)).max() < 1e-12
    assert np.allclose(m1.damping, m2.damping) and m1.hash == m2.hash


def test_structure_of_the_skewed_urdf():
    robot = loads_urdf(URDF)
    m = pack_robot(robot)                                  # validates S, X(q) = X_J(q) X(0), symmetry
    assert m.n == 4
    assert [l.name for l in robot.links] == ["j1", "j2", "j3", "j4"]       # depth-first, document order
    assert m.parent == [-1, 0, 1, 0]
    assert m.jtype == [0, 0, 0, 1]
    assert robot.get_damping_by_id(0) == 0.25
    # fixed links are folded into their carriers: l1 + flange, l3 + tool; world_link/pedestal are base
    assert abs(robot.get_Imat_by_id(0)[3, 3] - 3.5) < 1e-12 and abs(robot.get_Imat_by_id(2)[3, 3] - 1.9) < 1e-12


def test_gravity_torques_and_kinetic_energy_match_an_independent_model():
    robot = loads_urdf(URDF)
    om = orc.model_from_robot(robot)
    ind = IndependentModel(URDF)
    assert ind.movable == ["j1", "j2", "j3", "j4"]
    rng = np.random.default_rng(3)
    for _ in range(4):
        q = rng.uniform(-2.5, 2.5, 4); qd = rng.uniform(-1, 1, 4)
        # gravity compensation torque = dV/dq (RBDReference.rnea with qd = qdd = 0, a0 = +g z, :565-566)
        c = orc.rnea(om, q, np.zeros(4), np.zeros(4))[0]
        h = 1e-6
        dV = np.array([(ind.potential(q + h * e) - ind.potential(q - h * e)) / (2 * h) for e in np.eye(4)])
        assert np.abs(c - dV).max() < 1e-6 * max(1.0, np.abs(dV).max()), (c, dV)
        # kinetic energy: 1/2 qd^T H(q) qd with H from crba (:1091-1124)
        H = orc.crba(om, q)
        assert abs(0.5 * qd @ H @ qd - ind.kinetic(q, qd)) < 1e-6 * max(1.0, ind.kinetic(q, qd))


def test_rejections():
    with pytest.raises(ValueError, match="not supported"):
        loads_urdf(URDF.replace('type="continuous"', 'type="floating"'))
    with pytest.raises(ValueError, match="without mass"):
        loads_urdf(URDF.replace('<mass value="0.8"/>', '<mass value="0"/>'))
    with pytest.raises(ValueError, match="one root link"):
        loads_urdf(URDF.replace('<joint name="weld" type="fixed"><parent link="world_link"/><child link="pedestal"/>'
                                '<origin xyz="0.1 0 0.3" rpy="0 0 0.4"/></joint>', ""))
    with pytest.raises(ValueError, match="two parent joints"):
        loads_urdf(URDF.replace("</robot>", '<joint name="loop" type="fixed"><parent link="l2"/><child link="slider"/></joint></robot>'))


@pytest.mark.gpu
def test_urdf_robot_runs_through_the_hip_path():
    """A robot straight from URDF text, first use end to end: the model-handle library answers rnea_grad / minv at
    once while the robot's own library compiles in the background, `aba` (not served there) waits for its family
    library, and after the hand-over the specialised kernels give the same numbers -- all against the oracle."""
    import torch
    from rbdreference_amd import RBDReference
    robot = loads_urdf(URDF)
    rbd = RBDReference(robot)                      # returns at once; compiles the library for this robot in the background
    om = orc.model_from_robot(robot)
    rng = np.random.default_rng(9)
    q, qd, qdd = rng.uniform(-3, 3, (50, 4)), rng.uniform(-1, 1, (50, 4)), rng.uniform(-1, 1, (50, 4))
    tq, tqd, tqdd = (torch.tensor(x, device="cuda:0") for x in (q, qd, qdd))
    c, dc = rbd.rnea_grad(tq, tqd, tqdd, return_c=True)
    c_ref, dc_ref = orc.rnea_grad(om, q, qd, qdd, return_c=True)
    assert np.abs(c.cpu().numpy() - c_ref).max() < 1e-10 * np.abs(c_ref).max()
    assert np.abs(dc.cpu().numpy() - dc_ref).max() < 1e-10 * np.abs(dc_ref).max()
    assert np.abs(rbd.minv(tq).cpu().numpy() - orc.minv(om, q)).max() < 1e-9 * np.abs(orc.minv(om, q)).max()
    assert np.abs(rbd.aba(tq, tqd, tqdd).cpu().numpy() - orc.aba(om, q, qd, qdd)).max() < 1e-8 * np.abs(orc.aba(om, q, qd, qdd)).max()
    rbd._lib.wait_specialized()                    # hand-over: from here on the robot's own kernels
    c2, dc2 = rbd.rnea_grad(tq, tqd, tqdd, return_c=True)
    assert not rbd._lib.served_by_generic() and rbd._lib.kernel_name(1, 8, 50).startswith("rnea_grad")
    assert np.abs(dc2.cpu().numpy() - dc_ref).max() < 1e-10 * np.abs(dc_ref).max()
    assert np.abs(c2.cpu().numpy() - c_ref).max() < 1e-10 * np.abs(c_ref).max()
    assert np.abs(rbd.minv(tq).cpu().numpy() - orc.minv(om, q)).max() < 1e-9 * np.abs(orc.minv(om, q)).max()
