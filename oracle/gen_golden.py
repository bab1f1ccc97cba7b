#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (runs only in the build container).

The reference (`/root/reference/RBDReference.py`) is imported by path, fed this package's `Robot`
objects (the same objects the HIP path packs), and every pass of the rnea / rnea_grad / minv path
is run one configuration at a time, exactly as a user of the reference would.  Inputs and outputs
are stored as small float64 fixtures; the reference itself never leaves this container.

    python oracle/gen_golden.py [robot ...]   # rewrites tests/golden/golden_<robot>.npz (all by default)

Fixture contents (S = number of samples, n = DoF):
    q, qd, qdd                      [S, n]      inputs   (numpy default_rng(seed))
    fpass_v / fpass_a / fpass_f     [S, 6, n]   rnea_fpass            (RBDReference.py:559)
    c, f_acc                        [S, n], [S, 6, n]  rnea_bpass / rnea (:600, :623)
    c_noqdd                         [S, n]      rnea(q, qd)  (qdd=None, :589)
    dq_dv / dq_da / dq_df           [S, 6, n, n] rnea_grad_fpass_dq   (:1127)
    dqd_dv / dqd_da / dqd_df        [S, 6, n, n] rnea_grad_fpass_dqd  (:1189)
    dc_dq, dc_dqd, dc_dqd_damped    [S, n, n]   bpasses               (:1257, :1299)
    dc_du, dc_du_damped, dc_du_noqdd [S, n, 2n] rnea_grad             (:1345)
    mb_Minv, mb_F, mb_U, mb_Dinv    minv_bpass outputs                (:630)
    Minv_dense, Minv_upper          [S, n, n]   minv(q, True/False)   (:785)
    H                               [S, n, n]   crba(q) witness       (:1029)
    fd_qdd, fd_dq, fd_dqd           forward_dynamics / _grad          (:1371, :1376)
    aba_qdd                         [S, n]      aba(q, qd, tau=qdd)   (:817, fixed-base branch :940-1024), run
                                    on a view of the robot whose get_Imat_by_id returns np.matrix: the
                                    branch's ``np.matmul(temp, v)[0]`` (:984) is only the full bias force
                                    for that type (with ndarray inertias it degenerates to one scalar).
"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from RBDReference import RBDReference as RefRBD  # noqa: E402  (the real reference)
from rbdreference_amd.robot import BUILTIN_ROBOTS, FloatingBaseRobot, floating_quadruped_like, random_tree  # noqa: E402

N_SAMPLES = 16


def sample_inputs(n, seed, S=N_SAMPLES):
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (S, n))
    qd = rng.uniform(-1.0, 1.0, (S, n))
    qdd = rng.uniform(-1.0, 1.0, (S, n))
    return q, qd, qdd


class MatrixInertiaView:
    """The same robot with np.matrix inertias, for ``aba`` only (see the module docstring)."""

    def __init__(self, robot):
        self._robot = robot

    def __getattr__(self, name):
        return getattr(self._robot, name)

    def get_Imat_by_id(self, i):
        return np.matrix(self._robot.get_Imat_by_id(i))


def run_reference(robot, q, qd, qdd):
    ref = RefRBD(robot)
    ref_aba = RefRBD(MatrixInertiaView(robot))
    out = {k: [] for k in (
        "fpass_v fpass_a fpass_f c f_acc c_noqdd dq_dv dq_da dq_df dqd_dv dqd_da dqd_df dc_dq "
        "dc_dqd dc_dqd_damped dc_du dc_du_damped dc_du_noqdd mb_Minv mb_F mb_U mb_Dinv "
        "Minv_dense Minv_upper H fd_qdd fd_dq fd_dqd aba_qdd").split()}
    for s in range(q.shape[0]):
        qs, qds, qdds = q[s].copy(), qd[s].copy(), qdd[s].copy()
        v, a, f = ref.rnea_fpass(qs, qds, qdds)
        out["fpass_v"].append(v.copy()); out["fpass_a"].append(a.copy()); out["fpass_f"].append(f.copy())
        c, f_acc = ref.rnea_bpass(qs, f.copy())
        out["c"].append(c.copy()); out["f_acc"].append(f_acc.copy())
        c2, v2, a2, f2 = ref.rnea(qs, qds, qdds)
        assert np.array_equal(c2, c) and np.array_equal(f2, f_acc)
        out["c_noqdd"].append(ref.rnea(qs, qds)[0].copy())
        dv, da, df = ref.rnea_grad_fpass_dq(qs, qds, v2, a2)
        out["dq_dv"].append(dv.copy()); out["dq_da"].append(da.copy()); out["dq_df"].append(df.copy())
        dv2, da2, df2 = ref.rnea_grad_fpass_dqd(qs, qds, v2)
        out["dqd_dv"].append(dv2.copy()); out["dqd_da"].append(da2.copy()); out["dqd_df"].append(df2.copy())
        out["dc_dq"].append(ref.rnea_grad_bpass_dq(qs, f2, df.copy()).copy())
        out["dc_dqd"].append(ref.rnea_grad_bpass_dqd(qs, df2.copy()).copy())
        out["dc_dqd_damped"].append(ref.rnea_grad_bpass_dqd(qs, df2.copy(), True).copy())
        out["dc_du"].append(ref.rnea_grad(qs, qds, qdds).copy())
        out["dc_du_damped"].append(ref.rnea_grad(qs, qds, qdds, USE_VELOCITY_DAMPING=True).copy())
        out["dc_du_noqdd"].append(ref.rnea_grad(qs, qds).copy())
        Mb, F, U, D = ref.minv_bpass(qs)
        out["mb_Minv"].append(Mb.copy()); out["mb_F"].append(F.copy())
        out["mb_U"].append(U.copy()); out["mb_Dinv"].append(D.copy())
        out["Minv_dense"].append(ref.minv(qs, True).copy())
        out["Minv_upper"].append(ref.minv(qs, False).copy())
        out["H"].append(ref.crba(qs).copy())
        u = qdds  # any torque vector will do
        out["fd_qdd"].append(np.asarray(ref.forward_dynamics(qs, qds, u)).copy())
        a1, a2_ = ref.forward_dynamics_grad(qs, qds, u)
        out["fd_dq"].append(np.asarray(a1).copy()); out["fd_dqd"].append(np.asarray(a2_).copy())
        out["aba_qdd"].append(np.asarray(ref_aba.aba(qs, qds, u)).copy())
    return {k: np.stack(vv) for k, vv in out.items()}


def fb_robots():
    """Floating-base fixtures: the reference's floating-base branches of rnea / minv /
    forward_dynamics on duck-typed robots -- a trunk with four legs, and a 6-body tree with dense frames."""
    return [("fb_quadruped_like", floating_quadruped_like(), 301),
            ("fb_random_tree_n6", FloatingBaseRobot(random_tree([-1, 0, 1, 0, 3, 3], seed=5, name="t6"), "fb_random_tree_n6"), 302),
            # fewer than six bodies: the reference's rnea_grad raises (:1168), recorded in reference_raises
            ("fb_random_tree_n4", FloatingBaseRobot(random_tree([-1, 0, 1, 1], seed=9, name="t4"), "fb_random_tree_n4"), 303)]


def run_reference_fb(robot, q, qd, qdd):
    """Only what the reference can do with a floating base; where it raises, the fixture records the
    exception and the line (documentation, not a test vector)."""
    import traceback
    ref = RefRBD(robot)
    out = {k: [] for k in "fpass_v fpass_a fpass_f c f_acc c_noqdd mb_Minv mb_F mb_U mb_Dinv Minv_dense Minv_upper fd_qdd".split()}
    for s in range(q.shape[0]):
        qs, qds, qdds = q[s].copy(), qd[s].copy(), qdd[s].copy()
        v, a, f = ref.rnea_fpass(qs, qds, qdds)
        out["fpass_v"].append(v.copy()); out["fpass_a"].append(a.copy()); out["fpass_f"].append(f.copy())
        c, f_acc = ref.rnea_bpass(qs, f.copy())
        out["c"].append(c.copy()); out["f_acc"].append(f_acc.copy())
        out["c_noqdd"].append(ref.rnea(qs, qds)[0].copy())
        Mb, F, U, D = ref.minv_bpass(qs)
        out["mb_Minv"].append(Mb.copy()); out["mb_F"].append(F.copy()); out["mb_U"].append(U.copy()); out["mb_Dinv"].append(D.copy())
        out["Minv_dense"].append(ref.minv(qs, True).copy()); out["Minv_upper"].append(ref.minv(qs, False).copy())
        out["fd_qdd"].append(np.asarray(ref.forward_dynamics(qs, qds, qdds)).copy())
        if robot.get_num_bodies() >= 6:      # below, the reference's floating-base rnea_grad raises IndexError (:1168)
            out.setdefault("dc_du", []).append(ref.rnea_grad(qs, qds, qdds).copy())
            out.setdefault("dc_du_noqdd", []).append(ref.rnea_grad(qs, qds).copy())
            out.setdefault("dc_du_damped", []).append(ref.rnea_grad(qs, qds, qdds, USE_VELOCITY_DAMPING=True).copy())
            # the four gradient passes (README.md:19) and forward_dynamics_grad, which the reference also runs here
            c2, v2, a2, f2 = ref.rnea(qs, qds, qdds)
            dv, da, df = ref.rnea_grad_fpass_dq(qs, qds, v2, a2)
            out.setdefault("dq_dv", []).append(dv.copy()); out.setdefault("dq_da", []).append(da.copy()); out.setdefault("dq_df", []).append(df.copy())
            dv2, da2, df2 = ref.rnea_grad_fpass_dqd(qs, qds, v2)
            out.setdefault("dqd_dv", []).append(dv2.copy()); out.setdefault("dqd_da", []).append(da2.copy()); out.setdefault("dqd_df", []).append(df2.copy())
            dfm = df.copy()
            out.setdefault("dc_dq", []).append(ref.rnea_grad_bpass_dq(qs, f2, dfm).copy())
            out.setdefault("dq_df_after", []).append(dfm.copy())                  # the pass mutates df (:1291-1294)
            dfm2 = df2.copy()
            out.setdefault("dc_dqd", []).append(ref.rnea_grad_bpass_dqd(qs, dfm2).copy())
            out.setdefault("dqd_df_after", []).append(dfm2.copy())
            out.setdefault("dc_dqd_damped", []).append(ref.rnea_grad_bpass_dqd(qs, df2.copy(), True).copy())
            a1, a2_ = ref.forward_dynamics_grad(qs, qds, qdds)
            out.setdefault("fd_dq", []).append(np.asarray(a1).copy()); out.setdefault("fd_dqd", []).append(np.asarray(a2_).copy())
    data = {k: np.stack(vv) for k, vv in out.items()}
    raises = []
    for nm, fn in (("rnea_grad", lambda: ref.rnea_grad(q[0], qd[0], qdd[0])), ("crba", lambda: ref.crba(q[0])),
                   ("aba", lambda: ref.aba(q[0], qd[0], qdd[0])),
                   ("forward_dynamics_grad", lambda: ref.forward_dynamics_grad(q[0], qd[0], qdd[0]))):
        try:
            fn()
            raises.append(f"{nm}: ran")
        except Exception as e:      # noqa: BLE001  (recording whatever the reference does)
            ln = traceback.extract_tb(sys.exc_info()[2])[-1].lineno
            raises.append(f"{nm}: {type(e).__name__} at RBDReference.py:{ln}")
    data["reference_raises"] = np.array(raises)
    return data


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    robots = [(nm, mk(), 100 + k) for k, (nm, mk) in enumerate(BUILTIN_ROBOTS.items())]
    # extra fixtures with generic (dense) joint frames: a tree, a chain, a prismatic joint mix
    robots.append(("random_tree_n9", random_tree([-1, 0, 1, 1, 3, -1, 5, 5, 7], seed=7,
                                                  name="random_tree_n9"), 201))
    robots.append(("random_chain_n7", random_tree([-1, 0, 1, 2, 3, 4, 5], seed=21,
                                                   name="random_chain_n7"), 203))
    robots.append(("random_prismatic_n6", random_tree([-1, 0, 1, 2, 2, 4], seed=11,
                                                       prismatic_every=3,
                                                       name="random_prismatic_n6"), 202))
    # two roots whose subtrees INTERLEAVE in the numbering (0: {0,2,4,5,7}, 1: {1,3,6}) and a branch
    robots.append(("random_forest_n8", random_tree([-1, -1, 0, 1, 2, 0, 3, 5], seed=33,
                                                    name="random_forest_n8"), 204))
    # one root, a two-body stem and three four-body limbs under its second body (segment-wave kernels)
    robots.append(("random_limbs_n14", random_tree([-1, 0, 1, 2, 3, 4, 1, 6, 7, 8, 1, 10, 11, 12], seed=41,
                                                    name="random_limbs_n14"), 205))
    # two independent nine-body chains: a multi-root robot too deep / big for the register plans, i.e. the fp64 workspace
    # tree kernel on its single-wave (one block per root) layout (the blocks of the two roots shared a region)
    robots.append(("random_twochains_n18", random_tree([-1, 0, 1, 2, 3, 4, 5, 6, 7, -1, 9, 10, 11, 12, 13, 14, 15, 16], seed=57,
                                                        name="random_twochains_n18"), 206))
    only = set(sys.argv[1:])
    for nm, robot, seed in fb_robots():
        if only and nm not in only:
            continue
        n = robot.get_num_vel()
        q, qd, qdd = sample_inputs(n, seed, 8)
        data = run_reference_fb(robot, q, qd, qdd)
        data.update(q=q, qd=qd, qdd=qdd, seed=np.int64(seed),
                    parent=np.array([robot.get_parent_id(i) for i in range(robot.get_num_bodies())], dtype=np.int64))
        path = os.path.join(outdir, f"golden_{nm}.npz")
        np.savez_compressed(path, **data)
        print(f"{path}: nb={robot.get_num_bodies()} n={n} samples=8 size={os.path.getsize(path)} B  {list(data['reference_raises'])}")
    if only:
        robots = [r for r in robots if r[0] in only]
    for nm, robot, seed in robots:
        n = robot.get_num_bodies()
        S = N_SAMPLES if n <= 12 else (8 if n != 18 else 4)
        q, qd, qdd = sample_inputs(n, seed, S)
        data = run_reference(robot, q, qd, qdd)
        data.update(q=q, qd=qd, qdd=qdd, seed=np.int64(seed),
                    parent=np.array([robot.get_parent_id(i) for i in range(n)], dtype=np.int64))
        path = os.path.join(outdir, f"golden_{nm}.npz")
        np.savez_compressed(path, **data)
        print(f"{path}: n={n} samples={S} size={os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
