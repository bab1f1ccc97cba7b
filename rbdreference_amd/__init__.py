"""rbdreference_amd -- MI355X-native batched rigid-body dynamics behind the RBDReference API.

    from rbdreference_amd import RBDReference, iiwa_like
    rbd = RBDReference(iiwa_like())
    c, dc_du = rbd.rnea_grad(q, qd, qdd, return_c=True)      # q, qd, qdd: [B, 7] cuda tensors
"""
from .robot import (BUILTIN_ROBOTS, FloatingBaseRobot, Link, Robot, atlas_like, builtin_robot,
                    floating_quadruped_like, iiwa_like, quadruped_like, random_tree)
from .packer import PackedModel, pack_robot
from .urdf import load_urdf, loads_urdf, to_urdf

__all__ = ["RBDReference", "Robot", "FloatingBaseRobot", "floating_quadruped_like", "Link", "iiwa_like",
           "quadruped_like", "atlas_like",
           "random_tree", "builtin_robot", "BUILTIN_ROBOTS", "pack_robot", "PackedModel",
           "load_urdf", "loads_urdf", "to_urdf"]


def __getattr__(name):
    # api.py imports torch; keep `import rbdreference_amd` light for the packer / build tools
    if name == "RBDReference":
        from .api import RBDReference
        return RBDReference
    raise AttributeError(name)
