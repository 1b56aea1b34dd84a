"""Serial kinematic chain extracted from a URDF string.

Replaces, for the hot path only, the part of ``forwardkinematics``
(``GenericURDFFk(urdf, root_link, end_link, base_type)``, reference
``robotmpcs/models/mpcBase.py:46-51``) that the MPC model needs: the list of
joints between ``root_link`` and ``end_link`` with their fixed origin
transforms and axes, and ``n()`` = number of actuated joints on that chain.
The numeric constants are handed to the HIP solver in the model descriptor;
no symbolic expressions are built.
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

JOINT_FIXED = 0
JOINT_REVOLUTE = 1
JOINT_PRISMATIC = 2

_TYPE_MAP = {
    "fixed": JOINT_FIXED,
    "revolute": JOINT_REVOLUTE,
    "continuous": JOINT_REVOLUTE,
    "prismatic": JOINT_PRISMATIC,
}


def rpy_to_matrix(rpy) -> List[float]:
    """URDF fixed-axis roll/pitch/yaw -> row-major 3x3 (R = Rz(y) Ry(p) Rx(r))."""
    r, p, y = rpy
    cr, sr = math.cos(r), math.sin(r)
    cp, sp = math.cos(p), math.sin(p)
    cy, sy = math.cos(y), math.sin(y)
    return [
        cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
        sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
        -sp, cp * sr, cp * cr,
    ]


@dataclass
class ChainJoint:
    name: str
    type: int
    parent: str
    child: str
    xyz: List[float]
    rot: List[float]
    axis: List[float]
    dof: int = -1


@dataclass
class UrdfChain:
    root_link: str
    end_link: str
    joints: List[ChainJoint] = field(default_factory=list)

    def n(self) -> int:
        """Number of actuated joints on the chain (``fk.n()``)."""
        return sum(1 for j in self.joints if j.type != JOINT_FIXED)

    def frame_of(self, link: str) -> int:
        """Index of the joint whose child link is ``link``."""
        for i, j in enumerate(self.joints):
            if j.child == link:
                return i
        raise KeyError(f"link '{link}' is not on the chain {self.root_link} -> {self.end_link}")


def _floats(text: Optional[str], default) -> List[float]:
    if text is None:
        return list(default)
    return [float(v) for v in text.split()]


def parse_chain(urdf: str, root_link: str, end_link: str) -> UrdfChain:
    """Walk parent joints from ``end_link`` up to ``root_link``.

    If ``root_link`` is not a link of the URDF (the shipped
    ``pointRobotMpc.yaml`` names ``ee_link``, which ``pointRobot.urdf`` does not
    contain -- SURVEY.md 8a row A-FK) the walk continues to the URDF's absolute
    root link.
    """
    robot = ET.fromstring(urdf)
    by_child: Dict[str, ET.Element] = {}
    links = {l.get("name") for l in robot.findall("link")}
    for j in robot.findall("joint"):
        by_child[j.find("child").get("link")] = j
    if end_link not in links:
        raise KeyError(f"end_link '{end_link}' not found in URDF")
    rev: List[ChainJoint] = []
    cur = end_link
    while cur != root_link and cur in by_child:
        j = by_child[cur]
        jt = j.get("type")
        if jt not in _TYPE_MAP:
            raise ValueError(f"unsupported joint type '{jt}' on joint {j.get('name')}")
        origin = j.find("origin")
        xyz = _floats(origin.get("xyz") if origin is not None else None, (0, 0, 0))
        rpy = _floats(origin.get("rpy") if origin is not None else None, (0, 0, 0))
        axis_el = j.find("axis")
        axis = _floats(axis_el.get("xyz") if axis_el is not None else None, (1, 0, 0))
        nrm = math.sqrt(sum(a * a for a in axis))
        if _TYPE_MAP[jt] != JOINT_FIXED and nrm > 0:
            axis = [a / nrm for a in axis]
        rev.append(
            ChainJoint(
                name=j.get("name"), type=_TYPE_MAP[jt], parent=j.find("parent").get("link"),
                child=cur, xyz=xyz, rot=rpy_to_matrix(rpy), axis=axis,
            )
        )
        cur = j.find("parent").get("link")
    joints = list(reversed(rev))
    dof = 0
    for j in joints:
        if j.type != JOINT_FIXED:
            j.dof = dof
            dof += 1
    return UrdfChain(root_link=cur, end_link=end_link, joints=joints)


def fk_positions(joints, q, frames, diffdrive: bool = False):
    """Batched numpy forward kinematics, positions only.

    ``joints``: list of ``ChainJoint`` or descriptor dicts (type/dof/xyz/rot/axis);
    ``q``: (B, n); ``frames``: iterable of frame indices.  Host-side helper for
    scenario generation and debugging -- the solver's FK is the HIP device
    function in csrc/rmpc_model.hpp.
    """
    import numpy as np

    def get(j, k):
        return j[k] if isinstance(j, dict) else getattr(j, k)

    q = np.atleast_2d(np.asarray(q, dtype=float))
    B = q.shape[0]
    R = np.tile(np.eye(3), (B, 1, 1))
    o = np.zeros((B, 3))
    if diffdrive:
        c, s = np.cos(q[:, 2]), np.sin(q[:, 2])
        R[:, 0, 0] = c; R[:, 0, 1] = -s; R[:, 1, 0] = s; R[:, 1, 1] = c
        o[:, 0] = q[:, 0]; o[:, 1] = q[:, 1]
    out = {}
    frames = set(int(f) for f in frames)
    for idx, j in enumerate(joints):
        o = o + np.einsum("bij,j->bi", R, np.asarray(get(j, "xyz"), dtype=float))
        R = R @ np.asarray(get(j, "rot"), dtype=float).reshape(3, 3)
        jt, dof = get(j, "type"), get(j, "dof")
        ax = np.asarray(get(j, "axis"), dtype=float)
        if jt == JOINT_REVOLUTE:
            th = q[:, dof]
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            Rq = np.eye(3)[None] + np.sin(th)[:, None, None] * K[None] + (1 - np.cos(th))[:, None, None] * (K @ K)[None]
            R = R @ Rq
        elif jt == JOINT_PRISMATIC:
            o = o + np.einsum("bij,j->bi", R, ax) * q[:, dof][:, None]
        if idx in frames:
            out[idx] = o.copy()
    return out
