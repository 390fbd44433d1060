"""Kinematic-tree model for the regressor path (host side, no Pinocchio).

The reference obtains its model from Pinocchio's URDF parser
(``src/figaroh/tools/robot.py:52-58``) and only ever reads a handful of
attributes from it on the hot path (``src/figaroh/tools/regressor.py:36-42``,
``src/figaroh/tools/robot.py:102-119``): ``nq nv njoints inertias names
joints[j].idx_q/idx_v``.  This module builds the same flattened tree from a
URDF with the conventions the reference's committed artefacts depend on
(SURVEY.md Appendix A.2):

* joints are numbered depth-first, pre-order, children of a link visited in
  ascending *joint-name* order; joint 0 is the ``universe``;
* a fixed joint creates no model joint -- the child link's inertia is moved
  into the supporting joint's frame and added there, placements accumulate;
* without a free-flyer the root link's inertia lands on joint 0;
  ``root_joint=True`` inserts joint 1 ``root_joint`` (free-flyer, nq=7, nv=6);
* ``continuous`` joints are unbounded revolutes: nq=2 (cos, sin), nv=1;
* rpy -> R = Rz(yaw) Ry(pitch) Rx(roll); gravity (0, 0, -9.81).

The flattened form (``Model.to_flat`` / ``Model.from_flat``) is what the HIP
library receives through ``figh_model_create`` and what ships under
``figaroh_plus_amd/models/*.json`` for the five BASELINE.json robots, so the
GPU box never needs the URDF files.
"""
from __future__ import annotations

import json
import math
import xml.etree.ElementTree as ET

import numpy as np

# joint type codes shared with include/figh.h
JT_REVOLUTE = 0      # nq=1 nv=1, rotation about unit `axis`
JT_PRISMATIC = 1     # nq=1 nv=1, translation along unit `axis`
JT_CONTINUOUS = 2    # nq=2 (cos, sin) nv=1, rotation about `axis`
JT_FREEFLYER = 3     # nq=7 (p, qx qy qz qw) nv=6 (local lin, ang)
JT_UNIVERSE = -1

_NQ = {JT_REVOLUTE: 1, JT_PRISMATIC: 1, JT_CONTINUOUS: 2, JT_FREEFLYER: 7, JT_UNIVERSE: 0}
_NV = {JT_REVOLUTE: 1, JT_PRISMATIC: 1, JT_CONTINUOUS: 1, JT_FREEFLYER: 6, JT_UNIVERSE: 0}
_URDF_TYPES = {"revolute": JT_REVOLUTE, "prismatic": JT_PRISMATIC, "continuous": JT_CONTINUOUS}


def _skew(c):
    return np.array([[0.0, -c[2], c[1]], [c[2], 0.0, -c[0]], [-c[1], c[0], 0.0]])


def rpy_to_matrix(roll, pitch, yaw):
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr],
    ])


class SE3:
    """Placement (rotation, translation): child-frame coords -> parent-frame coords."""

    __slots__ = ("rotation", "translation")

    def __init__(self, rotation=None, translation=None):
        self.rotation = np.eye(3) if rotation is None else np.array(rotation, dtype=float)
        self.translation = np.zeros(3) if translation is None else np.array(translation, dtype=float)

    def __mul__(self, other):
        return SE3(self.rotation @ other.rotation, self.rotation @ other.translation + self.translation)

    def copy(self):
        return SE3(self.rotation.copy(), self.translation.copy())


class Inertia:
    """Spatial inertia: mass, lever (centre of mass) and 3x3 inertia about the centre of mass."""

    __slots__ = ("mass", "lever", "inertia")

    def __init__(self, mass=0.0, lever=None, inertia=None):
        self.mass = float(mass)
        self.lever = np.zeros(3) if lever is None else np.array(lever, dtype=float)
        self.inertia = np.zeros((3, 3)) if inertia is None else np.array(inertia, dtype=float)

    def se3_action(self, M):
        """Same body expressed in the parent frame of placement ``M``."""
        R = M.rotation
        return Inertia(self.mass, R @ self.lever + M.translation, R @ self.inertia @ R.T)

    def __add__(self, other):
        m = self.mass + other.mass
        if m == 0.0:
            return Inertia()
        c = (self.mass * self.lever + other.mass * other.lever) / m
        I = np.zeros((3, 3))
        for b in (self, other):
            d = _skew(b.lever - c)
            I += b.inertia + b.mass * (d.T @ d)
        return Inertia(m, c, I)

    def toDynamicParameters(self):
        """[m, m*c, Ixx, Ixy, Iyy, Ixz, Iyz, Izz] with I about the frame origin
        (the order consumed at ``src/figaroh/tools/robot.py:108-119``)."""
        s = _skew(self.lever)
        Io = self.inertia + self.mass * (s.T @ s)
        mc = self.mass * self.lever
        return np.array([self.mass, mc[0], mc[1], mc[2],
                         Io[0, 0], Io[0, 1], Io[1, 1], Io[0, 2], Io[1, 2], Io[2, 2]])

    def matrix(self):
        """6x6 spatial inertia, (linear, angular) ordering."""
        s = _skew(self.lever)
        Io = self.inertia + self.mass * (s.T @ s)
        M = np.zeros((6, 6))
        M[:3, :3] = self.mass * np.eye(3)
        M[:3, 3:] = -self.mass * s
        M[3:, :3] = self.mass * s
        M[3:, 3:] = Io
        return M


class _InertiaVector(list):
    """List with the ``.tolist()`` the reference calls (``regressor.py:36-39``)."""

    def tolist(self):
        return list(self)


class JointModel:
    __slots__ = ("id", "idx_q", "idx_v", "nq", "nv", "jtype", "axis")

    def __init__(self, jid, jtype, axis, idx_q, idx_v):
        self.id, self.jtype, self.axis = jid, jtype, axis
        self.idx_q, self.idx_v = idx_q, idx_v
        self.nq, self.nv = _NQ[jtype], _NV[jtype]

    def shortname(self):
        """Pinocchio's joint model names: axis-aligned joints carry their axis (JointModelRZ, JointModelPX,
        JointModelRUBY ...), anything else is the ...Unaligned variant."""
        if self.jtype == JT_FREEFLYER:
            return "JointModelFreeFlyer"
        if self.jtype == JT_UNIVERSE:
            return "JointModelUniverse"
        a = np.asarray(self.axis, dtype=np.float64)
        letter = None
        for k, name in enumerate("XYZ"):
            e = np.zeros(3)
            e[k] = 1.0
            if np.array_equal(a, e):
                letter = name
        if letter is None:
            return {JT_REVOLUTE: "JointModelRevoluteUnaligned", JT_PRISMATIC: "JointModelPrismaticUnaligned",
                    JT_CONTINUOUS: "JointModelRevoluteUnboundedUnaligned"}[self.jtype]
        return "JointModel" + {JT_REVOLUTE: "R", JT_PRISMATIC: "P", JT_CONTINUOUS: "RUB"}[self.jtype] + letter


class Data:
    """Placeholder for Pinocchio's scratch ``Data``; the HIP path keeps no host scratch."""

    def __init__(self, model):
        self.oMi = [SE3() for _ in range(model.njoints)]


class Model:
    def __init__(self, name="robot"):
        self.name = name
        self.names = ["universe"]
        self.parents = [0]
        self.jointPlacements = [SE3()]
        self.inertias = _InertiaVector([Inertia()])
        self.joints = [JointModel(0, JT_UNIVERSE, np.zeros(3), 0, 0)]
        self.gravity = np.array([0.0, 0.0, -9.81])
        self.nq = 0
        self.nv = 0
        self._lower, self._upper, self._vel, self._eff = [], [], [], []
        self.lowerPositionLimit = np.zeros(0)
        self.upperPositionLimit = np.zeros(0)
        self.velocityLimit = np.zeros(0)
        self.effortLimit = np.zeros(0)

    # ---------------------------------------------------------------- construction
    @property
    def njoints(self):
        return len(self.names)

    def add_joint(self, parent, jtype, axis, placement, name, limits=None):
        jid = len(self.names)
        axis = np.zeros(3) if axis is None else np.array(axis, dtype=float)
        self.names.append(name)
        self.parents.append(int(parent))
        self.jointPlacements.append(placement.copy())
        self.inertias.append(Inertia())
        self.joints.append(JointModel(jid, jtype, axis, self.nq, self.nv))
        nq, nv = _NQ[jtype], _NV[jtype]
        lo, up, vel, eff = limits if limits is not None else (-math.inf, math.inf, math.inf, math.inf)
        if jtype == JT_CONTINUOUS:
            self._lower += [-1.01, -1.01]
            self._upper += [1.01, 1.01]
        elif jtype == JT_FREEFLYER:
            self._lower += [-math.inf] * 3 + [-1.01] * 4
            self._upper += [math.inf] * 3 + [1.01] * 4
        else:
            self._lower += [lo] * nq
            self._upper += [up] * nq
        self._vel += [vel] * nv
        self._eff += [eff] * nv
        self.nq += nq
        self.nv += nv
        self._sync_limits()
        return jid

    def _sync_limits(self):
        self.lowerPositionLimit = np.array(self._lower, dtype=float)
        self.upperPositionLimit = np.array(self._upper, dtype=float)
        self.velocityLimit = np.array(self._vel, dtype=float)
        self.effortLimit = np.array(self._eff, dtype=float)

    def append_body(self, jid, inertia, placement):
        self.inertias[jid] = self.inertias[jid] + inertia.se3_action(placement)

    # ---------------------------------------------------------------- queries
    def getJointId(self, name):
        return self.names.index(name) if name in self.names else self.njoints

    def existJointName(self, name):
        return name in self.names

    def createData(self):
        return Data(self)

    def neutral(self):
        q = np.zeros(self.nq)
        for j in self.joints[1:]:
            if j.jtype == JT_CONTINUOUS:
                q[j.idx_q] = 1.0
            elif j.jtype == JT_FREEFLYER:
                q[j.idx_q + 6] = 1.0
        return q

    def depth(self, jid):
        d = 0
        while jid > 0:
            jid = self.parents[jid]
            d += 1
        return d

    # ---------------------------------------------------------------- flattened form
    def to_flat(self):
        """Plain arrays in the exact argument order of ``figh_model_create`` (include/figh.h)."""
        n = self.njoints
        placement = np.zeros((n, 12))
        axis = np.zeros((n, 3))
        for i in range(n):
            placement[i, :9] = self.jointPlacements[i].rotation.reshape(9)
            placement[i, 9:] = self.jointPlacements[i].translation
            axis[i] = self.joints[i].axis
        return {
            "name": self.name,
            "njoints": n,
            "nq": self.nq,
            "nv": self.nv,
            "names": list(self.names),
            "parents": np.array(self.parents, dtype=np.int32),
            "jtype": np.array([j.jtype for j in self.joints], dtype=np.int32),
            "axis": axis,
            "placement": placement,
            "idx_q": np.array([j.idx_q for j in self.joints], dtype=np.int32),
            "idx_v": np.array([j.idx_v for j in self.joints], dtype=np.int32),
            "gravity": self.gravity.copy(),
            "mass": np.array([Y.mass for Y in self.inertias]),
            "lever": np.array([Y.lever for Y in self.inertias]),
            "inertia": np.array([Y.inertia.reshape(9) for Y in self.inertias]),
            "lower": self.lowerPositionLimit.copy(),
            "upper": self.upperPositionLimit.copy(),
            "velocity": self.velocityLimit.copy(),
            "effort": self.effortLimit.copy(),
        }

    def save_flat(self, path):
        flat = self.to_flat()

        def enc(x):
            if isinstance(x, np.ndarray):
                return [enc(v) for v in x.tolist()]
            if isinstance(x, list):
                return [enc(v) for v in x]
            if isinstance(x, float) and math.isinf(x):
                return "inf" if x > 0 else "-inf"
            return x

        with open(path, "w") as f:
            json.dump({k: enc(v) for k, v in flat.items()}, f)

    @classmethod
    def from_flat(cls, flat):
        if isinstance(flat, str):
            with open(flat) as f:
                flat = json.load(f)

        def dec(x):
            return np.array([[float(v) for v in row] if isinstance(row, list) else float(row) for row in x],
                            dtype=float)

        m = cls(flat["name"])
        n = int(flat["njoints"])
        placement = dec(flat["placement"]).reshape(n, 12)
        axis = dec(flat["axis"]).reshape(n, 3)
        m.gravity = dec(flat["gravity"])
        for i in range(1, n):
            m.names.append(flat["names"][i])
            m.parents.append(int(flat["parents"][i]))
            m.jointPlacements.append(SE3(placement[i, :9].reshape(3, 3), placement[i, 9:]))
            m.inertias.append(Inertia())
            m.joints.append(JointModel(i, int(flat["jtype"][i]), axis[i],
                                       int(flat["idx_q"][i]), int(flat["idx_v"][i])))
        mass, lever, inertia = dec(flat["mass"]), dec(flat["lever"]).reshape(n, 3), dec(flat["inertia"]).reshape(n, 9)
        for i in range(n):
            m.inertias[i] = Inertia(mass[i], lever[i], inertia[i].reshape(3, 3))
        m.nq, m.nv = int(flat["nq"]), int(flat["nv"])
        m._lower, m._upper = list(dec(flat["lower"])), list(dec(flat["upper"]))
        m._vel, m._eff = list(dec(flat["velocity"])), list(dec(flat["effort"]))
        m._sync_limits()
        return m


# -------------------------------------------------------------------- URDF
def _floats(text, n, default):
    if text is None:
        return np.array(default, dtype=float)
    vals = [float(x) for x in text.split()]
    assert len(vals) == n, text
    return np.array(vals)


def _origin(el):
    o = el.find("origin") if el is not None else None
    if o is None:
        return SE3()
    xyz = _floats(o.get("xyz"), 3, [0, 0, 0])
    rpy = _floats(o.get("rpy"), 3, [0, 0, 0])
    return SE3(rpy_to_matrix(*rpy), xyz)


def _link_inertia(link):
    ine = link.find("inertial")
    if ine is None:
        return None
    M = _origin(ine)
    mass = float(ine.find("mass").get("value"))
    t = ine.find("inertia")
    g = (lambda k: float(t.get(k, 0.0))) if t is not None else (lambda k: 0.0)
    I = np.array([[g("ixx"), g("ixy"), g("ixz")],
                  [g("ixy"), g("iyy"), g("iyz")],
                  [g("ixz"), g("iyz"), g("izz")]])
    return Inertia(mass, M.translation, M.rotation @ I @ M.rotation.T)


def build_model_from_urdf(urdf_path, root_joint=False):
    """URDF file -> :class:`Model` following Pinocchio's parser conventions."""
    root = ET.parse(urdf_path).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.get("name"): j for j in root.findall("joint")}
    children = {name: [] for name in links}
    child_links = set()
    for jname in sorted(joints):
        j = joints[jname]
        children[j.find("parent").get("link")].append(jname)
        child_links.add(j.find("child").get("link"))
    roots = [name for name in links if name not in child_links]
    if len(roots) != 1:
        raise ValueError("URDF must have exactly one root link, found %r" % (roots,))

    model = Model(root.get("name", "robot"))

    def visit(link_name, jid, M):
        Y = _link_inertia(links[link_name])
        if Y is not None:
            model.append_body(jid, Y, M)
        for jname in children[link_name]:
            j = joints[jname]
            Mj = M * _origin(j)
            child = j.find("child").get("link")
            jt = j.get("type")
            if jt == "fixed":
                visit(child, jid, Mj)
                continue
            if jt not in _URDF_TYPES:
                raise ValueError("unsupported URDF joint type %r on %s" % (jt, jname))
            ax = j.find("axis")
            axis = _floats(ax.get("xyz") if ax is not None else None, 3, [1, 0, 0])
            axis = axis / np.linalg.norm(axis)
            lim = j.find("limit")
            if lim is not None:
                limits = (float(lim.get("lower", 0.0)), float(lim.get("upper", 0.0)),
                          float(lim.get("velocity", 0.0)), float(lim.get("effort", 0.0)))
            else:
                limits = None
            new = model.add_joint(jid, _URDF_TYPES[jt], axis, Mj, jname, limits)
            visit(child, new, SE3())

    if root_joint:
        rid = model.add_joint(0, JT_FREEFLYER, None, SE3(), "root_joint")
        visit(roots[0], rid, SE3())
    else:
        visit(roots[0], 0, SE3())
    return model
