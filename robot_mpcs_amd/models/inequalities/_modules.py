"""Inequality plug-ins, table driven.

The reference resolves the YAML ``constraints`` list to classes by name
(``getattr(robotmpcs.models.inequalities, name)``, ``InequalityManager.py:17-21``)
and asks each for its row count and its ``paramMap`` entries.  Here every plug-in
is ONE row of ``SPECS``: the HIP kernel kind (``rmpc.h`` ``RMPC_MOD_*``; the
arithmetic lives in ``csrc/rmpc_kernels.hip``), the number of rows and the
parameter entries in registration order.  A class per YAML name is generated from
the table so that the lookup by name -- and its ``AttributeError`` for unknown
names -- stays what users of the reference expect.

Row semantics (reference file:line):
  RadialConstraints                 h = ||fk_l(q) - c_i|| - r_i - r_body, links outer, obstacles inner
                                    (RadialConstraints.py:6-23, mpcBase.py:82-101; FIX undefined ``j`` at :22)
  LinearConstraints                 h = |a.fk_l(q) + d| / ||a|| - r_body        (LinearConstraints.py:8-40)
  SelfCollisionAvoidanceConstraints h = ||fk_a(q) - fk_b(q)|| - 2 r_body        (SelfCollision...py:8-27)
  JointLimitConstraints             [q_j - lo_j, hi_j - q_j] interleaved        (JointLimitConstraints.py:8-31)
  VelLimitConstraints               the same on the last two velocity states; FIX: 4 rows, the reference
                                    declares ``_n_ineq = 2`` but returns 4      (VelLimitConstraints.py:8-31)
  InputLimitConstraints             [u_j - lo_j, hi_j - u_j] interleaved        (InputLimitConstraints.py:7-29)
"""
from typing import Callable, List, NamedTuple, Tuple

from robot_mpcs_amd.models.mpcBase import ModelContext, ParamLayout


class Spec(NamedTuple):
    kind: int                                                  # RMPC_MOD_* of include/rmpc.h
    rows: Callable[[ModelContext], int]
    params: Callable[[ModelContext], List[Tuple[str, int]]]     # (name, count) in registration order


def _fk_rows(c: ModelContext) -> int:
    return c.config.number_obstacles * len(c.robot.collision_links)


SPECS = {
    "RadialConstraints": Spec(0, _fk_rows, lambda c: [("r_body", 1), ("obst", 4 * c.config.number_obstacles)]),
    "LinearConstraints": Spec(1, _fk_rows, lambda c: [("r_body", 1)] + [("lin_constrs_%d" % i, 4) for i in
                                                                        range(c.config.number_obstacles)]),
    "SelfCollisionAvoidanceConstraints": Spec(2, lambda c: len(c.robot.selfCollision['pairs']),
                                              lambda c: [("r_body", 1)]),
    "JointLimitConstraints": Spec(3, lambda c: 2 * c.n, lambda c: [("lower_limits", c.n), ("upper_limits", c.n)]),
    "VelLimitConstraints": Spec(4, lambda c: 4, lambda c: [("lower_limits_vel", 2), ("upper_limits_vel", 2)]),
    "InputLimitConstraints": Spec(5, lambda c: 2 * c.nu,
                                  lambda c: [("lower_limits_u", c.nu), ("upper_limits_u", c.nu)]),
}


class InequalityModule:
    """One constraint plug-in bound to a model context."""
    NAME = ""

    def __init__(self, ctx: ModelContext):
        self.spec = SPECS[self.NAME]
        self.KIND = self.spec.kind
        self._n_ineq = int(self.spec.rows(ctx))
        self._params = self.spec.params(ctx)

    def register(self, layout: ParamLayout) -> None:
        for name, count in self._params:
            layout.add(name, count)


def _make(name):
    return type(name, (InequalityModule,), {"NAME": name, "__doc__": "constraint plug-in '%s' (see SPECS)" % name})


class DescribedRows:
    """A constraint plug-in given as a ROW DESCRIPTION in the YAML (top-level block ``plugins: constraints: <name>:``)
    instead of a class: the counterpart of dropping a user class into ``robotmpcs/models/inequalities`` and naming it
    in ``mpc.constraints`` (``InequalityManager.py:17-21``).  The vocabulary is what the kernels evaluate
    (include/rmpc.h ``RMPC_ROW_*``) -- variants of the six modules on other links, pairs, joints or inputs, with
    parameter entries of their own -- lowered through the runtime row tables: no rebuild of the library.

      rows: radial    links: [...]            spheres: k     entry ``<name>`` (4 k: x y z r per sphere)
      rows: linear    links: [...]            planes: k      entry ``<name>`` (4 k: a b c d per plane)
      rows: self      pairs: [[a, b], ...]
      rows: limits    variables: q | qdot | u indices: [...] entries ``<name>_lower`` / ``<name>_upper``
                      (the key is not called ``on``: YAML 1.1 reads that as a boolean)
    ``params: <entry>`` instead of ``spheres`` / ``planes`` re-uses an entry registered earlier (e.g. ``obst``).
    Row order as in the built-in modules: links outer, objects inner; lower then upper per index."""
    KIND = 6    # RMPC_MOD_ROWS
    ROW_RADIAL, ROW_LINEAR, ROW_SELF, ROW_VAR = 0, 1, 2, 3

    def __init__(self, ctx: ModelContext, name: str, spec: dict):
        self.NAME, self._ctx, self._spec = name, ctx, dict(spec)
        kind = self._spec.get("rows")
        if kind in ("radial", "linear"):
            self._links = [ctx.chain.frame_of(l) for l in self._spec["links"]]
            self._count = int(self._spec.get("spheres", self._spec.get("planes", 0)))
            self._shared = self._spec.get("params")
            if not self._shared and self._count < 1:
                raise ValueError("plug-in '%s': give spheres / planes (a count) or params (an existing entry)" % name)
            self._n_ineq = None if self._shared else len(self._links) * self._count
        elif kind == "self":
            self._pairs = [[ctx.chain.frame_of(a), ctx.chain.frame_of(b)] for a, b in self._spec["pairs"]]
            self._n_ineq = len(self._pairs)
        elif kind == "limits":
            if "variables" not in self._spec:
                raise ValueError("plug-in '%s': variables must be q, qdot or u" % name)
            on = self._spec["variables"]
            base = {"q": 0, "qdot": ctx.n, "u": ctx.nx + ctx.ns}.get(on)
            if base is None:
                raise ValueError("plug-in '%s': variables must be q, qdot or u" % name)
            top = {"q": ctx.n, "qdot": ctx.nx - ctx.n, "u": ctx.nu}[on]
            self._vars = []
            for i in self._spec["indices"]:
                if not 0 <= int(i) < top:
                    raise ValueError("plug-in '%s': index %s out of range" % (name, i))
                self._vars.append(base + int(i))
            self._n_ineq = 2 * len(self._vars)
        else:
            raise ValueError("plug-in '%s': rows must be radial, linear, self or limits" % name)
        self._kind = kind

    def register(self, layout: ParamLayout) -> None:
        name = self.NAME
        if self._kind in ("radial", "linear"):
            layout.add("r_body", 1)
            if self._shared:
                if self._shared not in layout.entries or len(layout.entries[self._shared]) % 4:
                    raise ValueError("plug-in '%s': params '%s' is no registered entry of 4 k values" % (name, self._shared))
                self._count = len(layout.entries[self._shared]) // 4
                self._n_ineq = len(self._links) * self._count
                self._entry = self._shared
            else:
                # (the kernels address a sphere / plane as an entry of the obstacle / plane list: 4-aligned behind it)
                anchor = layout.offset("obst" if self._kind == "radial" else "lin_constrs_0")
                if anchor < 0:
                    anchor = layout.anchors.setdefault(self._kind, layout.size)
                layout.pad_to(anchor, 4)
                layout.add(name, 4 * self._count)
                self._entry = name
        elif self._kind == "self":
            layout.add("r_body", 1)
        else:
            layout.add(name + "_lower", len(self._vars))
            layout.add(name + "_upper", len(self._vars))

    def rows(self, layout: ParamLayout, module_index: int):
        """(module, kind, a, b, parameter offset) per row, in row order: rmpc_desc::xrow_*"""
        out = []
        if self._kind in ("radial", "linear"):
            off = layout.offset(self._entry)
            rk = self.ROW_RADIAL if self._kind == "radial" else self.ROW_LINEAR
            for f in self._links:
                for i in range(self._count):
                    out.append((module_index, rk, f, 0, off + 4 * i))
        elif self._kind == "self":
            for a, b in self._pairs:
                out.append((module_index, self.ROW_SELF, a, b, 0))
        else:
            lo, hi = layout.offset(self.NAME + "_lower"), layout.offset(self.NAME + "_upper")
            for i, v in enumerate(self._vars):
                out.append((module_index, self.ROW_VAR, v, +1, lo + i))
                out.append((module_index, self.ROW_VAR, v, -1, hi + i))
        return out


RadialConstraints = _make("RadialConstraints")
LinearConstraints = _make("LinearConstraints")
SelfCollisionAvoidanceConstraints = _make("SelfCollisionAvoidanceConstraints")
JointLimitConstraints = _make("JointLimitConstraints")
VelLimitConstraints = _make("VelLimitConstraints")
InputLimitConstraints = _make("InputLimitConstraints")
