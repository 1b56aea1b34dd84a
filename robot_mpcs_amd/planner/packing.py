"""Batched parameter packing and warm start (host logic, numpy only).

Vectorised counterpart of the per-stage Python loops of the reference planner
(``robotmpcs/planner/mpcPlanner.py``): ``reset`` (``:83-104``), the ``set*``
methods (``:120-210``), ``updateDynamicObstacles`` (``:144-161``) and
``setX0`` / ``shiftHorizon`` (``:215-236``), with a leading batch axis ``B``.

Layout kept from the reference: ``params[b]`` is the flat ``N * npar`` vector
with stride ``npar`` per stage (``:91``), ``x0[b]`` is ``(N, nx+ns+nu)``
(``:85``).  Intended-semantics fixes of reference bugs are marked ``FIX``.
"""
from __future__ import annotations

import numpy as np


class ParamPacker:
    def __init__(self, paramMap: dict, properties: dict, config, batch: int = 1):
        self._paramMap = paramMap
        self._properties = properties
        self._config = config
        self.B = int(batch)
        self.N = int(config.time_horizon)
        self.nx = int(properties['nx'])
        self.nu = int(properties['nu'])
        self.ns = int(properties['ns'])
        self.npar = int(properties['npar'])
        self.m = int(properties['m'])
        self.nvar = self.nx + self.nu + self.ns
        self._r = 0.1
        self.reset()

    # -- views ---------------------------------------------------------
    @property
    def params(self) -> np.ndarray:
        """(B, N*npar) flat parameter vectors (``all_parameters``)."""
        return self._params

    @property
    def p3(self) -> np.ndarray:
        return self._params.reshape(self.B, self.N, self.npar)

    def idx(self, name):
        return np.asarray(self._paramMap[name], dtype=np.int64)

    def _bcast(self, value, shape):
        a = np.asarray(value, dtype=np.float64)
        return np.broadcast_to(a, shape)

    # -- reset (reference mpcPlanner.py:83-104) --------------------------
    def reset(self):
        self._x0 = np.zeros((self.B, self.N, self.nvar))
        self._xinit = np.zeros((self.B, self.nx))
        self._initial_step = True
        self._params = np.zeros((self.B, self.N * self.npar), dtype=float)
        p3 = self.p3
        if "wgoal" in self._paramMap:
            p3[:, :, self.idx("wgoal")] = self._config.weights["w"]
        p3[:, :, self.idx("wu")] = self._config.weights["wu"]
        if self._config.slack:
            p3[:, :, self.idx("ws")] = self._config.weights["ws"]

    def setEntry(self, name, value):
        """Parameter entry ``name`` of every stage: ``value`` broadcast to (B, N, len(entry)) -- the setter of the
        plug-ins given as row descriptions (their entries: ``<name>``, ``<name>_lower`` / ``<name>_upper``)."""
        ix = self.idx(name)
        self.p3[:, :, ix] = self._bcast(value, (self.B, self.N, len(ix)))

    # -- setters (reference mpcPlanner.py:120-210) -----------------------
    def setRadialConstraints(self, obst_pos, obst_radius, r_body):
        """obst_pos (..., n_given, 3), obst_radius (..., n_given); slots beyond
        n_given are filled with the reference's ``EmptyObstacle`` (position -100,
        radius -100; ``mpcPlanner.py:18-26``)."""
        self._r = 0.1  # reference :121
        nob = int(self._config.number_obstacles)
        p3 = self.p3
        p3[:, :, self.idx("r_body")[0]] = self._bcast(r_body, (self.B,))[:, None]
        pos = np.asarray(obst_pos, dtype=np.float64).reshape((-1,) + np.shape(obst_pos)[-2:]) if np.size(obst_pos) else np.zeros((1, 0, 3))
        rad = np.asarray(obst_radius, dtype=np.float64).reshape((-1, pos.shape[1])) if np.size(obst_radius) else np.zeros((1, 0))
        full = np.full((self.B, nob, self.m + 1), -100.0)
        ng = min(nob, pos.shape[1])
        full[:, :ng, : pos.shape[2]] = np.broadcast_to(pos[:, :ng, :], (self.B, ng, pos.shape[2]))
        full[:, :ng, self.m] = np.broadcast_to(rad[:, :ng], (self.B, ng))
        p3[:, :, self.idx("obst")] = full.reshape(self.B, 1, nob * (self.m + 1))

    def setLinearConstraints(self, lin_constr, r_body):
        """lin_constr[..., j, i, 0:4]: plane i at stage j (reference :135-141)."""
        nob = int(self._config.number_obstacles)
        lc = self._bcast(lin_constr, (self.B, self.N, nob, 4)) if np.ndim(lin_constr) == 4 else \
            self._bcast(np.asarray(lin_constr, dtype=np.float64)[None], (self.B, self.N, nob, 4))
        p3 = self.p3
        p3[:, :, self.idx("r_body")[0]] = self._bcast(r_body, (self.B,))[:, None]
        for i in range(nob):
            p3[:, :, self.idx("lin_constrs_" + str(i))] = lc[:, :, i, :]

    def updateDynamicObstacles(self, obstArray):
        """obstArray (B, 9*k) = [pos(3), vel(3), acc(3)] per obstacle; stage i is
        predicted at time dt*i (reference :144-161).  FIX: obstacle j reads its
        own slice (the reference never offsets by j) and ``dt`` is the model
        time step (``self.dt()`` does not exist there)."""
        oa = np.asarray(obstArray, dtype=np.float64).reshape(self.B, -1)
        m = self.m
        nbDyn = oa.shape[1] // (3 * m)
        nob = int(self._config.number_obstacles)
        dt = float(self._config.time_step)
        istage = np.arange(self.N, dtype=np.float64)
        p3 = self.p3
        obst_idx = self.idx("obst")
        for j in range(nob):
            if j < nbDyn:
                sl = oa[:, 3 * m * j: 3 * m * (j + 1)]
                pos, vel, acc = sl[:, 0:m], sl[:, m:2 * m], sl[:, 2 * m:3 * m]
            else:
                pos = np.full((self.B, m), -100.0)
                vel = np.zeros((self.B, m))
                acc = np.zeros((self.B, m))
            # same association as the reference: pos + (vel*dt)*i + (0.5*(dt*i)**2)*acc
            pred = pos[:, None, :] + (vel[:, None, :] * dt) * istage[None, :, None] \
                + (0.5 * (dt * istage[None, :, None]) ** 2) * acc[:, None, :]
            p3[:, :, obst_idx[j * (m + 1): j * (m + 1) + m]] = pred
            p3[:, :, obst_idx[j * (m + 1) + m]] = self._r

    def setSelfCollisionAvoidanceConstraints(self, r_body):
        self.p3[:, :, self.idx("r_body")[0]] = self._bcast(r_body, (self.B,))[:, None]

    def setJointLimits(self, limits):
        """limits (..., 2, n) (reference :167-175)."""
        n = int(self._config.n)
        lim = self._bcast(limits, (self.B, 2, np.shape(limits)[-1]))
        self.p3[:, :, self.idx("lower_limits")[:n]] = lim[:, None, 0, :n]
        self.p3[:, :, self.idx("upper_limits")[:n]] = lim[:, None, 1, :n]

    def setVelLimits(self, limits_vel):
        lim = self._bcast(limits_vel, (self.B, 2, np.shape(limits_vel)[-1]))
        self.p3[:, :, self.idx("lower_limits_vel")] = lim[:, None, 0, :2]
        self.p3[:, :, self.idx("upper_limits_vel")] = lim[:, None, 1, :2]

    def setInputLimits(self, limits_u):
        lim = self._bcast(limits_u, (self.B, 2, np.shape(limits_u)[-1]))
        self.p3[:, :, self.idx("lower_limits_u")] = lim[:, None, 0, : self.nu]
        self.p3[:, :, self.idx("upper_limits_u")] = lim[:, None, 1, : self.nu]

    def setGoalReaching(self, goal_position):
        """goal (..., <=3), zero padded to m = 3 (reference :197-204)."""
        g = np.asarray(goal_position, dtype=np.float64)
        g = g.reshape(-1, g.shape[-1]) if g.ndim > 1 else g[None, :]
        full = np.zeros((self.B, self.m))
        k = min(self.m, g.shape[1])
        full[:, :k] = np.broadcast_to(g[:, :k], (self.B, k))
        self.p3[:, :, self.idx("goal")] = full[:, None, :]

    def setConstraintAvoidance(self):
        self.p3[:, :, self.idx("wconstr")] = np.asarray(self._config.weights["wconstr"], dtype=np.float64)

    # -- warm start (reference mpcPlanner.py:215-236) ---------------------
    def shiftHorizon(self, z_prev):
        """z_prev (B, N, nvar) = previous solution.  x0[k-2] = out[k] for
        k = 2..N and x0[N-1] = out[N] (net effect of reference :215-226)."""
        self._x0[:, : self.N - 1, :] = z_prev[:, 1:, :]
        self._x0[:, self.N - 1, :] = z_prev[:, self.N - 1, :]

    def setX0(self, xinit, z_prev=None, initialize_type="current_state"):
        self._xinit = np.asarray(xinit, dtype=np.float64).reshape(self.B, self.nx)
        if initialize_type == "current_state" or (initialize_type == "previous_plan" and self._initial_step):
            self._x0[:, :, 0: self.nx] = self._xinit[:, None, :]
            self._initial_step = False
        elif initialize_type == "previous_plan":
            self.shiftHorizon(z_prev)
        return self._x0
