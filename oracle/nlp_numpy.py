"""Independent numpy restatement of the reference NLP (TEST INFRASTRUCTURE).

Written separately from ``rmpc_oracle.c`` (homogeneous 4x4 transforms instead
of the recursive position/axis form, no analytic derivatives -- those are
checked against central finite differences of these functions) so that the two
oracles pin each other.  Also provides the scipy SLSQP cross-solve of the same
NLP.  PARITY UNPINNED against the reference itself: see rmpc_oracle.h.

Reference lines restated:
  z layout              robotmpcs/models/mpcBase.py:76-80
  obstacle distances    robotmpcs/models/mpcBase.py:82-101
  double integrator     robotmpcs/models/mpcModel.py:65-69
  unicycle              robotmpcs/models/diff_drive_mpc_model.py:24-41
  ERK2, 5 nodes         robotmpcs/models/mpcModel.py:118-120 (explicit midpoint)
  inequality modules    robotmpcs/models/inequalities/*.py
  objectives            robotmpcs/models/objectives/{ObjectiveManager,goal_reaching,constraint_avoidance}.py
  NLP assembly          robotmpcs/models/mpcModel.py:74-108
"""
from __future__ import annotations

import numpy as np

RADIAL, LINEAR, SELFCOLL, JOINTLIM, VELLIM, INPUTLIM = range(6)
FIXED, REVOLUTE, PRISMATIC = range(3)


def _hom(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def _axis_angle(ax, th):
    ax = np.asarray(ax, dtype=float)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def fk(desc, q, frame):
    """Position of the frame attached to the child link of joint ``frame``."""
    T = np.eye(4)
    if desc["robot"] == 1:  # diff-drive base pose prefixes the chain
        c, s = np.cos(q[2]), np.sin(q[2])
        T = _hom(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]), np.array([q[0], q[1], 0.0]))
    for j in desc["joints"][: frame + 1]:
        T = T @ _hom(np.asarray(j["rot"], dtype=float).reshape(3, 3), np.asarray(j["xyz"], dtype=float))
        if j["type"] == REVOLUTE:
            T = T @ _hom(_axis_angle(j["axis"], q[j["dof"]]), np.zeros(3))
        elif j["type"] == PRISMATIC:
            T = T @ _hom(np.eye(3), np.asarray(j["axis"], dtype=float) * q[j["dof"]])
    return T[:3, 3].copy()


def cont_dyn(desc, x, u):
    n = desc["n"]
    if desc["robot"] == 0:
        return np.concatenate([x[n: 2 * n], u])
    th, v, w = x[2], x[6], x[7]
    return np.array([np.cos(th) * v, np.sin(th) * v, w, 0.0, 0.0, 0.0, u[0], u[1]])


# ERK2 tableau of the discretisation.  FORCES Pro's choice is not verifiable here: the product and
# the C oracle use the explicit midpoint rule; "heun" (explicit trapezoid) exists only for the sensitivity line of
# tests/golden/make_scipy_golden.py (how far the first control moves if the other ERK2 tableau is the true one).
ERK2_TABLEAU = "midpoint"


def dynamics(desc, x, u, nodes=5):
    h = desc["dt"] / nodes
    x = np.array(x, dtype=float)
    for _ in range(nodes):
        k1 = cont_dyn(desc, x, u)
        if ERK2_TABLEAU == "heun":
            k2 = cont_dyn(desc, x + h * k1, u)
            x = x + 0.5 * h * (k1 + k2)
        else:
            k2 = cont_dyn(desc, x + 0.5 * h * k1, u)
            x = x + h * k2
    return x


def split(desc, z):
    nx, ns, nu, n = desc["nx"], desc["ns"], desc["nu"], desc["n"]
    return z[:n], z[:nx], (z[nx] if ns else 0.0), z[nx + ns: nx + ns + nu]


def module_rows(desc, z, p):
    """List (per module, YAML order) of un-slacked row values."""
    q, x, s, u = split(desc, z)
    nx, n, nu = desc["nx"], desc["n"], desc["nu"]
    out = []
    for kind in desc["module_kind"]:
        rows = []
        if kind == RADIAL:
            rb = p[desc["off_r_body"]]
            for fr in desc["link_frame"]:
                pos = fk(desc, q, fr)
                for i in range(desc["nobst"]):
                    ob = p[desc["off_obst"] + 4 * i: desc["off_obst"] + 4 * i + 4]
                    rows.append(np.linalg.norm(pos - ob[:3]) - ob[3] - rb)
        elif kind == LINEAR:
            rb = p[desc["off_r_body"]]
            for fr in desc["link_frame"]:
                pos = fk(desc, q, fr)
                for i in range(desc["nobst"]):
                    pl = p[desc["off_lin"] + 4 * i: desc["off_lin"] + 4 * i + 4]
                    rows.append(abs(pl[:3] @ pos + pl[3]) / np.linalg.norm(pl[:3]) - rb)
        elif kind == SELFCOLL:
            rb = p[desc["off_r_body"]]
            for a, b in desc["pair_frame"]:
                rows.append(np.linalg.norm(fk(desc, q, a) - fk(desc, q, b)) - 2 * rb)
        elif kind == JOINTLIM:
            lo = p[desc["off_lower"]: desc["off_lower"] + n]
            hi = p[desc["off_upper"]: desc["off_upper"] + n]
            for j in range(n):
                rows += [q[j] - lo[j], hi[j] - q[j]]
        elif kind == VELLIM:
            lo = p[desc["off_lower_vel"]: desc["off_lower_vel"] + 2]
            hi = p[desc["off_upper_vel"]: desc["off_upper_vel"] + 2]
            vel = z[n:nx][-2:]
            for j in range(2):
                rows += [vel[j] - lo[j], hi[j] - vel[j]]
        elif kind == INPUTLIM:
            lo = p[desc["off_lower_u"]: desc["off_lower_u"] + nu]
            hi = p[desc["off_upper_u"]: desc["off_upper_u"] + nu]
            for j in range(nu):
                rows += [u[j] - lo[j], hi[j] - u[j]]
        else:
            raise ValueError(kind)
        out.append(rows)
    return out


def stage_ineq(desc, z, p, with_bounds=True, fixed_state=False):
    """fixed_state: stage 1 (x pinned to xinit) -- state-only, unsoftened rows are constants of
    the problem and are neutralised (value 1), exactly as in rmpc_oracle.c."""
    q, x, s, u = split(desc, z)
    mods = module_rows(desc, z, p)
    if fixed_state and not desc["ns"]:
        mods = [[1.0] * len(rows) if kind != INPUTLIM else rows for kind, rows in zip(desc["module_kind"], mods)]
    rows = [r for mod in mods for r in mod]
    g = np.array(rows, dtype=float)
    if desc["ns"]:
        g = g + s
    if with_bounds:
        nv = desc["nx"] + desc["ns"] + desc["nu"]
        lb, ub = np.asarray(desc["lb"][:nv], dtype=float), np.asarray(desc["ub"][:nv], dtype=float)
        lo, hi = (z - lb), (ub - z)
        if fixed_state:
            lo[: desc["nx"]] = 1.0
            hi[: desc["nx"]] = 1.0
        g = np.concatenate([g, lo[np.isfinite(lb)], hi[np.isfinite(ub)]])
    return g


def stage_cost(desc, z, p, fixed_state=False):
    q, x, s, u = split(desc, z)
    J = 0.0
    if desc["has_goal"]:
        e = fk(desc, q, desc["end_frame"]) - p[desc["off_goal"]: desc["off_goal"] + 3]
        J += float(e @ (p[desc["off_wgoal"]: desc["off_wgoal"] + 3] * e))
    if desc["has_avoid"]:
        for i, rows in enumerate(module_rows(desc, z, p)):
            w = p[desc["off_wconstr"] + i]
            if rows and w != 0.0 and not (fixed_state and desc["module_kind"][i] != INPUTLIM):
                J += desc["N"] * w / rows[0]
    wu = p[desc["off_wu"]: desc["off_wu"] + desc["nu"]]
    J += float(u @ (wu * u))
    if desc["ns"]:
        J += p[desc["off_ws"]] * s * s
    return J


def fd_grad(fun, z, eps=1e-6):
    z = np.asarray(z, dtype=float)
    f0 = np.atleast_1d(fun(z))
    out = np.zeros((f0.size, z.size))
    for i in range(z.size):
        zp, zm = z.copy(), z.copy()
        zp[i] += eps
        zm[i] -= eps
        out[:, i] = (np.atleast_1d(fun(zp)) - np.atleast_1d(fun(zm))) / (2 * eps)
    return out


# ---------------------------------------------------------------------------
# whole-horizon NLP for scipy
# ---------------------------------------------------------------------------
class HorizonNLP:
    """min sum_k l(z_k, p_k)  s.t. x_1 = xinit, x_{k+1} = Phi(x_k, u_k),
    h(z_k, p_k) >= 0, lb <= z_k <= ub (reference mpcModel.py:74-108).
    x_1 is eliminated (fixed to xinit); decision vector = [w_1, z_2 .. z_N]."""

    def __init__(self, desc, xinit, params):
        self.d = desc
        self.N = desc["N"]
        self.nx, self.ns, self.nu = desc["nx"], desc["ns"], desc["nu"]
        self.nv = self.nx + self.ns + self.nu
        self.nw = self.ns + self.nu
        self.xinit = np.asarray(xinit, dtype=float)
        self.P = np.asarray(params, dtype=float).reshape(self.N, desc["npar"])

    def unpack(self, y):
        Z = np.zeros((self.N, self.nv))
        Z[0, : self.nx] = self.xinit
        Z[0, self.nx:] = y[: self.nw]
        Z[1:] = y[self.nw:].reshape(self.N - 1, self.nv)
        return Z

    def pack(self, Z):
        return np.concatenate([Z[0, self.nx:], Z[1:].reshape(-1)])

    def objective(self, y):
        Z = self.unpack(y)
        return sum(stage_cost(self.d, Z[k], self.P[k], fixed_state=(k == 0)) for k in range(self.N))

    def eq(self, y):
        Z = self.unpack(y)
        r = []
        for k in range(self.N - 1):
            u = Z[k, self.nx + self.ns:]
            r.append(dynamics(self.d, Z[k, : self.nx], u) - Z[k + 1, : self.nx])
        return np.concatenate(r) if r else np.zeros(0)

    def ineq(self, y):
        Z = self.unpack(y)
        return np.concatenate([stage_ineq(self.d, Z[k], self.P[k], with_bounds=False, fixed_state=(k == 0))
                               for k in range(self.N)])

    def bounds(self):
        nv = self.nv
        lb, ub = np.asarray(self.d["lb"][:nv], dtype=float), np.asarray(self.d["ub"][:nv], dtype=float)
        lo = np.concatenate([lb[self.nx:], np.tile(lb, self.N - 1)])
        hi = np.concatenate([ub[self.nx:], np.tile(ub, self.N - 1)])
        return [(a if np.isfinite(a) else None, b if np.isfinite(b) else None) for a, b in zip(lo, hi)]

    def solve_slsqp(self, Z0, maxiter=500, ftol=1e-12):
        from scipy.optimize import minimize

        y0 = self.pack(np.asarray(Z0, dtype=float).reshape(self.N, self.nv))
        res = minimize(
            self.objective, y0, method="SLSQP", bounds=self.bounds(),
            constraints=[{"type": "eq", "fun": self.eq}, {"type": "ineq", "fun": self.ineq}],
            options={"maxiter": maxiter, "ftol": ftol},
        )
        return self.unpack(res.x), res


# ---------------------------------------------------------------------------
# structured finite-difference Jacobians (stage-wise), so that a full-horizon scipy solve is
# affordable without borrowing any derivative from rmpc_oracle.c
# ---------------------------------------------------------------------------
class StructuredNLP(HorizonNLP):
    """HorizonNLP with dense Jacobians assembled from per-stage central differences of the numpy
    stage functions (each stage depends on its own nvar variables, the dynamics couple neighbours)."""

    EPS = 1e-6

    def _cols(self, k):
        """decision-vector columns of stage k's variables (stage 0: only s, u are free)"""
        if k == 0:
            return np.arange(self.nx, self.nv), np.arange(0, self.nw)
        off = self.nw + (k - 1) * self.nv
        return np.arange(0, self.nv), np.arange(off, off + self.nv)

    def grad(self, y):
        Z = self.unpack(y)
        g = np.zeros_like(y)
        for k in range(self.N):
            loc, col = self._cols(k)
            G = fd_grad(lambda zz: stage_cost(self.d, zz, self.P[k], fixed_state=(k == 0)), Z[k], self.EPS)[0]
            g[col] = G[loc]
        return g

    def eq_jac(self, y):
        Z = self.unpack(y)
        J = np.zeros(((self.N - 1) * self.nx, y.size))
        for k in range(self.N - 1):
            loc, col = self._cols(k)
            D = fd_grad(lambda zz: dynamics(self.d, zz[: self.nx], zz[self.nx + self.ns:]), Z[k], self.EPS)
            J[k * self.nx:(k + 1) * self.nx][:, col] = D[:, loc]
            _, coln = self._cols(k + 1)
            J[k * self.nx:(k + 1) * self.nx, coln[: self.nx]] -= np.eye(self.nx)
        return J

    def ineq_jac(self, y):
        Z = self.unpack(y)
        rows = []
        for k in range(self.N):
            loc, col = self._cols(k)
            D = fd_grad(lambda zz: stage_ineq(self.d, zz, self.P[k], with_bounds=False, fixed_state=(k == 0)),
                        Z[k], self.EPS)
            Jk = np.zeros((D.shape[0], y.size))
            Jk[:, col] = D[:, loc]
            rows.append(Jk)
        return np.vstack(rows)

    def var_scale(self):
        """Column scaling y = D y' handed to scipy: the slack variable carries the weight ws = 1e10 in the cost
        (ws s^2), which SLSQP's BFGS model cannot digest; with s = s' / sqrt(ws) the term is s'^2.  The NLP
        itself is unchanged (same optimum in the original variables)."""
        d = np.ones(self.nw + (self.N - 1) * self.nv)
        if self.ns:
            sc = 1.0 / np.sqrt(max(1.0, float(self.P[0, self.d["off_ws"]])))
            d[0] = sc                                  # stage 1: [s, u]
            d[self.nw + self.nx:: self.nv] = sc        # stages 2..N: position nx inside [x, s, u]
        return d

    def solve_slsqp(self, Z0, maxiter=400, ftol=1e-13):
        from scipy.optimize import minimize
        D = self.var_scale()
        y0 = self.pack(np.asarray(Z0, dtype=float).reshape(self.N, self.nv)) / D
        bnds = [(None if a is None else a / dd, None if b is None else b / dd) for (a, b), dd in zip(self.bounds(), D)]
        res = minimize(
            lambda y: self.objective(D * y), y0, jac=lambda y: D * self.grad(D * y), method="SLSQP", bounds=bnds,
            constraints=[{"type": "eq", "fun": lambda y: self.eq(D * y), "jac": lambda y: self.eq_jac(D * y) * D[None, :]},
                         {"type": "ineq", "fun": lambda y: self.ineq(D * y), "jac": lambda y: self.ineq_jac(D * y) * D[None, :]}],
            options={"maxiter": maxiter, "ftol": ftol},
        )
        res.x = D * res.x
        return self.unpack(res.x), res

    def solve_trust_constr(self, Z0, maxiter=3000, gtol=1e-9, xtol=1e-12):
        """The same NLP through scipy's trust-region interior-point method (``trust-constr``, BFGS model of the
        Lagrangian): a second algorithm family beside SLSQP's active-set SQP."""
        from scipy.optimize import BFGS, Bounds, NonlinearConstraint, minimize
        D = self.var_scale()
        y0 = self.pack(np.asarray(Z0, dtype=float).reshape(self.N, self.nv)) / D
        lo = np.array([-np.inf if a is None else a / dd for (a, b), dd in zip(self.bounds(), D)])
        hi = np.array([np.inf if b is None else b / dd for (a, b), dd in zip(self.bounds(), D)])
        cons = [NonlinearConstraint(lambda y: self.eq(D * y), 0.0, 0.0, jac=lambda y: self.eq_jac(D * y) * D[None, :], hess=BFGS()),
                NonlinearConstraint(lambda y: self.ineq(D * y), 0.0, np.inf, jac=lambda y: self.ineq_jac(D * y) * D[None, :], hess=BFGS())]
        res = minimize(lambda y: self.objective(D * y), y0, jac=lambda y: D * self.grad(D * y), hess=BFGS(), method="trust-constr",
                       bounds=Bounds(lo, hi, keep_feasible=False), constraints=cons,
                       options={"maxiter": maxiter, "gtol": gtol, "xtol": xtol, "initial_barrier_parameter": 0.1})
        res.x = D * res.x
        return self.unpack(res.x), res
