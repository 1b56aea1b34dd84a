"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Takes the plain model descriptor dict produced by
``robot_mpcs_amd.models.mpcModel.MpcModel.setModel()`` (or a hand-written one).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librmpc_oracle.so")

MAX_JOINTS, MAX_LINKS, MAX_PAIRS, MAX_MODULES, NV_MAX = 8, 8, 4, 8, 24
MAX_XROWS = 32
TRACE_W = 8


class OrcDesc(C.Structure):
    _fields_ = [
        ("robot", C.c_int32), ("N", C.c_int32),
        ("n", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32), ("ns", C.c_int32), ("npar", C.c_int32),
        ("dt", C.c_double),
        ("n_modules", C.c_int32), ("module_kind", C.c_int32 * MAX_MODULES),
        ("nobst", C.c_int32),
        ("n_links", C.c_int32), ("link_frame", C.c_int32 * MAX_LINKS),
        ("n_pairs", C.c_int32), ("pair_frame", (C.c_int32 * 2) * MAX_PAIRS),
        ("end_frame", C.c_int32),
        ("n_joints", C.c_int32), ("joint_type", C.c_int32 * MAX_JOINTS), ("joint_dof", C.c_int32 * MAX_JOINTS),
        ("joint_xyz", (C.c_double * 3) * MAX_JOINTS), ("joint_rot", (C.c_double * 9) * MAX_JOINTS),
        ("joint_axis", (C.c_double * 3) * MAX_JOINTS),
        ("off_r_body", C.c_int32), ("off_obst", C.c_int32), ("off_lin", C.c_int32),
        ("off_lower", C.c_int32), ("off_upper", C.c_int32), ("off_lower_u", C.c_int32),
        ("off_upper_u", C.c_int32), ("off_lower_vel", C.c_int32), ("off_upper_vel", C.c_int32),
        ("off_wu", C.c_int32), ("off_goal", C.c_int32), ("off_wgoal", C.c_int32),
        ("off_wconstr", C.c_int32), ("off_ws", C.c_int32),
        ("has_goal", C.c_int32), ("has_avoid", C.c_int32),
        ("lb", C.c_double * NV_MAX), ("ub", C.c_double * NV_MAX),
        ("max_iter", C.c_int32),
        ("tol_stat", C.c_double), ("tol_eq", C.c_double), ("tol_ineq", C.c_double), ("tol_comp", C.c_double),
        ("mu0", C.c_double),
        ("acc_iters", C.c_int32), ("acc_obj_tol", C.c_double), ("ls_max", C.c_int32),
        # rows of the row-described modules (include/rmpc.h RMPC_MOD_ROWS)
        ("n_xrows", C.c_int32), ("xrow_mod", C.c_int32 * MAX_XROWS), ("xrow_kind", C.c_int32 * MAX_XROWS),
        ("xrow_a", C.c_int32 * MAX_XROWS), ("xrow_b", C.c_int32 * MAX_XROWS), ("xrow_poff", C.c_int32 * MAX_XROWS),
    ]


class OrcStats(C.Structure):
    _fields_ = [
        ("exitflag", C.c_int32), ("iters", C.c_int32),
        ("res_stat", C.c_double), ("res_eq", C.c_double), ("res_ineq", C.c_double), ("res_comp", C.c_double),
        ("obj", C.c_double), ("mu", C.c_double),
    ]


def build(force: bool = False) -> str:
    if os.environ.get("RMPC_ORACLE_LIB"):   # (a build of the restatement made elsewhere: the sanitizer build, `make asan`)
        return os.environ["RMPC_ORACLE_LIB"]
    src = os.path.join(_HERE, "rmpc_oracle.c")
    hdr = os.path.join(_HERE, "rmpc_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librmpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L.orc_desc_size.restype = C.c_int
        L.orc_eval_stage.restype = C.c_int
        L.orc_eval_stage.argtypes = [C.POINTER(OrcDesc), dp, dp, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, C.c_int]
        L.orc_num_rows.restype = C.c_int
        L.orc_num_rows.argtypes = [C.POINTER(OrcDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_fk.restype = C.c_int
        L.orc_fk.argtypes = [C.POINTER(OrcDesc), dp, C.c_int, dp, dp]
        L.orc_fk_curv.restype = C.c_int
        L.orc_fk_curv.argtypes = [C.POINTER(OrcDesc), dp, C.c_int, dp, dp]
        L.orc_stage_curvature.restype = C.c_int
        L.orc_stage_curvature.argtypes = [C.POINTER(OrcDesc), dp, dp, dp, C.c_int, dp]
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [C.POINTER(OrcDesc), dp, dp, dp, dp, C.POINTER(OrcStats), dp]
        L.orc_solve_warm.restype = C.c_int
        L.orc_solve_warm.argtypes = [C.POINTER(OrcDesc), dp, dp, dp, dp, C.POINTER(OrcStats), dp, dp, C.c_double, dp, dp]
        L.orc_solve_batch.restype = C.c_int
        L.orc_solve_batch.argtypes = [C.POINTER(OrcDesc), C.c_int, dp, dp, dp, dp, C.POINTER(OrcStats), C.c_int]
        L.orc_dynamics.restype = C.c_int
        L.orc_dynamics.argtypes = [C.POINTER(OrcDesc), dp, dp, dp]
        assert L.orc_desc_size() == C.sizeof(OrcDesc), "orc_desc layout mismatch"
        _lib = L
    return _lib


def make_desc(d: dict) -> OrcDesc:
    o = OrcDesc()
    o.robot = d["robot"]; o.N = d["N"]
    o.n, o.nx, o.nu, o.ns, o.npar = d["n"], d["nx"], d["nu"], d["ns"], d["npar"]
    o.dt = d["dt"]
    o.n_modules = len(d["module_kind"])
    for i, k in enumerate(d["module_kind"]):
        o.module_kind[i] = k
    o.nobst = d["nobst"]
    o.n_links = len(d["link_frame"])
    for i, f in enumerate(d["link_frame"]):
        o.link_frame[i] = f
    o.n_pairs = len(d["pair_frame"])
    for i, (a, b) in enumerate(d["pair_frame"]):
        o.pair_frame[i][0] = a; o.pair_frame[i][1] = b
    o.end_frame = d["end_frame"]
    o.n_joints = len(d["joints"])
    for i, j in enumerate(d["joints"]):
        o.joint_type[i] = j["type"]; o.joint_dof[i] = j["dof"]
        for c in range(3):
            o.joint_xyz[i][c] = j["xyz"][c]; o.joint_axis[i][c] = j["axis"][c]
        for c in range(9):
            o.joint_rot[i][c] = j["rot"][c]
    for k in ("off_r_body", "off_obst", "off_lin", "off_lower", "off_upper", "off_lower_u", "off_upper_u",
              "off_lower_vel", "off_upper_vel", "off_wu", "off_goal", "off_wgoal", "off_wconstr", "off_ws",
              "has_goal", "has_avoid"):
        setattr(o, k, d[k])
    nv = d["nx"] + d["ns"] + d["nu"]
    for i in range(NV_MAX):
        o.lb[i] = float(d["lb"][i]) if i < nv else -np.inf
        o.ub[i] = float(d["ub"][i]) if i < nv else np.inf
    opt = d.get("options", {})
    o.max_iter = int(opt.get("max_iter", 200))
    o.tol_stat = float(opt.get("tol_stat", 1e-6)); o.tol_eq = float(opt.get("tol_eq", 1e-8))
    o.tol_ineq = float(opt.get("tol_ineq", 1e-8)); o.tol_comp = float(opt.get("tol_comp", 1e-6))
    o.mu0 = float(opt.get("mu0", 1.0))
    o.acc_iters = int(opt.get("acc_iters", 8)); o.acc_obj_tol = float(opt.get("acc_obj_tol", 1e-8))
    o.ls_max = int(opt.get("ls_max", 25))
    xrows = d.get("xrows", [])
    if len(xrows) > MAX_XROWS:
        raise ValueError("more than %d described rows" % MAX_XROWS)
    o.n_xrows = len(xrows)
    for i, r in enumerate(xrows):
        o.xrow_mod[i], o.xrow_kind[i], o.xrow_a[i], o.xrow_b[i], o.xrow_poff[i] = (int(v) for v in r)
    return o


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    def __init__(self, desc: dict):
        self.d = desc
        self.cd = make_desc(desc)
        self.nx, self.nu, self.ns, self.n = desc["nx"], desc["nu"], desc["ns"], desc["n"]
        self.nv = self.nx + self.ns + self.nu
        self.nw = self.ns + self.nu
        self.N, self.npar = desc["N"], desc["npar"]
        nh, m = C.c_int(), C.c_int()
        rc = lib().orc_num_rows(C.byref(self.cd), C.byref(nh), C.byref(m))
        assert rc == 0, rc
        self.nh, self.m = nh.value, m.value

    def fk(self, q, frame):
        q = np.ascontiguousarray(q, dtype=np.float64)
        pos = np.zeros(3); J = np.zeros((3, self.n))
        rc = lib().orc_fk(C.byref(self.cd), _p(q), frame, _p(pos), _p(J))
        assert rc == 0, rc
        return pos, J

    def fk_curv(self, q, frame, F):
        """sum_c F_c d2 pos_c / dq dq of a frame (n x n)"""
        q = np.ascontiguousarray(q, dtype=np.float64); F = np.ascontiguousarray(F, dtype=np.float64)
        Cm = np.zeros((self.n, self.n))
        rc = lib().orc_fk_curv(C.byref(self.cd), _p(q), frame, _p(F), _p(Cm))
        assert rc == 0, rc
        return Cm

    def stage_curvature(self, z, p, lam, fixed_state=False):
        """what the step computation subtracts from the Gauss-Newton q block when it uses the curvature terms"""
        z = np.ascontiguousarray(z, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        assert lam.size == self.nh
        Cm = np.zeros((self.n, self.n))
        rc = lib().orc_stage_curvature(C.byref(self.cd), _p(z), _p(p), _p(lam), 1 if fixed_state else 0, _p(Cm))
        assert rc == 0, rc
        return Cm

    def eval_stage(self, z, p, derivs=True, dynamics=True, fixed_state=False):
        z = np.ascontiguousarray(z, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
        nv, nx, nw = self.nv, self.nx, self.nw
        MR = 64 + 2 * NV_MAX
        f = np.zeros(1); gf = np.zeros(nv); H = np.zeros((nv, nv)); g = np.zeros(MR); Jg = np.zeros((MR, nv))
        xn = np.zeros(nx); A = np.zeros((nx, nx)); Bm = np.zeros((nx, nw))
        r = lib().orc_eval_stage(C.byref(self.cd), _p(z), _p(p), 1 if derivs else 0, _p(f),
                                 _p(gf) if derivs else None, _p(H) if derivs else None, _p(g),
                                 _p(Jg) if derivs else None, _p(xn) if dynamics else None, _p(A), _p(Bm),
                                 1 if fixed_state else 0)
        m = self.m
        return dict(rows=r, f=f[0], gf=gf, H=H, g=g[:m].copy(), Jg=Jg[:m].copy(), xnext=xn, A=A, B=Bm)

    def dynamics(self, x, u):
        x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        xn = np.zeros(self.nx)
        lib().orc_dynamics(C.byref(self.cd), _p(x), _p(u), _p(xn))
        return xn

    def solve(self, xinit, x0, params, trace=False):
        xinit = np.ascontiguousarray(xinit, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        assert xinit.size == self.nx and x0.size == self.N * self.nv and params.size == self.N * self.npar
        z = np.zeros(self.N * self.nv)
        st = OrcStats()
        tr = np.zeros((self.cd.max_iter + 2, TRACE_W)) if trace else None
        rc = lib().orc_solve(C.byref(self.cd), _p(xinit), _p(x0), _p(params), _p(z), C.byref(st),
                             _p(tr) if trace else None)
        assert rc == 0, rc
        out = dict(z=z.reshape(self.N, self.nv), exitflag=st.exitflag, iters=st.iters, res_stat=st.res_stat,
                   res_eq=st.res_eq, res_ineq=st.res_ineq, res_comp=st.res_comp, obj=st.obj, mu=st.mu)
        if trace:
            out["trace"] = tr[: st.iters + 1]
        return out

    def solve_warm(self, xinit, x0, params, duals=None):
        """One instance; ``duals`` = (lam [N, m], nu [N, nx], mu) of the previous solve or None (cold start).
        Returns the usual dict plus ``duals`` for the next call."""
        xinit = np.ascontiguousarray(xinit, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        z = np.zeros(self.N * self.nv)
        lam_out = np.zeros((self.N, self.m)); nu_out = np.zeros((self.N, self.nx))
        st = OrcStats()
        if duals is None:
            lw = nw = None; mw = 0.0
        else:
            lw = _p(np.ascontiguousarray(duals[0], dtype=np.float64)); nw = _p(np.ascontiguousarray(duals[1], dtype=np.float64))
            mw = float(duals[2])
        rc = lib().orc_solve_warm(C.byref(self.cd), _p(xinit), _p(x0), _p(params), _p(z), C.byref(st), lw, nw, mw,
                                  _p(lam_out), _p(nu_out))
        assert rc == 0, rc
        return dict(z=z.reshape(self.N, self.nv), exitflag=st.exitflag, iters=st.iters, res_stat=st.res_stat,
                    res_eq=st.res_eq, res_ineq=st.res_ineq, res_comp=st.res_comp, obj=st.obj, mu=st.mu,
                    duals=(lam_out, nu_out, st.mu))

    def solve_batch(self, xinit, x0, params, nthreads=0):
        xinit = np.ascontiguousarray(xinit, dtype=np.float64)
        B = xinit.shape[0]
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, -1)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(B, -1)
        assert x0.shape[1] == self.N * self.nv and params.shape[1] == self.N * self.npar
        z = np.zeros((B, self.N * self.nv))
        st = (OrcStats * B)()
        rc = lib().orc_solve_batch(C.byref(self.cd), B, _p(xinit), _p(x0), _p(params), _p(z), st, nthreads)
        assert rc == 0, rc
        return dict(
            z=z.reshape(B, self.N, self.nv),
            exitflag=np.array([s.exitflag for s in st], dtype=np.int32),
            iters=np.array([s.iters for s in st], dtype=np.int32),
            res_stat=np.array([s.res_stat for s in st]), res_eq=np.array([s.res_eq for s in st]),
            res_ineq=np.array([s.res_ineq for s in st]), res_comp=np.array([s.res_comp for s in st]),
            obj=np.array([s.obj for s in st]),
        )
