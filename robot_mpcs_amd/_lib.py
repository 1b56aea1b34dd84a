"""ctypes binding of the MI355X solver library (``csrc/librmpc_hip.so``).

This is the thin host layer above the C ABI declared in ``include/rmpc.h``;
it plays the role ``forcespro.nlp.Solver`` plays for the reference planner
(``robotmpcs/planner/mpcPlanner.py:73,262``).  There is no CPU fallback: if the
library is missing or no HIP device is present the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMPC_LIB_PATH") or os.path.join(_HERE, "csrc", "librmpc_hip.so")

MAX_JOINTS, MAX_LINKS, MAX_PAIRS, MAX_MODULES, NV_MAX = 8, 8, 4, 8, 24
MAX_XROWS = 32
NUM_KERNELS = 6


class RmpcError(RuntimeError):
    pass


class RmpcDesc(C.Structure):
    """Mirror of ``rmpc_desc`` (include/rmpc.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32),
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


class RmpcScene(C.Structure):
    """Mirror of ``rmpc_scene`` (include/rmpc.h): device pointers + broadcast weights."""
    _fields_ = [
        ("struct_size", C.c_int32),
        ("goal", C.c_void_p), ("r_body", C.c_void_p), ("obst", C.c_void_p), ("obst_dyn", C.c_void_p),
        ("dyn_radius", C.c_double),
        ("lower_limits", C.c_void_p), ("upper_limits", C.c_void_p),
        ("lower_limits_u", C.c_void_p), ("upper_limits_u", C.c_void_p),
        ("lower_limits_vel", C.c_void_p), ("upper_limits_vel", C.c_void_p),
        ("lin_constrs", C.c_void_p),
        ("w", C.c_double), ("wu", C.c_double), ("ws", C.c_double),
        ("wconstr", C.c_double * MAX_MODULES),
    ]


class RetargetArgs(C.Structure):
    """Mirror of ``rmpc_retarget`` (include/rmpc.h): the steady loop's goal hand-over, device pointers."""
    _fields_ = [
        ("struct_size", C.c_int32), ("pool_len", C.c_int32),
        ("xinit", C.c_void_p), ("x0", C.c_void_p), ("exitflag", C.c_void_p), ("iters", C.c_void_p), ("goal", C.c_void_p),
        ("goal_pool", C.c_void_p), ("x_start", C.c_void_p), ("lower_limits", C.c_void_p), ("upper_limits", C.c_void_p), ("cursor", C.c_void_p), ("dwell", C.c_void_p), ("failrun", C.c_void_p),
        ("tol", C.c_double), ("settle_vel", C.c_double), ("mu_regoal", C.c_double),
        ("settle_min_dwell", C.c_int32), ("max_dwell", C.c_int32), ("fail_reset_after", C.c_int32), ("reserved", C.c_int32),
        ("counts", C.c_void_p),
    ]


# every symbol include/rmpc.h declares
EXPORTED_SYMBOLS = [
    "rmpc_version", "rmpc_source_hash", "rmpc_last_error", "rmpc_desc_size", "rmpc_create", "rmpc_destroy", "rmpc_solve_batch",
    "rmpc_solve_batch_device", "rmpc_workspace_bytes", "rmpc_set_warm_start", "rmpc_set_pass_budget", "rmpc_is_fused", "rmpc_fused_kernel_name", "rmpc_is_async", "rmpc_set_profiling", "rmpc_get_profile",
    "rmpc_kernel_name", "rmpc_last_passes", "rmpc_debug_sweep", "rmpc_spec_source", "rmpc_spec_name", "rmpc_spec_for", "rmpc_debug_poison_lds",
    "rmpc_debug_fused_stamps", "rmpc_pack_scene_device", "rmpc_solve_batch_scene_device", "rmpc_pack_scene_workspace", "rmpc_solve_batch_packed_device", "rmpc_advance_device", "rmpc_advance_device_flags", "rmpc_retarget_device", "rmpc_advance_obstacles_device", "rmpc_free_space_device",
]

_lib = None


def _source_hash():
    """Hash of the sources next to the library (None when they are not shipped alongside)."""
    import hashlib
    root = os.path.dirname(_HERE)
    paths = [os.path.join(_HERE, "csrc", "rmpc_kernels.hip"), os.path.join(_HERE, "csrc", "rmpc_model.hpp"),
             os.path.join(_HERE, "csrc", "rmpc_spec_gen.hpp"), os.path.join(_HERE, "csrc", "rmpc_arm_fused.hpp"),
             os.path.join(root, "include", "rmpc.h")]
    if not all(os.path.exists(p) for p in paths):
        return None
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def spec_source(desc: dict, name: str) -> str:
    """C++ text of the generated view of ``desc`` (``rmpc_spec_source``; needs no GPU)."""
    L = load_library()
    cd = make_desc(desc, 0)
    n = L.rmpc_spec_source(C.byref(cd), name.encode(), None, 0)
    if n < 0:
        raise RmpcError("rmpc_spec_source failed: " + L.rmpc_last_error().decode())
    buf = C.create_string_buffer(int(n))
    L.rmpc_spec_source(C.byref(cd), name.encode(), buf, n)
    return buf.value.decode()


def spec_for(desc: dict) -> str:
    """Name of the generated view ``rmpc_create`` would select for ``desc`` ("" = runtime row tables); no GPU needed."""
    cd = make_desc(desc, 0)
    return load_library().rmpc_spec_for(C.byref(cd)).decode()


def source_hash() -> str:
    """Hash embedded in the loaded library (``rmpc_source_hash()``)."""
    return load_library().rmpc_source_hash().decode()


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library; raises ``RmpcError`` when it has not been built
    (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RmpcError(f"{path} not found: build the HIP extension first (__graft_entry__.build())")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP
    # runtimes in one process cannot both open the GPU, so when torch is installed
    # it is imported first: the solver library (NEEDED libamdhip64.so.7) then binds
    # to the runtime torch already loaded and device pointers / streams are shared.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.rmpc_version.restype = C.c_int
    L.rmpc_source_hash.restype = C.c_char_p
    L.rmpc_last_error.restype = C.c_char_p
    L.rmpc_desc_size.restype = C.c_int
    L.rmpc_create.restype = C.c_int
    L.rmpc_create.argtypes = [C.POINTER(RmpcDesc), C.c_int, C.POINTER(C.c_void_p)]
    L.rmpc_destroy.restype = None
    L.rmpc_destroy.argtypes = [C.c_void_p]
    L.rmpc_solve_batch.restype = C.c_int
    L.rmpc_solve_batch.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, ip, ip, dp, dp]
    L.rmpc_solve_batch_device.restype = C.c_int
    L.rmpc_solve_batch_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 9
    L.rmpc_workspace_bytes.restype = C.c_int64
    L.rmpc_workspace_bytes.argtypes = [C.POINTER(RmpcDesc), C.c_int]
    L.rmpc_set_warm_start.restype = C.c_int
    L.rmpc_set_warm_start.argtypes = [C.c_void_p, C.c_int]
    L.rmpc_set_profiling.restype = C.c_int
    L.rmpc_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.rmpc_get_profile.restype = C.c_int
    L.rmpc_get_profile.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64), dp, C.POINTER(C.c_int64)]
    L.rmpc_kernel_name.restype = C.c_char_p
    L.rmpc_kernel_name.argtypes = [C.c_int]
    L.rmpc_set_pass_budget.restype = C.c_int
    L.rmpc_set_pass_budget.argtypes = [C.c_void_p, C.c_int]
    L.rmpc_is_fused.restype = C.c_int
    L.rmpc_is_fused.argtypes = [C.c_void_p]
    L.rmpc_fused_kernel_name.restype = C.c_char_p
    L.rmpc_fused_kernel_name.argtypes = [C.c_void_p]
    L.rmpc_is_async.restype = C.c_int
    L.rmpc_is_async.argtypes = [C.c_void_p]
    L.rmpc_last_passes.restype = C.c_int
    L.rmpc_last_passes.argtypes = [C.c_void_p]
    L.rmpc_debug_sweep.restype = C.c_int
    L.rmpc_debug_sweep.argtypes = [C.c_void_p, C.c_int] + [dp] * 9
    L.rmpc_spec_source.restype = C.c_int64
    L.rmpc_spec_source.argtypes = [C.POINTER(RmpcDesc), C.c_char_p, C.c_char_p, C.c_int64]
    L.rmpc_spec_name.restype = C.c_char_p
    L.rmpc_spec_name.argtypes = [C.c_void_p]
    L.rmpc_debug_poison_lds.restype = C.c_int
    L.rmpc_debug_poison_lds.argtypes = [C.c_void_p]
    L.rmpc_spec_for.restype = C.c_char_p
    L.rmpc_spec_for.argtypes = [C.POINTER(RmpcDesc)]
    L.rmpc_debug_fused_stamps.restype = C.c_int
    L.rmpc_debug_fused_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
    L.rmpc_pack_scene_device.restype = C.c_int
    L.rmpc_pack_scene_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(RmpcScene), C.c_void_p, C.c_void_p]
    L.rmpc_pack_scene_workspace.restype = C.c_int
    L.rmpc_pack_scene_workspace.argtypes = [C.c_void_p, C.c_int, C.POINTER(RmpcScene), C.c_void_p]
    L.rmpc_solve_batch_packed_device.restype = C.c_int
    L.rmpc_solve_batch_packed_device.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
    L.rmpc_solve_batch_scene_device.restype = C.c_int
    L.rmpc_solve_batch_scene_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(RmpcScene)] + [C.c_void_p] * 8
    L.rmpc_advance_device.restype = C.c_int
    L.rmpc_advance_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.rmpc_advance_device_flags.restype = C.c_int
    L.rmpc_advance_device_flags.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.rmpc_retarget_device.restype = C.c_int
    L.rmpc_retarget_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(RetargetArgs), C.c_void_p]
    L.rmpc_advance_obstacles_device.restype = C.c_int
    L.rmpc_advance_obstacles_device.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.rmpc_free_space_device.restype = C.c_int
    L.rmpc_free_space_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if L.rmpc_desc_size() != C.sizeof(RmpcDesc):
        raise RmpcError("rmpc_desc layout mismatch between _lib.py and librmpc_hip.so")
    want = _source_hash()
    if os.environ.get("RMPC_ALLOW_STALE"):   # development aid (A/B against an older build); never set by product code
        want = None
    if want is not None and L.rmpc_source_hash().decode() != want:
        raise RmpcError("librmpc_hip.so is stale: built from other sources than the ones next to it "
                        f"({L.rmpc_source_hash().decode()} != {want}); run __graft_entry__.build()")
    _lib = L
    return L


def make_desc(d: dict, device: int = 0) -> RmpcDesc:
    """Descriptor dict (``rmpc_model.yaml``) -> ``rmpc_desc``."""
    o = RmpcDesc()
    o.struct_size = C.sizeof(RmpcDesc)
    o.device = int(device)
    o.robot = d["robot"]; o.N = d["N"]
    o.n, o.nx, o.nu, o.ns, o.npar = d["n"], d["nx"], d["nu"], d["ns"], d["npar"]
    o.dt = d["dt"]
    if len(d["module_kind"]) > MAX_MODULES or len(d["link_frame"]) > MAX_LINKS or \
            len(d["pair_frame"]) > MAX_PAIRS or len(d["joints"]) > MAX_JOINTS:
        raise RmpcError("descriptor exceeds the ABI's fixed capacities")
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
        raise RmpcError("more than %d described rows" % MAX_XROWS)
    o.n_xrows = len(xrows)
    for i, r in enumerate(xrows):
        o.xrow_mod[i], o.xrow_kind[i], o.xrow_a[i], o.xrow_b[i], o.xrow_poff[i] = (int(v) for v in r)
    return o


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _stream_arg(stream):
    """``stream``: an int / hipStream_t handle, or None = the stream the caller's torch ops run on
    (``torch.cuda.current_stream()``; the legacy null stream when torch is absent), so that the solver's
    kernels are ordered after the ops that produced the inputs and before the ops that read the outputs."""
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                stream = torch.cuda.current_stream().cuda_stream
        except ImportError:
            stream = 0
    return C.c_void_p(int(stream or 0))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def free_space_decomposition_device(points, seeds, planes_out, max_radius: float, stream=None):
    """points (B, P, 3), seeds (B, N, 3), planes_out (B, N, K, 4): contiguous fp64 device tensors.
    Device counterpart of ``FreeSpaceDecomposition.compute_constraints`` + ``asdict`` of the
    reference (``robotmpcs/utils/free_space_decomposition.py:79-116``) for B*N seeds at once."""
    L = load_library()
    B, P = int(points.shape[0]), int(points.shape[1])
    N, K = int(seeds.shape[1]), int(planes_out.shape[2])
    st = _stream_arg(stream)
    rc = L.rmpc_free_space_device(B, N, P, K, float(max_radius), C.c_void_p(points.data_ptr()),
                                  C.c_void_p(seeds.data_ptr()), C.c_void_p(planes_out.data_ptr()), st)
    if rc != 0:
        raise RmpcError("rmpc_free_space_device failed: " + L.rmpc_last_error().decode())


class Solver:
    """One handle = one model on one GPU (``rmpc_create`` / ``rmpc_destroy``)."""

    def __init__(self, desc: dict, max_batch: int = 1, device: int = 0):
        self._L = load_library()
        self.desc = desc
        self.cdesc = make_desc(desc, device)
        self.N, self.nx, self.nu, self.ns, self.npar = desc["N"], desc["nx"], desc["nu"], desc["ns"], desc["npar"]
        self.nvar = self.nx + self.ns + self.nu
        self.max_batch = int(max_batch)
        self.device = int(device)
        h = C.c_void_p()
        rc = self._L.rmpc_create(C.byref(self.cdesc), self.max_batch, C.byref(h))
        if rc != 0:
            raise RmpcError("rmpc_create failed: " + self._L.rmpc_last_error().decode())
        self._h = h

    def poison_lds(self):
        """Test aid: NaN patterns into the LDS of every CU (``rmpc_debug_poison_lds``)."""
        self._check(self._L.rmpc_debug_poison_lds(self._h), "rmpc_debug_poison_lds")

    def spec_name(self) -> str:
        """Name of the generated view this handle runs ("" = runtime row tables); ``rmpc_spec_name``."""
        return self._L.rmpc_spec_name(self._h).decode()

    def close(self):
        if getattr(self, "_h", None):
            self._L.rmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RmpcError(f"{what} failed: " + self._L.rmpc_last_error().decode())

    def workspace_bytes(self) -> int:
        return int(self._L.rmpc_workspace_bytes(C.byref(self.cdesc), self.max_batch))

    # -- host buffers (numpy) ------------------------------------------------
    def solve(self, xinit, x0, params):
        xinit = np.ascontiguousarray(xinit, dtype=np.float64).reshape(-1, self.nx)
        B = xinit.shape[0]
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, self.N * self.nvar)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(B, self.N * self.npar)
        z = np.empty((B, self.N, self.nvar))
        exitflag = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
        kkt = np.empty(B); obj = np.empty(B)
        rc = self._L.rmpc_solve_batch(self._h, B, _dp(xinit), _dp(x0), _dp(params), _dp(z), _ip(exitflag),
                                      _ip(iters), _dp(kkt), _dp(obj))
        self._check(rc, "rmpc_solve_batch")
        return dict(z=z, exitflag=exitflag, iters=iters, kkt=kkt, obj=obj)

    # -- device buffers (anything exposing data_ptr(), e.g. torch tensors) ---------
    def solve_device(self, B, xinit, x0, params, z_out, exitflag, iters, kkt, obj, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = _stream_arg(stream)
        rc = self._L.rmpc_solve_batch_device(self._h, int(B), ptr(xinit), ptr(x0), ptr(params), ptr(z_out),
                                             ptr(exitflag), ptr(iters), ptr(kkt), ptr(obj), st)
        self._check(rc, "rmpc_solve_batch_device")

    # -- scenes and closed loop on the device ----------------------------
    def make_scene(self, weights: dict, dyn_radius: float = 0.1, **tensors) -> RmpcScene:
        """``tensors``: goal, r_body, obst, obst_dyn, lower_limits, upper_limits, lower_limits_u,
        upper_limits_u, lower_limits_vel, upper_limits_vel, lin_constrs -- contiguous fp64 device
        tensors (anything with ``data_ptr()``).  ``weights`` = the YAML ``mpc.weights`` block."""
        s = RmpcScene()
        s.struct_size = C.sizeof(RmpcScene)
        for name, t in tensors.items():
            setattr(s, name, C.c_void_p(t.data_ptr()))
        s.dyn_radius = float(dyn_radius)
        s.w = float(weights.get("w", 0.0)); s.wu = float(weights.get("wu", 0.0)); s.ws = float(weights.get("ws", 0.0))
        wc = list(weights.get("wconstr", []))
        for i in range(MAX_MODULES):
            s.wconstr[i] = float(wc[i]) if i < len(wc) else 0.0
        s._keepalive = tensors  # the struct only holds raw pointers
        return s

    def pack_scene_device(self, B, scene: RmpcScene, params_out, stream=None):
        st = _stream_arg(stream)
        rc = self._L.rmpc_pack_scene_device(self._h, int(B), C.byref(scene), C.c_void_p(params_out.data_ptr()), st)
        self._check(rc, "rmpc_pack_scene_device")

    def solve_scene_device(self, B, scene: RmpcScene, xinit, x0, z_out, exitflag, iters, kkt, obj, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = _stream_arg(stream)
        rc = self._L.rmpc_solve_batch_scene_device(self._h, int(B), C.byref(scene), ptr(xinit), ptr(x0), ptr(z_out),
                                                   ptr(exitflag), ptr(iters), ptr(kkt), ptr(obj), st)
        self._check(rc, "rmpc_solve_batch_scene_device")

    def pack_scene_workspace(self, B, scene: RmpcScene, stream=None):
        """first half of ``solve_scene_device``: the scene's parameters into the solver's workspace"""
        rc = self._L.rmpc_pack_scene_workspace(self._h, int(B), C.byref(scene), _stream_arg(stream))
        self._check(rc, "rmpc_pack_scene_workspace")

    def solve_packed_device(self, B, xinit, x0, z_out, exitflag, iters, kkt, obj, stream=None):
        """second half: solve with the parameters ``pack_scene_workspace`` left in the workspace"""
        ptr = lambda t: C.c_void_p(t.data_ptr())
        rc = self._L.rmpc_solve_batch_packed_device(self._h, int(B), ptr(xinit), ptr(x0), ptr(z_out), ptr(exitflag), ptr(iters),
                                                    ptr(kkt), ptr(obj), _stream_arg(stream))
        self._check(rc, "rmpc_solve_batch_packed_device")

    def advance_device(self, B, z_prev, xinit, x0, previous_plan: bool, stream=None, exitflag=None):
        """``exitflag`` (device int32 [B], optional): instances whose solve failed restart from their state."""
        st = _stream_arg(stream)
        ef = C.c_void_p(exitflag.data_ptr()) if exitflag is not None else C.c_void_p(0)
        rc = self._L.rmpc_advance_device_flags(self._h, int(B), C.c_void_p(z_prev.data_ptr()), ef,
                                               C.c_void_p(xinit.data_ptr()), C.c_void_p(x0.data_ptr()),
                                               1 if previous_plan else 0, st)
        self._check(rc, "rmpc_advance_device_flags")

    def retarget_device(self, B, xinit, x0, exitflag, goal, goal_pool, cursor, dwell, x_start, tol, max_dwell=0, counts=None,
                        iters=None, mu_regoal=0.0, stream=None, failrun=None, fail_reset_after=0, settle_vel=0.0,
                        settle_min_dwell=0, lower_limits=None, upper_limits=None):
        """Steady closed loop (``rmpc_retarget_device``): next goal from the instance's pool on arrival / coming to rest /
        dwell time-out; a failed solve keeps its state (reset to the start state only after ``fail_reset_after`` failed
        control steps in a row, or at once when it has left the joint-limit box).  All arrays are device tensors;
        ``counts``: int64 [16]."""
        st = _stream_arg(stream)
        p = lambda t: t.data_ptr() if t is not None else None
        a = RetargetArgs()
        a.struct_size = C.sizeof(RetargetArgs)
        a.pool_len = int(goal_pool.shape[1])
        a.xinit, a.x0, a.exitflag, a.iters, a.goal = p(xinit), p(x0), p(exitflag), p(iters), p(goal)
        a.goal_pool, a.x_start, a.cursor, a.dwell, a.failrun = p(goal_pool), p(x_start), p(cursor), p(dwell), p(failrun)
        a.lower_limits, a.upper_limits = p(lower_limits), p(upper_limits)
        a.tol, a.settle_vel, a.mu_regoal = float(tol), float(settle_vel), float(mu_regoal)
        a.settle_min_dwell, a.max_dwell, a.fail_reset_after, a.reserved = int(settle_min_dwell), int(max_dwell), int(fail_reset_after), 0
        a.counts = p(counts)
        rc = self._L.rmpc_retarget_device(self._h, int(B), C.byref(a), st)
        self._check(rc, "rmpc_retarget_device")

    def advance_obstacles_device(self, obst_dyn, dt: float, arena: float = 0.0, stream=None):
        """``rmpc_advance_obstacles_device``: the moving obstacles [B, nobst, 9] one control step on (device tensor)."""
        st = _stream_arg(stream)
        B, nobst = int(obst_dyn.shape[0]), int(obst_dyn.numel() // (9 * obst_dyn.shape[0]))
        rc = self._L.rmpc_advance_obstacles_device(B, nobst, float(dt), float(arena), C.c_void_p(obst_dyn.data_ptr()), st)
        self._check(rc, "rmpc_advance_obstacles_device")

    def set_warm_start(self, enable: bool):
        """Closed loops: start every solve from the multipliers of the previous solve of the same batch
        (``rmpc_set_warm_start``); the plan itself is warm-started through ``x0`` as in the reference."""
        self._check(self._L.rmpc_set_warm_start(self._h, 1 if enable else 0), "rmpc_set_warm_start")

    def set_pass_budget(self, passes: int):
        """Real-time deadline of a solve in passes (``rmpc_set_pass_budget``; 0 = none): instances still iterating
        when it is spent return their last accepted iterate with exit flag 0."""
        self._check(self._L.rmpc_set_pass_budget(self._h, int(passes)), "rmpc_set_pass_budget")

    def is_fused(self) -> bool:
        """True when a solve is one launch that needs no look from the host (``rmpc_is_fused``)."""
        return bool(self._L.rmpc_is_fused(self._h))

    def is_async(self) -> bool:
        """True when a device solve only enqueues work and returns (``rmpc_is_async``): fused handles, and the pass
        kernels under a pass budget."""
        return bool(self._L.rmpc_is_async(self._h))

    def set_profiling(self, enable: bool):
        self._check(self._L.rmpc_set_profiling(self._h, 1 if enable else 0), "rmpc_set_profiling")

    def get_profile(self):
        ms = (C.c_double * NUM_KERNELS)(); n = (C.c_int64 * NUM_KERNELS)()
        by = (C.c_double * NUM_KERNELS)(); full = (C.c_int64 * NUM_KERNELS)()
        self._check(self._L.rmpc_get_profile(self._h, ms, n, by, full), "rmpc_get_profile")
        names = [self._L.rmpc_kernel_name(i).decode() for i in range(NUM_KERNELS)]
        fused = self._L.rmpc_fused_kernel_name(self._h).decode()
        if fused:
            names[NUM_KERNELS - 1] = fused   # ("k_fused" or "k_fused_arm": the kernel this handle's launches run)
        return {names[i]: dict(total_ms=ms[i], launches=n[i], total_alg_bytes=by[i], full_launch_bytes=full[i])
                for i in range(NUM_KERNELS)}

    def fused_stamps(self, nblocks: int):
        out = np.zeros((nblocks, 8), dtype=np.int64)
        self._check(self._L.rmpc_debug_fused_stamps(self._h, out.ctypes.data_as(C.POINTER(C.c_longlong)), nblocks),
                    "rmpc_debug_fused_stamps")
        return out

    def last_passes(self) -> int:
        return int(self._L.rmpc_last_passes(self._h))

    def debug_sweep(self, xinit, x0, params):
        xinit = np.ascontiguousarray(xinit, dtype=np.float64).reshape(-1, self.nx)
        B = xinit.shape[0]
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, self.N * self.nvar)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(B, self.N * self.npar)
        nv, N = self.nvar, self.N
        nh = int(self.desc["nh"])
        Q = np.zeros((B, N, nv, nv)); q0 = np.zeros((B, N, nv)); q1 = np.zeros((B, N, nv))
        rc_ = np.zeros((B, N, self.nx)); g = np.zeros((B, N, max(nh, 1))); f = np.zeros((B, N))
        rc = self._L.rmpc_debug_sweep(self._h, B, _dp(xinit), _dp(x0), _dp(params), _dp(Q), _dp(q0), _dp(q1),
                                      _dp(rc_), _dp(g), _dp(f))
        self._check(rc, "rmpc_debug_sweep")
        return dict(Q=Q, q0=q0, q1=q1, rc=rc_, g=g[:, :, :nh], f=f)
