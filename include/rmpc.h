/*
 * rmpc.h -- C ABI of the MI355X batched MPC solver (librmpc_hip.so).
 *
 * Drop-in boundary for the one native call the reference planner makes per
 * control step:
 *
 *   forcespro.nlp.Solver.from_directory(dir)            robotmpcs/planner/mpcPlanner.py:73
 *   output, exitflag, info = solver.solve(problem)      robotmpcs/planner/mpcPlanner.py:262
 *     problem = { "xinit": (nx,), "x0": (N*nvar,), "all_parameters": (N*npar,) }   :246-250
 *     output  = { "x01": (nvar,), ... }                                            :265-281
 *
 * The FORCES Pro generated solver is a per-model shared object loaded through
 * ctypes; this library is its replacement with a leading batch axis B.  Plain
 * pointers and sizes only -- no torch / numpy types cross this boundary.
 *
 * Layout at the ABI (row-major, instance-major, exactly x0.flatten() of the
 * reference, mpcPlanner.py:249):
 *   xinit  [B][nx]
 *   x0     [B][N][nvar]      nvar = nx + ns + nu, stage vector z = [x; s; u]   (mpcBase.py:76-80)
 *   params [B][N][npar]      stride npar per stage                             (mpcPlanner.py:91-104)
 *   z_out  [B][N][nvar]      row k = output["x%0*d" % (k+1)]                    (mpcPlanner.py:265-273)
 *
 * Ownership: the caller owns every in/out buffer; the library owns the handle
 * and its device workspace and keeps no caller pointer past return.
 * Threading: a handle is not thread-safe; distinct handles (one per GPU) may be
 * driven from distinct host threads or processes.
 * Errors: functions return 0 on success, non-zero on API misuse or HIP errors
 * (text via rmpc_last_error()).  Solver outcomes are per instance in exitflag:
 *    1 converged, 2 acceptable (feasible, objective stagnated -- cf. IPOPT's
 *    acceptable level), 0 iteration cap reached (plans with flag >= 0 are usable),
 *   <0 failure (-5 factorisation, -6 non-finite, -7 infeasible/diverged,
 *      -8 line search) -- the planner only tests the sign (mpcPlanner.py:263,
 *      examples/boxer_example.py:194).
 */
#ifndef RMPC_H
#define RMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.1.4 (104): rmpc_set_pass_budget; curvature terms of the arms.
 * 0.1.5 (105): rmpc_is_async, rmpc_retarget_device (round 3; the macro was not bumped then).
 * 0.2.0 (200): rmpc_retarget_device takes an rmpc_retarget struct (failed solves keep their state, settled robots,
 *              64-bit counters: NOT source compatible with 0.1.5), rmpc_advance_obstacles_device; rmpc_is_fused is 1
 *              for the arms with 5 .. 7 joints as well (k_fused_arm).
 * 0.2.1 (201): RMPC_MOD_ROWS -- constraint modules given as row descriptions (rmpc_desc grows by the xrow_* arrays at
 *              its end; a descriptor of the 0.2.0 size is still accepted and has no such rows). */
#define RMPC_VERSION 201

#define RMPC_MAX_JOINTS 8
#define RMPC_MAX_LINKS 8
#define RMPC_MAX_PAIRS 4
#define RMPC_MAX_MODULES 8
#define RMPC_NV_MAX 24

/* robot kinds (base_type in the YAML, mpcBase.py:52-60) */
#define RMPC_ROBOT_CHAIN 0     /* holonomic URDF chain: nx = 2n, nu = n (mpcModel.py:65-69) */
#define RMPC_ROBOT_DIFFDRIVE 1 /* diff-drive base, fk.n() == 0 (diff_drive_mpc_model.py:24-41) */

/* inequality modules = YAML class names (models/inequalities/__init__.py) */
#define RMPC_MOD_RADIAL 0
#define RMPC_MOD_LINEAR 1
#define RMPC_MOD_SELFCOLLISION 2
#define RMPC_MOD_JOINTLIMIT 3
#define RMPC_MOD_VELLIMIT 4
#define RMPC_MOD_INPUTLIMIT 5
/* A user-defined module (a YAML constraint name that is none of the six classes above): its rows are described one by
 * one in rmpc_desc::xrow_* -- variants of the six kinds (other frames, other joints, own parameter entries) need no
 * rebuild of the library.  The reference resolves such a name to a user class by getattr
 * (robotmpcs/models/inequalities/InequalityManager.py:17-21) and lets CasADi differentiate it; here the row kinds the
 * kernels evaluate are the vocabulary. */
#define RMPC_MOD_ROWS 6
#define RMPC_MAX_XROWS 32
/* row kinds of RMPC_MOD_ROWS (value h >= 0; r_body = the model's off_r_body entry) */
#define RMPC_ROW_RADIAL 0 /* ||fk_a(q) - c|| - r - r_body, (c, r) = 4 parameters at xrow_poff (RadialConstraints.py:6-23) */
#define RMPC_ROW_LINEAR 1 /* |n . fk_a(q) + d| / ||n|| - r_body, (n, d) = 4 parameters at xrow_poff (LinearConstraints.py:8-40) */
#define RMPC_ROW_SELF 2   /* ||fk_a(q) - fk_b(q)|| - 2 r_body (SelfCollisionAvoidanceConstraints.py:8-27) */
#define RMPC_ROW_VAR 3    /* b = +1: z_a - limit, b = -1: limit - z_a, limit = the parameter at xrow_poff
                             (JointLimitConstraints.py:8-31, InputLimitConstraints.py:7-29) */

#define RMPC_JOINT_FIXED 0
#define RMPC_JOINT_REVOLUTE 1
#define RMPC_JOINT_PRISMATIC 2

/* Model descriptor: the numeric content of <solver dir>/rmpc_model.yaml, i.e.
 * what the reference encodes in the generated solver (mpcModel.py:74-126). */
typedef struct rmpc_desc {
  int32_t struct_size; /* sizeof(rmpc_desc), checked */
  int32_t device;      /* HIP device ordinal */
  int32_t robot;
  int32_t N;           /* horizon (time_horizon) */
  int32_t n, nx, nu, ns, npar;
  double dt;           /* time_step; integrator ERK2 (explicit midpoint), 5 nodes */
  int32_t n_modules;
  int32_t module_kind[RMPC_MAX_MODULES];
  int32_t nobst;       /* number_obstacles */
  int32_t n_links;
  int32_t link_frame[RMPC_MAX_LINKS];
  int32_t n_pairs;
  int32_t pair_frame[RMPC_MAX_PAIRS][2];
  int32_t end_frame;
  int32_t n_joints;
  int32_t joint_type[RMPC_MAX_JOINTS];
  int32_t joint_dof[RMPC_MAX_JOINTS];
  double joint_xyz[RMPC_MAX_JOINTS][3];
  double joint_rot[RMPC_MAX_JOINTS][9];
  double joint_axis[RMPC_MAX_JOINTS][3];
  /* offsets into one stage's parameter slice (paramMap.yaml), -1 = absent */
  int32_t off_r_body, off_obst, off_lin, off_lower, off_upper, off_lower_u,
      off_upper_u, off_lower_vel, off_upper_vel, off_wu, off_goal, off_wgoal,
      off_wconstr, off_ws;
  int32_t has_goal, has_avoid;
  double lb[RMPC_NV_MAX], ub[RMPC_NV_MAX]; /* z bounds, +-inf allowed (mpcModel.py:91-104) */
  /* solver options */
  int32_t max_iter;
  double tol_stat, tol_eq, tol_ineq, tol_comp;
  double mu0;
  int32_t acc_iters;    /* acceptable termination: consecutive stagnant feasible iterations (0 = off, default 8) */
  double acc_obj_tol;   /* ... relative objective change (default 1e-8) */
  int32_t ls_max;       /* step halvings allowed in one line search (default 25); an instance that exhausts them
                           stops with exitflag -8 and its current iterate.  Small values bound the number of passes
                           of a real-time solve (examples/fleet_loop.py) */
  /* ---- 0.2.1: rows of the RMPC_MOD_ROWS modules, in row order; xrow_mod = index into module_kind.  The rows of one
   * module must all be on states (RADIAL / LINEAR / SELF, VAR with a < nx) or all on inputs (VAR with a >= nx + ns): the
   * stage-1 neutralisation and the inverse-barrier objective are per module (InequalityManager.py:25-33).  RADIAL and
   * LINEAR rows address their parameters as entries of the obstacle / plane lists: xrow_poff - off_obst (off_lin) must
   * be a multiple of 4 in [0, 252] (an absent list starts at the first such row). */
  int32_t n_xrows;
  int32_t xrow_mod[RMPC_MAX_XROWS];
  int32_t xrow_kind[RMPC_MAX_XROWS];
  int32_t xrow_a[RMPC_MAX_XROWS];
  int32_t xrow_b[RMPC_MAX_XROWS];
  int32_t xrow_poff[RMPC_MAX_XROWS];
} rmpc_desc;

typedef struct rmpc_handle rmpc_handle;

int rmpc_version(void);
/* sha256 (first 16 hex digits) of the sources this binary was built from (csrc/rmpc_kernels.hip,
 * csrc/rmpc_model.hpp, csrc/rmpc_spec_gen.hpp, include/rmpc.h), embedded by __graft_entry__.build(); the Python binding
 * refuses a library whose hash differs from the sources next to it. */
const char *rmpc_source_hash(void);
const char *rmpc_last_error(void);
int rmpc_desc_size(void);

/* Replaces forcespro.nlp.Solver.from_directory (mpcPlanner.py:73): validates
 * the descriptor, selects the device and allocates the HBM workspace for up to
 * max_batch instances. */
int rmpc_create(const rmpc_desc *desc, int max_batch, rmpc_handle **out);
void rmpc_destroy(rmpc_handle *h);

/* Replaces solver.solve(problem) (mpcPlanner.py:262) for B instances; host
 * pointers, blocks until the results are in host memory. */
int rmpc_solve_batch(rmpc_handle *h, int B, const double *xinit, const double *x0,
                     const double *params, double *z_out, int32_t *exitflag,
                     int32_t *iters, double *kkt_res, double *obj);

/* Same, device pointers (e.g. torch tensor data_ptr()).  All work is enqueued on `stream`
 * (a hipStream_t; NULL = the legacy null stream, i.e. ordered with the caller's default-stream
 * work such as the torch ops that produced the inputs).  Returns when the iteration loop has
 * drained and the final kernel (the one that writes d_z_out ... d_obj) is ENQUEUED: the outputs
 * are valid for later work on the same stream, or for the host after a stream synchronize. */
int rmpc_solve_batch_device(rmpc_handle *h, int B, const double *d_xinit,
                            const double *d_x0, const double *d_params,
                            double *d_z_out, int32_t *d_exitflag, int32_t *d_iters,
                            double *d_kkt_res, double *d_obj, void *stream);

/* Warm start of the multipliers for closed loops (no counterpart in the reference, which warm-starts the plan
 * only: setX0 / shiftHorizon, mpcPlanner.py:215-236).  mode 1: a solve of B instances that follows a finished solve
 * of the same B instances on this handle starts instance b from the multipliers of ITS previous solve, shifted by one
 * stage like the plan (lambda_k <- lambda_{k+1}, nu_k <- nu_{k+1}, last stage repeated), with mu = clamp(1000 *
 * previous final mu, 1e-6, mu0), slacks t = max(g(x0), 1e-4) and lambda = max(previous, mu / t); an instance whose
 * previous solve failed starts from zero multipliers and mu0.  mode 0 (default): every solve starts cold.  Changing the mode, or solving another
 * batch size, forgets the stored multipliers. */
int rmpc_set_warm_start(rmpc_handle *h, int mode);

/* Real-time deadline of a solve, in passes (0 = none, the default).  A pass is one evaluation of the whole horizon:
 * the start point, every trial point of a line search, every step recomputed with the Gauss-Newton blocks -- the
 * unit the device pays for (an iteration costs one pass plus one per backtracking).  An instance still iterating
 * when the budget is spent returns its last accepted iterate with exit flag 0, exactly as at the iteration limit;
 * instances that finish earlier are unaffected.  No reference counterpart (FORCES Pro offers a wall-clock
 * `solver_timeout`); used by the fleet loop, where a few instances in hopeless states would otherwise hold the
 * control step of thousands.  Budgets below max_iter + 1 also bound the iterations. */
int rmpc_set_pass_budget(rmpc_handle *h, int passes);

/* 1 when the solves of this handle run as ONE launch that needs no look from the host (k_fused: point robot and
 * diff-drive base, N <= 32) -- rmpc_solve_batch_device then only enqueues work on the stream and returns; 0 when they
 * run as pass kernels, whose host loop reads a counter every few passes and returns when the batch is done. */
int rmpc_is_fused(const rmpc_handle *h);
/* the kernel behind a fused handle: "k_fused" (point robot, diff-drive base: two instances per wavefront, lane = stage),
 * "k_fused_arm" (arms with 5 .. 7 joints: one instance per wavefront, a stage per two lanes), "" for the pass kernels */
const char *rmpc_fused_kernel_name(const rmpc_handle *h);

/* 1 when rmpc_solve_batch_device (and the scene / packed variants) of this handle only ENQUEUES work on the stream and
 * returns -- no look from the host, the solve is ordered with the caller's stream like any kernel: fused handles
 * always, handles on the pass kernels (the arm, N > 32) when a pass budget is set (the budgeted passes are enqueued
 * whole; kernels leave at once when no instance iterates any more).  Without a budget the pass kernels' host loop
 * reads a counter every few passes, because only it can know how many passes to enqueue (a solve then returns when
 * the batch is done; the reference's solver.solve() is synchronous too, mpcPlanner.py:262). */
int rmpc_is_async(const rmpc_handle *h);

/* Workspace size in bytes for a given descriptor / batch (no allocation). */
int64_t rmpc_workspace_bytes(const rmpc_desc *desc, int max_batch);

/* Per-kernel timing with HIP events on the solver's stream.
 * kernels: 0 pack, 1 sweep, 2 riccati, 3 step, 4 unpack (pass kernels: large horizons, the arm),
 * 5 fused (whole solves inside one wavefront, N <= 32: point robot, diff-drive base, arms with 5 .. 7 joints --
 *   rmpc_fused_kernel_name tells which kernel). */
#define RMPC_NUM_KERNELS 6
int rmpc_set_profiling(rmpc_handle *h, int enable);
/* total_ms / launches: summed HIP-event durations and launch counts per kernel;
 * total_alg_bytes: algorithmic bytes of those launches, counting only the lanes
 * that were still active in each launch; full_launch_bytes: algorithmic bytes
 * of one launch with every lane of a max_batch batch active. */
int rmpc_get_profile(rmpc_handle *h, double *total_ms, int64_t *launches,
                     double *total_alg_bytes, int64_t *full_launch_bytes);
const char *rmpc_kernel_name(int idx);
/* number of sweep/riccati/step passes of the last solve, and instance-iterations */
int rmpc_last_passes(rmpc_handle *h);

/* ---- next rows of the hot path: inputs produced on the device ---------- */

/* Compact scene of B instances; every pointer is a DEVICE pointer and may be NULL (field left at
 * zero).  Device counterpart of MPCPlanner.reset() + setGoalReaching / setRadialConstraints /
 * setLinearConstraints / setJointLimits / setInputLimits / setVelLimits / setConstraintAvoidance /
 * updateDynamicObstacles (robotmpcs/planner/mpcPlanner.py:83-210). */
typedef struct rmpc_scene {
  int32_t struct_size;        /* sizeof(rmpc_scene) */
  const double *goal;         /* [B][3]            setGoalReaching, zero padded (:197-204) */
  const double *r_body;       /* [B]               (:122,137,165) */
  const double *obst;         /* [B][nobst][4]     static obstacles: position, radius (:120-133) */
  const double *obst_dyn;     /* [B][nobst][9]     position, velocity, acceleration; stage k predicted at dt*k (:144-161) */
  double dyn_radius;          /*                   radius of the predicted obstacles, self._r = 0.1 (:121) */
  const double *lower_limits, *upper_limits;         /* [B][n]   (:167-175) */
  const double *lower_limits_u, *upper_limits_u;     /* [B][nu]  (:187-195) */
  const double *lower_limits_vel, *upper_limits_vel; /* [B][2]   (:177-185) */
  const double *lin_constrs;  /* [B][N][nobst][4]  planes per stage (:135-141) */
  double w, wu, ws;           /*                   weights["w"], ["wu"], ["ws"], broadcast by reset() (:92-104) */
  double wconstr[RMPC_MAX_MODULES]; /*             weights["wconstr"] (:206-210) */
} rmpc_scene;

/* all_parameters [B][N][npar] (ABI layout) from a scene; bit-identical to the host packer. */
int rmpc_pack_scene_device(rmpc_handle *h, int B, const rmpc_scene *scene, double *d_params, void *stream);

/* rmpc_solve_batch_device with the parameters expanded from a scene straight into the solver's
 * workspace: the B*N*npar array never exists in the ABI layout. */
int rmpc_solve_batch_scene_device(rmpc_handle *h, int B, const rmpc_scene *scene, const double *d_xinit,
                                  const double *d_x0, double *d_z_out, int32_t *d_exitflag, int32_t *d_iters,
                                  double *d_kkt_res, double *d_obj, void *stream);

/* The two halves of rmpc_solve_batch_scene_device, for callers that drive several handles on several streams: the
 * packing kernels of all of them can be enqueued before the first solve (a fused solve fills every SIMD for
 * milliseconds; a small kernel enqueued behind it on another stream waits for a free slot first).
 * rmpc_solve_batch_packed_device solves with the parameters rmpc_pack_scene_workspace left in the workspace (same
 * handle, same B, same stream). */
int rmpc_pack_scene_workspace(rmpc_handle *h, int B, const rmpc_scene *scene, void *stream);
int rmpc_solve_batch_packed_device(rmpc_handle *h, int B, const double *d_xinit, const double *d_x0, double *d_z_out,
                                   int32_t *d_exitflag, int32_t *d_iters, double *d_kkt_res, double *d_obj, void *stream);

/* Closed loop between two solves, on the device: xinit <- Phi(xinit, u_1 of the previous plan)
 * with the model's own ERK2 map (the plant of the examples' env.step), and the warm start
 * x0 <- shifted plan (shiftHorizon, mpcPlanner.py:215-226) when previous_plan != 0, else the new
 * state repeated over the horizon with zero controls (setX0 "current_state", :228-232). */
int rmpc_advance_device(rmpc_handle *h, int B, const double *d_z_prev, double *d_xinit, double *d_x0,
                        int previous_plan, void *stream);
/* The same with the exit flags of the solve that produced d_z_prev ([B], device): an instance whose solve failed
 * (exitflag < 0) restarts from its new state ("current_state" initialisation) whatever previous_plan says -- the
 * closed-loop fallback of the reference's boxer example (examples/boxer_example.py:194-198). */
int rmpc_advance_device_flags(rmpc_handle *h, int B, const double *d_z_prev, const int32_t *d_exitflag, double *d_xinit,
                              double *d_x0, int previous_plan, void *stream);

/* Steady closed loop (fleet harness; the examples of the reference hand the planner its next goal whenever the driver
 * has one -- setGoalReaching every control step with the next waypoint, examples/boxer_example_global.py:203-212).
 * Called after rmpc_advance_device_flags, once per control step.  Every pointer is a DEVICE pointer.
 *  - an instance whose end link is within `tol` of its goal has ARRIVED; one that has come to rest (largest joint /
 *    wheel speed below settle_vel after at least settle_min_dwell control steps on this goal; settle_vel = 0: off) has
 *    SETTLED -- with the reference's objective (N w / h on the first row of a module, constraint_avoidance.py:22-31) a
 *    goal next to an obstacle is an equilibrium at a distance, not a point that is reached; one that has spent
 *    max_dwell control steps on its goal (0: no limit) has TIMED OUT.  All three take the next goal of their pool
 *    goal_pool [B][pool_len][3] (cursor [B], dwell [B]: int32, zero-initialised by the caller).
 *  - an instance whose solve FAILED (exitflag < 0) keeps its state: the reference prints the flag and applies the action
 *    it got (mpcPlanner.py:263-264), the next solve starts cold from the new state (rmpc_advance_device_flags =
 *    examples/boxer_example.py:194-198).  Only after fail_reset_after failed control steps IN A ROW (failrun [B], int32,
 *    zero-initialised; 0: never) is it put back to x_start [B][nx] with a cold plan and its next goal: a RESET.  The same
 *    happens at once to an instance whose configuration has left the joint-limit box lower_limits / upper_limits [B][n]
 *    (may be NULL: no check) by more than 5 % of its width: the plant of the loop is the bare integrator, the examples'
 *    simulator would have stopped the joint at its limit (counted in counts[12] as well).
 *  - mu_regoal > 0 (with rmpc_set_warm_start(1)): the next solve of an instance that has just taken a new goal keeps its
 *    multipliers but restarts its barrier parameter from mu_regoal; 0: plain warm start.
 *  - goal [B][3] is the array the scene (rmpc_scene.goal) points at.
 *  - counts (may be NULL): twelve int64 counters, incremented: [0] arrivals, [1] settled, [2] time-outs, [3] resets,
 *    [4..7] the control step's exit flags (1, 2, 0, < 0), [8] sum of iters (may be NULL), [9] sum of the distance to the
 *    goal at the hand-overs in micrometres, [10] hand-overs counted in [9], [11] instances inside a run of failed
 *    solves this step, [12] resets because the robot had left its workspace, [13..15] reserved -- loop statistics
 *    without a host read per control step (64-bit: a loop may run for days). */
typedef struct rmpc_retarget {
  int32_t struct_size;            /* sizeof(rmpc_retarget) */
  int32_t pool_len;
  double *xinit, *x0;             /* [B][nx], [B][N][nvar] */
  const int32_t *exitflag, *iters;
  double *goal;
  const double *goal_pool, *x_start;
  const double *lower_limits, *upper_limits;
  int32_t *cursor, *dwell, *failrun;
  double tol, settle_vel, mu_regoal;
  int32_t settle_min_dwell, max_dwell, fail_reset_after, reserved;
  int64_t *counts;
} rmpc_retarget;
int rmpc_retarget_device(rmpc_handle *h, int B, const rmpc_retarget *r, void *stream);

/* The moving obstacles between two control steps (what the examples' simulator does before the driver hands the planner
 * ob[nx:], mpcPlanner.py:243-244): d_obst_dyn [B][nobst][9] = position, velocity, acceleration (rmpc_scene.obst_dyn):
 * pos += vel dt + acc dt^2 / 2, vel += acc dt.  arena > 0: an obstacle that leaves [-arena, arena] in x or y comes back
 * with that velocity component mirrored.  Needs no handle. */
int rmpc_advance_obstacles_device(int B, int nobst, double dt, double arena, double *d_obst_dyn, void *stream);

/* Free-space decomposition on the device: for each of the B*N seed points
 * (e.g. the planned lidar position of instance b at stage k) at most K half-planes
 * [a(3), d] from instance b's point cloud of P <= 64 points, greedy nearest-point rule and dummy
 * planes of robotmpcs/utils/free_space_decomposition.py:79-116.  d_points [B][P][3],
 * d_seeds [B][N][3], d_planes [B][N][K][4] = the lin_constrs field of rmpc_scene
 * (setLinearConstraints, mpcPlanner.py:135-141; called N times per control step by
 * examples/boxer_example.py:193-203).  Needs no handle. */
int rmpc_free_space_device(int B, int N, int P, int K, double max_radius, const double *d_points,
                           const double *d_seeds, double *d_planes, void *stream);

/* Debug / parity hooks (used by tests through the same ABI): evaluate one
 * stage-parallel sweep at z = x0 (first-pass semantics) and return the
 * condensed stage blocks in instance-major order.
 * out_Q [B][N][nvar*nvar] dense symmetric, out_q0/out_q1 [B][N][nvar],
 * out_rc [B][N][nx], out_g [B][N][nh], out_f [B][N]. */
int rmpc_debug_sweep(rmpc_handle *h, int B, const double *xinit, const double *x0,
                     const double *params, double *out_Q, double *out_q0,
                     double *out_q1, double *out_rc, double *out_g, double *out_f);

/* Generated solvers.  The reference has FORCES Pro generate C code for ONE problem (mpcModel.py:139-160
 * generateSolver, examples/makeSolver.py); here the kernels exist in two forms: over runtime row tables (any
 * descriptor) and over "generated views" -- the same tables as compile-time constants, so that the row loops of
 * the hot kernels become straight-line code (csrc/rmpc_spec_gen.hpp: views of the point-robot configurations).
 * rmpc_spec_source writes the C++ text of the view of `desc` (struct `name`) into out (cap bytes incl. the
 * terminating 0) and returns the size it needs, or -1; scripts/gen_specs.py assembles the header from it.
 * rmpc_create selects a view when every table entry equals the descriptor's (RMPC_NO_SPEC=1 in the environment
 * forces the runtime tables; measured: same throughput, one batch alone 6 % faster on the point robot, 20 % slower
 * on the boxer, whose views are therefore not generated: DESIGN.md 5.1); rmpc_spec_name returns the view's name
 * ("" = runtime tables). */
int64_t rmpc_spec_source(const rmpc_desc *desc, const char *name, char *out, int64_t cap);
const char *rmpc_spec_name(rmpc_handle *h);
const char *rmpc_spec_for(const rmpc_desc *desc);   /* the view rmpc_create would select (no GPU needed) */

/* Test aid: overwrites the LDS of every CU of the handle's device with NaN patterns (tests/test_gpu_parity.py:
 * a solve must not depend on what a previous kernel left in LDS). */
int rmpc_debug_poison_lds(rmpc_handle *h);

/* Development aid: per-block phase cycle counts of the last fused launch (8 words per block: sweep, decisions,
 * recursion, step, total, passes, start, -); all zero unless the library was built with -DRMPC_STAMPS. */
int rmpc_debug_fused_stamps(rmpc_handle *h, long long *out, int nblocks);

#ifdef __cplusplus
}
#endif
#endif
