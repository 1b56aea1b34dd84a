/*
 * rmpc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the NLP that maxspahn/robot_mpcs builds with CasADi
 * and hands to FORCES Pro, plus a restatement of the published FORCES-NLP
 * algorithm class (primal-dual interior point on the multistage NLP with a
 * stage-wise block factorisation of the KKT system; Zanelli, Domahidi, Jerez,
 * Morari, "FORCES NLP: an efficient implementation of interior-point methods
 * for multistage nonlinear nonconvex programs", Int. J. Control 2017).
 *
 * PARITY UNPINNED: the reference holds no tests / golden vectors for this path
 *  and its solver (forcespro, proprietary; casadi 3.5.5;
 * forwardkinematics 1.1.5) is absent from this image, so the oracle cannot be
 * checked against reference outputs.  It is pinned instead by (i) an independent
 * numpy restatement of the model functions (oracle/nlp_numpy.py) with
 * finite-difference and sympy checks and (ii) an independent scipy SLSQP solve
 * of the same NLP (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Reference files restated (all under /root/reference/robotmpcs):
 *   models/mpcBase.py:52-101        dims, z layout, obstacle distances
 *   models/mpcModel.py:65-126       double integrator, NLP assembly, ERK2 x 5 nodes
 *   models/diff_drive_mpc_model.py:24-41   unicycle dynamics
 *   models/inequalities/<module>.py       inequality modules (YAML order)
 *   models/objectives/<module>.py         GoalReaching, ConstraintAvoidance, wu, ws
 *   utils/utils.py:48-52            point_to_plane
 */
#ifndef RMPC_ORACLE_H
#define RMPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_JOINTS 8
#define ORC_MAX_LINKS 8
#define ORC_MAX_PAIRS 4
#define ORC_MAX_MODULES 8
#define ORC_NX_MAX 16
#define ORC_NW_MAX 8
#define ORC_NV_MAX 24
#define ORC_NH_MAX 64
#define ORC_MR_MAX (ORC_NH_MAX + 2 * ORC_NV_MAX)

/* robot kinds (mpcBase.py:52-60 base_type) */
#define ORC_ROBOT_CHAIN 0     /* holonomic URDF chain: nx = 2n, nu = n */
#define ORC_ROBOT_DIFFDRIVE 1 /* diff-drive base, fk.n() == 0: n = 3, nx = 8, nu = 2 */

/* inequality module kinds, names = YAML class names (inequalities/__init__.py) */
#define ORC_MOD_RADIAL 0
#define ORC_MOD_LINEAR 1
#define ORC_MOD_SELFCOLLISION 2
#define ORC_MOD_JOINTLIMIT 3
#define ORC_MOD_VELLIMIT 4
#define ORC_MOD_INPUTLIMIT 5
/* a module given as row descriptions (include/rmpc.h RMPC_MOD_ROWS; rows in orc_desc::xrow_*) */
#define ORC_MOD_ROWS 6
#define ORC_MAX_XROWS 32
#define ORC_ROW_RADIAL 0
#define ORC_ROW_LINEAR 1
#define ORC_ROW_SELF 2
#define ORC_ROW_VAR 3

/* joint types of the kinematic chain */
#define ORC_JOINT_FIXED 0
#define ORC_JOINT_REVOLUTE 1
#define ORC_JOINT_PRISMATIC 2

typedef struct orc_desc {
  int32_t robot;
  int32_t N;
  int32_t n, nx, nu, ns, npar;
  double dt;
  /* constraint modules in YAML order (InequalityManager.py:15-23) */
  int32_t n_modules;
  int32_t module_kind[ORC_MAX_MODULES];
  int32_t nobst;
  /* kinematics: frame index f = frame attached to the child link of joint f */
  int32_t n_links;
  int32_t link_frame[ORC_MAX_LINKS];
  int32_t n_pairs;
  int32_t pair_frame[ORC_MAX_PAIRS][2];
  int32_t end_frame;
  int32_t n_joints;
  int32_t joint_type[ORC_MAX_JOINTS];
  int32_t joint_dof[ORC_MAX_JOINTS]; /* index into q, -1 for fixed */
  double joint_xyz[ORC_MAX_JOINTS][3];
  double joint_rot[ORC_MAX_JOINTS][9]; /* row-major R(rpy) of the joint origin */
  double joint_axis[ORC_MAX_JOINTS][3];
  /* parameter offsets inside one stage slice p_k (-1 = absent); paramMap.yaml */
  int32_t off_r_body, off_obst, off_lin, off_lower, off_upper, off_lower_u,
      off_upper_u, off_lower_vel, off_upper_vel, off_wu, off_goal, off_wgoal,
      off_wconstr, off_ws;
  int32_t has_goal, has_avoid; /* objectives present (ObjectiveManager.py:18-26) */
  double lb[ORC_NV_MAX], ub[ORC_NV_MAX]; /* z bounds, order x,s,u (mpcModel.py:91-104) */
  /* solver options */
  int32_t max_iter;
  double tol_stat, tol_eq, tol_ineq, tol_comp;
  double mu0;
  int32_t acc_iters;
  double acc_obj_tol;
  int32_t ls_max; /* halvings allowed in one line search (<= 0: ORC_LS_MAX) */
  /* rows of the ORC_MOD_ROWS modules (include/rmpc.h: xrow_*): module index, row kind, a, b, parameter offset */
  int32_t n_xrows;
  int32_t xrow_mod[ORC_MAX_XROWS], xrow_kind[ORC_MAX_XROWS], xrow_a[ORC_MAX_XROWS], xrow_b[ORC_MAX_XROWS],
      xrow_poff[ORC_MAX_XROWS];
} orc_desc;

/* Per-stage model evaluation (dense).  All matrices row-major.
 * H is the generalised Gauss-Newton Hessian of the stage cost.
 * g holds the general rows (YAML order, slack added when ns==1) followed by
 * the finite lower bounds (z-lb) and the finite upper bounds (ub-z).
 * fixed_state != 0: stage 1 (state pinned to xinit), state-only unsoftened rows are
 * neutralised (value 1, zero gradient, no inverse-barrier term).
 * Returns the number of rows m, or <0 on error. */
int orc_eval_stage(const orc_desc *d, const double *z, const double *p,
                   int want_derivs, double *f, double *gf, double *H,
                   double *g, double *Jg, double *xnext, double *A, double *Bm, int fixed_state);

/* number of general rows nh and total rows m (incl. finite bounds) */
int orc_num_rows(const orc_desc *d, int *nh, int *m);

/* forward kinematics of frame `frame` (position, 3 x n Jacobian row-major) */
int orc_fk(const orc_desc *d, const double *q, int frame, double *pos, double *Jp);
/* Cacc [n*n] += sum_c F_c d2 pos_c / dq dq (holonomic chain) */
int orc_last_passes(void);
int orc_stage_curvature(const orc_desc *d, const double *z, const double *p, const double *lam, int fixed_state,
                        double *Cout);
int orc_fk_curv(const orc_desc *d, const double *q, int frame, const double *F, double *Cacc);

typedef struct orc_stats {
  int32_t exitflag; /* 1 converged, 2 acceptable (objective stagnated at a feasible point), 0 iteration cap, <0 failure */
  int32_t iters;
  double res_stat, res_eq, res_ineq, res_comp;
  double obj;
  double mu;
} orc_stats;

/* One instance.  xinit[nx], x0[N*nvar] (stage-major), params[N*npar].
 * zout[N*nvar].  trace (optional, may be NULL): per-iteration
 * [res_stat,res_eq,res_ineq,res_comp,mu,alpha,obj,ls_trials] * (max_iter+1). */
int orc_solve(const orc_desc *d, const double *xinit, const double *x0,
              const double *params, double *zout, orc_stats *st, double *trace);

/* The same with a warm start of the multipliers (see rmpc_oracle.c): lam_w [N][m], nu_w [N][nx], mu_w from the
 * previous solve of this instance (its lam_out / nu_out / st->mu), shifted by one stage inside. */
int orc_solve_warm(const orc_desc *d, const double *xinit, const double *x0, const double *params, double *zout,
                   orc_stats *st, const double *lam_w, const double *nu_w, double mu_w, double *lam_out,
                   double *nu_out);

/* Batch, instance-major arrays; nthreads <= 0 -> OpenMP default. */
int orc_solve_batch(const orc_desc *d, int B, const double *xinit,
                    const double *x0, const double *params, double *zout,
                    orc_stats *st, int nthreads);

/* discrete dynamics only (plant model for closed-loop tests) */
int orc_dynamics(const orc_desc *d, const double *x, const double *u, double *xnext);

int orc_desc_size(void);

#ifdef __cplusplus
}
#endif
#endif
