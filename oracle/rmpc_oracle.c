/*
 * rmpc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See rmpc_oracle.h for scope, the "parity unpinned" statement and the list
 * of reference files restated.  Dense, runtime-dimensioned, deliberately plain.
 */
#include "rmpc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NVM ORC_NV_MAX
#define NXM ORC_NX_MAX
#define NWM ORC_NW_MAX
#define MRM ORC_MR_MAX

int orc_desc_size(void) { return (int)sizeof(orc_desc); }

static int nvar_of(const orc_desc *d) { return d->nx + d->ns + d->nu; }

/* ------------------------------------------------------------------ */
/* small dense helpers                                                  */
/* ------------------------------------------------------------------ */
static void mat3_mul(const double *a, const double *b, double *c) {
  double r[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
      r[3 * i + j] = s;
    }
  memcpy(c, r, sizeof r);
}
static void mat3_vec(const double *a, const double *v, double *o) {
  double r[3];
  for (int i = 0; i < 3; i++) r[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
/* Rodrigues rotation about unit axis k by angle th */
static void rot_axis(const double *k, double th, double *R) {
  double c = cos(th), s = sin(th), v = 1.0 - c;
  R[0] = c + k[0] * k[0] * v;        R[1] = k[0] * k[1] * v - k[2] * s; R[2] = k[0] * k[2] * v + k[1] * s;
  R[3] = k[1] * k[0] * v + k[2] * s; R[4] = c + k[1] * k[1] * v;        R[5] = k[1] * k[2] * v - k[0] * s;
  R[6] = k[2] * k[0] * v - k[1] * s; R[7] = k[2] * k[1] * v + k[0] * s; R[8] = c + k[2] * k[2] * v;
}

/* ------------------------------------------------------------------ */
/* forward kinematics (restates the .fk(q, root, link, positionOnly=True)
 * call sites: mpcBase.py:89-94, goal_reaching.py:22-27,
 * LinearConstraints.py:30-35, SelfCollisionAvoidanceConstraints.py:23-24) */
/* ------------------------------------------------------------------ */
static int fk_axes(const orc_desc *d, const double *q, int frame, double *pos, double *Jp, double (*ax)[3]);

int orc_fk(const orc_desc *d, const double *q, int frame, double *pos, double *Jp) {
  return fk_axes(d, q, frame, pos, Jp, 0);
}

/* ax (optional): world axis of the revolute joint behind each degree of freedom on the chain to `frame`
 * (zero for prismatic joints and for degrees of freedom beyond the frame) */
static int fk_axes(const orc_desc *d, const double *q, int frame, double *pos, double *Jp, double (*ax)[3]) {
  const int n = d->n;
  if (frame < 0 || frame >= d->n_joints) return -1;
  if (ax) memset(ax, 0, sizeof(double) * 3 * n);
  for (int i = 0; i < 3 * n; i++) Jp[i] = 0.0;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
  if (d->robot == ORC_ROBOT_DIFFDRIVE) {
    /* base pose (x, y, theta) prefixes a chain of fixed joints (fk.n() == 0) */
    for (int j = 0; j <= frame; j++) {
      if (d->joint_type[j] != ORC_JOINT_FIXED) return -2;
      double t[3];
      mat3_vec(R, d->joint_xyz[j], t);
      o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
      mat3_mul(R, d->joint_rot[j], R);
    }
    double c = cos(q[2]), s = sin(q[2]);
    pos[0] = q[0] + c * o[0] - s * o[1];
    pos[1] = q[1] + s * o[0] + c * o[1];
    pos[2] = o[2];
    Jp[0 * n + 0] = 1.0;
    Jp[1 * n + 1] = 1.0;
    Jp[0 * n + 2] = -s * o[0] - c * o[1];
    Jp[1 * n + 2] = c * o[0] - s * o[1];
    return 0;
  }
  double oj[ORC_MAX_JOINTS][3], aj[ORC_MAX_JOINTS][3];
  for (int j = 0; j <= frame; j++) {
    double t[3];
    mat3_vec(R, d->joint_xyz[j], t);
    o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
    mat3_mul(R, d->joint_rot[j], R);
    mat3_vec(R, d->joint_axis[j], aj[j]);
    oj[j][0] = o[0]; oj[j][1] = o[1]; oj[j][2] = o[2];
    if (d->joint_type[j] == ORC_JOINT_REVOLUTE) {
      double Rq[9];
      rot_axis(d->joint_axis[j], q[d->joint_dof[j]], Rq);
      mat3_mul(R, Rq, R);
    } else if (d->joint_type[j] == ORC_JOINT_PRISMATIC) {
      double qq = q[d->joint_dof[j]];
      o[0] += aj[j][0] * qq; o[1] += aj[j][1] * qq; o[2] += aj[j][2] * qq;
    }
  }
  pos[0] = o[0]; pos[1] = o[1]; pos[2] = o[2];
  for (int j = 0; j <= frame; j++) {
    int dof = d->joint_dof[j];
    if (d->joint_type[j] == ORC_JOINT_REVOLUTE) {
      double r[3] = {o[0] - oj[j][0], o[1] - oj[j][1], o[2] - oj[j][2]};
      Jp[0 * n + dof] = aj[j][1] * r[2] - aj[j][2] * r[1];
      Jp[1 * n + dof] = aj[j][2] * r[0] - aj[j][0] * r[2];
      Jp[2 * n + dof] = aj[j][0] * r[1] - aj[j][1] * r[0];
      if (ax) { ax[dof][0] = aj[j][0]; ax[dof][1] = aj[j][1]; ax[dof][2] = aj[j][2]; }
    } else if (d->joint_type[j] == ORC_JOINT_PRISMATIC) {
      Jp[0 * n + dof] = aj[j][0];
      Jp[1 * n + dof] = aj[j][1];
      Jp[2 * n + dof] = aj[j][2];
    }
  }
  return 0;
}

/* Second derivatives of a frame position, contracted with a vector: C[a][b] += sum_c F_c d2 p_c / dq_a dq_b
 * (C: n x n, symmetric).  For degrees of freedom a before b on the chain, dJ_b/dq_a = axis_a x J_b when joint a
 * is revolute (it turns everything behind it, the column J_b included) and 0 when it is prismatic.  Degrees of
 * freedom are numbered along the chain (joint_dof is increasing).  No reference counterpart: CasADi differentiates
 * the forward kinematics symbolically for FORCES Pro's exact Hessian (mpcModel.py:139-160 generateSolver). */
int orc_fk_curv(const orc_desc *d, const double *q, int frame, const double *F, double *Cacc) {
  const int n = d->n;
  double pos[3], Jp[3 * 8], ax[8][3];
  int rc = fk_axes(d, q, frame, pos, Jp, d->robot == ORC_ROBOT_CHAIN ? ax : 0);
  if (rc != 0) return rc;
  if (d->robot == ORC_ROBOT_DIFFDRIVE) {
    /* p = (x, y) + R(theta) o: only d2 p / dtheta2 = -(p - (x, y)) is non-zero (diff_drive_mpc_model.py: the frames ride
     * on the base) */
    Cacc[2 * n + 2] -= F[0] * (pos[0] - q[0]) + F[1] * (pos[1] - q[1]);
    return 0;
  }
  for (int a = 0; a < n; a++) {
    /* G = F x axis_a, then F . (axis_a x J_b) = G . J_b */
    const double G[3] = {F[1] * ax[a][2] - F[2] * ax[a][1], F[2] * ax[a][0] - F[0] * ax[a][2],
                         F[0] * ax[a][1] - F[1] * ax[a][0]};
    for (int b = a; b < n; b++) {
      const double h = G[0] * Jp[0 * n + b] + G[1] * Jp[1 * n + b] + G[2] * Jp[2 * n + b];
      Cacc[a * n + b] += h;
      if (b != a) Cacc[b * n + a] += h;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* dynamics: continuous model + ERK2 (explicit midpoint) x 5 nodes      */
/* mpcModel.py:65-69, diff_drive_mpc_model.py:24-41, mpcModel.py:118-120 */
/* ------------------------------------------------------------------ */
static void cont_dyn(const orc_desc *d, const double *x, const double *u,
                     double *xd, double *fx, double *fu) {
  const int nx = d->nx, nu = d->nu, n = d->n;
  if (fx) memset(fx, 0, sizeof(double) * nx * nx);
  if (fu) memset(fu, 0, sizeof(double) * nx * nu);
  if (d->robot == ORC_ROBOT_CHAIN) {
    for (int i = 0; i < n; i++) {
      xd[i] = x[n + i];
      xd[n + i] = u[i];
      if (fx) fx[i * nx + n + i] = 1.0;
      if (fu) fu[(n + i) * nu + i] = 1.0;
    }
  } else {
    double th = x[2], v = x[6], w = x[7];
    double c = cos(th), s = sin(th);
    xd[0] = c * v; xd[1] = s * v; xd[2] = w;
    xd[3] = 0; xd[4] = 0; xd[5] = 0;
    xd[6] = u[0]; xd[7] = u[1];
    if (fx) {
      fx[0 * nx + 2] = -s * v; fx[0 * nx + 6] = c;
      fx[1 * nx + 2] = c * v;  fx[1 * nx + 6] = s;
      fx[2 * nx + 7] = 1.0;
    }
    if (fu) { fu[6 * nu + 0] = 1.0; fu[7 * nu + 1] = 1.0; }
  }
}

#define ORC_ERK_NODES 5
/* A: nx x nx, Bu: nx x nu (w.r.t. the control only; slack never enters) */
static void disc_dyn(const orc_desc *d, const double *x0, const double *u,
                     double *xn, double *A, double *Bu) {
  const int nx = d->nx, nu = d->nu;
  const double h = d->dt / ORC_ERK_NODES;
  double x[NXM];
  memcpy(x, x0, sizeof(double) * nx);
  if (A) {
    memset(A, 0, sizeof(double) * nx * nx);
    for (int i = 0; i < nx; i++) A[i * nx + i] = 1.0;
    memset(Bu, 0, sizeof(double) * nx * nu);
  }
  for (int it = 0; it < ORC_ERK_NODES; it++) {
    double k1[NXM], k2[NXM], xm[NXM];
    double fx1[NXM * NXM], fu1[NXM * NWM], fx2[NXM * NXM], fu2[NXM * NWM];
    cont_dyn(d, x, u, k1, A ? fx1 : 0, A ? fu1 : 0);
    for (int i = 0; i < nx; i++) xm[i] = x[i] + 0.5 * h * k1[i];
    cont_dyn(d, xm, u, k2, A ? fx2 : 0, A ? fu2 : 0);
    if (A) {
      /* As = I + h fx2 (I + h/2 fx1);  Bs = h (fx2 (h/2 fu1) + fu2) */
      double M[NXM * NXM], As[NXM * NXM], Bs[NXM * NWM], T[NXM * NXM], TB[NXM * NWM];
      for (int i = 0; i < nx; i++)
        for (int j = 0; j < nx; j++) M[i * nx + j] = (i == j ? 1.0 : 0.0) + 0.5 * h * fx1[i * nx + j];
      for (int i = 0; i < nx; i++)
        for (int j = 0; j < nx; j++) {
          double s = 0;
          for (int k = 0; k < nx; k++) s += fx2[i * nx + k] * M[k * nx + j];
          As[i * nx + j] = (i == j ? 1.0 : 0.0) + h * s;
        }
      for (int i = 0; i < nx; i++)
        for (int j = 0; j < nu; j++) {
          double s = 0;
          for (int k = 0; k < nx; k++) s += fx2[i * nx + k] * (0.5 * h * fu1[k * nu + j]);
          Bs[i * nu + j] = h * (s + fu2[i * nu + j]);
        }
      for (int i = 0; i < nx; i++)
        for (int j = 0; j < nx; j++) {
          double s = 0;
          for (int k = 0; k < nx; k++) s += As[i * nx + k] * A[k * nx + j];
          T[i * nx + j] = s;
        }
      for (int i = 0; i < nx; i++)
        for (int j = 0; j < nu; j++) {
          double s = 0;
          for (int k = 0; k < nx; k++) s += As[i * nx + k] * Bu[k * nu + j];
          TB[i * nu + j] = s + Bs[i * nu + j];
        }
      memcpy(A, T, sizeof(double) * nx * nx);
      memcpy(Bu, TB, sizeof(double) * nx * nu);
    }
    for (int i = 0; i < nx; i++) x[i] += h * k2[i];
  }
  memcpy(xn, x, sizeof(double) * nx);
}

int orc_dynamics(const orc_desc *d, const double *x, const double *u, double *xnext) {
  disc_dyn(d, x, u, xnext, 0, 0);
  return 0;
}

/* ------------------------------------------------------------------ */
/* rows                                                                 */
/* ------------------------------------------------------------------ */
static int module_rows(const orc_desc *d, int mi) {
  switch (d->module_kind[mi]) {
    case ORC_MOD_RADIAL: return d->nobst * d->n_links;          /* RadialConstraints.py:9 */
    case ORC_MOD_LINEAR: return d->nobst * d->n_links;          /* LinearConstraints.py:13 */
    case ORC_MOD_SELFCOLLISION: return d->n_pairs;              /* SelfCollision...py:13 */
    case ORC_MOD_JOINTLIMIT: return 2 * d->n;                   /* JointLimitConstraints.py:10 */
    case ORC_MOD_VELLIMIT: return 4;                            /* VelLimitConstraints.py:28-31 (4 rows; _n_ineq=2 is a bug) */
    case ORC_MOD_INPUTLIMIT: return 2 * d->nu;                  /* InputLimitConstraints.py:7 */
    case ORC_MOD_ROWS: {                                        /* a user class of InequalityManager.py:17-21, as rows */
      int c = 0;
      for (int r = 0; r < d->n_xrows; r++) c += d->xrow_mod[r] == mi;
      return c;
    }
  }
  return -1;
}

int orc_num_rows(const orc_desc *d, int *nh_out, int *m_out) {
  int nh = 0;
  for (int i = 0; i < d->n_modules; i++) {
    int r = module_rows(d, i);
    if (r < 0) return -1;
    nh += r;
  }
  int m = nh;
  const int nv = nvar_of(d);
  for (int j = 0; j < nv; j++) {
    if (isfinite(d->lb[j])) m++;
    if (isfinite(d->ub[j])) m++;
  }
  if (nh > ORC_NH_MAX || m > MRM) return -2;
  if (nh_out) *nh_out = nh;
  if (m_out) *m_out = m;
  return 0;
}

/* ------------------------------------------------------------------ */
/* stage evaluation                                                     */
/* ------------------------------------------------------------------ */
#define ORC_EVAL_BAD_AVOID (-100) /* an inverse-barrier row has h <= 0 */
/* second-order terms of the distance rows, filled by orc_eval_stage(want = 1) */
static __thread double tl_C[ORC_NH_MAX][64];   /* n x n curvature matrix per general row (zero if none) */
static __thread double tl_cw[ORC_NH_MAX];      /* extra weight from the inverse-barrier objective */
static __thread double tl_G[64];               /* second-order term of the goal cost in q (models with_fk_curv) */

/* the arms (n > 3, no slack): exact second-order terms with the latch-release rule of round 2 */
static int arm_curv(const orc_desc *d) { return d->robot == ORC_ROBOT_CHAIN && d->ns == 0 && d->n > 3; }
/* the diff-drive base (round 4): exact second-order terms of the unicycle -- nu . grad^2 Phi of the discrete dynamics
 * and the rotation of off-centre frames (ORC_DD_CURV=0 in the environment: Gauss-Newton blocks only, for A/B runs) */
static int dd_curv(const orc_desc *d) {
  static int on = -1;
  if (on < 0) { const char *e = getenv("ORC_DD_CURV"); on = (e && e[0] == '0') ? 0 : 1; }
  return d->robot == ORC_ROBOT_DIFFDRIVE && on;
}
/* Models whose failed curvature steps are BACKED OFF (the terms are switched off for 1, 2, 4 .. 16 iterations, see
 * solve_impl) instead of latched off: the unicycle, and (round 4) the small holonomic chains -- on the BASELINE point
 * robots the slowest instance of a batch of 4096 needs 66 - 68 passes instead of 82 - 95 at an unchanged mean (13.97 vs
 * 13.99; three seeds; warm closed loops: 7.34 vs 7.40 passes per control step).  The arms keep the latch with its
 * release rule (round 2): their cold solves see no failure at all. */
static int backoff_model(const orc_desc *d) {
  return dd_curv(d) || (d->robot == ORC_ROBOT_CHAIN && d->ns == 0 && d->n <= 3);
}
/* models whose curvature terms are scaled down before they are given up for an iteration (ORC_CS_*) */
static int cscale_model(const orc_desc *d) { return d->robot == ORC_ROBOT_CHAIN && d->ns == 0 && d->n <= 3; }
/* models whose distance rows and goal cost carry second derivatives of the kinematics */
static int with_fk_curv(const orc_desc *d) { return arm_curv(d) || dd_curv(d); }


/* fixed_state != 0 (stage 1, whose state is pinned to xinit): rows that depend on the state
 * only and are not softened are constants of the problem -- they are neutralised (value 1,
 * zero gradient, no inverse-barrier term), the standard presolve for constraints on fixed
 * variables.  Without it a start state that sits on a constraint boundary (every closed loop
 * whose previous plan had an active state constraint) makes the NLP infeasible by rounding. */
int orc_eval_stage(const orc_desc *d, const double *z, const double *p,
                   int want, double *f_out, double *gf, double *H, double *g,
                   double *Jg, double *xnext, double *A, double *Bm, int fixed_state) {
  const int n = d->n, nx = d->nx, ns = d->ns, nu = d->nu, nv = nx + ns + nu, nw = ns + nu;
  const double *q = z;
  const double *u = z + nx + ns;
  const double s = ns ? z[nx] : 0.0;
  double f = 0.0;
  int bad = 0;
  if (want) {
    memset(gf, 0, sizeof(double) * nv);
    memset(H, 0, sizeof(double) * nv * nv);
  }
  /* --- kinematics cache --- */
  double fpos[ORC_MAX_JOINTS][3], fJ[ORC_MAX_JOINTS][3 * 8];
  int fdone[ORC_MAX_JOINTS] = {0};
#define NEED_FK(fr)                                         \
  do {                                                      \
    if (!fdone[fr]) {                                       \
      if (orc_fk(d, q, fr, fpos[fr], fJ[fr]) != 0) return -1; \
      fdone[fr] = 1;                                        \
    }                                                       \
  } while (0)

  /* --- GoalReaching (goal_reaching.py:19-33) --- */
  if (d->has_goal) {
    const double *goal = p + d->off_goal, *w = p + d->off_wgoal;
    NEED_FK(d->end_frame);
    const double *pe = fpos[d->end_frame], *J = fJ[d->end_frame];
    double e[3];
    for (int i = 0; i < 3; i++) { e[i] = pe[i] - goal[i]; f += w[i] * e[i] * e[i]; }
    if (want) {
      for (int a = 0; a < n; a++) {
        double ga = 0;
        for (int i = 0; i < 3; i++) ga += 2.0 * w[i] * e[i] * J[i * n + a];
        gf[a] += ga;
        for (int b = 0; b < n; b++) {
          double hab = 0;
          for (int i = 0; i < 3; i++) hab += 2.0 * w[i] * J[i * n + a] * J[i * n + b];
          H[a * nv + b] += hab;
        }
      }
      /* what Gauss-Newton leaves out: sum_i 2 w_i e_i grad^2 p_i (added to the Hessian with the curvature terms) */
      memset(tl_G, 0, sizeof(double) * n * n);
      if (with_fk_curv(d)) {
        const double F[3] = {2.0 * w[0] * e[0], 2.0 * w[1] * e[1], 2.0 * w[2] * e[2]};
        orc_fk_curv(d, q, d->end_frame, F, tl_G);
      }
    }
  }
  /* --- control effort + slack penalty (ObjectiveManager.py:28-42) --- */
  {
    const double *wu = p + d->off_wu;
    for (int j = 0; j < nu; j++) {
      f += wu[j] * u[j] * u[j];
      if (want) {
        gf[nx + ns + j] += 2.0 * wu[j] * u[j];
        H[(nx + ns + j) * nv + nx + ns + j] += 2.0 * wu[j];
      }
    }
    if (ns) {
      double ws = p[d->off_ws];
      f += ws * s * s;
      if (want) { gf[nx] += 2.0 * ws * s; H[nx * nv + nx] += 2.0 * ws; }
    }
  }
  /* --- inequality modules, YAML order (InequalityManager.py:25-33) --- */
  int row = 0;
  if (want) {
    /* (only the part in use is cleared: the full scratch arrays are 32 KB per stage evaluation) */
    int nh_used = 0, m_used = 0;
    orc_num_rows(d, &nh_used, &m_used);
    memset(Jg, 0, sizeof(double) * (size_t)m_used * nv);
    for (int r = 0; r < nh_used; r++) memset(tl_C[r], 0, sizeof(double) * n * n);
    memset(tl_cw, 0, sizeof(double) * nh_used);
  }
  for (int mi = 0; mi < d->n_modules; mi++) {
    const int kind = d->module_kind[mi];
    const int row0 = row;
    int xrows_on_u = 0;   /* (ORC_MOD_ROWS) the module's rows are on inputs */
    if (kind == ORC_MOD_RADIAL) {
      /* mpcBase.py:82-101: links outer, obstacles inner */
      const double rb = p[d->off_r_body];
      for (int l = 0; l < d->n_links; l++) {
        int fr = d->link_frame[l];
        NEED_FK(fr);
        for (int i = 0; i < d->nobst; i++) {
          const double *ob = p + d->off_obst + 4 * i;
          double dv[3] = {fpos[fr][0] - ob[0], fpos[fr][1] - ob[1], fpos[fr][2] - ob[2]};
          double dist = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
          g[row] = dist - ob[3] - rb;
          if (want) {
            for (int a = 0; a < n; a++)
              Jg[row * nv + a] = (dv[0] * fJ[fr][0 * n + a] + dv[1] * fJ[fr][1 * n + a] + dv[2] * fJ[fr][2 * n + a]) / dist;
            for (int a = 0; a < n; a++)
              for (int bb = 0; bb < n; bb++) {
                double jj = fJ[fr][0 * n + a] * fJ[fr][0 * n + bb] + fJ[fr][1 * n + a] * fJ[fr][1 * n + bb] + fJ[fr][2 * n + a] * fJ[fr][2 * n + bb];
                tl_C[row][a * n + bb] = (jj - Jg[row * nv + a] * Jg[row * nv + bb]) / dist;
              }
            if (with_fk_curv(d)) {
              const double F[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist};
              orc_fk_curv(d, q, fr, F, tl_C[row]);
            }
          }
          row++;
        }
      }
    } else if (kind == ORC_MOD_LINEAR) {
      /* LinearConstraints.py:25-40, utils.py:48-52 */
      const double rb = p[d->off_r_body];
      for (int l = 0; l < d->n_links; l++) {
        int fr = d->link_frame[l];
        NEED_FK(fr);
        for (int i = 0; i < d->nobst; i++) {
          const double *pl = p + d->off_lin + 4 * i;
          double nn = sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
          double sd = pl[0] * fpos[fr][0] + pl[1] * fpos[fr][1] + pl[2] * fpos[fr][2] + pl[3];
          double sg = sd < 0 ? -1.0 : 1.0;
          g[row] = fabs(sd) / nn - rb;
          if (want)
            for (int a = 0; a < n; a++)
              Jg[row * nv + a] = sg * (pl[0] * fJ[fr][0 * n + a] + pl[1] * fJ[fr][1 * n + a] + pl[2] * fJ[fr][2 * n + a]) / nn;
          if (want && with_fk_curv(d)) {
            const double F[3] = {sg * pl[0] / nn, sg * pl[1] / nn, sg * pl[2] / nn};
            orc_fk_curv(d, q, fr, F, tl_C[row]);
          }
          row++;
        }
      }
    } else if (kind == ORC_MOD_SELFCOLLISION) {
      /* SelfCollisionAvoidanceConstraints.py:19-27 */
      const double rb = p[d->off_r_body];
      for (int pi = 0; pi < d->n_pairs; pi++) {
        int fa = d->pair_frame[pi][0], fb = d->pair_frame[pi][1];
        NEED_FK(fa);
        NEED_FK(fb);
        double dv[3] = {fpos[fa][0] - fpos[fb][0], fpos[fa][1] - fpos[fb][1], fpos[fa][2] - fpos[fb][2]};
        double dist = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        g[row] = dist - 2.0 * rb;
        if (want) {
          for (int a = 0; a < n; a++) {
            double s3 = 0;
            for (int i = 0; i < 3; i++) s3 += dv[i] * (fJ[fa][i * n + a] - fJ[fb][i * n + a]);
            Jg[row * nv + a] = s3 / dist;
          }
          for (int a = 0; a < n; a++)
            for (int bb = 0; bb < n; bb++) {
              double jj = 0;
              for (int i = 0; i < 3; i++) jj += (fJ[fa][i * n + a] - fJ[fb][i * n + a]) * (fJ[fa][i * n + bb] - fJ[fb][i * n + bb]);
              tl_C[row][a * n + bb] = (jj - Jg[row * nv + a] * Jg[row * nv + bb]) / dist;
            }
          if (with_fk_curv(d)) {
            const double Fa[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist}, Fb[3] = {-Fa[0], -Fa[1], -Fa[2]};
            orc_fk_curv(d, q, fa, Fa, tl_C[row]);
            orc_fk_curv(d, q, fb, Fb, tl_C[row]);
          }
        }
        row++;
      }
    } else if (kind == ORC_MOD_JOINTLIMIT) {
      /* JointLimitConstraints.py:21-31 */
      const double *lo = p + d->off_lower, *hi = p + d->off_upper;
      for (int j = 0; j < n; j++) {
        g[row] = q[j] - lo[j];
        if (want) Jg[row * nv + j] = 1.0;
        row++;
        g[row] = hi[j] - q[j];
        if (want) Jg[row * nv + j] = -1.0;
        row++;
      }
    } else if (kind == ORC_MOD_VELLIMIT) {
      /* VelLimitConstraints.py:20-31: vel = z[n:nx][-2:] */
      const double *lo = p + d->off_lower_vel, *hi = p + d->off_upper_vel;
      for (int j = 0; j < 2; j++) {
        int idx = nx - 2 + j;
        g[row] = z[idx] - lo[j];
        if (want) Jg[row * nv + idx] = 1.0;
        row++;
        g[row] = hi[j] - z[idx];
        if (want) Jg[row * nv + idx] = -1.0;
        row++;
      }
    } else if (kind == ORC_MOD_INPUTLIMIT) {
      /* InputLimitConstraints.py:19-29 */
      const double *lo = p + d->off_lower_u, *hi = p + d->off_upper_u;
      for (int j = 0; j < nu; j++) {
        int idx = nx + ns + j;
        g[row] = u[j] - lo[j];
        if (want) Jg[row * nv + idx] = 1.0;
        row++;
        g[row] = hi[j] - u[j];
        if (want) Jg[row * nv + idx] = -1.0;
        row++;
      }
    } else if (kind == ORC_MOD_ROWS) {
      /* a module given as row descriptions (orc_desc::xrow_*): the arithmetic of the row kinds above, one row each */
      for (int xr = 0; xr < d->n_xrows; xr++) {
        if (d->xrow_mod[xr] != mi) continue;
        const int a0 = d->xrow_a[xr], b0 = d->xrow_b[xr], po = d->xrow_poff[xr];
        if (d->xrow_kind[xr] == ORC_ROW_RADIAL) {
          const double rb = p[d->off_r_body];
          const int fr = a0;
          NEED_FK(fr);
          const double *ob = p + po;
          double dv[3] = {fpos[fr][0] - ob[0], fpos[fr][1] - ob[1], fpos[fr][2] - ob[2]};
          double dist = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
          g[row] = dist - ob[3] - rb;
          if (want) {
            for (int a = 0; a < n; a++)
              Jg[row * nv + a] = (dv[0] * fJ[fr][0 * n + a] + dv[1] * fJ[fr][1 * n + a] + dv[2] * fJ[fr][2 * n + a]) / dist;
            for (int a = 0; a < n; a++)
              for (int bb = 0; bb < n; bb++) {
                double jj = fJ[fr][0 * n + a] * fJ[fr][0 * n + bb] + fJ[fr][1 * n + a] * fJ[fr][1 * n + bb] + fJ[fr][2 * n + a] * fJ[fr][2 * n + bb];
                tl_C[row][a * n + bb] = (jj - Jg[row * nv + a] * Jg[row * nv + bb]) / dist;
              }
            if (with_fk_curv(d)) {
              const double F[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist};
              orc_fk_curv(d, q, fr, F, tl_C[row]);
            }
          }
        } else if (d->xrow_kind[xr] == ORC_ROW_LINEAR) {
          const double rb = p[d->off_r_body];
          const int fr = a0;
          NEED_FK(fr);
          const double *pl = p + po;
          double nn = sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
          double sd = pl[0] * fpos[fr][0] + pl[1] * fpos[fr][1] + pl[2] * fpos[fr][2] + pl[3];
          double sg = sd < 0 ? -1.0 : 1.0;
          g[row] = fabs(sd) / nn - rb;
          if (want)
            for (int a = 0; a < n; a++)
              Jg[row * nv + a] = sg * (pl[0] * fJ[fr][0 * n + a] + pl[1] * fJ[fr][1 * n + a] + pl[2] * fJ[fr][2 * n + a]) / nn;
          if (want && with_fk_curv(d)) {
            const double F[3] = {sg * pl[0] / nn, sg * pl[1] / nn, sg * pl[2] / nn};
            orc_fk_curv(d, q, fr, F, tl_C[row]);
          }
        } else if (d->xrow_kind[xr] == ORC_ROW_SELF) {
          const double rb = p[d->off_r_body];
          const int fa = a0, fb = b0;
          NEED_FK(fa);
          NEED_FK(fb);
          double dv[3] = {fpos[fa][0] - fpos[fb][0], fpos[fa][1] - fpos[fb][1], fpos[fa][2] - fpos[fb][2]};
          double dist = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
          g[row] = dist - 2.0 * rb;
          if (want) {
            for (int a = 0; a < n; a++) {
              double s3 = 0;
              for (int i = 0; i < 3; i++) s3 += dv[i] * (fJ[fa][i * n + a] - fJ[fb][i * n + a]);
              Jg[row * nv + a] = s3 / dist;
            }
            for (int a = 0; a < n; a++)
              for (int bb = 0; bb < n; bb++) {
                double jj = 0;
                for (int i = 0; i < 3; i++) jj += (fJ[fa][i * n + a] - fJ[fb][i * n + a]) * (fJ[fa][i * n + bb] - fJ[fb][i * n + bb]);
                tl_C[row][a * n + bb] = (jj - Jg[row * nv + a] * Jg[row * nv + bb]) / dist;
              }
            if (with_fk_curv(d)) {
              const double Fa[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist}, Fb[3] = {-Fa[0], -Fa[1], -Fa[2]};
              orc_fk_curv(d, q, fa, Fa, tl_C[row]);
              orc_fk_curv(d, q, fb, Fb, tl_C[row]);
            }
          }
        } else if (d->xrow_kind[xr] == ORC_ROW_VAR) {
          if (a0 < 0 || a0 >= nv) return -3;
          g[row] = b0 > 0 ? z[a0] - p[po] : p[po] - z[a0];
          if (want) Jg[row * nv + a0] = b0 > 0 ? 1.0 : -1.0;
          if (a0 >= nx + ns) xrows_on_u = 1;
        } else {
          return -3;
        }
        row++;
      }
    } else {
      return -3;
    }
    const int x_only = (kind == ORC_MOD_ROWS) ? !xrows_on_u : (kind != ORC_MOD_INPUTLIMIT);
    const int neutral = fixed_state && x_only && !ns;
    if (neutral) {
      for (int r = row0; r < row; r++) {
        g[r] = 1.0;
        if (want) {
          for (int a = 0; a < nv; a++) Jg[r * nv + a] = 0.0;
          memset(tl_C[r], 0, sizeof tl_C[r]);
        }
      }
    }
    /* --- ConstraintAvoidance (constraint_avoidance.py:22-31): N * w_i / h_first --- */
    if (d->has_avoid && row > row0 && !(fixed_state && x_only)) {
      double wi = p[d->off_wconstr + mi];
      if (wi != 0.0) {
        double c = (double)d->N * wi;
        double h = g[row0];
        if (!(h > 0.0)) bad = 1;
        f += c / h;
        if (want) {
          const double *jr = Jg + row0 * nv;
          double c1 = -c / (h * h), c2 = 2.0 * c / (h * h * h);
          tl_cw[row0] = c / (h * h);
          for (int a = 0; a < nv; a++) {
            gf[a] += c1 * jr[a];
            if (jr[a] != 0.0)
              for (int b = 0; b < nv; b++) H[a * nv + b] += c2 * jr[a] * jr[b];
          }
        }
      }
    }
  }
  /* slack softening of every general row (intended semantics of
   * InequalityManager.py:29-32) */
  if (ns) {
    for (int r = 0; r < row; r++) {
      g[r] += s;
      if (want) Jg[r * nv + nx] = 1.0;
    }
  }
  /* simple bounds lb <= z <= ub (mpcModel.py:91-104) */
  for (int j = 0; j < nv; j++)
    if (isfinite(d->lb[j])) {
      const int neutral = fixed_state && j < nx;
      g[row] = neutral ? 1.0 : z[j] - d->lb[j];
      if (want && !neutral) Jg[row * nv + j] = 1.0;
      row++;
    }
  for (int j = 0; j < nv; j++)
    if (isfinite(d->ub[j])) {
      const int neutral = fixed_state && j < nx;
      g[row] = neutral ? 1.0 : d->ub[j] - z[j];
      if (want && !neutral) Jg[row * nv + j] = -1.0;
      row++;
    }
  /* dynamics */
  if (xnext) {
    double Bu[NXM * NWM];
    disc_dyn(d, z, u, xnext, want ? A : 0, want ? Bu : 0);
    if (want) {
      for (int i = 0; i < nx; i++) {
        for (int j = 0; j < nw; j++) Bm[i * nw + j] = 0.0;
        for (int j = 0; j < nu; j++) Bm[i * nw + ns + j] = Bu[i * nu + j];
      }
    }
  }
  *f_out = f;
  return bad ? ORC_EVAL_BAD_AVOID : row;
#undef NEED_FK
}

/* ------------------------------------------------------------------ */
/* primal-dual interior point with stage-wise Riccati factorisation     */
/* ------------------------------------------------------------------ */
#define ORC_TMIN 1e-2      /* slack push at initialisation */
/* fraction to the boundary: a step may take a slack or a multiplier to (1 - tau) of its value at most.  0.995 until the end
 * of round 4; measured then over the BASELINE scenes: 0.985 for the small holonomic chains (passes 13.95 ->
 * 13.27 on the point robot, same tails), 0.98 for the unicycle and the arms (22.1 -> 18.0 and 13.06 -> 12.12 passes): a
 * slack that is not driven to 1/200 of its value in one step leaves the next steps better conditioned */
#define ORC_TAU_CHAIN3 0.985
#define ORC_TAU_OTHER 0.98
/* the complementarity test of the convergence check is made against this share of tol_comp: with the fractions above
 * the last step no longer overshoots the tolerance by an order of magnitude, and a solve that stops at 0.9 tol_comp
 * is up to 3.5e-4 away from the exact solution in the applied control (scipy golden vectors, bar 1e-4); at 0.3 the
 * largest difference is 8.4e-5 (tests/test_scipy_golden.py) */
#define ORC_COMP_FRAC 0.3
#define ORC_LS_MAX 25      /* max halvings in the line search */
#define ORC_LS_GROW 1      /* step-length memory: a line search starts this many halvings above the last accepted one */
#define ORC_ARMIJO 1e-4
#define ORC_MU_DIVERGED 1e12
#define ORC_CURV_MU 1e-2    /* curvature terms only once the barrier parameter is this small */
#define ORC_LS_CURV 2       /* trials granted to a step computed with constraint curvature */
#define ORC_CURV_FAIL_MAX 2 /* consecutive curvature-step failures before latching Gauss-Newton */
#define ORC_ACC_FEAS 1e-6   /* ... feasibility / complementarity level */
/* barrier restart (round 4): from iteration ORC_RS_IT on, ORC_RS_N accepted steps in a row shorter than ORC_RS_ALPHA
 * while mu < ORC_RS_MU mean the iterate crawls along a boundary with the barrier at its floor: mu is held at ORC_RS_MU
 * and released by the factor ORC_RS_DECAY per iteration */
#define ORC_RS_IT 8
#define ORC_RS_N 3
#define ORC_RS_ALPHA 0.2
#define ORC_RS_MU 1e-3
#define ORC_RS_DECAY 0.3
/* scaled curvature (round 4, the small holonomic chains): a failed factorisation with the exact constraint curvature is
 * retried with the curvature terms scaled by theta = 1/2, 1/4 (Q = Gauss-Newton blocks + theta * curvature: the blocks
 * are positive definite at theta = 0) before the iteration falls back to Gauss-Newton; the scale that worked is kept
 * for the next iteration and doubled again after ORC_CS_CLEAN accepted curvature steps in a row that needed no retry
 *  */
#define ORC_CS_MIN 0.3
#define ORC_CS_CLEAN 3
#define ORC_TRACE_W 8

typedef struct {
  int N, nx, nw, nv, m;
  double *z, *t, *lam, *nu;
  double *f, *gf, *H, *g, *Jg, *xn, *A, *Bm;
  double *Q, *qv, *rc;
  double *K, *kff, *P, *pv;
  double *dz, *dtt, *dlam, *nunew;
  double *zt, *tt, *gt, *xnt;
  double *Cc, *cw, *Gc;
} orc_work;

/* The workspace of a solve is kept per thread and reused while the dimensions stay the same: a fresh set of
 * callocs per solve costs 1.3 MB of page faults and, with many OpenMP threads, serialises on the process's
 * address-space lock (the thread-scaling row of bench.py's cpu_baseline).  Every array is zeroed on reuse as calloc
 * did, except Cc (1 MB), whose entries eval_all writes before the step computation reads them. */
typedef struct { double **p; size_t n; int zero; } orc_slot;
static __thread orc_work tl_work;
static __thread int tl_work_sig[5] = {0, 0, 0, 0, 0};
static __thread int tl_work_live = 0;

static int work_slots(orc_work *w, const orc_desc *d, int m, orc_slot *sl) {
  const size_t N = d->N, nx = d->nx, nv = nvar_of(d), nw = d->ns + d->nu, M = m;
  int n = 0;
#define SLOT(f, cnt, z) do { sl[n].p = &w->f; sl[n].n = (cnt); sl[n].zero = (z); n++; } while (0)
  SLOT(z, N * nv, 1); SLOT(t, N * M, 1); SLOT(lam, N * M, 1); SLOT(nu, N * nx, 1);
  SLOT(f, N, 1); SLOT(gf, N * nv, 1); SLOT(H, N * nv * nv, 1);
  SLOT(g, N * MRM, 1); SLOT(Jg, N * MRM * nv, 1);
  SLOT(xn, N * nx, 1); SLOT(A, N * nx * nx, 1); SLOT(Bm, N * nx * nw, 1);
  SLOT(Q, N * nv * nv, 1); SLOT(qv, N * nv, 1); SLOT(rc, N * nx, 1);
  SLOT(K, N * nw * nx, 1); SLOT(kff, N * nw, 1); SLOT(P, N * nx * nx, 1); SLOT(pv, N * nx, 1);
  SLOT(dz, N * nv, 1); SLOT(dtt, N * M, 1); SLOT(dlam, N * M, 1); SLOT(nunew, N * nx, 1);
  SLOT(zt, N * nv, 1); SLOT(tt, N * M, 1); SLOT(gt, N * MRM, 1); SLOT(xnt, N * nx, 1);
  SLOT(Cc, N * ORC_NH_MAX * 64, 0); SLOT(cw, N * ORC_NH_MAX, 1); SLOT(Gc, N * 64, 1);
#undef SLOT
  return n;
}

static void work_free(orc_work *w) {
  free(w->z); free(w->t); free(w->lam); free(w->nu); free(w->f); free(w->gf); free(w->H);
  free(w->g); free(w->Jg); free(w->xn); free(w->A); free(w->Bm); free(w->Q); free(w->qv);
  free(w->rc); free(w->K); free(w->kff); free(w->P); free(w->pv); free(w->dz); free(w->dtt);
  free(w->dlam); free(w->nunew); free(w->zt); free(w->tt); free(w->gt); free(w->xnt); free(w->Cc); free(w->cw); free(w->Gc);
}

/* the calling thread's workspace for these dimensions, zeroed like a fresh calloc (owned by the thread: not freed) */
static orc_work *work_get(const orc_desc *d, int m) {
  orc_work *w = &tl_work;
  orc_slot sl[32];
  const int sig[5] = {d->N, d->nx, d->ns + d->nu, nvar_of(d), m};
  const int n = work_slots(w, d, m, sl);
  if (tl_work_live && memcmp(sig, tl_work_sig, sizeof sig) == 0) {
    for (int i = 0; i < n; i++)
      if (sl[i].zero) memset(*sl[i].p, 0, sizeof(double) * sl[i].n);
  } else {
    if (tl_work_live) work_free(w);
    for (int i = 0; i < n; i++) *sl[i].p = (double *)calloc(sl[i].n ? sl[i].n : 1, sizeof(double));
    memcpy(tl_work_sig, sig, sizeof sig);
    tl_work_live = 1;
  }
  w->N = d->N; w->nx = d->nx; w->nw = d->ns + d->nu; w->nv = nvar_of(d); w->m = m;
  return w;
}

/* evaluate every stage with derivatives at w->z; returns 0, or <0 */
static int eval_all(const orc_desc *d, orc_work *w, const double *params) {
  const int N = w->N, nv = w->nv, nx = w->nx, nw = w->nw;
  int rc = 0;
  for (int k = 0; k < N; k++) {
    int r = orc_eval_stage(d, w->z + (size_t)k * nv, params + (size_t)k * d->npar, 1, &w->f[k],
                           w->gf + (size_t)k * nv, w->H + (size_t)k * nv * nv, w->g + (size_t)k * MRM,
                           w->Jg + (size_t)k * MRM * nv, k < N - 1 ? w->xn + (size_t)k * nx : 0,
                           w->A + (size_t)k * nx * nx, w->Bm + (size_t)k * nx * nw, k == 0);
    {
      int nh_used = 0;
      orc_num_rows(d, &nh_used, 0);
      for (int r = 0; r < nh_used; r++)
        memcpy(w->Cc + ((size_t)k * ORC_NH_MAX + r) * 64, tl_C[r], sizeof(double) * d->n * d->n);
      memcpy(w->cw + (size_t)k * ORC_NH_MAX, tl_cw, sizeof(double) * nh_used);
      if (d->has_goal) memcpy(w->Gc + (size_t)k * 64, tl_G, sizeof(double) * d->n * d->n);
    }
    if (r == ORC_EVAL_BAD_AVOID) rc = ORC_EVAL_BAD_AVOID;
    else if (r < 0) return r;
  }
  return rc;
}

/* Test access to the curvature terms of one stage: Cout [n*n] = sum_i (lam_i + cN/h_i^2) grad^2 h_i - (second-order
 * term of the goal cost), i.e. what the step computation subtracts from the Gauss-Newton block when it uses them;
 * lam [nh] multipliers of the general rows.  The exact Hessian of f - lam^T g in q is H_qq - Cout. */
int orc_stage_curvature(const orc_desc *d, const double *z, const double *p, const double *lam, int fixed_state,
                        double *Cout) {
  static __thread double gf[NVM], H[NVM * NVM], g[MRM], Jg[MRM * NVM];
  double f;
  int nh = 0;
  if (orc_num_rows(d, &nh, 0) != 0) return -1;
  int r = orc_eval_stage(d, z, p, 1, &f, gf, H, g, Jg, 0, 0, 0, fixed_state);
  if (r < 0 && r != ORC_EVAL_BAD_AVOID) return r;
  const int n = d->n;
  for (int a = 0; a < n * n; a++) Cout[a] = d->has_goal ? -tl_G[a] : 0.0;
  for (int i = 0; i < nh; i++)
    for (int a = 0; a < n * n; a++) Cout[a] += (lam[i] + tl_cw[i]) * tl_C[i][a];
  return 0;
}

/* Cholesky of the nw x nw block, in place (lower). returns 0 ok */
static int chol(double *M, int n) {
  for (int j = 0; j < n; j++) {
    double dg = M[j * n + j];
    for (int k = 0; k < j; k++) dg -= M[j * n + k] * M[j * n + k];
    if (!(dg > 0.0)) return -1;
    dg = sqrt(dg);
    M[j * n + j] = dg;
    for (int i = j + 1; i < n; i++) {
      double s = M[i * n + j];
      for (int k = 0; k < j; k++) s -= M[i * n + k] * M[j * n + k];
      M[i * n + j] = s / dg;
    }
  }
  return 0;
}
static void chol_solve(const double *L, int n, double *b) {
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

/* Riccati backward + forward on the condensed stage QPs. */
static int riccati(orc_work *w) {
  const int N = w->N, nx = w->nx, nw = w->nw, nv = w->nv;
  double Pn[NXM * NXM], pn[NXM];
  memset(Pn, 0, sizeof Pn);
  memset(pn, 0, sizeof pn);
  for (int k = N - 1; k >= 0; k--) {
    const double *Q = w->Q + (size_t)k * nv * nv, *qv = w->qv + (size_t)k * nv;
    double Qxx[NXM * NXM], Qxw[NXM * NWM], Qww[NWM * NWM], qx[NXM], qw[NWM];
    for (int i = 0; i < nx; i++) {
      for (int j = 0; j < nx; j++) Qxx[i * nx + j] = Q[i * nv + j];
      for (int j = 0; j < nw; j++) Qxw[i * nw + j] = Q[i * nv + nx + j];
      qx[i] = qv[i];
    }
    for (int i = 0; i < nw; i++) {
      for (int j = 0; j < nw; j++) Qww[i * nw + j] = Q[(nx + i) * nv + nx + j];
      qw[i] = qv[nx + i];
    }
    if (k < N - 1) {
      const double *A = w->A + (size_t)k * nx * nx, *Bm = w->Bm + (size_t)k * nx * nw;
      const double *rc = w->rc + (size_t)k * nx;
      double PA[NXM * NXM], PB[NXM * NWM], Pc[NXM];
      for (int i = 0; i < nx; i++) {
        for (int j = 0; j < nx; j++) {
          double s = 0;
          for (int l = 0; l < nx; l++) s += Pn[i * nx + l] * A[l * nx + j];
          PA[i * nx + j] = s;
        }
        for (int j = 0; j < nw; j++) {
          double s = 0;
          for (int l = 0; l < nx; l++) s += Pn[i * nx + l] * Bm[l * nw + j];
          PB[i * nw + j] = s;
        }
        double s = pn[i];
        for (int l = 0; l < nx; l++) s += Pn[i * nx + l] * rc[l];
        Pc[i] = s;
      }
      for (int i = 0; i < nx; i++) {
        for (int j = 0; j < nx; j++) {
          double s = 0;
          for (int l = 0; l < nx; l++) s += A[l * nx + i] * PA[l * nx + j];
          Qxx[i * nx + j] += s;
        }
        for (int j = 0; j < nw; j++) {
          double s = 0;
          for (int l = 0; l < nx; l++) s += A[l * nx + i] * PB[l * nw + j];
          Qxw[i * nw + j] += s;
        }
        double s = 0;
        for (int l = 0; l < nx; l++) s += A[l * nx + i] * Pc[l];
        qx[i] += s;
      }
      for (int i = 0; i < nw; i++) {
        for (int j = 0; j < nw; j++) {
          double s = 0;
          for (int l = 0; l < nx; l++) s += Bm[l * nw + i] * PB[l * nw + j];
          Qww[i * nw + j] += s;
        }
        double s = 0;
        for (int l = 0; l < nx; l++) s += Bm[l * nw + i] * Pc[l];
        qw[i] += s;
      }
    }
    if (chol(Qww, nw) != 0) return -1;
    double *K = w->K + (size_t)k * nw * nx, *kff = w->kff + (size_t)k * nw;
    for (int j = 0; j < nx; j++) {
      double col[NWM];
      for (int i = 0; i < nw; i++) col[i] = -Qxw[j * nw + i];
      chol_solve(Qww, nw, col);
      for (int i = 0; i < nw; i++) K[i * nx + j] = col[i];
    }
    for (int i = 0; i < nw; i++) kff[i] = -qw[i];
    chol_solve(Qww, nw, kff);
    double *P = w->P + (size_t)k * nx * nx, *pv = w->pv + (size_t)k * nx;
    for (int i = 0; i < nx; i++) {
      for (int j = 0; j < nx; j++) {
        double s = Qxx[i * nx + j];
        for (int l = 0; l < nw; l++) s += Qxw[i * nw + l] * K[l * nx + j];
        P[i * nx + j] = s;
      }
      double s = qx[i];
      for (int l = 0; l < nw; l++) s += Qxw[i * nw + l] * kff[l];
      pv[i] = s;
    }
    for (int i = 0; i < nx; i++)
      for (int j = i + 1; j < nx; j++) {
        double a = 0.5 * (P[i * nx + j] + P[j * nx + i]);
        P[i * nx + j] = a;
        P[j * nx + i] = a;
      }
    memcpy(Pn, P, sizeof(double) * nx * nx);
    memcpy(pn, pv, sizeof(double) * nx);
  }
  /* forward */
  double dx[NXM];
  memset(dx, 0, sizeof dx);
  for (int k = 0; k < N; k++) {
    const double *K = w->K + (size_t)k * nw * nx, *kff = w->kff + (size_t)k * nw;
    double *dz = w->dz + (size_t)k * nv;
    for (int i = 0; i < nx; i++) dz[i] = dx[i];
    for (int i = 0; i < nw; i++) {
      double s = kff[i];
      for (int j = 0; j < nx; j++) s += K[i * nx + j] * dx[j];
      dz[nx + i] = s;
    }
    const double *P = w->P + (size_t)k * nx * nx, *pv = w->pv + (size_t)k * nx;
    for (int i = 0; i < nx; i++) {
      double s = pv[i];
      for (int j = 0; j < nx; j++) s += P[i * nx + j] * dx[j];
      w->nunew[(size_t)k * nx + i] = s;
    }
    if (k < N - 1) {
      const double *A = w->A + (size_t)k * nx * nx, *Bm = w->Bm + (size_t)k * nx * nw;
      const double *rc = w->rc + (size_t)k * nx;
      double dxn[NXM];
      for (int i = 0; i < nx; i++) {
        double s = rc[i];
        for (int j = 0; j < nx; j++) s += A[i * nx + j] * dx[j];
        for (int j = 0; j < nw; j++) s += Bm[i * nw + j] * dz[nx + j];
        dxn[i] = s;
      }
      memcpy(dx, dxn, sizeof(double) * nx);
    }
  }
  return 0;
}

/* merit pieces at a trial point (function values only):
 * returns 0 ok, 1 rejected (bad avoidance row / non-finite) */
static int trial_merit(const orc_desc *d, orc_work *w, const double *params, double mu,
                       double *f_out, double *theta_out, double *logsum_out) {
  const int N = w->N, nv = w->nv, nx = w->nx, m = w->m;
  double f = 0, th = 0, ls = 0;
  for (int k = 0; k < N; k++) {
    double fk;
    int r = orc_eval_stage(d, w->zt + (size_t)k * nv, params + (size_t)k * d->npar, 0, &fk, 0, 0,
                           w->gt + (size_t)k * MRM, 0, k < N - 1 ? w->xnt + (size_t)k * nx : 0, 0, 0, k == 0);
    if (r < 0) return 1;
    f += fk;
    for (int i = 0; i < m; i++) {
      th += fabs(w->gt[(size_t)k * MRM + i] - w->tt[(size_t)k * m + i]);
      ls += log(w->tt[(size_t)k * m + i]);
    }
  }
  for (int k = 0; k < N - 1; k++)
    for (int i = 0; i < nx; i++) th += fabs(w->xnt[(size_t)k * nx + i] - w->zt[(size_t)(k + 1) * nv + i]);
  (void)mu;
  if (!isfinite(f) || !isfinite(th) || !isfinite(ls)) return 1;
  *f_out = f; *theta_out = th; *logsum_out = ls;
  return 0;
}


/* Second-order terms of the distance rows are used when they are the exact
 * constraint Hessians: holonomic chain, no slack, n <= 3, and every frame a row
 * refers to moves affinely with q (prismatic joints, or a revolute joint at the
 * frame itself) -- the point robot.  DESIGN.md, section "Algorithm". */
static int frame_is_affine(const orc_desc *d, int f) {
  for (int j = 0; j <= f; j++)
    if (d->joint_type[j] == ORC_JOINT_REVOLUTE && j != f) return 0;
  return 1;
}
/* nu . grad^2 Phi of the unicycle's discrete dynamics (ERK2, explicit midpoint, ORC_ERK_NODES nodes of h = dt / nodes:
 * disc_dyn above).  In closed form the map is x+ = x + h sum_n cos(alpha_n) beta_n, y+ = y + h sum_n sin(alpha_n) beta_n
 * with alpha_n = theta + a_n omega + b_n u1 (midpoint heading of node n), beta_n = v + a_n u0 (midpoint speed),
 * a_n = (n + 1/2) h, b_n = h^2 n (n + 1) / 2; theta+, v+, omega+ are linear.  So only the costates of x and y carry
 * curvature:  D = h sum_n [ (-nx cos - ny sin) beta_n ga ga^T + (-nx sin + ny cos) (ga gb^T + gb ga^T) ],
 * ga = d alpha_n / d(theta, omega, u1) = (1, a_n, b_n), gb = d beta_n / d(v, u0) = (1, a_n).  Added to Q (nv x nv). */
static void dd_dyn_curv(const orc_desc *d, const double *z, const double *nun, double *Q) {
  const int nx = d->nx, nv = nvar_of(d), iu = nx + d->ns;
  const double h = d->dt / ORC_ERK_NODES;
  const double th = z[2], v = z[6], om = z[7], u0 = z[iu], u1 = z[iu + 1];
  const int ia[3] = {2, 7, iu + 1}, ib[2] = {6, iu};
  for (int n = 0; n < ORC_ERK_NODES; n++) {
    const double an = (n + 0.5) * h, bn = h * h * n * (n + 1) * 0.5;
    const double al = th + an * om + bn * u1, be = v + an * u0;
    const double c = cos(al), s = sin(al);
    const double P = h * (-nun[0] * c - nun[1] * s) * be, S = h * (-nun[0] * s + nun[1] * c);
    const double ga[3] = {1.0, an, bn}, gb[2] = {1.0, an};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Q[ia[i] * nv + ia[j]] += P * ga[i] * ga[j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 2; j++) {
        Q[ia[i] * nv + ib[j]] += S * ga[i] * gb[j];
        Q[ib[j] * nv + ia[i]] += S * ga[i] * gb[j];
      }
  }
}

static int model_uses_curvature(const orc_desc *d) {
  if (dd_curv(d)) return 1;
  if (d->robot != ORC_ROBOT_CHAIN || d->ns != 0) return 0;
  if (with_fk_curv(d)) return 1;   /* the kinematics' own second derivatives are part of the terms */
  for (int mi = 0; mi < d->n_modules; mi++) {
    if (d->module_kind[mi] == ORC_MOD_RADIAL)
      for (int l = 0; l < d->n_links; l++) if (!frame_is_affine(d, d->link_frame[l])) return 0;
    if (d->module_kind[mi] == ORC_MOD_SELFCOLLISION)
      for (int p = 0; p < d->n_pairs; p++)
        if (!frame_is_affine(d, d->pair_frame[p][0]) || !frame_is_affine(d, d->pair_frame[p][1])) return 0;
  }
  for (int r = 0; r < d->n_xrows; r++) {
    if (d->xrow_kind[r] == ORC_ROW_RADIAL && !frame_is_affine(d, d->xrow_a[r])) return 0;
    if (d->xrow_kind[r] == ORC_ROW_SELF && (!frame_is_affine(d, d->xrow_a[r]) || !frame_is_affine(d, d->xrow_b[r]))) return 0;
  }
  return 1;
}

/* Warm start of the multipliers (closed loops; no reference counterpart -- the reference warm-starts the plan
 * only, mpcPlanner.py:215-236): lam_w [N][m], nu_w [N][nx], mu_w = multipliers and final barrier parameter of the
 * previous solve of the same instance.  Stage k takes the values of stage k+1 (the last stage is repeated), like
 * the shifted plan.  Start: mu = clamp(ORC_WARM_KAPPA * mu_w, ORC_WARM_MU_MIN, mu0), t = max(g(z0), ORC_WARM_TMIN),
 * lam = max(shifted lam, mu / t), nu = shifted nu. */
#define ORC_WARM_KAPPA 1000.0
#define ORC_WARM_MU_MIN 1e-6
#define ORC_WARM_TMIN 1e-4
static int solve_impl(const orc_desc *d, const double *xinit, const double *x0, const double *params,
                      double *zout, orc_stats *st, double *trace, const double *lam_w, const double *nu_w,
                      double mu_w, double *lam_out, double *nu_out);

/* development aid: ORC_TRACE=1 prints the residuals of every iteration to stderr */
/* development aid: sweeps the GPU's pass kernels would need for the calling thread's last solve (one per trial point,
 * one more per step recomputed with the Gauss-Newton blocks) */
static __thread int tl_passes;
int orc_last_passes(void) { return tl_passes; }
static int orc_trace_stderr(void) {
  return getenv("ORC_TRACE") != 0;
}

int orc_solve(const orc_desc *d, const double *xinit, const double *x0, const double *params,
              double *zout, orc_stats *st, double *trace) {
  return solve_impl(d, xinit, x0, params, zout, st, trace, 0, 0, 0.0, 0, 0);
}

int orc_solve_warm(const orc_desc *d, const double *xinit, const double *x0, const double *params, double *zout,
                   orc_stats *st, const double *lam_w, const double *nu_w, double mu_w, double *lam_out,
                   double *nu_out) {
  return solve_impl(d, xinit, x0, params, zout, st, 0, lam_w, nu_w, mu_w, lam_out, nu_out);
}

static int solve_impl(const orc_desc *d, const double *xinit, const double *x0, const double *params,
                      double *zout, orc_stats *st, double *trace, const double *lam_w, const double *nu_w,
                      double mu_w, double *lam_out, double *nu_out) {
  int nh, m;
  if (orc_num_rows(d, &nh, &m) != 0) return -1;
  const int N = d->N, nx = d->nx, nv = nvar_of(d), nw = d->ns + d->nu;
  if (nx > NXM || nw > NWM || nv > NVM || N < 1) return -1;
  orc_work *w = work_get(d, m);
  memcpy(w->z, x0, sizeof(double) * N * nv);
  memcpy(w->z, xinit, sizeof(double) * nx); /* x_1 = xinit (mpcModel.py:108 xinitidx) */
  memset(st, 0, sizeof *st);
  double mu = d->mu0, rho = 0.0;
  int small_steps = 0;   /* barrier restart: accepted short steps in a row, the level mu is held at */
  double mu_hold = 0.0;
  double theta_mem = 1.0;   /* scaled curvature: the scale the next curvature step starts from, clean successes in a row */
  int theta_clean = 0;
  const int cscale = cscale_model(d);
  const double tau = cscale ? ORC_TAU_CHAIN3 : ORC_TAU_OTHER;
  int exitflag = 0, it = 0;
  const int curv_ok = model_uses_curvature(d);
  int gn_sticky = 0, curv_fail = 0, stall = 0;
  /* (the unicycle) after a curvature step that failed -- exact Hessian not positive definite, or its two-trial line
   * search -- the next curvature steps are skipped: 1, 2, 4 ... 16 iterations, the count doubling with every failure in
   * a row and starting over after a success.  In a non-convex region every iteration would otherwise pay a failed
   * factorisation or two failed trials plus a null pass before its Gauss-Newton step (one instance of the BASELINE
   * batch: 99 iterations in 315 passes); a permanent latch (the arms' rule) would give up the quadratic end game. */
  int curv_skip = 0, curv_back = 0;
  int ls_start = 0; /* step-length memory: halvings the next Gauss-Newton line search starts from */
  double obj_prev = 0.0;
  tl_passes = 1;
  int ev = eval_all(d, w, params);
  if (ev != 0) { exitflag = (ev == ORC_EVAL_BAD_AVOID) ? -7 : -10; goto done; }
  if (lam_w) {
    mu = ORC_WARM_KAPPA * mu_w;
    if (mu < ORC_WARM_MU_MIN) mu = ORC_WARM_MU_MIN;
    if (mu > d->mu0) mu = d->mu0;
  }
  for (int k = 0; k < N; k++) {
    const int ks = k < N - 1 ? k + 1 : k; /* shifted like the plan */
    for (int i = 0; i < m; i++) {
      double gv = w->g[(size_t)k * MRM + i];
      if (lam_w) {
        double tv = gv > ORC_WARM_TMIN ? gv : ORC_WARM_TMIN;
        double lc = mu / tv, lw = lam_w[(size_t)ks * m + i];
        w->t[(size_t)k * m + i] = tv;
        w->lam[(size_t)k * m + i] = lw > lc ? lw : lc;
      } else {
        double tv = gv > ORC_TMIN ? gv : ORC_TMIN;
        w->t[(size_t)k * m + i] = tv;
        w->lam[(size_t)k * m + i] = mu / tv;
      }
    }
    if (nu_w)
      for (int i = 0; i < nx; i++) w->nu[(size_t)k * nx + i] = (k == 0) ? 0.0 : nu_w[(size_t)ks * nx + i];
  }
  for (it = 0;; it++) {
    /* ---- residuals at the current iterate ---- */
    double res_stat = 0, res_eq = 0, res_ineq = 0, res_comp = 0, obj = 0, theta = 0, logsum = 0;
    for (int k = 0; k < N; k++) {
      const double *Jg = w->Jg + (size_t)k * MRM * nv;
      obj += w->f[k];
      if (k < N - 1)
        for (int i = 0; i < nx; i++) {
          double r = w->xn[(size_t)k * nx + i] - w->z[(size_t)(k + 1) * nv + i];
          w->rc[(size_t)k * nx + i] = r;
          if (fabs(r) > res_eq) res_eq = fabs(r);
          theta += fabs(r);
        }
      for (int i = 0; i < m; i++) {
        double rg = w->g[(size_t)k * MRM + i] - w->t[(size_t)k * m + i];
        if (fabs(rg) > res_ineq) res_ineq = fabs(rg);
        theta += fabs(rg);
        double c = w->t[(size_t)k * m + i] * w->lam[(size_t)k * m + i];
        if (c > res_comp) res_comp = c;
        logsum += log(w->t[(size_t)k * m + i]);
      }
      for (int a = 0; a < nv; a++) {
        double r = w->gf[(size_t)k * nv + a];
        for (int i = 0; i < m; i++) r -= Jg[i * nv + a] * w->lam[(size_t)k * m + i];
        if (k < N - 1) {
          const double *nun = w->nu + (size_t)(k + 1) * nx;
          if (a < nx) { for (int l = 0; l < nx; l++) r += w->A[(size_t)k * nx * nx + l * nx + a] * nun[l]; }
          else { for (int l = 0; l < nx; l++) r += w->Bm[(size_t)k * nx * nw + l * nw + (a - nx)] * nun[l]; }
        }
        if (a < nx) {
          if (k == 0) continue; /* x_1 is fixed: no stationarity condition */
          r -= w->nu[(size_t)k * nx + a];
        }
        if (fabs(r) > res_stat) res_stat = fabs(r);
      }
    }
    st->res_stat = res_stat; st->res_eq = res_eq; st->res_ineq = res_ineq; st->res_comp = res_comp;
    st->obj = obj; st->mu = mu;
    if (trace) {
      double *tr = trace + (size_t)it * ORC_TRACE_W;
      tr[0] = res_stat; tr[1] = res_eq; tr[2] = res_ineq; tr[3] = res_comp; tr[4] = mu; tr[5] = 0; tr[6] = obj; tr[7] = 0;
    }
    if (orc_trace_stderr())
      fprintf(stderr, "orc it %2d stat %.2e eq %.2e ineq %.2e comp %.2e mu %.2e obj %.10e rho %.2e\n", it, res_stat, res_eq,
              res_ineq, res_comp, mu, obj, rho);
    if (!isfinite(res_stat) || !isfinite(res_eq) || !isfinite(res_ineq)) { exitflag = -6; break; }
    if (res_stat <= d->tol_stat && res_eq <= d->tol_eq && res_ineq <= d->tol_ineq && res_comp <= ORC_COMP_FRAC * d->tol_comp) {
      exitflag = 1;
      break;
    }
    /* acceptable termination (cf. IPOPT acceptable_iter): feasible, complementary and the
     * objective has not moved for acc_iters consecutive iterations */
    if (it >= 1 && res_eq <= ORC_ACC_FEAS && res_ineq <= ORC_ACC_FEAS && res_comp <= ORC_ACC_FEAS &&
        fabs(obj - obj_prev) <= d->acc_obj_tol * fmax(1.0, fabs(obj)))
      stall++;
    else
      stall = 0;
    obj_prev = obj;
    if (d->acc_iters > 0 && stall >= d->acc_iters) { exitflag = 2; break; }
    if (it >= d->max_iter) { exitflag = 0; break; }
    /* ---- step computation: exact constraint curvature first (when the model
     * qualifies and no fallback is latched), Gauss-Newton blocks otherwise ---- */
    int use_curv = curv_ok && !gn_sticky && mu <= ORC_CURV_MU;
    if (use_curv && backoff_model(d) && curv_skip > 0) { curv_skip--; use_curv = 0; }
    double alpha = 0.0, ad = 1.0;
    int ls = 0, accepted = 0, fatal = 0;
    double theta_c = cscale ? theta_mem : 1.0;
    int theta_retry = 0;
    for (;;) {
      for (int k = 0; k < N; k++) {
        const double *Jg = w->Jg + (size_t)k * MRM * nv;
        double *Q = w->Q + (size_t)k * nv * nv, *qv = w->qv + (size_t)k * nv;
        memcpy(Q, w->H + (size_t)k * nv * nv, sizeof(double) * nv * nv);
        memcpy(qv, w->gf + (size_t)k * nv, sizeof(double) * nv);
        for (int i = 0; i < m; i++) {
          double tv = w->t[(size_t)k * m + i], lv = w->lam[(size_t)k * m + i];
          double rg = w->g[(size_t)k * MRM + i] - tv;
          double sig = lv / tv, cq = (mu - lv * rg) / tv;
          const double *jr = Jg + i * nv;
          for (int a2 = 0; a2 < nv; a2++) {
            if (jr[a2] == 0.0) continue;
            qv[a2] -= jr[a2] * cq;
            for (int b2 = 0; b2 < nv; b2++) Q[a2 * nv + b2] += sig * jr[a2] * jr[b2];
          }
          if (use_curv && i < nh) {
            /* - (lambda_i + cN/h^2) * grad^2 h_i, distance rows with affine kinematics */
            const double *Cm = w->Cc + ((size_t)k * ORC_NH_MAX + i) * 64;
            double wgt = lv + w->cw[(size_t)k * ORC_NH_MAX + i];
            for (int a2 = 0; a2 < d->n; a2++)
              for (int b2 = 0; b2 < d->n; b2++) Q[a2 * nv + b2] -= theta_c * (wgt * Cm[a2 * d->n + b2]);
          }
        }
        if (use_curv && d->has_goal) {
          const double *G = w->Gc + (size_t)k * 64;
          for (int a2 = 0; a2 < d->n; a2++)
            for (int b2 = 0; b2 < d->n; b2++) Q[a2 * nv + b2] += theta_c * G[a2 * d->n + b2];
        }
        if (use_curv && dd_curv(d) && k < N - 1) dd_dyn_curv(d, w->z + (size_t)k * nv, w->nu + (size_t)(k + 1) * nx, Q);
      }
      if (riccati(w) != 0) {
        if (use_curv && cscale && theta_c > ORC_CS_MIN) { theta_c *= 0.5; theta_retry = 1; tl_passes++; continue; }
        if (use_curv) {
          use_curv = 0;
          tl_passes++;
          if (backoff_model(d)) { curv_back = curv_back ? (curv_back < 16 ? 2 * curv_back : 16) : 1; curv_skip = curv_back; }
          continue; /* this iteration only */
        }
        fatal = 1;
        break;
      }
      /* ---- slack / multiplier steps, fraction to the boundary ---- */
      double ap = 1.0, gphi = 0.0;
      ad = 1.0;
      for (int k = 0; k < N; k++) {
        const double *Jg = w->Jg + (size_t)k * MRM * nv, *dz = w->dz + (size_t)k * nv;
        for (int a2 = 0; a2 < nv; a2++) gphi += w->gf[(size_t)k * nv + a2] * dz[a2];
        for (int i = 0; i < m; i++) {
          double tv = w->t[(size_t)k * m + i], lv = w->lam[(size_t)k * m + i];
          double dt = w->g[(size_t)k * MRM + i] - tv;
          for (int a2 = 0; a2 < nv; a2++) dt += Jg[i * nv + a2] * dz[a2];
          double dl = (mu - tv * lv - lv * dt) / tv;
          w->dtt[(size_t)k * m + i] = dt;
          w->dlam[(size_t)k * m + i] = dl;
          if (dt < 0) { double a3 = -tau * tv / dt; if (a3 < ap) ap = a3; }
          if (dl < 0) { double a3 = -tau * lv / dl; if (a3 < ad) ad = a3; }
          gphi -= mu * dt / tv;
        }
      }
      /* ---- l1 merit, Armijo backtracking ---- */
      if (theta > 1e-13) {
        double need = gphi / (0.9 * theta);
        if (rho < need) rho = need + 1.0;
      }
      double D = gphi - rho * theta;
      double phi0 = obj - mu * logsum + rho * theta;
      accepted = 0;
      const int ls_cap = use_curv ? ORC_LS_CURV - 1 : (d->ls_max > 0 ? d->ls_max : ORC_LS_MAX);
      /* step-length memory: a Gauss-Newton line search starts one halving above the one accepted last;
       * a step with the exact curvature is tried at full length first */
      const int ls_begin = use_curv ? 0 : ls_start;
      alpha = ldexp(ap, -ls_begin);
      for (ls = ls_begin; ls <= ls_cap; ls++) {
        for (size_t i = 0; i < (size_t)N * nv; i++) w->zt[i] = w->z[i] + alpha * w->dz[i];
        for (size_t i = 0; i < (size_t)N * m; i++) w->tt[i] = w->t[i] + alpha * w->dtt[i];
        double ft, tht, lst;
        tl_passes++;
        if (trial_merit(d, w, params, mu, &ft, &tht, &lst) == 0) {
          double phi = ft - mu * lst + rho * tht;
          if (phi <= phi0 + ORC_ARMIJO * alpha * D + 1e-13 * fabs(phi0)) { accepted = 1; break; }
        }
        alpha *= 0.5;
      }
      if (!accepted && use_curv) {
        /* Gauss-Newton fallback for this iteration; latched after repeated failures (the unicycle: backed off instead) */
        if (backoff_model(d)) { curv_back = curv_back ? (curv_back < 16 ? 2 * curv_back : 16) : 1; curv_skip = curv_back; }
        else if (++curv_fail >= ORC_CURV_FAIL_MAX) gn_sticky = 1;
        tl_passes++;
        use_curv = 0;
        continue;
      }
      if (accepted && use_curv) { curv_fail = 0; curv_back = 0; }
      if (cscale) {
        if (accepted && use_curv) {
          if (theta_retry) { theta_mem = theta_c; theta_clean = 0; }
          else if (++theta_clean >= ORC_CS_CLEAN) { theta_mem = theta_c < 0.75 ? 2.0 * theta_c : 1.0; theta_clean = 0; }
        } else if (accepted && theta_retry) { theta_mem = theta_c; theta_clean = 0; }
      }
      /* the arms: a Gauss-Newton step accepted at full length releases the latch (the failures that set it
       * belong to the first iterations of a warm start, where the fraction to the boundary cuts the steps) */
      if (accepted && !use_curv && ls == 0 && arm_curv(d)) { gn_sticky = 0; curv_fail = 0; }
      if (accepted) ls_start = ls > ORC_LS_GROW ? ls - ORC_LS_GROW : 0;
      break;
    }
    if (fatal) { exitflag = -5; break; }
    if (!accepted) { exitflag = -8; break; } /* line search failure */
    if (trace) { trace[(size_t)it * ORC_TRACE_W + 5] = alpha; trace[(size_t)it * ORC_TRACE_W + 7] = ls; }
    if (orc_trace_stderr()) fprintf(stderr, "       step alpha %.3e ls %d ad %.3e curv %d\n", alpha, ls, ad, use_curv);
    memcpy(w->z, w->zt, sizeof(double) * N * nv);
    memcpy(w->t, w->tt, sizeof(double) * N * m);
    for (size_t i = 0; i < (size_t)N * m; i++) w->lam[i] += ad * w->dlam[i];
    for (size_t i = 0; i < (size_t)N * nx; i++) w->nu[i] += alpha * (w->nunew[i] - w->nu[i]);
    /* ---- barrier update: LOQO-style centrality rule with floor ---- */
    {
      double sum = 0, mn = 1e300;
      size_t cnt = (size_t)N * m;
      for (size_t i = 0; i < cnt; i++) {
        double c = w->t[i] * w->lam[i];
        sum += c;
        if (c < mn) mn = c;
      }
      double avg = sum / (double)cnt;
      double xi = mn / avg;
      double sg = 0.05 * (1.0 - xi) / xi;
      if (sg > 2.0) sg = 2.0;
      sg = 0.1 * sg * sg * sg;
      if (sg < 0.02) sg = 0.02; /* floor: never drop mu by more than 50x in one step */
      if (sg > 0.8) sg = 0.8;
      mu = sg * avg;
      if (mu < 0.1 * d->tol_comp) mu = 0.1 * d->tol_comp;
    }
    /* ---- barrier restart on stalled steps ---- */
    {
      small_steps = (it >= ORC_RS_IT && alpha < ORC_RS_ALPHA) ? small_steps + 1 : 0;
      if (small_steps >= ORC_RS_N && mu < ORC_RS_MU && !(mu_hold > 0.0)) { mu_hold = ORC_RS_MU; small_steps = 0; }
      if (mu_hold > 0.0) {
        if (mu < mu_hold) mu = mu_hold;
        mu_hold *= ORC_RS_DECAY;
        if (mu_hold < 0.1 * d->tol_comp) mu_hold = 0.0;
      }
    }
    if (!(mu < ORC_MU_DIVERGED)) { exitflag = -7; it++; break; } /* infeasible / diverged */
    ev = eval_all(d, w, params);
    if (ev != 0) { exitflag = (ev == ORC_EVAL_BAD_AVOID) ? -7 : -10; break; }
  }
done:
  st->exitflag = exitflag;
  st->iters = it;
  st->mu = mu;
  memcpy(zout, w->z, sizeof(double) * N * nv);
  if (lam_out) memcpy(lam_out, w->lam, sizeof(double) * N * m);
  if (nu_out) memcpy(nu_out, w->nu, sizeof(double) * N * nx);
  return 0;
}

int orc_solve_batch(const orc_desc *d, int B, const double *xinit, const double *x0,
                    const double *params, double *zout, orc_stats *st, int nthreads) {
  const int N = d->N, nx = d->nx, nv = nvar_of(d);
  int err = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 4)
  for (int b = 0; b < B; b++) {
    int r = orc_solve(d, xinit + (size_t)b * nx, x0 + (size_t)b * N * nv,
                      params + (size_t)b * N * d->npar, zout + (size_t)b * N * nv, st + b, 0);
    if (r != 0) err = r;
  }
  return err;
}
