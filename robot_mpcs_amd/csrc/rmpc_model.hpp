// rmpc_model.hpp -- hand-lowered robot models for the MI355X MPC solver.
//
// What the reference expresses as CasADi SX graphs and hands to FORCES Pro
// (robotmpcs/models/*.py) is written here as plain fp64 device functions over
// compile-time dimensions: forward kinematics with analytic position Jacobians
// (forwardkinematics' GenericURDFFk call sites mpcBase.py:89-94,
// goal_reaching.py:22-27, LinearConstraints.py:30-35,
// SelfCollisionAvoidanceConstraints.py:23-24), the continuous models
// (mpcModel.py:65-69, diff_drive_mpc_model.py:24-41) and their ERK2 / 5-node
// discretisation (mpcModel.py:118-120).  Everything a lane touches lives in
// registers: arrays are statically indexed in fully unrolled loops.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rmpc.h"

namespace rmpc {

constexpr int kMaxRows = 64;  // general inequality rows per stage
constexpr int kErkNodes = 5;

enum RowKind : int8_t { ROW_RADIAL = 0, ROW_LINEAR = 1, ROW_SELF = 2, ROW_SINGLE = 3 };

// Uniform (kernarg) model: the descriptor plus derived row tables.
struct DevModel {
  int robot, N, n, nx, nu, ns, nv, nw, npar;
  int nh, m, nlb, nub, nfk;
  double dt;
  int n_modules;
  int mod_kind[RMPC_MAX_MODULES], mod_row0[RMPC_MAX_MODULES], mod_rows[RMPC_MAX_MODULES];
  int nobst;
  int end_frame;
  int n_joints;
  int joint_type[RMPC_MAX_JOINTS];
  double joint_xyz[RMPC_MAX_JOINTS][3];
  double joint_rot[RMPC_MAX_JOINTS][9];
  double joint_axis[RMPC_MAX_JOINTS][3];
  double dd_off[RMPC_MAX_JOINTS][3];  // diff-drive: frame offset in the base frame
  int off_r_body, off_obst, off_lin, off_wu, off_goal, off_wgoal, off_wconstr, off_ws;
  int has_goal, has_avoid;
  // general rows, YAML module order (InequalityManager.py:25-33)
  int8_t row_kind[kMaxRows];
  int8_t row_a[kMaxRows];     // radial/linear: frame; self: frame A; single: variable index
  int8_t row_b[kMaxRows];     // radial/linear: obstacle; self: frame B; single: sign
  int16_t row_poff[kMaxRows]; // single: parameter offset of the limit value
  int8_t row_fk[kMaxRows];    // index among the FK rows, -1 otherwise
  int8_t row_mod[kMaxRows];   // owning module
  // finite simple bounds (mpcModel.py:91-104): lower rows first, then upper rows
  int8_t lb_var[RMPC_NV_MAX], ub_var[RMPC_NV_MAX];
  double lb_val[RMPC_NV_MAX], ub_val[RMPC_NV_MAX];
  int max_iter;
  double tol_stat, tol_eq, tol_ineq, tol_comp, mu0;
  int use_curv;      // exact curvature of the distance rows (affine kinematics, no slack, n <= 3)
  int acc_iters;     // acceptable termination window (0 = off)
  double acc_obj_tol;
  int ls_max;        // step halvings allowed in one line search
};

// Row tables that are too large for the kernel argument live in device memory; they are
// uniform, so the kernels read them through the scalar cache.
constexpr int kMaxSlots = 4;   // distinct points (frame, or frame pair) the FK rows and the goal refer to
constexpr int kMaxFkRows = 40;
constexpr int kVarRows = 4;    // single-variable rows per variable: limit lower/upper, bound lower/upper
struct DevTables {
  // All fields are 32-bit (or double): sub-dword fields cannot be fetched through the scalar
  // cache on gfx950 and would turn every table access into a vector-memory round trip.
  // kinematic slots: slot 0 is the goal's end frame when GoalReaching is present
  int nslots;
  int slot_fa[kMaxSlots], slot_fb[kMaxSlots];        // frame A, frame B (-1: single frame)
  int slot_row_begin[kMaxSlots + 1];                 // FK rows sorted by slot
  int nfkrows;
  int fk_row[kMaxFkRows];                            // storage index of the row (YAML order)
  int fk_kind[kMaxFkRows], fk_obst[kMaxFkRows], fk_mod[kMaxFkRows], fk_first[kMaxFkRows];
  int fk_idx[kMaxFkRows];                            // index among the FK rows (Jq storage)
  // rows that depend on one variable only, grouped by variable
  int v_row[RMPC_NV_MAX][kVarRows];                  // storage index or -1
  int v_sgn[RMPC_NV_MAX][kVarRows];                  // +1: z - limit, -1: limit - z
  int v_poff[RMPC_NV_MAX][kVarRows];                 // parameter offset of the limit, -1: constant bound
  int v_soft[RMPC_NV_MAX][kVarRows];                 // general row (softened when ns = 1)
  int v_mod[RMPC_NV_MAX][kVarRows];                  // owning module, -1 for simple bounds
  int v_first[RMPC_NV_MAX][kVarRows];                // first row of its module (inverse-barrier objective)
  double v_val[RMPC_NV_MAX][kVarRows];               // constant bound value
  // the same tables packed, one word per row, for lanes that index them with a per-lane variable / row (the (stage,
  // part) lanes of the fused arm kernel: one vector request instead of six).  v_desc: bits 0-7 storage index, 8 present,
  // 9 sign negative, 10 first row of its module, 11 general row (limit from the parameters), 12-14 module, 16-31
  // parameter offset.  fk_desc: bits 0-7 storage index, 8-9 kind, 10-15 obstacle, 16-18 module, 19 first row of its
  // module, 20-25 index among the FK rows.
  int v_desc[RMPC_NV_MAX][kVarRows];
  int fk_desc[kMaxFkRows];
  int slot_rows_max;                                 // most FK rows of one slot
};

// ---------------------------------------------------------------------------
// Model views.  The device functions read the structure of the problem (row tables, parameter offsets, joint
// chain) through a "view":
//   * RtView  -- the runtime tables: any model the descriptor can express runs through it;
//   * a generated view (rmpc_spec_gen.hpp, written by rmpc_spec_source from the descriptors of the shipped
//     configurations) -- the same accessors as constexpr functions over literal tables.  After unrolling, the
//     row loops of the sweep and the step phase are straight-line code: no table reads, no uniform branches
//     around the requests, no masked rows, constant joint transforms.  This is the counterpart of the code
//     the reference has FORCES Pro GENERATE for one problem (makeSolver.py / mpcModel.py:139-160 generateSolver);
//     rmpc_create picks a generated view when its tables equal the descriptor's, the runtime view otherwise.
// ---------------------------------------------------------------------------
struct RtView {
  static constexpr bool SPEC = false;
  const DevModel &M;
  const DevTables &T;
  __device__ __forceinline__ RtView(const DevModel &m, const DevTables &t) : M(m), T(t) {}
  __device__ __forceinline__ int nslots() const { return T.nslots; }
  __device__ __forceinline__ int slot_fa(int s) const { return T.slot_fa[s]; }
  __device__ __forceinline__ int slot_fb(int s) const { return T.slot_fb[s]; }
  __device__ __forceinline__ int slot_row_begin(int s) const { return T.slot_row_begin[s]; }
  __device__ __forceinline__ int nfkrows() const { return T.nfkrows; }
  __device__ __forceinline__ int fk_row(int r) const { return T.fk_row[r]; }
  __device__ __forceinline__ int fk_kind(int r) const { return T.fk_kind[r]; }
  __device__ __forceinline__ int fk_obst(int r) const { return T.fk_obst[r]; }
  __device__ __forceinline__ int fk_mod(int r) const { return T.fk_mod[r]; }
  __device__ __forceinline__ int fk_first(int r) const { return T.fk_first[r]; }
  __device__ __forceinline__ int fk_idx(int r) const { return T.fk_idx[r]; }
  __device__ __forceinline__ int v_row(int j, int u) const { return T.v_row[j][u]; }
  __device__ __forceinline__ int v_sgn(int j, int u) const { return T.v_sgn[j][u]; }
  __device__ __forceinline__ int v_poff(int j, int u) const { return T.v_poff[j][u]; }
  __device__ __forceinline__ int v_soft(int j, int u) const { return T.v_soft[j][u]; }
  __device__ __forceinline__ int v_mod(int j, int u) const { return T.v_mod[j][u]; }
  __device__ __forceinline__ int v_first(int j, int u) const { return T.v_first[j][u]; }
  __device__ __forceinline__ double v_val(int j, int u) const { return T.v_val[j][u]; }
  __device__ __forceinline__ int v_desc(int j, int u) const { return T.v_desc[j][u]; }
  __device__ __forceinline__ int fk_desc(int r) const { return T.fk_desc[r]; }
  __device__ __forceinline__ int slot_rows_max() const { return T.slot_rows_max; }
  __device__ __forceinline__ int n_rows() const { return M.m; }
  __device__ __forceinline__ int n_general() const { return M.nh; }
  __device__ __forceinline__ int n_fk() const { return M.nfk; }
  __device__ __forceinline__ int off_r_body() const { return M.off_r_body; }
  __device__ __forceinline__ int off_obst() const { return M.off_obst; }
  __device__ __forceinline__ int off_lin() const { return M.off_lin; }
  __device__ __forceinline__ int off_wu() const { return M.off_wu; }
  __device__ __forceinline__ int off_goal() const { return M.off_goal; }
  __device__ __forceinline__ int off_wgoal() const { return M.off_wgoal; }
  __device__ __forceinline__ int off_wconstr() const { return M.off_wconstr; }
  __device__ __forceinline__ int off_ws() const { return M.off_ws; }
  __device__ __forceinline__ int has_goal() const { return M.has_goal; }
  __device__ __forceinline__ int has_avoid() const { return M.has_avoid; }
  __device__ __forceinline__ int joint_type(int j) const { return M.joint_type[j]; }
  __device__ __forceinline__ double joint_xyz(int j, int c) const { return M.joint_xyz[j][c]; }
  __device__ __forceinline__ double joint_rot(int j, int c) const { return M.joint_rot[j][c]; }
  __device__ __forceinline__ double joint_axis(int j, int c) const { return M.joint_axis[j][c]; }
  __device__ __forceinline__ double dd_off(int f, int c) const { return M.dd_off[f][c]; }
};

// The runtime tables for the phase FUNCTIONS of the fused arm kernel: model and tables are copies in device memory,
// reached through uniform pointers in the constant address space (a callee gets its arguments in vector registers;
// through the constant address space the tables still come by scalar loads).  Same accessors as RtView.
struct GView {
  static constexpr bool SPEC = false;
  typedef const __attribute__((address_space(4))) DevModel cModel;
  typedef const __attribute__((address_space(4))) DevTables cTables;
  cModel *M;
  cTables *T;
  __device__ __forceinline__ GView(cModel *m, cTables *t) : M(m), T(t) {}
  __device__ __forceinline__ int nslots() const { return T->nslots; }
  __device__ __forceinline__ int slot_fa(int s) const { return T->slot_fa[s]; }
  __device__ __forceinline__ int slot_fb(int s) const { return T->slot_fb[s]; }
  __device__ __forceinline__ int slot_row_begin(int s) const { return T->slot_row_begin[s]; }
  __device__ __forceinline__ int nfkrows() const { return T->nfkrows; }
  __device__ __forceinline__ int fk_row(int r) const { return T->fk_row[r]; }
  __device__ __forceinline__ int fk_kind(int r) const { return T->fk_kind[r]; }
  __device__ __forceinline__ int fk_obst(int r) const { return T->fk_obst[r]; }
  __device__ __forceinline__ int fk_mod(int r) const { return T->fk_mod[r]; }
  __device__ __forceinline__ int fk_first(int r) const { return T->fk_first[r]; }
  __device__ __forceinline__ int fk_idx(int r) const { return T->fk_idx[r]; }
  __device__ __forceinline__ int v_row(int j, int u) const { return T->v_row[j][u]; }
  __device__ __forceinline__ int v_sgn(int j, int u) const { return T->v_sgn[j][u]; }
  __device__ __forceinline__ int v_poff(int j, int u) const { return T->v_poff[j][u]; }
  __device__ __forceinline__ int v_soft(int j, int u) const { return T->v_soft[j][u]; }
  __device__ __forceinline__ int v_mod(int j, int u) const { return T->v_mod[j][u]; }
  __device__ __forceinline__ int v_first(int j, int u) const { return T->v_first[j][u]; }
  __device__ __forceinline__ double v_val(int j, int u) const { return T->v_val[j][u]; }
  __device__ __forceinline__ int v_desc(int j, int u) const { return T->v_desc[j][u]; }
  __device__ __forceinline__ int fk_desc(int r) const { return T->fk_desc[r]; }
  __device__ __forceinline__ int slot_rows_max() const { return T->slot_rows_max; }
  __device__ __forceinline__ int n_rows() const { return M->m; }
  __device__ __forceinline__ int n_general() const { return M->nh; }
  __device__ __forceinline__ int n_fk() const { return M->nfk; }
  __device__ __forceinline__ int off_r_body() const { return M->off_r_body; }
  __device__ __forceinline__ int off_obst() const { return M->off_obst; }
  __device__ __forceinline__ int off_lin() const { return M->off_lin; }
  __device__ __forceinline__ int off_wu() const { return M->off_wu; }
  __device__ __forceinline__ int off_goal() const { return M->off_goal; }
  __device__ __forceinline__ int off_wgoal() const { return M->off_wgoal; }
  __device__ __forceinline__ int off_wconstr() const { return M->off_wconstr; }
  __device__ __forceinline__ int off_ws() const { return M->off_ws; }
  __device__ __forceinline__ double dd_off(int f, int c) const { return M->dd_off[f][c]; }
  __device__ __forceinline__ int has_goal() const { return M->has_goal; }
  __device__ __forceinline__ int has_avoid() const { return M->has_avoid; }
  __device__ __forceinline__ int joint_type(int j) const { return M->joint_type[j]; }
  __device__ __forceinline__ double joint_xyz(int j, int c) const { return M->joint_xyz[j][c]; }
  __device__ __forceinline__ double joint_rot(int j, int c) const { return M->joint_rot[j][c]; }
  __device__ __forceinline__ double joint_axis(int j, int c) const { return M->joint_axis[j][c]; }
};

struct Vec3 {
  double x, y, z;
};
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Compile-time configuration of one robot family.
template <int ROBOT_, int NQ_, int NS_>
struct Cfg {
  static constexpr int ROBOT = ROBOT_;
  static constexpr int NQ = NQ_;
  static constexpr int NS = NS_;
  static constexpr int NX = ROBOT_ == RMPC_ROBOT_CHAIN ? 2 * NQ_ : 2 * NQ_ + 2;
  static constexpr int NU = ROBOT_ == RMPC_ROBOT_CHAIN ? NQ_ : 2;
  static constexpr int NV = NX + NS_ + NU;
  static constexpr int NW = NS_ + NU;
  static constexpr int NQ2 = NQ_ * (NQ_ + 1) / 2;
  static constexpr int NR = 5;  // reduced diff-drive state (x, y, theta, v, omega)
  // exact curvature of the distance rows / inverse-barrier objective (Cqq block of the stage record): holonomic
  // chains without slack.  Three joints: only when every frame moves affinely with q (decided per descriptor,
  // DevModel::use_curv).  The arms: always, with the second derivatives of the kinematics themselves (FKCURV:
  // distance rows, inverse-barrier objective and the goal cost) -- without them a warm-started arm crawls to the
  // tolerance at a linear rate of 0.6 per iteration.
  static constexpr bool CURV = (ROBOT_ == RMPC_ROBOT_CHAIN) && (NS_ == 0);
  static constexpr bool FKCURV = CURV && (NQ_ > 3);
  // The diff-drive base (round 4): exact second-order terms of the unicycle -- nu . grad^2 Phi of the discrete dynamics
  // (only x+ and y+ are curved: block over theta, omega, u1 | v, u0, record entries R_D) and the rotation of off-centre
  // frames (d2 p / dtheta2, in the q block R_C together with the distance rows' own curvature).  Without them the
  // Gauss-Newton blocks of the unicycle converge linearly: cold boxers 20.6 -> 15.9 iterations, every instance to the
  // 1e-6 tolerance instead of 9 in 10 (oracle, 256 instances of BASELINE configs[2]).
  static constexpr bool DDCURV = (ROBOT_ == RMPC_ROBOT_DIFFDRIVE);
  // a failed curvature step switches the terms off for 1, 2, 4 .. 16 iterations (back-off) instead of latching them
  // off: the unicycle and the small chains; the arms keep the latch with its release rule (oracle: backoff_model)
  static constexpr bool BACKOFF = DDCURV || (CURV && !FKCURV);
  // scaled curvature (the small chains; oracle: cscale_model, ORC_CS_*): a failed factorisation with the exact curvature is
  // retried with the curvature terms at 1/2, 1/4 of their weight before the iteration falls back to Gauss-Newton
  static constexpr bool CSCALE = CURV && !FKCURV;
  // fraction to the boundary (oracle: ORC_TAU_*): a step takes a slack or a multiplier to (1 - TAU) of its value at most
  static constexpr double TAU = CSCALE ? 0.985 : 0.98;
  static constexpr int ND = 11;   // (th,om) (th,u1) (om,om) (om,u1) (u1,u1) | (th,v) (th,u0) (om,v) (om,u0) (u1,v) (u1,u0)
  // instances (wavefronts) per block of the grouped Riccati kernel.  Small blocks: with the instance-major
  // records neighbouring instances no longer share cache lines, a 4-wavefront block fits beside a k_sweep
  // wavefront of another batch on every SIMD (a 16-wavefront block needs four free slots per SIMD at once);
  // the arm runs one-wavefront blocks only (no 512-thread register limit, no spills, one launch per pass)
  static constexpr int IPB = (NX > 8) ? 1 : 4;
  // k_sweep register budget: uncapped (one wavefront per SIMD, ~390 VGPRs, no spills).  A 256-VGPR cap (two
  // wavefronts per SIMD) shortens a lone sweep slightly but spills and fills every register file, so that no
  // wavefront of another batch's k_riccati (96 VGPRs) can run beside it: -13 % throughput with four batches in flight
  static constexpr int SWEEP_WPE = 1;
  // k_riccati: no register cap (94 VGPRs for the point robot) -- at 80 the recursion spills inside its stage
  // loop: -3 % throughput on cfg2, -20 % on cfg3 (measured); the arm at two wavefronts per SIMD (256 registers,
  // 380 B scratch): 84 -> 154 us per launch of 1024 instances in round 2; with round 3's recursion (300 registers
  // uncapped; capped 256 + 108 B scratch): four batches in flight 0.497 -> 0.455 M solves/s, a lone batch 4.50 -> 4.61 ms
  static constexpr int RIC_WPE = 1;
  // lanes per instance in the grouped Riccati blocks: half a wavefront for the small models
  static constexpr int RIC_LPI = (NX + NS_ + NU <= 16) ? 32 : 64;
  // Stage record handed from k_sweep to k_riccati, one per (instance, stage), instance-major:
  //   Qqq (upper triangle) | Cqq | Dg (variables >= NQ) | cs (softened models) | q0 | q1 | rc | A5 B5 (diff-drive)
  // A lane of k_sweep scatters its stage's entries into the record; a wavefront of k_riccati then
  // fetches a whole record with one contiguous request instead of one cache line per entry.
  static constexpr int R_Q = 0;
  static constexpr int R_C = R_Q + NQ2;
  static constexpr int R_DG = R_C + NQ2;
  static constexpr int R_CS = R_DG + (NV - NQ_);
  static constexpr int R_Q0 = R_CS + (NS_ > 0 ? NV : 0);
  static constexpr int R_Q1 = R_Q0 + NV;
  static constexpr int R_RC = R_Q1 + NV;
  static constexpr int R_A5 = R_RC + NX;
  static constexpr int R_B5 = R_A5 + 25;
  static constexpr int R_D = R_B5 + 10;           // (diff-drive) MINUS the dynamics' curvature entries: Q = r0 - cwt r1 adds them
  static constexpr int RW = R_A5 + (ROBOT_ == RMPC_ROBOT_DIFFDRIVE ? 35 + ND : 0);
  static constexpr int R_ZERO = RW;             // always 0.0: source of the structural zeros of the dense blocks
  static constexpr int RS = (RW + 1 + 7) / 8 * 8;   // record stride (doubles)
  // fused kernel: the stage records of the owner wavefront's two instances stay in LDS when they are small
  // (point robot: 2 x 32 x 48 doubles = 24 KB per wavefront; the diff-drive records carry A5 | B5 and would
  // leave room for two wavefronts per CU only -- they go through the instance's block of the workspace)
  static constexpr bool FUSED_REC_LDS = (ROBOT_ == RMPC_ROBOT_CHAIN) && (NQ_ <= 3);
  static constexpr bool FUSED_OK = !(ROBOT_ == RMPC_ROBOT_CHAIN && NQ_ > 3);   // models k_fused is built for
  // the arms k_fused_arm is built for (rmpc_arm_fused.hpp): holonomic chains without slack whose state and gradient
  // column fit one 16 x 16 tile of the matrix cores (n = 5, 6, 7) -- the models of riccati_recursion's arm path
  static constexpr bool ARM_FUSED = (ROBOT_ == RMPC_ROBOT_CHAIN) && NS_ == 0 && NX > 8 && NX < 16;
};

// ---------------------------------------------------------------------------
// Kinematics: one pass over the chain.  Everything is indexed by unrolled loop
// counters only (a uniform *runtime* index into a register array is turned into
// scratch memory by the compiler), so the points the rows need are captured
// into "slots" while the chain is walked.
// ---------------------------------------------------------------------------
template <class C>
struct Kin {
  static constexpr bool CHAIN = (C::ROBOT == RMPC_ROBOT_CHAIN);
  Vec3 oj[CHAIN ? C::NQ : 1];   // joint origins
  Vec3 aj[CHAIN ? C::NQ : 1];   // joint axes (world)
  Vec3 pa[kMaxSlots], pb[kMaxSlots];  // positions of frame A / frame B of every slot
  double c, s, qx, qy;          // diff-drive base pose

  template <class V>
  __device__ __forceinline__ void compute(const V &v, const double (&q)[C::NQ]) {
#pragma unroll
    for (int sl = 0; sl < kMaxSlots; sl++) { pa[sl] = {0, 0, 0}; pb[sl] = {0, 0, 0}; }
    if constexpr (CHAIN) {
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      Vec3 o = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < C::NQ; j++) {
        const double t[3] = {v.joint_xyz(j, 0), v.joint_xyz(j, 1), v.joint_xyz(j, 2)};
        o.x += R[0] * t[0] + R[1] * t[1] + R[2] * t[2];
        o.y += R[3] * t[0] + R[4] * t[1] + R[5] * t[2];
        o.z += R[6] * t[0] + R[7] * t[1] + R[8] * t[2];
        {
          double Rj[9];
#pragma unroll
          for (int c = 0; c < 9; c++) Rj[c] = v.joint_rot(j, c);
          mul33(R, Rj);
        }
        const double ax[3] = {v.joint_axis(j, 0), v.joint_axis(j, 1), v.joint_axis(j, 2)};
        Vec3 a = {R[0] * ax[0] + R[1] * ax[1] + R[2] * ax[2], R[3] * ax[0] + R[4] * ax[1] + R[5] * ax[2],
                  R[6] * ax[0] + R[7] * ax[1] + R[8] * ax[2]};
        aj[j] = a;
        oj[j] = o;
        if (v.joint_type(j) == RMPC_JOINT_REVOLUTE) {
          double Rq[9];
          rodrigues(ax, q[j], Rq);
          mul33(R, Rq);
        } else if (v.joint_type(j) == RMPC_JOINT_PRISMATIC) {
          o.x += a.x * q[j];
          o.y += a.y * q[j];
          o.z += a.z * q[j];
        }
#pragma unroll
        for (int sl = 0; sl < kMaxSlots; sl++) {
          if (v.slot_fa(sl) == j) pa[sl] = o;
          if (v.slot_fb(sl) == j) pb[sl] = o;
        }
      }
    } else {
      sincos(q[2], &s, &c);
      qx = q[0];
      qy = q[1];
#pragma unroll
      for (int sl = 0; sl < kMaxSlots; sl++) {
        if (sl < v.nslots()) {
          const int fa = v.slot_fa(sl), fb = v.slot_fb(sl);
          const double o[3] = {v.dd_off(fa, 0), v.dd_off(fa, 1), v.dd_off(fa, 2)};
          pa[sl] = {qx + c * o[0] - s * o[1], qy + s * o[0] + c * o[1], o[2]};
          if (fb >= 0) {
            const double ob[3] = {v.dd_off(fb, 0), v.dd_off(fb, 1), v.dd_off(fb, 2)};
            pb[sl] = {qx + c * ob[0] - s * ob[1], qy + s * ob[0] + c * ob[1], ob[2]};
          }
        }
      }
    }
  }

  // Point of slot sl (static sl): P = pos(A) - pos(B) (B absent: pos(A)) and dP/dq.
  template <int SL, class V>
  __device__ __forceinline__ Vec3 point(const V &v, Vec3 (&J)[C::NQ]) const {
    const int fa = v.slot_fa(SL), fb = v.slot_fb(SL);
    Vec3 P = pa[SL];
    if constexpr (CHAIN) {
#pragma unroll
      for (int d = 0; d < C::NQ; d++) {
        Vec3 col = {0, 0, 0};
        if (d <= fa) {
          if (v.joint_type(d) == RMPC_JOINT_REVOLUTE) col = cross(aj[d], pa[SL] - oj[d]);
          else if (v.joint_type(d) == RMPC_JOINT_PRISMATIC) col = aj[d];
        }
        if (fb >= 0 && d <= fb) {
          Vec3 cb = {0, 0, 0};
          if (v.joint_type(d) == RMPC_JOINT_REVOLUTE) cb = cross(aj[d], pb[SL] - oj[d]);
          else if (v.joint_type(d) == RMPC_JOINT_PRISMATIC) cb = aj[d];
          col = col - cb;
        }
        J[d] = col;
      }
    } else {
      const double o[3] = {v.dd_off(fa, 0), v.dd_off(fa, 1), v.dd_off(fa, 2)};
      J[0] = {1, 0, 0};
      J[1] = {0, 1, 0};
      J[2] = {-s * o[0] - c * o[1], c * o[0] - s * o[1], 0};
      if (fb >= 0) {
        const double ob[3] = {v.dd_off(fb, 0), v.dd_off(fb, 1), v.dd_off(fb, 2)};
        J[0] = {0, 0, 0};
        J[1] = {0, 0, 0};
        J[2] = {J[2].x - (-s * ob[0] - c * ob[1]), J[2].y - (c * ob[0] - s * ob[1]), 0};
      }
    }
    if (fb >= 0) P = P - pb[SL];
    return P;
  }

 public:   // (also used by the part-wise sweep of the fused arm kernel)
  __device__ __forceinline__ static void mul33(double (&R)[9], const double *Bm) {
    double r[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) r[3 * i + j] = R[3 * i] * Bm[j] + R[3 * i + 1] * Bm[3 + j] + R[3 * i + 2] * Bm[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = r[i];
  }
  __device__ __forceinline__ static void rodrigues(const double *k, double th, double (&R)[9]) {
    double cs, sn;
    sincos(th, &sn, &cs);   // one range reduction for both
    const double v = 1.0 - cs;
    R[0] = cs + k[0] * k[0] * v;        R[1] = k[0] * k[1] * v - k[2] * sn; R[2] = k[0] * k[2] * v + k[1] * sn;
    R[3] = k[1] * k[0] * v + k[2] * sn; R[4] = cs + k[1] * k[1] * v;        R[5] = k[1] * k[2] * v - k[0] * sn;
    R[6] = k[2] * k[0] * v - k[1] * sn; R[7] = k[2] * k[1] * v + k[0] * sn; R[8] = cs + k[2] * k[2] * v;
  }
};

// ---------------------------------------------------------------------------
// Discrete dynamics x+ = Phi(x, u): ERK2 (explicit midpoint), 5 nodes.
// ---------------------------------------------------------------------------
// Holonomic chain, double integrator (mpcModel.py:65-69).  The Jacobians are
// the constants A = [I dt I; 0 I], B = [dt^2/2 I; dt I] and are never stored.
template <class C>
__device__ __forceinline__ void chain_step(const double dt, const double (&z)[C::NV], double (&xn)[C::NX]) {
  const double h = dt / kErkNodes;
#pragma unroll
  for (int i = 0; i < C::NX; i++) xn[i] = z[i];
#pragma unroll
  for (int it = 0; it < kErkNodes; it++) {
#pragma unroll
    for (int i = 0; i < C::NQ; i++) {
      const double u = z[C::NX + C::NS + i];
      const double vm = xn[C::NQ + i] + 0.5 * h * u;  // midpoint velocity
      xn[i] += h * vm;
      xn[C::NQ + i] += h * u;
    }
  }
}

// Diff-drive unicycle (diff_drive_mpc_model.py:24-41) on the reduced state
// r = (x, y, theta, v, omega) = x[{0,1,2,6,7}]; x[3:6] is carried unchanged.
// A5 (5x5) and B5 (5x2) are the Jacobians of the reduced map.
template <class C>
__device__ __forceinline__ void diffdrive_step(const double dt, const double (&z)[C::NV], double (&xn)[C::NX],
                                               double (&A5)[25], double (&B5)[10], bool want) {
  const double h = dt / kErkNodes;
  double r[5] = {z[0], z[1], z[2], z[6], z[7]};
  const double u0 = z[C::NX + C::NS], u1 = z[C::NX + C::NS + 1];
  if (want) {
#pragma unroll
    for (int i = 0; i < 25; i++) A5[i] = (i % 6 == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 10; i++) B5[i] = 0.0;
  }
#pragma unroll 1
  for (int it = 0; it < kErkNodes; it++) {
    double c1, s1;
    sincos(r[2], &s1, &c1);
    const double k1[5] = {c1 * r[3], s1 * r[3], r[4], u0, u1};
    double rm[5];
#pragma unroll
    for (int i = 0; i < 5; i++) rm[i] = r[i] + 0.5 * h * k1[i];
    double c2, s2;
    sincos(rm[2], &s2, &c2);
    const double k2[5] = {c2 * rm[3], s2 * rm[3], rm[4], u0, u1};
    if (want) {
      // fx at (r) and (rm): nonzeros (0,2) (0,3) (1,2) (1,3) (2,4)
      const double f1_02 = -s1 * r[3], f1_03 = c1, f1_12 = c1 * r[3], f1_13 = s1;
      const double f2_02 = -s2 * rm[3], f2_03 = c2, f2_12 = c2 * rm[3], f2_13 = s2;
      // M = I + h/2 fx1 ; As = I + h fx2 M ; Bs = h (fx2 (h/2 fu1) + fu2), fu = e3 e0^T + e4 e1^T
      double As[25], Bs[10];
#pragma unroll
      for (int i = 0; i < 25; i++) As[i] = (i % 6 == 0) ? 1.0 : 0.0;
#pragma unroll
      for (int i = 0; i < 10; i++) Bs[i] = 0.0;
      // rows of M that fx2 reads: row 2 = e2 + h/2 * e4 ; row 3 = e3 ; row 4 = e4
      // row 0 of fx2*M = f2_02 * M[2,:] + f2_03 * M[3,:]
      As[0 * 5 + 2] += h * f2_02;
      As[0 * 5 + 4] += h * f2_02 * (0.5 * h);
      As[0 * 5 + 3] += h * f2_03;
      As[1 * 5 + 2] += h * f2_12;
      As[1 * 5 + 4] += h * f2_12 * (0.5 * h);
      As[1 * 5 + 3] += h * f2_13;
      As[2 * 5 + 4] += h;
      (void)f1_02; (void)f1_03; (void)f1_12; (void)f1_13;
      // M rows 0,1 (which carry fx1) are never read by fx2 (its columns 0,1 are zero),
      // so fx1 only enters through M[2,4] = h/2.
      // Bs = h * (fx2 * (h/2) fu1 + fu2): fu1 = fu2 = [e3->u0, e4->u1]
      Bs[0 * 2 + 0] = h * (f2_03 * 0.5 * h);
      Bs[1 * 2 + 0] = h * (f2_13 * 0.5 * h);
      Bs[2 * 2 + 1] = h * (0.5 * h);
      Bs[3 * 2 + 0] = h;
      Bs[4 * 2 + 1] = h;
      double T[25], TB[10];
#pragma unroll
      for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
          double sacc = 0;
#pragma unroll
          for (int l = 0; l < 5; l++) sacc += As[i * 5 + l] * A5[l * 5 + j];
          T[i * 5 + j] = sacc;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          double sacc = Bs[i * 2 + j];
#pragma unroll
          for (int l = 0; l < 5; l++) sacc += As[i * 5 + l] * B5[l * 2 + j];
          TB[i * 2 + j] = sacc;
        }
      }
#pragma unroll
      for (int i = 0; i < 25; i++) A5[i] = T[i];
#pragma unroll
      for (int i = 0; i < 10; i++) B5[i] = TB[i];
    }
#pragma unroll
    for (int i = 0; i < 5; i++) r[i] += h * k2[i];
  }
  xn[0] = r[0]; xn[1] = r[1]; xn[2] = r[2];
  xn[3] = z[3]; xn[4] = z[4]; xn[5] = z[5];
  xn[6] = r[3]; xn[7] = r[4];
}

}  // namespace rmpc
