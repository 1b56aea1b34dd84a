// rmpc_kernels.hip -- batched multiple-shooting interior-point MPC solver for
// MI355X (gfx950).  Replaces the FORCES Pro generated solver behind
// robotmpcs.planner.mpcPlanner.MPCPlanner.solve() (mpcPlanner.py:262).
//
// One solve = pack, then passes of three kernels until every instance has
// stopped, then unpack:
//
//   k_sweep   one lane per (instance, stage): forms the trial point
//             z + alpha dz (t, lambda, nu likewise), evaluates dynamics, cost,
//             inequality rows and their Jacobians there, condenses the barrier
//             terms into the stage record (Hessian / gradient blocks) and writes
//             the merit and KKT partial sums of the stage.   [HBM / request bound]
//   k_riccati one wavefront per instance, stage matrices in LDS: reduces the
//             stage partials, runs the Armijo test on the l1 merit, updates the
//             barrier parameter, checks convergence and runs the block-
//             tridiagonal Riccati recursion (backward, forward, costates).
//                                    [LDS throughput / LDS round-trip latency]
//   k_step    one lane per (instance, stage): fraction-to-the-boundary partial
//             minima of the slack and multiplier steps, merit slope partials.
//                                                                  [HBM bound]
//   (+ k_compact: list of the instances still iterating; k_migrate: their move
//    to a small dense workspace once few are left)
//
// Data layout: what the stage-parallel kernels exchange is [slot][stage][instance]
// with the instance index contiguous (a wavefront's 64 lanes move 512 contiguous
// bytes per slot); what the wave-per-instance kernel reads or keeps is
// [instance][stage][record] (one request per record).  Iterates are double
// buffered per instance (cur / cur^1): a trial point is written once and
// accepted by flipping a bit.  DESIGN.md, sections 4 and 5.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "rmpc_model.hpp"
#include "rmpc_spec_gen.hpp"   // generated views of the shipped configurations (scripts/gen_specs.py)

// ---- kernel variants and translation units ---------------------------------------------------------------------
// One kernel variant per robot family and size: X(id, robot kind, n, ns).  ids 0 .. 5 are the shipped configurations
// (point robot, panda, boxer, each without / with the slack variable); 6 .. 10 further holonomic chains (mpcBase.py:52-55:
// n = fk.n() of whatever URDF chain the YAML names), the sizes the test suite can build from the shipped URDFs
// (tests: chain2, chain4, chain5, chain6), and n = 8 = RMPC_MAX_JOINTS (test: chain8, the panda's chain with one more
// revolute joint).  Another size is one more line here, one more #if block below
// (instantiation list) and one more bit in __graft_entry__.TU_MASKS.
#define RMPC_VARIANTS(X)                                                                                                  \
  X(0, RMPC_ROBOT_CHAIN, 3, 0) X(1, RMPC_ROBOT_CHAIN, 3, 1) X(2, RMPC_ROBOT_CHAIN, 7, 0) X(3, RMPC_ROBOT_CHAIN, 7, 1)    \
  X(4, RMPC_ROBOT_DIFFDRIVE, 3, 0) X(5, RMPC_ROBOT_DIFFDRIVE, 3, 1) X(6, RMPC_ROBOT_CHAIN, 2, 0) X(7, RMPC_ROBOT_CHAIN, 4, 0) \
  X(8, RMPC_ROBOT_CHAIN, 5, 0) X(9, RMPC_ROBOT_CHAIN, 6, 0) X(10, RMPC_ROBOT_CHAIN, 8, 0)
// The library is built from this one source compiled several times in parallel (__graft_entry__.build: one translation
// unit per group of variants, minutes of compile time otherwise): RMPC_DEV_VARIANTS is the bit mask of the variants whose
// kernels THIS translation unit instantiates, RMPC_ALL_VARIANTS the mask of the variants the library holds (what the
// dispatch offers), RMPC_TU_MAIN says whether this unit carries the host side and the variant-independent kernels.
// A plain one-file build (hipcc rmpc_kernels.hip) has all three at their defaults: everything in one unit.
#ifndef RMPC_DEV_VARIANTS
#define RMPC_DEV_VARIANTS 0x7ff
#endif
#ifndef RMPC_ALL_VARIANTS
#define RMPC_ALL_VARIANTS RMPC_DEV_VARIANTS
#endif
#ifndef RMPC_TU_MAIN
#define RMPC_TU_MAIN 1
#endif

namespace rmpc {

// solver constants (DESIGN.md, section "Algorithm")
constexpr double kTMin = 1e-2;
// warm start of the multipliers (rmpc_set_warm_start; oracle: ORC_WARM_*): mu = clamp(kappa * previous final mu),
// slacks pushed to kWarmTMin only, multipliers max(previous, mu / t)
constexpr double kWarmKappa = 1000.0;
constexpr double kWarmMuMin = 1e-6;
constexpr double kWarmTMin = 1e-4;
// (the fraction to the boundary is per model: Cfg::TAU)
// barrier restart on stalled steps (oracle: ORC_RS_IT, ORC_RS_N, ORC_RS_ALPHA, ORC_RS_MU, ORC_RS_DECAY)
constexpr int kRsIt = 8, kRsN = 3;
constexpr double kRsAlpha = 0.2, kRsMu = 1e-3, kRsDecay = 0.3;
constexpr int kSweepBlock = 64;     // threads per k_sweep / k_step block: one wavefront, so that small batches spread over all CUs
constexpr int kLsMax = 25;
constexpr int kLsGrow = 1;         // step-length memory: a line search starts this many halvings above the last accepted one
constexpr double kArmijo = 1e-4;
constexpr double kMuDiverged = 1e12;
constexpr double kCurvMu = 1e-2; // curvature terms only once the barrier parameter is this small
constexpr int kLsCurv = 2;        // trials granted to a step computed with constraint curvature
constexpr int kCurvFailMax = 2;   // consecutive curvature-step failures before Gauss-Newton is latched
constexpr int kCurvBackMax = 16;  // (diff-drive) longest run of iterations a failed curvature step switches the terms off
constexpr int kGroupedMin = 512;    // list length from which the grouped Riccati blocks are used
constexpr double kCompFrac = 0.3; // share of tol_comp the convergence test asks for (oracle: ORC_COMP_FRAC)
constexpr double kCsMin = 0.3;    // scaled curvature (oracle: ORC_CS_MIN, ORC_CS_CLEAN)
constexpr int kCsClean = 3;
constexpr double kAccFeas = 1e-6; // acceptable termination: feasibility / complementarity level
constexpr int kDenseDiv = 8;      // identity list while more than B / kDenseDiv instances iterate; below: compacted list,
                                  // and the survivors move to the compact workspace at the host's next look
constexpr int kMigrateMin = 1024; // batches smaller than this never migrate

enum Status : int { ST_ACTIVE = 100 };

enum Part : int { P_F = 0, P_TH, P_LOGS, P_RSTAT, P_REQ, P_RINEQ, P_RCOMP, P_SUMC, P_MINC, P_BAD, P_COUNT };

// Explicit address spaces for what the fused kernel addresses: pointers that travel through structs or are
// selected at run time are otherwise compiled to FLAT accesses, which count on both memory counters and so
// serialise global-memory and LDS waits.
typedef __attribute__((address_space(1))) double gdouble;   // global memory
typedef __attribute__((address_space(3))) double ldouble;   // LDS

// Device workspace (all pointers into one allocation).
struct Ws {
  int N, Bp;
  double *p;                      // [npar][N][Bp]
  double *z[2], *t[2], *lam[2], *nu[2];
  double *dz, *nunew;
  double *grow[2], *Jq[2];        // row values / FK-row gradients at the iterate of the same buffer index
  double *R;                      // [Bp][N][rs] stage records k_sweep -> k_riccati (layout: Cfg::R_*)
  int rs;
  double *gfa;
  double *KP;                     // [Bp][N][kps] per instance and stage: gains K | kff | cost-to-go P (dense) | p --
                                  // private to k_riccati, instance-major so that a wavefront moves a record in one request
  int kps;                        // record stride (doubles, multiple of 8)
  double *part;                   // [P_COUNT][N][Bp]
  double *gphi;                   // [N][Bp]
  unsigned long long *amin_p, *amin_d;  // [Bp] fraction-to-the-boundary step lengths (bits of a positive double)
  // per instance [Bp]
  double *mu, *rho, *phi0, *Dd, *fcur, *thcur, *logcur;
  double *mu_hold;                // barrier restart (inst_decide): the level mu is held at, 0 = none
  double *res_stat, *res_eq, *res_ineq, *res_comp, *obj;
  int *status, *iters, *ls, *cur, *newstep;
  int *redo, *force_gn, *gn_sticky, *curv_fail, *usedc, *stall;
  int *curv_skip, *curv_back;     // (diff-drive) curvature steps still to be skipped / length of the last skip (back-off)
  int *small_steps;               // barrier restart: accepted short steps in a row
  double *theta_mem, *theta_c;    // scaled curvature (Cfg::CSCALE): the scale the next curvature step starts from / of this iteration
  int *theta_clean, *theta_retry; // ... accepted curvature steps in a row without a retry / this iteration has retried
  int *ls0, *lsst;                // halvings the current line search started from / the next one starts from
  int *active_hist;               // [max_passes] instances still iterating after each pass
  int *act_idx, *n_act;           // compacted list of the instances still iterating, its length
  int *orig;                      // [Bp] compact workspace only: column -> instance of the caller's batch
  // multipliers of the last solve (warm start of the next one): [m][N][Bp], [nx][N][Bp], final barrier
  // parameter [Bp]
  double *wlam, *wnu, *wmu;
};

#define IDX(slot, k, b) (((size_t)(slot) * W.N + (size_t)(k)) * W.Bp + (size_t)(b))
// the same with the lane's (stage, instance) offset precomputed (k_sweep / k_step): uniform slot base + 32-bit lane offset
#define IDXL(slot) ((size_t)(slot) * SS + loff)
#define IDXL1(slot) ((size_t)(slot) * SS + loff1)

#if RMPC_TU_MAIN
// ===========================================================================
// pack / unpack: instance-major ABI layout <-> batch-minor SoA (LDS transpose)
// ===========================================================================
// in[b][c], c = k*inner + j  ->  out[(j*N + k)*Bp + b]
__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ in, double *__restrict__ out, int B,
                                              int C, int inner, int N, int Bp) {
  __shared__ double tile[64][65];
  const int b0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  // unconditional requests with clamped indices, all issued before the first LDS store (a branch around a
  // load makes the compiler wait for each element separately)
  double v[16];
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int r = ty + 4 * u;
    const int b = b0 + r < B ? b0 + r : B - 1, c = c0 + tx < C ? c0 + tx : C - 1;
    v[u] = in[(size_t)b * C + c];
  }
#pragma unroll
  for (int u = 0; u < 16; u++) tile[ty + 4 * u][tx] = v[u];
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    int c = c0 + r, b = b0 + tx;
    if (c < C && b < B) {
      int k = c / inner, j = c - k * inner;
      out[((size_t)j * N + k) * Bp + b] = tile[tx][r];
    }
  }
}

#endif  // RMPC_TU_MAIN

// stage-0 state := xinit (mpcModel.py:108 xinitidx), per-instance state reset
__device__ __forceinline__ double warm_mu(double wmu, double mu0) {
  double mu = kWarmKappa * wmu;
  if (mu < kWarmMuMin) mu = kWarmMuMin;
  if (mu > mu0) mu = mu0;
  return mu;
}

#if RMPC_TU_MAIN
__global__ __launch_bounds__(256) void k_init(Ws W, const double *__restrict__ xinit, int B, int nx, double mu0, int warm) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  for (int j = 0; j < nx; j++) W.z[0][IDX(j, 0, b)] = xinit[(size_t)b * nx + j];
  W.status[b] = ST_ACTIVE;
  W.act_idx[b] = b;
  if (b == 0) *W.n_act = B;
  W.iters[b] = 0;
  W.ls[b] = 0;
  W.cur[b] = 0;
  W.newstep[b] = 0;
  W.amin_p[b] = (unsigned long long)__double_as_longlong(1.0);
  W.amin_d[b] = (unsigned long long)__double_as_longlong(1.0);
  W.redo[b] = 0; W.force_gn[b] = 0; W.gn_sticky[b] = 0; W.curv_fail[b] = 0; W.usedc[b] = 0; W.stall[b] = 0;
  W.curv_skip[b] = 0; W.curv_back[b] = 0;
  W.small_steps[b] = 0; W.mu_hold[b] = 0.0;
  W.theta_mem[b] = 1.0; W.theta_c[b] = 1.0; W.theta_clean[b] = 0; W.theta_retry[b] = 0;
  W.ls0[b] = 0; W.lsst[b] = 0;
  W.mu[b] = warm ? warm_mu(W.wmu[b], mu0) : mu0;
  W.rho[b] = 0.0;
  W.phi0[b] = 0.0;
  W.Dd[b] = 0.0;
  W.fcur[b] = 0.0;
  W.thcur[b] = 0.0;
  W.logcur[b] = 0.0;
  W.res_stat[b] = 0.0; W.res_eq[b] = 0.0; W.res_ineq[b] = 0.0; W.res_comp[b] = 0.0; W.obj[b] = 0.0;
}

// z (current buffer of each instance) -> z_out[b][k][v]; stats
__global__ __launch_bounds__(256) void k_unpack(Ws W, double *__restrict__ zout, int *__restrict__ exitflag,
                                                int *__restrict__ iters, double *__restrict__ kkt,
                                                double *__restrict__ obj, int B, int nv,
                                                const int *__restrict__ orig) {
  // orig != nullptr: W is the compact workspace, column b belongs to instance orig[b] of the batch
  __shared__ double tile[64][65];
  const int N = W.N;
  const int C = N * nv;
  const int b0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  {
    const int b = b0 + tx < B ? b0 + tx : B - 1;   // clamped: the requests stay unconditional
    const double *__restrict__ zb = W.z[W.cur[b]];
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int c = c0 + ty + 4 * u < C ? c0 + ty + 4 * u : C - 1;
      const int k = c / nv, j = c - k * nv;
      v[u] = zb[IDX(j, k, b)];
    }
#pragma unroll
    for (int u = 0; u < 16; u++) tile[ty + 4 * u][tx] = v[u];
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    int b = b0 + r, c = c0 + tx;
    if (b < B && c < C) zout[(size_t)(orig ? orig[b] : b) * C + c] = tile[tx][r];
  }
  if (blockIdx.y == 0 && threadIdx.x < 64) {
    int b = b0 + threadIdx.x;
    if (b < B) {
      const int ob = orig ? orig[b] : b;
      int st = W.status[b];
      exitflag[ob] = (st == ST_ACTIVE) ? 0 : st;
      iters[ob] = W.iters[b];
      double r = fmax(fmax(W.res_stat[b], W.res_eq[b]), fmax(W.res_ineq[b], W.res_comp[b]));
      kkt[ob] = r;
      obj[ob] = W.obj[b];
    }
  }
}

// Multipliers of the finished solve -> the warm-start arrays of the batch's workspace D (W may be the compact
// workspace: column b then belongs to instance orig[b]).  One lane per (column, stage).
// d[j * ds] = s[j * ss], j < cnt, eight requests in flight (source and destination never alias: the copies below are
// chains of dependent latencies otherwise -- 50 us for the arm's multipliers, 105 us for a migration of 128 instances)
__device__ __forceinline__ void copy_strided(double *__restrict__ d, const double *__restrict__ s, const int cnt, const size_t ds,
                                             const size_t ss) {
  int j = 0;
  for (; j + 8 <= cnt; j += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = s[(size_t)(j + u) * ss];
#pragma unroll
    for (int u = 0; u < 8; u++) d[(size_t)(j + u) * ds] = v[u];
  }
  for (; j < cnt; j++) d[(size_t)j * ds] = s[(size_t)j * ss];
}
__device__ __forceinline__ void fill_strided(double *__restrict__ d, const double val, const int cnt, const size_t ds) {
  for (int j = 0; j < cnt; j++) d[(size_t)j * ds] = val;
}

__global__ __launch_bounds__(256) void k_save_duals(const Ws W, const Ws D, int B, int m, int nx, const int *__restrict__ orig,
                                                    double mu0) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int b = gid % W.Bp, k = gid / W.Bp;
  if (b >= B || k >= W.N) return;
  const int cur = W.cur[b];
  const int ob = orig ? orig[b] : b;
  // a failed solve leaves nothing to start from: zero multipliers and mu0 (the warm start then degenerates to
  // lambda = mu0 / t, nu = 0)
  const int st = W.status[b];
  const double mu = W.mu[b];
  const bool ok = (st == ST_ACTIVE || st >= 0) && isfinite(mu) && mu > 0.0;
  double *const dl = D.wlam + (size_t)k * D.Bp + ob, *const dn = D.wnu + (size_t)k * D.Bp + ob;
  const size_t ds = (size_t)D.N * D.Bp, ss = (size_t)W.N * W.Bp;
  if (ok) {
    copy_strided(dl, W.lam[cur] + (size_t)k * W.Bp + b, m, ds, ss);
    copy_strided(dn, W.nu[cur] + (size_t)k * W.Bp + b, nx, ds, ss);
  } else {
    fill_strided(dl, 0.0, m, ds);
    fill_strided(dn, 0.0, nx, ds);
  }
  if (k == 0) D.wmu[ob] = ok ? mu : mu0;
}

// ===========================================================================
// k_compact: ordered list of the instances that are still iterating.  All pass
// kernels index their lanes through it, so wavefronts beyond the list exit at
// once and the passes of the iteration tail touch a few wavefronts only.
// ===========================================================================
__global__ __launch_bounds__(1024) void k_compact(Ws W, int B, int pass) {
  __shared__ int sums[1024];
  const int tid = threadIdx.x;
  // (passes enqueued without a host look, rmpc_set_pass_budget: once nothing iterates any more the remaining passes
  //  are empty launches -- this one too; active_hist was zeroed before the solve)
  if (pass > 0 && *W.n_act == 0) return;
  const int per = (B + 1023) / 1024;
  const int lo = tid * per, hi = (lo + per < B) ? lo + per : B;
  int cnt = 0;
  for (int b = lo; b < hi; b++) cnt += (W.status[b] == ST_ACTIVE);
  sums[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (tid >= off) ? sums[tid - off] : 0;
    __syncthreads();
    sums[tid] += v;
    __syncthreads();
  }
  const int total = sums[1023];
  // while most instances are still iterating the identity list keeps every access coalesced
  const bool dense = (total * kDenseDiv > B);
  int base = dense ? lo : sums[tid] - cnt;
  for (int b = lo; b < hi; b++)
    if (dense || W.status[b] == ST_ACTIVE) W.act_idx[base++] = b;
  if (tid == 1023) {
    *W.n_act = dense ? B : total;
    W.active_hist[pass] = total;
  }
}

// The same list from ONE wavefront (batches up to kCompactWaveMax instances).  With other handles' kernels on the chip
// every SIMD holds a long-lived 512-register wavefront, and the 16-wavefront block above waits until a whole compute
// unit has drained: in a trace of four arm batches in flight k_compact took 25 us on average (p90 93 us) for 5 us of
// work -- once per pass, on the critical path of its stream.  A single wavefront takes the first SIMD that frees.
// 64 instances per round (one coalesced request, ballot + popcount instead of a scan), eight rounds in flight.
constexpr int kCompactWaveMax = 8192;
__global__ __launch_bounds__(64) void k_compact_wave(Ws W, int B, int pass) {
  const int lane = threadIdx.x;
  if (pass > 0 && *W.n_act == 0) return;
  const int rounds = (B + 63) / 64;
  int total = 0;
  for (int r0 = 0; r0 < rounds; r0 += 8) {
    int st[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = (r0 + u) * 64 + lane;
      st[u] = W.status[b < B ? b : B - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = (r0 + u) * 64 + lane;
      total += __popcll(__ballot(b < B && st[u] == ST_ACTIVE));
    }
  }
  const bool dense = (total * kDenseDiv > B);
  const unsigned long long below = (1ull << lane) - 1ull;
  int base = 0;
  for (int r0 = 0; r0 < rounds; r0 += 8) {
    int st[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = (r0 + u) * 64 + lane;
      st[u] = W.status[b < B ? b : B - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = (r0 + u) * 64 + lane;
      const bool on = b < B && (dense || st[u] == ST_ACTIVE);
      const unsigned long long mk = __ballot(on);
      if (on) W.act_idx[base + __popcll(mk & below)] = b;
      base += __popcll(mk);
    }
  }
  if (lane == 0) {
    *W.n_act = dense ? B : total;
    W.active_hist[pass] = total;
  }
}

// ===========================================================================
// k_migrate: once few instances are left their whole iteration state moves to the
// dense columns 0..n-1 of a small second workspace.  Indexing scattered survivors
// through the list costs a 64-byte sector per 8-byte element (every pass then moves
// as many bytes as a full batch); one gather of that kind pays for itself in the
// next pass.  Runs between k_step and the next k_sweep: what crosses that boundary
// is the current iterate, the step, the parameters and the per-instance words.
// ===========================================================================
__global__ __launch_bounds__(64) void k_migrate(const Ws S, const Ws D, int n, int nv, int m, int nx, int npar, int nh,
                                                int njq) {
  const int li = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (li >= n) return;
  const int b = S.act_idx[li];
  const int cur = S.cur[b];
  auto si = [&](int slot) { return ((size_t)slot * S.N + k) * S.Bp + b; };
  auto di = [&](int slot) { return ((size_t)slot * D.N + k) * D.Bp + li; };
  {
    const size_t ds = (size_t)D.N * D.Bp, ss = (size_t)S.N * S.Bp, d0 = di(0), s0 = si(0);
    copy_strided(D.z[0] + d0, S.z[cur] + s0, nv, ds, ss);
    copy_strided(D.dz + d0, S.dz + s0, nv, ds, ss);
    copy_strided(D.t[0] + d0, S.t[cur] + s0, m, ds, ss);
    copy_strided(D.lam[0] + d0, S.lam[cur] + s0, m, ds, ss);
    copy_strided(D.grow[0] + d0, S.grow[cur] + s0, nh, ds, ss);
    copy_strided(D.Jq[0] + d0, S.Jq[cur] + s0, njq, ds, ss);
    copy_strided(D.nu[0] + d0, S.nu[cur] + s0, nx, ds, ss);
    copy_strided(D.nunew + d0, S.nunew + s0, nx, ds, ss);
    copy_strided(D.p + d0, S.p + s0, npar, ds, ss);
  }
  D.gphi[di(0)] = S.gphi[si(0)];
  if (k == 0) {
    D.amin_p[li] = S.amin_p[b]; D.amin_d[li] = S.amin_d[b];
    D.mu[li] = S.mu[b]; D.rho[li] = S.rho[b]; D.phi0[li] = S.phi0[b]; D.Dd[li] = S.Dd[b];
    D.fcur[li] = S.fcur[b]; D.thcur[li] = S.thcur[b]; D.logcur[li] = S.logcur[b];
    D.res_stat[li] = S.res_stat[b]; D.res_eq[li] = S.res_eq[b]; D.res_ineq[li] = S.res_ineq[b];
    D.res_comp[li] = S.res_comp[b]; D.obj[li] = S.obj[b];
    D.status[li] = S.status[b]; D.iters[li] = S.iters[b]; D.ls[li] = S.ls[b]; D.newstep[li] = S.newstep[b];
    D.redo[li] = S.redo[b]; D.force_gn[li] = S.force_gn[b]; D.gn_sticky[li] = S.gn_sticky[b];
    D.curv_fail[li] = S.curv_fail[b]; D.usedc[li] = S.usedc[b]; D.stall[li] = S.stall[b];
    D.curv_skip[li] = S.curv_skip[b]; D.curv_back[li] = S.curv_back[b];
    D.small_steps[li] = S.small_steps[b]; D.mu_hold[li] = S.mu_hold[b];
    D.theta_mem[li] = S.theta_mem[b]; D.theta_c[li] = S.theta_c[b]; D.theta_clean[li] = S.theta_clean[b]; D.theta_retry[li] = S.theta_retry[b];
    D.ls0[li] = S.ls0[b]; D.lsst[li] = S.lsst[b];
    D.cur[li] = 0;
    D.orig[li] = b;
    D.act_idx[li] = li;
    if (li == 0) *D.n_act = n;
  }
}

#endif  // RMPC_TU_MAIN

// 1/x for normal positive x: hardware estimate + two Newton steps (about 1 ulp; a full fp64 division costs three
// times as many instructions and the sweep performs one or two per constraint row)
__device__ __forceinline__ double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// compile-time loop: fn(integral_constant<int, L>) ... fn(integral_constant<int, H-1>)
template <int L, class F, int... I>
__device__ __forceinline__ void for_range_impl(F &&fn, std::integer_sequence<int, I...>) {
  (fn(std::integral_constant<int, L + I>{}), ...);
}
template <int L, int H, class F>
__device__ __forceinline__ void for_range(F &&fn) {
  for_range_impl<L>(fn, std::make_integer_sequence<int, (H > L ? H - L : 0)>{});
}

// ===========================================================================
// k_sweep: stage-parallel function / Jacobian evaluation + condensing
// ===========================================================================
// Rows are processed in two groups so that every register array is indexed by an
// unrolled loop counter only and loads can be issued in batches:
//   * FK rows (distance / plane rows), grouped by kinematic slot (static slot loop,
//     short runtime loop over the rows of the slot);
//   * single-variable rows (limits and simple bounds), grouped by variable (static
//     loops; absent entries load row 0 and are masked -- a branch around a load,
//     even a wave-uniform one, makes hipcc wait for every element separately).
// What one lane -- one (instance, stage) pair -- of the stage-parallel sweep addresses.  Element `slot` of an
// array is ptr[slot * SS + loff]: the batch-minor SoA of the pass kernels (SS = N * Bp, loff = k * Bp + b,
// next stage kstride = Bp) and the per-instance layout of the fused kernel ([instance][slot][32 stages]:
// SS = 32, loff = k, kstride = 1, pointers advanced to the instance) run the same code.
template <class RP = gdouble, class SP = RP>   // RP / SP: where the stage record / the step live (gdouble, or ldouble in the fused kernels)
struct SweepIO {
  const gdouble *zc, *tc, *lc, *nc, *pp, *gro, *jqo;   // iterate (current buffer), parameters
  gdouble *zn, *tn, *ln, *nn, *grn, *jqn, *gfa;        // trial point (other buffer), cost gradient
  const SP *dzp, *nup;                                 // step
  RP *rec;                                             // this lane's stage record
  size_t SS;
  unsigned loff, kstride;
  // the step (dzp, nup) may live elsewhere (fused kernel: in the LDS slots of the instance): own strides
  size_t SSd;
  unsigned loffd, kstrided;
  // first pass of a warm-started solve: multipliers / costates of the previous solve (same addressing as lc / nc;
  // stage k takes the values of stage k + 1, like the shifted plan)
  const gdouble *wl, *wn;
  int warm;
};
// merit / KKT partial sums of one stage (order = enum Part)
struct Partials {
  double f, th, logs, rstat, req, rineq, rcomp, sumc, minc, bad;
#ifdef RMPC_STAMPS
  long long tk[6];   // development builds: cycles of the sections of the sweep ([4], [5]: step lengths and their reduction, fused_sweep_step_call)
#endif
};
#ifdef RMPC_STAMPS
// development aid: cycles per section of k_sweep (pass kernels), summed over the wavefronts of all launches
__device__ long long g_sst[8];
#endif
#ifdef RMPC_STAMPS
#define SW_STAMP(i) do { long long t_ = __builtin_amdgcn_s_memtime(); out.tk[i] = t_ - sw_t0; sw_t0 = t_; } while (0)
#else
#define SW_STAMP(i)
#endif

// Order in which sweep_body takes the variables of a stage (positions 0 .. NV-1; the first NFIRST of them before
// the kinematics: see EARLY in sweep_body).
template <class C, bool EARLY>
struct SweepOrder {
  static constexpr int NFIRST = EARLY ? (C::NV - C::NQ - (C::NS > 0 ? 1 : 0)) : 0;
  __host__ __device__ static constexpr int at(int p) {
    int idx[C::NV] = {};
    int n = 0;
    if (EARLY) {
      for (int j = C::NQ; j < C::NV; j++)
        if (!(C::NS > 0 && j == C::NX)) idx[n++] = j;
      for (int j = 0; j < C::NQ; j++) idx[n++] = j;
      if (C::NS > 0) idx[n++] = C::NX;
    } else {
      for (int j = 0; j < C::NV; j++) idx[n++] = j;
    }
    return idx[p];
  }
};

// The scalars of the model the sweep needs (everything else comes through the view)
struct SweepK { int N; double dt; int use_curv; };

// FIRSTC: 1 / 0 = the first pass of a solve (or not) known at compile time, -1 = taken from first_rt.  The rows
// branch on it; callers that can afford two copies of the body (every kernel here) pass it as a constant so that
// the rows of a stage form one basic block and their requests are issued together.
template <class C, int EARLY_MODE = -1, class RP = gdouble, class V = RtView, int FIRSTC = -1>
__device__ __forceinline__ void sweep_body(const SweepK M, const V &v, const SweepIO<RP> &io, const int k,
                                           const bool first_rt, const bool nostep, const double alpha, const double adual,
                                           const double mu, Partials &out, ldouble *const qacc = nullptr) {
  const bool first = FIRSTC < 0 ? first_rt : (FIRSTC != 0);
  // (FKCURV, k_sweep) the two 7 x 7 blocks of the q variables are accumulated in LDS, one column of 2 x 28 doubles per
  // lane (qacc, lane stride kSweepBlock): they are touched once per FK point and by the joint-limit rows only, and the
  // kernel has no register to spare for them
  constexpr bool QLDS = C::FKCURV;
  auto qtri = [](int a, int c) __attribute__((always_inline)) { return a * C::NQ - a * (a - 1) / 2 + (c - a); };
#ifdef RMPC_STAMPS
  long long sw_t0 = __builtin_amdgcn_s_memtime();
#endif
  constexpr int NQ = C::NQ, NX = C::NX, NS = C::NS, NU = C::NU, NV = C::NV;
  const int N = M.N;
  const unsigned loff = io.loff;
  const size_t SS = io.SS;
  const gdouble *__restrict__ zc = io.zc;
  const gdouble *__restrict__ tc = io.tc;
  const gdouble *__restrict__ lc = io.lc;
  const gdouble *__restrict__ nc = io.nc;
  gdouble *__restrict__ zn = io.zn;
  gdouble *__restrict__ tn = io.tn;
  gdouble *__restrict__ ln = io.ln;
  gdouble *__restrict__ nn = io.nn;
  const gdouble *__restrict__ pp = io.pp;
  const RP *__restrict__ dzp = io.dzp;
  const gdouble *__restrict__ gro = io.gro;   // row values and FK-row gradients at the current iterate:
  const gdouble *__restrict__ jqo = io.jqo;   //  the slack / multiplier steps are recomputed from them
  gdouble *__restrict__ grn = io.grn;
  gdouble *__restrict__ jqn = io.jqn;
  const RP *__restrict__ nup = io.nup;
  gdouble *__restrict__ gfa = io.gfa;
  RP *__restrict__ rec = (RP *)__builtin_assume_aligned(io.rec, 64);   // 64-byte aligned: neighbouring entries leave as 16-byte stores

  // ---- trial stage vector, costates, next stage's state ------------------------
  double z[NV], xk1[NX], nuk[NX], nun[NX];
  double zo[NV], dzo[NV];   // current iterate and step of this stage (the row steps are recomputed from them)
  const unsigned loff1 = loff + (k < N - 1 ? io.kstride : 0u);  // next stage, clamped: loads stay unconditional
  const bool warm = first && (io.warm != 0);
  // multipliers the rows start from: the current buffer, or (warm first pass) the previous solve's, one stage on
  const gdouble *__restrict__ lsrc = warm ? io.wl : lc;
  const unsigned loffl = warm ? loff1 : loff;
#define IDXLL(slot) ((size_t)(slot) * SS + loffl)
  double x1[NX], dx1[NX], n0[NX], n0n[NX], n1[NX], n1n[NX];
  {
    const size_t SSd = io.SSd;
    const unsigned loffd = io.loffd, loffd1 = io.loffd + (k < N - 1 ? io.kstrided : 0u);
#pragma unroll
    for (int j = 0; j < NV; j++) { zo[j] = zc[IDXL(j)]; dzo[j] = dzp[(size_t)j * SSd + loffd]; }
#pragma unroll
    for (int j = 0; j < NX; j++) {
      x1[j] = zc[IDXL1(j)]; dx1[j] = dzp[(size_t)j * SSd + loffd1];
      n0[j] = nc[IDXL(j)];  n0n[j] = nup[(size_t)j * SSd + loffd];
      n1[j] = nc[IDXL1(j)]; n1n[j] = nup[(size_t)j * SSd + loffd1];
    }
  }
  auto P = [&](int off) __attribute__((always_inline)) -> double { return pp[IDXL(off)]; };
  // Request batching (generated views: PIPE).  One wavefront per SIMD hides no latency by itself, so the body issues
  // what it will need well before it needs it: the objective parameters and every distance row's inputs here, the
  // single-variable rows two variables ahead of the arithmetic (var_load / var_compute below).  With the runtime
  // tables the requests stay where the arithmetic is, as before.
  // (the arms too, over the runtime tables: their sweep waits on memory for 63 % of its cycles -- 114 -> 110 us)
  constexpr bool PIPE = V::SPEC || C::FKCURV || std::is_same<V, GView>::value;
  double wuv[NU], wsv = 0.0, goalv[3] = {0, 0, 0}, wgoalv[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < NU; j++) wuv[j] = P(v.off_wu() + j);
  if constexpr (NS > 0) wsv = P(v.off_ws());
  const double rbody = (v.off_r_body() >= 0) ? P(v.off_r_body()) : 0.0;
  if (v.has_goal()) {
#pragma unroll
    for (int c = 0; c < 3; c++) { goalv[c] = P(v.off_goal() + c); wgoalv[c] = P(v.off_wgoal() + c); }
  }
  // inputs of a distance row: slack, multiplier, value and gradient at the current iterate, obstacle, weight
  struct FkBuf { double tcv, lcv, gold, jo[NQ], op[4], wi; };
  auto fk_load = [&](const int r, FkBuf &Bf) __attribute__((always_inline)) {
    const int i = v.fk_row(r), kind = v.fk_kind(r), ob = v.fk_obst(r), fi = v.fk_idx(r);
    Bf.tcv = tc[IDXL(i)]; Bf.lcv = lsrc[IDXLL(i)]; Bf.gold = gro[IDXL(i)];
#pragma unroll
    for (int a = 0; a < NQ; a++) Bf.jo[a] = jqo[IDXL(fi * NQ + a)];
#pragma unroll
    for (int c = 0; c < 4; c++) Bf.op[c] = 0.0;
    if (kind == ROW_RADIAL) {
#pragma unroll
      for (int c = 0; c < 4; c++) Bf.op[c] = P(v.off_obst() + 4 * ob + c);
    } else if (kind == ROW_LINEAR) {
#pragma unroll
      for (int c = 0; c < 4; c++) Bf.op[c] = P(v.off_lin() + 4 * ob + c);
    }
    Bf.wi = 0.0;
    if (v.has_avoid() && v.fk_first(r)) Bf.wi = P(v.off_wconstr() + v.fk_mod(r));
  };
  constexpr int NFKC = []() { if constexpr (V::SPEC) return V::nfkrows() > 0 ? V::nfkrows() : 1; else return 1; }();
  FkBuf fkb[NFKC];
  if constexpr (V::SPEC) {   // (the rows of a generated view are static: their inputs are requested here, all at once)
    for_range<0, NFKC>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      if constexpr (r < V::nfkrows()) fk_load(r, fkb[r]);
    });
  }

  struct VarBuf { double tcv[kVarRows], lcv[kVarRows], lim[kVarRows], wi[kVarRows]; };
  auto var_load = [&](auto jc, VarBuf &Bv) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    // unconditional, clamped requests for the (up to) four rows of variable j
#pragma unroll
    for (int u = 0; u < kVarRows; u++) {
      const int i = v.v_row(j, u);
      const int ii = i >= 0 ? i : 0;
      const int po = v.v_poff(j, u);
      Bv.tcv[u] = tc[IDXL(ii)];
      Bv.lcv[u] = lsrc[IDXLL(ii)];
      const double pl = pp[IDXL(po >= 0 ? po : 0)];
      Bv.lim[u] = po >= 0 ? pl : v.v_val(j, u);
      Bv.wi[u] = 0.0;
      if (i >= 0 && v.has_avoid() && v.v_first(j, u)) Bv.wi[u] = P(v.off_wconstr() + v.v_mod(j, u));
    }
  };
  const double al = alpha, adl = adual;
  // ---- trial point -----------------------------------------------------------------------------------
  {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      z[j] = nostep ? zo[j] : zo[j] + al * dzo[j];
      zn[IDXL(j)] = z[j];
    }
#pragma unroll
    for (int j = 0; j < NX; j++) {
      xk1[j] = nostep ? x1[j] : x1[j] + al * dx1[j];
      if constexpr (QLDS) qacc[(2 * C::NQ2 + j) * kSweepBlock] = xk1[j];   // (read back for the defect, at the end)
      double v = 0.0, w = 0.0;
      if (!first && k >= 1) v = nostep ? n0[j] : n0[j] + al * (n0n[j] - n0[j]);
      if (!first && k < N - 1) w = nostep ? n1[j] : n1[j] + al * (n1n[j] - n1[j]);
      if (warm) {
        // costates of the previous solve, shifted: nu_k <- nu_{k+1}, nu_{k+1} <- nu_{k+2} (last stage repeated)
        const unsigned loff2 = loff1 + (k < N - 2 ? io.kstride : 0u);
        if (k >= 1) v = io.wn[IDXL1(j)];
        if (k < N - 1) w = io.wn[(size_t)j * SS + loff2];
      }
      nuk[j] = v;
      if constexpr (QLDS) { if (j < NQ) qacc[(2 * C::NQ2 + NX + (j < NQ ? j : 0)) * kSweepBlock] = v; }
      nun[j] = w;
      nn[IDXL(j)] = v;
    }
  }
  // ---- accumulators --------------------------------------------------------
  double gf[NV], q0[NV], q1[NV], rs[NV], Dg[NV], cs[NV];
  double Qqq[NQ][NQ];
  constexpr bool QC = C::CURV || C::DDCURV;   // the record carries a curvature block of the q variables
  double Cqq[QC ? NQ : 1][QC ? NQ : 1];  // sum_i (lambda_i + cN/h^2) grad^2 h_i of the distance rows
#pragma unroll
  for (int a = 0; a < (QC ? NQ : 1); a++)
#pragma unroll
    for (int c = 0; c < (QC ? NQ : 1); c++) Cqq[a][c] = 0;
#pragma unroll
  for (int j = 0; j < NV; j++) { gf[j] = 0; q0[j] = 0; q1[j] = 0; rs[j] = 0; Dg[j] = 0; cs[j] = 0; }
#pragma unroll
  for (int a = 0; a < NQ; a++)
#pragma unroll
    for (int c = 0; c < NQ; c++) Qqq[a][c] = 0;
  if constexpr (QLDS) {
#pragma unroll
    for (int s2 = 0; s2 < 2 * C::NQ2; s2++) qacc[s2 * kSweepBlock] = 0.0;
  }
  double f = 0.0;
  int bad = 0;
  double theta = 0.0, rineq = 0.0, rcomp = 0.0, sumc = 0.0, minc = 1e300;
  // sum of log t over the rows, kept as log(prod of mantissas) + ln2 * (sum of exponents): one log per lane
  // instead of one per row (a software log is ~70 instructions; the rows of a stage are the bulk of this kernel)
  double lprod = 1.0;
  int lexp = 0;

  SW_STAMP(0);
  // ---- control effort and slack penalty (ObjectiveManager.py:28-42) ----------
#pragma unroll
  for (int j = 0; j < NU; j++) {
    const double wu = wuv[j], u = z[NX + NS + j];
    f += wu * u * u;
    gf[NX + NS + j] += 2.0 * wu * u;
    Dg[NX + NS + j] += 2.0 * wu;
  }
  double sl = 0.0;
  if constexpr (NS > 0) {
    const double ws = wsv;
    sl = z[NX];
    f += ws * sl * sl;
    gf[NX] += 2.0 * ws * sl;
    Dg[NX] += 2.0 * ws;
  }
  // trial slack / multiplier of row i and their bookkeeping; returns sigma, ca, cb, lv
  struct RowW { double sig, ca, cb, lv; };
  // (gold, gdz: row value at the current iterate and J_i dz -- the slack / multiplier steps of the row are
  //  recomputed with the very expressions k_step took its step lengths from)
  auto row_core = [&](int i, double g, double tcv, double lcv, double gold, double gdz) __attribute__((always_inline)) -> RowW {
    double tv, lv;
    if (first) {
      const double tmin = warm ? kWarmTMin : kTMin;
      tv = g > tmin ? g : tmin;
      lv = mu * frcp(tv);
      if (warm) lv = lcv > lv ? lcv : lv;   // (lcv: the previous solve's multiplier of this row, one stage on)
    } else {
      const double dtv = gdz + (gold - tcv);
      const double dlv = (mu - tcv * lcv - lcv * dtv) * frcp(tcv);
      // (null passes keep the point by selection, not by a zero step length: the step they would multiply
      //  may be stale -- after a failed factorisation of the fused kernel even non-finite)
      tv = nostep ? tcv : tcv + al * dtv;
      lv = nostep ? lcv : lcv + adl * dlv;
    }
    tn[IDXL(i)] = tv;
    ln[IDXL(i)] = lv;
    const double rg = g - tv;
    theta += fabs(rg);
    bad |= (int)!(tv > 0.0);   // (cannot happen: fraction to the boundary; keeps the product's sign meaningful.  |=: no branch)
    {
      int ex;
      lprod *= frexp(tv, &ex);
      lexp += ex;
    }
    rineq = fmax(rineq, fabs(rg));
    const double cmp = tv * lv;
    rcomp = fmax(rcomp, cmp);
    sumc += cmp;
    minc = fmin(minc, cmp);
    const double it = frcp(tv);
    return {lv * it, lv * rg * it, it, lv};
  };

  // ---- single-variable rows: limits (general rows) and simple bounds, by variable ----
  // (generic lambda over a compile-time variable index: every array index stays a constant)
  auto var_compute = [&](auto jc, const VarBuf &Bv) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
#pragma unroll
    for (int u = 0; u < kVarRows; u++) {
      const int i = v.v_row(j, u);
      if (i < 0) continue;  // uniform
      const double sg = (double)v.v_sgn(j, u);
      const bool soft = (NS > 0) && v.v_soft(j, u);
      const bool neutral = (k == 0) && (j < NX) && !soft;  // constant of the problem at the pinned stage
      const double h = neutral ? 1.0 : sg * (z[j] - Bv.lim[u]);
      if (v.has_avoid() && v.v_first(j, u)) {
        // (selects, not a branch on the weight: a data-dependent branch would cut the stage's rows into
        //  separate basic blocks and with them the batches of requests)
        const double wi = Bv.wi[u];
        const bool on = (wi != 0.0) && !(k == 0 && j < NX);
        const double cN = (double)M.N * wi;
        bad |= (int)(on & !(h > 0.0));   // (bitwise: a short-circuit branch would cut the rows into separate basic blocks)
        const double ih = frcp(h);
        f += on ? cN * ih : 0.0;
        gf[j] += on ? -cN * (ih * ih) * sg : 0.0;
        const double c2 = on ? 2.0 * cN * (ih * ih * ih) : 0.0;
        if (j < NQ) {
          if constexpr (QLDS) qacc[qtri(j < NQ ? j : 0, j < NQ ? j : 0) * kSweepBlock] += c2;
          else Qqq[j < NQ ? j : 0][j < NQ ? j : 0] += c2;
        } else Dg[j] += c2;
      }
      double g = h;
      if constexpr (NS > 0) { if (soft) g += sl; }
      if (v.v_poff(j, u) >= 0) grn[IDXL(i)] = g;  // general rows keep their value for k_step
      // the same row at the current iterate (what k_step read back or recomputed)
      double gold = neutral ? 1.0 : sg * (zo[j] - Bv.lim[u]);
      double gdz = sg * dzo[j];
      if constexpr (NS > 0) { if (soft) { gold += zo[NX]; gdz += dzo[NX]; } }
      const RowW rw = row_core(i, g, Bv.tcv[u], Bv.lcv[u], gold, gdz);
      // (a neutralised row contributes nothing; by selection, not by a branch: in the fused kernel the stage differs
      //  from lane to lane and a divergent `continue` cuts the rows of a variable into exec-masked blocks)
      q0[j] = neutral ? q0[j] : q0[j] + sg * rw.ca;
      q1[j] = neutral ? q1[j] : q1[j] + sg * rw.cb;
      rs[j] = neutral ? rs[j] : rs[j] - sg * rw.lv;
      const double sigc = neutral ? 0.0 : rw.sig;
      if (j < NQ) {
        if constexpr (QLDS) qacc[qtri(j < NQ ? j : 0, j < NQ ? j : 0) * kSweepBlock] += sigc;
        else Qqq[j < NQ ? j : 0][j < NQ ? j : 0] = neutral ? Qqq[j < NQ ? j : 0][j < NQ ? j : 0] : Qqq[j < NQ ? j : 0][j < NQ ? j : 0] + rw.sig;
      } else Dg[j] = neutral ? Dg[j] : Dg[j] + rw.sig;
      if constexpr (NS > 0) {
        if (soft) {
          cs[j] += rw.sig * sg;
          q0[NX] += rw.ca;
          q1[NX] += rw.cb;
          rs[NX] -= rw.lv;
          Dg[NX] += rw.sig;
        }
      }
    }
  };
  // Everything variable j contributes to is complete: stationarity residual of the variable and its entries
  // of the stage record.  (Holonomic chain: A^T nu = [nu_q ; dt nu_q + nu_v], B^T nu = dt^2/2 nu_q + dt nu_v;
  // the diff-drive model needs its Jacobians first and is finalised in one go further down.)
  double rstat = 0.0;
  auto finalize_var = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    double r = rs[j] + gf[j];
    if constexpr (C::ROBOT == RMPC_ROBOT_CHAIN) {
      if (k < N - 1) {
        const double hh = M.dt, hh2 = 0.5 * M.dt * M.dt;
        if constexpr (j < NQ) r += nun[j];
        else if constexpr (j < NX) r += hh * nun[j - NQ] + nun[j];
        else if constexpr (j >= NX + NS) r += hh2 * nun[j - NX - NS] + hh * nun[NQ + (j - NX - NS)];
      }
    }
    if (!(j < NX && k == 0)) {   // x_1 is fixed: no stationarity condition
      if constexpr (j < NX) {
        if constexpr (QLDS && j < NQ) r -= qacc[(2 * C::NQ2 + NX + j) * kSweepBlock];
        else r -= nuk[j];
      }
      rstat = fmax(rstat, fabs(r));
    }
    if constexpr (j >= NQ) rec[C::R_DG + j - NQ] = Dg[j];
    if constexpr (NS > 0) rec[C::R_CS + j] = cs[j];
    rec[C::R_Q0 + j] = gf[j] + q0[j];
    rec[C::R_Q1 + j] = q1[j];
    gfa[IDXL(j)] = gf[j];
  };
  constexpr bool CHAIN = (C::ROBOT == RMPC_ROBOT_CHAIN);
  // The arm: velocity and input variables first, so that their accumulators are dead before the kinematics
  // start (the slack variable collects from every softened row and waits for the end): 1.2 KB less scratch
  // per lane, sweep 190 -> 139 us on cfg4.  The three-joint models do not spill and lose 7 % this way.
  constexpr bool EARLY = (EARLY_MODE >= 0) ? (CHAIN && EARLY_MODE != 0) : (CHAIN && (NQ > 3));
  using Ord = SweepOrder<C, EARLY>;
  // positions [P0, P1) of the order; PIPE: the requests of a variable are issued two variables ahead (the first two
  // of the range by the caller when PRE is set)
  // (PD: how many variables ahead.  Two: the boxer over the runtime tables at three and four -- a round trip to the
  //  instance's block is 2 - 3 us with the chip full, the rows of a variable 0.5 us -- spills 428 / 556 B per lane instead
  //  of 296 and loses 4 - 7 %: 0.68 -> 0.64 M solves/s with four batches in flight)
#ifdef RMPC_PIPE_DEPTH
  constexpr int PD = RMPC_PIPE_DEPTH;
#else
  constexpr int PD = 2;
#endif
  VarBuf vring[PIPE ? PD + 1 : 1];
  auto run_vars = [&](auto p0c, auto p1c, auto finc, auto prec) __attribute__((always_inline)) {
    constexpr int P0 = decltype(p0c)::value, P1 = decltype(p1c)::value;
    constexpr bool FIN = decltype(finc)::value, PRE = decltype(prec)::value;
    if constexpr (P1 > P0) {
      if constexpr (PIPE && !PRE) {
        for_range<0, PD>([&](auto dc) __attribute__((always_inline)) {
          constexpr int d = decltype(dc)::value;
          if constexpr (P0 + d < P1) var_load(std::integral_constant<int, Ord::at(P0 + d < P1 ? P0 + d : P0)>{}, vring[d]);
        });
      }
      for_range<P0, P1>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr int j = Ord::at(p);
        if constexpr (PIPE) {
          if constexpr (p + PD < P1) var_load(std::integral_constant<int, Ord::at(p + PD < P1 ? p + PD : p)>{}, vring[(p + PD - P0) % (PD + 1)]);
          __builtin_amdgcn_sched_barrier(0);
          var_compute(std::integral_constant<int, j>{}, vring[(p - P0) % (PD + 1)]);
        } else {
          var_load(std::integral_constant<int, j>{}, vring[0]);
          var_compute(std::integral_constant<int, j>{}, vring[0]);
        }
        if constexpr (FIN) finalize_var(std::integral_constant<int, j>{});
      });
    }
  };
  using TrueT = std::integral_constant<bool, true>;
  using FalseT = std::integral_constant<bool, false>;
  run_vars(std::integral_constant<int, 0>{}, std::integral_constant<int, Ord::NFIRST>{}, TrueT{}, FalseT{});
  // (no variable goes first: the requests of the first PD variables leave before the kinematics)
  constexpr bool PRE2 = PIPE && (Ord::NFIRST == 0);
  if constexpr (PRE2) {
    for_range<0, PD>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      if constexpr (d < NV) var_load(std::integral_constant<int, Ord::at(d < NV ? d : 0)>{}, vring[d]);
    });
  }

  // ---- kinematics, GoalReaching and the FK rows, slot by slot -------------------
  Kin<C> kin;
  {
    double q[NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) q[j] = z[j];
    kin.compute(v, q);
  }
  auto do_slot = [&](auto slc) __attribute__((always_inline)) {
    constexpr int SL = decltype(slc)::value;
    if constexpr (V::SPEC) {
      if constexpr (SL >= V::nslots()) return;
    }
    if (SL >= v.nslots()) return;
    Vec3 J[NQ];
    const Vec3 Pt = kin.template point<SL>(v, J);
    // (DDCURV) the frames ride on the base, p = (x, y) + R(theta) o: d2 p / dtheta2 = -(p - (x, y)); a pair: -(pa - pb)
    Vec3 ddP = {0, 0, 0};
    if constexpr (C::DDCURV) {
      if (v.slot_fb(SL) >= 0) ddP = {-Pt.x, -Pt.y, 0.0};
      else ddP = {-(kin.pa[SL].x - kin.qx), -(kin.pa[SL].y - kin.qy), 0.0};
    }
    // (FKCURV) sum over the slot's rows of (multiplier + inverse-barrier weight) x unit direction of the row, minus
    // the goal cost's 2 w e: what the second derivatives of the slot's point are contracted with
    Vec3 Fc = {0, 0, 0};
    // (FKCURV) every term a row of the slot adds to the q block has the form J^T (w n n^T) J with the row's unit
    // direction n in the slot's point: the rows accumulate 3 x 3 symmetric matrices (xx xy xz yy yz zz) and the
    // 7 x 7 blocks are formed once per slot -- 6 multiply-adds per row instead of 28, and the 2 x 28 block entries
    // are not read-modify-written inside the row loop (the arm's sweep lives in scratch: 1276 -> 1140 bytes per lane, 126 -> 115 us)
    double TQ[6] = {0, 0, 0, 0, 0, 0}, TC[6] = {0, 0, 0, 0, 0, 0};
    double Wsum = 0.0;
    auto addsym = [](double (&T)[6], const double w, const Vec3 &n) __attribute__((always_inline)) {
      const double wx = w * n.x, wy = w * n.y, wz = w * n.z;
      T[0] += wx * n.x; T[1] += wx * n.y; T[2] += wx * n.z; T[3] += wy * n.y; T[4] += wy * n.z; T[5] += wz * n.z;
    };
    if (SL == 0 && v.has_goal()) {
      // GoalReaching (goal_reaching.py:19-33), Gauss-Newton Hessian
      const double e0 = Pt.x - goalv[0], e1 = Pt.y - goalv[1], e2 = Pt.z - goalv[2];
      const double w0 = wgoalv[0], w1 = wgoalv[1], w2 = wgoalv[2];
      f += w0 * e0 * e0 + w1 * e1 * e1 + w2 * e2 * e2;
#pragma unroll
      for (int a = 0; a < NQ; a++) {
        gf[a] += 2.0 * (w0 * e0 * J[a].x + w1 * e1 * J[a].y + w2 * e2 * J[a].z);
        if constexpr (!C::FKCURV) {
#pragma unroll
          for (int c = a; c < NQ; c++)
            Qqq[a][c] += 2.0 * (w0 * J[a].x * J[c].x + w1 * J[a].y * J[c].y + w2 * J[a].z * J[c].z);
        }
      }
      if constexpr (C::FKCURV) {
        TQ[0] += 2.0 * w0; TQ[3] += 2.0 * w1; TQ[5] += 2.0 * w2;
        Fc = {-2.0 * w0 * e0, -2.0 * w1 * e1, -2.0 * w2 * e2};
      }
      if constexpr (C::DDCURV) {
        // what Gauss-Newton leaves out: 2 sum_c w_c e_c d2 p_c / dtheta2 (added to Q: subtracted from the block that is subtracted)
        Cqq[2][2] -= 2.0 * (w0 * e0 * ddP.x + w1 * e1 * ddP.y);
      }
    }
    auto fk_row_body = [&](const int r, const FkBuf &Bf) __attribute__((always_inline)) {
      const int i = v.fk_row(r), kind = v.fk_kind(r);
      const int fi = v.fk_idx(r);
      const double tcv = Bf.tcv, lcv = Bf.lcv, gold = Bf.gold;
      double gdz = 0.0;
      {
#pragma unroll
        for (int a = 0; a < NQ; a++) gdz += Bf.jo[a] * dzo[a];
        if constexpr (NS > 0) gdz += dzo[NX];
      }
      double gq[NQ];
      double h, cinv = 0.0;
      double ndd = 0.0;      // unit direction of the row . d2 p / dtheta2 (DDCURV)
      Vec3 nd = {0, 0, 0};   // unit direction of the row in the slot's point (FKCURV)
      if (kind == ROW_RADIAL) {
        // ||fk_l(q) - c_i|| - r_i - r_body (mpcBase.py:82-101)
        const Vec3 dv = {Pt.x - Bf.op[0], Pt.y - Bf.op[1], Pt.z - Bf.op[2]};
        const double dist = sqrt(dot(dv, dv));
        h = dist - Bf.op[3] - rbody;
        cinv = 1.0 / dist;
        if constexpr (C::FKCURV) nd = {dv.x * cinv, dv.y * cinv, dv.z * cinv};
        if constexpr (C::DDCURV) ndd = dot(dv, ddP) * cinv;
#pragma unroll
        for (int a = 0; a < NQ; a++) gq[a] = dot(dv, J[a]) * cinv;
      } else if (kind == ROW_LINEAR) {
        // |a.fk_l(q) + d| / ||a|| - r_body (LinearConstraints.py:25-40, utils.py:48-52)
        const Vec3 av = {Bf.op[0], Bf.op[1], Bf.op[2]};
        const double nrm = sqrt(dot(av, av));
        const double sd = dot(av, Pt) + Bf.op[3];
        const double sgn = sd < 0 ? -1.0 : 1.0;
        h = fabs(sd) / nrm - rbody;
        if constexpr (C::FKCURV) nd = {sgn * av.x / nrm, sgn * av.y / nrm, sgn * av.z / nrm};
        if constexpr (C::DDCURV) ndd = sgn * dot(av, ddP) / nrm;
#pragma unroll
        for (int a = 0; a < NQ; a++) gq[a] = sgn * dot(av, J[a]) / nrm;
      } else {
        // ||fk_a(q) - fk_b(q)|| - 2 r_body (SelfCollisionAvoidanceConstraints.py:19-27)
        const double dist = sqrt(dot(Pt, Pt));
        h = dist - 2.0 * rbody;
        cinv = 1.0 / dist;
        if constexpr (C::FKCURV) nd = {Pt.x * cinv, Pt.y * cinv, Pt.z * cinv};
        if constexpr (C::DDCURV) ndd = dot(Pt, ddP) * cinv;
#pragma unroll
        for (int a = 0; a < NQ; a++) gq[a] = dot(Pt, J[a]) * cinv;
      }
      // stage 1 (state pinned to xinit): state-only, unsoftened rows are constants of the
      // problem -- neutralised (value 1, zero gradient, no inverse-barrier term); DESIGN.md 2
      if (k == 0 && NS == 0) {
        h = 1.0;
        cinv = 0.0;
        ndd = 0.0;
        nd = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < NQ; a++) gq[a] = 0.0;
      }
      double cw = 0.0, c2row = 0.0;
      if (v.has_avoid() && v.fk_first(r)) {
        // inverse-barrier objective N w_i / h on the first row of a module (constraint_avoidance.py:22-31)
        const double wi = Bf.wi;
        const bool on = (wi != 0.0) && (k != 0);   // (selects: see the single-variable rows)
        const double cN = (double)M.N * wi;
        bad |= (int)(on & !(h > 0.0));   // (bitwise: a short-circuit branch would cut the rows into separate basic blocks)
        const double ih = frcp(h);
        f += on ? cN * ih : 0.0;
        const double c1 = on ? -cN * (ih * ih) : 0.0, c2 = on ? 2.0 * cN * (ih * ih * ih) : 0.0;
        cw = on ? cN * (ih * ih) : 0.0;
        c2row = c2;
#pragma unroll
        for (int a = 0; a < NQ; a++) {
          gf[a] += c1 * gq[a];
          if constexpr (!C::FKCURV) {
#pragma unroll
            for (int c = a; c < NQ; c++) Qqq[a][c] += c2 * gq[a] * gq[c];
          }
        }
      }
      double g = h;
      if constexpr (NS > 0) g += sl;  // softened rows (intended InequalityManager.py:29-32)
      grn[IDXL(i)] = g;
#pragma unroll
      for (int a = 0; a < NQ; a++) jqn[IDXL(fi * NQ + a)] = gq[a];
      const RowW rw = row_core(i, g, tcv, lcv, gold, gdz);
#pragma unroll
      for (int a = 0; a < NQ; a++) {
        q0[a] += gq[a] * rw.ca;
        q1[a] += gq[a] * rw.cb;
        rs[a] -= gq[a] * rw.lv;
        if constexpr (!C::FKCURV) {
#pragma unroll
          for (int c = a; c < NQ; c++) Qqq[a][c] += rw.sig * gq[a] * gq[c];
        }
        if constexpr (NS > 0) cs[a] += rw.sig * gq[a];
      }
      if constexpr (C::FKCURV) addsym(TQ, rw.sig + c2row, nd);
      if constexpr (NS > 0) {
        q0[NX] += rw.ca;
        q1[NX] += rw.cb;
        rs[NX] -= rw.lv;
        Dg[NX] += rw.sig;
      }
      if constexpr (QC) {
        // exact Hessian of the distance rows when the kinematics are affine in q:
        // grad^2 h = (J^T J - g g^T) / dist, weighted by the multiplier and the inverse-barrier term
        // (weight selected, not branched on: the rows of the slot stay one basic block)
        const double wgt = (M.use_curv && kind != ROW_LINEAR) ? (rw.lv + cw) * cinv : 0.0;
        if constexpr (C::FKCURV) {
          // (J^T J - g g^T) / dist = J^T (I - n n^T) J / dist
          Wsum += wgt;
          addsym(TC, wgt, nd);
          const double wf = rw.lv + cw;
          Fc.x += wf * nd.x; Fc.y += wf * nd.y; Fc.z += wf * nd.z;
        } else {
#pragma unroll
          for (int a = 0; a < NQ; a++)
#pragma unroll
            for (int c = a; c < NQ; c++) Cqq[a][c] += wgt * (dot(J[a], J[c]) - gq[a] * gq[c]);
          // (the unicycle: the frame turns with the base -- the row's direction times d2 p / dtheta2)
          if constexpr (C::DDCURV) Cqq[2][2] += M.use_curv ? (rw.lv + cw) * ndd : 0.0;
        }
      }
    };
    if constexpr (V::SPEC) {
      // generated view: the rows of the slot are known at compile time -- straight-line code
      for_range<0, V::nfkrows()>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r >= V::slot_row_begin(SL) && r < V::slot_row_begin(SL + 1)) fk_row_body(r, fkb[r]);
      });
    } else {
      // (runtime tables: the requests of the next row of the slot leave before this row's arithmetic)
      const int rb0 = v.slot_row_begin(SL), re0 = v.slot_row_begin(SL + 1);
      if constexpr (C::FKCURV) {
        // (the arms: no register to spare for a second row's inputs -- 420 -> 564 B of scratch, sweep 98 -> 102 us)
        for (int r = rb0; r < re0; r++) {
          fk_load(r, fkb[0]);   // requests first, arithmetic after
          fk_row_body(r, fkb[0]);
        }
      } else if (rb0 < re0) {
        FkBuf nxt;
        fk_load(rb0, nxt);
        for (int r = rb0; r < re0; r++) {
          fkb[0] = nxt;
          fk_load(r + 1 < re0 ? r + 1 : r, nxt);
          __builtin_amdgcn_sched_barrier(0);
          fk_row_body(r, fkb[0]);
        }
      }
    }
    if constexpr (C::FKCURV) {
      // second derivatives of the slot's point: for joints a before c on the chain dJ_c/dq_a = axis_a x J_c when
      // joint a is revolute (it turns everything behind it, the column J_c included), 0 when it is prismatic;
      // Fc . (axis_a x J_c) = (Fc x axis_a) . J_c.  Columns beyond the slot's frames are zero.
      auto symv = [](const double (&T)[6], const Vec3 &x) __attribute__((always_inline)) -> Vec3 {
        return {T[0] * x.x + T[1] * x.y + T[2] * x.z, T[1] * x.x + T[3] * x.y + T[4] * x.z, T[2] * x.x + T[4] * x.y + T[5] * x.z};
      };
#pragma unroll
      for (int a = 0; a < NQ; a++) {
        const Vec3 u = symv(TQ, J[a]);
#pragma unroll
        for (int c = a; c < NQ; c++) qacc[qtri(a, c) * kSweepBlock] += dot(u, J[c]);
      }
      if (M.use_curv) {
#pragma unroll
        for (int a = 0; a < NQ; a++) {
          const Vec3 t = symv(TC, J[a]);
          Vec3 w = {Wsum * J[a].x - t.x, Wsum * J[a].y - t.y, Wsum * J[a].z - t.z};
          if (v.joint_type(a) == RMPC_JOINT_REVOLUTE) {
            const Vec3 G = cross(Fc, kin.aj[a]);
            w = {w.x + G.x, w.y + G.y, w.z + G.z};
          }
#pragma unroll
          for (int c = a; c < NQ; c++) qacc[(C::NQ2 + qtri(a, c)) * kSweepBlock] += dot(w, J[c]);
        }
      }
    }
  };
  do_slot(std::integral_constant<int, 0>{});
  do_slot(std::integral_constant<int, 1>{});
  do_slot(std::integral_constant<int, 2>{});
  do_slot(std::integral_constant<int, 3>{});

  SW_STAMP(1);
  // ---- the remaining single-variable rows -----------------------------------------------
  run_vars(std::integral_constant<int, Ord::NFIRST>{}, std::integral_constant<int, NV>{}, FalseT{},
           std::integral_constant<bool, PRE2>{});

  SW_STAMP(2);
  // ---- dynamics defect and stationarity -------------------------------------------
  double req = 0.0;
  if constexpr (CHAIN) {
    if (k < N - 1) {
      double xn[NX];
      chain_step<C>(M.dt, z, xn);
#pragma unroll
      for (int j = 0; j < NX; j++) {
        const double r = xn[j] - (QLDS ? (double)qacc[(2 * C::NQ2 + j) * kSweepBlock] : xk1[j]);
        rec[C::R_RC + j] = r;
        req = fmax(req, fabs(r));
        theta += fabs(r);
      }
    } else {
      // (the last stage has no defect, but the recursion reads the entries -- times a zero cost-to-go; records in
      //  LDS start from whatever the previous kernel left there, and 0 * NaN is not 0)
#pragma unroll
      for (int j = 0; j < NX; j++) rec[C::R_RC + j] = 0.0;
    }
    if constexpr (EARLY) {
      for_range<0, NQ>(finalize_var);
      if constexpr (NS > 0) finalize_var(std::integral_constant<int, NX>{});
    } else {
      for_range<0, NV>(finalize_var);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NV; j++) rs[j] += gf[j];
    if (k < N - 1) {
      double xn[NX];
      double A5[25], B5[10];
      diffdrive_step<C>(M.dt, z, xn, A5, B5, true);
      constexpr int map[5] = {0, 1, 2, 6, 7};
#pragma unroll
      for (int i = 0; i < 25; i++) rec[C::R_A5 + i] = A5[i];
#pragma unroll
      for (int i = 0; i < 10; i++) rec[C::R_B5 + i] = B5[i];
      {
        // nu . grad^2 Phi of the discrete dynamics (ERK2 midpoint, 5 nodes; closed form of diffdrive_step):
        // x+ = x + h sum_n cos(al_n) be_n, y+ = y + h sum_n sin(al_n) be_n, al_n = theta + a_n omega + b_n u1,
        // be_n = v + a_n u0, a_n = (n + 1/2) h, b_n = h^2 n (n + 1) / 2 -- only the costates of x and y carry curvature:
        // D = h sum_n [(-nx cos - ny sin) be_n ga ga^T + (-nx sin + ny cos)(ga gb^T + gb ga^T)], ga = (1, a_n, b_n) over
        // (theta, omega, u1), gb = (1, a_n) over (v, u0).  Stored negated (the recursion subtracts cwt x the entry).
        const double hn = M.dt / kErkNodes;
        const double th = z[2], vv = z[6], om = z[7], u0 = z[NX + NS], u1 = z[NX + NS + 1];
        double Dd[C::ND + 1];
#pragma unroll
        for (int i = 0; i <= C::ND; i++) Dd[i] = 0.0;
#pragma unroll 1
        for (int nn_ = 0; nn_ < kErkNodes; nn_++) {
          const double an = (nn_ + 0.5) * hn, bn = hn * hn * (double)(nn_ * (nn_ + 1)) * 0.5;
          double sn, cn;
          sincos(th + an * om + bn * u1, &sn, &cn);
          const double be = vv + an * u0;
          const double Pn = hn * (-nun[0] * cn - nun[1] * sn) * be, Sn = hn * (-nun[0] * sn + nun[1] * cn);
          Dd[0] += Pn * an; Dd[1] += Pn * bn; Dd[2] += Pn * an * an; Dd[3] += Pn * an * bn; Dd[4] += Pn * bn * bn;
          Dd[5] += Sn; Dd[6] += Sn * an; Dd[7] += Sn * an; Dd[8] += Sn * an * an; Dd[9] += Sn * bn; Dd[10] += Sn * bn * an;
          Dd[C::ND] += Pn;   // (theta, theta): into the q block
        }
#pragma unroll
        for (int i = 0; i < C::ND; i++) rec[C::R_D + i] = M.use_curv ? -Dd[i] : 0.0;
        Cqq[2][2] -= M.use_curv ? Dd[C::ND] : 0.0;
      }
      // A = I outside the reduced block
#pragma unroll
      for (int j = 3; j < 6; j++) rs[j] += nun[j];
#pragma unroll
      for (int c = 0; c < 5; c++) {
        double acc = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) acc += A5[r * 5 + c] * nun[map[r]];
        rs[map[c]] += acc;
      }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        double acc = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) acc += B5[r * 2 + c] * nun[map[r]];
        rs[NX + NS + c] += acc;
      }
#pragma unroll
      for (int j = 0; j < NX; j++) {
        const double r = xn[j] - xk1[j];
        rec[C::R_RC + j] = r;
        req = fmax(req, fabs(r));
        theta += fabs(r);
      }
    } else {
      // (see the holonomic chain: every entry the recursion reads is written)
#pragma unroll
      for (int i = 0; i < 35 + C::ND; i++) rec[C::R_A5 + i] = 0.0;
#pragma unroll
      for (int j = 0; j < NX; j++) rec[C::R_RC + j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
      double r = rs[j];
      if (j < NX) {
        if (k == 0) continue;  // x_1 is fixed: no stationarity condition
        r -= nuk[j];
      }
      rstat = fmax(rstat, fabs(r));
    }
#pragma unroll
    for (int j = NQ; j < NV; j++) rec[C::R_DG + j - NQ] = Dg[j];
    if constexpr (NS > 0) {
#pragma unroll
      for (int j = 0; j < NV; j++) rec[C::R_CS + j] = cs[j];
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
      rec[C::R_Q0 + j] = gf[j] + q0[j];
      rec[C::R_Q1 + j] = q1[j];
      gfa[IDXL(j)] = gf[j];
    }
  }

  // ---- write the q block of the condensed stage ------------------------------------------
  {
    int s = 0;
#pragma unroll
    for (int a = 0; a < NQ; a++)
#pragma unroll
      for (int c = a; c < NQ; c++) {
        if constexpr (QLDS) rec[C::R_Q + s] = qacc[s * kSweepBlock];
        else rec[C::R_Q + s] = Qqq[a][c];
        s++;
      }
  }
  {
    // (zero when the model or this solve does not use the curvature terms: k_riccati reads the slot regardless)
    int s = 0;
#pragma unroll
    for (int a = 0; a < NQ; a++)
#pragma unroll
      for (int c = a; c < NQ; c++) {
        if constexpr (QLDS) rec[C::R_C + s] = M.use_curv ? (double)qacc[(C::NQ2 + s) * kSweepBlock] : 0.0;
        else rec[C::R_C + s] = (QC && M.use_curv) ? Cqq[QC ? a : 0][QC ? c : 0] : 0.0;
        s++;
      }
  }
  rec[C::R_ZERO] = 0.0;
  const double logsum = log(lprod) + 0.6931471805599453094 * (double)lexp;
  bad |= (int)(!isfinite(f) | !isfinite(theta) | !isfinite(logsum));
  SW_STAMP(3);
  out.f = f; out.th = theta; out.logs = logsum; out.rstat = rstat; out.req = req; out.rineq = rineq;
  out.rcomp = rcomp; out.sumc = sumc; out.minc = minc; out.bad = (double)bad;
}

template <class C, class V>
__global__ __launch_bounds__(kSweepBlock, C::SWEEP_WPE) void k_sweep(const DevModel M, const DevTables *__restrict__ Tp, const Ws W,
                                               const int B, const int first, const int warm) {
#ifdef RMPC_STAMPS
  const long long ks_t0 = __builtin_amdgcn_s_memtime();
#endif
  const int gid = blockIdx.x * kSweepBlock + threadIdx.x;
  const int li = gid % W.Bp;   // position in the compacted list of iterating instances
  // (Bp % 64 == 0: the stage is the same for the 64 lanes of a wavefront; as a scalar, every test on it is a scalar
  //  branch taken by the whole wavefront instead of a masked region)
  const int k = __builtin_amdgcn_readfirstlane(gid / W.Bp);
  if (li >= *W.n_act || k >= M.N) return;
  const int b = W.act_idx[li];
  if (W.status[b] != ST_ACTIVE) return;
  const int N = M.N;
  (void)B;
  const int cur = W.cur[b], nxt = cur ^ 1;
  SweepIO<gdouble> io;
  io.zc = (gdouble *)W.z[cur]; io.tc = (gdouble *)W.t[cur]; io.lc = (gdouble *)W.lam[cur]; io.nc = (gdouble *)W.nu[cur];
  io.zn = (gdouble *)W.z[nxt]; io.tn = (gdouble *)W.t[nxt]; io.ln = (gdouble *)W.lam[nxt]; io.nn = (gdouble *)W.nu[nxt];
  io.pp = (gdouble *)W.p; io.dzp = (gdouble *)W.dz; io.gro = (gdouble *)W.grow[cur]; io.jqo = (gdouble *)W.Jq[cur];
  io.grn = (gdouble *)W.grow[nxt]; io.jqn = (gdouble *)W.Jq[nxt];
  io.nup = (gdouble *)W.nunew; io.gfa = (gdouble *)W.gfa;
  io.rec = (gdouble *)(W.R + ((size_t)b * N + k) * C::RS);   // this lane's stage record
  // element offset of this lane inside a slot (32-bit, so that accesses become uniform base + lane offset) and slot size
  io.loff = (unsigned)k * (unsigned)W.Bp + (unsigned)b;
  io.kstride = (unsigned)W.Bp;
  io.SS = (size_t)N * W.Bp;
  io.SSd = io.SS; io.loffd = io.loff; io.kstrided = io.kstride;
  io.wl = (gdouble *)W.wlam; io.wn = (gdouble *)W.wnu;
  io.warm = warm;   // (wave uniform; instances without usable multipliers hold zeros and mu0 in the warm arrays)
  // ---- step lengths of this trial --------------------------------------
  // null pass: the current point is re-evaluated unchanged so that the step can be
  // recomputed with the Gauss-Newton blocks (fallback of a failed curvature step)
  const bool nostep = first || (W.redo[b] != 0);
  double alpha = 0.0, adual = 0.0;
  if (!nostep) {
    alpha = ldexp(__longlong_as_double((long long)W.amin_p[b]), -W.ls[b]);
    adual = __longlong_as_double((long long)W.amin_d[b]);
  }
  Partials pt;
  const V v(M, *Tp);
  const SweepK sk = {M.N, M.dt, M.use_curv};
  __shared__ double sq[C::FKCURV ? (2 * C::NQ2 + C::NX + C::NQ) * kSweepBlock : 1];
  ldouble *const qacc = (ldouble *)sq + threadIdx.x;
  if (first) sweep_body<C, -1, gdouble, V, 1>(sk, v, io, k, true, nostep, alpha, adual, W.mu[b], pt, qacc);
  else sweep_body<C, -1, gdouble, V, 0>(sk, v, io, k, false, nostep, alpha, adual, W.mu[b], pt, qacc);
  const unsigned loff = io.loff;
  const size_t SS = io.SS;
  W.part[IDXL(P_F)] = pt.f;
  W.part[IDXL(P_TH)] = pt.th;
  W.part[IDXL(P_LOGS)] = pt.logs;
  W.part[IDXL(P_RSTAT)] = pt.rstat;
  W.part[IDXL(P_REQ)] = pt.req;
  W.part[IDXL(P_RINEQ)] = pt.rineq;
  W.part[IDXL(P_RCOMP)] = pt.rcomp;
  W.part[IDXL(P_SUMC)] = pt.sumc;
  W.part[IDXL(P_MINC)] = pt.minc;
  W.part[IDXL(P_BAD)] = pt.bad;
#ifdef RMPC_STAMPS
  if ((threadIdx.x & 63) == 0) {
    for (int i = 0; i < 4; i++) atomicAdd((unsigned long long *)&g_sst[i], (unsigned long long)pt.tk[i]);
    atomicAdd((unsigned long long *)&g_sst[4], (unsigned long long)(__builtin_amdgcn_s_memtime() - ks_t0));
    atomicAdd((unsigned long long *)&g_sst[7], 1ull);
  }
#endif
}

// ===========================================================================
// k_riccati: per-instance decisions + block-tridiagonal Riccati recursion
// ===========================================================================
// One 64-lane wavefront per instance.  The stage matrices live in LDS and every
// small dense operation of the recursion is spread over the lanes (one output
// entry per lane and pass), so the dependent chain per stage is a handful of
// LDS round trips instead of ~1500 serial fp64 instructions of one lane.
//   backward, stage k:  fill Q_k, q_k      (compact blocks -> dense (nx+nw)^2, lanes over entries)
//                       T = P [A|B], Pc = P rc + p
//                       Q += [A|B]^T T, q += [A|B]^T Pc
//                       Cholesky of Qww (every lane, registers), gains K | kff (one column per lane)
//                       P = sym(Qxx + Qxw K), p = qx + Qxw kff
//   forward, stage k:   dw = K dx + kff, nu+ = P dx + p, dx+ = [A|B][dx; dw] + rc
template <int NW>
__device__ __forceinline__ void chol_solve(const double (&L)[NW][NW], const double (&invd)[NW], double (&v)[NW]) {
  // L L^T x = v with the reciprocals of the diagonal supplied (no divisions on the chain)
#pragma unroll
  for (int i = 0; i < NW; i++) {
    double s = v[i];
#pragma unroll
    for (int l = 0; l < i; l++) s -= L[i][l] * v[l];
    v[i] = s * invd[i];
  }
#pragma unroll
  for (int i = NW - 1; i >= 0; i--) {
    double s = v[i];
#pragma unroll
    for (int l = i + 1; l < NW; l++) s -= L[l][i] * v[l];
    v[i] = s * invd[i];
  }
}

// LDS hand-off inside ONE wavefront: DS instructions of a wave execute in issue order, so a
// compiler-level ordering point is all that is needed (a __syncthreads() would also drain the
// global loads that are deliberately left in flight as the next stage's prefetch).
#define WSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// reductions over the LPI consecutive lanes that work on one instance (a whole wavefront or half of one)
template <int LPI>
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = LPI / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int LPI>
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = LPI / 2; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
template <int LPI>
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = LPI / 2; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  return v;
}

// Several reductions at once, step by step: the exchanges of one step of all of them are issued together (a single
// reduction is a chain of dependent LDS-crossbar round trips; done one after the other, eleven of them cost eleven
// chains).  Same partner pattern, hence the same rounding, as wave_sum / wave_max / wave_min.
template <int LPI, int NS_, int NM_, int NN_>
__device__ __forceinline__ void wave_reduce_many(double (&sums)[NS_], double (&maxs)[NM_], double (&mins)[NN_]) {
#pragma unroll
  for (int off = LPI / 2; off >= 1; off >>= 1) {
    double ts[NS_], tm[NM_], tn[NN_];
#pragma unroll
    for (int i = 0; i < NS_; i++) ts[i] = __shfl_xor(sums[i], off, 64);
#pragma unroll
    for (int i = 0; i < NM_; i++) tm[i] = __shfl_xor(maxs[i], off, 64);
#pragma unroll
    for (int i = 0; i < NN_; i++) tn[i] = __shfl_xor(mins[i], off, 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NS_; i++) sums[i] += ts[i];
#pragma unroll
    for (int i = 0; i < NM_; i++) maxs[i] = fmax(maxs[i], tm[i]);
#pragma unroll
    for (int i = 0; i < NN_; i++) mins[i] = fmin(mins[i], tn[i]);
  }
}

// ---- per-instance solver state ------------------------------------------------------------------------
// One set of words per instance.  The pass kernels keep them in the workspace (arrays over the batch), the
// fused kernel in registers of the wavefront that owns the instance; the decision logic is the same code.
struct Inst {
  double mu, rho, phi0, Dd, fcur, thcur, logcur, res_stat, res_eq, res_ineq, res_comp, obj;
  double amin_p, amin_d;   // fraction-to-the-boundary step lengths of the current step
  double mu_hold;          // barrier restart: the level mu is held at (0: none)
  int status, iters, ls, ls0, lsst, cur, newstep, redo, force_gn, gn_sticky, curv_fail, usedc, stall, curv_skip, curv_back;
  int small_steps;         // barrier restart: accepted short steps in a row
  double theta_mem, theta_c;      // scaled curvature: the scale the next curvature step starts from / of this iteration
  int theta_clean, theta_retry;   // accepted curvature steps in a row without a retry / this iteration has retried
};
__device__ __forceinline__ void inst_init(Inst &s, double mu0) {
  s.mu = mu0; s.rho = 0.0; s.phi0 = 0.0; s.Dd = 0.0; s.fcur = 0.0; s.thcur = 0.0; s.logcur = 0.0;
  s.res_stat = 0.0; s.res_eq = 0.0; s.res_ineq = 0.0; s.res_comp = 0.0; s.obj = 0.0;
  s.amin_p = 1.0; s.amin_d = 1.0;
  s.status = ST_ACTIVE; s.iters = 0; s.ls = 0; s.ls0 = 0; s.lsst = 0; s.cur = 0; s.newstep = 0; s.redo = 0;
  s.force_gn = 0; s.gn_sticky = 0; s.curv_fail = 0; s.usedc = 0; s.stall = 0; s.curv_skip = 0; s.curv_back = 0;
  s.small_steps = 0; s.mu_hold = 0.0;
  s.theta_mem = 1.0; s.theta_c = 1.0; s.theta_clean = 0; s.theta_retry = 0;
}
__device__ __forceinline__ void inst_load(Inst &s, const Ws &W, int b) {
  s.mu = W.mu[b]; s.rho = W.rho[b]; s.phi0 = W.phi0[b]; s.Dd = W.Dd[b]; s.fcur = W.fcur[b]; s.thcur = W.thcur[b];
  s.logcur = W.logcur[b]; s.res_stat = W.res_stat[b]; s.res_eq = W.res_eq[b]; s.res_ineq = W.res_ineq[b];
  s.res_comp = W.res_comp[b]; s.obj = W.obj[b];
  s.amin_p = __longlong_as_double((long long)W.amin_p[b]); s.amin_d = __longlong_as_double((long long)W.amin_d[b]);
  s.status = W.status[b]; s.iters = W.iters[b]; s.ls = W.ls[b]; s.ls0 = W.ls0[b]; s.lsst = W.lsst[b]; s.cur = W.cur[b];
  s.newstep = W.newstep[b]; s.redo = W.redo[b]; s.force_gn = W.force_gn[b]; s.gn_sticky = W.gn_sticky[b];
  s.curv_fail = W.curv_fail[b]; s.usedc = W.usedc[b]; s.stall = W.stall[b]; s.curv_skip = W.curv_skip[b]; s.curv_back = W.curv_back[b];
  s.small_steps = W.small_steps[b]; s.mu_hold = W.mu_hold[b];
  s.theta_mem = W.theta_mem[b]; s.theta_c = W.theta_c[b]; s.theta_clean = W.theta_clean[b]; s.theta_retry = W.theta_retry[b];
}
__device__ __forceinline__ void inst_store(const Inst &s, const Ws &W, int b) {
  W.mu[b] = s.mu; W.rho[b] = s.rho; W.phi0[b] = s.phi0; W.Dd[b] = s.Dd; W.fcur[b] = s.fcur; W.thcur[b] = s.thcur;
  W.logcur[b] = s.logcur; W.res_stat[b] = s.res_stat; W.res_eq[b] = s.res_eq; W.res_ineq[b] = s.res_ineq;
  W.res_comp[b] = s.res_comp; W.obj[b] = s.obj;
  W.amin_p[b] = (unsigned long long)__double_as_longlong(s.amin_p); W.amin_d[b] = (unsigned long long)__double_as_longlong(s.amin_d);
  W.status[b] = s.status; W.iters[b] = s.iters; W.ls[b] = s.ls; W.ls0[b] = s.ls0; W.lsst[b] = s.lsst; W.cur[b] = s.cur;
  W.newstep[b] = s.newstep; W.redo[b] = s.redo; W.force_gn[b] = s.force_gn; W.gn_sticky[b] = s.gn_sticky;
  W.curv_fail[b] = s.curv_fail; W.usedc[b] = s.usedc; W.stall[b] = s.stall; W.curv_skip[b] = s.curv_skip; W.curv_back[b] = s.curv_back;
  W.small_steps[b] = s.small_steps; W.mu_hold[b] = s.mu_hold;
  W.theta_mem[b] = s.theta_mem; W.theta_c[b] = s.theta_c; W.theta_clean[b] = s.theta_clean; W.theta_retry[b] = s.theta_retry;
}

// whole-horizon sums / maxima of the trial point the last sweep evaluated (+ the merit slope of the step)
struct Reduced { double f, th, lgs, rstat, req, rineq, rcomp, sumc, minc, badf, gphi; };

// Armijo test of the trial point, acceptance, barrier update, convergence tests.  Returns true when a new
// step has to be computed (Riccati recursion next; `usec`: with the exact constraint curvature); false when
// the instance retries with a shorter step, re-evaluates (null pass) or has stopped (s.status).
template <class C>
__device__ __forceinline__ bool inst_decide(const DevModel &M, Inst &s, const Reduced &r, const bool first, bool &usec) {
  const int N = M.N;
  s.newstep = 0;
  double mu = s.mu;
  int status = ST_ACTIVE;
  int iters = s.iters;
  const bool redo = (!first) && (s.redo != 0);
  int lsst = first ? 0 : s.lsst;
  double alpha_acc = 1.0;   // length of the step accepted in this pass (barrier restart)
  usec = false;
  if (first) {
    if (r.badf != 0.0) status = -7;  // inverse-barrier row not strictly feasible at the start
  } else if (redo) {
    // null pass: same point, the step is recomputed below with the Gauss-Newton blocks
    s.redo = 0;
  } else {
    const double a0 = s.amin_p;
    int ls = s.ls;
    double rho = s.rho, phi0 = s.phi0, Dd = s.Dd;
    if (ls == s.ls0) {   // first trial of this line search
      const double thc = s.thcur;
      if (thc > 1e-13) {
        const double need = r.gphi / (0.9 * thc);
        if (rho < need) rho = need + 1.0;
      }
      Dd = r.gphi - rho * thc;
      phi0 = s.fcur - mu * s.logcur + rho * thc;
      s.rho = rho; s.phi0 = phi0; s.Dd = Dd;
    }
    const double alpha = ldexp(a0, -ls);
    const double phi = r.f - mu * r.lgs + rho * r.th;
    const bool ok = (r.badf == 0.0) && (phi <= phi0 + kArmijo * alpha * Dd + 1e-13 * fabs(phi0));
    const int usedc = s.usedc;
    if (!ok) {
      ls++;
      if (ls > (usedc ? kLsCurv - 1 : M.ls_max)) {
        if (usedc) {
          // the curvature step failed its line search: recompute this iteration's step with
          // the Gauss-Newton blocks (null pass next); latch after repeated failures
          if constexpr (C::BACKOFF) {
            // (the unicycle, the small chains: the next curvature steps are skipped -- 1, 2, 4 .. 16 iterations, doubling
            //  with every failure in a row, over after a success -- instead of a latch: DESIGN.md 3)
            s.curv_back = s.curv_back ? (s.curv_back < kCurvBackMax ? 2 * s.curv_back : kCurvBackMax) : 1;
            s.curv_skip = s.curv_back;
          } else {
            const int cf = s.curv_fail + 1;
            s.curv_fail = cf;
            if (cf >= kCurvFailMax) s.gn_sticky = 1;
          }
          s.redo = 1;
          s.force_gn = 1;
          s.ls = 0;
          return false;
        }
        s.status = -8;  // line search failure; the current iterate is returned
        return false;
      }
      s.ls = ls;
      return false;  // next sweep retries with alpha / 2
    }
    if (usedc) { s.curv_fail = 0; s.curv_back = 0; }
    if constexpr (C::CSCALE) {
      // scaled curvature: the scale that needed a retry is kept, kCsClean accepted curvature steps in a row without a
      // retry double it again; an iteration whose retries all failed (Gauss-Newton step accepted) keeps the last scale
      if (usedc) {
        if (s.theta_retry) { s.theta_mem = s.theta_c; s.theta_clean = 0; }
        else if (++s.theta_clean >= kCsClean) { s.theta_mem = s.theta_c < 0.75 ? 2.0 * s.theta_c : 1.0; s.theta_clean = 0; }
      } else if (s.theta_retry) { s.theta_mem = s.theta_c; s.theta_clean = 0; }
    }
    // the arms: a Gauss-Newton step accepted at full length releases the latch (the failures that set it belong to
    // the first iterations of a warm start, where the fraction to the boundary cuts the steps)
    if constexpr (C::FKCURV) {
      if (!usedc && ls == 0) { s.gn_sticky = 0; s.curv_fail = 0; }
    }
    // step-length memory: the next Gauss-Newton line search starts one halving above the accepted one (models
    // whose steps overshoot every iteration -- the unicycle -- otherwise pay a pass per halving per iteration)
    lsst = ls > kLsGrow ? ls - kLsGrow : 0;
    s.lsst = lsst;
    alpha_acc = alpha;
    iters++;
  }
  // ---- accept the trial point ------------------------------------------------------
  if (status == ST_ACTIVE) {
    const double f_prev = s.fcur;
    const int stall0 = s.stall;
    s.cur ^= 1;
    s.fcur = r.f;
    s.thcur = r.th;
    s.logcur = r.lgs;
    if (!redo) {
      s.iters = iters;
      s.res_stat = r.rstat; s.res_eq = r.req; s.res_ineq = r.rineq; s.res_comp = r.rcomp; s.obj = r.f;
      if (!first) {
        // LOQO-style centrality rule with floors (DESIGN.md, section "Algorithm")
        const double cnt = (double)N * (double)M.m;
        const double avg = r.sumc / cnt;
        const double xi = r.minc / avg;
        double sg = 0.05 * (1.0 - xi) / xi;
        if (sg > 2.0) sg = 2.0;
        sg = 0.1 * sg * sg * sg;
        if (sg < 0.02) sg = 0.02;
        if (sg > 0.8) sg = 0.8;
        mu = sg * avg;
        if (mu < 0.1 * M.tol_comp) mu = 0.1 * M.tol_comp;
        // barrier restart on stalled steps (oracle: ORC_RS_*): from iteration kRsIt on, kRsN accepted steps in
        // a row shorter than kRsAlpha while mu < kRsMu -- the iterate crawls along a boundary with the barrier at its
        // floor -- hold mu at kRsMu, released by the factor kRsDecay per iteration
        {
          int ss = (iters - 1 >= kRsIt && alpha_acc < kRsAlpha) ? s.small_steps + 1 : 0;
          double mh = s.mu_hold;
          if (ss >= kRsN && mu < kRsMu && !(mh > 0.0)) { mh = kRsMu; ss = 0; }
          if (mh > 0.0) {
            if (mu < mh) mu = mh;
            mh *= kRsDecay;
            if (mh < 0.1 * M.tol_comp) mh = 0.0;
          }
          s.small_steps = ss; s.mu_hold = mh;
        }
        s.mu = mu;
        if (!(mu < kMuDiverged)) status = -7;
      }
      if (status == ST_ACTIVE) {
        if (!isfinite(r.rstat) || !isfinite(r.req) || !isfinite(r.rineq)) status = -6;
        else if (r.rstat <= M.tol_stat && r.req <= M.tol_eq && r.rineq <= M.tol_ineq && r.rcomp <= kCompFrac * M.tol_comp) status = 1;
        else {
          // acceptable termination: feasible, complementary, objective stagnant for acc_iters iterations
          int stall = stall0;
          if (!first && r.req <= kAccFeas && r.rineq <= kAccFeas && r.rcomp <= kAccFeas &&
              fabs(r.f - f_prev) <= M.acc_obj_tol * fmax(1.0, fabs(r.f)))
            stall++;
          else
            stall = 0;
          s.stall = stall;
          if (M.acc_iters > 0 && stall >= M.acc_iters) status = 2;
          else if (iters >= M.max_iter) status = 0;
        }
      }
    }
  }
  if (status != ST_ACTIVE) {
    s.status = status;
    return false;
  }
  // exact constraint curvature unless latched off or this is the fallback pass
  if constexpr (C::CURV || C::DDCURV) usec = M.use_curv && !s.gn_sticky && !s.force_gn && (mu <= kCurvMu);
  if constexpr (C::BACKOFF) {
    // (a fallback pass -- force_gn -- is not an iteration of its own: the skip counter moves once per iteration)
    if (usec && s.curv_skip > 0) { s.curv_skip--; usec = false; }
  }
  s.force_gn = 0;
  if constexpr (C::CSCALE) {
    if (!redo) { s.theta_c = s.theta_mem; s.theta_retry = 0; }   // (a null pass belongs to the iteration that asked for it)
  }
  // a step with the exact curvature is tried at full length first
  const int lsb = usec ? 0 : lsst;
  s.ls = lsb;
  s.ls0 = lsb;
  return true;
}
// after the recursion: a failed factorisation either falls back to Gauss-Newton (null pass) or stops the instance
__device__ __forceinline__ void inst_after_recursion(Inst &s, const bool chol_ok, const bool usec, const bool backoff = false,
                                                     const bool cscale = false) {
  if (!chol_ok) {
    if (usec) {
      if (cscale && s.theta_c > kCsMin) {
        // scaled curvature: the same iteration again (null pass next) with the curvature terms at half their weight
        s.theta_c *= 0.5; s.theta_retry = 1; s.redo = 1; s.usedc = 0;
        return;
      }
      if (backoff) {   // (diff-drive: see inst_decide)
        s.curv_back = s.curv_back ? (s.curv_back < kCurvBackMax ? 2 * s.curv_back : kCurvBackMax) : 1;
        s.curv_skip = s.curv_back;
      }
      // reduced Hessian not positive definite with the curvature terms: recompute this
      // iteration's step with the Gauss-Newton blocks (null pass next); not counted as a
      // line-search failure
      s.redo = 1; s.force_gn = 1; s.usedc = 0;
      return;
    }
    s.status = -5;
    return;
  }
  s.usedc = usec ? 1 : 0;
  s.newstep = 1;
  s.amin_p = 1.0;   // the step kernel takes the minima next
  s.amin_d = 1.0;
}

// where the recursion leaves the step: dz[slot * SS + k * KS], nunew likewise (pointers advanced to the instance)
template <class RP = gdouble>
struct StepOut {
  RP *dz, *nunew;
  size_t SS, KS;
};

#ifdef RMPC_RIC_STAMPS
// development aid: cycles per phase of the generic recursion path, summed over the wavefronts of all launches
__device__ long long g_rst[8];
#define RST_DECL() long long rst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rst_t0 = __builtin_amdgcn_s_memtime()
#define RST(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); rst_acc[i] += t_ - rst_t0; rst_t0 = t_; } while (0)
#define RST_FLUSH() do { if (lane == 0) { for (int i_ = 0; i_ < 7; i_++) atomicAdd((unsigned long long *)&g_rst[i_], (unsigned long long)rst_acc[i_]); atomicAdd((unsigned long long *)&g_rst[7], 1ull); } } while (0)
#else
#define RST_DECL()
#define RST(i)
#define RST_FLUSH()
#endif
template <class C, int LPI>
struct RicLds {
  static constexpr int NX = C::NX, NV = C::NV, NW = C::NW;
  static constexpr int NP2 = NX * (NX + 1) / 2;
  static constexpr int KPW = NW * NX + NW + NP2 + NX + NX;
  // The arms (one wavefront per instance, pass kernels): the chain model uses neither the [A|B] area nor T, and what
  // that saves holds gain images: IMG_SLOTS stages' images stay in LDS between the backward and the forward pass
  // instead of going through the gain record in global memory (budget: a quarter of the CU's 160 KB per wavefront).
  static constexpr bool LIMG = (C::ROBOT == RMPC_ROBOT_CHAIN) && LPI == 64 && NX > 8;
  static constexpr int ABW = LIMG ? 0 : NX * NV;    // [A|B]
  static constexpr int TW = LIMG ? 64 : NX * NV;    // T (chains: only the idle lanes' words)
  // The arms without slack, 9 .. 15 states (n = 5, 6, 7): the Schur-complement path of riccati_recursion (ARMB) with
  // a work area of its own -- image staging | P / Qxx (NX rows of APS doubles: 8-lane groups of a half wavefront on
  // distinct banks) | [Qux | qu] (NW rows of 16) | Quu (NW rows of 8) | Y operands (8 x 16) | p (16) | dx (2 x 16) |
  // a word per idle lane (64).  The stage records do not pass through LDS there.
  static constexpr bool ARMB = LIMG && C::NS == 0 && NX < 16;
  static constexpr int APS = 24;
  static constexpr int LDSW_ARM = KPW + APS * NX + 16 * NW + 8 * NW + 128 + 16 + 32 + 64;
  static constexpr int LDSW = ARMB ? LDSW_ARM
                                   : KPW + NX * NX + ABW + NV * NV + NV + TW + NX + NX + NW + C::RS;   // doubles per instance
  static constexpr int IMG_SLOTS = LIMG ? (40960 / 8 - LDSW) / KPW : 0;
};

// v moved between lanes by a DPP control word (quad permutations, row mirrors): full-rate vector moves, no LDS
// crossbar round trip.  All lanes of the wavefront must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_move(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over the 4 lanes of a quad; every lane of the quad ends with the total
__device__ __forceinline__ double dpp_sum4(double v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_move<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  return v;
}
// sum over the 8 aligned consecutive lanes a lane belongs to; every lane of the group ends with the total
__device__ __forceinline__ double dpp_sum8(double v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_move<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_move<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of its 8
  return v;
}

// Block-tridiagonal Riccati recursion of one instance, LPI lanes.  img: the instance's LDS row (RicLds::LDSW
// doubles); rb: its stage records (stage 0, stride C::RS; global memory or LDS); kpb: its gain records (stride
// kps, global memory).  Returns false when a stage's control block is not positive definite.
// Fused kernel, small models: everything the recursion exchanges stays in LDS.  An instance owns 32 slots of
// GS doubles, one per stage; slot k holds, in turn, the stage record the sweep wrote ([0, RW]), the gain image
// of the stage that the backward pass leaves for the forward pass ([0, KPW): the record is dead by then), and
// the step of the stage (dz | nu+ at [DZ_OFF, DZ_OFF + NV + NX): behind the record, so that the next sweeps
// read the step while they write their records).
template <class C>
struct FusedSlots {
  static constexpr int KPW = RicLds<C, 32>::KPW;
  static constexpr int DZ_OFF = C::RW + 1;
  static constexpr int NEED = (KPW > DZ_OFF + C::NV + C::NX) ? KPW : DZ_OFF + C::NV + C::NX;
  // 16-byte aligned slots (the sweep stores its record with aligned 16-byte writes) whose stride is NOT a multiple of
  // the 256 bytes the 64 LDS banks span: in the sweep and the step phase lane = stage, i.e. the lanes of an instance
  // address the same word of 32 different slots -- with a stride of 512 bytes every one of those requests was a
  // 32-way bank conflict (SQ_LDS_BANK_CONFLICT: 20 % of the LDS cycles of the kernel)
  // (+2 doubles: a stride of 4 banks -- 16-byte writes of 16 lanes cover the 64 banks once; A/B on one box, four
  //  batches in flight: 2.45-2.52 -> 2.61-2.62 M solves/s)
  static constexpr int GS = (NEED + 7) / 8 * 8 + 2;
};

template <class C, int LPI, bool SLOTS = false, class RP = gdouble, bool OWNER = false, class SP = RP>
__device__ __forceinline__ bool riccati_recursion(const int N, const double dt, const double mu, const double cw, const int lane,
                                                  ldouble *const img, const RP *const rb, gdouble *const kpb,
                                                  const int kps, const StepOut<SP> so, ldouble *const slots = nullptr,
                                                  ldouble *const limg = nullptr, const int lcap_rt = -1) {
  // SP: where the step goes (the fused arm kernel keeps it in LDS while its records are in global memory);
  // lcap_rt >= 0: number of gain-image slots behind limg (the arm path; otherwise RicLds::IMG_SLOTS)
  // SLOTS: rb == slots (records), gains go to the slots as well (kpb unused)
  // limg (RicLds::LIMG): RicLds::IMG_SLOTS gain images of KPW doubles in LDS, stages 1 .. IMG_SLOTS
  constexpr int GS = FusedSlots<C>::GS;
  constexpr int NQ = C::NQ, NX = C::NX, NS = C::NS, NV = C::NV, NW = C::NW;
  constexpr bool DD = (C::ROBOT == RMPC_ROBOT_DIFFDRIVE);
  const double cwt = cw;  // weight of the curvature terms (0: Gauss-Newton blocks, 1: exact, 1/2, 1/4: Cfg::CSCALE); Cqq is zero-filled when the model does not use it
  // ---- LDS images -------------------------------------------------------------------------
  // img = [K | kff | P (upper triangle) | p | rc]: what the forward pass needs of a stage, contiguous in
  // LDS so that it leaves for (and returns from) the instance's gain record KP in one request
  constexpr int NP2 = NX * (NX + 1) / 2;
  constexpr int KPW = NW * NX + NW + NP2 + NX + NX;
  constexpr int KPL = (KPW + LPI - 1) / LPI;
  constexpr bool LIMG = !SLOTS && RicLds<C, LPI>::LIMG;
  constexpr int LCAP = LIMG ? RicLds<C, LPI>::IMG_SLOTS : 0;   // stages 1 .. LCAP keep their image in LDS (limg)
  ldouble *const sK = img, *const skf = sK + NW * NX, *const sPt = skf + NW, *const sp = sPt + NP2,
               *const src = sp + NX, *const sP = img + KPW, *const sAB = sP + NX * NX, *const sQ = sAB + RicLds<C, LPI>::ABW,
               *const sq = sQ + NV * NV, *const sT = sq + NV, *const sPc = sT + RicLds<C, LPI>::TW, *const sdx = sPc + NX,
               *const sdw = sdx + NX, *const srec = sdw + NW;   // srec: the stage record as fetched
  auto tri = [](int i, int j) __attribute__((always_inline)) {   // index of (i, j) in the packed upper triangle
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return lo * NX - lo * (lo - 1) / 2 + (hi - lo);
  };
  const double h = dt, h2 = 0.5 * dt * dt;

  // ([A | B] of the holonomic chain is constant, A = [I hI; 0 I], B = [h2 I; h I] on the u columns: every
  //  product with it is written out in closed form below and sAB is used by the diff-drive model only)
  // Loop-invariant source of every LDS entry this lane fills: a pointer into the instance's stage
  // records (stage 0; an entry of the record, or its zero slot) plus a constant.  The per-stage fetch
  // is then an unconditional load per entry -- no branch around any load -- and all lanes of the
  // wavefront address the same few cache lines.
  constexpr size_t sstr = SLOTS ? (size_t)GS : (size_t)C::RS;   // stage stride of the records
  constexpr int EPL = (NV * NV + LPI - 1) / LPI;   // stage Hessian entries per lane
  constexpr int TPL = (NX * NV + LPI - 1) / LPI;   // entries of T = P [A|B] (and of [A|B]) per lane
  constexpr int RPL = (C::RS + LPI - 1) / LPI;     // record entries per lane
  int qp[EPL], cp[EPL];                      // record entries a dense-block entry of this lane is made of
#pragma unroll
  for (int u = 0; u < EPL; u++) {
    const int e = lane + LPI * u;
    qp[u] = C::R_ZERO; cp[u] = C::R_ZERO;
    if (e < NV * NV) {
      const int i = e / NV, j = e - i * NV;
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      if (hi < NQ) {
        const int s = lo * NQ - lo * (lo - 1) / 2 + (hi - lo);
        qp[u] = C::R_Q + s;
        if constexpr (C::CURV || C::DDCURV) cp[u] = C::R_C + s;
      } else if (lo == hi) {
        qp[u] = C::R_DG + (lo - NQ);
      } else if (NS > 0 && lo == NX) {
        qp[u] = C::R_CS + hi;
      } else if (NS > 0 && hi == NX) {
        qp[u] = C::R_CS + lo;
      }
      if constexpr (C::DDCURV) {
        // curvature of the unicycle's dynamics outside the q block: variables theta (2), omega (7), u1 | v (6), u0
        auto cls = [](int j) __attribute__((always_inline)) {   // 0 theta, 1 omega, 2 u1, 3 v, 4 u0, -1 none
          return j == 2 ? 0 : (j == 7 ? 1 : (j == NX + NS + 1 ? 2 : (j == 6 ? 3 : (j == NX + NS ? 4 : -1))));
        };
        const int ci = cls(i), cj = cls(j);
        if (ci >= 0 && cj >= 0 && !(ci == 0 && cj == 0)) {
          const int a = ci < cj ? ci : cj, b = ci < cj ? cj : ci;
          // (a, b): alpha-alpha pairs (0,1) (0,2) (1,1) (1,2) (2,2) -> 0 .. 4; alpha-beta pairs (a, 3 + t) -> 5 + 2 a + t
          int idx = -1;
          if (b <= 2) idx = a == 0 ? b - 1 : (a == 1 ? 1 + b : 4);
          else if (a <= 2) idx = 5 + 2 * a + (b - 3);
          if (idx >= 0) cp[u] = C::R_D + idx;
        }
      }
    }
  }
  // [A|B] of the diff-drive model: identity outside the reduced (x, y, theta, v, omega) block
  int abp[DD ? TPL : 1];
  double abc[DD ? TPL : 1];
  if constexpr (DD) {
#pragma unroll
    for (int u = 0; u < TPL; u++) {
      const int e = lane + LPI * u;
      abp[u] = C::R_ZERO; abc[u] = 0.0;
      if (e < NX * NV) {
        const int i = e / NV, j = e - i * NV;
        const int ri = (i < 3) ? i : (i >= 6 ? i - 3 : -1);
        if (j < NX) {
          const int rj = (j < 3) ? j : (j >= 6 ? j - 3 : -1);
          if (ri >= 0 && rj >= 0) abp[u] = C::R_A5 + (ri * 5 + rj);
          else abc[u] = (i == j) ? 1.0 : 0.0;
        } else if (j >= NX + NS && ri >= 0) {
          abp[u] = C::R_B5 + (ri * 2 + (j - NX - NS));
        }
      }
    }
  }
  // [A|B] of the stage whose record is in srec
  auto fill_AB_dd = [&]() __attribute__((always_inline)) {
    if constexpr (DD) {
#pragma unroll
      for (int u = 0; u < TPL; u++) {
        const int e = lane + LPI * u;
        const double v = srec[abp[u]] + abc[u];
        if (e < NX * NV) sAB[e] = v;
      }
    }
  };

  for (int e = lane; e < NX * NX; e += LPI) sP[e] = 0.0;
  if (lane < NX) sp[lane] = 0.0;
  bool chol_ok = true;

  // ---- fused kernel, holonomic chain: the backward pass on the instance's LDS slots ---------------------
  // Same arithmetic per entry as the generic path below, organised for a wavefront that runs alone on its
  // SIMD: the record is read where the sweep left it (slot k), the image of the stage is written where the
  // forward pass will read it (slot k: the record is dead by then), every phase issues all its LDS reads
  // before the first use (one wait per phase instead of one per entry), nothing is predicated except the
  // stores, and the cost-to-go products of a lane's q entry are formed by the lane itself instead of going
  // through another LDS exchange: three waits per stage instead of about twenty.
  // ---- fused kernel, holonomic chain without slack, n <= 3 (the point robot): Schur-complement form on the slots --------
  // Round 4.  At four wavefronts per CU the recursion of k_fused is bound by the LDS: the path below it (kept for the
  // chains with the slack variable) reads 78 doubles per lane and backward stage and 24 per forward stage; this one
  // reads 37 and 17 -- the block form of the arms' path (riccati_recursion's ARMB) at half-wavefront width:
  //   A  lane (i, j) of an n x n grid (8-lane groups) reads S, T, T', V of the cost-to-go once and forms the seven
  //      block entries of [A|B]^T P [A|B]; its eight record entries come from the stage's slot one stage ahead;
  //      g = P rc + p by DPP sums over the group, lanes j = 0, 1, 2 finish the gradient entries q_i, v_i, u_i;
  //   B  Cholesky of Quu in every lane, Y = L^-1 [Qux | qu] (a column per lane), the gains K = -L^-T Y behind it;
  //   C  [P | p] = [Qxx | qx] - Y^T [Y | y]: an entry per lane and turn, (i, j) and (j, i) the same products in the
  //      same order (Qxx is formed symmetrically): symmetric without the 0.5 (a + a^T) of the gain form.
  // The rollout forms dw, nu+ and dx+ from dx alone with one role per lane (one row of the image per lane).
#ifdef RMPC_NO_FASTB
  constexpr bool FASTB = false;
#else
  constexpr bool FASTB = SLOTS && !DD && NS == 0 && NQ <= 3 && LPI == 32;
#endif
  if constexpr (FASTB) {
    constexpr int n = NQ;
    constexpr int OFF_KFF = NW * NX, OFF_PT = OFF_KFF + NW, OFF_P = OFF_PT + NP2, OFF_RC = OFF_P + NX;
    static_assert(NW == n && NX == 2 * n && NX + 1 <= 8, "point-robot path: holonomic chain without slack, n <= 3");
    static_assert(20 * NW + 48 <= RicLds<C, LPI>::LDSW, "point-robot path: work area");
    // work area: [Qux | qu] (NW rows of 8) | Quu (NW rows of 4) | Y (NW rows of 8) | dx (2 x 8) | a word per idle lane
    ldouble *const aQux = img, *const aQuu = aQux + 8 * NW, *const aY = aQuu + 4 * NW, *const adx = aY + 8 * NW,
                 *const adum = adx + 16;
    ldouble *const dummy = adum + lane;
    // -- lane (gi, gj), block position (ii, jj): the lane keeps its entries S, T, T', V of the cost-to-go (and p_i in
    //    lanes gj = 0, p_{n+i} in lanes gj = 1) in registers from stage to stage -----------------------------------------
    const int gi = lane >> 3, gj = lane & 7;
    const bool gval = gi < n, gon = gval && gj < n;
    const int ii = gval ? gi : 0, jj = gj < n ? gj : 0;
    const bool gdiag = gon && ii == jj;
    const int qlo = ii < jj ? ii : jj, qhi = ii < jj ? jj : ii;
    const int tq = qlo * n - qlo * (qlo - 1) / 2 + (qhi - qlo);
    const int jme = gj == 0 ? ii : (gj == 1 ? n + ii : (gj == 2 ? 2 * n + ii : 0));   // gradient entry of lanes gj <= 2
    const double gc1 = gj == 0 ? 1.0 : (gj == 1 ? h : h2), gc2 = gj == 0 ? 0.0 : (gj == 1 ? 1.0 : h);
    int ro[8];   // record entries of this lane
    ro[0] = C::R_Q + tq; ro[1] = C::R_C + tq; ro[2] = C::R_DG + ii; ro[3] = C::R_DG + n + ii;
    ro[4] = C::R_RC + jj; ro[5] = C::R_RC + n + jj; ro[6] = C::R_Q0 + jme; ro[7] = C::R_Q1 + jme;
    ldouble *const dUq = gon ? aQux + ii * 8 + jj : dummy, *const dUv = gon ? aQux + ii * 8 + n + jj : dummy,
                 *const dUu = gon ? aQuu + ii * 4 + jj : dummy;
    ldouble *const dqu = (gval && gj == 2) ? aQux + ii * 8 + NX : dummy;   // gradient of u_i (q_i, v_i stay in registers)
    const bool rcw = lane < n;   // lanes (0, jj) put the defect of the stage into the image
    const int myrow = gj == 1 ? n + ii : ii;   // row of [P | p] whose p entry this lane forms (lanes gj = 0, 1)
    // where the lane's entries of the new cost-to-go go in the image of the stage (packed upper triangle; p)
    const int sS = (gon && ii <= jj) ? OFF_PT + tri(ii, jj) : -1, sT = gon ? OFF_PT + tri(ii, n + jj) : -1,
              sV = (gon && ii <= jj) ? OFF_PT + tri(n + ii, n + jj) : -1, sp_ = (gval && gj <= 1) ? OFF_P + myrow : -1;
    // -- phase B: gain column of this lane (NX: the gradient column) ---------------------------------------------------
    const int bc = lane <= NX ? lane : 0;
    ldouble *const dY = lane <= NX ? aY + bc : dummy;
    const int ystr = lane <= NX ? 8 : 0;
    const int koff = lane < NX ? lane : (lane == NX ? OFF_KFF : -1), kstr = lane < NX ? NX : (lane == NX ? 1 : 0);
    double rn[8];
#pragma unroll
    for (int u = 0; u < 8; u++) rn[u] = slots[(size_t)(N - 1) * GS + ro[u]];
    double S = 0.0, T = 0.0, U = 0.0, V = 0.0, p1 = 0.0, p2 = 0.0;   // P = 0, p = 0 behind the last stage
    WSYNC();
    RST_DECL();
    for (int k = N - 1; k >= 0; k--) {
      ldouble *const slot = slots + (size_t)k * GS;   // record of stage k (in registers by now); becomes its image
      double rc_[8];
#pragma unroll
      for (int u = 0; u < 8; u++) rc_[u] = rn[u];
      {
        const int kn = k > 0 ? k - 1 : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) rn[u] = slots[(size_t)kn * GS + ro[u]];   // arrives while this stage is computed
      }
      // ---- phase A: the blocks of [A|B]^T P [A|B] at (ii, jj); Qxx stays in registers --------------------------------
      const double rcj = gj < n ? rc_[4] : 0.0, rcnj = gj < n ? rc_[5] : 0.0;
      const double tv = h * S + T, tu = h2 * S + h * T, bv = h * U + V, bu = h2 * U + h * V;
      const double qq = S + (rc_[0] - cwt * rc_[1]);
      const double vq = h * S + U;
      const double vv = (h * (h * S + (T + U)) + V) + (gdiag ? rc_[2] : 0.0);
      const double uq = h2 * S + h * U, uv = h2 * tv + h * bv;
      const double uu = (h2 * tu + h * bu) + (gdiag ? rc_[3] : 0.0);
      *dUq = uq; *dUv = uv; *dUu = uu;
      // g = P rc + p: the group's partial products, summed over its lanes (gj < n <= 3: one quad)
      const double g1 = p1 + dpp_sum4(S * rcj + T * rcnj), g2 = p2 + dpp_sum4(U * rcj + V * rcnj);
      const double gme = (rc_[6] - mu * rc_[7]) + (gc1 * g1 + gc2 * g2);   // gradient entry q_i / v_i / u_i (gj = 0, 1, 2)
      *dqu = gme;
      *(rcw ? slot + OFF_RC + lane : dummy) = rc_[4];
      *(rcw ? slot + OFF_RC + n + lane : dummy) = rc_[5];
      WSYNC();
      RST(1);
      // ---- phase B: Cholesky of Quu (every lane), Y = L^-1 [Qux | qu] (one column per lane) ------------------------------
      {
        double qw[NW][NW], colv[NW];
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
          for (int i = j; i < NW; i++) qw[i][j] = aQuu[i * 4 + j];
#pragma unroll
        for (int i = 0; i < NW; i++) colv[i] = aQux[i * 8 + bc];
        __builtin_amdgcn_sched_barrier(0);
        double L[NW][NW], invd[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
          double dg = qw[j][j];
#pragma unroll
          for (int l = 0; l < j; l++) dg -= L[j][l] * L[j][l];
          if (!(dg > 0.0)) chol_ok = false;
          double inv = __builtin_amdgcn_rsq(dg);
          inv = inv * (1.5 - 0.5 * dg * inv * inv);
          inv = inv * (1.5 - 0.5 * dg * inv * inv);
          L[j][j] = dg * inv;
          invd[j] = inv;
#pragma unroll
          for (int i = j + 1; i < NW; i++) {
            double sacc = qw[i][j];
#pragma unroll
            for (int l = 0; l < j; l++) sacc -= L[i][l] * L[j][l];
            L[i][j] = sacc * inv;
          }
        }
        double y[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) {
          double sacc = colv[i];
#pragma unroll
          for (int l = 0; l < i; l++) sacc -= L[i][l] * y[l];
          y[i] = sacc * invd[i];
          dY[i * ystr] = y[i];
        }
        WSYNC();
        RST(3);
        // ---- phase C: [P | p] = [Qxx | qx] - Y^T [Y | y] at the lane's own block position: the next stage's phase A
        //      starts from registers (no store of P, no ordering point, no read-back) ---------------------------------------
        double yiq[NW], yiv[NW], yjq[NW], yjv[NW], yg[NW];
#pragma unroll
        for (int l = 0; l < NW; l++) {
          yiq[l] = aY[l * 8 + ii]; yiv[l] = aY[l * 8 + n + ii]; yjq[l] = aY[l * 8 + jj]; yjv[l] = aY[l * 8 + n + jj];
          yg[l] = aY[l * 8 + NX];
        }
        __builtin_amdgcn_sched_barrier(0);
        // (while the operands arrive: the gains K = -L^-T Y of this lane's column, for the rollout only)
        double x[NW];
#pragma unroll
        for (int i = NW - 1; i >= 0; i--) {
          double sacc = y[i];
#pragma unroll
          for (int l = i + 1; l < NW; l++) sacc -= L[l][i] * x[l];
          x[i] = sacc * invd[i];
        }
        ldouble *const kd = koff >= 0 ? slot + koff : dummy;
#pragma unroll
        for (int i = 0; i < NW; i++) kd[i * kstr] = -x[i];
        double sn = qq, tn_ = tv, un = vq, vn = vv, pn = gme;
#pragma unroll
        for (int l = 0; l < NW; l++) {
          sn -= yiq[l] * yjq[l]; tn_ -= yiq[l] * yjv[l]; un -= yiv[l] * yjq[l]; vn -= yiv[l] * yjv[l];
          pn -= (gj == 1 ? yiv[l] : yiq[l]) * yg[l];
        }
        S = sn; T = tn_; U = un; V = vn;
        p1 = dpp_move<0x00>(pn);   // quad_perm [0, 0, 0, 0]: p_i from lane gj = 0 of the quad
        p2 = dpp_move<0x55>(pn);   // quad_perm [1, 1, 1, 1]: p_{n+i} from lane gj = 1
        *(sS >= 0 ? slot + sS : dummy) = sn;
        *(sT >= 0 ? slot + sT : dummy) = tn_;
        *(sV >= 0 ? slot + sV : dummy) = vn;
        *(sp_ >= 0 ? slot + sp_ : dummy) = pn;
      }
      RST(4);
    }
    WSYNC();
    if (!chol_ok) return false;
    // ---- rollout: dw = kff + K dx, nu+ = p + P dx, dx+ = rc + [A|B][dx; dw], one ordering point per stage ---------------
    const bool fA = lane < NW, fB = lane >= NW && lane < NW + NX, fC = lane >= NW + NX && lane < NW + 2 * NX;
    const int fi = fA ? lane : (fB ? lane - NW : (fC ? lane - NW - NX : 0));   // entry of dw / nu+ / dx+
    const int fw = fC ? (fi < n ? fi : fi - n) : fi;                             // the entry of dw a dx+ lane needs
    const int foff = fB ? OFF_P + fi : OFF_KFF + fw;
    int frow[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) frow[j] = fB ? OFF_PT + tri(fi, j) : fw * NX + j;
    const int fx1 = fi < n ? n + fi : fi;
    const double fca = fi < n ? h : 0.0, fcb = fi < n ? h2 : h;
    const int dzslot = fA ? NX + lane : fi;
    // (dx through the crossbar -- ds_bpermute, no store / ordering point / read-back, the image rows requested a stage
    //  ahead -- was measured slower: 16 crossbar instructions per stage cost more LDS issue than the exchange saves,
    //  2.85 -> 2.75 M solves/s on cfg2 with four batches in flight)
    if (lane < 16) adx[lane] = 0.0;
    for (int k = 0; k < N; k++) {
      const ldouble *const im = slots + (size_t)k * GS;
      const ldouble *const dxc = adx + 8 * (k & 1);
      ldouble *const dxn = adx + 8 * ((k & 1) ^ 1);
      // (the lane's row of the image does not depend on dx: requested before the ordering point, it arrives with it)
      double rowv[NX];
#pragma unroll
      for (int j = 0; j < NX; j++) rowv[j] = im[frow[j]];
      double sacc = im[foff];
      const double rcv = im[OFF_RC + fi];
      WSYNC();
      double dxv[NX];
#pragma unroll
      for (int j = 0; j < NX; j++) dxv[j] = dxc[j];
      const double d0 = dxc[fi], d1 = dxc[fx1];
      __builtin_amdgcn_sched_barrier(0);   // (the image of the stage is read before its step is stored over it)
#pragma unroll
      for (int j = 0; j < NX; j++) sacc += rowv[j] * dxv[j];
      *((fA || fC) ? so.dz + dzslot + (size_t)k * GS : dummy) = fA ? sacc : d0;
      *((fB && k >= 1) ? so.nunew + fi + (size_t)k * GS : dummy) = sacc;
      double sx = rcv;
      sx += d0;
      sx += fca * d1;
      sx += fcb * sacc;
      *((fC && k < N - 1) ? dxn + fi : dummy) = sx;
      RST(6);
    }
    RST_FLUSH();
    return true;
  }
  constexpr bool FAST = SLOTS && !DD && !FASTB;
  // the arms' cost-to-go update on the matrix cores (v_mfma_f64_16x16x4_f64): one wavefront per instance, a state of
  // 9 .. 15 entries (one tile with the gradient column), at most 8 inputs (two k-steps)
  constexpr bool MFMA_P = !DD && !SLOTS && LPI == 64 && NX > 8 && NX < 16 && NW <= 8;
  if constexpr (FAST) {
    constexpr int OFF_KFF = NW * NX, OFF_PT = NW * NX + NW, OFF_P = OFF_PT + NP2, OFF_RC = OFF_P + NX;
    // loop-invariant per-lane constants of the Q entries (sum_{a,b} l_a c_b P_ab, see the generic path)
    int o11[EPL];
    double l1[EPL], l2[EPL], c1[EPL], c2[EPL];
    bool qok[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) {
      const int e = lane + LPI * u;
      qok[u] = e < NV * NV;
      const int ec = qok[u] ? e : 0;
      const int i = ec / NV, j = ec - i * NV;
      const int ki = i < NQ ? 0 : (i < NX ? 1 : (i >= NX + NS ? 2 : 3));
      const int kj = j < NQ ? 0 : (j < NX ? 1 : (j >= NX + NS ? 2 : 3));
      const bool on = ki != 3 && kj != 3;
      const int ii = !on ? 0 : (ki == 0 ? i : (ki == 1 ? i - NQ : i - NX - NS));
      const int jj = !on ? 0 : (kj == 0 ? j : (kj == 1 ? j - NQ : j - NX - NS));
      o11[u] = ii * NX + jj;
      l1[u] = !on ? 0.0 : (ki == 0 ? 1.0 : (ki == 1 ? h : h2)); l2[u] = !on ? 0.0 : (ki == 0 ? 0.0 : (ki == 1 ? 1.0 : h));
      c1[u] = kj == 0 ? 1.0 : (kj == 1 ? h : h2); c2[u] = kj == 0 ? 0.0 : (kj == 1 ? 1.0 : h);
    }
    // the lane's gradient entry (lanes < NV)
    const int lv = lane < NV ? lane : 0;
    const int kq = lv < NQ ? 0 : (lv < NX ? 1 : (lv >= NX + NS ? 2 : 3));
    const int iq = kq == 3 ? 0 : (kq == 0 ? lv : (kq == 1 ? lv - NQ : lv - NX - NS));
    const double l1q = kq == 3 ? 0.0 : (kq == 0 ? 1.0 : (kq == 1 ? h : h2)), l2q = kq == 3 ? 0.0 : (kq == 0 ? 0.0 : (kq == 1 ? 1.0 : h));
    const int lr = lane < NX ? lane : 0;
    // gains: lane c <= NX solves for column c of K (c < NX) or for kff (c == NX); sq follows sQ in the work area
    const int lc = lane <= NX ? lane : 0;
    // Every store of a phase is unconditional: a lane without an entry writes to a word of its own in the staging
    // area of the generic path (srec, unused here) instead of skipping the store -- a predicated store makes the
    // compiler cut the phase into exec-masked blocks with their own waits (15 of them per stage before).
    ldouble *const dummy = srec + lane;
    ldouble *qdst[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) qdst[u] = (lane + LPI * u < NV * NV) ? sQ + lane + LPI * u : dummy;
    ldouble *const sqdst = lane < NV ? sq + lane : dummy;
    // cost-to-go entries of this lane (see the generic path): P(i, j) for e < NX*NX, then p(i)
    constexpr int PPL2 = (NX * NX + NX + LPI - 1) / LPI;
    int pa0[PPL2], pc0[PPL2], pqa[PPL2], pqc[PPL2], pka[PPL2], pkc[PPL2], pks[PPL2], pslot[PPL2];
    ldouble *pdst1[PPL2];
    bool pisP[PPL2], pok[PPL2];
#pragma unroll
    for (int u = 0; u < PPL2; u++) {
      const int e = lane + LPI * u;
      pok[u] = e < NX * NX + NX;
      const int ec = pok[u] ? e : 0;
      const bool isP = ec < NX * NX;
      pisP[u] = isP;
      const int i = isP ? ec / NX : ec - NX * NX, j = isP ? ec - i * NX : 0;
      // offsets relative to sQ (sq = sQ + NV*NV) and to the slot (K at 0, kff at OFF_KFF)
      pa0[u] = isP ? i * NV + j : NV * NV + i;
      pc0[u] = isP ? j * NV + i : NV * NV + i;
      pqa[u] = i * NV + NX;                          // sQ[i][NX + l]
      pqc[u] = isP ? j * NV + NX : i * NV + NX;      // cq[l]
      pka[u] = isP ? j : OFF_KFF;                    // K[l][j] = slot[l*NX + j]  |  kff[l] = slot[OFF_KFF + l]
      pkc[u] = isP ? i : OFF_KFF;
      pks[u] = isP ? NX : 1;                         // stride over l
      // where the entry goes: work area (sP / sp) and the stage's image (upper triangle of P packed, p); -1: nowhere
      pdst1[u] = !pok[u] ? dummy : (isP ? sP + ec : sp + (ec - NX * NX));
      pslot[u] = !pok[u] ? -1 : (isP ? (i <= j ? OFF_PT + tri(i, j) : -1) : OFF_P + (ec - NX * NX));
    }
    for (int k = N - 1; k >= 0; k--) {
      ldouble *const slot = slots + (size_t)k * GS;   // record of stage k; becomes its image [K | kff | Pt | p | rc]
      // ---- phase A: stage Hessian and gradient, with [A|B]^T P [A|B] and [A|B]^T (P rc + p) in closed form ----
      double r0[EPL], r1[EPL], a11[EPL], a12[EPL], a21[EPL], a22[EPL];
#pragma unroll
      for (int u = 0; u < EPL; u++) {
        r0[u] = slot[qp[u]]; r1[u] = slot[cp[u]];
        a11[u] = sP[o11[u]]; a12[u] = sP[o11[u] + NQ]; a21[u] = sP[o11[u] + NQ * NX]; a22[u] = sP[o11[u] + NQ * NX + NQ];
      }
      double rcl[NX], pr1[NX], pr2[NX];
#pragma unroll
      for (int l = 0; l < NX; l++) { rcl[l] = slot[C::R_RC + l]; pr1[l] = sP[iq * NX + l]; pr2[l] = sP[(NQ + iq) * NX + l]; }
      double pc1 = sp[iq], pc2 = sp[NQ + iq];
      const double q0v = slot[C::R_Q0 + lv], q1v = slot[C::R_Q1 + lv], rcme = slot[C::R_RC + lr];
      __builtin_amdgcn_sched_barrier(0);   // (every read of the phase is issued before the first use: one counted wait instead of a wait per use)
#pragma unroll
      for (int u = 0; u < EPL; u++) {
        double v = r0[u] - cwt * r1[u];
        v += l1[u] * (c1[u] * a11[u] + c2[u] * a12[u]) + l2[u] * (c1[u] * a21[u] + c2[u] * a22[u]);
        *qdst[u] = v;
      }
#pragma unroll
      for (int l = 0; l < NX; l++) { pc1 += pr1[l] * rcl[l]; pc2 += pr2[l] * rcl[l]; }
      {
        double v = q0v - mu * q1v;
        v += l1q * pc1 + l2q * pc2;
        *sqdst = v;
      }
      *(lane < NX ? slot + OFF_RC + lane : dummy) = rcme;   // (behind the record: [OFF_RC, OFF_RC + NX) is step space, dead now)
      WSYNC();
      // ---- phase B: Cholesky of Qww (every lane, registers) and the gains (one column per lane) -------------
      double qw[NW][NW], colv[NW];
#pragma unroll
      for (int j = 0; j < NW; j++)
#pragma unroll
        for (int i = j; i < NW; i++) qw[i][j] = sQ[(NX + i) * NV + NX + j];
#pragma unroll
      for (int i = 0; i < NW; i++) colv[i] = sQ[lc < NX ? (NX + i) * NV + lc : NV * NV + NX + i];
      __builtin_amdgcn_sched_barrier(0);   // (every read of the phase is issued before the first use: one counted wait instead of a wait per use)
      double L[NW][NW], invd[NW];
#pragma unroll
      for (int j = 0; j < NW; j++) {
        double dg = qw[j][j];
#pragma unroll
        for (int l = 0; l < j; l++) dg -= L[j][l] * L[j][l];
        if (!(dg > 0.0)) chol_ok = false;
        double inv = __builtin_amdgcn_rsq(dg);
        inv = inv * (1.5 - 0.5 * dg * inv * inv);
        inv = inv * (1.5 - 0.5 * dg * inv * inv);
        L[j][j] = dg * inv;
        invd[j] = inv;
#pragma unroll
        for (int i = j + 1; i < NW; i++) {
          double sacc = qw[i][j];
#pragma unroll
          for (int l = 0; l < j; l++) sacc -= L[i][l] * L[j][l];
          L[i][j] = sacc * inv;
        }
      }
      {
        double col[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) col[i] = -colv[i];
        chol_solve<NW>(L, invd, col);
        {
          ldouble *const kdst = lane <= NX ? slot + (lane < NX ? lane : OFF_KFF) : dummy;
          const int kstr = lane < NX ? NX : (lane == NX ? 1 : 0);
#pragma unroll
          for (int i = 0; i < NW; i++) kdst[i * kstr] = col[i];
        }
      }
      WSYNC();
      // ---- phase C: cost-to-go P = sym(Qxx + Qxw K), p = qx + Qxw kff ---------------------------------------
      double pa[PPL2], pcc[PPL2], qa[PPL2][NW], qc[PPL2][NW], ka[PPL2][NW], kc[PPL2][NW];
#pragma unroll
      for (int u = 0; u < PPL2; u++) {
        pa[u] = sQ[pa0[u]]; pcc[u] = sQ[pc0[u]];
#pragma unroll
        for (int l = 0; l < NW; l++) {
          qa[u][l] = sQ[pqa[u] + l]; qc[u][l] = sQ[pqc[u] + l];
          ka[u][l] = slot[pka[u] + l * pks[u]]; kc[u][l] = slot[pkc[u] + l * pks[u]];
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // (every read of the phase is issued before the first use: one counted wait instead of a wait per use)
#pragma unroll
      for (int u = 0; u < PPL2; u++) {
        double a = pa[u], c = pcc[u];
#pragma unroll
        for (int l = 0; l < NW; l++) {
          a += qa[u][l] * ka[u][l];
          c += qc[u][l] * kc[u][l];
        }
        const double pn = 0.5 * (a + c);
        *pdst1[u] = pn;                                      // work area: sP / sp, read by the next stage's phase A
        *(pslot[u] >= 0 ? slot + pslot[u] : dummy) = pn;     // image of the stage: packed triangle / p
      }
      WSYNC();   // (sP / sp of this stage are read by the next stage's phase A)
    }
  }
  // prefetched record of the stage about to be processed (lane e holds entry e)
  double recv[RPL];
  auto fetch_stage = [&](int k) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < RPL; u++) {
      const int e = lane + LPI * u;
      recv[u] = rb[(size_t)k * sstr + (e < C::RS ? e : 0)];
    }
  };
  // ---- fused kernel, diff-drive: backward pass with the structure of [A | B] used ------------------------------
  // [A | B] of the unicycle is the identity outside the reduced block (x, y, theta, v, omega) = x[{0,1,2,6,7}] and its
  // two input columns: T = P [A|B] has 7 computed columns (the identity columns are columns of P, the slack column
  // is zero) and Q += [A|B]^T T has 7 computed rows, each a 5-term sum -- a third of the dense products, with the
  // terms in the dense order (zeros and ones drop out exactly), so the values are those of the generic path.  The
  // record entries a lane needs come straight from the record into its registers one stage ahead (no staging copy of
  // the record in LDS), every producer stores its part of the gain image to the gain record itself (no read-back),
  // and a stage has four ordering points instead of five.
  constexpr bool DDFAST = DD && OWNER && !SLOTS;
  if constexpr (DDFAST) {
    constexpr int NR = 5, NT = NR + 2;            // reduced states; computed columns of T (reduced + inputs)
    constexpr int OFF_KFF = NW * NX, OFF_PT = NW * NX + NW, OFF_P = OFF_PT + NP2, OFF_RC = OFF_P + NX;
    ldouble *const sA5 = sAB, *const sB5 = sAB + 25, *const sZ = sAB + 35;   // sZ: one zero word
    ldouble *const sT7 = sT;                                                   // T[l][jc], 8 x 7
    auto Rl = [](int r) __attribute__((always_inline)) { return r < 3 ? r : r + 3; };        // reduced index -> state
    auto rid = [](int j) __attribute__((always_inline)) { return j < 3 ? j : (j >= 6 && j < 8 ? j - 3 : -1); };
    // phase A: entries of T this lane forms
    constexpr int TA = (NX * NT + LPI - 1) / LPI;
    int tpo[TA], tco[TA], tcs[TA], tst[TA];
    bool tok[TA];
#pragma unroll
    for (int u = 0; u < TA; u++) {
      const int t = lane + LPI * u;
      tok[u] = t < NX * NT;
      const int tc = tok[u] ? t : 0;
      const int i = tc / NT, jc = tc - i * NT;
      tpo[u] = i * NX;                                   // row i of P
      tco[u] = jc < NR ? jc : 25 + (jc - NR);            // M[rl][jc]: A5[rl][jc] | B5[rl][jc - 5]   (offset in sAB)
      tcs[u] = jc < NR ? NR : 2;
      tst[u] = tc;
    }
    // phase B: the lane's entries of the stage Hessian and its gradient entry
    int bto[EPL], bts[EPL], bco[EPL], bcs[EPL], bio[EPL];
    bool bok[EPL], bI[EPL];
    auto colsrc = [&](int j, int &to, int &ts) __attribute__((always_inline)) {
      // T[l][j] for l = 0..7: computed column | column of P | zero
      if (j < NX) {
        if (rid(j) >= 0) { to = (int)(sT7 - img) + rid(j); ts = NT; }
        else { to = (int)(sP - img) + j; ts = NX; }
      } else if (j >= NX + NS) { to = (int)(sT7 - img) + NR + (j - NX - NS); ts = NT; }
      else { to = (int)(sZ - img); ts = 0; }
    };
    auto rowsrc = [&](int i, int &co, int &cs, bool &isI) __attribute__((always_inline)) {
      // [A|B][l][i] for l in the reduced rows: column of A5 | column of B5 | nothing
      isI = false;
      if (i < NX) {
        if (rid(i) >= 0) { co = (int)(sA5 - img) + rid(i); cs = NR; }
        else { co = (int)(sZ - img); cs = 0; isI = true; }
      } else if (i >= NX + NS) { co = (int)(sB5 - img) + (i - NX - NS); cs = 2; }
      else { co = (int)(sZ - img); cs = 0; }
    };
#pragma unroll
    for (int u = 0; u < EPL; u++) {
      const int e = lane + LPI * u;
      bok[u] = e < NV * NV;
      const int ec = bok[u] ? e : 0;
      const int i = ec / NV, j = ec - i * NV;
      colsrc(j, bto[u], bts[u]);
      rowsrc(i, bco[u], bcs[u], bI[u]);
      bio[u] = bto[u] + (i < NX ? i : 0) * bts[u];      // T[i][j] (identity rows)
    }
    const int lq = lane < NV ? lane : 0;
    int qco, qcs;
    bool qI;
    rowsrc(lq, qco, qcs, qI);
    const int lr = lane < NX ? lane : 0;
    // record entries of the stage about to be processed, one stage ahead
    double rq[EPL], rqk[EPL], q0n = 0, q1n = 0, rcn = 0, abn[2] = {0, 0};
    auto fetch_dd = [&](int k) __attribute__((always_inline)) {
      const RP *const r = rb + (size_t)k * sstr;
#pragma unroll
      for (int u = 0; u < EPL; u++) { rq[u] = r[qp[u]]; rqk[u] = r[cp[u]]; }
      q0n = r[C::R_Q0 + lq]; q1n = r[C::R_Q1 + lq]; rcn = r[C::R_RC + lr];
#pragma unroll
      for (int u = 0; u < 2; u++) abn[u] = r[C::R_A5 + (lane + LPI * u < 35 ? lane + LPI * u : 0)];
    };
    // cost-to-go entries of this lane (as in the generic path): offsets of what an entry is made of, relative to the
    // work area (sQ, sq, sK, skf all live in it), and where it goes
    constexpr int PPL2 = (NX * NX + NX + LPI - 1) / LPI;
    // (the gain record always has a spare word behind the image: kps = (KPW + 8) / 8 * 8)
    // Every store of a phase is unconditional (a lane without an entry writes to a word of its own in the unused
    // staging area, or to the spare word behind the stage's gain record) and every read of a phase is issued before
    // its first use: a stage is straight-line code with one counted wait per phase instead of two dozen exec-masked
    // blocks that each wait for their own reads (as for the chain's path above).
    ldouble *const dummy = srec + lane;
    int da0[PPL2], dc0[PPL2], dqa[PPL2], dqc[PPL2], dka[PPL2], dkc[PPL2], dks[PPL2], dkp[PPL2];
    ldouble *dd1[PPL2], *dd2[PPL2];
#pragma unroll
    for (int u = 0; u < PPL2; u++) {
      const int e = lane + LPI * u;
      const bool okp = e < NX * NX + NX;
      const int ec = okp ? e : 0;
      const bool isP = ec < NX * NX;
      const int i = isP ? ec / NX : ec - NX * NX, j = isP ? ec - i * NX : 0;
      da0[u] = isP ? (int)(sQ - img) + i * NV + j : (int)(sq - img) + i;
      dc0[u] = isP ? (int)(sQ - img) + j * NV + i : (int)(sq - img) + i;
      dqa[u] = (int)(sQ - img) + i * NV + NX;
      dqc[u] = (int)(sQ - img) + (isP ? j : i) * NV + NX;
      dka[u] = isP ? (int)(sK - img) + j : (int)(skf - img);
      dkc[u] = isP ? (int)(sK - img) + i : (int)(skf - img);
      dks[u] = isP ? NX : 1;
      dd1[u] = !okp ? dummy : (isP ? sP + ec : sp + (ec - NX * NX));
      dd2[u] = (okp && isP && i <= j) ? sPt + tri(i, j) : dummy;
      dkp[u] = !okp ? KPW : (isP ? (i <= j ? OFF_PT + tri(i, j) : KPW) : OFF_P + (ec - NX * NX));
    }
    ldouble *tdst[TA];
#pragma unroll
    for (int u = 0; u < TA; u++) tdst[u] = tok[u] ? sT7 + tst[u] : dummy;
    ldouble *qdst[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) qdst[u] = bok[u] ? sQ + lane + LPI * u : dummy;
    ldouble *const pcdst = lane < NX ? sPc + lane : dummy, *const sqdst = lane < NV ? sq + lane : dummy;
    ldouble *const srcdst = lane < NX ? src + lane : dummy;
    ldouble *abdst[2];
#pragma unroll
    for (int u = 0; u < 2; u++) abdst[u] = lane + LPI * u < 35 ? sAB + lane + LPI * u : dummy;
    const int lc = lane <= NX ? lane : 0;                           // gain column of this lane (NX: kff)
    ldouble *const kdst = lane <= NX ? img + (lane < NX ? lane : OFF_KFF) : dummy;
    const int kgo = lane <= NX ? (lane < NX ? lane : OFF_KFF) : KPW, kstr = lane < NX ? NX : (lane == NX ? 1 : 0);
    const int rco = lane < NX ? OFF_RC + lane : KPW;
    if (lane == 0) sZ[0] = 0.0;
    fetch_dd(N - 1);
    for (int k = N - 1; k >= 0; k--) {
      gdouble *const kpk = kpb + (size_t)k * kps;
      // this stage's record entries are in registers; the next one's leave now
      // (with the second-order terms of the unicycle: record entry minus the curvature entry when this step uses them)
      double rqc[EPL];
#pragma unroll
      for (int u = 0; u < EPL; u++) rqc[u] = rq[u] - cwt * rqk[u];
      const double q0c = q0n, q1c = q1n, rcc = rcn;
      // ([A5 | B5] and rc of THIS stage are in LDS since the last phase of the previous stage)
      if (k > 0) fetch_dd(k - 1);
      const bool rec_cost = k < N - 1;
      WSYNC();   // P, p of stage k+1 and [A5 | B5], rc of this stage are in LDS
      if (rec_cost) {
        // ---- phase A: T = P [A|B] (computed columns), Pc = P rc + p ------------------------------------------
        double ta[TA][NR], tb[TA][NR], pr[NX], rcl[NX];
#pragma unroll
        for (int u = 0; u < TA; u++)
#pragma unroll
          for (int r = 0; r < NR; r++) { ta[u][r] = sP[tpo[u] + Rl(r)]; tb[u][r] = sAB[tco[u] + r * tcs[u]]; }
#pragma unroll
        for (int l = 0; l < NX; l++) { pr[l] = sP[lr * NX + l]; rcl[l] = src[l]; }
        double pcv = sp[lr];
        __builtin_amdgcn_sched_barrier(0);
        double tv[TA];
#pragma unroll
        for (int u = 0; u < TA; u++) {
          double sacc = 0.0;
#pragma unroll
          for (int r = 0; r < NR; r++) sacc += ta[u][r] * tb[u][r];
          tv[u] = sacc;
        }
#pragma unroll
        for (int l = 0; l < NX; l++) pcv += pr[l] * rcl[l];
#pragma unroll
        for (int u = 0; u < TA; u++) *tdst[u] = tv[u];
        *pcdst = pcv;
        WSYNC();
      }
      // ---- phase B: Q = record + [A|B]^T T, q = q0 - mu q1 + [A|B]^T Pc ----------------------------------------
      {
        double qv[EPL];
        double gq = q0c - mu * q1c;
        if (rec_cost) {
          double ba[EPL][NR], bb[EPL][NR], bi[EPL], ga[NR], gb[NR];
#pragma unroll
          for (int u = 0; u < EPL; u++) {
#pragma unroll
            for (int r = 0; r < NR; r++) { ba[u][r] = img[bco[u] + r * bcs[u]]; bb[u][r] = img[bto[u] + Rl(r) * bts[u]]; }
            bi[u] = img[bio[u]];
          }
#pragma unroll
          for (int r = 0; r < NR; r++) { ga[r] = img[qco + r * qcs]; gb[r] = sPc[Rl(r)]; }
          const double gi = sPc[lq < NX ? lq : 0];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < EPL; u++) {
            double v = rqc[u];
#pragma unroll
            for (int r = 0; r < NR; r++) v += ba[u][r] * bb[u][r];
            v += bI[u] ? bi[u] : 0.0;
            qv[u] = v;
          }
#pragma unroll
          for (int r = 0; r < NR; r++) gq += ga[r] * gb[r];
          gq += qI ? gi : 0.0;
        } else {
#pragma unroll
          for (int u = 0; u < EPL; u++) qv[u] = rqc[u];
        }
#pragma unroll
        for (int u = 0; u < EPL; u++) *qdst[u] = qv[u];
        *sqdst = gq;
        kpk[rco] = rcc;   // the stage's defect: part of its gain image
      }
      WSYNC();
      // ---- phase C: Cholesky of Qww (every lane, registers) and the gains (one column per lane) ----------------
      {
        double qw[NW][NW], colv[NW];
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
          for (int i = j; i < NW; i++) qw[i][j] = sQ[(NX + i) * NV + NX + j];
#pragma unroll
        for (int i = 0; i < NW; i++) colv[i] = lc < NX ? sQ[(NX + i) * NV + lc] : sq[NX + i];
        __builtin_amdgcn_sched_barrier(0);
        double L[NW][NW], invd[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
          double dg = qw[j][j];
#pragma unroll
          for (int l = 0; l < j; l++) dg -= L[j][l] * L[j][l];
          if (!(dg > 0.0)) chol_ok = false;
          double inv = __builtin_amdgcn_rsq(dg);
          inv = inv * (1.5 - 0.5 * dg * inv * inv);
          inv = inv * (1.5 - 0.5 * dg * inv * inv);
          L[j][j] = dg * inv;
          invd[j] = inv;
#pragma unroll
          for (int i = j + 1; i < NW; i++) {
            double sacc = qw[i][j];
#pragma unroll
            for (int l = 0; l < j; l++) sacc -= L[i][l] * L[j][l];
            L[i][j] = sacc * inv;
          }
        }
        double col[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) col[i] = -colv[i];
        chol_solve<NW>(L, invd, col);
#pragma unroll
        for (int i = 0; i < NW; i++) {
          kdst[i * kstr] = col[i];
          kpk[kgo + i * kstr] = col[i];
        }
      }
      WSYNC();
      // ---- phase D: cost-to-go P = sym(Qxx + Qxw K), p = qx + Qxw kff; [A5 | B5], rc of the next stage to LDS -----
      {
        double a0[PPL2], c0[PPL2], qa[PPL2][NW], qc[PPL2][NW], ka[PPL2][NW], kc[PPL2][NW];
#pragma unroll
        for (int u = 0; u < PPL2; u++) {
          a0[u] = img[da0[u]]; c0[u] = img[dc0[u]];
#pragma unroll
          for (int l = 0; l < NW; l++) {
            qa[u][l] = img[dqa[u] + l]; qc[u][l] = img[dqc[u] + l];
            ka[u][l] = img[dka[u] + l * dks[u]]; kc[u][l] = img[dkc[u] + l * dks[u]];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PPL2; u++) {
          double a = a0[u], c = c0[u];
#pragma unroll
          for (int l = 0; l < NW; l++) {
            a += qa[u][l] * ka[u][l];
            c += qc[u][l] * kc[u][l];
          }
          const double pn = 0.5 * (a + c);
          *dd1[u] = pn;
          *dd2[u] = pn;
          kpk[dkp[u]] = pn;
        }
        // (what fetch_dd(k - 1) brought: nobody reads sAB / src again before the ordering point at the loop top;
        //  single-stage horizon: the forward pass reads the defect of stage 0 from the image)
#pragma unroll
        for (int u = 0; u < 2; u++) *(k > 0 ? abdst[u] : dummy) = abn[u];
        *srcdst = k > 0 ? rcn : rcc;
      }
    }
  }
  // ---- the arms (pass kernels, one wavefront per instance, n = 5 .. 7 without slack): Schur-complement form ---------
  // Round 4.  Stamps of the generic path on 1024 arms: stage Hessian 1.46 k, Cholesky + gains 1.44 k, cost-to-go 1.74 k
  // cycles per backward stage, rollout 2.4 k per forward stage -- 134 LDS reads per lane and stage, five ordering
  // points.  This path:
  //   A  lane (i, j) of an n x n grid (8-lane groups) reads the four blocks S, T, T', V of the cost-to-go at (i, j) ONCE
  //      and forms the seven block entries of [A|B]^T P [A|B] that are needed -- qq, qv, vq, vv in place over P, uq, uv,
  //      uu for the control block -- instead of every dense entry fetching its four; its record entries come straight
  //      from the stage record in global memory, requested one stage ahead (the record never passes through LDS);
  //      g = P rc + p is summed over the 8 lanes of a group by DPP moves, and lanes j = 0, 1, 2 of group i finish
  //      the gradient entries q_i, v_i, u_i: no separate gradient phase;
  //   B  Cholesky of Quu in every lane (as before), then only the FORWARD substitution Y = L^-1 [Qux | qu] is on the
  //      way to the next stage; the backward substitution that yields the gains K | kff goes to the image behind it;
  //   C  P = Qxx - Y^T Y, p = qx - Y^T y on the matrix cores: two v_mfma_f64_16x16x4_f64 with the SAME register as
  //      both operands (A = -Y^T, B = Y), accumulated onto [Qxx | qx]; the products of (i, j) and (j, i) are the same
  //      numbers in the same order, and Qxx is formed symmetrically: the result is symmetric without the 0.5 (a + a^T).
  // Three ordering points per backward stage, ~55 LDS reads per lane; the image of a stage is written where the
  // rollout reads it.  The rollout forms dw, nu+ and dx+ from dx alone (one ordering point per stage: the lane of an
  // entry of dx+ computes the entry of dw it needs itself), as the fused chain path does.
  if constexpr (RicLds<C, LPI>::ARMB && !SLOTS) {
    constexpr int n = NQ, PS = RicLds<C, LPI>::APS, QS = 16;
    constexpr int OFF_KFF = NW * NX, OFF_PT = OFF_KFF + NW, OFF_P = OFF_PT + NP2, OFF_RC = OFF_P + NX;
    static_assert(NW == n && NX == 2 * n && NX + 1 <= 16 && PS >= 16, "arm path: holonomic chain without slack, one MFMA tile");
    ldouble *const aP = img + KPW, *const aQux = aP + PS * NX, *const aQuu = aQux + QS * NW, *const aY = aQuu + 8 * NW,
                 *const ap = aY + 128, *const adx = ap + 16, *const adum = adx + 32;
    ldouble *const dummy = adum + lane;
    const int LCAPr = lcap_rt >= 0 ? lcap_rt : LCAP;   // stages 1 .. LCAPr keep their image in LDS
    for (int e = lane; e < PS * NX; e += LPI) aP[e] = 0.0;   // P = 0 behind the last stage (columns NX, NX + 1: gradient / unused)
    aY[lane] = 0.0; aY[64 + lane] = 0.0;                     // (row 7 and column 15 of the operand tile stay zero)
    if (lane < 16) ap[lane] = 0.0;
    // -- lane (gi, gj): block position (ii, jj) ----------------------------------------------------------------
    const int gi = lane >> 3, gj = lane & 7;
    const bool gval = gi < n, gon = gval && gj < n;
    const int ii = gval ? gi : 0, jj = gj < n ? gj : 0;
    const bool gdiag = gon && ii == jj;
    const int qlo = ii < jj ? ii : jj, qhi = ii < jj ? jj : ii;
    const int tq = qlo * n - qlo * (qlo - 1) / 2 + (qhi - qlo);
    const int jme = gj == 0 ? ii : (gj == 1 ? n + ii : (gj == 2 ? 2 * n + ii : 0));   // gradient entry of lanes gj <= 2
    const double gc1 = gj == 0 ? 1.0 : (gj == 1 ? h : h2), gc2 = gj == 0 ? 0.0 : (gj == 1 ? 1.0 : h);
    int ro[8];   // record entries of this lane
    ro[0] = C::R_Q + tq; ro[1] = C::R_C + tq; ro[2] = C::R_DG + ii; ro[3] = C::R_DG + n + ii;
    ro[4] = C::R_RC + jj; ro[5] = C::R_RC + n + jj; ro[6] = C::R_Q0 + jme; ro[7] = C::R_Q1 + jme;
    const int oS = ii * PS + jj, oT = oS + n, oU = (n + ii) * PS + jj, oV = oU + n;
    ldouble *const dS = gon ? aP + oS : dummy, *const dT = gon ? aP + oT : dummy, *const dU = gon ? aP + oU : dummy,
                 *const dV = gon ? aP + oV : dummy;
    ldouble *const dUq = gon ? aQux + ii * QS + jj : dummy, *const dUv = gon ? aQux + ii * QS + n + jj : dummy,
                 *const dUu = gon ? aQuu + ii * 8 + jj : dummy;
    ldouble *const dq = (gval && gj <= 2) ? (gj == 2 ? aQux + ii * QS + NX : aP + (gj == 1 ? n + ii : ii) * PS + NX) : dummy;
    const bool rcw = lane < n;   // lanes (0, jj) put the defect of the stage into the image
    // -- phase B: gain column of this lane (NX: the gradient column) -------------------------------------------
    const int bc = lane <= NX ? lane : 0;
    const bool bon = lane <= NX;
    ldouble *const dY = bon ? aY + bc : dummy;
    const int ystr = bon ? 16 : 0;
    const int koff = lane < NX ? lane : (lane == NX ? OFF_KFF : -1), kstr = lane < NX ? NX : (lane == NX ? 1 : 0);
    // -- phase C: tile position of this lane -------------------------------------------------------------------
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int c16 = lane & 15, kq = lane >> 4;
    int co[4], po2[4];
    bool cin[4];
    ldouble *cd1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = kq + 4 * r;
      const bool rin = row < NX;
      cin[r] = rin;
      co[r] = (rin ? row : 0) * PS + c16;
      cd1[r] = !rin ? dummy : (c16 < NX ? aP + row * PS + c16 : (c16 == NX ? ap + row : dummy));
      po2[r] = !rin ? -1 : (c16 < NX ? (row <= c16 ? OFF_PT + tri(row, c16) : -1) : (c16 == NX ? OFF_P + row : -1));
    }
    // record entries of the stage about to be processed, requested one stage ahead
    double rn[8];
#pragma unroll
    for (int u = 0; u < 8; u++) rn[u] = rb[(size_t)(N - 1) * sstr + ro[u]];
    RST_DECL();
    for (int k = N - 1; k >= 0; k--) {
      // image of this stage: stages 1 .. LCAP where the rollout reads them, the others in the staging area (stage 0
      // stays there; later ones leave for the gain record at the top of the next stage)
      ldouble *const imk = (k >= 1 && k <= LCAPr) ? limg + (size_t)(k - 1) * KPW : img;
      if (k + 1 < N && k + 1 > LCAPr) {   // (uniform) the image of stage k + 1 is complete in the staging area
        gdouble *const kp1 = kpb + (size_t)(k + 1) * kps;
#pragma unroll
        for (int u = 0; u < KPL; u++) {
          const int e = lane + LPI * u;
          kp1[e < KPW ? e : KPW] = img[e < KPW ? e : 0];   // (KPW: the record's spare word)
        }
      }
      double rc_[8];
#pragma unroll
      for (int u = 0; u < 8; u++) rc_[u] = rn[u];
      {
        const int kn = k > 0 ? k - 1 : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) rn[u] = rb[(size_t)kn * sstr + ro[u]];   // travels while this stage is computed
      }
      RST(0);
      // ---- phase A -------------------------------------------------------------------------------------------
      {
        const double S = aP[oS], T = aP[oT], U = aP[oU], V = aP[oV], p1 = ap[ii], p2 = ap[n + ii];
        __builtin_amdgcn_sched_barrier(0);
        const double rcj = gj < n ? rc_[4] : 0.0, rcnj = gj < n ? rc_[5] : 0.0;
        const double tv = h * S + T, tu = h2 * S + h * T, bv = h * U + V, bu = h2 * U + h * V;
        const double qq = S + (rc_[0] - cwt * rc_[1]);
        const double vq = h * S + U;
        const double vv = (h * (h * S + (T + U)) + V) + (gdiag ? rc_[2] : 0.0);
        const double uq = h2 * S + h * U, uv = h2 * tv + h * bv;
        const double uu = (h2 * tu + h * bu) + (gdiag ? rc_[3] : 0.0);
        *dS = qq; *dT = tv; *dU = vq; *dV = vv;
        *dUq = uq; *dUv = uv; *dUu = uu;
        // g = P rc + p: the group's partial products, summed over its 8 lanes
        const double g1 = p1 + dpp_sum8(S * rcj + T * rcnj), g2 = p2 + dpp_sum8(U * rcj + V * rcnj);
        *dq = (rc_[6] - mu * rc_[7]) + (gc1 * g1 + gc2 * g2);
        *(rcw ? imk + OFF_RC + lane : dummy) = rc_[4];
        *(rcw ? imk + OFF_RC + n + lane : dummy) = rc_[5];
      }
      WSYNC();
      RST(1);
      // ---- phase B: Cholesky of Quu (every lane), Y = L^-1 [Qux | qu] (one column per lane), gains -------------
      v4d acc;
      {
        double qw[NW][NW], colv[NW], ac[4];
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
          for (int i = j; i < NW; i++) qw[i][j] = aQuu[i * 8 + j];
#pragma unroll
        for (int i = 0; i < NW; i++) colv[i] = aQux[i * QS + bc];
#pragma unroll
        for (int r = 0; r < 4; r++) ac[r] = aP[co[r]];   // (accumulator of phase C: arrives during the factorisation)
        __builtin_amdgcn_sched_barrier(0);
        double L[NW][NW], invd[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
          double dg = qw[j][j];
#pragma unroll
          for (int l = 0; l < j; l++) dg -= L[j][l] * L[j][l];
          if (!(dg > 0.0)) chol_ok = false;
          // 1/sqrt(dg): hardware estimate + two Newton steps (full double precision), then sqrt = dg * rsqrt
          double inv = __builtin_amdgcn_rsq(dg);
          inv = inv * (1.5 - 0.5 * dg * inv * inv);
          inv = inv * (1.5 - 0.5 * dg * inv * inv);
          L[j][j] = dg * inv;
          invd[j] = inv;
#pragma unroll
          for (int i = j + 1; i < NW; i++) {
            double s = qw[i][j];
#pragma unroll
            for (int l = 0; l < j; l++) s -= L[i][l] * L[j][l];
            L[i][j] = s * inv;
          }
        }
        double y[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) {
          double s = colv[i];
#pragma unroll
          for (int l = 0; l < i; l++) s -= L[i][l] * y[l];
          y[i] = s * invd[i];
          dY[i * ystr] = y[i];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = cin[r] ? ac[r] : 0.0;
        WSYNC();
        RST(3);
        // ---- phase C: [P | p] = [Qxx | qx] - Y^T [Y | y] --------------------------------------------------------
        const double ya = aY[kq * 16 + c16], yb = aY[(4 + kq) * 16 + c16];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-ya, ya, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-yb, yb, acc, 0, 0, 0);
        // (behind the matrix instructions: the gains K = -L^-T Y of this lane's column, for the rollout only)
        double x[NW];
#pragma unroll
        for (int i = NW - 1; i >= 0; i--) {
          double s = y[i];
#pragma unroll
          for (int l = i + 1; l < NW; l++) s -= L[l][i] * x[l];
          x[i] = s * invd[i];
        }
        ldouble *const kd = koff >= 0 ? imk + koff : dummy;
#pragma unroll
        for (int i = 0; i < NW; i++) kd[i * kstr] = -x[i];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          *cd1[r] = acc[r];
          *(po2[r] >= 0 ? imk + po2[r] : dummy) = acc[r];
        }
      }
      WSYNC();   // (P, p of this stage are read by the next stage's phase A)
      RST(4);
    }
    if (!chol_ok) return false;
    // ---- rollout: dw = kff + K dx, nu+ = p + P dx, dx+ = rc + [A|B][dx; dw], one ordering point per stage --------
    const bool fA = lane < NW, fB = lane >= NW && lane < NW + NX, fC = lane >= NW + NX && lane < NW + 2 * NX;
    const int fi = fA ? lane : (fB ? lane - NW : (fC ? lane - NW - NX : 0));   // entry of dw / nu+ / dx+
    const int fw = fC ? (fi < n ? fi : fi - n) : fi;                             // the entry of dw a dx+ lane needs
    const int foff = fB ? OFF_P + fi : OFF_KFF + fw;
    int frow[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) frow[j] = fB ? OFF_PT + tri(fi, j) : fw * NX + j;
    const int fx1 = fi < n ? n + fi : fi;
    const double fca = fi < n ? h : 0.0, fcb = fi < n ? h2 : h;
    const size_t dzslot = fA ? (size_t)(NX + lane) : (size_t)fi;
    if (lane < 32) adx[lane] = 0.0;
    constexpr int FD = 4;
    double fvq[FD][KPL];
    auto fetch_fwd = [&](int k, double (&fv)[KPL]) __attribute__((always_inline)) {
      const int kk = k < N ? k : N - 1;
#pragma unroll
      for (int u = 0; u < KPL; u++) {
        const int e = lane + LPI * u;
        fv[u] = kpb[(size_t)kk * kps + (e < KPW ? e : 0)];
      }
    };
#pragma unroll
    for (int d = 0; d < FD; d++) fetch_fwd(LCAPr + 1 + d, fvq[d]);
    auto fwd_stage = [&](const int k, double (&fv)[KPL], const bool from_mem) __attribute__((always_inline)) {
      const ldouble *const im = (!from_mem && k > 0) ? limg + (size_t)(k - 1) * KPW : img;
      if (from_mem) {
#pragma unroll
        for (int u = 0; u < KPL; u++) *(lane + LPI * u < KPW ? img + lane + LPI * u : dummy) = fv[u];
        fetch_fwd(k + FD, fv);
      }
      const ldouble *const dxc = adx + 16 * (k & 1);
      ldouble *const dxn = adx + 16 * ((k & 1) ^ 1);
      WSYNC();
      RST(5);
      double dxv[NX], rowv[NX];
#pragma unroll
      for (int j = 0; j < NX; j++) { dxv[j] = dxc[j]; rowv[j] = im[frow[j]]; }
      double sacc = im[foff];
      const double rcv = im[OFF_RC + fi], d0 = dxc[fi], d1 = dxc[fx1];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NX; j++) sacc += rowv[j] * dxv[j];
      // (unconditional stores, idle lanes to the spare word of the stage's gain record)
      SP *sink;
      if constexpr (std::is_same<SP, ldouble>::value) sink = dummy;
      else sink = (SP *)(kpb + (size_t)k * kps + KPW);
      *((fA || fC) ? so.dz + dzslot * so.SS + (size_t)k * so.KS : sink) = fA ? sacc : d0;
      *((fB && k >= 1) ? so.nunew + (size_t)fi * so.SS + (size_t)k * so.KS : sink) = sacc;
      double sx = rcv;
      sx += d0;
      sx += fca * d1;
      sx += fcb * sacc;
      *((fC && k < N - 1) ? dxn + fi : dummy) = sx;
      RST(6);
    };
    for (int k = 0; k < N && k <= LCAPr; k++) fwd_stage(k, fvq[0], false);
    for (int k0 = LCAPr + 1; k0 < N; k0 += FD) {
#pragma unroll
      for (int d = 0; d < FD; d++) {
        if (k0 + d < N) fwd_stage(k0 + d, fvq[d], true);   // (uniform branch)
      }
    }
    RST_FLUSH();
    return true;
  }
  // ---- generic path (pass kernels; fused kernel of models without a path of their own): per-lane constants ----------
  // idle lanes store to a word of the unused T area (several lanes may share one: the value is never read)
  ldouble *const gdummy = sT + lane % RicLds<C, LPI>::TW;
  ldouble *gsrdst[RPL];
#pragma unroll
  for (int u = 0; u < RPL; u++) gsrdst[u] = lane + LPI * u < C::RS ? srec + lane + LPI * u : gdummy;
  int go11[EPL];
  double gl1[EPL], gl2[EPL], gc1[EPL], gc2[EPL];
  bool gon[EPL];
  ldouble *gqdst[EPL];
#pragma unroll
  for (int u = 0; u < EPL; u++) {
    const int e = lane + LPI * u;
    const bool ok = e < NV * NV;
    const int ec = ok ? e : 0;
    const int i = ec / NV, j = ec - i * NV;
    // row / column kind: 0 = q, 1 = v, 2 = u, 3 = slack (no contribution)
    const int ki = i < NQ ? 0 : (i < NX ? 1 : (i >= NX + NS ? 2 : 3));
    const int kj = j < NQ ? 0 : (j < NX ? 1 : (j >= NX + NS ? 2 : 3));
    gon[u] = ok && ki != 3 && kj != 3;
    const int ii = !gon[u] ? 0 : (ki == 0 ? i : (ki == 1 ? i - NQ : i - NX - NS));
    const int jj = !gon[u] ? 0 : (kj == 0 ? j : (kj == 1 ? j - NQ : j - NX - NS));
    go11[u] = ii * NX + jj;
    // left factor: row kind picks the combination of the two block rows -- q: (1, 0); v: (h, 1); u: (h2, h)
    gl1[u] = ki == 0 ? 1.0 : (ki == 1 ? h : h2); gl2[u] = ki == 0 ? 0.0 : (ki == 1 ? 1.0 : h);
    gc1[u] = kj == 0 ? 1.0 : (kj == 1 ? h : h2); gc2[u] = kj == 0 ? 0.0 : (kj == 1 ? 1.0 : h);
    gqdst[u] = ok ? sQ + e : gdummy;
  }
  const int glr = lane < NX ? lane : 0, glv = lane < NV ? lane : 0;
  ldouble *const gsrc = lane < NX ? src + lane : gdummy, *const gpcdst = lane < NX ? sPc + lane : gdummy;
  ldouble *const gsqdst = lane < NV ? sq + lane : gdummy;
  const int gkq = glv < NQ ? 0 : (glv < NX ? 1 : (glv >= NX + NS ? 2 : 3));
  const bool gqon = lane < NV && gkq != 3;
  const int giq = !gqon ? 0 : (gkq == 0 ? glv : (gkq == 1 ? glv - NQ : glv - NX - NS));
  const double gl1q = gkq == 0 ? 1.0 : (gkq == 1 ? h : h2), gl2q = gkq == 0 ? 0.0 : (gkq == 1 ? 1.0 : h);
  const int glc = lane <= NX ? lane : 0;   // gain column of this lane (NX: kff)
  ldouble *const gkdst = lane <= NX ? (lane < NX ? sK + lane : skf) : gdummy;
  const int gkstr = lane < NX ? NX : (lane == NX ? 1 : 0);
  if constexpr (!FAST && !DDFAST) fetch_stage(N - 1);
  RST_DECL();
  for (int k = (FAST || DDFAST) ? -1 : N - 1; k >= 0; k--) {
    // -- the image of stage k+1 is complete: it leaves for the gain record (read now, stored after the
    //    barrier); the stage record goes to LDS, the request for the next one leaves ----------------
    double kpv[KPL];
#pragma unroll
    for (int u = 0; u < KPL; u++) {
      const int e = lane + LPI * u;
      kpv[u] = img[e < KPW ? e : 0];
    }
#pragma unroll
    for (int u = 0; u < RPL; u++) *gsrdst[u] = recv[u];
    if (k > 0) fetch_stage(k - 1);  // travels while this stage is computed
    WSYNC();
    RST(0);
    if (k < N - 1) {
      if constexpr (SLOTS) {
        ldouble *const kp1 = slots + (size_t)(k + 1) * GS;
#pragma unroll
        for (int u = 0; u < KPL; u++) {
          const int e = lane + LPI * u;
          if (e < KPW) kp1[e] = kpv[u];
        }
      } else if (LIMG && k + 1 <= LCAP) {
        ldouble *const kl = limg + (size_t)k * KPW;   // (slot of stage k + 1)
#pragma unroll
        for (int u = 0; u < KPL; u++) *(lane + LPI * u < KPW ? kl + lane + LPI * u : gdummy) = kpv[u];
      } else {
        gdouble *const kp1 = kpb + (size_t)(k + 1) * kps;
#pragma unroll
        for (int u = 0; u < KPL; u++) kp1[lane + LPI * u < KPW ? lane + LPI * u : KPW] = kpv[u];   // (KPW: the record's spare word)
      }
    }
    if constexpr (!DD) {
      // Holonomic chain: A = [I hI; 0 I], B = [h2 I; h I].  The dense stage Hessian is formed in one step
      // from the record blocks and [A|B]^T P [A|B] in closed form (each entry from at most four entries of
      // P: blocks 11, 12, 21, 22 at (ii, jj)); rc of the stage goes into the image (stage N-1: finite, unused).
      // Straight-line phases: what an entry is made of and where it goes are loop-invariant per-lane constants
      // (go*), every read of a phase is issued before its first use, every store is unconditional (idle lanes write
      // to gdummy) -- as for the fused kernel's paths above; same arithmetic per entry.
      const bool rec_cost = k < N - 1;   // a cost-to-go of stage k+1 exists
      {
        double r0[EPL], r1[EPL], a11[EPL], a12[EPL], a21[EPL], a22[EPL], pr[NX], rcl[NX];
#pragma unroll
        for (int u = 0; u < EPL; u++) {
          r0[u] = srec[qp[u]]; r1[u] = srec[cp[u]];
          a11[u] = sP[go11[u]]; a12[u] = sP[go11[u] + NQ]; a21[u] = sP[go11[u] + NQ * NX]; a22[u] = sP[go11[u] + NQ * NX + NQ];
        }
#pragma unroll
        for (int l = 0; l < NX; l++) { pr[l] = sP[glr * NX + l]; rcl[l] = srec[C::R_RC + l]; }
        const double rcme = srec[C::R_RC + glr];
        double pcs = sp[glr];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < EPL; u++) {
          double v = r0[u] - cwt * r1[u];
          const double add = gl1[u] * (gc1[u] * a11[u] + gc2[u] * a12[u]) + gl2[u] * (gc1[u] * a21[u] + gc2[u] * a22[u]);
          v = (rec_cost && gon[u]) ? v + add : v;
          *gqdst[u] = v;
        }
#pragma unroll
        for (int l = 0; l < NX; l++) pcs += pr[l] * rcl[l];
        *gsrc = rcme;
        *gpcdst = pcs;     // (stage N-1: P = 0, p = 0 -- the product is not used)
      }
      WSYNC();
      RST(1);
      {
        const double q0v = srec[C::R_Q0 + glv], q1v = srec[C::R_Q1 + glv], pc1 = sPc[giq], pc2 = sPc[NQ + giq];
        __builtin_amdgcn_sched_barrier(0);
        double v = q0v - mu * q1v;
        v = (rec_cost && gqon) ? v + (gl1q * pc1 + gl2q * pc2) : v;
        *gsqdst = v;
      }
      WSYNC();
      RST(2);
    } else {
      // -- fill: dense stage Hessian, gradient, defect (rc of stage N-1: finite, unused), [A|B] -------
#pragma unroll
      for (int u = 0; u < EPL; u++) {
        const int e = lane + LPI * u;
        const double v = srec[qp[u]] - cwt * srec[cp[u]];
        if (e < NV * NV) sQ[e] = v;
      }
      if (lane < NV) sq[lane] = srec[C::R_Q0 + lane] - mu * srec[C::R_Q1 + lane];
      if (lane < NX) src[lane] = srec[C::R_RC + lane];
      if (k < N - 1) fill_AB_dd();
      WSYNC();
      if (k < N - 1) {
      // -- T = P [A|B], Pc = P rc + p ---------------------------------------------------------
#pragma unroll
      for (int u = 0; u < TPL; u++) {
        const int e = lane + LPI * u;
        if (e < NX * NV) {
          const int i = e / NV, j = e - i * NV;
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < NX; l++) s += sP[i * NX + l] * sAB[l * NV + j];
          sT[e] = s;
        }
      }
      if (lane < NX) {
        double s = sp[lane];
#pragma unroll
        for (int l = 0; l < NX; l++) s += sP[lane * NX + l] * src[l];
        sPc[lane] = s;
      }
      WSYNC();
      // -- Q += [A|B]^T T, q += [A|B]^T Pc ---------------------------------------------------------
#pragma unroll
      for (int u = 0; u < EPL; u++) {
        const int e = lane + LPI * u;
        if (e < NV * NV) {
          const int i = e / NV, j = e - i * NV;
          double s = sQ[e];
#pragma unroll
          for (int l = 0; l < NX; l++) s += sAB[l * NV + i] * sT[l * NV + j];
          sQ[e] = s;
        }
      }
      if (lane < NV) {
        double s = sq[lane];
#pragma unroll
        for (int l = 0; l < NX; l++) s += sAB[l * NV + lane] * sPc[l];
        sq[lane] = s;
      }
      WSYNC();
      }
    }
    // -- Cholesky of Qww: every lane factors the small block in registers (its entries and the lane's right-hand
    //    side are requested first, all at once) -----------------------------------------------------------------
    double qw[NW][NW], colv[NW];
#pragma unroll
    for (int j = 0; j < NW; j++)
#pragma unroll
      for (int i = j; i < NW; i++) qw[i][j] = sQ[(NX + i) * NV + NX + j];
#pragma unroll
    for (int i = 0; i < NW; i++) colv[i] = glc < NX ? sQ[(NX + i) * NV + glc] : sq[NX + i];
    __builtin_amdgcn_sched_barrier(0);
    double L[NW][NW], invd[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) {
      double dg = qw[j][j];
#pragma unroll
      for (int l = 0; l < j; l++) dg -= L[j][l] * L[j][l];
      if (!(dg > 0.0)) chol_ok = false;
      // 1/sqrt(dg): hardware estimate + two Newton steps (full double precision), then sqrt = dg * rsqrt
      double inv = __builtin_amdgcn_rsq(dg);
      inv = inv * (1.5 - 0.5 * dg * inv * inv);
      inv = inv * (1.5 - 0.5 * dg * inv * inv);
      L[j][j] = dg * inv;
      invd[j] = inv;
#pragma unroll
      for (int i = j + 1; i < NW; i++) {
        double s = qw[i][j];
#pragma unroll
        for (int l = 0; l < j; l++) s -= L[i][l] * L[j][l];
        L[i][j] = s * inv;
      }
    }
    // -- gains: lane c < NX solves for column c of K, lane NX for kff (the other lanes solve column 0 again and
    //    store to gdummy) ---------------------------------------------------------------------------------------
    {
      double col[NW];
#pragma unroll
      for (int i = 0; i < NW; i++) col[i] = -colv[i];
      chol_solve<NW>(L, invd, col);
#pragma unroll
      for (int i = 0; i < NW; i++) gkdst[i * gkstr] = col[i];
    }
    WSYNC();
    RST(3);
    // -- cost-to-go: P = sym(Qxx + Qxw K), p = qx + Qxw kff ---------------------------------------------
    if constexpr (MFMA_P) {
      // The arms, one wavefront per instance: the 14 x 7 x 15 product on the matrix cores.  M = Qxw [K | kff] is one
      // 16 x 16 tile of v_mfma_f64_16x16x4_f64 (two k-steps of four), accumulated onto C = [Qxx | qx]; its transpose
      // M^T = [K | kff]^T Qxw^T comes from the same two operand registers swapped, accumulated onto Qxx^T, so that
      // a lane holds M(i, j) and M(j, i) for its four entries: 14 LDS reads and 4 matrix instructions per lane
      // instead of 120 reads and 56 multiply-adds (the sums keep the order l = 0 .. NW-1 on top of the Qxx entry).
      // Operand maps (guide, "Fragment layout"): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
      // C / D[row = (lane >> 4) + 4 r][col = lane & 15], r = 0 .. 3.
      typedef double v4d __attribute__((ext_vector_type(4)));
      const int c16 = lane & 15, kq = lane >> 4;
      const bool cin = c16 < NX;
      const int cc = cin ? c16 : 0;
      double a0 = sQ[cc * NV + NX + kq];                                   // Qxw(c16, kq)
      double a1 = sQ[cc * NV + NX + (4 + kq < NW ? 4 + kq : 0)];           // Qxw(c16, 4 + kq)
      double b0 = c16 == NX ? skf[kq] : sK[kq * NX + cc];                  // K(kq, c16) | kff(kq)
      double b1 = c16 == NX ? skf[4 + kq < NW ? 4 + kq : 0] : sK[(4 + kq < NW ? 4 + kq : 0) * NX + cc];
      a0 = cin ? a0 : 0.0;
      a1 = (cin && 4 + kq < NW) ? a1 : 0.0;
      b0 = c16 <= NX ? b0 : 0.0;
      b1 = (c16 <= NX && 4 + kq < NW) ? b1 : 0.0;
      v4d cacc, tacc;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = kq + 4 * r;
        const bool rin = row < NX;
        const int rr = rin ? row : 0;
        const double qe = sQ[rr * NV + cc], qt = sQ[cc * NV + rr], qv = sq[rr];
        cacc[r] = !rin ? 0.0 : (cin ? qe : (c16 == NX ? qv : 0.0));
        tacc[r] = (rin && cin) ? qt : 0.0;
      }
      cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, cacc, 0, 0, 0);
      tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, tacc, 0, 0, 0);
      cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cacc, 0, 0, 0);
      tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, tacc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = kq + 4 * r;
        if (row < NX) {
          if (cin) {
            const double pn = 0.5 * (cacc[r] + tacc[r]);
            sP[row * NX + c16] = pn;
            if (row <= c16) sPt[tri(row, c16)] = pn;
          } else if (c16 == NX) {
            sp[row] = cacc[r];
          }
        }
      }
    } else {
    // entries e < NX*NX are P(i, j); the next NX entries are p(i), written as the same expression with the
    // "column" kff and no transposed partner (a == c, 0.5 (a + a) = a exactly): one instruction stream
    constexpr int PPL2 = (NX * NX + NX + LPI - 1) / LPI;
    double pn[PPL2];
#pragma unroll
    for (int u = 0; u < PPL2; u++) {
      const int e = lane + LPI * u;
      pn[u] = 0.0;
      if (e < NX * NX + NX) {
        const bool isP = e < NX * NX;
        const int i = isP ? e / NX : e - NX * NX, j = isP ? e - i * NX : 0;
        const ldouble *const a0 = isP ? sQ + i * NV + j : sq + i;
        const ldouble *const c0 = isP ? sQ + j * NV + i : sq + i;
        const ldouble *const cq = isP ? sQ + j * NV + NX : sQ + i * NV + NX;
        double a = *a0, c = *c0;
#pragma unroll
        for (int l = 0; l < NW; l++) {
          a += sQ[i * NV + NX + l] * (isP ? sK[l * NX + j] : skf[l]);
          c += cq[l] * (isP ? sK[l * NX + i] : skf[l]);
        }
        pn[u] = 0.5 * (a + c);
      }
    }
    // (no ordering point needed: what is written now -- sP, sPt, sp -- is not read in this phase, and the
    //  LDS instructions of a wavefront execute in program order)
#pragma unroll
    for (int u = 0; u < PPL2; u++) {
      const int e = lane + LPI * u;
      if (e < NX * NX) {
        sP[e] = pn[u];
        const int i = e / NX, j = e - i * NX;
        if (i <= j) sPt[tri(i, j)] = pn[u];
      } else if (e < NX * NX + NX) {
        sp[e - NX * NX] = pn[u];
      }
    }
    }
    // (the fill of the next stage touches sQ / sq / src only; its barrier orders the sP writes)
    RST(4);
  }
  if (!chol_ok) return false;

  // ---- fused kernel, holonomic chain without slack: forward rollout with ONE exchange per stage ------------
  // dw = kff + K dx (lanes < NW), nu+ = p + P dx (the next NX lanes) and dx+ = rc + [A|B][dx; dw] (lanes < NX) are
  // all formed from dx alone: a lane that needs an entry of dw for its dx+ computes that entry itself (same
  // expression, same value) instead of waiting for the lane that stores it.  dx ping-pongs between two buffers.
  constexpr bool FASTF = FAST && (NS == 0);
  if constexpr (FASTF) {
    constexpr int OFF_KFF = NW * NX, OFF_PT = NW * NX + NW, OFF_P = OFF_PT + NP2, OFF_RC = OFF_P + NX;
    // two dx buffers in the work area: sdx (NX) and sT (large, unused by the chain model)
    ldouble *const dx0 = sdx, *const dx1 = sT;
    WSYNC();
    if (lane < NX) dx0[lane] = 0.0;
    const bool isq = lane < NQ;
    const int lx = lane < NX ? lane : 0;                    // dx+ entry of this lane
    const int iw = isq ? lx : lx - NQ;                      // the entry of dw it needs (NS == 0: u_i pairs with q_i and v_i)
    const bool isn = lane >= NW && lane < NW + NX;          // nu+ entry lane - NW
    const int in = isn ? lane - NW : 0;
    int tro[NX];                                            // packed-triangle offsets of row `in` of P
#pragma unroll
    for (int j = 0; j < NX; j++) tro[j] = tri(in, j);
    for (int k = 0; k < N; k++) {
      const ldouble *const im = slots + (size_t)k * GS;
      const ldouble *const dxc = (k & 1) ? dx1 : dx0;
      ldouble *const dxn = (k & 1) ? dx0 : dx1;
      WSYNC();
      double dx[NX], kr[NX], pr[NX];
#pragma unroll
      for (int j = 0; j < NX; j++) { dx[j] = dxc[j]; kr[j] = im[iw * NX + j]; pr[j] = im[OFF_PT + tro[j]]; }
      const double kf = im[OFF_KFF + iw], pv = im[OFF_P + in], rcv = im[OFF_RC + lx];
      const double dxme = dxc[lx], dxv = dxc[isq ? NQ + lx : lx], dxo = dxc[in];
      __builtin_amdgcn_sched_barrier(0);   // (every read of the phase is issued before the first use: one counted wait instead of a wait per use)
      double dw = kf, nup = pv;
#pragma unroll
      for (int j = 0; j < NX; j++) { dw += kr[j] * dx[j]; nup += pr[j] * dx[j]; }
      // step of the stage: lanes < NW hold dw (slots NX..), the next NX lanes dx (slots 0..); nu+ for k >= 1
      // (the step lives in the slots: its strides are constants -- with the run-time strides of StepOut the address
      //  arithmetic was 82 of the 176 instructions of a forward stage)
      if (lane < NW + NX) so.dz[(lane < NW ? NX + lane : lane - NW) + k * GS] = lane < NW ? dw : dxo;
      if (isn && k >= 1) so.nunew[in + k * GS] = nup;
      if (k < N - 1 && lane < NX) {
        double sx = rcv;
        sx += dxme;
        sx += (isq ? h : 0.0) * dxv;
        sx += (isq ? h2 : h) * dw;
        dxn[lane] = sx;
      }
    }
    return true;
  }
  // ---- fused kernel, diff-drive: forward rollout with ONE ordering point per stage ---------------------------------
  // As in the chain's path, the lane that forms an entry of dx+ computes the two input steps it needs itself (same
  // expression, same value as the lane that stores them); dx ping-pongs between two buffers; the image and [A5 | B5]
  // of the next stage travel from the gain record / the stage record while this stage is computed; the products
  // with [A | B] keep only its non-trivial entries, in the dense order.
  if constexpr (DDFAST) {
    constexpr int NR = 5;
    constexpr int OFF_KFF = NW * NX, OFF_PT = NW * NX + NW, OFF_P = OFF_PT + NP2, OFF_RC = OFF_P + NX;
    ldouble *const sA5 = sAB, *const sB5 = sAB + 25, *const sZ = sAB + 35;
    ldouble *const dx0 = sdx, *const dx1 = sPc;
    auto Rl = [](int r) __attribute__((always_inline)) { return r < 3 ? r : r + 3; };
    auto rid = [](int j) __attribute__((always_inline)) { return j < 3 ? j : (j >= 6 && j < 8 ? j - 3 : -1); };
    double fv[KPL], abf[2] = {0, 0};
    auto fetch_f = [&](int k) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < KPL; u++) {
        const int e = lane + LPI * u;
        fv[u] = kpb[(size_t)k * kps + (e < KPW ? e : 0)];
      }
#pragma unroll
      for (int u = 0; u < 2; u++) abf[u] = rb[(size_t)k * sstr + C::R_A5 + (lane + LPI * u < 35 ? lane + LPI * u : 0)];
    };
    WSYNC();
    if (lane < NX) dx0[lane] = 0.0;
    if (N > 1) fetch_f(1);
    // role of the lane in the "offset + row . dx" stream: an input / slack step (lanes < NW) or a costate
    const bool isw = lane < NW, isn = lane >= NW && lane < NW + NX;
    const int in = isn ? lane - NW : 0;
    int ro[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) ro[j] = isw ? lane * NX + j : OFF_PT + tri(in, j);
    const int oo = isw ? OFF_KFF + lane : OFF_P + in;
    // dx+ entry of the lane
    const int lx = lane < NX ? lane : 0;
    const bool xI = rid(lx) < 0;
    const int ao = xI ? 35 : rid(lx) * NR, as = xI ? 0 : 1;   // row of A5 (offsets in sAB; 35 = the zero word)
    const int bo = xI ? 35 : 25 + rid(lx) * 2, bs = xI ? 0 : 1;
    // (unconditional stores and batched reads as in the backward pass)
    ldouble *const fdummy = srec + lane;
    ldouble *fdst[KPL], *fab[2];
#pragma unroll
    for (int u = 0; u < KPL; u++) fdst[u] = lane + LPI * u < KPW ? img + lane + LPI * u : fdummy;
#pragma unroll
    for (int u = 0; u < 2; u++) fab[u] = lane + LPI * u < 35 ? sAB + lane + LPI * u : fdummy;
    // (the two global stores of the step stay predicated: there is no spare word in the step arrays)
    for (int k = 0; k < N; k++) {
      const ldouble *const dxc = (k & 1) ? dx1 : dx0;
      ldouble *const dxn = (k & 1) ? dx0 : dx1;
      // image and [A5 | B5] of this stage (what the previous iteration requested; stage 0's are in place)
#pragma unroll
      for (int u = 0; u < KPL; u++) *(k > 0 ? fdst[u] : fdummy) = fv[u];
#pragma unroll
      for (int u = 0; u < 2; u++) *(k > 0 ? fab[u] : fdummy) = abf[u];
      if (k + 1 < N) fetch_f(k + 1);
      WSYNC();
      double dx[NX], rw[NX], ku[2][NX], ar[NR], br[2];
#pragma unroll
      for (int j = 0; j < NX; j++) { dx[j] = dxc[j]; rw[j] = img[ro[j]]; }
      const double own0 = img[oo];
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int j = 0; j < NX; j++) ku[c][j] = img[(NS + c) * NX + j];
      const double kf0 = img[OFF_KFF + NS], kf1 = img[OFF_KFF + NS + 1];
#pragma unroll
      for (int r = 0; r < NR; r++) ar[r] = sAB[ao + r * as];
#pragma unroll
      for (int c = 0; c < 2; c++) br[c] = sAB[bo + c * bs];
      const double rcv = img[OFF_RC + lx];
      const double dxo = dxc[in], dxme = dxc[lx];
      __builtin_amdgcn_sched_barrier(0);
      // own entry of the step / costate
      double sown = own0;
#pragma unroll
      for (int j = 0; j < NX; j++) sown += rw[j] * dx[j];
      // the two input steps (every lane: the dx+ lanes need them)
      double du[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        double sacc = c == 0 ? kf0 : kf1;
#pragma unroll
        for (int j = 0; j < NX; j++) sacc += ku[c][j] * dx[j];
        du[c] = sacc;
      }
      if (lane < NW + NX) so.dz[(size_t)(isw ? NX + lane : in) * so.SS + (size_t)k * so.KS] = isw ? sown : dxo;
      if (isn && k >= 1) so.nunew[(size_t)in * so.SS + (size_t)k * so.KS] = sown;
      {
        double sx = rcv;
#pragma unroll
        for (int r = 0; r < NR; r++) sx += ar[r] * dx[Rl(r)];
        sx += xI ? dxme : 0.0;
#pragma unroll
        for (int c = 0; c < 2; c++) sx += br[c] * du[c];
        *((k < N - 1 && lane < NX) ? dxn + lane : fdummy) = sx;
      }
    }
    return true;
  }
  // ---- forward rollout + costates nu+_k = P_k dx_k + p_k ------------------------------------------
  // the image of stage 0 is still in LDS; later stages come back from the gain record (one request each)
  WSYNC();
  if (lane < NX) sdx[lane] = 0.0;
  // Gain images from global memory: FD stages in flight.  (One stage ahead was not enough: a stage of the rollout is
  // ~600 cycles of work and an image takes several thousand cycles to come back from the Infinity Cache -- the gain
  // records of a launch, 39 MB for 1024 arms, do not fit the L2 -- so that the arm's rollout was 3.4 k cycles per
  // stage, 40 % of its recursion: tests/tools/dev_ric_stamps.py.)  The stage loop is unrolled FD times so that the
  // buffer index is static; requests beyond the horizon are clamped, not skipped.
  constexpr int FD = SLOTS ? 1 : 4;
  double fvq[FD][KPL];
  auto fetch_fwd = [&](int k, double (&fv)[KPL]) __attribute__((always_inline)) {
    const int kk = k < N ? k : N - 1;
#pragma unroll
    for (int u = 0; u < KPL; u++) {
      const int e = lane + LPI * u;
      fv[u] = kpb[(size_t)kk * kps + (e < KPW ? e : 0)];
    }
  };
  if constexpr (!SLOTS) {
#pragma unroll
    for (int d = 0; d < FD; d++) fetch_fwd(LCAP + 1 + d, fvq[d]);
  }
  // (straight-line stages as in the backward pass: clamped per-lane rows, reads before the first use, idle lanes
  //  store to gdummy; the two stores of the step to global memory stay predicated)
  const bool fisw = lane < NW, fact = lane < NW + NX;
  const int fi = fisw ? lane : (fact ? lane - NW : 0);
  const int foff = fisw ? (NW * NX + lane) : (NW * NX + NW + NP2 + fi);
  int frow[NX];
#pragma unroll
  for (int j = 0; j < NX; j++) frow[j] = fisw ? (lane * NX + j) : (NW * NX + NW + tri(fi, j));
  ldouble *const fdwdst = fisw ? sdw + lane : gdummy;
  ldouble *fimg[KPL];
#pragma unroll
  for (int u = 0; u < KPL; u++) fimg[u] = lane + LPI * u < KPW ? img + lane + LPI * u : gdummy;
  const bool fisq = glr < NQ;
  auto fwd_stage = [&](const int k, double (&fv)[KPL], const bool from_mem) __attribute__((always_inline)) {
    // image of this stage: stage 0's is still in the work area; later ones are read where the backward pass left
    // them (SLOTS; limg for stages 1 .. LCAP), or come back from the gain record (copied into the work area)
    const ldouble *const im = (SLOTS && (k > 0 || FAST)) ? slots + (size_t)k * GS
                              : ((LIMG && !from_mem && k > 0) ? limg + (size_t)(k - 1) * KPW : img);
    if constexpr (!SLOTS) {
      if (from_mem) {
#pragma unroll
        for (int u = 0; u < KPL; u++) *fimg[u] = fv[u];
        fetch_fwd(k + FD, fv);   // (the buffer is free again: the image of stage k + FD takes its place)
      }
    }
    if constexpr (DD) {
      if (k < N - 1) {   // [A|B] of stage k straight from its record (diff-drive only)
#pragma unroll
        for (int u = 0; u < TPL; u++) {
          const int e = lane + LPI * u;
          const double v = rb[(size_t)k * sstr + abp[u]] + abc[u];
          if (e < NX * NV) sAB[e] = v;
        }
      }
    }
    WSYNC();
    RST(5);
    // (the defect of the stage, read before the step of the stage may overwrite it: SLOTS)
    const double rcv = im[NW * NX + NW + NP2 + NX + glr];
    // dw = kff + K dx (lanes < NW) and nu+ = p + P dx (the next NX lanes) as ONE instruction stream: both
    // are "offset + row . dx" over the image, only the per-lane addresses differ (LDS instruction count
    // is what bounds this kernel when the whole batch iterates)
    double dxv[NX], rowv[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) { dxv[j] = sdx[j]; rowv[j] = im[frow[j]]; }
    double sacc = im[foff];
    const double dxi = sdx[fi < NX ? fi : 0];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) sacc += rowv[j] * dxv[j];
    *fdwdst = sacc;
    const double dzv = fisw ? sacc : dxi;
    // dz of the stage in one request: lanes < NW hold dw (slots NX..), the next NX lanes dx (slots 0..)
    if constexpr (SLOTS) {
      if (fact && !fisw && k >= 1) so.nunew[(size_t)fi * so.SS + (size_t)k * so.KS] = sacc;
      if (fact) so.dz[(size_t)(fisw ? NX + lane : lane - NW) * so.SS + (size_t)k * so.KS] = dzv;
    } else {
      // (unconditional stores, idle lanes to the spare word of the stage's gain record: behind a predicated store the
      //  compiler no longer knows how many requests are in flight and waits for ALL of them -- the images of the next
      //  stages included -- before it touches the oldest)
      SP *sink;
      if constexpr (std::is_same<SP, ldouble>::value) sink = gdummy;
      else sink = (SP *)(kpb + (size_t)k * kps + KPW);
      *((fact && !fisw && k >= 1) ? so.nunew + (size_t)fi * so.SS + (size_t)k * so.KS : sink) = sacc;
      *(fact ? so.dz + (size_t)(fisw ? NX + lane : lane - NW) * so.SS + (size_t)k * so.KS : sink) = dzv;
    }
    WSYNC();
    double dxn = 0.0;
    if constexpr (!DD) {
      // holonomic chain, closed form of rc + [A|B][dx; dw] (same order of the non-zero terms as the dense
      // product): q rows dx_i + h dx_{n+i} + h2 dw_i, v rows dx_i + h dw_{i-n}
      const double d0 = sdx[glr], d1 = sdx[fisq ? NQ + glr : glr], w0 = sdw[NS + (fisq ? glr : glr - NQ)];
      __builtin_amdgcn_sched_barrier(0);
      double sx = rcv;
      sx += d0;
      sx += (fisq ? h : 0.0) * d1;
      sx += (fisq ? h2 : h) * w0;
      dxn = sx;
    } else {
      double sx = rcv;
#pragma unroll
      for (int j = 0; j < NX; j++) sx += sAB[glr * NV + j] * sdx[j];
#pragma unroll
      for (int j = 0; j < NW; j++) sx += sAB[glr * NV + NX + j] * sdw[j];
      dxn = sx;
    }
    // (all lanes have issued their reads of sdx before this store: same wavefront, program order)
    *((k < N - 1 && lane < NX) ? sdx + lane : gdummy) = dxn;
    // (next stage's barrier orders this write before the reads)
    RST(6);
  };
  // stage 0 and the stages whose image stayed in LDS, then the stages from the gain record: stage LCAP + 1 + d (+ FD,
  // + 2 FD ...) waits in buffer d
  for (int k = 0; k < N && k <= LCAP; k++) fwd_stage(k, fvq[0], false);
  if constexpr (!SLOTS) {
    for (int k0 = LCAP + 1; k0 < N; k0 += FD) {
#pragma unroll
      for (int d = 0; d < FD; d++) {
        if (k0 + d < N) fwd_stage(k0 + d, fvq[d], true);   // (uniform branch: N and k0 are wave-uniform)
      }
    }
  } else {
    for (int k = 1; k < N; k++) fwd_stage(k, fvq[0], false);
  }
  RST_FLUSH();
  return true;
}

template <class C, int IPB>
__global__ __launch_bounds__(64 * IPB, C::RIC_WPE) void k_riccati(const DevModel M, const Ws W, const int B, const int first,
                                                const int pass) {
  // IPB wavefronts per block work on IPB consecutive list entries: neighbouring instances share
  // the 128-byte lines of the batch-minor arrays, so most of a wave's requests hit the CU's L1
  // Two instantiations are launched every pass and pick their regime from the list length:
  // the grouped one (IPB = C::IPB) while many instances iterate, the one-wave blocks (IPB = 1,
  // static LDS addresses, lowest latency) in the iteration tail.
  const int nact = *W.n_act;
  if constexpr (C::IPB > 1) {
    if ((IPB > 1) != (nact >= kGroupedMin)) return;
  }
  // lanes per instance: a whole wavefront, or half of one in the grouped regime of the small models (their
  // dense blocks have few rows: two instances per wavefront halve the LDS instructions an instance costs, and
  // LDS instruction throughput is what bounds this kernel when the whole batch iterates)
  constexpr int LPI = (IPB > 1) ? C::RIC_LPI : 64;
  constexpr int IPW = 64 / LPI;
  const int wv = threadIdx.x / LPI;   // instance slot within the block
  const int li = blockIdx.x * (IPB * IPW) + wv;
  if (li >= nact) return;
  const int b = W.act_idx[li];
  if (W.status[b] != ST_ACTIVE) return;  // uniform over the lanes of an instance (whole wavefront, or one half in the grouped regime)
  const int lane = threadIdx.x & (LPI - 1);
  const int N = M.N;
  (void)B; (void)pass;

  // ---- reduce the stage partials of the trial point --------------------------------
  Reduced r = {0, 0, 0, 0, 0, 0, 0, 0, 1e300, 0, 0};
  for (int k = lane; k < N; k += LPI) {
    r.f += W.part[IDX(P_F, k, b)];
    r.th += W.part[IDX(P_TH, k, b)];
    r.lgs += W.part[IDX(P_LOGS, k, b)];
    r.rstat = fmax(r.rstat, W.part[IDX(P_RSTAT, k, b)]);
    r.req = fmax(r.req, W.part[IDX(P_REQ, k, b)]);
    r.rineq = fmax(r.rineq, W.part[IDX(P_RINEQ, k, b)]);
    r.rcomp = fmax(r.rcomp, W.part[IDX(P_RCOMP, k, b)]);
    r.sumc += W.part[IDX(P_SUMC, k, b)];
    r.minc = fmin(r.minc, W.part[IDX(P_MINC, k, b)]);
    r.badf += W.part[IDX(P_BAD, k, b)];
    r.gphi += first ? 0.0 : W.gphi[(size_t)k * W.Bp + b];
  }
  {
    // (all quantities through the xor tree together, step by step: 6 exchange rounds instead of 11 x 6 dependent ones;
    //  the same trees as wave_sum / wave_max / wave_min)
    double rs6[6] = {r.f, r.th, r.lgs, r.sumc, r.badf, r.gphi}, rm4[4] = {r.rstat, r.req, r.rineq, r.rcomp}, rn1[1] = {r.minc};
    wave_reduce_many<LPI>(rs6, rm4, rn1);
    r.f = rs6[0]; r.th = rs6[1]; r.lgs = rs6[2]; r.sumc = rs6[3]; r.badf = rs6[4]; r.gphi = rs6[5];
    r.rstat = rm4[0]; r.req = rm4[1]; r.rineq = rm4[2]; r.rcomp = rm4[3]; r.minc = rn1[0];
  }

  // ---- decisions: every lane computes them (identical values), lane 0 stores ---------
  const bool L0 = (lane == 0);
  Inst s;
  inst_load(s, W, b);
  bool usec = false;
  const bool recurse = inst_decide<C>(M, s, r, first != 0, usec);
  if (L0) inst_store(s, W, b);   // (every lane has loaded the words above: same wavefront, program order)
  if (!recurse) return;
  const double mu = s.mu;        // nothing else of the instance state stays live across the recursion
  __shared__ double lds[IPB * IPW][RicLds<C, LPI>::LDSW];
  constexpr int IMGW = RicLds<C, LPI>::IMG_SLOTS * RicLds<C, LPI>::KPW;
  __shared__ double limg[IMGW > 0 ? IMGW : 1];   // (the arms: gain images of the first IMG_SLOTS stages, one-wavefront blocks)
  static_assert(IMGW == 0 || IPB * IPW == 1, "image slots: one instance per block");
  StepOut<gdouble> so;
  so.dz = (gdouble *)(W.dz + b); so.nunew = (gdouble *)(W.nunew + b); so.SS = (size_t)N * W.Bp; so.KS = (size_t)W.Bp;
  const double cw = usec ? (C::CSCALE ? s.theta_c : 1.0) : 0.0;
  const bool chol_ok = riccati_recursion<C, LPI, false, gdouble>(M.N, M.dt, mu, cw, lane, (ldouble *)lds[wv],
                                                                 (const gdouble *)(W.R + (size_t)b * N * C::RS),
                                                                 (gdouble *)(W.KP + (size_t)b * N * W.kps), W.kps, so,
                                                                 nullptr, (ldouble *)limg);
  if (L0) {
    // = inst_after_recursion on the stored words
    if (!chol_ok) {
      if (usec) {
        if (C::CSCALE && s.theta_c > kCsMin) {   // scaled curvature: the iteration again at half the weight
          W.theta_c[b] = 0.5 * s.theta_c; W.theta_retry[b] = 1; W.redo[b] = 1; W.usedc[b] = 0;
        } else {
          W.redo[b] = 1; W.force_gn[b] = 1; W.usedc[b] = 0;
          if constexpr (C::BACKOFF) {
            const int cb = s.curv_back ? (s.curv_back < kCurvBackMax ? 2 * s.curv_back : kCurvBackMax) : 1;
            W.curv_back[b] = cb; W.curv_skip[b] = cb;
          }
        }
      }
      else W.status[b] = -5;
    } else {
      W.usedc[b] = usec ? 1 : 0;
      W.newstep[b] = 1;
      W.amin_p[b] = (unsigned long long)__double_as_longlong(1.0);  // k_step takes the minima next
      W.amin_d[b] = (unsigned long long)__double_as_longlong(1.0);
    }
  }
}

// ===========================================================================
// k_riccati_lane: the same decisions and recursion with ONE LANE PER INSTANCE
// ===========================================================================
// The "tiny batched" layout of the recursion (round 3, review item 1a): 64 instances per wavefront, the cost-to-go,
// the dense stage block and the gains of an instance in its lane's registers, no LDS, no exchange between lanes; a
// stage is a few hundred dependent-free multiply-adds per lane.  A wavefront costs the same ~900 instructions per
// stage whether 2 or 64 of its lanes hold an instance, so the layout pays once the batch fills wavefronts that would
// otherwise each carry one instance: it is selected for lists of at least kLaneMin instances (holonomic chains with
// n <= 3; the arm's blocks do not fit a lane's registers), k_riccati's one-instance-per-wavefront blocks below that.
// Same arithmetic per entry as riccati_recursion's generic path (closed-form [A|B]^T P [A|B], Cholesky with Newton
// reciprocal square roots, symmetrised cost-to-go); the stage partials are summed in stage order instead of by a
// shuffle tree (a rounding-level difference in the merit value).
constexpr int kLaneMin = 16384;
template <class C>
__global__ __launch_bounds__(64) void k_riccati_lane(const DevModel M, const Ws W, const int B, const int first) {
  constexpr int NQ = C::NQ, NX = C::NX, NS = C::NS, NV = C::NV, NW = C::NW;
  static_assert(C::ROBOT == RMPC_ROBOT_CHAIN && NQ <= 3, "lane-per-instance recursion: small holonomic chains only");
  const int li = blockIdx.x * 64 + threadIdx.x;
  if (li >= *W.n_act) return;
  const int b = W.act_idx[li];
  if (W.status[b] != ST_ACTIVE) return;
  const int N = M.N;
  (void)B;
  Reduced r = {0, 0, 0, 0, 0, 0, 0, 0, 1e300, 0, 0};
  for (int k = 0; k < N; k++) {
    r.f += W.part[IDX(P_F, k, b)];
    r.th += W.part[IDX(P_TH, k, b)];
    r.lgs += W.part[IDX(P_LOGS, k, b)];
    r.rstat = fmax(r.rstat, W.part[IDX(P_RSTAT, k, b)]);
    r.req = fmax(r.req, W.part[IDX(P_REQ, k, b)]);
    r.rineq = fmax(r.rineq, W.part[IDX(P_RINEQ, k, b)]);
    r.rcomp = fmax(r.rcomp, W.part[IDX(P_RCOMP, k, b)]);
    r.sumc += W.part[IDX(P_SUMC, k, b)];
    r.minc = fmin(r.minc, W.part[IDX(P_MINC, k, b)]);
    r.badf += W.part[IDX(P_BAD, k, b)];
    r.gphi += first ? 0.0 : W.gphi[(size_t)k * W.Bp + b];
  }
  Inst s;
  inst_load(s, W, b);
  bool usec = false;
  const bool recurse = inst_decide<C>(M, s, r, first != 0, usec);
  inst_store(s, W, b);
  if (!recurse) return;
  const double mu = s.mu, cwt = usec ? (C::CSCALE ? s.theta_c : 1.0) : 0.0;
  const double h = M.dt, h2 = 0.5 * M.dt * M.dt;
  constexpr int NP2 = NX * (NX + 1) / 2;
  constexpr int OFF_KFF = NW * NX, OFF_PT = NW * NX + NW, OFF_P = OFF_PT + NP2, OFF_RC = OFF_P + NX;
  auto tri = [](int i, int j) __attribute__((always_inline)) {
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return lo * NX - lo * (lo - 1) / 2 + (hi - lo);
  };
  // kind of a variable (0 q, 1 v, 2 u, 3 slack) and its joint: rows of [A | B]^T are (1, 0), (h, 1), (h2, h) on the
  // (q+, v+) block rows
  auto kind = [](int i) __attribute__((always_inline)) { return i < NQ ? 0 : (i < NX ? 1 : (i >= NX + NS ? 2 : 3)); };
  auto joint = [](int i) __attribute__((always_inline)) { return i < NQ ? i : (i < NX ? i - NQ : (i >= NX + NS ? i - NX - NS : 0)); };
  double P[NX][NX], pv[NX];
#pragma unroll
  for (int i = 0; i < NX; i++) {
    pv[i] = 0.0;
#pragma unroll
    for (int j = 0; j < NX; j++) P[i][j] = 0.0;
  }
  bool chol_ok = true;
  const gdouble *const rb = (const gdouble *)(W.R + (size_t)b * N * C::RS);
  gdouble *const kpb = (gdouble *)(W.KP + (size_t)b * N * W.kps);
  for (int k = N - 1; k >= 0; k--) {
    const gdouble *const rec = rb + (size_t)k * C::RS;
    gdouble *const kpk = kpb + (size_t)k * W.kps;
    const bool rec_cost = k < N - 1;
    double rcv[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) rcv[j] = rec[C::R_RC + j];
    // ---- dense stage block Q (NV x NV) and gradient q ------------------------------------------------------
    double Q[NV][NV], q[NV];
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        double v = 0.0;
        if (hi < NQ) {
          const int t = lo * NQ - lo * (lo - 1) / 2 + (hi - lo);
          v = rec[C::R_Q + t] - cwt * (C::CURV ? (double)rec[C::R_C + t] : 0.0);
        } else if (lo == hi) {
          v = rec[C::R_DG + (lo - NQ)];
        } else if (NS > 0 && lo == NX) {
          v = rec[C::R_CS + hi];
        } else if (NS > 0 && hi == NX) {
          v = rec[C::R_CS + lo];
        }
        const int ki = kind(i), kj = kind(j);
        if (ki != 3 && kj != 3) {
          const int ii = joint(i), jj = joint(j);
          const double l1 = ki == 0 ? 1.0 : (ki == 1 ? h : h2), l2 = ki == 0 ? 0.0 : (ki == 1 ? 1.0 : h);
          const double c1 = kj == 0 ? 1.0 : (kj == 1 ? h : h2), c2 = kj == 0 ? 0.0 : (kj == 1 ? 1.0 : h);
          const double add = l1 * (c1 * P[ii][jj] + c2 * P[ii][NQ + jj]) + l2 * (c1 * P[NQ + ii][jj] + c2 * P[NQ + ii][NQ + jj]);
          v += rec_cost ? add : 0.0;
        }
        Q[i][j] = v;
      }
    double Pc[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
      double sacc = pv[i];
#pragma unroll
      for (int l = 0; l < NX; l++) sacc += P[i][l] * rcv[l];
      Pc[i] = sacc;
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      double v = rec[C::R_Q0 + i] - mu * rec[C::R_Q1 + i];
      const int ki = kind(i);
      if (ki != 3) {
        const int ii = joint(i);
        const double l1 = ki == 0 ? 1.0 : (ki == 1 ? h : h2), l2 = ki == 0 ? 0.0 : (ki == 1 ? 1.0 : h);
        const double add = l1 * Pc[ii] + l2 * Pc[NQ + ii];
        v += rec_cost ? add : 0.0;
      }
      q[i] = v;
    }
    // ---- Cholesky of Qww, gains ----------------------------------------------------------------------------------
    double L[NW][NW], invd[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) {
      double dg = Q[NX + j][NX + j];
#pragma unroll
      for (int l = 0; l < j; l++) dg -= L[j][l] * L[j][l];
      if (!(dg > 0.0)) chol_ok = false;
      double inv = __builtin_amdgcn_rsq(dg);
      inv = inv * (1.5 - 0.5 * dg * inv * inv);
      inv = inv * (1.5 - 0.5 * dg * inv * inv);
      L[j][j] = dg * inv;
      invd[j] = inv;
#pragma unroll
      for (int i = j + 1; i < NW; i++) {
        double sacc = Q[NX + i][NX + j];
#pragma unroll
        for (int l = 0; l < j; l++) sacc -= L[i][l] * L[j][l];
        L[i][j] = sacc * inv;
      }
    }
    double K[NW][NX], kff[NW];
#pragma unroll
    for (int c = 0; c < NX; c++) {
      double col[NW];
#pragma unroll
      for (int i = 0; i < NW; i++) col[i] = -Q[NX + i][c];
      chol_solve<NW>(L, invd, col);
#pragma unroll
      for (int i = 0; i < NW; i++) K[i][c] = col[i];
    }
    {
      double col[NW];
#pragma unroll
      for (int i = 0; i < NW; i++) col[i] = -q[NX + i];
      chol_solve<NW>(L, invd, col);
#pragma unroll
      for (int i = 0; i < NW; i++) kff[i] = col[i];
    }
    // ---- cost-to-go P = sym(Qxx + Qxw K), p = qx + Qxw kff ----------------------------------------------------------
    double Pa[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
#pragma unroll
      for (int j = 0; j < NX; j++) {
        double a = Q[i][j];
#pragma unroll
        for (int l = 0; l < NW; l++) a += Q[i][NX + l] * K[l][j];
        Pa[i][j] = a;
      }
      double a = q[i];
#pragma unroll
      for (int l = 0; l < NW; l++) a += Q[i][NX + l] * kff[l];
      pv[i] = a;
    }
#pragma unroll
    for (int i = 0; i < NX; i++)
#pragma unroll
      for (int j = 0; j < NX; j++) P[i][j] = 0.5 * (Pa[i][j] + Pa[j][i]);
    // ---- gain image of the stage: K | kff | P (upper triangle) | p | rc ---------------------------------------------
#pragma unroll
    for (int i = 0; i < NW; i++) {
#pragma unroll
      for (int c = 0; c < NX; c++) kpk[i * NX + c] = K[i][c];
      kpk[OFF_KFF + i] = kff[i];
    }
#pragma unroll
    for (int i = 0; i < NX; i++) {
#pragma unroll
      for (int j = i; j < NX; j++) kpk[OFF_PT + tri(i, j)] = P[i][j];
      kpk[OFF_P + i] = pv[i];
      kpk[OFF_RC + i] = rcv[i];
    }
  }
  if (chol_ok) {
    // ---- forward rollout: dw = kff + K dx, nu+ = p + P dx, dx+ = rc + [A | B][dx; dw] (closed form) --------------
    double dx[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) dx[j] = 0.0;
    const size_t SS = (size_t)N * W.Bp;
    for (int k = 0; k < N; k++) {
      const gdouble *const im = kpb + (size_t)k * W.kps;
      double dw[NW];
#pragma unroll
      for (int i = 0; i < NW; i++) {
        double sacc = im[OFF_KFF + i];
#pragma unroll
        for (int j = 0; j < NX; j++) sacc += im[i * NX + j] * dx[j];
        dw[i] = sacc;
      }
      const size_t o = (size_t)k * W.Bp + b;
#pragma unroll
      for (int j = 0; j < NX; j++) W.dz[(size_t)j * SS + o] = dx[j];
#pragma unroll
      for (int i = 0; i < NW; i++) W.dz[(size_t)(NX + i) * SS + o] = dw[i];
      if (k >= 1) {
#pragma unroll
        for (int i = 0; i < NX; i++) {
          double sacc = im[OFF_P + i];
#pragma unroll
          for (int j = 0; j < NX; j++) sacc += im[OFF_PT + tri(i, j)] * dx[j];
          W.nunew[(size_t)i * SS + o] = sacc;
        }
      }
      if (k < N - 1) {
        double dxn[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) {
          const bool isq = i < NQ;
          double sacc = im[OFF_RC + i];
          sacc += dx[i];
          sacc += (isq ? h : 0.0) * dx[isq ? NQ + i : i];
          sacc += (isq ? h2 : h) * dw[NS + (isq ? i : i - NQ)];
          dxn[i] = sacc;
        }
#pragma unroll
        for (int i = 0; i < NX; i++) dx[i] = dxn[i];
      }
    }
  }
  // = inst_after_recursion on the stored words
  if (!chol_ok) {
    if (usec) {
      if (C::CSCALE && s.theta_c > kCsMin) {   // scaled curvature: the iteration again at half the weight
        W.theta_c[b] = 0.5 * s.theta_c; W.theta_retry[b] = 1; W.redo[b] = 1; W.usedc[b] = 0;
      } else {
        W.redo[b] = 1; W.force_gn[b] = 1; W.usedc[b] = 0;
        if constexpr (C::BACKOFF) {
          const int cb = s.curv_back ? (s.curv_back < kCurvBackMax ? 2 * s.curv_back : kCurvBackMax) : 1;
          W.curv_back[b] = cb; W.curv_skip[b] = cb;
        }
      }
    }
    else W.status[b] = -5;
  } else {
    W.usedc[b] = usec ? 1 : 0;
    W.newstep[b] = 1;
    W.amin_p[b] = (unsigned long long)__double_as_longlong(1.0);
    W.amin_d[b] = (unsigned long long)__double_as_longlong(1.0);
  }
}

// ===========================================================================
// k_step: slack / multiplier steps and step-length partials, stage parallel
// ===========================================================================
// What one lane of the step kernel addresses (same convention as SweepIO).
template <class RP = gdouble>   // RP: where the step lives
struct StepIO {
  const gdouble *zc, *tc, *lc, *grow, *Jq, *gfa;
  const RP *dz;
  size_t SS;
  unsigned loff;
  size_t SSd;       // addressing of dz (see SweepIO)
  unsigned loffd;
};

// ap, ad: fraction-to-the-boundary step lengths of this stage (1 when no row binds); gphi: its merit slope partial
template <class C, class RP = gdouble, class V = RtView>
__device__ __forceinline__ void step_body(const V &v, const StepIO<RP> &io, const int k, const double mu,
                                          double &ap_out, double &ad_out, double &gphi_out) {
  constexpr int NQ = C::NQ, NX = C::NX, NS = C::NS, NV = C::NV;
  const unsigned loff = io.loff;
  const size_t SS = io.SS;
  const gdouble *__restrict__ zc = io.zc;
  const gdouble *__restrict__ tc = io.tc;
  const gdouble *__restrict__ lc = io.lc;
  const gdouble *__restrict__ grow = io.grow;
  const gdouble *__restrict__ Jq = io.Jq;
  double dz[NV], z[NV], gfv[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) {
    dz[j] = io.dz[(size_t)j * io.SSd + io.loffd];
    z[j] = zc[IDXL(j)];
    gfv[j] = io.gfa[IDXL(j)];
  }
  double gphi = 0.0;
#pragma unroll
  for (int j = 0; j < NV; j++) gphi += gfv[j] * dz[j];
  double ap = 1.0, ad = 1.0;
  auto row = [&](int i, double gdz, double g, double tv, double lv) __attribute__((always_inline)) {
    const double dt = gdz + (g - tv);
    const double itv = frcp(tv);
    const double dl = (mu - tv * lv - lv * dt) * itv;   // (same expression as in sweep_body's row_core)
    (void)i;  // the steps themselves are not stored: the sweep recomputes them from the same inputs
    // ratio tests with Newton reciprocals (the quotient of a non-negative step is discarded by the select)
    const double rp = -C::TAU * tv * frcp(dt), rd = -C::TAU * lv * frcp(dl);
    // (bitwise and: no short-circuit branch -- the rows of a stage stay one basic block)
    ap = ((dt < 0) & (rp < ap)) ? rp : ap;
    ad = ((dl < 0) & (rd < ad)) ? rd : ad;
    gphi -= mu * dt * itv;
  };
  // Every request of the phase leaves before the first row is evaluated (one wavefront per SIMD hides no latency by
  // itself; left where the arithmetic is, the compiler waits for each small group of loads in turn: a dozen round
  // trips to L2 per call instead of one).  The rows are then evaluated in the old order (the merit slope is a sum).
  struct FkIn { double g, tv, lv, jq[NQ]; };
  auto fk_load = [&](const int r, FkIn &f) __attribute__((always_inline)) {
    const int i = v.fk_row(r), fi = v.fk_idx(r);
    f.g = grow[IDXL(i)]; f.tv = tc[IDXL(i)]; f.lv = lc[IDXL(i)];
#pragma unroll
    for (int a = 0; a < NQ; a++) f.jq[a] = Jq[IDXL(fi * NQ + a)];
  };
  auto fk_row_body = [&](const int r, const FkIn &f) __attribute__((always_inline)) {
    double gdz = 0.0;
#pragma unroll
    for (int a = 0; a < NQ; a++) gdz += f.jq[a] * dz[a];
    if constexpr (NS > 0) gdz += dz[NX];
    row(v.fk_row(r), gdz, f.g, f.tv, f.lv);
  };
  constexpr int NFKC = []() { if constexpr (V::SPEC) return V::nfkrows() > 0 ? V::nfkrows() : 1; else return 1; }();
  FkIn fkin[NFKC];
  if constexpr (V::SPEC) {
    for_range<0, V::nfkrows()>([&](auto rc) __attribute__((always_inline)) { fk_load(decltype(rc)::value, fkin[decltype(rc)::value]); });
  }
  // single-variable rows, by variable (unconditional clamped requests, see sweep_body), in chunks of VCH variables whose
  // requests leave together: all of them for the small models, one variable at a time for the arms (12 requests per
  // variable: more in flight cost the arm's kernel registers it does not have -- k_step 30 -> 33 us with six)
  constexpr int VCH = NV <= 12 ? NV : 1;
  double tvv[VCH][kVarRows], lvv[VCH][kVarRows], glv[VCH][kVarRows];
  auto chunk_load = [&](auto c0c) __attribute__((always_inline)) {
    constexpr int c0 = decltype(c0c)::value;
#pragma unroll
    for (int jj = 0; jj < VCH; jj++) {
      const int j = c0 + jj < NV ? c0 + jj : NV - 1;
#pragma unroll
      for (int u = 0; u < kVarRows; u++) {
        const int i = v.v_row(j, u);
        const int ii = i >= 0 ? i : 0;
        const bool general = v.v_poff(j, u) >= 0;
        tvv[jj][u] = tc[IDXL(ii)];
        lvv[jj][u] = lc[IDXL(ii)];
        glv[jj][u] = grow[IDXL(general ? ii : 0)];
      }
    }
  };
  auto chunk_rows = [&](auto c0c) __attribute__((always_inline)) {
    constexpr int c0 = decltype(c0c)::value;
#pragma unroll
    for (int jj = 0; jj < VCH; jj++) {
      const int j = c0 + jj;
      if (j >= NV) continue;
#pragma unroll
      for (int u = 0; u < kVarRows; u++) {
        const int i = v.v_row(j, u);
        if (i < 0) continue;
        const bool general = v.v_poff(j, u) >= 0;
        const double gvv = general ? glv[jj][u] : ((k == 0 && j < NX) ? 1.0 : (double)v.v_sgn(j, u) * (z[j] - v.v_val(j, u)));
        double gdz = (double)v.v_sgn(j, u) * dz[j];
        if constexpr (NS > 0) { if (v.v_soft(j, u)) gdz += dz[NX]; }
        row(i, gdz, gvv, tvv[jj][u], lvv[jj][u]);
      }
    }
  };
  chunk_load(std::integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (V::SPEC) {
    for_range<0, V::nfkrows()>([&](auto rc) __attribute__((always_inline)) { fk_row_body(decltype(rc)::value, fkin[decltype(rc)::value]); });
  } else {
    // (runtime tables: four rows' requests at a time, clamped to the last row; the row count is uniform)
    const int nfk = v.nfkrows();
    constexpr int FCH = NV <= 12 ? 4 : 1;
    for (int r0 = 0; r0 < nfk; r0 += FCH) {
      FkIn f4[FCH];
#pragma unroll
      for (int u = 0; u < FCH; u++) fk_load(r0 + u < nfk ? r0 + u : nfk - 1, f4[u]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < FCH; u++)
        if (r0 + u < nfk) fk_row_body(r0 + u, f4[u]);
    }
  }
  chunk_rows(std::integral_constant<int, 0>{});
  for_range<1, (NV + VCH - 1) / VCH>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c0 = decltype(cc)::value * VCH;
    chunk_load(std::integral_constant<int, c0>{});
    __builtin_amdgcn_sched_barrier(0);
    chunk_rows(std::integral_constant<int, c0>{});
  });
  ap_out = ap; ad_out = ad; gphi_out = gphi;
}

template <class C, class V>
__global__ __launch_bounds__(kSweepBlock) void k_step(const DevModel M, const DevTables *__restrict__ Tp, const Ws W,
                                              const int B) {
  const int gid = blockIdx.x * kSweepBlock + threadIdx.x;
  const int li = gid % W.Bp;
  const int k = __builtin_amdgcn_readfirstlane(gid / W.Bp);   // (uniform per wavefront, see k_sweep)
  if (li >= *W.n_act || k >= M.N) return;
  const int b = W.act_idx[li];
  if (W.status[b] != ST_ACTIVE || !W.newstep[b]) return;
  (void)B;
  const int cur = W.cur[b];
  StepIO<gdouble> io;
  io.zc = (gdouble *)W.z[cur]; io.tc = (gdouble *)W.t[cur]; io.lc = (gdouble *)W.lam[cur]; io.grow = (gdouble *)W.grow[cur];
  io.Jq = (gdouble *)W.Jq[cur]; io.dz = (gdouble *)W.dz; io.gfa = (gdouble *)W.gfa;
  io.SS = (size_t)M.N * W.Bp;
  io.loff = (unsigned)k * (unsigned)W.Bp + (unsigned)b;
  io.SSd = io.SS; io.loffd = io.loff;
  double ap, ad, gphi;
  const V v(M, *Tp);
  step_body<C, gdouble, V>(v, io, k, W.mu[b], ap, ad, gphi);
  // partial minima -> per-instance step lengths (min is order independent: deterministic)
  atomicMin(&W.amin_p[b], (unsigned long long)__double_as_longlong(ap));
  atomicMin(&W.amin_d[b], (unsigned long long)__double_as_longlong(ad));
  W.gphi[(size_t)k * W.Bp + b] = gphi;
}


// ===========================================================================
// k_fused: whole interior-point iterations of an instance inside ONE wavefront
// ===========================================================================
// The pass kernels above run the batch in lock step: every pass is four launches, every launch streams the
// whole iterate through HBM, and the last few stragglers of a batch cost a full launch chain per iteration.
// Here a wavefront OWNS two instances (32 lanes each, lane = stage) from the first sweep to the converged
// plan: sweep -> reduction (shuffles) -> decisions (registers) -> Riccati recursion (the instance's 32 lanes,
// stage blocks in LDS) -> step lengths (shuffles) -> next sweep, with no kernel boundary, no host look and no
// other wavefront involved.  An instance's state lives in its own contiguous block of the workspace
// ([instance][slot][32 stages]: a half-wavefront moves 256 contiguous bytes per slot), which only its owner
// touches, so it is served by the XCD's L2 / the Infinity Cache; stage records go through LDS (point robot)
// and the per-instance solver words through registers.  Results are bit-identical to the pass kernels: the
// same sweep_body / step_body / inst_decide / riccati_recursion run, and the reductions use the same trees.
// Blocks are independent and of one wavefront: the dispatcher backfills a CU as soon as a pair finishes.
// the sweep and the step phase of the runtime-table models as real functions (GView); 0: inlined into k_fused (round 3)
#ifndef RMPC_FUSED_CALLS
#define RMPC_FUSED_CALLS 1
#endif
constexpr bool kFusedCalls = RMPC_FUSED_CALLS != 0;
constexpr int kFusedStages = 32;   // stage stride of the per-instance layout = lanes per instance

struct FusedWs {
  double *p;                      // [B][npar][32]
  double *z[2], *t[2], *lam[2], *nu[2];
  double *dz, *nunew, *gfa;
  double *grow[2], *Jq[2];
  double *wlam, *wnu, *wmu;       // [B][m][32], [B][nx][32], [B]: multipliers of the last solve (warm start)
  double *R;                      // [B][N][rs]   (models whose records do not fit LDS)
  double *KP;                     // [B][N][kps]
  int *passes;                    // [0] most passes any instance of the last launch needed, [1] the launch's queue counter
  int *lastp;                     // [B] passes of every instance in the last launch
  int *order;                     // [B] launch order of the next warm-started launch: instances by lastp, longest first
  int *ckey;                      // [B] launch-order keys of a cold launch (k_difficulty)
  long long *stamps;              // [blocks][8] cycles per phase (builds with -DRMPC_STAMPS only; development aid)
  int rs, kps, nv, m, nx, npar, nhs, njqs;
};

// ordering point for data one lane writes to the workspace and another lane of the same wavefront reads later
#define GSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_s_waitcnt(0);                           \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// The phases of the fused kernel are real functions, not inlined bodies: one 500-register function with the sweep,
// the recursion and the step phase inside lets the register allocator spill the loop-carried values of the
// recursion's stage loop to make room for the sweep's straight-line code (measured with the generated views:
// recursion 83 k -> 125 k cycles per pass).  As callees every phase gets the whole register file to itself and the
// few words that live across a call are saved once around it.
#define RMPC_ONE_WAVE   // (occupancy attributes are kernel-only in clang: the phase functions inherit k_fused's, see there)
#ifdef RMPC_NOINLINE_OFF
#define RMPC_PHASE __forceinline__
#else
#define RMPC_PHASE __noinline__ RMPC_ONE_WAVE
#endif
template <class C>
__device__ RMPC_PHASE bool fused_recursion_lds(const int N, const double dt, const double mu, const double cw, const int lane,
                                               ldouble *const work, ldouble *const slots, const StepOut<ldouble> so) {
  return riccati_recursion<C, kFusedStages, true, ldouble>(N, dt, mu, cw, lane, work, slots, nullptr, 0, so, slots);
}
template <class C>
__device__ RMPC_PHASE bool fused_recursion_mem(const int N, const double dt, const double mu, const double cw, const int lane,
                                               ldouble *const work, const gdouble *const grec, gdouble *const kpb,
                                               const int kps, const StepOut<gdouble> so) {
  return riccati_recursion<C, kFusedStages, false, gdouble, true>(N, dt, mu, cw, lane, work, grec, kpb, kps, so);
}

// Bases of an instance's block in every array of the fused workspace.  They are recomputed from the instance index
// where a phase needs them (a handful of integer operations) instead of living in registers across the phase calls.
struct FusedPtrs {
  gdouble *pz[2], *pt[2], *pl[2], *pn[2], *pg[2], *pj[2], *pp, *pdz, *pnn, *pgf, *pwl, *pwn;
};
__device__ __forceinline__ FusedPtrs fused_ptrs(const FusedWs &F, size_t b) {
  asm volatile("" : "+v"(b));   // opaque: the bases must not be hoisted out of the pass loop (and spilled there)
  const size_t S = kFusedStages;
  FusedPtrs P;
  P.pz[0] = (gdouble *)F.z[0] + b * F.nv * S; P.pz[1] = (gdouble *)F.z[1] + b * F.nv * S;
  P.pt[0] = (gdouble *)F.t[0] + b * F.m * S; P.pt[1] = (gdouble *)F.t[1] + b * F.m * S;
  P.pl[0] = (gdouble *)F.lam[0] + b * F.m * S; P.pl[1] = (gdouble *)F.lam[1] + b * F.m * S;
  P.pn[0] = (gdouble *)F.nu[0] + b * F.nx * S; P.pn[1] = (gdouble *)F.nu[1] + b * F.nx * S;
  P.pg[0] = (gdouble *)F.grow[0] + b * F.nhs * S; P.pg[1] = (gdouble *)F.grow[1] + b * F.nhs * S;
  P.pj[0] = (gdouble *)F.Jq[0] + b * F.njqs * S; P.pj[1] = (gdouble *)F.Jq[1] + b * F.njqs * S;
  P.pp = (gdouble *)F.p + b * F.npar * S;
  P.pdz = (gdouble *)F.dz + b * F.nv * S;
  P.pnn = (gdouble *)F.nunew + b * F.nx * S;
  P.pgf = (gdouble *)F.gfa + b * F.nv * S;
  P.pwl = (gdouble *)F.wlam + b * F.m * S;
  P.pwn = (gdouble *)F.wnu + b * F.nx * S;
  return P;
}

// The callees of the fused kernel get the pointer block's address as an ordinary (vector register) argument.  Read
// through it as it is, the block came in by eleven vector loads and a full wait before the first useful request of
// the phase, and picking the current / next buffers of an array pair by a run-time index sent the pairs through scratch
// (store, wait, indexed load: a second round trip).  The address is the same in every lane: as a scalar in the constant
// address space the block arrives by scalar loads, and the buffers are picked by selects.
typedef const __attribute__((address_space(4))) FusedWs cFusedWs;
__device__ __forceinline__ cFusedWs *uniform_block(const FusedWs *p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (cFusedWs *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void load_block(FusedWs &F, const FusedWs *p) {
  static_assert(sizeof(FusedWs) % 8 == 0, "FusedWs is copied in 8-byte words");
  const __attribute__((address_space(4))) unsigned long long *src = (const __attribute__((address_space(4))) unsigned long long *)uniform_block(p);
  unsigned long long *dst = (unsigned long long *)&F;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(FusedWs) / 8); i++) dst[i] = src[i];
}
struct FusedCur {   // an instance's bases with the current / next buffers resolved
  gdouble *zc, *zn, *tc, *tn, *lc, *ln, *nc, *nn, *gc, *gn, *jc, *jn, *pp, *pdz, *pnn, *pgf, *pwl, *pwn;
};
__device__ __forceinline__ FusedCur fused_cur(const FusedWs &F, const size_t b, const int cur) {
  const FusedPtrs P = fused_ptrs(F, b);
  const bool c1 = cur != 0;
  FusedCur Q;
  Q.zc = c1 ? P.pz[1] : P.pz[0]; Q.zn = c1 ? P.pz[0] : P.pz[1];
  Q.tc = c1 ? P.pt[1] : P.pt[0]; Q.tn = c1 ? P.pt[0] : P.pt[1];
  Q.lc = c1 ? P.pl[1] : P.pl[0]; Q.ln = c1 ? P.pl[0] : P.pl[1];
  Q.nc = c1 ? P.pn[1] : P.pn[0]; Q.nn = c1 ? P.pn[0] : P.pn[1];
  Q.gc = c1 ? P.pg[1] : P.pg[0]; Q.gn = c1 ? P.pg[0] : P.pg[1];
  Q.jc = c1 ? P.pj[1] : P.pj[0]; Q.jn = c1 ? P.pj[0] : P.pj[1];
  Q.pp = P.pp; Q.pdz = P.pdz; Q.pnn = P.pnn; Q.pgf = P.pgf; Q.pwl = P.pwl; Q.pwn = P.pwn;
  return Q;
}

// The sweep and the step phase are real functions for the generated views only: with the runtime tables they would
// need the model and the tables through memory instead of through the scalar registers of the kernel.  A call takes
// a handful of scalars -- the callee derives the instance's bases from the pointer block in device memory (scalar
// loads) -- and returns its results by value: with the SweepIO / StepIO structs as arguments and the partials behind
// a reference, the argument and result traffic through scratch was 1.4 KB per lane and pass, more than the 0.9 KB
// the sweep stores by design (round 2, L2 counters: 60 % of the fabric traffic of a launch were writes).
// behind the row tables in device memory: the workspace block, then a copy of the model (rmpc_create)
struct ArmBlock {
  FusedWs F;
  DevModel M;
};
// The view a phase FUNCTION reads the problem's structure through: a generated view is a set of constants; the runtime
// tables come through uniform pointers in the constant address space (GView: tables in front of the pointer block,
// the model's copy behind it), i.e. by scalar loads -- round 4: with that the sweep and the step phase of the models
// WITHOUT a generated view (the boxer, the weighted / 2-joint chains) are real functions as well, each with the register
// file to itself, and their requests can leave ahead of the arithmetic (PIPE in sweep_body).
template <class V>
__device__ __forceinline__ V call_view(const FusedWs *Fp) {
  if constexpr (std::is_same<V, GView>::value) {
    const unsigned long long a = (unsigned long long)Fp;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long u = ((unsigned long long)hi << 32) | lo;
    typedef const __attribute__((address_space(4))) ArmBlock cArmBlock;
    cArmBlock *const blk = (cArmBlock *)u;
    return GView(&blk->M, (GView::cTables *)(u - sizeof(DevTables)));
  } else {
    return V{};
  }
}
struct StepRes { double ap, ad, gp; };
template <class C, class V, int FIRSTC, bool REC_LDS>
__device__ __noinline__ RMPC_ONE_WAVE Partials fused_sweep_call(const FusedWs *Fp, const int N, const double dt, const int use_curv,
                                                  const size_t b, const int cur, const int k, ldouble *const slots,
                                                  const bool nostep, const double alpha, const double adual, const double mu,
                                                  const int warm) {
  using RP = typename std::conditional<REC_LDS, ldouble, gdouble>::type;
  constexpr int GS = FusedSlots<C>::GS, DZ_OFF = FusedSlots<C>::DZ_OFF, NV = C::NV;
  FusedWs F;
  load_block(F, Fp);   // (scalar loads: uniform address, constant address space)
  const size_t S = kFusedStages;
  const FusedCur Pw = fused_cur(F, b, cur);
  SweepIO<RP> io;
  io.zc = Pw.zc; io.tc = Pw.tc; io.lc = Pw.lc; io.nc = Pw.nc;
  io.zn = Pw.zn; io.tn = Pw.tn; io.ln = Pw.ln; io.nn = Pw.nn;
  io.pp = Pw.pp; io.gro = Pw.gc; io.jqo = Pw.jc; io.grn = Pw.gn; io.jqn = Pw.jn;
  io.gfa = Pw.pgf;
  io.SS = S; io.loff = (unsigned)k; io.kstride = 1u;
  if constexpr (REC_LDS) {
    io.rec = slots + k * GS;
    io.dzp = slots + DZ_OFF; io.nup = slots + DZ_OFF + NV;
    io.SSd = 1; io.loffd = (unsigned)(k * GS); io.kstrided = (unsigned)GS;
  } else {
    io.rec = (gdouble *)F.R + (b * (size_t)N + k) * C::RS;
    io.dzp = Pw.pdz; io.nup = Pw.pnn;
    io.SSd = S; io.loffd = (unsigned)k; io.kstrided = 1u;
  }
  io.wl = Pw.pwl; io.wn = Pw.pwn; io.warm = warm;
  const SweepK sk = {N, dt, use_curv};
  const V v = call_view<V>(Fp);
  Partials q = {0, 0, 0, 0, 0, 0, 0, 0, 1e300, 0};
  sweep_body<C, -1, RP, V, FIRSTC>(sk, v, io, k, FIRSTC != 0, nostep, alpha, adual, mu, q);
  return q;
}
template <class C, class V, bool REC_LDS>
__device__ __noinline__ RMPC_ONE_WAVE StepRes fused_step_call(const FusedWs *Fp, const size_t b, const int cur, const int k,
                                                ldouble *const slots, const double mu) {
  using RP = typename std::conditional<REC_LDS, ldouble, gdouble>::type;
  constexpr int GS = FusedSlots<C>::GS, DZ_OFF = FusedSlots<C>::DZ_OFF;
  FusedWs F;
  load_block(F, Fp);   // (scalar loads: uniform address, constant address space)
  const size_t S = kFusedStages;
  const FusedCur Ps = fused_cur(F, b, cur);
  StepIO<RP> io;
  io.zc = Ps.zc; io.tc = Ps.tc; io.lc = Ps.lc; io.grow = Ps.gc; io.Jq = Ps.jc;
  io.gfa = Ps.pgf;
  io.SS = S; io.loff = (unsigned)k;
  if constexpr (REC_LDS) { io.dz = slots + DZ_OFF; io.SSd = 1; io.loffd = (unsigned)(k * GS); }
  else { io.dz = Ps.pdz; io.SSd = S; io.loffd = (unsigned)k; }
  const V v = call_view<V>(Fp);
  StepRes r = {1.0, 1.0, 0.0};
  step_body<C, RP, V>(v, io, k, mu, r.ap, r.ad, r.gp);
  return r;
}

// Generated views, records in LDS: the step lengths of a fresh step are formed at the beginning of the sweep call
// instead of after the recursion -- the whole wavefront calls (the reductions over the 32
// lanes of the instance run inside), lanes without work skip the bodies.  The slacks and multipliers the step phase
// reads are then read again by the sweep a few thousand cycles later (L2) instead of a whole recursion later (fabric),
// and a pass is two calls: this one and the recursion (1.90-1.94 -> 1.97-2.03 M solves/s, same results).
// What the call hands back, per instance (identical in the 32 lanes of a half: the reductions over the stages run
// inside the call): the reduced partials of the sweep and the step lengths.  Through LDS, not by value -- an
// aggregate of this size is returned in memory, i.e. through scratch: a store, a full wait before the return, and a
// load plus wait in the caller, per pass.
struct SweepStepOut { double f, th, lgs, sumc, badf, rstat, req, rineq, rcomp, minc, amin_p, amin_d, gphi;
#ifdef RMPC_STAMPS
  long long tk[6];
#endif
};
struct SweepStepRes { Partials q; double amin_p, amin_d, gphi; };
template <class C, class V, int FIRSTC>
__device__ __noinline__ RMPC_ONE_WAVE void fused_sweep_step_call(__attribute__((address_space(3))) SweepStepOut *const out,
                                                                 const FusedWs *Fp, const int N, const double dt, const int use_curv,
                                                           const size_t b, const int cur, const int k, ldouble *const slots,
                                                           const bool live, const bool nostep, const bool fresh, const int ls,
                                                           const double amin_p_in, const double amin_d_in, const double gphi_in,
                                                           const double mu, const int warm) {
  constexpr int GS = FusedSlots<C>::GS, DZ_OFF = FusedSlots<C>::DZ_OFF, NV = C::NV;
  FusedWs F;
  load_block(F, Fp);   // (scalar loads: uniform address, constant address space)
  const size_t S = kFusedStages;
  const FusedCur Pw = fused_cur(F, b, cur);
  const V v{};
  double ap = 1.0, ad = 1.0, gp = 0.0;
#ifdef RMPC_STAMPS
  const long long ss_t0 = __builtin_amdgcn_s_memtime();
#endif
  if (fresh && live) {
    StepIO<ldouble> io;
    io.zc = Pw.zc; io.tc = Pw.tc; io.lc = Pw.lc; io.grow = Pw.gc; io.Jq = Pw.jc;
    io.gfa = Pw.pgf;
    io.SS = S; io.loff = (unsigned)k;
    io.dz = slots + DZ_OFF; io.SSd = 1; io.loffd = (unsigned)(k * GS);
    step_body<C, ldouble, V>(v, io, k, mu, ap, ad, gp);
  }
#ifdef RMPC_STAMPS
  const long long ss_t1 = __builtin_amdgcn_s_memtime();
#endif
  {
    double rs1[1] = {gp}, rm0[1] = {0.0}, rn2[2] = {ap, ad};
    wave_reduce_many<kFusedStages>(rs1, rm0, rn2);
    gp = rs1[0]; ap = rn2[0]; ad = rn2[1];
  }
#ifdef RMPC_STAMPS
  const long long ss_t2 = __builtin_amdgcn_s_memtime();
#endif
  SweepStepRes r;
  r.amin_p = fresh ? fmin(amin_p_in, ap) : amin_p_in;
  r.amin_d = fresh ? fmin(amin_d_in, ad) : amin_d_in;
  r.gphi = fresh ? gp : gphi_in;
  const double alpha = nostep ? 0.0 : ldexp(r.amin_p, -ls), adual = nostep ? 0.0 : r.amin_d;
  const Partials qn = {0, 0, 0, 0, 0, 0, 0, 0, 1e300, 0};
  r.q = qn;
  if (live) {
    SweepIO<ldouble> io;
    io.zc = Pw.zc; io.tc = Pw.tc; io.lc = Pw.lc; io.nc = Pw.nc;
    io.zn = Pw.zn; io.tn = Pw.tn; io.ln = Pw.ln; io.nn = Pw.nn;
    io.pp = Pw.pp; io.gro = Pw.gc; io.jqo = Pw.jc; io.grn = Pw.gn; io.jqn = Pw.jn;
    io.gfa = Pw.pgf;
    io.SS = S; io.loff = (unsigned)k; io.kstride = 1u;
    io.rec = slots + k * GS;
    io.dzp = slots + DZ_OFF; io.nup = slots + DZ_OFF + NV;
    io.SSd = 1; io.loffd = (unsigned)(k * GS); io.kstrided = (unsigned)GS;
    io.wl = Pw.pwl; io.wn = Pw.pwn; io.warm = warm;
    const SweepK sk = {N, dt, use_curv};
    sweep_body<C, -1, ldouble, V, FIRSTC>(sk, v, io, k, FIRSTC != 0, nostep, alpha, adual, mu, r.q);
  }
#ifdef RMPC_STAMPS
  r.q.tk[4] = ss_t1 - ss_t0; r.q.tk[5] = ss_t2 - ss_t1;
#endif
  {
    // (idle lanes and idle halves contribute the neutral elements: their sums are discarded by the caller)
    const Partials &q = r.q;
    double rs5[5] = {q.f, q.th, q.logs, q.sumc, q.bad}, rm4[4] = {q.rstat, q.req, q.rineq, q.rcomp}, rn1[1] = {q.minc};
    wave_reduce_many<kFusedStages>(rs5, rm4, rn1);
    // (every lane of the half stores the same words: no divergence, one LDS request each)
    out->f = rs5[0]; out->th = rs5[1]; out->lgs = rs5[2]; out->sumc = rs5[3]; out->badf = rs5[4];
    out->rstat = rm4[0]; out->req = rm4[1]; out->rineq = rm4[2]; out->rcomp = rm4[3]; out->minc = rn1[0];
    out->amin_p = r.amin_p; out->amin_d = r.amin_d; out->gphi = r.gphi;
#ifdef RMPC_STAMPS
    if (k == 0) { for (int i = 0; i < 6; i++) out->tk[i] = q.tk[i]; }
#endif
  }
}

// (disable_tail_calls: a phase call that hands the callee nothing of the caller's stack gets the `tail` marker, and a
//  function with a tail-marked call site is not eligible for the no-callee-saved-registers optimisation of internal
//  functions: the sweep call then saved and restored 300 registers through scratch on every pass.)
// (amdgpu_waves_per_eu(1, 1): __launch_bounds__' second argument only sets the MINIMUM of waves per SIMD; with the
//  maximum open the instruction scheduler still plans the phase functions -- which inherit the attribute -- for as
//  many waves as it can reach and keeps their register pressure down by serialising the LDS reads of a phase:
//  load, wait, use, load, wait, use.  One wave per SIMD is what the kernel gets anyway: 38 KB of LDS.)
template <class C, bool REC_LDS, class V>
__global__ __launch_bounds__(64, 1) __attribute__((amdgpu_waves_per_eu(1, 1), disable_tail_calls)) void k_fused(const DevModel M, const DevTables *__restrict__ Tp, const FusedWs F, const int B,
                                              const double *__restrict__ xinit, const double *__restrict__ x0,
                                              const double *__restrict__ params, double *__restrict__ zout,
                                              int *__restrict__ exitflag, int *__restrict__ iters_out,
                                              double *__restrict__ kkt, double *__restrict__ obj, const int max_passes,
                                              const int warm_mode, const int use_order) {
  constexpr int LPI = kFusedStages;
  constexpr int IPW = 2;   // instances per wavefront
  constexpr int NX = C::NX, NV = C::NV;
  using VC = typename std::conditional<V::SPEC, V, GView>::type;   // the view of the phase functions
  const V v(M, *Tp);
  const int half = threadIdx.x / LPI;
  const int k = threadIdx.x & (LPI - 1);       // stage of this lane; also its lane index inside the instance
  // The launch is a queue of instances, not a grid of pairs: a half-wavefront takes instance after instance until the
  // queue is empty (its first one by its position in the grid, the following ones from an atomic counter), so a
  // finished instance never holds its 32 lanes until its partner has finished too, and the grid is no larger than the
  // chip.  In a closed loop (use_order) the queue holds the instances in the order of their previous solve's passes,
  // longest first (k_order): longest-processing-time-first scheduling.  The arithmetic of an instance depends neither
  // on its position in the queue nor on its partner.
  const int N = M.N;
  const bool stage = k < N;

  // LDS of an instance: the work area of the recursion, and (REC_LDS) its 32 stage slots (FusedSlots)
  constexpr int LW = RicLds<C, LPI>::LDSW;
  constexpr int GS = FusedSlots<C>::GS;
  constexpr int DZ_OFF = FusedSlots<C>::DZ_OFF;
  constexpr int RECW = REC_LDS ? kFusedStages * GS : 0;
  __shared__ double lds[IPW * (LW + RECW)];
  ldouble *const work = (ldouble *)lds + half * (LW + RECW);
  ldouble *const slots = work + LW;

  // ---- per-instance bases of the workspace: fused_ptrs(F, b) where a phase needs them ---------------------
  const size_t S = kFusedStages;
  // the solver words of the two instances are parked here around the phase calls (the callees own the register file)
  __shared__ Inst sinst[IPW];
  __shared__ SweepStepOut sres[IPW];   // what the sweep call hands back (generated views with LDS records)
  Inst s;
  const bool warm = warm_mode != 0;
  // (every lane of an instance holds the same words: its lane 0 parks them, all lanes take them back)
  auto park = [&]() __attribute__((always_inline)) { if (k == 0) sinst[half] = s; };
  // (lane 0's store and the other lanes' loads are ordered by the wavefront fence: without it the compiler may
  //  keep a lane's copy from the previous unpark -- nothing in that lane's own program wrote the words since)
  auto unpark = [&]() __attribute__((always_inline)) { WSYNC(); s = sinst[half]; };
  inst_init(s, M.mu0);
  s.status = 0;                 // (no instance yet)
  size_t b = (size_t)(B - 1);   // instance of this half (none: clamped -- addresses stay legal, nothing is written)
  bool valid = false;           // the half holds an instance
  bool retired = false;         // the queue was empty when the half asked: it stays idle
  bool first = true;            // the instance's next pass is its first
  int ipass = 0;                // passes of the instance so far
  int nextslot = blockIdx.x * IPW + half;   // queue position of the half's first instance (-1: ask the counter)
  int *const qhead = F.passes + 1;          // positions handed out beyond the grid's own (zeroed before the launch)
  double gphi_sum = 0.0;   // merit slope of the current step (sum over the stages; step phase)

#ifdef RMPC_STAMPS
  long long st_sweep = 0, st_dec = 0, st_ric = 0, st_step = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_a, st_b;
  long long st_sw[6] = {0, 0, 0, 0, 0, 0}, st_sw2[2] = {0, 0};
  int st_ipass = 0;   // instance passes of this wavefront (both halves)
#define STAMP_A() st_a = __builtin_amdgcn_s_memtime()
#define STAMP_B(acc) do { st_b = __builtin_amdgcn_s_memtime(); acc += st_b - st_a; st_a = st_b; } while (0)
#else
#define STAMP_A()
#define STAMP_B(acc)
#endif
  int pass = 0;   // passes of the wavefront
  for (;; pass++) {
    // ---- finished instances leave, idle halves take the next instance of the queue -----------------------------------
    {
      const bool over = valid && (s.status == ST_ACTIVE) && ipass >= max_passes;   // deadline (rmpc_set_pass_budget) or cap
      const bool done = valid && (s.status != ST_ACTIVE || over);
      if (__ballot(done || (!valid && !retired)) != 0ull) {
        if (done) {
          // epilogue: plan in the ABI layout, statistics (the trial point and the step were made visible to the whole
          // wavefront by the ordering points of the pass that ended the solve)
          const FusedPtrs Pe = fused_ptrs(F, b);
          if (stage) {
            const gdouble *zf = Pe.pz[s.cur];
            double *zr = zout + (b * N + k) * NV;
#pragma unroll
            for (int j = 0; j < NV; j++) zr[j] = zf[j * S + k];
            // multipliers for a warm start of the next solve of this instance (a failed solve leaves zeros and mu0)
            const bool okd = (s.status == ST_ACTIVE || s.status >= 0) && isfinite(s.mu) && s.mu > 0.0;
            const gdouble *lf = Pe.pl[s.cur], *nf = Pe.pn[s.cur];
            {
              int i = 0;
              for (; i + 8 <= F.m; i += 8) {   // (eight requests in flight, as for the parameters)
                double lv8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) lv8[u] = lf[(i + u) * S + k];
#pragma unroll
                for (int u = 0; u < 8; u++) Pe.pwl[(i + u) * S + k] = okd ? lv8[u] : 0.0;
              }
              for (; i < F.m; i++) Pe.pwl[i * S + k] = okd ? lf[i * S + k] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < NX; j++) Pe.pwn[j * S + k] = okd ? nf[j * S + k] : 0.0;
          }
          if (k == 0) {
            exitflag[b] = (s.status == ST_ACTIVE) ? 0 : s.status;
            iters_out[b] = s.iters;
            kkt[b] = fmax(fmax(s.res_stat, s.res_eq), fmax(s.res_ineq, s.res_comp));
            obj[b] = s.obj;
            F.wmu[b] = ((s.status == ST_ACTIVE || s.status >= 0) && isfinite(s.mu) && s.mu > 0.0) ? s.mu : M.mu0;
            F.lastp[b] = ipass;
            atomicMax(F.passes, ipass);
          }
          valid = false;
          s.status = 0;
        }
        if (!valid && !retired) {
          int pos = nextslot;
          nextslot = -1;
          if (pos < 0) {
            int t = 0;
            if (k == 0) t = atomicAdd(qhead, 1);
            pos = (int)gridDim.x * IPW + __shfl(t, half * LPI, 64);
          }
          if (pos < B) {
            b = (size_t)(use_order ? F.order[pos] : pos);
            valid = true;
            // prologue: ABI rows of this stage -> the instance's block (x_1 := xinit, mpcModel.py:108)
            const FusedPtrs P0 = fused_ptrs(F, b);
            if (stage) {
              const double *zr = x0 + (b * N + k) * NV;
#pragma unroll
              for (int j = 0; j < NV; j++) {
                double v = zr[j];
                if (k == 0 && j < NX) v = xinit[b * NX + j];
                P0.pz[0][j * S + k] = v;
              }
              if constexpr (REC_LDS) {   // the step slots are read (and discarded) by the first sweep: keep them finite
#pragma unroll
                for (int j = 0; j < NV + NX; j++) slots[k * GS + DZ_OFF + j] = 0.0;
              }
              if (params) {
                // (eight requests in flight: one by one the copy is npar dependent round trips to memory)
                const double *pr = params + (b * N + k) * M.npar;
                int j = 0;
                for (; j + 8 <= M.npar; j += 8) {
                  double pv8[8];
#pragma unroll
                  for (int u = 0; u < 8; u++) pv8[u] = pr[j + u];
#pragma unroll
                  for (int u = 0; u < 8; u++) P0.pp[(j + u) * S + k] = pv8[u];
                }
                for (; j < M.npar; j++) P0.pp[j * S + k] = pr[j];
              }
            }
            inst_init(s, warm ? warm_mu(F.wmu[b], M.mu0) : M.mu0);
            first = true;
            ipass = 0;
            gphi_sum = 0.0;
          } else {
            retired = true;
            b = (size_t)(B - 1);
          }
        }
        GSYNC();   // the new instance's block is complete before any lane reads another lane's part
      }
    }
    const bool act = valid && (s.status == ST_ACTIVE);
    if (__ballot(act) == 0ull) break;   // both halves are idle and the queue is empty
    if (act) ipass++;
#ifdef RMPC_STAMPS
    st_ipass += __popcll(__ballot(act && k == 0));
#endif
    // which copy of the sweep the lane runs this pass (first pass of its instance or not): the two halves of the
    // wavefront may differ (both copies then run, one after the other); an idle half follows its partner
    const bool v1 = act ? first : (__ballot(act && first) != 0ull);
    STAMP_A();
    // ---- sweep: trial point, model functions, condensing, stage partials -------------------------------
    Partials q = {0, 0, 0, 0, 0, 0, 0, 0, 1e300, 0};
#ifdef RMPC_STAMPS
    q.tk[0] = q.tk[1] = q.tk[2] = q.tk[3] = q.tk[4] = q.tk[5] = 0;
#endif
    // Generated views with LDS records: the step lengths of a fresh step are formed inside the sweep call (MERGE2).
    // (The same reordering for the runtime tables, inline, is bit-identical too and no faster: boxer 0.48 vs 0.50 M.)
    constexpr bool MERGE2 = V::SPEC && REC_LDS;
    park();
    bool fresh = false;
    if constexpr (MERGE2) {
      const bool nostep = first || (s.redo != 0);
      fresh = act && !nostep && (s.newstep != 0);
      const FusedWs *const Fp = (const FusedWs *)(Tp + 1);
      __attribute__((address_space(3))) SweepStepOut *const so = (__attribute__((address_space(3))) SweepStepOut *)&sres[half];
      if (v1) fused_sweep_step_call<C, V, 1>(so, Fp, M.N, M.dt, M.use_curv, b, s.cur, k, slots, act && stage, nostep, fresh, s.ls, s.amin_p, s.amin_d, gphi_sum, s.mu, warm ? 1 : 0);
      else fused_sweep_step_call<C, V, 0>(so, Fp, M.N, M.dt, M.use_curv, b, s.cur, k, slots, act && stage, nostep, fresh, s.ls, s.amin_p, s.amin_d, gphi_sum, s.mu, warm ? 1 : 0);
    } else if constexpr (V::SPEC || kFusedCalls) {
      // the sweep is a call (scalars in, partials out): a generated view, or the runtime tables through GView
      if (act && stage) {
        const bool nostep = first || (s.redo != 0);
        double alpha = 0.0, adual = 0.0;
        if (!nostep) {
          alpha = ldexp(s.amin_p, -s.ls);
          adual = s.amin_d;
        }
        const FusedWs *const Fp = (const FusedWs *)(Tp + 1);   // (the pointer block behind the row tables)
        if (first) q = fused_sweep_call<C, VC, 1, REC_LDS>(Fp, M.N, M.dt, M.use_curv, b, s.cur, k, slots, nostep, alpha, adual, s.mu, warm ? 1 : 0);
        else q = fused_sweep_call<C, VC, 0, REC_LDS>(Fp, M.N, M.dt, M.use_curv, b, s.cur, k, slots, nostep, alpha, adual, s.mu, warm ? 1 : 0);
      }
    } else if (act && stage) {
      const int cur = s.cur, nxt = cur ^ 1;
      using RP = typename std::conditional<REC_LDS, ldouble, gdouble>::type;
      const FusedPtrs Pw = fused_ptrs(F, b);
      SweepIO<RP> io;
      io.zc = Pw.pz[cur]; io.tc = Pw.pt[cur]; io.lc = Pw.pl[cur]; io.nc = Pw.pn[cur];
      io.zn = Pw.pz[nxt]; io.tn = Pw.pt[nxt]; io.ln = Pw.pl[nxt]; io.nn = Pw.pn[nxt];
      io.pp = Pw.pp; io.gro = Pw.pg[cur]; io.jqo = Pw.pj[cur]; io.grn = Pw.pg[nxt]; io.jqn = Pw.pj[nxt];
      io.gfa = Pw.pgf;
      io.SS = S; io.loff = (unsigned)k; io.kstride = 1u;
      if constexpr (REC_LDS) {
        io.rec = slots + k * GS;
        io.dzp = slots + DZ_OFF; io.nup = slots + DZ_OFF + NV;
        io.SSd = 1; io.loffd = (unsigned)(k * GS); io.kstrided = (unsigned)GS;
      } else {
        io.rec = (gdouble *)F.R + (b * (size_t)N + k) * C::RS;
        io.dzp = Pw.pdz; io.nup = Pw.pnn;
        io.SSd = S; io.loffd = (unsigned)k; io.kstrided = 1u;
      }
      io.wl = Pw.pwl; io.wn = Pw.pwn; io.warm = warm ? 1 : 0;
      const bool nostep = first || (s.redo != 0);
      double alpha = 0.0, adual = 0.0;
      if (!nostep) {
        alpha = ldexp(s.amin_p, -s.ls);
        adual = s.amin_d;
      }
      const SweepK sk = {M.N, M.dt, M.use_curv};
      if (first) sweep_body<C, -1, RP, V, 1>(sk, v, io, k, true, nostep, alpha, adual, s.mu, q);
      else sweep_body<C, -1, RP, V, 0>(sk, v, io, k, false, nostep, alpha, adual, s.mu, q);
    }
#ifdef RMPC_STAMPS
    const long long st_ret = __builtin_amdgcn_s_memtime();   // (the sweep call has returned)
#endif
    unpark();
    Reduced r;
    if constexpr (MERGE2) {
      // (the call has reduced over the stages and left the instance's words in LDS: unpark's fence orders the reads)
      const SweepStepOut o = sres[half];
      if (fresh) { s.amin_p = o.amin_p; s.amin_d = o.amin_d; gphi_sum = o.gphi; }
      r.f = o.f; r.th = o.th; r.lgs = o.lgs; r.sumc = o.sumc; r.badf = o.badf;
      r.rstat = o.rstat; r.req = o.req; r.rineq = o.rineq; r.rcomp = o.rcomp; r.minc = o.minc;
#ifdef RMPC_STAMPS
      for (int i = 0; i < 6; i++) q.tk[i] = o.tk[i];
#endif
    } else {
      double rs5[5] = {q.f, q.th, q.logs, q.sumc, q.bad}, rm4[4] = {q.rstat, q.req, q.rineq, q.rcomp}, rn1[1] = {q.minc};
      wave_reduce_many<LPI>(rs5, rm4, rn1);
      r.f = rs5[0]; r.th = rs5[1]; r.lgs = rs5[2]; r.sumc = rs5[3]; r.badf = rs5[4];
      r.rstat = rm4[0]; r.req = rm4[1]; r.rineq = rm4[2]; r.rcomp = rm4[3]; r.minc = rn1[0];
    }
    r.gphi = first ? 0.0 : gphi_sum;
#ifdef RMPC_STAMPS
    const long long st_red = __builtin_amdgcn_s_memtime();
#endif
    GSYNC();   // trial point and records are complete before any lane reads another lane's part
    STAMP_B(st_sweep);
#ifdef RMPC_STAMPS
    for (int i = 0; i < 6; i++) st_sw[i] += __builtin_amdgcn_readfirstlane((int)q.tk[i]);
    st_sw2[0] += st_red - st_ret; st_sw2[1] += st_b - st_red;   // unpark + reductions; the ordering point's wait
#endif
    // ---- decisions, then a new step when the trial was accepted --------------------------------------
    bool usec = false;
    bool recurse = false;
    if (act) recurse = inst_decide<C>(M, s, r, first, usec);
    if (act) first = false;
    STAMP_B(st_dec);
    park();
    const double mu_r = s.mu;
    const double cw_r = usec ? (C::CSCALE ? s.theta_c : 1.0) : 0.0;   // weight of the curvature terms in this recursion
    bool rec_ok = true;
    double ap = 1.0, ad = 1.0, gp = 0.0;
    if (recurse) {
      bool ok;
      if constexpr (REC_LDS) {
        StepOut<ldouble> so;
        so.dz = slots + DZ_OFF; so.nunew = slots + DZ_OFF + NV; so.SS = 1; so.KS = GS;
        ok = fused_recursion_lds<C>(M.N, M.dt, mu_r, cw_r, k, work, slots, so);
      } else {
        const FusedPtrs Pr = fused_ptrs(F, b);
        StepOut<gdouble> so;
        so.dz = Pr.pdz; so.nunew = Pr.pnn; so.SS = S; so.KS = 1;
        ok = fused_recursion_mem<C>(M.N, M.dt, mu_r, cw_r, k, work, (gdouble *)F.R + b * (size_t)N * C::RS,
                                    (gdouble *)F.KP + b * (size_t)N * F.kps, F.kps, so);
      }
      rec_ok = ok;
    }
    unpark();
    if (recurse) inst_after_recursion(s, rec_ok, usec, C::BACKOFF, C::CSCALE);
    GSYNC();   // dz, nunew
    STAMP_B(st_ric);
    // ---- step lengths of the new step -----------------------------------------------------------------
    const bool stepping = !MERGE2 && act && (s.status == ST_ACTIVE) && (s.newstep != 0);
    park();
    if constexpr (V::SPEC || kFusedCalls) {
      if (stepping && stage) {
        const FusedWs *const Fp = (const FusedWs *)(Tp + 1);
        const StepRes sr = fused_step_call<C, VC, REC_LDS>(Fp, b, s.cur, k, slots, s.mu);
        ap = sr.ap; ad = sr.ad; gp = sr.gp;
      }
    } else if (stepping && stage) {
      const int cur = s.cur;
      using RP = typename std::conditional<REC_LDS, ldouble, gdouble>::type;
      const FusedPtrs Ps = fused_ptrs(F, b);
      StepIO<RP> io;
      io.zc = Ps.pz[cur]; io.tc = Ps.pt[cur]; io.lc = Ps.pl[cur]; io.grow = Ps.pg[cur]; io.Jq = Ps.pj[cur];
      io.gfa = Ps.pgf;
      io.SS = S; io.loff = (unsigned)k;
      if constexpr (REC_LDS) { io.dz = slots + DZ_OFF; io.SSd = 1; io.loffd = (unsigned)(k * GS); }
      else { io.dz = Ps.pdz; io.SSd = S; io.loffd = (unsigned)k; }
      step_body<C, RP, V>(v, io, k, s.mu, ap, ad, gp);
    }
    unpark();
    {
      double rs1[1] = {gp}, rm0[1] = {0.0}, rn2[2] = {ap, ad};
      wave_reduce_many<LPI>(rs1, rm0, rn2);
      gp = rs1[0]; ap = rn2[0]; ad = rn2[1];
    }
    if (stepping) {
      s.amin_p = fmin(s.amin_p, ap);
      s.amin_d = fmin(s.amin_d, ad);
      gphi_sum = gp;
    }
    STAMP_B(st_step);
  }
#ifdef RMPC_STAMPS
  if (threadIdx.x == 0) {
    long long *o = F.stamps + (size_t)blockIdx.x * 8;
    {
      long long *o2 = F.stamps + (size_t)(gridDim.x + blockIdx.x) * 8;   // second half of the array: sweep sections
      for (int i = 0; i < 6; i++) o2[i] = st_sw[i];
      o2[6] = st_sw2[0]; o2[7] = st_sw2[1];
    }
    o[0] = st_sweep; o[1] = st_dec; o[2] = st_ric; o[3] = st_step; o[4] = __builtin_amdgcn_s_memtime() - st_t0; o[5] = pass;
    o[6] = st_t0; o[7] = st_ipass;
  }
#endif
}

#include "rmpc_arm_fused.hpp"   // the arms in one launch (k_fused_arm: a wavefront per instance, a stage per P lanes)

// Launch order of a fused launch: the instances sorted by a key (the passes of their previous solve for a warm start,
// k_difficulty's estimate for a cold one), largest first (counting sort, one block; the order inside a bucket is
// whatever the atomics give -- it changes which instances share a wavefront, never what an instance computes).
// (static: every translation unit of the build launches it from its variants' launchers and keeps its own copy)
// (NT = 64 for the order of a cold launch, which runs IN FRONT of the fused launch: with other handles' fused launches
//  on the chip every SIMD is held by one long-lived 512-register wavefront, and a block of several wavefronts would
//  wait until a whole compute unit has drained; a single wavefront takes the first SIMD that frees)
template <int NT>
static __global__ __launch_bounds__(NT) void k_order_t(const int *__restrict__ key, int *__restrict__ order, int B) {
  __shared__ int cnt[256];
  const int tid = threadIdx.x;
  for (int i = tid; i < 256; i += NT) cnt[i] = 0;
  __syncthreads();
  // (eight keys per lane and round: the requests of a round are in flight together -- one by one the single
  //  wavefront of the cold order spent 33 us on 4096 keys, most of it waiting for one key at a time)
  constexpr int U = 8;
  for (int b0 = tid; b0 < B; b0 += NT * U) {
    int kq[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int b = b0 + u * NT; kq[u] = key[b < B ? b : B - 1]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int kk = kq[u] < 0 ? 0 : (kq[u] > 255 ? 255 : kq[u]);
      if (b0 + u * NT < B) atomicAdd(&cnt[255 - kk], 1);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; i++) { const int c = cnt[i]; cnt[i] = run; run += c; }
  }
  __syncthreads();
  for (int b0 = tid; b0 < B; b0 += NT * U) {
    int kq[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int b = b0 + u * NT; kq[u] = key[b < B ? b : B - 1]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int kk = kq[u] < 0 ? 0 : (kq[u] > 255 ? 255 : kq[u]);
      if (b0 + u * NT < B) order[atomicAdd(&cnt[255 - kk], 1)] = b0 + u * NT;
    }
  }
}

// ===========================================================================
// Scene packing and closed-loop advance: device
// counterparts of the planner's host loops, so that neither the N*npar
// parameter vectors nor the plans have to cross PCIe between control steps.
// ===========================================================================
struct SceneDev {
  const double *goal, *r_body, *obst, *obst_dyn, *lower, *upper, *lower_u, *upper_u, *lower_vel, *upper_vel, *lin;
  double dyn_radius, w, wu, ws;
  double wconstr[RMPC_MAX_MODULES];
};
struct SceneOff {
  int r_body, obst, lin, lower, upper, lower_u, upper_u, lower_vel, upper_vel, wu, goal, wgoal, wconstr, ws;
  int n, nu, nobst, n_modules, npar, N;
  double dt;
};

// One lane per (instance, stage).  SOA = 0: ABI layout params[b][k][npar] (what
// MPCPlanner.reset() + set*() + updateDynamicObstacles() produce, mpcPlanner.py:83-210);
// SOA = 1: straight into the pass kernels' batch-minor parameter array; SOA = 2: into the fused kernel's
// per-instance layout.
template <int SOA>
__global__ __launch_bounds__(256) void k_scene(const SceneDev S, const SceneOff O, double *__restrict__ out, int B, int Bp) {
#pragma clang fp contract(off)
  const int gid = blockIdx.x * 256 + threadIdx.x;
  int b, k;
  if (SOA == 1) { b = gid % Bp; k = gid / Bp; } else { k = gid % O.N; b = gid / O.N; }
  if (b >= B || k >= O.N) return;
  auto put = [&](int off, double v) __attribute__((always_inline)) {
    if (SOA == 1) out[((size_t)off * O.N + k) * Bp + b] = v;
    else if (SOA == 2) out[((size_t)b * O.npar + off) * kFusedStages + k] = v;   // fused kernel: [instance][slot][32 stages]
    else out[((size_t)b * O.N + k) * O.npar + off] = v;
  };
  // reset(): zeros, then the broadcast weights (mpcPlanner.py:91-104)
  for (int j = 0; j < O.npar; j++) put(j, 0.0);
  if (O.wgoal >= 0) for (int j = 0; j < 3; j++) put(O.wgoal + j, S.w);
  for (int j = 0; j < O.nu; j++) put(O.wu + j, S.wu);
  if (O.ws >= 0) put(O.ws, S.ws);
  if (O.wconstr >= 0) for (int j = 0; j < O.n_modules; j++) put(O.wconstr + j, S.wconstr[j]);
  if (O.goal >= 0 && S.goal) for (int j = 0; j < 3; j++) put(O.goal + j, S.goal[(size_t)b * 3 + j]);
  if (O.r_body >= 0 && S.r_body) put(O.r_body, S.r_body[b]);
  if (O.obst >= 0) {
    if (S.obst_dyn) {
      // updateDynamicObstacles (mpcPlanner.py:144-161): c = pos + (vel*dt)*k + (0.5*(dt*k)^2)*acc
      const double kk = (double)k;
      for (int j = 0; j < O.nobst; j++) {
        const double *o = S.obst_dyn + ((size_t)b * O.nobst + j) * 9;
        for (int c = 0; c < 3; c++) {
          // every product and sum rounded separately (fp contraction is switched off for this
          // kernel): bit-identical to the reference's numpy expression pos + vel*dt*i + 0.5*(dt*i)**2*acc
          const double tk = O.dt * kk;
          const double lin = (o[3 + c] * O.dt) * kk;
          const double quad = (0.5 * (tk * tk)) * o[6 + c];
          put(O.obst + 4 * j + c, (o[c] + lin) + quad);
        }
        put(O.obst + 4 * j + 3, S.dyn_radius);
      }
    } else if (S.obst) {
      for (int j = 0; j < 4 * O.nobst; j++) put(O.obst + j, S.obst[(size_t)b * 4 * O.nobst + j]);
    } else {
      // no obstacles given: every slot is the reference's EmptyObstacle (position -100, radius -100;
      // mpcPlanner.py:18-26,127-133), as the host packer writes
      for (int j = 0; j < 4 * O.nobst; j++) put(O.obst + j, -100.0);
    }
  }
  if (O.lin >= 0 && S.lin)
    for (int j = 0; j < 4 * O.nobst; j++) put(O.lin + j, S.lin[((size_t)b * O.N + k) * 4 * O.nobst + j]);
  if (O.lower >= 0 && S.lower) for (int j = 0; j < O.n; j++) put(O.lower + j, S.lower[(size_t)b * O.n + j]);
  if (O.upper >= 0 && S.upper) for (int j = 0; j < O.n; j++) put(O.upper + j, S.upper[(size_t)b * O.n + j]);
  if (O.lower_u >= 0 && S.lower_u) for (int j = 0; j < O.nu; j++) put(O.lower_u + j, S.lower_u[(size_t)b * O.nu + j]);
  if (O.upper_u >= 0 && S.upper_u) for (int j = 0; j < O.nu; j++) put(O.upper_u + j, S.upper_u[(size_t)b * O.nu + j]);
  if (O.lower_vel >= 0 && S.lower_vel) for (int j = 0; j < 2; j++) put(O.lower_vel + j, S.lower_vel[(size_t)b * 2 + j]);
  if (O.upper_vel >= 0 && S.upper_vel) for (int j = 0; j < 2; j++) put(O.upper_vel + j, S.upper_vel[(size_t)b * 2 + j]);
}

// Closed loop between two solves: the plant is the model's own ERK2 map applied to the first
// control of the previous plan, the warm start is the shifted plan (shiftHorizon,
// mpcPlanner.py:215-226) or the current state repeated (setX0 "current_state", :228-232).
constexpr int kAdvanceIB = 16;   // instances per block of k_advance
template <class C>
__global__ __launch_bounds__(256) void k_advance(const DevModel M, const double *__restrict__ zprev, double *__restrict__ xinit,
                                                 double *__restrict__ x0, int B, int previous_plan_all,
                                                 const int *__restrict__ exitflag) {
  // A block takes kAdvanceIB instances: one lane each for the plant step, then all 256 lanes shift the plans
  // element by element (contiguous in the ABI layout [b][k][j]: coalesced; one lane per instance walking its
  // N x nvar plan took 76 us for 1024 arms).
  constexpr int NX = C::NX, NS = C::NS, NV = C::NV, IB = kAdvanceIB;
  __shared__ double sx[IB][NX];
  __shared__ int spp[IB];
  const int b0 = blockIdx.x * IB, t = threadIdx.x;
  const int N = M.N;
  if (t < IB && b0 + t < B) {
    const int b = b0 + t;
    double z[NV], xn[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) z[j] = xinit[(size_t)b * NX + j];
#pragma unroll
    for (int j = NX; j < NV; j++) z[j] = zprev[(size_t)b * N * NV + j];  // slack and first control of the plan
    if constexpr (C::ROBOT == RMPC_ROBOT_CHAIN) {
      chain_step<C>(M.dt, z, xn);
    } else {
      double A5[25], B5[10];
      diffdrive_step<C>(M.dt, z, xn, A5, B5, false);
    }
#pragma unroll
    for (int j = 0; j < NX; j++) { xinit[(size_t)b * NX + j] = xn[j]; sx[t][j] = xn[j]; }
    // an instance whose last solve failed (exitflag < 0) has no plan worth shifting: it restarts from its state,
    // as the boxer example of the reference does for its linearisation point (boxer_example.py:194-198)
    spp[t] = (previous_plan_all && !(exitflag && exitflag[b] < 0)) ? 1 : 0;
  }
  __syncthreads();
  const int nb = (B - b0) < IB ? (B - b0) : IB;
  const int per = N * NV;
  for (int e = t; e < nb * per; e += 256) {
    const int ib = e / per, r = e - ib * per, k = r / NV, j = r - k * NV;
    const size_t base = (size_t)(b0 + ib) * per;
    double val;
    if (spp[ib]) val = zprev[base + (size_t)(k + 1 < N ? k + 1 : N - 1) * NV + j];
    else val = j < NX ? sx[ib][j] : 0.0;
    x0[base + r] = val;
  }
  (void)NS;
}

// Steady closed loop (the examples hand the planner a new goal whenever the driver has one,
// setGoalReaching every control step in examples/boxer_example_global.py:203-212): one lane per instance looks at the
// state the plant step has just produced and gives the instance its next goal from its pool when the end link has
// arrived (within tol of the goal) or has dwelt max_dwell control steps on this goal; an instance whose solve FAILED
// (exitflag < 0: infeasible or diverged, a state no plan leads out of) is put back to its start state with a cold
// plan and takes its next goal too.  The goals live in the scene's goal array, so the next parameter packing sees them.
struct RetargetDev {   // rmpc_retarget, device side
  double *xinit, *x0, *goal;
  const int *exitflag, *iters;
  const double *pool, *x_start, *lower, *upper;
  int P;
  int *cursor, *dwell, *failrun;
  double tol, settle_vel;
  int settle_min_dwell, max_dwell, fail_reset_after;
  long long *counts;
  double *wmu;
  double wmu_regoal;
};
template <class C>
__global__ __launch_bounds__(256) void k_retarget(const DevModel M, const DevTables *__restrict__ Tp, int B, const RetargetDev R) {
  constexpr int NQ = C::NQ, NX = C::NX, NV = C::NV;
  const int b = blockIdx.x * 256 + threadIdx.x;
  const bool in = b < B;
  long long *const counts = R.counts;
  // statistics of the control step, summed on the device (no host read inside the loop): exit flags and iterations
  if (counts && R.exitflag) {
    const int ef = in ? R.exitflag[b] : -1000;
    const int it = (in && R.iters) ? R.iters[b] : 0;
    const int cls[4] = {ef == 1, ef == 2, ef == 0, ef < 0 && ef > -1000};
    for (int c = 0; c < 4; c++) {
      const unsigned long long mk = __ballot(cls[c]);
      if ((threadIdx.x & 63) == 0 && mk) atomicAdd((unsigned long long *)&counts[4 + c], (unsigned long long)__popcll(mk));
    }
    int si = it;
    for (int off = 32; off >= 1; off >>= 1) si += __shfl_xor(si, off, 64);
    if ((threadIdx.x & 63) == 0 && si) atomicAdd((unsigned long long *)&counts[8], (unsigned long long)si);
  }
  if (!in) return;
  const RtView v(M, *Tp);
  double *const xi = R.xinit + (size_t)b * NX;
  const bool failed = R.exitflag && R.exitflag[b] < 0;
  // A failed solve (infeasible, diverged, line search): the reference prints the flag and drives on with the action it
  // got (mpcPlanner.py:263-264), its boxer example takes the current pose as the next linearisation point
  // (boxer_example.py:194-198) -- the plant step has applied the returned control, the next solve starts cold from the
  // new state (rmpc_advance_device_flags).  Only an instance that has failed fail_reset_after control steps IN A ROW is
  // put back to its start state (a reset; 0: never).
  int fr = R.failrun ? R.failrun[b] : 0;
  fr = failed ? fr + 1 : 0;
  // ... or one whose configuration has left the joint-limit box by more than a limit row's reach (a robot outside its
  // workspace: the examples' simulator stops a joint at its limit, the plant here is the bare integrator -- a short
  // horizon without a terminal set does overshoot a far goal; counted on its own, counts[12])
  bool oob = false;
  if (R.lower && R.upper) {
    for (int j = 0; j < M.n; j++) {
      const double lo = R.lower[(size_t)b * M.n + j], hi = R.upper[(size_t)b * M.n + j];
      const double margin = 0.05 * (hi - lo);
      oob |= (xi[j] < lo - margin) | (xi[j] > hi + margin);
    }
  }
  const bool reset = oob || (failed && R.fail_reset_after > 0 && fr >= R.fail_reset_after);
  if (counts && oob) atomicAdd((unsigned long long *)&counts[12], 1ull);
  if (reset) {
    for (int j = 0; j < NX; j++) xi[j] = R.x_start[(size_t)b * NX + j];
    for (int k = 0; k < M.N; k++)
      for (int j = 0; j < NV; j++) R.x0[((size_t)b * M.N + k) * NV + j] = j < NX ? R.x_start[(size_t)b * NX + j] : 0.0;
    fr = 0;
  }
  if (R.failrun) R.failrun[b] = fr;
  if (counts && fr > 0) atomicAdd((unsigned long long *)&counts[11], 1ull);
  double q[NQ];
#pragma unroll
  for (int j = 0; j < NQ; j++) q[j] = xi[j];
  Kin<C> kin;
  kin.compute(v, q);
  Vec3 J[NQ];
  const Vec3 pt = kin.template point<0>(v, J);   // slot 0: the goal's end frame (build_tables)
  double *const g = R.goal + (size_t)b * 3;
  const double dx = pt.x - g[0], dy = pt.y - g[1], dz = pt.z - g[2];
  const double dist = sqrt(dx * dx + dy * dy + dz * dz);
  const bool arrived = dist < R.tol;
  // settled: the robot has come to rest on this goal -- with the reference's objective (N w / h on the first row of a
  // module, constraint_avoidance.py:22-31) a goal next to an obstacle is an equilibrium at a distance, not a point reached
  double vmax = 0.0;
  if constexpr (C::ROBOT == RMPC_ROBOT_CHAIN) {
#pragma unroll
    for (int j = 0; j < NQ; j++) vmax = fmax(vmax, fabs(xi[NQ + j]));
  } else {
    vmax = fmax(fabs(xi[6]), fabs(xi[7]));
  }
  int dw = R.dwell[b] + 1;
  const bool settled = !arrived && R.settle_vel > 0.0 && dw >= R.settle_min_dwell && vmax < R.settle_vel;
  const bool late = R.max_dwell > 0 && dw >= R.max_dwell;
  if (arrived || settled || late || reset) {
    const int c = R.cursor[b] + 1;
    R.cursor[b] = c;
    const double *gn = R.pool + ((size_t)b * R.P + (size_t)(c % R.P)) * 3;
    g[0] = gn[0]; g[1] = gn[1]; g[2] = gn[2];
    dw = 0;
    // a new goal moves the optimum: the multipliers of the last solve stay, the barrier parameter of the next solve
    // restarts from mu_regoal (stored so that warm_mu() yields it) instead of 1000 x the converged one
    if (R.wmu && R.wmu_regoal > 0.0 && !failed) R.wmu[b] = R.wmu_regoal;
    if (counts) {
      atomicAdd((unsigned long long *)&counts[reset ? 3 : (arrived ? 0 : (settled ? 1 : 2))], 1ull);
      if (!reset) {
        atomicAdd((unsigned long long *)&counts[9], (unsigned long long)(dist * 1e6));   // distance at the hand-over [um]
        atomicAdd((unsigned long long *)&counts[10], 1ull);
      }
    }
  }
  R.dwell[b] = dw;
}

// The environment of the moving obstacles between two control steps (what the examples' simulator does before the driver
// hands the planner ob[nx:], mpcPlanner.py:243-244): pos += vel dt + acc dt^2 / 2, vel += acc dt, one lane per
// (instance, obstacle); arena > 0: an obstacle that leaves [-arena, arena] in x or y comes back (velocity component
// mirrored), so that a loop that runs for hours keeps its obstacles.
static __global__ __launch_bounds__(256) void k_obst_advance(double *__restrict__ od, int n, double dt, double arena) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double *o = od + (size_t)i * 9;
  for (int c = 0; c < 3; c++) {
    double pos = o[c] + o[3 + c] * dt + 0.5 * o[6 + c] * dt * dt;
    double vel = o[3 + c] + o[6 + c] * dt;
    if (arena > 0.0 && c < 2) {
      if (pos > arena) { pos = 2.0 * arena - pos; vel = -vel; }
      else if (pos < -arena) { pos = -2.0 * arena - pos; vel = -vel; }
    }
    o[c] = pos; o[3 + c] = vel;
  }
}

// Launch order of a COLD fused launch that is larger than the chip (more instances than half-wavefronts: the rest wait
// in the queue).  A lone launch lasts as long as its slowest instance needs from the moment it is dequeued, and the
// slow ones of a cold batch are mostly those that start close to a constraint boundary (the interior-point method's
// first steps are cut by the fraction to the boundary): on the BASELINE scenarios 37-40 of the 40 slowest of 4096
// point robots are in the closer half.  One lane per instance evaluates the distance rows (obstacle, plane, self
// collision) of the start state with the parameters of the second stage and hands k_order_t a key, closest first --
// longest-processing-time-first scheduling with an estimate instead of the previous solve's count.  What an instance
// computes does not depend on its place in the queue (test_launch_order_...).
template <class C>
__global__ __launch_bounds__(64) void k_difficulty(const DevModel M, const DevTables *__restrict__ Tp, const int B,
                                                    const double *__restrict__ xinit, const double *__restrict__ params,
                                                    int *__restrict__ key) {
  constexpr int NQ = C::NQ, NX = C::NX;
  const int b = blockIdx.x * 64 + threadIdx.x;   // (one-wavefront blocks: see k_order_t)
  if (b >= B) return;
  const RtView v(M, *Tp);
  const double *const P = params + ((size_t)b * M.N + (M.N > 1 ? 1 : 0)) * M.npar;
  double q[NQ];
#pragma unroll
  for (int j = 0; j < NQ; j++) q[j] = xinit[(size_t)b * NX + j];
  Kin<C> kin;
  kin.compute(v, q);
  const double rbody = (v.off_r_body() >= 0) ? P[v.off_r_body()] : 0.0;
  double dmin = 1e30, dseg = 1e30;
  // (slot 0 is the goal's end frame when the model has a GoalReaching objective: for its spherical obstacles also the
  //  clearance of the straight segment from the point to the goal -- an instance whose way is blocked takes longer
  //  than one that merely starts next to an obstacle; the sum of the two clearances orders the BASELINE batches
  //  almost as well as the iteration counts themselves: simulated makespans 36 / 60 / 61 / 59 / 102 / 46 iterations
  //  against 35 / 60 / 61 / 59 / 102 / 46 for the true longest-first order and 48 / 72 / 61 / 73 / 102 / 54 by index)
  const bool goal = v.has_goal() != 0;
  Vec3 gv = {0, 0, 0};
  if (goal) gv = {P[v.off_goal()], P[v.off_goal() + 1], P[v.off_goal() + 2]};
  for_range<0, kMaxSlots>([&](auto slc) __attribute__((always_inline)) {
    constexpr int SL = decltype(slc)::value;
    if (SL < v.nslots()) {
      Vec3 J[NQ];
      const Vec3 Pt = kin.template point<SL>(v, J);
      for (int r = v.slot_row_begin(SL); r < v.slot_row_begin(SL + 1); r++) {
        const int kind = v.fk_kind(r), ob = v.fk_obst(r);
        double h;
        if (kind == ROW_RADIAL) {
          const double *o = P + v.off_obst() + 4 * ob;
          const Vec3 dv = {Pt.x - o[0], Pt.y - o[1], Pt.z - o[2]};
          h = sqrt(dot(dv, dv)) - o[3] - rbody;
          if (SL == 0 && goal) {
            const Vec3 sg = {gv.x - Pt.x, gv.y - Pt.y, gv.z - Pt.z}, so = {o[0] - Pt.x, o[1] - Pt.y, o[2] - Pt.z};
            const double l2 = dot(sg, sg);
            double t = l2 > 0.0 ? dot(so, sg) / l2 : 0.0;
            t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
            const Vec3 cv = {so.x - t * sg.x, so.y - t * sg.y, so.z - t * sg.z};
            dseg = fmin(dseg, sqrt(dot(cv, cv)) - o[3] - rbody);
          }
        } else if (kind == ROW_LINEAR) {
          const double *o = P + v.off_lin() + 4 * ob;
          const Vec3 av = {o[0], o[1], o[2]};
          h = fabs(dot(av, Pt) + o[3]) / sqrt(dot(av, av)) - rbody;
        } else {
          h = sqrt(dot(Pt, Pt)) - 2.0 * rbody;
        }
        dmin = fmin(dmin, h);
      }
    }
  });
  // 256 classes of 4 cm of (clearance at the start + clearance of the way), the smallest (and every infeasible start)
  // in the class that is dequeued first
  const double dsum = (dmin > 0.0 ? dmin : 0.0) + (dseg < 1e29 ? (dseg > 0.0 ? dseg : 0.0) : (dmin > 0.0 ? dmin : 0.0));
  const double c = dsum * 25.0;
  key[b] = 255 - (c < 255.0 ? (int)c : 255);
}

#if RMPC_TU_MAIN
// ===========================================================================
// Free-space decomposition: lidar point cloud -> at most K half-planes
// around a seed point, one lane per (instance, stage) seed.  Greedy rule of the reference
// (robotmpcs/utils/free_space_decomposition.py:79-97): the closest remaining point inside
// max_radius defines the plane through it with normal (seed - point); points on or behind the
// plane are discarded; unused slots get the dummy plane of asdict() (:110-114).  The sort of
// the reference is replaced by K arg-min sweeps over a keep-mask (P <= 64 points).
// ===========================================================================
__global__ __launch_bounds__(256) void k_fsd(const double *__restrict__ points, const double *__restrict__ seeds,
                                             double *__restrict__ out, int B, int N, int P, int K, double max_radius) {
#pragma clang fp contract(off)
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= B * N) return;
  const int b = gid / N;
  const double *pc = points + (size_t)b * P * 3;
  const double s0 = seeds[(size_t)gid * 3], s1 = seeds[(size_t)gid * 3 + 1], s2 = seeds[(size_t)gid * 3 + 2];
  double *o = out + (size_t)gid * K * 4;
  unsigned long long keep = 0ull;
  for (int i = 0; i < P; i++) {
    const double d0 = pc[3 * i] - s0, d1 = pc[3 * i + 1] - s1, d2 = pc[3 * i + 2] - s2;
    if (sqrt(d0 * d0 + d1 * d1 + d2 * d2) < max_radius) keep |= (1ull << i);
  }
  int nc = 0;
  while (keep && nc < K) {
    int best = -1;
    double bd = 0.0;
    for (int i = 0; i < P; i++)
      if (keep & (1ull << i)) {
        const double d0 = pc[3 * i] - s0, d1 = pc[3 * i + 1] - s1, d2 = pc[3 * i + 2] - s2;
        const double dd = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        if (best < 0 || dd < bd) { best = i; bd = dd; }
      }
    const double p0 = pc[3 * best], p1 = pc[3 * best + 1], p2 = pc[3 * best + 2];
    const double n0 = s0 - p0, n1 = s1 - p1, n2 = s2 - p2;
    const double c = -((n0 * p0 + n1 * p1) + n2 * p2);
    o[4 * nc] = n0; o[4 * nc + 1] = n1; o[4 * nc + 2] = n2; o[4 * nc + 3] = c;
    nc++;
    for (int i = 0; i < P; i++)
      if (keep & (1ull << i)) {
        const double v = ((n0 * pc[3 * i] + n1 * pc[3 * i + 1]) + n2 * pc[3 * i + 2]) + c;
        if (v <= 0.0) keep &= ~(1ull << i);
      }
  }
  for (; nc < K; nc++) {
    // HalfPlane(seed + (20, 20, 0), seed): normal = seed - point
    const double p0 = s0 + 20.0, p1 = s1 + 20.0, p2 = s2 + 0.0;
    const double n0 = s0 - p0, n1 = s1 - p1, n2 = s2 - p2;
    o[4 * nc] = n0; o[4 * nc + 1] = n1; o[4 * nc + 2] = n2; o[4 * nc + 3] = -((n0 * p0 + n1 * p1) + n2 * p2);
  }
}

#endif  // RMPC_TU_MAIN

}  // namespace rmpc

// ===========================================================================
// host side: handle, workspace, launch loop, C ABI
// ===========================================================================
using namespace rmpc;

inline thread_local std::string g_err;   // (one object for all translation units of the library)
inline int fail(const std::string &m) {
  g_err = m;
  return -1;
}
#define HIPCHK(x)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess)                                                                 \
      return fail(std::string(#x) + ": " + hipGetErrorString(e_));                        \
  } while (0)

enum KernelId { K_PACK = 0, K_SWEEP, K_RICCATI, K_STEP, K_UNPACK, K_FUSED };
static const char *kKernelNames[RMPC_NUM_KERNELS] = {"k_pack", "k_sweep", "k_riccati", "k_step", "k_unpack", "k_fused"};

struct rmpc_handle {
  rmpc_desc desc;
  DevModel M;
  DevTables T;
  DevTables *d_T = nullptr;
  Ws W;
  Ws Wc;            // compact workspace the last survivors of a batch migrate to (Bpc columns; Bpc == 0: none)
  int Bpc = 0;
  int warm_mode = 0;        // rmpc_set_warm_start
  bool have_duals = false;  // the warm-start arrays hold the multipliers of a finished solve of duals_B instances
  int duals_B = 0;
  int spec = -1;        // generated view whose tables equal this descriptor's (rmpc_spec_gen.hpp), -1: runtime tables
  bool fused = false;   // this model runs the fused kernel (small models, N <= 32); the pass kernels otherwise
  int ric_lane = 1;       // lane-per-instance recursion of the pass kernels: 0 never, 1 large lists, 2 always (RMPC_RIC_LANE)
  int fused_grid = 1024;  // wavefronts the chip holds at one per SIMD (4 x compute units): grid of a fused launch
  FusedWs F;
  int *h_passes = nullptr;  // pinned
  int device = 0;
  int max_batch = 0;
  int Bp = 0;
  int variant = -1;
  int max_passes = 0;
  int pass_budget = 0;            // rmpc_set_pass_budget (0: none)
  int packed_B = 0;               // batch size of the parameters rmpc_pack_scene_workspace left in the workspace
  void *ws_base = nullptr;
  size_t ws_bytes = 0;
  hipStream_t stream = nullptr;
  // staging for the host-pointer entry point
  double *d_xinit = nullptr, *d_x0 = nullptr, *d_params = nullptr, *d_zout = nullptr, *d_kkt = nullptr,
         *d_obj = nullptr;
  int *d_exit = nullptr, *d_iters = nullptr;
  int *h_active = nullptr;  // pinned
  int last_passes = 0;
  int last_cap = 0;             // passes enqueued by the last solve of the pass kernels
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev;   // pool, reused across solves
  size_t ev_used = 0;
  std::vector<int> ev_kind;
  double prof_ms[RMPC_NUM_KERNELS] = {0};
  int64_t prof_n[RMPC_NUM_KERNELS] = {0};
  int64_t lane_bytes[RMPC_NUM_KERNELS] = {0};   // per active lane (pack/unpack: per call)
  double prof_bytes[RMPC_NUM_KERNELS] = {0};     // accumulated algorithmic bytes of the profiled launches
  std::vector<int> h_hist;
  // debugging switches, read once at rmpc_create (never set by the product code)
  bool env_no_migrate = false, env_dump_hist = false, env_no_order = false, env_no_cold_order = false, env_arm_two_parts = false;
};

// ---- per-variant launchers -----------------------------------------------------------------------------------------
// Function templates that reference the kernels of ONE variant: instantiated in the translation unit that builds the
// variant (explicit instantiations below), only declared (extern template) in the others.
// The workspace the passes currently run in: the batch's own, or the compact one after migration
// (B = number of columns in use).
struct Phase {
  Ws W;
  int B;
};
constexpr int variant_id(int R, int NQ, int NS) {
#define RMPC_VID(ID, r, nq, ns) \
  if (R == r && NQ == nq && NS == ns) return ID;
  RMPC_VARIANTS(RMPC_VID)
#undef RMPC_VID
  return -1;
}
template <int R, int NQ, int NS>
constexpr bool variant_built() {       // kernels of the variant are instantiated in this translation unit
  return variant_id(R, NQ, NS) >= 0 && ((RMPC_DEV_VARIANTS >> variant_id(R, NQ, NS)) & 1);
}
template <int R, int NQ, int NS>
constexpr bool variant_available() {   // the library holds the variant (possibly in another translation unit)
  return variant_id(R, NQ, NS) >= 0 && ((RMPC_ALL_VARIANTS >> variant_id(R, NQ, NS)) & 1);
}

template <class C, class V>
int launch_pass(rmpc_handle *h, const Phase &ph, int first, int pass, hipStream_t st, int which) {
  const int B = ph.B;
  const int lanes = ph.W.Bp * h->M.N;
  if (ph.W.rs != C::RS) return fail("stage-record layout mismatch between the workspace and the kernel variant");
  if (which == K_SWEEP) hipLaunchKernelGGL((k_sweep<C, V>), dim3((lanes + kSweepBlock - 1) / kSweepBlock), dim3(kSweepBlock), 0, st, h->M, h->d_T, ph.W, B, first,
                                           (first && h->warm_mode && h->have_duals) ? 1 : 0);
  else if (which == K_RICCATI) {
    if constexpr (C::ROBOT == RMPC_ROBOT_CHAIN && C::NQ <= 3) {
      // lane-per-instance recursion for large lists (h->ric_lane: 0 never, 1 from kLaneMin instances on, 2 always)
      if (h->ric_lane == 2 || (h->ric_lane == 1 && B >= kLaneMin)) {
        hipLaunchKernelGGL((k_riccati_lane<C>), dim3((B + 63) / 64), dim3(64), 0, st, h->M, ph.W, B, first);
        return 0;
      }
    }
    if (C::IPB > 1 && B >= kGroupedMin)
    {
      constexpr int per_block = C::IPB * (64 / C::RIC_LPI);   // instances per block
      hipLaunchKernelGGL((k_riccati<C, C::IPB>), dim3((B + per_block - 1) / per_block), dim3(64 * C::IPB), 0, st, h->M, ph.W, B, first, pass);
    }
    const int tail_blocks = (C::IPB == 1 || B < kGroupedMin) ? B : kGroupedMin;
    hipLaunchKernelGGL((k_riccati<C, 1>), dim3(tail_blocks), dim3(64), 0, st, h->M, ph.W, B, first, pass);
  }
  else hipLaunchKernelGGL((k_step<C, V>), dim3((lanes + kSweepBlock - 1) / kSweepBlock), dim3(kSweepBlock), 0, st, h->M, h->d_T, ph.W, B);
  return 0;
}

template <class C, class V>
int launch_fused_t(rmpc_handle *h, int B, const double *d_xinit, const double *d_x0, const double *d_params,
                          double *d_zout, int *d_exit, int *d_iters, double *d_kkt, double *d_obj, hipStream_t st, int cap) {
  const int warm = (h->warm_mode && h->have_duals) ? 1 : 0;
  if (h->F.rs != C::RS) return fail("stage-record layout mismatch between the workspace and the kernel variant");
  // closed loop: the previous solve of this batch tells which instances take long (k_fused: launch order)
  int use_order = (warm && !h->env_no_order) ? 1 : 0;
  // the grid is the chip (one wavefront per SIMD: __launch_bounds__(64, 1)), the batch is a queue its halves drain
  const int pairs = (B + 1) / 2;
  const int grid = pairs < h->fused_grid ? pairs : h->fused_grid;
  // cold launch with a queue behind the grid: the instances closest to a constraint boundary first (k_difficulty)
  // (holonomic chains only: measured on the BASELINE batches, one launch alone 3.05 -> 2.59 ms for the point robots,
  //  14.7 -> 15.0 ms for the boxers, whose slow instances are slow for another reason -- the Gauss-Newton blocks of the
  //  unicycle converge linearly -- and gain nothing from the two little kernels in front of the launch)
  if (C::ROBOT == RMPC_ROBOT_CHAIN && !warm && !h->env_no_order && !h->env_no_cold_order && d_params && pairs > grid) {
    hipLaunchKernelGGL((k_difficulty<C>), dim3((B + 63) / 64), dim3(64), 0, st, h->M, h->d_T, B, d_xinit, d_params, h->F.ckey);
    hipLaunchKernelGGL(k_order_t<64>, dim3(1), dim3(64), 0, st, (const int *)h->F.ckey, h->F.order, B);
    use_order = 1;
  }
  hipLaunchKernelGGL((k_fused<C, C::FUSED_REC_LDS, V>), dim3(grid), dim3(64), 0, st, h->M, h->d_T, h->F, B, d_xinit, d_x0,
                     d_params, d_zout, d_exit, d_iters, d_kkt, d_obj, cap, warm, use_order);
  return 0;
}
template <class C>
int launch_fused_arm_t(rmpc_handle *h, int B, const double *d_xinit, const double *d_x0, const double *d_params,
                       double *d_zout, int *d_exit, int *d_iters, double *d_kkt, double *d_obj, hipStream_t st, int cap) {
  if constexpr (C::ARM_FUSED) {
    const int warm = (h->warm_mode && h->have_duals) ? 1 : 0;
    if (h->F.rs != C::RS) return fail("stage-record layout mismatch between the workspace and the kernel variant");
    const int use_order = (warm && !h->env_no_order) ? 1 : 0;
    // the grid is the chip (one wavefront per SIMD), the batch a queue its wavefronts drain
    const int grid = B < h->fused_grid ? B : h->fused_grid;
    // parts per stage: three when the horizon leaves room for them (3 N <= 64 lanes), else two
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void *)k_fused_arm<C, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ArmLds<C>::TOTAL * 8);
      (void)hipFuncSetAttribute((const void *)k_fused_arm<C, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, ArmLds<C>::TOTAL * 8);
      attr_set = true;
    }
    if (3 * h->M.N <= 64 && !h->env_arm_two_parts)
      hipLaunchKernelGGL((k_fused_arm<C, 3>), dim3(grid), dim3(64), ArmLds<C>::TOTAL * 8, st, h->M, h->d_T, h->F, B, d_xinit, d_x0,
                         d_params, d_zout, d_exit, d_iters, d_kkt, d_obj, cap, warm, use_order);
    else
      hipLaunchKernelGGL((k_fused_arm<C, 2>), dim3(grid), dim3(64), ArmLds<C>::TOTAL * 8, st, h->M, h->d_T, h->F, B, d_xinit, d_x0,
                         d_params, d_zout, d_exit, d_iters, d_kkt, d_obj, cap, warm, use_order);
    return 0;
  } else {
    return fail("no fused arm kernel for this model");
  }
}
template <class C>
int launch_advance(rmpc_handle *h, int B, const double *d_z_prev, const int *ef, double *d_xinit, double *d_x0, int previous_plan,
                   hipStream_t st) {
  hipLaunchKernelGGL((k_advance<C>), dim3((B + kAdvanceIB - 1) / kAdvanceIB), dim3(256), 0, st, h->M, d_z_prev, d_xinit, d_x0, B,
                     previous_plan, ef);
  return 0;
}
template <class C>
int launch_retarget(rmpc_handle *h, int B, const RetargetDev &R, hipStream_t st) {
  hipLaunchKernelGGL((k_retarget<C>), dim3((B + 255) / 256), dim3(256), 0, st, h->M, h->d_T, B, R);
  return 0;
}

// Explicit instantiations for the variants of this translation unit, extern declarations for the variants built in
// another one (the preprocessor cannot pick the keyword inside RMPC_VARIANTS: one #if per variant id; the list must
// follow RMPC_VARIANTS, static_assert below).  Generated views are instantiated implicitly by the dispatch of the main
// unit: a view is offered only when its variant is built there.
#define RMPC_SIG_PASS (rmpc_handle *, const Phase &, int, int, hipStream_t, int)
#define RMPC_SIG_FUSED (rmpc_handle *, int, const double *, const double *, const double *, double *, int *, int *, double *, double *, hipStream_t, int)
#define RMPC_SIG_ADV (rmpc_handle *, int, const double *, const int *, double *, double *, int, hipStream_t)
#define RMPC_SIG_RET (rmpc_handle *, int, const RetargetDev &, hipStream_t)
#define RMPC_INST(KW, R, NQ, NS)                                      \
  KW int launch_pass<Cfg<R, NQ, NS>, RtView> RMPC_SIG_PASS;           \
  KW int launch_advance<Cfg<R, NQ, NS>> RMPC_SIG_ADV;                 \
  KW int launch_retarget<Cfg<R, NQ, NS>> RMPC_SIG_RET;
#define RMPC_INST_F(KW, R, NQ, NS) KW int launch_fused_t<Cfg<R, NQ, NS>, RtView> RMPC_SIG_FUSED;
#define RMPC_INST_FA(KW, R, NQ, NS) KW int launch_fused_arm_t<Cfg<R, NQ, NS>> RMPC_SIG_FUSED;
static_assert(variant_id(RMPC_ROBOT_CHAIN, 3, 0) == 0, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 0) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 3, 0) RMPC_INST_F(template, RMPC_ROBOT_CHAIN, 3, 0)
#elif (RMPC_ALL_VARIANTS >> 0) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 3, 0) RMPC_INST_F(extern template, RMPC_ROBOT_CHAIN, 3, 0)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 3, 1) == 1, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 1) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 3, 1) RMPC_INST_F(template, RMPC_ROBOT_CHAIN, 3, 1)
#elif (RMPC_ALL_VARIANTS >> 1) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 3, 1) RMPC_INST_F(extern template, RMPC_ROBOT_CHAIN, 3, 1)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 7, 0) == 2, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 2) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 7, 0) RMPC_INST_FA(template, RMPC_ROBOT_CHAIN, 7, 0)
#elif (RMPC_ALL_VARIANTS >> 2) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 7, 0) RMPC_INST_FA(extern template, RMPC_ROBOT_CHAIN, 7, 0)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 7, 1) == 3, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 3) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 7, 1)
#elif (RMPC_ALL_VARIANTS >> 3) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 7, 1)
#endif
static_assert(variant_id(RMPC_ROBOT_DIFFDRIVE, 3, 0) == 4, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 4) & 1
RMPC_INST(template, RMPC_ROBOT_DIFFDRIVE, 3, 0) RMPC_INST_F(template, RMPC_ROBOT_DIFFDRIVE, 3, 0)
#elif (RMPC_ALL_VARIANTS >> 4) & 1
RMPC_INST(extern template, RMPC_ROBOT_DIFFDRIVE, 3, 0) RMPC_INST_F(extern template, RMPC_ROBOT_DIFFDRIVE, 3, 0)
#endif
static_assert(variant_id(RMPC_ROBOT_DIFFDRIVE, 3, 1) == 5, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 5) & 1
RMPC_INST(template, RMPC_ROBOT_DIFFDRIVE, 3, 1) RMPC_INST_F(template, RMPC_ROBOT_DIFFDRIVE, 3, 1)
#elif (RMPC_ALL_VARIANTS >> 5) & 1
RMPC_INST(extern template, RMPC_ROBOT_DIFFDRIVE, 3, 1) RMPC_INST_F(extern template, RMPC_ROBOT_DIFFDRIVE, 3, 1)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 2, 0) == 6, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 6) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 2, 0) RMPC_INST_F(template, RMPC_ROBOT_CHAIN, 2, 0)
#elif (RMPC_ALL_VARIANTS >> 6) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 2, 0) RMPC_INST_F(extern template, RMPC_ROBOT_CHAIN, 2, 0)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 4, 0) == 7, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 7) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 4, 0)
#elif (RMPC_ALL_VARIANTS >> 7) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 4, 0)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 5, 0) == 8, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 8) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 5, 0) RMPC_INST_FA(template, RMPC_ROBOT_CHAIN, 5, 0)
#elif (RMPC_ALL_VARIANTS >> 8) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 5, 0) RMPC_INST_FA(extern template, RMPC_ROBOT_CHAIN, 5, 0)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 6, 0) == 9, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 9) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 6, 0) RMPC_INST_FA(template, RMPC_ROBOT_CHAIN, 6, 0)
#elif (RMPC_ALL_VARIANTS >> 9) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 6, 0) RMPC_INST_FA(extern template, RMPC_ROBOT_CHAIN, 6, 0)
#endif
static_assert(variant_id(RMPC_ROBOT_CHAIN, 8, 0) == 10, "instantiation list out of step with RMPC_VARIANTS");
#if (RMPC_DEV_VARIANTS >> 10) & 1
RMPC_INST(template, RMPC_ROBOT_CHAIN, 8, 0)
#elif (RMPC_ALL_VARIANTS >> 10) & 1
RMPC_INST(extern template, RMPC_ROBOT_CHAIN, 8, 0)
#endif
#undef RMPC_INST
#undef RMPC_INST_F
#undef RMPC_INST_FA

#if RMPC_TU_MAIN

// Row tables in device memory (DevTables): kinematic slots with their FK rows, and the
// single-variable rows grouped by variable.
static int build_tables(const rmpc_desc &d, const DevModel &M, DevTables &T, std::string &err) {
  memset(&T, 0, sizeof T);
  for (int s = 0; s < kMaxSlots; s++) { T.slot_fa[s] = -1; T.slot_fb[s] = -1; }
  for (int j = 0; j < RMPC_NV_MAX; j++)
    for (int u = 0; u < kVarRows; u++) { T.v_row[j][u] = -1; T.v_poff[j][u] = -1; T.v_mod[j][u] = -1; }
  auto slot_of = [&](int fa, int fb) -> int {
    for (int s = 0; s < T.nslots; s++)
      if (T.slot_fa[s] == fa && T.slot_fb[s] == fb) return s;
    if (T.nslots >= kMaxSlots) return -1;
    T.slot_fa[T.nslots] = fa; T.slot_fb[T.nslots] = fb;
    return T.nslots++;
  };
  if (d.has_goal && slot_of(d.end_frame, -1) != 0) { err = "slot table"; return -1; }
  // FK rows with their slots, then sorted by slot
  struct FkRow { int row, kind, obst, mod, first, idx, slot; };
  std::vector<FkRow> rows;
  for (int i = 0; i < M.nh; i++) {
    if (M.row_kind[i] == ROW_SINGLE) continue;
    const int fb = (M.row_kind[i] == ROW_SELF) ? M.row_b[i] : -1;
    const int s = slot_of(M.row_a[i], fb);
    if (s < 0) { err = "more than 4 distinct collision points (links / link pairs / end link)"; return -1; }
    rows.push_back({i, M.row_kind[i], M.row_kind[i] == ROW_SELF ? 0 : M.row_b[i], M.row_mod[i],
                    i == M.mod_row0[M.row_mod[i]] ? 1 : 0, M.row_fk[i], s});
  }
  if ((int)rows.size() > kMaxFkRows) { err = "too many distance rows"; return -1; }
  int r = 0;
  for (int s = 0; s < kMaxSlots; s++) {
    T.slot_row_begin[s] = r;
    for (const FkRow &fr : rows)
      if (fr.slot == s) {
        T.fk_row[r] = fr.row; T.fk_kind[r] = fr.kind; T.fk_obst[r] = fr.obst;
        T.fk_mod[r] = fr.mod; T.fk_first[r] = fr.first; T.fk_idx[r] = fr.idx;
        r++;
      }
  }
  T.slot_row_begin[kMaxSlots] = r;
  T.nfkrows = r;
  // single-variable rows
  auto add_var_row = [&](int var, int row, int sgn, int poff, double val, int soft, int mod, int firstrow) -> bool {
    for (int u = 0; u < kVarRows; u++)
      if (T.v_row[var][u] < 0) {
        T.v_row[var][u] = row; T.v_sgn[var][u] = sgn; T.v_poff[var][u] = poff;
        T.v_val[var][u] = val; T.v_soft[var][u] = soft; T.v_mod[var][u] = mod;
        T.v_first[var][u] = firstrow;
        return true;
      }
    return false;
  };
  bool ok = true;
  for (int i = 0; i < M.nh && ok; i++)
    if (M.row_kind[i] == ROW_SINGLE)
      ok = add_var_row(M.row_a[i], i, M.row_b[i], M.row_poff[i], 0.0, 1, M.row_mod[i], i == M.mod_row0[M.row_mod[i]] ? 1 : 0);
  int i = M.nh;
  for (int q = 0; q < M.nlb && ok; q++, i++) ok = add_var_row(M.lb_var[q], i, +1, -1, M.lb_val[q], 0, -1, 0);
  for (int q = 0; q < M.nub && ok; q++, i++) ok = add_var_row(M.ub_var[q], i, -1, -1, M.ub_val[q], 0, -1, 0);
  if (!ok) { err = "more than 4 limit / bound rows on one variable"; return -1; }
  // packed copies (fused arm kernel)
  for (int j = 0; j < RMPC_NV_MAX; j++)
    for (int u = 0; u < kVarRows; u++) {
      int w = 0;
      if (T.v_row[j][u] >= 0 && T.v_row[j][u] < 256 && T.v_poff[j][u] < 65536) {
        w = T.v_row[j][u] | (1 << 8) | ((T.v_sgn[j][u] < 0 ? 1 : 0) << 9) | ((T.v_first[j][u] ? 1 : 0) << 10) |
            ((T.v_poff[j][u] >= 0 ? 1 : 0) << 11) | ((T.v_mod[j][u] >= 0 ? T.v_mod[j][u] & 7 : 0) << 12) |
            ((T.v_poff[j][u] >= 0 ? T.v_poff[j][u] : 0) << 16);
      }
      T.v_desc[j][u] = w;
    }
  for (int q = 0; q < T.nfkrows; q++)
    T.fk_desc[q] = (T.fk_row[q] & 255) | ((T.fk_kind[q] & 3) << 8) | ((T.fk_obst[q] & 63) << 10) | ((T.fk_mod[q] & 7) << 16) |
                   ((T.fk_first[q] ? 1 : 0) << 19) | ((T.fk_idx[q] & 63) << 20);
  T.slot_rows_max = 0;
  for (int s = 0; s < kMaxSlots; s++)
    if (T.slot_row_begin[s + 1] - T.slot_row_begin[s] > T.slot_rows_max) T.slot_rows_max = T.slot_row_begin[s + 1] - T.slot_row_begin[s];
  return 0;
}

static int variant_of(const rmpc_desc &d) {
  // id of the kernel variant (RMPC_VARIANTS) this library holds for the robot, -1: none
#define RMPC_VO(ID, R, NQ, NS) \
  if (((RMPC_ALL_VARIANTS >> ID) & 1) && d.robot == R && d.n == NQ && (d.ns ? 1 : 0) == NS) return ID;
  RMPC_VARIANTS(RMPC_VO)
#undef RMPC_VO
  return -1;
}

static int build_model(const rmpc_desc &d, DevModel &M, std::string &err) {
  memset(&M, 0, sizeof M);
  M.robot = d.robot; M.N = d.N; M.n = d.n; M.nx = d.nx; M.nu = d.nu; M.ns = d.ns;
  M.nv = d.nx + d.ns + d.nu; M.nw = d.ns + d.nu; M.npar = d.npar; M.dt = d.dt;
  if (d.N < 1 || d.N > 1000) { err = "horizon out of range"; return -1; }
  if (d.n_joints < 1 || d.n_joints > RMPC_MAX_JOINTS) { err = "n_joints out of range"; return -1; }
  if (d.ns != 0 && d.ns != 1) { err = "ns must be 0 or 1"; return -1; }
  if (d.robot == RMPC_ROBOT_CHAIN) {
    if (d.nx != 2 * d.n || d.nu != d.n) { err = "holonomic chain needs nx = 2n, nu = n"; return -1; }
    if (d.n_joints != d.n) { err = "chain with fixed joints between root and end link is not supported"; return -1; }
    for (int j = 0; j < d.n_joints; j++)
      if (d.joint_type[j] == RMPC_JOINT_FIXED || d.joint_dof[j] != j) { err = "chain joints must all be actuated, in order"; return -1; }
  } else if (d.robot == RMPC_ROBOT_DIFFDRIVE) {
    if (d.n != 3 || d.nx != 8 || d.nu != 2) { err = "diff-drive needs n = 3, nx = 8, nu = 2 (fk.n() == 0)"; return -1; }
    for (int j = 0; j < d.n_joints; j++)
      if (d.joint_type[j] != RMPC_JOINT_FIXED) { err = "diff-drive chain must consist of fixed joints"; return -1; }
  } else { err = "unknown robot kind"; return -1; }
  if (d.n_joints < 1 || d.n_joints > RMPC_MAX_JOINTS) { err = "n_joints out of range"; return -1; }
  if (M.nv > RMPC_NV_MAX) { err = "nvar too large"; return -1; }
  M.n_modules = d.n_modules; M.nobst = d.nobst; M.end_frame = d.end_frame; M.n_joints = d.n_joints;
  if (d.n_modules < 0 || d.n_modules > RMPC_MAX_MODULES) { err = "n_modules out of range"; return -1; }
  if (d.n_xrows < 0 || d.n_xrows > RMPC_MAX_XROWS) { err = "n_xrows out of range"; return -1; }
  for (int r = 0; r < d.n_xrows; r++)
    if (d.xrow_mod[r] < 0 || d.xrow_mod[r] >= d.n_modules || d.module_kind[d.xrow_mod[r]] != RMPC_MOD_ROWS) { err = "row description: xrow_mod must name a module of kind RMPC_MOD_ROWS"; return -1; }
  if (d.n_links < 0 || d.n_links > RMPC_MAX_LINKS || d.n_pairs < 0 || d.n_pairs > RMPC_MAX_PAIRS) { err = "links/pairs out of range"; return -1; }
  auto frame_ok = [&](int f) { return f >= 0 && f < d.n_joints; };
  if (!frame_ok(d.end_frame)) { err = "end_frame out of range"; return -1; }
  for (int j = 0; j < d.n_joints; j++) {
    M.joint_type[j] = d.joint_type[j];
    for (int c = 0; c < 3; c++) { M.joint_xyz[j][c] = d.joint_xyz[j][c]; M.joint_axis[j][c] = d.joint_axis[j][c]; }
    for (int c = 0; c < 9; c++) M.joint_rot[j][c] = d.joint_rot[j][c];
  }
  if (d.robot == RMPC_ROBOT_DIFFDRIVE) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
    for (int j = 0; j < d.n_joints; j++) {
      const double *t = d.joint_xyz[j];
      for (int r = 0; r < 3; r++) o[r] += R[3 * r] * t[0] + R[3 * r + 1] * t[1] + R[3 * r + 2] * t[2];
      double Rn[9];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          Rn[3 * r + c] = R[3 * r] * d.joint_rot[j][c] + R[3 * r + 1] * d.joint_rot[j][3 + c] + R[3 * r + 2] * d.joint_rot[j][6 + c];
      memcpy(R, Rn, sizeof R);
      for (int c = 0; c < 3; c++) M.dd_off[j][c] = o[c];
    }
  }
  M.off_r_body = d.off_r_body; M.off_obst = d.off_obst; M.off_lin = d.off_lin; M.off_wu = d.off_wu;
  M.off_goal = d.off_goal; M.off_wgoal = d.off_wgoal; M.off_wconstr = d.off_wconstr; M.off_ws = d.off_ws;
  M.has_goal = d.has_goal; M.has_avoid = d.has_avoid;
  auto off_ok = [&](int off, int len) { return off >= 0 && off + len <= d.npar; };
  if (!off_ok(d.off_wu, d.nu)) { err = "off_wu"; return -1; }
  if (d.ns && !off_ok(d.off_ws, 1)) { err = "off_ws"; return -1; }
  if (d.has_goal && (!off_ok(d.off_goal, 3) || !off_ok(d.off_wgoal, 3))) { err = "goal offsets"; return -1; }
  if (d.has_avoid && !off_ok(d.off_wconstr, d.n_modules)) { err = "off_wconstr"; return -1; }
  // general rows in module order
  int row = 0, nfk = 0;
  for (int mi = 0; mi < d.n_modules; mi++) {
    M.mod_kind[mi] = d.module_kind[mi];
    M.mod_row0[mi] = row;
    auto push = [&](int kind, int a, int bb, int poff, bool fk) -> bool {
      if (row >= kMaxRows) return false;
      M.row_kind[row] = (int8_t)kind; M.row_a[row] = (int8_t)a; M.row_b[row] = (int8_t)bb;
      M.row_poff[row] = poff; M.row_fk[row] = fk ? (int8_t)nfk++ : (int8_t)-1; M.row_mod[row] = (int8_t)mi;
      row++;
      return true;
    };
    bool ok = true;
    switch (d.module_kind[mi]) {
      case RMPC_MOD_RADIAL:
        if (!off_ok(d.off_r_body, 1) || !off_ok(d.off_obst, 4 * d.nobst)) { err = "radial offsets"; return -1; }
        for (int l = 0; l < d.n_links && ok; l++) {
          if (!frame_ok(d.link_frame[l])) { err = "link frame"; return -1; }
          for (int i = 0; i < d.nobst && ok; i++) ok = push(ROW_RADIAL, d.link_frame[l], i, 0, true);
        }
        break;
      case RMPC_MOD_LINEAR:
        if (!off_ok(d.off_r_body, 1) || !off_ok(d.off_lin, 4 * d.nobst)) { err = "linear offsets"; return -1; }
        for (int l = 0; l < d.n_links && ok; l++) {
          if (!frame_ok(d.link_frame[l])) { err = "link frame"; return -1; }
          for (int i = 0; i < d.nobst && ok; i++) ok = push(ROW_LINEAR, d.link_frame[l], i, 0, true);
        }
        break;
      case RMPC_MOD_SELFCOLLISION:
        if (d.n_pairs > 0 && !off_ok(d.off_r_body, 1)) { err = "self collision offsets"; return -1; }
        for (int pi = 0; pi < d.n_pairs && ok; pi++) {
          if (!frame_ok(d.pair_frame[pi][0]) || !frame_ok(d.pair_frame[pi][1])) { err = "pair frame"; return -1; }
          ok = push(ROW_SELF, d.pair_frame[pi][0], d.pair_frame[pi][1], 0, true);
        }
        break;
      case RMPC_MOD_JOINTLIMIT:
        if (!off_ok(d.off_lower, d.n) || !off_ok(d.off_upper, d.n)) { err = "joint limit offsets"; return -1; }
        for (int j = 0; j < d.n && ok; j++) {
          ok = push(ROW_SINGLE, j, +1, d.off_lower + j, false);
          ok = ok && push(ROW_SINGLE, j, -1, d.off_upper + j, false);
        }
        break;
      case RMPC_MOD_VELLIMIT:
        if (!off_ok(d.off_lower_vel, 2) || !off_ok(d.off_upper_vel, 2)) { err = "velocity limit offsets"; return -1; }
        for (int j = 0; j < 2 && ok; j++) {
          ok = push(ROW_SINGLE, d.nx - 2 + j, +1, d.off_lower_vel + j, false);
          ok = ok && push(ROW_SINGLE, d.nx - 2 + j, -1, d.off_upper_vel + j, false);
        }
        break;
      case RMPC_MOD_INPUTLIMIT:
        if (!off_ok(d.off_lower_u, d.nu) || !off_ok(d.off_upper_u, d.nu)) { err = "input limit offsets"; return -1; }
        for (int j = 0; j < d.nu && ok; j++) {
          ok = push(ROW_SINGLE, d.nx + d.ns + j, +1, d.off_lower_u + j, false);
          ok = ok && push(ROW_SINGLE, d.nx + d.ns + j, -1, d.off_upper_u + j, false);
        }
        break;
      case RMPC_MOD_ROWS: {
        // a module given as row descriptions (rmpc.h): variants of the six kinds through the same row tables
        int on_x = 0, on_u = 0;
        for (int r = 0; r < d.n_xrows && ok; r++) {
          if (d.xrow_mod[r] != mi) continue;
          const int a = d.xrow_a[r], b = d.xrow_b[r], po = d.xrow_poff[r];
          switch (d.xrow_kind[r]) {
            case RMPC_ROW_RADIAL:
            case RMPC_ROW_LINEAR: {
              const bool radial = d.xrow_kind[r] == RMPC_ROW_RADIAL;
              int &base = radial ? M.off_obst : M.off_lin;
              if (!frame_ok(a)) { err = "row description: frame"; return -1; }
              if (!off_ok(d.off_r_body, 1) || !off_ok(po, 4)) { err = "row description: parameter offsets"; return -1; }
              if (base < 0) base = po;   // (no module of the kind: the list starts at the first described row)
              if (po < base || (po - base) % 4 != 0 || (po - base) / 4 > 63) {
                err = "row description: a sphere / plane must lie a multiple of 4 (at most 252) parameters behind the obstacle / plane list";
                return -1;
              }
              ok = push(radial ? ROW_RADIAL : ROW_LINEAR, a, (po - base) / 4, 0, true);
              on_x++;
              break;
            }
            case RMPC_ROW_SELF:
              if (!frame_ok(a) || !frame_ok(b) || a == b) { err = "row description: pair frames"; return -1; }
              if (!off_ok(d.off_r_body, 1)) { err = "row description: r_body"; return -1; }
              ok = push(ROW_SELF, a, b, 0, true);
              on_x++;
              break;
            case RMPC_ROW_VAR:
              if (a < 0 || a >= M.nv || (d.ns && a == d.nx)) { err = "row description: variable"; return -1; }
              if (b != 1 && b != -1) { err = "row description: sign must be +1 or -1"; return -1; }
              if (!off_ok(po, 1)) { err = "row description: limit offset"; return -1; }
              ok = push(ROW_SINGLE, a, b, po, false);
              (a < d.nx ? on_x : on_u)++;
              break;
            default:
              err = "row description: unknown row kind";
              return -1;
          }
        }
        if (on_x && on_u) { err = "row description: the rows of a module must all be on states or all on inputs"; return -1; }
        break;
      }
      default:
        err = "unknown constraint module";
        return -1;
    }
    if (!ok) { err = "too many inequality rows"; return -1; }
    M.mod_rows[mi] = row - M.mod_row0[mi];
  }
  M.nh = row; M.nfk = nfk;
  for (int j = 0; j < M.nv; j++)
    if (std::isfinite(d.lb[j])) { M.lb_var[M.nlb] = (int8_t)j; M.lb_val[M.nlb] = d.lb[j]; M.nlb++; }
  for (int j = 0; j < M.nv; j++)
    if (std::isfinite(d.ub[j])) { M.ub_var[M.nub] = (int8_t)j; M.ub_val[M.nub] = d.ub[j]; M.nub++; }
  M.m = M.nh + M.nlb + M.nub;
  M.max_iter = d.max_iter > 0 ? d.max_iter : 200;
  M.tol_stat = d.tol_stat > 0 ? d.tol_stat : 1e-6;
  M.tol_eq = d.tol_eq > 0 ? d.tol_eq : 1e-8;
  M.tol_ineq = d.tol_ineq > 0 ? d.tol_ineq : 1e-8;
  M.tol_comp = d.tol_comp > 0 ? d.tol_comp : 1e-6;
  M.mu0 = d.mu0 > 0 ? d.mu0 : 1.0;
  M.acc_iters = d.acc_iters < 0 ? 0 : d.acc_iters;
  M.acc_obj_tol = d.acc_obj_tol > 0 ? d.acc_obj_tol : 1e-8;
  M.ls_max = d.ls_max > 0 ? d.ls_max : kLsMax;
  // exact curvature of the distance rows: holonomic chain, no slack, and for n <= 3 every frame a
  // distance row refers to moves affinely with q (prismatic joints, or revolute at the frame itself)
  auto affine = [&](int f) {
    for (int j = 0; j <= f; j++)
      if (d.joint_type[j] == RMPC_JOINT_REVOLUTE && j != f) return false;
    return true;
  };
  bool curv = d.robot == RMPC_ROBOT_CHAIN && d.ns == 0;
  // (the arms carry the kinematics' own second derivatives: Cfg::FKCURV)
  // (by row: the sphere and pair rows of the built-in modules and of the row-described ones alike)
  for (int r = 0; r < M.nh && curv && d.n <= 3; r++) {
    if (M.row_kind[r] == ROW_RADIAL) curv = curv && affine(M.row_a[r]);
    if (M.row_kind[r] == ROW_SELF) curv = curv && affine(M.row_a[r]) && affine(M.row_b[r]);
  }
  if (d.robot == RMPC_ROBOT_DIFFDRIVE) curv = true;   // exact second-order terms of the unicycle (Cfg::DDCURV)
  M.use_curv = curv ? 1 : 0;
  if (getenv("RMPC_NO_CURV")) M.use_curv = 0;  // debugging aid
  return 0;
}

// ---- generated views (rmpc_spec_gen.hpp) --------------------------------------------------
// Source text of the view of one descriptor: the accessors of RtView as constexpr functions over literal tables
// (doubles as hex floats: exact).  scripts/gen_specs.py writes rmpc_spec_gen.hpp from it for the shipped
// configurations; the library is then built with those views next to the runtime one.
static std::string spec_source(const rmpc_desc &d, const DevModel &M, const DevTables &T, const std::string &name) {
  std::string o;
  char buf[128];
  auto fi = [&](int v) { snprintf(buf, sizeof buf, "%d", v); return std::string(buf); };
  auto fd = [&](double v) { snprintf(buf, sizeof buf, "%a", v); return std::string(buf); };
  auto scalar = [&](const char *nm, int v) {
    o += "  __host__ __device__ static constexpr int " + std::string(nm) + "() { return " + fi(v) + "; }\n";
  };
  auto arr1 = [&](const char *nm, const int *p, int n) {
    o += "  __host__ __device__ static constexpr int " + std::string(nm) + "(int i) { constexpr int t[" + fi(n) + "] = {";
    for (int i = 0; i < n; i++) o += (i ? ", " : "") + fi(p[i]);
    o += "}; return t[i]; }\n";
  };
  auto arr2i = [&](const char *nm, const int *p, int n0, int n1) {
    o += "  __host__ __device__ static constexpr int " + std::string(nm) + "(int i, int j) { constexpr int t[" + fi(n0) + "][" + fi(n1) + "] = {";
    for (int i = 0; i < n0; i++) {
      o += (i ? ", {" : "{");
      for (int j = 0; j < n1; j++) o += (j ? ", " : "") + fi(p[i * n1 + j]);
      o += "}";
    }
    o += "}; return t[i][j]; }\n";
  };
  auto arr2d = [&](const char *nm, const double *p, int n0, int n1) {
    o += "  __host__ __device__ static constexpr double " + std::string(nm) + "(int i, int j) { constexpr double t[" + fi(n0) + "][" + fi(n1) + "] = {";
    for (int i = 0; i < n0; i++) {
      o += (i ? ", {" : "{");
      for (int j = 0; j < n1; j++) o += (j ? ", " : "") + fd(p[i * n1 + j]);
      o += "}";
    }
    o += "}; return t[i][j]; }\n";
  };
  o += "struct " + name + " {\n  static constexpr bool SPEC = true;\n";
  o += "  static constexpr int ROBOT = " + fi(d.robot) + ", NQ = " + fi(d.n) + ", NS = " + fi(d.ns) + ";\n";
  o += "  __host__ __device__ " + name + "() {}\n  __host__ __device__ " + name + "(const DevModel &, const DevTables &) {}\n";
  scalar("nslots", T.nslots);
  arr1("slot_fa", T.slot_fa, kMaxSlots);
  arr1("slot_fb", T.slot_fb, kMaxSlots);
  arr1("slot_row_begin", T.slot_row_begin, kMaxSlots + 1);
  scalar("nfkrows", T.nfkrows);
  arr1("fk_row", T.fk_row, kMaxFkRows);
  arr1("fk_kind", T.fk_kind, kMaxFkRows);
  arr1("fk_obst", T.fk_obst, kMaxFkRows);
  arr1("fk_mod", T.fk_mod, kMaxFkRows);
  arr1("fk_first", T.fk_first, kMaxFkRows);
  arr1("fk_idx", T.fk_idx, kMaxFkRows);
  arr2i("v_row", &T.v_row[0][0], RMPC_NV_MAX, kVarRows);
  arr2i("v_sgn", &T.v_sgn[0][0], RMPC_NV_MAX, kVarRows);
  arr2i("v_poff", &T.v_poff[0][0], RMPC_NV_MAX, kVarRows);
  arr2i("v_soft", &T.v_soft[0][0], RMPC_NV_MAX, kVarRows);
  arr2i("v_mod", &T.v_mod[0][0], RMPC_NV_MAX, kVarRows);
  arr2i("v_first", &T.v_first[0][0], RMPC_NV_MAX, kVarRows);
  arr2d("v_val", &T.v_val[0][0], RMPC_NV_MAX, kVarRows);
  scalar("off_r_body", M.off_r_body); scalar("off_obst", M.off_obst); scalar("off_lin", M.off_lin);
  scalar("off_wu", M.off_wu); scalar("off_goal", M.off_goal); scalar("off_wgoal", M.off_wgoal);
  scalar("off_wconstr", M.off_wconstr); scalar("off_ws", M.off_ws);
  scalar("has_goal", M.has_goal); scalar("has_avoid", M.has_avoid);
  arr1("joint_type", M.joint_type, RMPC_MAX_JOINTS);
  arr2d("joint_xyz", &M.joint_xyz[0][0], RMPC_MAX_JOINTS, 3);
  arr2d("joint_rot", &M.joint_rot[0][0], RMPC_MAX_JOINTS, 9);
  arr2d("joint_axis", &M.joint_axis[0][0], RMPC_MAX_JOINTS, 3);
  arr2d("dd_off", &M.dd_off[0][0], RMPC_MAX_JOINTS, 3);
  o += "};\n";
  return o;
}

// true when every accessor of the generated view S returns what the runtime tables hold
template <class S>
static bool spec_matches(const rmpc_desc &d, const DevModel &M, const DevTables &T) {
  if (S::ROBOT != d.robot || S::NQ != d.n || S::NS != d.ns) return false;
  bool ok = S::nslots() == T.nslots && S::nfkrows() == T.nfkrows;
  for (int i = 0; i < kMaxSlots; i++) ok = ok && S::slot_fa(i) == T.slot_fa[i] && S::slot_fb(i) == T.slot_fb[i];
  for (int i = 0; i <= kMaxSlots; i++) ok = ok && S::slot_row_begin(i) == T.slot_row_begin[i];
  for (int i = 0; i < kMaxFkRows; i++)
    ok = ok && S::fk_row(i) == T.fk_row[i] && S::fk_kind(i) == T.fk_kind[i] && S::fk_obst(i) == T.fk_obst[i] &&
         S::fk_mod(i) == T.fk_mod[i] && S::fk_first(i) == T.fk_first[i] && S::fk_idx(i) == T.fk_idx[i];
  for (int j = 0; j < RMPC_NV_MAX; j++)
    for (int u = 0; u < kVarRows; u++)
      ok = ok && S::v_row(j, u) == T.v_row[j][u] && S::v_sgn(j, u) == T.v_sgn[j][u] && S::v_poff(j, u) == T.v_poff[j][u] &&
           S::v_soft(j, u) == T.v_soft[j][u] && S::v_mod(j, u) == T.v_mod[j][u] && S::v_first(j, u) == T.v_first[j][u] &&
           S::v_val(j, u) == T.v_val[j][u];
  ok = ok && S::off_r_body() == M.off_r_body && S::off_obst() == M.off_obst && S::off_lin() == M.off_lin &&
       S::off_wu() == M.off_wu && S::off_goal() == M.off_goal && S::off_wgoal() == M.off_wgoal &&
       S::off_wconstr() == M.off_wconstr && S::off_ws() == M.off_ws && S::has_goal() == M.has_goal &&
       S::has_avoid() == M.has_avoid;
  for (int j = 0; j < RMPC_MAX_JOINTS; j++) {
    ok = ok && S::joint_type(j) == M.joint_type[j];
    for (int c = 0; c < 3; c++)
      ok = ok && S::joint_xyz(j, c) == M.joint_xyz[j][c] && S::joint_axis(j, c) == M.joint_axis[j][c] && S::dd_off(j, c) == M.dd_off[j][c];
    for (int c = 0; c < 9; c++) ok = ok && S::joint_rot(j, c) == M.joint_rot[j][c];
  }
  return ok;
}
static int find_spec(const rmpc_desc &d, const DevModel &M, const DevTables &T) {
#define RMPC_X(ID, S, R, NQ, NS)                 \
  if constexpr (variant_built<R, NQ, NS>()) {    \
    if (spec_matches<S>(d, M, T)) return ID;     \
  }
  RMPC_SPECS(RMPC_X)
#undef RMPC_X
  return -1;
}
static const char *spec_name(int id) {
#define RMPC_X(ID, S, R, NQ, NS) \
  if (id == ID) return #S;
  RMPC_SPECS(RMPC_X)
#undef RMPC_X
  return "";
}

// ---- workspace carving ---------------------------------------------------------------
struct Carver {
  char *base;
  size_t off = 0;
  explicit Carver(void *b) : base((char *)b) {}
  template <class T>
  T *take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T *p = base ? (T *)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// Host view of the stage-record layout (Cfg::R_* of the model's kernel variant).
struct RecLayout { int q, c, dg, cs, q0, q1, rc, a5, b5, rw, rs; };
static RecLayout rec_layout(const DevModel &M) {
  RecLayout L;
  const int nq2 = M.n * (M.n + 1) / 2;
  L.q = 0; L.c = nq2; L.dg = 2 * nq2; L.cs = L.dg + (M.nv - M.n);
  L.q0 = L.cs + (M.ns > 0 ? M.nv : 0); L.q1 = L.q0 + M.nv; L.rc = L.q1 + M.nv;
  L.a5 = L.rc + M.nx; L.b5 = L.a5 + 25;
  L.rw = L.a5 + (M.robot == RMPC_ROBOT_DIFFDRIVE ? 35 + 11 : 0);   // (+ Cfg::ND curvature entries)
  L.rs = (L.rw + 1 + 7) / 8 * 8;
  return L;
}

static size_t carve(const DevModel &M, int Bp, int max_passes, void *base, Ws &W) {
  Carver c(base);
  const size_t S = (size_t)M.N * Bp;  // one slot
  const int nq = M.n, nq2 = nq * (nq + 1) / 2;
  W.N = M.N; W.Bp = Bp;
  W.p = c.take<double>(S * M.npar);
  for (int i = 0; i < 2; i++) {
    W.z[i] = c.take<double>(S * M.nv);
    W.t[i] = c.take<double>(S * M.m);
    W.lam[i] = c.take<double>(S * M.m);
    W.nu[i] = c.take<double>(S * M.nx);
  }
  W.dz = c.take<double>(S * M.nv);
  W.nunew = c.take<double>(S * M.nx);
  W.rs = rec_layout(M).rs;
  W.R = c.take<double>(S * W.rs);
  W.gfa = c.take<double>(S * M.nv);
  for (int i = 0; i < 2; i++) {
    W.grow[i] = c.take<double>(S * (M.nh > 0 ? M.nh : 1));
    W.Jq[i] = c.take<double>(S * (M.nfk > 0 ? M.nfk * nq : 1));
  }
  W.kps = (M.nw * M.nx + M.nw + M.nx * (M.nx + 1) / 2 + M.nx + M.nx + 8) / 8 * 8;   // (>= one spare word behind the image: stores of idle lanes)
  W.KP = c.take<double>(S * W.kps);
  W.part = c.take<double>(S * P_COUNT);
  W.gphi = c.take<double>(S);
  W.amin_p = c.take<unsigned long long>(Bp);
  W.amin_d = c.take<unsigned long long>(Bp);
  double **per[] = {&W.mu, &W.rho, &W.phi0, &W.Dd, &W.fcur, &W.thcur, &W.logcur,
                    &W.res_stat, &W.res_eq, &W.res_ineq, &W.res_comp, &W.obj, &W.mu_hold, &W.theta_mem, &W.theta_c};
  for (auto pp : per) *pp = c.take<double>(Bp);
  int **peri[] = {&W.status, &W.iters, &W.ls, &W.cur, &W.newstep, &W.redo, &W.force_gn, &W.gn_sticky, &W.curv_fail, &W.usedc, &W.stall,
                  &W.ls0, &W.lsst, &W.curv_skip, &W.curv_back, &W.small_steps, &W.theta_clean, &W.theta_retry};
  for (auto pp : peri) *pp = c.take<int>(Bp);
  W.active_hist = c.take<int>(max_passes + 8);
  W.act_idx = c.take<int>(Bp);
  W.n_act = c.take<int>(64);
  W.orig = c.take<int>(Bp);
  W.wlam = c.take<double>(S * M.m);
  W.wnu = c.take<double>(S * M.nx);
  W.wmu = c.take<double>(Bp);
  return (c.off + 255) & ~(size_t)255;
}

static int passes_cap(const DevModel &M) { return 4 * M.max_iter + 64; }
// columns of the compact workspace (0: batches of this handle never migrate)
static int compact_columns(int max_batch) {
  return max_batch >= kMigrateMin ? ((max_batch + kDenseDiv - 1) / kDenseDiv + 63) / 64 * 64 : 0;
}

// Algorithmic bytes (DESIGN.md, section "Kernels"): what one ACTIVE lane must read and
// write by design.  Sweep / step lanes are (instance, stage) pairs, riccati lanes are
// instances (bytes already multiplied by N stages).  pack / unpack are per call.
static void fill_lane_bytes(rmpc_handle *h, int B) {
  const DevModel &M = h->M;
  const int nq2 = M.n * (M.n + 1) / 2;
  const int64_t dd = (M.robot == RMPC_ROBOT_DIFFDRIVE) ? 35 : 0;
  const int64_t sweep_rd = M.nv * 2 + M.m * 2 + M.nfk * (1 + M.n) + M.nx * 4 + M.npar + M.nx * 2 + 2;
  // stage record (k_sweep -> k_riccati): Qqq, Cqq, Dg, cs, q0, q1, rc, A5 B5, zero slot
  const int64_t rec = 2 * nq2 + (M.nv - M.n) + (M.ns ? M.nv : 0) + 2 * M.nv + M.nx + dd + 1;
  const int64_t sweep_wr = M.nv + 2 * M.m + M.nx + rec + M.nv + M.nh + M.nfk * M.n + P_COUNT;
  // gain record (k_riccati backward -> forward): K, kff, P (packed), p, rc
  const int64_t kpw = M.nw * M.nx + M.nw + M.nx * (M.nx + 1) / 2 + 2 * M.nx;
  const int64_t ric_rd = P_COUNT + 3 + rec + kpw + dd;
  const int64_t ric_wr = kpw + M.nv + M.nx;
  const int64_t step_rd = 3 * M.nv + 2 * M.m + M.nh + M.nfk * M.n;
  const int64_t step_wr = 3;  // gphi and two atomic minima (the row steps are recomputed by k_sweep, not stored)
  h->lane_bytes[K_PACK] = 16 * ((int64_t)B * (M.nx + (int64_t)M.N * (M.nv + M.npar)));
  h->lane_bytes[K_SWEEP] = 8 * (sweep_rd + sweep_wr);
  h->lane_bytes[K_RICCATI] = 8 * (int64_t)M.N * (ric_rd + ric_wr);
  h->lane_bytes[K_STEP] = 8 * (step_rd + step_wr);
  h->lane_bytes[K_UNPACK] = 16 * (int64_t)B * M.N * M.nv;
  // fused kernel, per instance: the compulsory I/O of a solve: xinit, x0, parameters in, plan and
  // four statistics out -- everything in between is the owner wavefront's private state
  h->lane_bytes[K_FUSED] = 8 * ((int64_t)M.nx + 2 * (int64_t)M.N * M.nv + (int64_t)M.N * M.npar) + 24;
}

static int launch_variant(rmpc_handle *h, const Phase &ph, int first, int pass, hipStream_t st, int which) {
  // a generated view when the descriptor's tables equal one (rmpc_create), the runtime tables otherwise
#define RMPC_X(ID, S, R, NQ, NS)                                                            \
  if constexpr (variant_built<R, NQ, NS>()) {                                               \
    if (h->spec == ID) return launch_pass<Cfg<R, NQ, NS>, S>(h, ph, first, pass, st, which); \
  }
  RMPC_SPECS(RMPC_X)
#undef RMPC_X
#define RMPC_V(ID, R, NQ, NS)                                                                         \
  if constexpr (variant_available<R, NQ, NS>()) {                                                     \
    if (h->variant == ID) return launch_pass<Cfg<R, NQ, NS>, RtView>(h, ph, first, pass, st, which);  \
  }
  RMPC_VARIANTS(RMPC_V)
#undef RMPC_V
  return fail("no kernel variant (built with RMPC_DEV_VARIANTS?)");
}

static hipEvent_t prof_event(rmpc_handle *h) {
  if (h->ev_used == h->ev.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    h->ev.push_back(e);
  }
  return h->ev[h->ev_used++];
}

struct ProfScope {
  rmpc_handle *h;
  hipStream_t st;
  int kind;
  ProfScope(rmpc_handle *h_, hipStream_t st_, int kind_) : h(h_), st(st_), kind(kind_) {
    if (h->profiling) (void)hipEventRecord(prof_event(h), st);
  }
  ~ProfScope() {
    if (h->profiling) {
      (void)hipEventRecord(prof_event(h), st);
      h->ev_kind.push_back(kind);
    }
  }
};

static void prof_collect(rmpc_handle *h) {
  for (size_t i = 0; i < h->ev_kind.size(); i++) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]);
    h->prof_ms[h->ev_kind[i]] += ms;
    h->prof_n[h->ev_kind[i]]++;
  }
  h->ev_used = 0;
  h->ev_kind.clear();
}

// ---- fused path ------------------------------------------------------------------------------------------
static int fused_columns(int max_batch) { return (max_batch + 1) / 2 * 2; }
static bool arm_fused_model(const DevModel &M) {   // = Cfg::ARM_FUSED of the model's variant
  return M.robot == RMPC_ROBOT_CHAIN && M.ns == 0 && M.nx > 8 && M.nx < 16;
}
static size_t carve_fused(const DevModel &M, int Bcap, void *base, FusedWs &F) {
  Carver c(base);
  const size_t S = (size_t)Bcap * kFusedStages;
  const bool arm = arm_fused_model(M);
  F.nv = M.nv; F.m = M.m; F.nx = M.nx; F.npar = M.npar;
  // (the arms: one spare row behind the general rows' values, where the rows that keep no value are stored)
  F.nhs = arm ? M.nh + 1 : (M.nh > 0 ? M.nh : 1); F.njqs = M.nfk > 0 ? M.nfk * M.n : 1;
  F.p = c.take<double>(S * M.npar);
  // (the arms: F.m counts the spare row behind the slacks / multipliers, where the rows a joint does not have are evaluated)
  if (arm) F.m = M.m + 1;
  for (int i = 0; i < 2; i++) {
    F.z[i] = c.take<double>(S * M.nv);
    F.t[i] = c.take<double>(S * F.m);
    F.lam[i] = c.take<double>(S * F.m);
    F.nu[i] = c.take<double>(S * M.nx);
    F.grow[i] = c.take<double>(S * F.nhs);
    F.Jq[i] = c.take<double>(S * F.njqs);
  }
  F.dz = c.take<double>(S * M.nv);
  F.nunew = c.take<double>(S * M.nx);
  F.gfa = c.take<double>(S * M.nv);
  F.wlam = c.take<double>(S * F.m);
  F.wnu = c.take<double>(S * M.nx);
  F.wmu = c.take<double>(Bcap);
  F.rs = rec_layout(M).rs;
  // (the arms: a record slot for every one of the 32 stage columns -- lanes without a stage work on the padding columns)
  F.R = c.take<double>((size_t)Bcap * (arm ? kFusedStages : M.N) * F.rs);
  F.kps = (M.nw * M.nx + M.nw + M.nx * (M.nx + 1) / 2 + M.nx + M.nx + 8) / 8 * 8;
  F.KP = c.take<double>((size_t)Bcap * M.N * F.kps);
  F.passes = c.take<int>(64);
  F.lastp = c.take<int>(Bcap);
  F.order = c.take<int>(Bcap);
  F.ckey = c.take<int>(Bcap);
  F.stamps = c.take<long long>((size_t)Bcap * 8);
  return (c.off + 255) & ~(size_t)255;
}
// k_fused_arm reads the row STRUCTURE of a joint's variables (which limit / bound rows q_a, v_a, u_a have, whether a
// row's limit is a parameter, its sign) from joint 0 and takes scalar branches on it; only row index, parameter offset
// and module differ from joint to joint.  True for every module of the reference (they loop over all joints:
// JointLimitConstraints.py:8-31, InputLimitConstraints.py:7-29, bounds mpcModel.py:91-104); checked here all the same.
static bool arm_rows_uniform(const DevModel &M, const DevTables &T) {
  const int keep = (1 << 8) | (1 << 9) | (1 << 11);
  for (int a = 1; a < M.n; a++)
    for (int c = 0; c < 3; c++)
      for (int u = 0; u < kVarRows; u++)
        if ((T.v_desc[a + c * M.n][u] & keep) != (T.v_desc[c * M.n][u] & keep)) return false;
  return true;
}
static bool fused_supported(int variant, const DevModel &M, const DevTables &T) {
  // chain n = 3 and the diff-drive base, horizons that fit the 32 lanes of an instance.  The arm stays with the pass
  // kernels: in the fused kernel (measured twice in round 2, the second time with the recursion as a real function) a
  // pass takes 670 k cycles -- its recursion on the 32 lanes of an instance alone 347 k, twice the 64-lane pass kernel
  // -- against ~480 k for the four pass kernels.  Restructuring the arm's recursion like the chain's (records straight
  // into registers, three ordering points per stage) changed nothing either (6.8 vs 7.0 ms per batch): it is bound by
  // the 7 x 7 factorisation and the cost-to-go update, not by its ordering points.
  bool ok = false;
#define RMPC_FS(ID, R, NQ, NS) \
  if (variant == ID) ok = Cfg<R, NQ, NS>::FUSED_OK || Cfg<R, NQ, NS>::ARM_FUSED;
  RMPC_VARIANTS(RMPC_FS)
#undef RMPC_FS
  if (ok && arm_fused_model(M) && !arm_rows_uniform(M, T)) ok = false;
  return ok && M.N <= kFusedStages;
}

static int launch_fused(rmpc_handle *h, int B, const double *d_xinit, const double *d_x0, const double *d_params,
                        double *d_zout, int *d_exit, int *d_iters, double *d_kkt, double *d_obj, hipStream_t st, int cap) {
  int rc = -2;
#define RMPC_X(ID, S, R, NQ, NS)                                                                                    \
  if constexpr (Cfg<R, NQ, NS>::FUSED_OK && variant_built<R, NQ, NS>()) {                                           \
    if (rc == -2 && h->spec == ID)                                                                                  \
      rc = launch_fused_t<Cfg<R, NQ, NS>, S>(h, B, d_xinit, d_x0, d_params, d_zout, d_exit, d_iters, d_kkt, d_obj, st, cap); \
  }
  RMPC_SPECS(RMPC_X)
#undef RMPC_X
#define RMPC_V(ID, R, NQ, NS)                                                                                       \
  if constexpr (Cfg<R, NQ, NS>::FUSED_OK && variant_available<R, NQ, NS>()) {                                       \
    if (rc == -2 && h->variant == ID)                                                                               \
      rc = launch_fused_t<Cfg<R, NQ, NS>, RtView>(h, B, d_xinit, d_x0, d_params, d_zout, d_exit, d_iters, d_kkt, d_obj, st, cap); \
  }
  RMPC_VARIANTS(RMPC_V)
#undef RMPC_V
#define RMPC_VA(ID, R, NQ, NS)                                                                                      \
  if constexpr (Cfg<R, NQ, NS>::ARM_FUSED && variant_available<R, NQ, NS>()) {                                      \
    if (rc == -2 && h->variant == ID)                                                                               \
      rc = launch_fused_arm_t<Cfg<R, NQ, NS>>(h, B, d_xinit, d_x0, d_params, d_zout, d_exit, d_iters, d_kkt, d_obj, st, cap); \
  }
  RMPC_VARIANTS(RMPC_VA)
#undef RMPC_VA
  if (rc == 0) {
    // the order of the NEXT warm-started launch, right behind this one (in front of it the little kernel would wait
    // for a free SIMD whenever another handle's fused launch fills the chip: 130 us in the fleet loop)
    if (h->warm_mode && !h->env_no_order)
      hipLaunchKernelGGL(k_order_t<1024>, dim3(1), dim3(1024), 0, st, (const int *)h->F.lastp, h->F.order, B);
    return 0;
  }
  if (rc != -2) return rc;
  return fail("no fused kernel for this model");
}

static int solve_device(rmpc_handle *h, int B, const double *d_xinit, const double *d_x0, const double *d_params,
                        double *d_zout, int *d_exit, int *d_iters, double *d_kkt, double *d_obj, hipStream_t st,
                        int max_passes_override) {
  if (B < 1 || B > h->max_batch) return fail("batch size out of range for this handle");
  const DevModel &M = h->M;
  HIPCHK(hipSetDevice(h->device));
  fill_lane_bytes(h, B);
  if (h->have_duals && h->duals_B != B) h->have_duals = false;   // multipliers of another batch: cold start
  if (d_params) h->packed_B = 0;   // (the workspace parameters are about to be overwritten)
  if (h->fused) {
    // one launch: every wavefront carries its two instances from the first sweep to the plan
    int cap = max_passes_override > 0 ? max_passes_override : h->max_passes;
    if (h->pass_budget > 0 && h->pass_budget < cap) cap = h->pass_budget;
    HIPCHK(hipMemsetAsync(h->F.passes, 0, 16, st));   // [0] most passes of an instance, [1] queue counter
    {
      ProfScope ps(h, st, K_FUSED);
      if (launch_fused(h, B, d_xinit, d_x0, d_params, d_zout, d_exit, d_iters, d_kkt, d_obj, st, cap)) return -1;
    }
    HIPCHK(hipGetLastError());
    h->have_duals = true; h->duals_B = B;   // (the kernel has left the multipliers in the warm-start arrays)
    h->last_passes = -1;   // on the device (rmpc_last_passes fetches it)
    if (h->profiling) {
      HIPCHK(hipStreamSynchronize(st));
      prof_collect(h);
      h->prof_bytes[K_FUSED] += (double)B * (double)h->lane_bytes[K_FUSED];
    }
    return 0;
  }
  HIPCHK(hipMemsetAsync(h->W.active_hist, 0, sizeof(int) * (h->max_passes + 8), st));
  {
    ProfScope ps(h, st, K_PACK);
    dim3 g1((B + 63) / 64, (M.N * M.npar + 63) / 64);
    if (d_params)  // nullptr: the parameters were written straight into W.p by rmpc_solve_batch_scene_device
      hipLaunchKernelGGL(k_pack, g1, dim3(256), 0, st, d_params, h->W.p, B, M.N * M.npar, M.npar, M.N, h->Bp);
    dim3 g2((B + 63) / 64, (M.N * M.nv + 63) / 64);
    hipLaunchKernelGGL(k_pack, g2, dim3(256), 0, st, d_x0, h->W.z[0], B, M.N * M.nv, M.nv, M.N, h->Bp);
    hipLaunchKernelGGL(k_init, dim3((B + 255) / 256), dim3(256), 0, st, h->W, d_xinit, B, M.nx, M.mu0,
                       (h->warm_mode && h->have_duals) ? 1 : 0);
  }
  int cap = max_passes_override > 0 ? max_passes_override : h->max_passes;
  if (h->pass_budget > 0 && h->pass_budget < cap) cap = h->pass_budget;
  int pass = 0, next_check = 8;
  Phase ph{h->W, B};
  bool migrated = false;
  const bool may_migrate = h->Bpc > 0 && B >= kMigrateMin && !h->env_no_migrate;
  // A solve with a deadline in passes (rmpc_set_pass_budget) is enqueued whole, without a host look: every kernel
  // leaves at once when the list of iterating instances is empty, so the call returns immediately and the solve is
  // ordered with the caller's stream like a fused launch (rmpc_is_async).  Without a deadline the host reads one
  // counter every few passes (it cannot know how many passes to enqueue) and moves the survivors to the compact
  // workspace.
  const bool async = h->pass_budget > 0 && max_passes_override <= 0;
  for (; pass < cap; pass++) {
    const int first = pass == 0;
    { ProfScope ps(h, st, K_SWEEP); if (launch_variant(h, ph, first, pass, st, K_SWEEP)) return -1; }
    { ProfScope ps(h, st, K_RICCATI); if (launch_variant(h, ph, first, pass, st, K_RICCATI)) return -1; }
    if (ph.B <= kCompactWaveMax) hipLaunchKernelGGL(k_compact_wave, dim3(1), dim3(64), 0, st, ph.W, ph.B, pass);
    else hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, st, ph.W, ph.B, pass);
    { ProfScope ps(h, st, K_STEP); if (launch_variant(h, ph, first, pass, st, K_STEP)) return -1; }
    if (h->profiling) HIPCHK(hipGetLastError());   // per pass when profiling is on (otherwise once after the loop)
    if (!async && pass + 1 == next_check && max_passes_override <= 0) {
      HIPCHK(hipMemcpyAsync(h->h_active, h->W.active_hist + pass, sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      const int n = *h->h_active;
      if (n == 0) { pass++; break; }
      if (may_migrate && !migrated && n * kDenseDiv <= B) {
        // k_compact has just left the compacted list of the n survivors in the batch's workspace
        hipLaunchKernelGGL(k_migrate, dim3((n + 63) / 64, M.N), dim3(64), 0, st, h->W, h->Wc, n, M.nv, M.m, M.nx, M.npar,
                           M.nh, M.nfk * M.n);
        ph = Phase{h->Wc, n};
        migrated = true;
      }
      next_check += (may_migrate && !migrated) ? 2 : 4;  // look more often while the migration is still ahead
    }
  }
  h->last_passes = async ? -2 : pass;   // (-2: on the device, rmpc_last_passes counts the non-empty passes of active_hist)
  h->last_cap = pass;
  {
    ProfScope ps(h, st, K_UNPACK);
    dim3 g((B + 63) / 64, (M.N * M.nv + 63) / 64);
    hipLaunchKernelGGL(k_unpack, g, dim3(256), 0, st, h->W, d_zout, d_exit, d_iters, d_kkt, d_obj, B, M.nv, (const int *)nullptr);
    if (migrated) {
      // the survivors' results overwrite the stale rows the first launch wrote for them
      dim3 gc((ph.B + 63) / 64, (M.N * M.nv + 63) / 64);
      hipLaunchKernelGGL(k_unpack, gc, dim3(256), 0, st, h->Wc, d_zout, d_exit, d_iters, d_kkt, d_obj, ph.B, M.nv,
                         (const int *)h->Wc.orig);
    }
  }
  if (h->warm_mode) {
    // multipliers for the next solve (the survivors' from the compact workspace, over the stale ones of the first launch)
    const int lanes = h->W.Bp * M.N;
    hipLaunchKernelGGL(k_save_duals, dim3((lanes + 255) / 256), dim3(256), 0, st, h->W, h->W, B, M.m, M.nx, (const int *)nullptr, M.mu0);
    if (migrated) {
      const int lc = h->Wc.Bp * M.N;
      hipLaunchKernelGGL(k_save_duals, dim3((lc + 255) / 256), dim3(256), 0, st, h->Wc, h->W, ph.B, M.m, M.nx, (const int *)h->Wc.orig, M.mu0);
    }
    h->have_duals = true; h->duals_B = B;
  } else {
    h->have_duals = false;
  }
  HIPCHK(hipGetLastError());
  if (h->profiling) {
    // active (instance) lanes per pass: the sweep of pass p works on what was still
    // active after pass p-1, riccati likewise; step on what riccati p left active.
    h->h_hist.resize(pass + 1);
    HIPCHK(hipMemcpyAsync(h->h_hist.data(), h->W.active_hist, sizeof(int) * pass, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    prof_collect(h);
    if (h->env_dump_hist) {  // development aid: instances still iterating after each pass
      fprintf(stderr, "rmpc active after pass:");
      for (int p = 0; p < pass; p++) fprintf(stderr, " %d", h->h_hist[p]);
      fprintf(stderr, "\n");
    }
    double act_in = 0, act_out = 0;
    for (int p = 0; p < pass; p++) {
      act_in += (p == 0) ? B : h->h_hist[p - 1];
      act_out += h->h_hist[p];
    }
    h->prof_bytes[K_PACK] += (double)h->lane_bytes[K_PACK];
    h->prof_bytes[K_UNPACK] += (double)h->lane_bytes[K_UNPACK];
    h->prof_bytes[K_SWEEP] += act_in * M.N * (double)h->lane_bytes[K_SWEEP];
    h->prof_bytes[K_RICCATI] += act_in * (double)h->lane_bytes[K_RICCATI];
    h->prof_bytes[K_STEP] += act_out * M.N * (double)h->lane_bytes[K_STEP];
  }
  return 0;
}

extern "C" {

int rmpc_version(void) { return RMPC_VERSION; }
#ifndef RMPC_SOURCE_HASH
#define RMPC_SOURCE_HASH "unhashed"
#endif
const char *rmpc_source_hash(void) { return RMPC_SOURCE_HASH; }
const char *rmpc_last_error(void) { return g_err.c_str(); }
/* a descriptor of this version, or of 0.2.0 (the struct without the xrow_* arrays at its end: no row-described modules) */
static bool take_desc(const rmpc_desc *in, rmpc_desc &full) {
  if (!in) return false;
  const int old_size = (int)offsetof(rmpc_desc, n_xrows);
  if (in->struct_size != (int)sizeof(rmpc_desc) && in->struct_size != old_size) return false;
  memset(&full, 0, sizeof full);
  memcpy(&full, in, (size_t)in->struct_size);
  full.struct_size = (int)sizeof(rmpc_desc);
  return true;
}
/* generated views: source text for one descriptor, and which view a handle runs (see rmpc.h) */
int64_t rmpc_spec_source(const rmpc_desc *desc_in, const char *name, char *out, int64_t cap) {
  rmpc_desc dfull;
  if (!name || !take_desc(desc_in, dfull)) return fail("rmpc_spec_source: bad arguments");
  const rmpc_desc *desc = &dfull;
  DevModel M;
  DevTables T;
  std::string err;
  if (build_model(*desc, M, err) != 0 || build_tables(*desc, M, T, err) != 0) return fail("invalid descriptor: " + err);
  const std::string src = spec_source(*desc, M, T, name);
  if (out && cap > (int64_t)src.size()) memcpy(out, src.c_str(), src.size() + 1);
  return (int64_t)src.size() + 1;
}
const char *rmpc_spec_name(rmpc_handle *h) { return h ? spec_name(h->spec) : ""; }
const char *rmpc_spec_for(const rmpc_desc *desc_in) {
  rmpc_desc dfull;
  if (!take_desc(desc_in, dfull)) return "";
  const rmpc_desc *desc = &dfull;
  DevModel M;
  DevTables T;
  std::string err;
  if (build_model(*desc, M, err) != 0 || build_tables(*desc, M, T, err) != 0) return "";
  return spec_name(find_spec(*desc, M, T));
}
int rmpc_desc_size(void) { return (int)sizeof(rmpc_desc); }
const char *rmpc_kernel_name(int idx) { return (idx >= 0 && idx < RMPC_NUM_KERNELS) ? kKernelNames[idx] : ""; }

int64_t rmpc_workspace_bytes(const rmpc_desc *desc_in, int max_batch) {
  rmpc_desc dfull;
  if (!take_desc(desc_in, dfull) || max_batch < 1) return -1;
  const rmpc_desc *desc = &dfull;
  DevModel M;
  std::string err;
  if (build_model(*desc, M, err) != 0) { g_err = err; return -1; }
  DevTables T;
  if (build_tables(*desc, M, T, err) != 0) { g_err = err; return -1; }
  Ws W;
  const int Bp = (max_batch + 63) / 64 * 64;
  const int Bpc = compact_columns(max_batch);
  FusedWs F;
  const size_t fused = fused_supported(variant_of(*desc), M, T) ? carve_fused(M, fused_columns(max_batch), nullptr, F) : 0;
  return (int64_t)(carve(M, Bp, passes_cap(M), nullptr, W) + (Bpc ? carve(M, Bpc, 0, nullptr, W) : 0) + fused);
}

int rmpc_create(const rmpc_desc *desc_in, int max_batch, rmpc_handle **out) {
  if (!desc_in || !out) return fail("null argument");
  rmpc_desc dfull;
  if (!take_desc(desc_in, dfull)) return fail("rmpc_desc size mismatch (ABI version?)");
  const rmpc_desc *desc = &dfull;
  if (max_batch < 1) return fail("max_batch must be >= 1");
  rmpc_handle *h = new rmpc_handle();
  h->desc = *desc;
  std::string err;
  if (build_model(*desc, h->M, err) != 0) { delete h; return fail("invalid descriptor: " + err); }
  if (build_tables(*desc, h->M, h->T, err) != 0) { delete h; return fail("invalid descriptor: " + err); }
  h->variant = variant_of(*desc);
  if (h->variant < 0) { delete h; return fail("no kernel variant for this robot (built: holonomic chains n = 2 .. 7, with the slack variable n = 3 and 7; diff-drive base without an arm)"); }
  // Generated views (rmpc_spec_gen.hpp: the point-robot configurations).  Measured in round 2: with the chip full the
  // throughput is the same as with the runtime tables (1.60-1.65 M solves/s either way: the fused kernel is bound by
  // the traffic of the iterate, not by its instruction count), one batch alone is 6 % faster (4.42 vs 4.69 ms: the
  // requests of the sweep leave ahead of the arithmetic).  RMPC_NO_SPEC=1 (read here once) forces the runtime tables.
  h->spec = getenv("RMPC_NO_SPEC") ? -1 : find_spec(*desc, h->M, h->T);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { delete h; return fail("no HIP device available"); }
  if (desc->device < 0 || desc->device >= ndev) { delete h; return fail("device ordinal out of range"); }
  h->device = desc->device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->fused_grid = 4 * cus;
    if (const char *g = getenv("RMPC_FUSED_GRID")) { const int v = atoi(g); if (v > 0) h->fused_grid = v; }   // (development switch)
  }
  h->max_batch = max_batch;
  h->Bp = (max_batch + 63) / 64 * 64;
  h->max_passes = passes_cap(h->M);
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) { delete h; return fail(std::string("hipSetDevice: ") + hipGetErrorString(e)); }
  Ws tmp;
  h->Bpc = compact_columns(max_batch);
  const size_t big = carve(h->M, h->Bp, h->max_passes, nullptr, tmp);
  const size_t small = h->Bpc ? carve(h->M, h->Bpc, 0, nullptr, tmp) : 0;
  h->fused = fused_supported(h->variant, h->M, h->T) && !getenv("RMPC_NO_FUSED");   // (debugging switch: pass kernels only)
  FusedWs ftmp;
  h->ws_bytes = big + small + (h->fused ? carve_fused(h->M, fused_columns(max_batch), nullptr, ftmp) : 0);
  e = hipMalloc(&h->ws_base, h->ws_bytes);
  if (e != hipSuccess) { delete h; return fail(std::string("hipMalloc workspace: ") + hipGetErrorString(e)); }
  e = hipMemset(h->ws_base, 0, h->ws_bytes);
  // (the fill runs on the null stream and may still be in flight when hipMemset returns; solves run on
  //  non-blocking streams that do not wait for it)
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(h->ws_base); delete h; return fail(std::string("hipMemset workspace: ") + hipGetErrorString(e)); }
  carve(h->M, h->Bp, h->max_passes, h->ws_base, h->W);
  if (h->Bpc) {
    carve(h->M, h->Bpc, 0, (char *)h->ws_base + big, h->Wc);
    h->Wc.active_hist = h->W.active_hist;  // one history per batch, whichever workspace the pass ran in
  }
  if (h->fused) carve_fused(h->M, fused_columns(max_batch), (char *)h->ws_base + big + small, h->F);
  // (the row tables, and behind them a copy of the fused workspace's pointer block: the phase functions of the
  //  fused kernel take the instance's bases from there instead of receiving two dozen pointers per call)
  // behind the row tables: the fused workspace block and a copy of the model (ArmBlock: the phase functions of the fused
  // kernels read both through uniform pointers)
  e = hipMalloc((void **)&h->d_T, sizeof(DevTables) + sizeof(ArmBlock));
  if (e == hipSuccess) e = hipMemcpy(h->d_T, &h->T, sizeof(DevTables), hipMemcpyHostToDevice);
  if (e == hipSuccess && h->fused) e = hipMemcpy((void *)(h->d_T + 1), &h->F, sizeof(FusedWs), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy((char *)(h->d_T + 1) + offsetof(ArmBlock, M), &h->M, sizeof(DevModel), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(h->ws_base); delete h; return fail(std::string("row tables: ") + hipGetErrorString(e)); }
  e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_active, sizeof(int), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_passes, sizeof(int), hipHostMallocDefault);
  if (e != hipSuccess) { rmpc_destroy(h); return fail(std::string("stream / pinned word: ") + hipGetErrorString(e)); }
  if (const char *rl = getenv("RMPC_RIC_LANE")) h->ric_lane = atoi(rl);   // (development switch)
  h->env_no_migrate = getenv("RMPC_NO_MIGRATE") != nullptr;
  h->env_arm_two_parts = getenv("RMPC_ARM_TWO_PARTS") != nullptr;   // (development switch: k_fused_arm with two parts per stage at every horizon)
  h->env_no_order = getenv("RMPC_NO_ORDER") != nullptr;   // (development switch: fused launches in index order)
  h->env_no_cold_order = getenv("RMPC_NO_COLD_ORDER") != nullptr;   // (development switch: cold fused launches in index order)
  h->env_dump_hist = getenv("RMPC_DUMP_HIST") != nullptr;
  *out = h;
  return 0;
}

void rmpc_destroy(rmpc_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();   // a solve enqueued on a caller's stream may still be reading the workspace
  void *bufs[] = {h->ws_base, (void *)h->d_T, h->d_xinit, h->d_x0, h->d_params, h->d_zout, h->d_kkt, h->d_obj, h->d_exit, h->d_iters};
  for (void *p : bufs) (void)hipFree(p);
  for (auto e : h->ev) (void)hipEventDestroy(e);
  if (h->h_active) (void)hipHostFree(h->h_active);
  if (h->h_passes) (void)hipHostFree(h->h_passes);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int ensure_staging(rmpc_handle *h) {
  if (h->d_xinit) return 0;
  const DevModel &M = h->M;
  const size_t B = h->max_batch;
  HIPCHK(hipMalloc((void **)&h->d_xinit, sizeof(double) * B * M.nx));
  HIPCHK(hipMalloc((void **)&h->d_x0, sizeof(double) * B * M.N * M.nv));
  HIPCHK(hipMalloc((void **)&h->d_params, sizeof(double) * B * M.N * M.npar));
  HIPCHK(hipMalloc((void **)&h->d_zout, sizeof(double) * B * M.N * M.nv));
  HIPCHK(hipMalloc((void **)&h->d_kkt, sizeof(double) * B));
  HIPCHK(hipMalloc((void **)&h->d_obj, sizeof(double) * B));
  HIPCHK(hipMalloc((void **)&h->d_exit, sizeof(int) * B));
  HIPCHK(hipMalloc((void **)&h->d_iters, sizeof(int) * B));
  return 0;
}

int rmpc_solve_batch(rmpc_handle *h, int B, const double *xinit, const double *x0, const double *params,
                     double *z_out, int32_t *exitflag, int32_t *iters, double *kkt_res, double *obj) {
  if (!h || !xinit || !x0 || !params || !z_out || !exitflag) return fail("null argument");
  if (B < 1 || B > h->max_batch) return fail("batch size out of range for this handle");
  HIPCHK(hipSetDevice(h->device));
  if (ensure_staging(h) != 0) return -1;
  const DevModel &M = h->M;
  hipStream_t st = h->stream;
  HIPCHK(hipMemcpyAsync(h->d_xinit, xinit, sizeof(double) * B * M.nx, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(h->d_x0, x0, sizeof(double) * (size_t)B * M.N * M.nv, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(h->d_params, params, sizeof(double) * (size_t)B * M.N * M.npar, hipMemcpyHostToDevice, st));
  if (solve_device(h, B, h->d_xinit, h->d_x0, h->d_params, h->d_zout, h->d_exit, h->d_iters, h->d_kkt, h->d_obj, st, 0) != 0)
    return -1;
  HIPCHK(hipMemcpyAsync(z_out, h->d_zout, sizeof(double) * (size_t)B * M.N * M.nv, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(exitflag, h->d_exit, sizeof(int) * B, hipMemcpyDeviceToHost, st));
  if (iters) HIPCHK(hipMemcpyAsync(iters, h->d_iters, sizeof(int) * B, hipMemcpyDeviceToHost, st));
  if (kkt_res) HIPCHK(hipMemcpyAsync(kkt_res, h->d_kkt, sizeof(double) * B, hipMemcpyDeviceToHost, st));
  if (obj) HIPCHK(hipMemcpyAsync(obj, h->d_obj, sizeof(double) * B, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

int rmpc_solve_batch_device(rmpc_handle *h, int B, const double *d_xinit, const double *d_x0, const double *d_params,
                            double *d_z_out, int32_t *d_exitflag, int32_t *d_iters, double *d_kkt_res,
                            double *d_obj, void *stream) {
  if (!h || !d_xinit || !d_x0 || !d_params || !d_z_out || !d_exitflag || !d_iters || !d_kkt_res || !d_obj)
    return fail("null argument");
  hipStream_t st = (hipStream_t)stream;   // NULL: the legacy null stream, ordered with the caller's default-stream work
  return solve_device(h, B, d_xinit, d_x0, d_params, d_z_out, d_exitflag, d_iters, d_kkt_res, d_obj, st, 0);
}

static void scene_args(const rmpc_handle *h, const rmpc_scene *s, SceneDev &S, SceneOff &O) {
  const rmpc_desc &d = h->desc;
  S.goal = s->goal; S.r_body = s->r_body; S.obst = s->obst; S.obst_dyn = s->obst_dyn;
  S.lower = s->lower_limits; S.upper = s->upper_limits; S.lower_u = s->lower_limits_u; S.upper_u = s->upper_limits_u;
  S.lower_vel = s->lower_limits_vel; S.upper_vel = s->upper_limits_vel; S.lin = s->lin_constrs;
  S.dyn_radius = s->dyn_radius; S.w = s->w; S.wu = s->wu; S.ws = s->ws;
  for (int i = 0; i < RMPC_MAX_MODULES; i++) S.wconstr[i] = s->wconstr[i];
  O.r_body = d.off_r_body; O.obst = d.off_obst; O.lin = d.off_lin; O.lower = d.off_lower; O.upper = d.off_upper;
  O.lower_u = d.off_lower_u; O.upper_u = d.off_upper_u; O.lower_vel = d.off_lower_vel; O.upper_vel = d.off_upper_vel;
  O.wu = d.off_wu; O.goal = d.has_goal ? d.off_goal : -1; O.wgoal = d.has_goal ? d.off_wgoal : -1;
  O.wconstr = d.has_avoid ? d.off_wconstr : -1; O.ws = d.ns ? d.off_ws : -1;
  O.n = d.n; O.nu = d.nu; O.nobst = d.nobst; O.n_modules = d.n_modules; O.npar = d.npar; O.N = d.N; O.dt = d.dt;
}

int rmpc_pack_scene_device(rmpc_handle *h, int B, const rmpc_scene *scene, double *d_params, void *stream) {
  if (!h || !scene || !d_params) return fail("null argument");
  if (scene->struct_size != (int)sizeof(rmpc_scene)) return fail("rmpc_scene size mismatch");
  if (B < 1 || B > h->max_batch) return fail("batch size out of range for this handle");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;   // NULL: the legacy null stream, ordered with the caller's default-stream work
  SceneDev S; SceneOff O;
  scene_args(h, scene, S, O);
  const int lanes = B * h->M.N;
  hipLaunchKernelGGL((k_scene<0>), dim3((lanes + 255) / 256), dim3(256), 0, st, S, O, d_params, B, h->Bp);
  HIPCHK(hipGetLastError());
  return 0;
}

int rmpc_pack_scene_workspace(rmpc_handle *h, int B, const rmpc_scene *scene, void *stream) {
  if (!h || !scene) return fail("null argument");
  if (scene->struct_size != (int)sizeof(rmpc_scene)) return fail("rmpc_scene size mismatch");
  if (B < 1 || B > h->max_batch) return fail("batch size out of range for this handle");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;   // NULL: the legacy null stream, ordered with the caller's default-stream work
  SceneDev S; SceneOff O;
  scene_args(h, scene, S, O);
  if (h->fused) {
    const int lanes = B * h->M.N;
    hipLaunchKernelGGL((k_scene<2>), dim3((lanes + 255) / 256), dim3(256), 0, st, S, O, h->F.p, B, h->Bp);
  } else {
    const int lanes = h->Bp * h->M.N;
    hipLaunchKernelGGL((k_scene<1>), dim3((lanes + 255) / 256), dim3(256), 0, st, S, O, h->W.p, B, h->Bp);
  }
  HIPCHK(hipGetLastError());
  h->packed_B = B;
  return 0;
}

int rmpc_solve_batch_packed_device(rmpc_handle *h, int B, const double *d_xinit, const double *d_x0, double *d_z_out,
                                   int32_t *d_exitflag, int32_t *d_iters, double *d_kkt_res, double *d_obj, void *stream) {
  if (!h || !d_xinit || !d_x0 || !d_z_out || !d_exitflag || !d_iters || !d_kkt_res || !d_obj) return fail("null argument");
  if (h->packed_B != B) return fail("no parameters of this batch size in the workspace (rmpc_pack_scene_workspace first)");
  return solve_device(h, B, d_xinit, d_x0, nullptr, d_z_out, d_exitflag, d_iters, d_kkt_res, d_obj, (hipStream_t)stream, 0);
}

int rmpc_solve_batch_scene_device(rmpc_handle *h, int B, const rmpc_scene *scene, const double *d_xinit,
                                  const double *d_x0, double *d_z_out, int32_t *d_exitflag, int32_t *d_iters,
                                  double *d_kkt_res, double *d_obj, void *stream) {
  if (!h || !scene || !d_xinit || !d_x0 || !d_z_out || !d_exitflag || !d_iters || !d_kkt_res || !d_obj)
    return fail("null argument");
  if (rmpc_pack_scene_workspace(h, B, scene, stream)) return -1;
  return rmpc_solve_batch_packed_device(h, B, d_xinit, d_x0, d_z_out, d_exitflag, d_iters, d_kkt_res, d_obj, stream);
}

int rmpc_advance_device_flags(rmpc_handle *h, int B, const double *d_z_prev, const int32_t *d_exitflag, double *d_xinit,
                              double *d_x0, int previous_plan, void *stream) {
  if (!h || !d_z_prev || !d_xinit || !d_x0) return fail("null argument");
  if (B < 1 || B > h->max_batch) return fail("batch size out of range for this handle");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;   // NULL: the legacy null stream, ordered with the caller's default-stream work
  const int *ef = (const int *)d_exitflag;
#define RMPC_AD(ID, R, NQ, NS)                                                                                       \
  if constexpr (variant_available<R, NQ, NS>()) {                                                                    \
    if (h->variant == ID) launch_advance<Cfg<R, NQ, NS>>(h, B, d_z_prev, ef, d_xinit, d_x0, previous_plan, st);      \
  }
  RMPC_VARIANTS(RMPC_AD)
#undef RMPC_AD
  HIPCHK(hipGetLastError());
  return 0;
}

int rmpc_advance_device(rmpc_handle *h, int B, const double *d_z_prev, double *d_xinit, double *d_x0,
                        int previous_plan, void *stream) {
  return rmpc_advance_device_flags(h, B, d_z_prev, nullptr, d_xinit, d_x0, previous_plan, stream);
}

int rmpc_retarget_device(rmpc_handle *h, int B, const rmpc_retarget *r, void *stream) {
  if (!h || !r) return fail("null argument");
  if (r->struct_size != (int)sizeof(rmpc_retarget)) return fail("rmpc_retarget.struct_size mismatch");
  if (!r->xinit || !r->x0 || !r->goal || !r->goal_pool || !r->cursor || !r->dwell || !r->x_start) return fail("null argument");
  if (B < 1 || B > h->max_batch) return fail("batch size out of range for this handle");
  if (r->pool_len < 1) return fail("goal pool must hold at least one goal per instance");
  if (!h->desc.has_goal) return fail("the model has no GoalReaching objective");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  RetargetDev R;
  R.xinit = r->xinit; R.x0 = r->x0; R.goal = r->goal; R.exitflag = (const int *)r->exitflag; R.iters = (const int *)r->iters;
  R.pool = r->goal_pool; R.x_start = r->x_start; R.P = r->pool_len; R.lower = r->lower_limits; R.upper = r->upper_limits;
  R.cursor = (int *)r->cursor; R.dwell = (int *)r->dwell; R.failrun = (int *)r->failrun;
  R.tol = r->tol; R.settle_vel = r->settle_vel; R.settle_min_dwell = r->settle_min_dwell; R.max_dwell = r->max_dwell;
  R.fail_reset_after = r->fail_reset_after; R.counts = (long long *)r->counts;
  R.wmu = h->warm_mode ? (h->fused ? h->F.wmu : h->W.wmu) : nullptr;
  R.wmu_regoal = r->mu_regoal > 0.0 ? r->mu_regoal / kWarmKappa : 0.0;
#define RMPC_RT(ID, R_, NQ, NS)                                                   \
  if constexpr (variant_available<R_, NQ, NS>()) {                                \
    if (h->variant == ID) launch_retarget<Cfg<R_, NQ, NS>>(h, B, R, st);          \
  }
  RMPC_VARIANTS(RMPC_RT)
#undef RMPC_RT
  HIPCHK(hipGetLastError());
  return 0;
}

int rmpc_advance_obstacles_device(int B, int nobst, double dt, double arena, double *d_obst_dyn, void *stream) {
  if (B < 1 || nobst < 1 || !d_obst_dyn) return fail("bad argument");
  const int n = B * nobst;
  hipLaunchKernelGGL(k_obst_advance, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_obst_dyn, n, dt, arena);
  HIPCHK(hipGetLastError());
  return 0;
}

int rmpc_free_space_device(int B, int N, int P, int K, double max_radius, const double *d_points,
                           const double *d_seeds, double *d_planes, void *stream) {
  if (!d_points || !d_seeds || !d_planes) return fail("null argument");
  if (B < 1 || N < 1 || K < 1 || P < 1 || P > 64) return fail("free space decomposition: need 1 <= P <= 64 points, K >= 1");
  hipLaunchKernelGGL(k_fsd, dim3((B * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_points, d_seeds, d_planes,
                     B, N, P, K, max_radius);
  HIPCHK(hipGetLastError());
  return 0;
}

int rmpc_set_warm_start(rmpc_handle *h, int mode) {
  if (!h) return fail("null handle");
  h->warm_mode = mode ? 1 : 0;
  h->have_duals = false;
  return 0;
}

int rmpc_set_pass_budget(rmpc_handle *h, int passes) {
  if (!h) return fail("null handle");
  if (passes < 0) return fail("pass budget must be >= 0");
  h->pass_budget = passes;
  return 0;
}

int rmpc_is_fused(const rmpc_handle *h) { return (h && h->fused) ? 1 : 0; }
const char *rmpc_fused_kernel_name(const rmpc_handle *h) {
  if (!h || !h->fused) return "";
  return arm_fused_model(h->M) ? "k_fused_arm" : "k_fused";
}
int rmpc_is_async(const rmpc_handle *h) { return (h && (h->fused || h->pass_budget > 0)) ? 1 : 0; }

int rmpc_set_profiling(rmpc_handle *h, int enable) {
  if (!h) return fail("null handle");
  h->profiling = enable != 0;
  for (int i = 0; i < RMPC_NUM_KERNELS; i++) { h->prof_ms[i] = 0; h->prof_n[i] = 0; h->prof_bytes[i] = 0; }
  return 0;
}

int rmpc_get_profile(rmpc_handle *h, double *total_ms, int64_t *launches, double *total_alg_bytes,
                     int64_t *full_launch_bytes) {
  if (!h) return fail("null handle");
  const int64_t L = (int64_t)h->max_batch * h->M.N;
  for (int i = 0; i < RMPC_NUM_KERNELS; i++) {
    if (total_ms) total_ms[i] = h->prof_ms[i];
    if (launches) launches[i] = h->prof_n[i];
    if (total_alg_bytes) total_alg_bytes[i] = h->prof_bytes[i];
    if (full_launch_bytes)
      full_launch_bytes[i] = (i == K_SWEEP || i == K_STEP) ? L * h->lane_bytes[i]
                             : (i == K_RICCATI || i == K_FUSED) ? (int64_t)h->max_batch * h->lane_bytes[i]
                                                                : h->lane_bytes[i];
  }
  return 0;
}

int rmpc_last_passes(rmpc_handle *h) {
  if (!h) return -1;
  if (h->fused && h->last_passes < 0) {
    // the fused kernel counts on the device: the most passes any instance of the last launch needed
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(h->h_passes, h->F.passes, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
      return -1;
    h->last_passes = *h->h_passes;
  }
  if (!h->fused && h->last_passes == -2) {
    // a solve enqueued without a host look: passes after which instances were still iterating, plus the one that
    // found them all stopped
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
    h->h_hist.resize(h->last_cap + 1);
    if (hipMemcpy(h->h_hist.data(), h->W.active_hist, sizeof(int) * h->last_cap, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int p = 0;
    while (p < h->last_cap && h->h_hist[p] > 0) p++;
    h->last_passes = p < h->last_cap ? p + 1 : h->last_cap;
  }
  return h->last_passes;
}

/* development aid (builds with -DRMPC_STAMPS): per-block phase cycles of the last fused launch, 8 words per block */
#ifdef RMPC_STAMPS
int rmpc_debug_sweep_stamps(long long *out) {   // reads and clears k_sweep's section counters
  long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rmpc::g_sst), sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(rmpc::g_sst), z, sizeof(z)) != hipSuccess;
}
#endif
#ifdef RMPC_RIC_STAMPS
int rmpc_debug_ric_stamps(long long *out) {   // reads and clears the recursion's phase counters
  long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rmpc::g_rst), sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(rmpc::g_rst), z, sizeof(z)) != hipSuccess;
}
#endif
int rmpc_debug_fused_stamps(rmpc_handle *h, long long *out, int nblocks) {
  if (!h || !h->fused) return fail("no fused workspace");
  if (nblocks > fused_columns(h->max_batch)) return fail("too many blocks");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->F.stamps, sizeof(long long) * 8 * (size_t)nblocks, hipMemcpyDeviceToHost));
  return 0;
}

/* test aid: NaN patterns into everything a solve could read without having written it -- the LDS of every CU (blocks
 * that own all 160 KB of a CU, then 64 KB blocks, so that whatever offset a solver kernel's allocation starts at has
 * been covered), the scratch (private) memory the wavefronts spill to, and the handle's whole device workspace (all
 * bytes 0xff: NaN as a double, -1 as an int; stored multipliers and the parameters of rmpc_pack_scene_workspace are
 * thereby forgotten).  A test then shows that no result depends on what a kernel finds in any of them. */
__global__ __launch_bounds__(64) void k_poison_lds(double *sink, int nbytes) {
  extern __shared__ double pl[];
  const int n = nbytes / 8;
  for (int i = threadIdx.x; i < n; i += 64) pl[i] = __longlong_as_double(0x7ff8dead0000beefLL);
  __syncthreads();
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = pl[n - 1];
}
__global__ __launch_bounds__(64) void k_poison_scratch(double *sink, int salt) {
  // 4 KB of private memory per lane, indexed at run time (so that it lives in scratch), filled with NaN patterns
  // (launched with 40 KB of LDS per block: four wavefronts per CU, like the solver kernels that spill)
  volatile double buf[512];
  for (int i = 0; i < 512; i++) buf[i] = __longlong_as_double(0x7ff8dead0000beefLL + i);
  if (sink && salt == 12345) sink[threadIdx.x] = buf[(salt + threadIdx.x) % 512];
}
int rmpc_debug_poison_lds(rmpc_handle *h) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const int sizes[2] = {160 * 1024, 64 * 1024};
  for (int si = 0; si < 2; si++) {
    HIPCHK(hipFuncSetAttribute((const void *)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, sizes[si]));
    for (int rep = 0; rep < 4; rep++)
      hipLaunchKernelGGL(k_poison_lds, dim3(4096), dim3(64), sizes[si], h->stream, (double *)nullptr, sizes[si]);
    HIPCHK(hipGetLastError());
  }
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_poison_scratch, dim3(8192), dim3(64), 40 * 1024, h->stream, (double *)nullptr, rep);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(h->ws_base, 0xff, h->ws_bytes, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->have_duals = false;
  h->packed_B = 0;
  return 0;
}

int rmpc_debug_sweep(rmpc_handle *h, int B, const double *xinit, const double *x0, const double *params,
                     double *out_Q, double *out_q0, double *out_q1, double *out_rc, double *out_g, double *out_f) {
  if (!h) return fail("null handle");
  if (B < 1 || B > h->max_batch) return fail("batch size out of range for this handle");
  HIPCHK(hipSetDevice(h->device));
  if (ensure_staging(h) != 0) return -1;
  const DevModel &M = h->M;
  hipStream_t st = h->stream;
  HIPCHK(hipMemcpyAsync(h->d_xinit, xinit, sizeof(double) * B * M.nx, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(h->d_x0, x0, sizeof(double) * (size_t)B * M.N * M.nv, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(h->d_params, params, sizeof(double) * (size_t)B * M.N * M.npar, hipMemcpyHostToDevice, st));
  dim3 g1((B + 63) / 64, (M.N * M.npar + 63) / 64);
  hipLaunchKernelGGL(k_pack, g1, dim3(256), 0, st, h->d_params, h->W.p, B, M.N * M.npar, M.npar, M.N, h->Bp);
  dim3 g2((B + 63) / 64, (M.N * M.nv + 63) / 64);
  hipLaunchKernelGGL(k_pack, g2, dim3(256), 0, st, h->d_x0, h->W.z[0], B, M.N * M.nv, M.nv, M.N, h->Bp);
  hipLaunchKernelGGL(k_init, dim3((B + 255) / 256), dim3(256), 0, st, h->W, h->d_xinit, B, M.nx, M.mu0, 0);
  h->have_duals = false;
  {
    const int wm = h->warm_mode;
    h->warm_mode = 0;
    const int rc = launch_variant(h, Phase{h->W, B}, 1, 0, st, K_SWEEP);
    h->warm_mode = wm;
    if (rc) return -1;
  }
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  // gather SoA -> instance-major on the host (debug path, not timed)
  const size_t S = (size_t)M.N * h->Bp;
  auto fetch = [&](const double *dptr, size_t slots, std::vector<double> &v) -> int {
    v.resize(S * slots);
    HIPCHK(hipMemcpy(v.data(), dptr, sizeof(double) * S * slots, hipMemcpyDeviceToHost));
    return 0;
  };
  std::vector<double> R, gr, part;
  const int nq = M.n, nv = M.nv;
  const RecLayout L = rec_layout(M);
  if (fetch(h->W.R, L.rs, R) || fetch(h->W.grow[1], M.nh > 0 ? M.nh : 1, gr)  /* first pass: cur = 0, written to buffer 1 */ ||
      fetch(h->W.part, P_COUNT, part))
    return -1;
  auto at = [&](const std::vector<double> &v, int slot, int k, int b) { return v[((size_t)slot * M.N + k) * h->Bp + b]; };
  auto rec = [&](int off, int k, int b) { return R[((size_t)b * M.N + k) * L.rs + off]; };
  for (int b = 0; b < B; b++)
    for (int k = 0; k < M.N; k++) {
      const size_t sb = (size_t)b * M.N + k;
      if (out_Q) {
        double *Q = out_Q + sb * nv * nv;
        for (int i = 0; i < nv * nv; i++) Q[i] = 0.0;
        int s = 0;
        for (int a = 0; a < nq; a++)
          for (int c = a; c < nq; c++) { double v = rec(L.q + s++, k, b); Q[a * nv + c] = v; Q[c * nv + a] = v; }
        for (int j = nq; j < nv; j++) Q[j * nv + j] = rec(L.dg + j - nq, k, b);
        if (M.ns)
          for (int j = 0; j < nv; j++)
            if (j != M.nx) { double v = rec(L.cs + j, k, b); Q[j * nv + M.nx] = v; Q[M.nx * nv + j] = v; }
      }
      for (int j = 0; j < nv; j++) {
        if (out_q0) out_q0[sb * nv + j] = rec(L.q0 + j, k, b);
        if (out_q1) out_q1[sb * nv + j] = rec(L.q1 + j, k, b);
      }
      if (out_rc) for (int j = 0; j < M.nx; j++) out_rc[sb * M.nx + j] = (k < M.N - 1) ? rec(L.rc + j, k, b) : 0.0;
      if (out_g) for (int j = 0; j < M.nh; j++) out_g[sb * M.nh + j] = at(gr, j, k, b);
      if (out_f) out_f[sb] = at(part, P_F, k, b);
    }
  return 0;
}

}  // extern "C"
#endif  // RMPC_TU_MAIN
