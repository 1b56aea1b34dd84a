// rmpc_arm_fused.hpp -- the arms in ONE launch: a wavefront owns an instance from the first sweep to the plan.
// Included by rmpc_kernels.hip (behind k_fused: it uses that kernel's workspace, queue and decision logic).
//
// Round 4.  The pass kernels of the arm (k_sweep, k_riccati, k_compact, k_step: four launches per pass) spend, per
// instance and pass, ~70 k SIMD cycles in a sweep of 28 k instructions per lane with 420 B of scratch, 94 k in the
// recursion and 24 k in the step kernel, queue behind each other when several batches are in flight, and exchange the
// step and the partial sums through HBM.  Both earlier attempts to put the arm into k_fused kept that kernel's
// geometry -- 32 lanes per instance, lane = stage -- and were slower than the pass kernels: one lane evaluating a whole
// stage of the arm is what does not fit.  Here
//   * an instance has the whole wavefront: the recursion runs on 64 lanes (riccati_recursion's arm path, round 4);
//   * a stage is evaluated by P lanes ("parts", P = 2 for horizons up to 32): a part owns the kinematic slots
//     s = p, p + P, ... (frame points, their Jacobians, the distance rows and their 7 x 7 curvature blocks) and the
//     joints a = p, p + P, ... (the variables q_a, v_a, u_a: limit and bound rows, dynamics defect, stationarity,
//     record entries).  What the parts of a stage have to add up -- the two q blocks and four q vectors -- crosses
//     the lanes by DPP moves (dpp_move: quad permutations, no LDS round trip);
//   * tables indexed by a per-lane joint or row come as one packed word per row (DevTables::v_desc / fk_desc);
//   * the step dz | nu+ stays in LDS between the recursion and the sweep, the partial sums in registers; the stage
//     records and the gain images that do not fit LDS (the last stages) go through the instance's block in global memory.
// Same algorithm, same decisions (inst_decide) as the pass kernels; sums are formed in another order.
#pragma once
// (included inside namespace rmpc)

#ifdef RMPC_STAMPS
// development aid: cycles per section of the part-wise sweep, summed over the calls of all wavefronts into g_sst
// ([0] q + chain walk, [1] slots, [2] sums over the parts, [3] joints, [4] block stores, [5] step lengths + reduction,
//  [6] reduction of the partials, [7] calls)
#define AP_STAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); ap_acc[i] += t_ - ap_t0; ap_t0 = t_; } while (0)
#else
#define AP_STAMP(i)
#endif
// Element `slot` of an instance-block array for the lane's stage: the arrays of the fused layout are [slot][32 stages],
// their bases are uniform (one instance per wavefront) -- 32-bit lane offsets on scalar bases, so that a request is
// "scalar base + vector offset" instead of a 64-bit address computed per lane.
#define AIDX(slot) ((unsigned)(slot) * (unsigned)kFusedStages + loff)
#define AIDX1(slot) ((unsigned)(slot) * (unsigned)kFusedStages + loff1)
#define AIDXL(slot) ((unsigned)(slot) * (unsigned)kFusedStages + loffl)
template <int P>
__device__ __forceinline__ double part_sum(double v) {   // sum over the P aligned consecutive lanes of a stage
  if constexpr (P == 3) {
    // three parts (horizons of 21 stages at most: 63 of the 64 lanes carry a row instead of 40 at two parts): the
    // group is no power of two, so the terms travel through the crossbar (ds_bpermute), and every lane adds them in
    // the same order -- the three lanes of a stage must hold the same bits
    const int base = ((int)threadIdx.x / 3) * 3;
    const double v0 = __shfl(v, base, 64), v1 = __shfl(v, base + 1, 64), v2 = __shfl(v, base + 2, 64);
    return (v0 + v1) + v2;
  } else {
    if constexpr (P >= 2) v += dpp_move<0xB1>(v);
    if constexpr (P >= 4) v += dpp_move<0x4E>(v);
    return v;
  }
}
// v[p] for a per-lane p out of statically indexed values
template <int P>
__device__ __forceinline__ double part_pick(const int p, const double v0, const double v1, const double v2, const double v3) {
  if constexpr (P == 1) return v0;
  else if constexpr (P == 2) return (p & 1) ? v1 : v0;
  else if constexpr (P == 3) return p == 0 ? v0 : (p == 1 ? v1 : v2);
  else return (p & 2) ? ((p & 1) ? v3 : v2) : ((p & 1) ? v1 : v0);
}

// ---------------------------------------------------------------------------------------------------------------------
// sweep_part: what sweep_body does for a stage, for part p of the stage (holonomic chains without slack, FKCURV).
// Addressing as in SweepIO.  Rows that a lane does not have are skipped by branches (the same for every lane in
// practice: all joints carry the same modules).  The t / lambda arrays have one spare row behind the last (n_rows()).
// ---------------------------------------------------------------------------------------------------------------------
template <class C, int P, class RP, class SP, class V, int FIRSTC>
__device__ __forceinline__ void sweep_part(const SweepK M, const V &v, const SweepIO<RP, SP> &io, const int k, const int p,
                                           const bool nostep, const double alpha, const double adual, const double mu,
                                           Partials &out
#ifdef RMPC_STAMPS
                                           , long long (&ap_acc)[8], long long &ap_t0
#endif
                                           ) {
  static_assert(C::ROBOT == RMPC_ROBOT_CHAIN && C::NS == 0 && C::FKCURV, "part-wise sweep: the arms (holonomic chain, no slack)");
  constexpr int NQ = C::NQ, NX = C::NX, NQ2 = C::NQ2;
  constexpr int NJP = (NQ + P - 1) / P, NSP = (kMaxSlots + P - 1) / P;
  static_assert(NQ >= P, "every part owns a joint");
  constexpr bool first = FIRSTC != 0;
  auto qtri = [](int a, int c) __attribute__((always_inline)) { return a * NQ - a * (a - 1) / 2 + (c - a); };
  const int N = M.N;
  const unsigned loff = io.loff;
  const gdouble *__restrict__ zc = io.zc;
  const gdouble *__restrict__ tc = io.tc;
  const gdouble *__restrict__ lc = io.lc;
  const gdouble *__restrict__ nc = io.nc;
  gdouble *__restrict__ zn = io.zn;
  gdouble *__restrict__ tn = io.tn;
  gdouble *__restrict__ ln = io.ln;
  gdouble *__restrict__ nn = io.nn;
  const gdouble *__restrict__ pp = io.pp;
  const SP *__restrict__ dzp = io.dzp;
  const SP *__restrict__ nup = io.nup;
  const gdouble *__restrict__ gro = io.gro;
  const gdouble *__restrict__ jqo = io.jqo;
  gdouble *__restrict__ grn = io.grn;
  gdouble *__restrict__ jqn = io.jqn;
  gdouble *__restrict__ gfa = io.gfa;
  RP *__restrict__ rec = io.rec;
  const unsigned loff1 = loff + (k < N - 1 ? io.kstride : 0u);
  const bool warm = first && (io.warm != 0);
  const gdouble *__restrict__ lsrc = warm ? io.wl : lc;
  const unsigned loffl = warm ? loff1 : loff;
  const size_t SSd = io.SSd;
  const unsigned loffd = io.loffd, loffd1 = io.loffd + (k < N - 1 ? io.kstrided : 0u);
  const double al = alpha, adl = adual;
  const double hh = M.dt, hh2 = 0.5 * M.dt * M.dt;
  auto PR = [&](int off) __attribute__((always_inline)) -> double { return pp[AIDX(off)]; };

  double f = 0.0, theta = 0.0, rineq = 0.0, rcomp = 0.0, sumc = 0.0, minc = 1e300, lprod = 1.0, rstat = 0.0, req = 0.0;
  int lexp = 0, bad = 0;
  struct RowW { double sig, ca, cb, lv; };
  auto row_core = [&](const int i, const double g, const double tcv, const double lcv, const double gold, const double gdz)
                      __attribute__((always_inline)) -> RowW {
    double tv, lv;
    if (first) {
      const double tmin = warm ? kWarmTMin : kTMin;
      tv = g > tmin ? g : tmin;
      lv = mu * frcp(tv);
      if (warm) lv = lcv > lv ? lcv : lv;
    } else {
      const double dtv = gdz + (gold - tcv);
      const double dlv = (mu - tcv * lcv - lcv * dtv) * frcp(tcv);
      tv = nostep ? tcv : tcv + al * dtv;
      lv = nostep ? lcv : lcv + adl * dlv;
    }
    tn[AIDX(i)] = tv;
    ln[AIDX(i)] = lv;
    const double rg = g - tv;
    theta += fabs(rg);
    bad |= (int)!(tv > 0.0);
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

  // ---- q of the trial point and its step: every part needs them for its frames -------------------------------
  double qz[NQ], qdz[NQ];
#pragma unroll
  for (int a = 0; a < NQ; a++) {
    const double zo = zc[AIDX(a)];
    qdz[a] = dzp[(size_t)a * SSd + loffd];
    qz[a] = nostep ? zo : zo + al * qdz[a];
  }
  const double rbody = (v.off_r_body() >= 0) ? PR(v.off_r_body()) : 0.0;

  // ---- kinematics: one walk of the chain, the points of this part's slots captured on the way -------------------
  const int nsl = v.nslots();
  int sfa[NSP], sfb[NSP], srb[NSP], sre[NSP];
  bool sv[NSP];
#pragma unroll
  for (int s = 0; s < NSP; s++) {
    const int sl = p + P * s;
    sv[s] = sl < nsl;
    int fa = v.slot_fa(0), fb = v.slot_fb(0), rb0 = v.slot_row_begin(0), re0 = v.slot_row_begin(1);
#pragma unroll
    for (int t = 1; t < kMaxSlots; t++) {
      fa = sl == t ? v.slot_fa(t) : fa; fb = sl == t ? v.slot_fb(t) : fb;
      rb0 = sl == t ? v.slot_row_begin(t) : rb0; re0 = sl == t ? v.slot_row_begin(t + 1) : re0;
    }
    sfa[s] = fa; sfb[s] = fb; srb[s] = rb0; sre[s] = re0;
  }
  // (descriptor of the first row of every slot: requested now, so that the row's own requests need not wait for it)
  int sd0[NSP];
  const int nfkr = v.nfkrows();
#pragma unroll
  for (int s = 0; s < NSP; s++) sd0[s] = v.fk_desc(srb[s] < nfkr ? srb[s] : 0);
  Vec3 oj[NQ], aj[NQ], pa[NSP], pb[NSP];
#pragma unroll
  for (int s = 0; s < NSP; s++) { pa[s] = {0, 0, 0}; pb[s] = {0, 0, 0}; }
  {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    Vec3 o = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      const double t[3] = {v.joint_xyz(j, 0), v.joint_xyz(j, 1), v.joint_xyz(j, 2)};
      o.x += R[0] * t[0] + R[1] * t[1] + R[2] * t[2];
      o.y += R[3] * t[0] + R[4] * t[1] + R[5] * t[2];
      o.z += R[6] * t[0] + R[7] * t[1] + R[8] * t[2];
      {
        double Rj[9];
#pragma unroll
        for (int c = 0; c < 9; c++) Rj[c] = v.joint_rot(j, c);
        Kin<C>::mul33(R, Rj);
      }
      const double ax[3] = {v.joint_axis(j, 0), v.joint_axis(j, 1), v.joint_axis(j, 2)};
      const Vec3 a = {R[0] * ax[0] + R[1] * ax[1] + R[2] * ax[2], R[3] * ax[0] + R[4] * ax[1] + R[5] * ax[2],
                      R[6] * ax[0] + R[7] * ax[1] + R[8] * ax[2]};
      aj[j] = a;
      oj[j] = o;
      if (v.joint_type(j) == RMPC_JOINT_REVOLUTE) {
        double Rq[9];
        Kin<C>::rodrigues(ax, qz[j], Rq);
        Kin<C>::mul33(R, Rq);
      } else if (v.joint_type(j) == RMPC_JOINT_PRISMATIC) {
        o.x += a.x * qz[j];
        o.y += a.y * qz[j];
        o.z += a.z * qz[j];
      }
#pragma unroll
      for (int s = 0; s < NSP; s++) {
        const bool ca = sfa[s] == j, cb = sfb[s] == j;
        pa[s] = {ca ? o.x : pa[s].x, ca ? o.y : pa[s].y, ca ? o.z : pa[s].z};
        pb[s] = {cb ? o.x : pb[s].x, cb ? o.y : pb[s].y, cb ? o.z : pb[s].z};
      }
    }
  }

  AP_STAMP(0);
  // ---- this part's slots: GoalReaching (slot 0), distance rows, curvature blocks -----------------------------------
  double bq[NQ2], bc[NQ2], qgf[NQ], qq0[NQ], qq1[NQ], qrs[NQ];
#pragma unroll
  for (int e = 0; e < NQ2; e++) { bq[e] = 0.0; bc[e] = 0.0; }
#pragma unroll
  for (int a = 0; a < NQ; a++) { qgf[a] = 0.0; qq0[a] = 0.0; qq1[a] = 0.0; qrs[a] = 0.0; }
  const int rows_max = v.slot_rows_max();
  auto addsym = [](double (&T)[6], const double w, const Vec3 &n) __attribute__((always_inline)) {
    const double wx = w * n.x, wy = w * n.y, wz = w * n.z;
    T[0] += wx * n.x; T[1] += wx * n.y; T[2] += wx * n.z; T[3] += wy * n.y; T[4] += wy * n.z; T[5] += wz * n.z;
  };
  auto symv = [](const double (&T)[6], const Vec3 &x) __attribute__((always_inline)) -> Vec3 {
    return {T[0] * x.x + T[1] * x.y + T[2] * x.z, T[1] * x.x + T[3] * x.y + T[4] * x.z, T[2] * x.x + T[4] * x.y + T[5] * x.z};
  };
  // requests of the first distance row of every slot of this part (the usual case: one row per slot), all slots at once
  struct FkRowIn { double tcv, lcv, gold, jo[NQ], op[4], wir; };
  auto fk_row_load = [&](const int d, FkRowIn &in) __attribute__((always_inline)) {
    const int i = d & 255, kind = (d >> 8) & 3, ob = (d >> 10) & 63, mod = (d >> 16) & 7, fi = (d >> 20) & 63;
    in.tcv = tc[AIDX(i)]; in.lcv = lsrc[AIDXL(i)]; in.gold = gro[AIDX(i)];
#pragma unroll
    for (int a = 0; a < NQ; a++) in.jo[a] = jqo[AIDX(fi * NQ + a)];
    const int obase = kind == ROW_LINEAR ? v.off_lin() + 4 * ob : (kind == ROW_RADIAL ? v.off_obst() + 4 * ob : 0);
#pragma unroll
    for (int c = 0; c < 4; c++) in.op[c] = PR(obase + c);
    in.wir = PR(v.has_avoid() ? v.off_wconstr() + mod : 0);
  };
  FkRowIn fk0[NSP];
#pragma unroll
  for (int s = 0; s < NSP; s++) fk_row_load(sd0[s], fk0[s]);   // (a slot without rows: row 0's words, unused)
#pragma unroll
  for (int s = 0; s < NSP; s++) {
    if (sv[s]) {
      const int sl = p + P * s;
      const int fa = sfa[s], fb = sfb[s];
      // point of the slot and its Jacobian (Kin::point with a per-lane frame)
      Vec3 J[NQ];
#pragma unroll
      for (int d = 0; d < NQ; d++) {
        Vec3 col = {0, 0, 0};
        if (v.joint_type(d) == RMPC_JOINT_REVOLUTE) {
          const Vec3 ca = cross(aj[d], pa[s] - oj[d]), cb = cross(aj[d], pb[s] - oj[d]);
          col = {d <= fa ? ca.x : 0.0, d <= fa ? ca.y : 0.0, d <= fa ? ca.z : 0.0};
          const bool ub = fb >= 0 && d <= fb;
          col = {ub ? col.x - cb.x : col.x, ub ? col.y - cb.y : col.y, ub ? col.z - cb.z : col.z};
        } else if (v.joint_type(d) == RMPC_JOINT_PRISMATIC) {
          const double wa = d <= fa ? 1.0 : 0.0, wb = (fb >= 0 && d <= fb) ? 1.0 : 0.0;
          col = {(wa - wb) * aj[d].x, (wa - wb) * aj[d].y, (wa - wb) * aj[d].z};
        }
        J[d] = col;
      }
      const Vec3 Pt = fb >= 0 ? pa[s] - pb[s] : pa[s];
      Vec3 Fc = {0, 0, 0};
      double TQ[6] = {0, 0, 0, 0, 0, 0}, TC[6] = {0, 0, 0, 0, 0, 0};
      double Wsum = 0.0;
      if (sl == 0 && v.has_goal()) {
        // GoalReaching (goal_reaching.py:19-33)
        const double g0 = PR(v.off_goal()), g1 = PR(v.off_goal() + 1), g2 = PR(v.off_goal() + 2);
        const double w0 = PR(v.off_wgoal()), w1 = PR(v.off_wgoal() + 1), w2 = PR(v.off_wgoal() + 2);
        const double e0 = Pt.x - g0, e1 = Pt.y - g1, e2 = Pt.z - g2;
        f += w0 * e0 * e0 + w1 * e1 * e1 + w2 * e2 * e2;
#pragma unroll
        for (int a = 0; a < NQ; a++) qgf[a] += 2.0 * (w0 * e0 * J[a].x + w1 * e1 * J[a].y + w2 * e2 * J[a].z);
        TQ[0] += 2.0 * w0; TQ[3] += 2.0 * w1; TQ[5] += 2.0 * w2;
        Fc = {-2.0 * w0 * e0, -2.0 * w1 * e1, -2.0 * w2 * e2};
      }
      for (int t = 0; t < rows_max; t++) {
        const int r = srb[s] + t;
        if (r < sre[s]) {
          const int d = t == 0 ? sd0[s] : v.fk_desc(r);
          const int i = d & 255, kind = (d >> 8) & 3, fi = (d >> 20) & 63;
          const bool firstr = ((d >> 19) & 1) != 0;
          FkRowIn in = fk0[s];
          if (t > 0) fk_row_load(d, in);   // (uniform: further rows of a slot)
          const double tcv = in.tcv, lcv = in.lcv, gold = in.gold;
          const double op0 = in.op[0], op1 = in.op[1], op2 = in.op[2], op3 = in.op[3], wir = in.wir;
          double jo[NQ];
#pragma unroll
          for (int a = 0; a < NQ; a++) jo[a] = in.jo[a];
          const double wi = (v.has_avoid() && firstr) ? wir : 0.0;
          double gdz = 0.0;
#pragma unroll
          for (int a = 0; a < NQ; a++) gdz += jo[a] * qdz[a];
          // radial / self-collision: distance of a point; linear: distance to a plane
          const bool isl = kind == ROW_LINEAR, isr = kind == ROW_RADIAL;
          const Vec3 dv = {isr ? Pt.x - op0 : Pt.x, isr ? Pt.y - op1 : Pt.y, isr ? Pt.z - op2 : Pt.z};
          const double dist = sqrt(dot(dv, dv));
          const double idist = 1.0 / dist;
          const Vec3 av = {op0, op1, op2};
          const double nrm = sqrt(dot(av, av));
          const double sd = dot(av, Pt) + op3;
          const double sgn = sd < 0 ? -1.0 : 1.0;
          const double inrm = sgn / nrm;
          double h = isl ? fabs(sd) / nrm - rbody : dist - (isr ? op3 + rbody : 2.0 * rbody);
          double cinv = isl ? 0.0 : idist;
          Vec3 nd = {isl ? av.x * inrm : dv.x * idist, isl ? av.y * inrm : dv.y * idist, isl ? av.z * inrm : dv.z * idist};
          // stage 1 (state pinned to xinit): state-only rows are constants of the problem -- neutralised
          if (k == 0) { h = 1.0; cinv = 0.0; nd = {0, 0, 0}; }
          double gq[NQ];
#pragma unroll
          for (int a = 0; a < NQ; a++) gq[a] = dot(nd, J[a]);
          double cw = 0.0, c2row = 0.0;
          {
            const bool on = (wi != 0.0) && (k != 0);
            const double cN = (double)M.N * wi;
            bad |= (int)(on & !(h > 0.0));
            const double ih = frcp(h);
            f += on ? cN * ih : 0.0;
            const double c1 = on ? -cN * (ih * ih) : 0.0;
            c2row = on ? 2.0 * cN * (ih * ih * ih) : 0.0;
            cw = on ? cN * (ih * ih) : 0.0;
#pragma unroll
            for (int a = 0; a < NQ; a++) qgf[a] += c1 * gq[a];
          }
          grn[AIDX(i)] = h;
#pragma unroll
          for (int a = 0; a < NQ; a++) jqn[AIDX(fi * NQ + a)] = gq[a];
          const RowW rw = row_core(i, h, tcv, lcv, gold, gdz);
#pragma unroll
          for (int a = 0; a < NQ; a++) {
            qq0[a] += gq[a] * rw.ca;
            qq1[a] += gq[a] * rw.cb;
            qrs[a] -= gq[a] * rw.lv;
          }
          addsym(TQ, rw.sig + c2row, nd);
          const double wgt = (M.use_curv && !isl) ? (rw.lv + cw) * cinv : 0.0;
          Wsum += wgt;
          addsym(TC, wgt, nd);
          const double wf = rw.lv + cw;
          Fc.x += wf * nd.x; Fc.y += wf * nd.y; Fc.z += wf * nd.z;
        }
      }
      // the slot's terms of the two q blocks: J^T (sum w n n^T) J, and the curvature J^T (W I - sum w n n^T) J plus
      // the second derivatives of the kinematics, (Fc x axis_a) . J_c for revolute joints a before c
#pragma unroll
      for (int a = 0; a < NQ; a++) {
        const Vec3 u = symv(TQ, J[a]);
#pragma unroll
        for (int c = a; c < NQ; c++) bq[qtri(a, c)] += dot(u, J[c]);
      }
      if (M.use_curv) {
#pragma unroll
        for (int a = 0; a < NQ; a++) {
          const Vec3 t = symv(TC, J[a]);
          Vec3 w = {Wsum * J[a].x - t.x, Wsum * J[a].y - t.y, Wsum * J[a].z - t.z};
          if (v.joint_type(a) == RMPC_JOINT_REVOLUTE) {
            const Vec3 G = cross(Fc, aj[a]);
            w = {w.x + G.x, w.y + G.y, w.z + G.z};
          }
#pragma unroll
          for (int c = a; c < NQ; c++) bc[qtri(a, c)] += dot(w, J[c]);
        }
      }
    }
  }
  AP_STAMP(1);
  // ---- row structure of a joint's variables (uniform: rmpc_create admits a model to this kernel only when every joint
  //      carries the same rows -- arm_rows_uniform -- so the descriptors of joint 0 tell which rows exist, whether their
  //      limit is a parameter and their sign, by scalar loads and scalar branches), then the per-lane descriptors of this
  //      part's joints (row index, parameter offset, module; their requests travel during the sums below)
  int vdu[3][kVarRows];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int u = 0; u < kVarRows; u++) vdu[c][u] = v.v_desc(c * NQ, u);
  // ---- (per-lane descriptors) -----------------------
  int vds[NJP][3][kVarRows];
#pragma unroll
  for (int i = 0; i < NJP; i++) {
    const int a = p + P * i < NQ ? p + P * i : p;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int u = 0; u < kVarRows; u++) vds[i][c][u] = v.v_desc(a + c * NQ, u);
  }
  // ---- the parts' terms of a stage added up (every part ends with the totals) ------------------------------------
  if constexpr (P > 1) {
#pragma unroll
    for (int e = 0; e < NQ2; e++) { bq[e] = part_sum<P>(bq[e]); bc[e] = part_sum<P>(bc[e]); }
#pragma unroll
    for (int a = 0; a < NQ; a++) {
      qgf[a] = part_sum<P>(qgf[a]); qq0[a] = part_sum<P>(qq0[a]); qq1[a] = part_sum<P>(qq1[a]); qrs[a] = part_sum<P>(qrs[a]);
    }
  }

  // ---- the q blocks of the record: each part stores its share (the diagonal of the first block is stored again behind
  //      the joints, with the terms of their own rows; the blocks' registers are free for the joints' requests) ----------
  double bqd[NQ];
#pragma unroll
  for (int a = 0; a < NQ; a++) bqd[a] = bq[qtri(a, a)];
  {
    constexpr int NST = (NQ2 + P - 1) / P;
#pragma unroll
    for (int t = 0; t < NST; t++) {
      constexpr int last = NQ2 - 1;
      const int e0 = P * t;
      const double vq = part_pick<P>(p, bq[e0 < NQ2 ? e0 : last], bq[e0 + 1 < NQ2 ? e0 + 1 : last], bq[e0 + 2 < NQ2 ? e0 + 2 : last],
                                     bq[e0 + 3 < NQ2 ? e0 + 3 : last]);
      const double vc = part_pick<P>(p, bc[e0 < NQ2 ? e0 : last], bc[e0 + 1 < NQ2 ? e0 + 1 : last], bc[e0 + 2 < NQ2 ? e0 + 2 : last],
                                     bc[e0 + 3 < NQ2 ? e0 + 3 : last]);
      const int e = e0 + p < NQ2 ? e0 + p : last;   // (a part beyond the end repeats the last entry: same value, same word)
      rec[C::R_Q + e] = vq;
      rec[C::R_C + e] = M.use_curv ? vc : 0.0;
    }
  }
  AP_STAMP(2);
  // ---- this part's joints: the variables q_a, v_a, u_a -------------------------------------------------------------
  double djq[NJP];   // what the single-variable rows of q_a add to the diagonal of the q block
#pragma unroll
  for (int i = 0; i < NJP; i++) djq[i] = 0.0;
  for_range<0, NJP>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const int a = p + P * i;
    if (a < NQ) {
      const int jv[3] = {a, NQ + a, NX + a};
      // every request of the joint first -- its variables, the next stage's state, the costates, and the slack,
      // multiplier, limit and inverse-barrier weight of its 3 x 4 rows (absent rows on the spare row): one round trip
      // instead of one per row
      double zo[3], dzo[3], x1[2], dx1[2], n0[2], n0n[2], n1[2], n1n[2], wn0[2] = {0, 0}, wn1[2] = {0, 0};
#pragma unroll
      for (int c = 0; c < 3; c++) {
        zo[c] = zc[AIDX(jv[c])];
        dzo[c] = dzp[(size_t)jv[c] * SSd + loffd];
      }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        x1[c] = zc[AIDX1(jv[c])]; dx1[c] = dzp[(size_t)jv[c] * SSd + loffd1];
        n0[c] = nc[AIDX(jv[c])]; n0n[c] = nup[(size_t)jv[c] * SSd + loffd];
        n1[c] = nc[AIDX1(jv[c])]; n1n[c] = nup[(size_t)jv[c] * SSd + loffd1];
        if (warm) {
          const unsigned loff2 = loff1 + (k < N - 2 ? io.kstride : 0u);
          wn0[c] = io.wn[AIDX1(jv[c])];
          wn1[c] = io.wn[(unsigned)jv[c] * (unsigned)kFusedStages + loff2];
        }
      }
      double gf[3] = {0, 0, 0}, q0[3] = {0, 0, 0}, q1[3] = {0, 0, 0}, rs[3] = {0, 0, 0}, dg[3] = {0, 0, 0};
      const double wuv = PR(v.off_wu() + a);
      double tcv[3][kVarRows], lcv[3][kVarRows], plim[3][kVarRows], wiv[3][kVarRows];
      const int owc = v.has_avoid() ? v.off_wconstr() : 0;
#pragma unroll
      for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int u = 0; u < kVarRows; u++) {
          const int du = vdu[c][u];   // (uniform: which rows a variable of this kind has, whether their limit is a parameter)
          if (!((du >> 8) & 1)) continue;
          const int d = vds[i][c][u];
          const int irow = d & 255;
          tcv[c][u] = tc[AIDX(irow)];
          lcv[c][u] = lsrc[AIDXL(irow)];
          if ((du >> 11) & 1) plim[c][u] = pp[AIDX((int)((unsigned)d >> 16))];
          else plim[c][u] = v.v_val(jv[c], u);
          wiv[c][u] = v.has_avoid() ? pp[AIDX(owc + ((d >> 12) & 7))] : 0.0;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      double z[3], xk1[2], nuk[2], nun[2];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        z[c] = nostep ? zo[c] : zo[c] + al * dzo[c];
        zn[AIDX(jv[c])] = z[c];
      }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        xk1[c] = nostep ? x1[c] : x1[c] + al * dx1[c];
        double vv = 0.0, ww = 0.0;
        if (!first && k >= 1) vv = nostep ? n0[c] : n0[c] + al * (n0n[c] - n0[c]);
        if (!first && k < N - 1) ww = nostep ? n1[c] : n1[c] + al * (n1n[c] - n1[c]);
        if (warm) {
          // costates of the previous solve, shifted: nu_k <- nu_{k+1}, nu_{k+1} <- nu_{k+2} (last stage repeated)
          if (k >= 1) vv = wn0[c];
          if (k < N - 1) ww = wn1[c];
        }
        nuk[c] = vv;
        nun[c] = ww;
        nn[AIDX(jv[c])] = vv;
      }
      {
        // control effort (ObjectiveManager.py:28-42)
        const double u = z[2];
        f += wuv * u * u;
        gf[2] += 2.0 * wuv * u;
        dg[2] += 2.0 * wuv;
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int u = 0; u < kVarRows; u++) {
          const int du = vdu[c][u];
          if (!((du >> 8) & 1)) continue;   // (scalar branch: the row structure is the same for every joint)
          const bool general = ((du >> 11) & 1) != 0;
          const double sg = ((du >> 9) & 1) ? -1.0 : 1.0;
          const int d = vds[i][c][u];
          const int irow = d & 255;
          const bool firstr = ((d >> 10) & 1) != 0;
          const double lim = plim[c][u];
          const double wi = (v.has_avoid() && firstr) ? wiv[c][u] : 0.0;
          const bool neutral = (k == 0) && (c < 2);   // constant of the problem at the pinned stage
          const double h = neutral ? 1.0 : sg * (z[c] - lim);
          // (inverse-barrier objective on the first row of a module: one joint's row at most -- skipped by a scalar
          //  branch when no lane of the wavefront holds such a row with a non-zero weight)
          if (v.has_avoid() && __ballot(firstr && wi != 0.0) != 0ull) {
            const bool on = (wi != 0.0) && !neutral;
            const double cN = (double)M.N * wi;
            bad |= (int)(on & !(h > 0.0));
            const double ih = frcp(h);
            f += on ? cN * ih : 0.0;
            gf[c] += on ? -cN * (ih * ih) * sg : 0.0;
            dg[c] += on ? 2.0 * cN * (ih * ih * ih) : 0.0;
          }
          if (general) grn[AIDX(irow)] = h;   // general rows keep their value for the step phase
          const double gold = neutral ? 1.0 : sg * (zo[c] - lim);
          const RowW rw = row_core(irow, h, tcv[c][u], lcv[c][u], gold, sg * dzo[c]);
          q0[c] += neutral ? 0.0 : sg * rw.ca;
          q1[c] += neutral ? 0.0 : sg * rw.cb;
          rs[c] -= neutral ? 0.0 : sg * rw.lv;
          dg[c] += neutral ? 0.0 : rw.sig;
        }
      }
      djq[i] = dg[0];
      // the frames' terms of q_a: the totals over the parts, entry a = p + P i
      {
        constexpr int b0 = P * i;
        auto pick = [&](const double (&arr)[NQ]) __attribute__((always_inline)) -> double {
          return part_pick<P>(p, arr[b0 < NQ ? b0 : NQ - 1], arr[b0 + 1 < NQ ? b0 + 1 : NQ - 1], arr[b0 + 2 < NQ ? b0 + 2 : NQ - 1],
                              arr[b0 + 3 < NQ ? b0 + 3 : NQ - 1]);
        };
        gf[0] += pick(qgf); q0[0] += pick(qq0); q1[0] += pick(qq1); rs[0] += pick(qrs);
      }
      // stationarity of the three variables (A^T nu = [nu_q ; dt nu_q + nu_v], B^T nu = dt^2/2 nu_q + dt nu_v)
      {
        double r0 = rs[0] + gf[0], r1 = rs[1] + gf[1], r2 = rs[2] + gf[2];
        if (k < N - 1) {
          r0 += nun[0];
          r1 += hh * nun[0] + nun[1];
          r2 += hh2 * nun[0] + hh * nun[1];
        }
        if (k != 0) {   // x_1 is fixed: no stationarity condition
          rstat = fmax(rstat, fabs(r0 - nuk[0]));
          rstat = fmax(rstat, fabs(r1 - nuk[1]));
        }
        rstat = fmax(rstat, fabs(r2));
      }
      rec[C::R_DG + a] = dg[1];
      rec[C::R_DG + NQ + a] = dg[2];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        rec[C::R_Q0 + jv[c]] = gf[c] + q0[c];
        rec[C::R_Q1 + jv[c]] = q1[c];
        gfa[AIDX(jv[c])] = gf[c];
      }
      // dynamics defect of the joint (ERK2, 5 nodes: chain_step's arithmetic)
      if (k < N - 1) {
        const double hn = M.dt / kErkNodes;
        double xq = z[0], xv = z[1];
#pragma unroll
        for (int it = 0; it < kErkNodes; it++) {
          const double vm = xv + 0.5 * hn * z[2];
          xq += hn * vm;
          xv += hn * z[2];
        }
        const double rq = xq - xk1[0], rv = xv - xk1[1];
        rec[C::R_RC + a] = rq;
        rec[C::R_RC + NQ + a] = rv;
        req = fmax(req, fmax(fabs(rq), fabs(rv)));
        theta += fabs(rq) + fabs(rv);
      } else {
        rec[C::R_RC + a] = 0.0;
        rec[C::R_RC + NQ + a] = 0.0;
      }
    }
  });
  AP_STAMP(3);
  // ---- the diagonal of the q block again, with the terms of the joints' own rows (every part stores the same values) ----
#pragma unroll
  for (int a = 0; a < NQ; a++) {
    const double mine = (p == a % P) ? djq[a / P] : 0.0;
    rec[C::R_Q + qtri(a, a)] = bqd[a] + part_sum<P>(mine);
  }
  rec[C::R_ZERO] = 0.0;
  const double logsum = log(lprod) + 0.6931471805599453094 * (double)lexp;
  bad |= (int)(!isfinite(f) | !isfinite(theta) | !isfinite(logsum));
  out.f = f; out.th = theta; out.logs = logsum; out.rstat = rstat; out.req = req; out.rineq = rineq;
  out.rcomp = rcomp; out.sumc = sumc; out.minc = minc; out.bad = (double)bad;
  AP_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------------------
// step_part: step_body for part p of a stage -- fraction-to-the-boundary ratios of the rows of this part's slots and
// joints, and its share of the merit slope.
// ---------------------------------------------------------------------------------------------------------------------
template <class C, int P, class SP, class V>
__device__ __forceinline__ void step_part(const V &v, const StepIO<SP> &io, const int k, const int p, const double mu,
                                          double &ap_out, double &ad_out, double &gphi_out) {
  constexpr int NQ = C::NQ, NX = C::NX;
  constexpr int NJP = (NQ + P - 1) / P, NSP = (kMaxSlots + P - 1) / P;
  const unsigned loff = io.loff;
  const gdouble *__restrict__ zc = io.zc;
  const gdouble *__restrict__ tc = io.tc;
  const gdouble *__restrict__ lc = io.lc;
  const gdouble *__restrict__ grow = io.grow;
  const gdouble *__restrict__ Jq = io.Jq;
  double gphi = 0.0, ap = 1.0, ad = 1.0;
  auto row = [&](const double gdz, const double g, const double tv, const double lv) __attribute__((always_inline)) {
    const double dt = gdz + (g - tv);
    const double itv = frcp(tv);
    const double dl = (mu - tv * lv - lv * dt) * itv;   // (same expression as in the sweep's row_core)
    const double rp = -C::TAU * tv * frcp(dt), rd = -C::TAU * lv * frcp(dl);
    ap = ((dt < 0) & (rp < ap)) ? rp : ap;
    ad = ((dl < 0) & (rd < ad)) ? rd : ad;
    gphi -= mu * dt * itv;
  };
  // row structure of a joint's variables (uniform, see sweep_part), row descriptors of this part's joints, step of q
  int vdu[3][kVarRows];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int u = 0; u < kVarRows; u++) vdu[c][u] = v.v_desc(c * NQ, u);
  int vds[NJP][3][kVarRows];
#pragma unroll
  for (int i = 0; i < NJP; i++) {
    const int a = p + P * i < NQ ? p + P * i : p;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int u = 0; u < kVarRows; u++) vds[i][c][u] = v.v_desc(a + c * NQ, u);
  }
  double qdz[NQ];
#pragma unroll
  for (int a = 0; a < NQ; a++) qdz[a] = io.dz[(size_t)a * io.SSd + io.loffd];
  // distance rows of this part's slots (requests of a row together)
  const int nsl = v.nslots();
  const int rows_max = v.slot_rows_max();
#pragma unroll
  for (int s = 0; s < NSP; s++) {
    const int sl = p + P * s;
    if (sl < nsl) {
      int rb0 = v.slot_row_begin(0), re0 = v.slot_row_begin(1);
#pragma unroll
      for (int t = 1; t < kMaxSlots; t++) { rb0 = sl == t ? v.slot_row_begin(t) : rb0; re0 = sl == t ? v.slot_row_begin(t + 1) : re0; }
      for (int t = 0; t < rows_max; t++) {
        const int r = rb0 + t;
        if (r < re0) {
          const int d = v.fk_desc(r);
          const int i = d & 255, fi = (d >> 20) & 63;
          const double g = grow[AIDX(i)], tv = tc[AIDX(i)], lv = lc[AIDX(i)];
          double jq[NQ];
#pragma unroll
          for (int a = 0; a < NQ; a++) jq[a] = Jq[AIDX(fi * NQ + a)];
          __builtin_amdgcn_sched_barrier(0);
          double gdz = 0.0;
#pragma unroll
          for (int a = 0; a < NQ; a++) gdz += jq[a] * qdz[a];
          row(gdz, g, tv, lv);
        }
      }
    }
  }
  // rows of this part's joints: every request of a joint first
  for_range<0, NJP>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const int a = p + P * i;
    if (a < NQ) {
      const int jv[3] = {a, NQ + a, NX + a};
      double dzv[3], zv[3], gfv[3], tvv[3][kVarRows], lvv[3][kVarRows], glv[3][kVarRows];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        dzv[c] = io.dz[(size_t)jv[c] * io.SSd + io.loffd];
        zv[c] = zc[AIDX(jv[c])];
        gfv[c] = io.gfa[AIDX(jv[c])];
#pragma unroll
        for (int u = 0; u < kVarRows; u++) {
          const int du = vdu[c][u];
          if (!((du >> 8) & 1)) continue;
          const int irow = vds[i][c][u] & 255;
          tvv[c][u] = tc[AIDX(irow)];
          lvv[c][u] = lc[AIDX(irow)];
          if ((du >> 11) & 1) glv[c][u] = grow[AIDX(irow)];
          else glv[c][u] = v.v_val(jv[c], u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        gphi += gfv[c] * dzv[c];
#pragma unroll
        for (int u = 0; u < kVarRows; u++) {
          const int du = vdu[c][u];
          if (!((du >> 8) & 1)) continue;
          const double sg = ((du >> 9) & 1) ? -1.0 : 1.0;
          const double g = ((du >> 11) & 1) ? glv[c][u] : ((k == 0 && c < 2) ? 1.0 : sg * (zv[c] - glv[c][u]));
          row(sg * dzv[c], g, tvv[c][u], lvv[c][u]);
        }
      }
    }
  });
  ap_out = ap; ad_out = ad; gphi_out = gphi;
}

// ---------------------------------------------------------------------------------------------------------------------
// The phase functions (real functions, as in k_fused: each gets the register file to itself) and the kernel.
// ---------------------------------------------------------------------------------------------------------------------

// LDS of the wavefront (doubles): [ recursion work area | step N x (NV + NX) | hand-over words | gain images ]
template <class C>
struct ArmLds {
  static constexpr int WORK = RicLds<C, 64>::LDSW_ARM;
  static constexpr int KPW = RicLds<C, 64>::KPW;
  static constexpr int SW = C::NV + C::NX;          // step of a stage: dz | nu+
  static constexpr int HAND = 64;                   // solver words parked around the calls, results of the sweep call
  static constexpr int TOTAL = 40960 / 8;           // a quarter of the CU's LDS: one wavefront per SIMD
  __host__ __device__ static constexpr int step_off() { return WORK; }
  __host__ __device__ static constexpr int hand_off(int N) { return WORK + N * SW; }
  __host__ __device__ static constexpr int img_off(int N) { return WORK + N * SW + HAND; }
  __host__ __device__ static constexpr int img_slots(int N) { return (TOTAL - img_off(N)) / KPW; }
};

struct ArmSweepOut { double f, th, lgs, sumc, badf, rstat, req, rineq, rcomp, minc, amin_p, amin_d, gphi; };

// One call per pass: the step lengths of a fresh step (step_part, reduced over the wavefront), then the sweep at the
// trial point, reduced over the wavefront; the instance's words go back through LDS.
template <class C, int P, int FIRSTC>
__device__ __noinline__ void arm_sweep_call(__attribute__((address_space(3))) ArmSweepOut *const out, const ArmBlock *blkp,
                                            const DevTables *tabp, const size_t b, const int cur, ldouble *const lstep,
                                            const bool nostep, const bool fresh, const int ls, const double amin_p_in,
                                            const double amin_d_in, const double gphi_in, const double mu, const int warm) {
  constexpr int NV = C::NV, SW = ArmLds<C>::SW;
  // uniform addresses in the constant address space: the block and the tables come by scalar loads
  // (readfirstlane returns a signed int: through `unsigned` first, or a low half with bit 31 set smears over the high half)
  auto uniform64 = [](const void *ptr) __attribute__((always_inline)) -> unsigned long long {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  typedef const __attribute__((address_space(4))) ArmBlock cArmBlock;
  cArmBlock *const blk = (cArmBlock *)uniform64(blkp);
  GView::cTables *const tab = (GView::cTables *)uniform64(tabp);
  const GView v(&blk->M, tab);
  FusedWs F;
  load_block(F, &blkp->F);
  const int N = blk->M.N;
  const int lane = threadIdx.x;
  const int p = lane % P, k = lane / P;
  const bool live = k < N;
  const int ks = live ? k : N - 1;   // (lanes without a stage read the last stage's step: finite values, nothing stored)
  const size_t S = kFusedStages;
  // (instance and buffer index are the same in every lane: as scalars, every base below is a scalar)
  const size_t bu = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
  const bool c1 = __builtin_amdgcn_readfirstlane(cur) != 0;
  FusedCur Pw;
  {
    gdouble *const z0 = (gdouble *)F.z[0] + bu * F.nv * S, *const z1 = (gdouble *)F.z[1] + bu * F.nv * S;
    gdouble *const t0 = (gdouble *)F.t[0] + bu * F.m * S, *const t1 = (gdouble *)F.t[1] + bu * F.m * S;
    gdouble *const l0 = (gdouble *)F.lam[0] + bu * F.m * S, *const l1 = (gdouble *)F.lam[1] + bu * F.m * S;
    gdouble *const n0 = (gdouble *)F.nu[0] + bu * F.nx * S, *const n1 = (gdouble *)F.nu[1] + bu * F.nx * S;
    gdouble *const g0 = (gdouble *)F.grow[0] + bu * F.nhs * S, *const g1 = (gdouble *)F.grow[1] + bu * F.nhs * S;
    gdouble *const j0 = (gdouble *)F.Jq[0] + bu * F.njqs * S, *const j1 = (gdouble *)F.Jq[1] + bu * F.njqs * S;
    Pw.zc = c1 ? z1 : z0; Pw.zn = c1 ? z0 : z1; Pw.tc = c1 ? t1 : t0; Pw.tn = c1 ? t0 : t1;
    Pw.lc = c1 ? l1 : l0; Pw.ln = c1 ? l0 : l1; Pw.nc = c1 ? n1 : n0; Pw.nn = c1 ? n0 : n1;
    Pw.gc = c1 ? g1 : g0; Pw.gn = c1 ? g0 : g1; Pw.jc = c1 ? j1 : j0; Pw.jn = c1 ? j0 : j1;
    Pw.pp = (gdouble *)F.p + bu * F.npar * S; Pw.pdz = nullptr; Pw.pnn = nullptr;
    Pw.pgf = (gdouble *)F.gfa + bu * F.nv * S;
    Pw.pwl = (gdouble *)F.wlam + bu * F.m * S; Pw.pwn = (gdouble *)F.wnu + bu * F.nx * S;
  }
#ifdef RMPC_STAMPS
  long long ap_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ap_t0 = __builtin_amdgcn_s_memtime();
#endif
  double ap = 1.0, ad = 1.0, gp = 0.0;
  if (fresh) {
    StepIO<ldouble> io;
    io.zc = Pw.zc; io.tc = Pw.tc; io.lc = Pw.lc; io.grow = Pw.gc; io.Jq = Pw.jc;
    io.gfa = Pw.pgf;
    io.SS = S; io.loff = (unsigned)k;
    io.dz = lstep; io.SSd = 1; io.loffd = (unsigned)(ks * SW);
    step_part<C, P, ldouble, GView>(v, io, k, p, mu, ap, ad, gp);
    ap = live ? ap : 1.0; ad = live ? ad : 1.0; gp = live ? gp : 0.0;
  }
  {
    double rs1[1] = {gp}, rm0[1] = {0.0}, rn2[2] = {ap, ad};
    wave_reduce_many<64>(rs1, rm0, rn2);
    gp = rs1[0]; ap = rn2[0]; ad = rn2[1];
  }
  const double amin_p = fresh ? fmin(amin_p_in, ap) : amin_p_in;
  const double amin_d = fresh ? fmin(amin_d_in, ad) : amin_d_in;
  const double gphi = fresh ? gp : gphi_in;
  const double alpha = nostep ? 0.0 : ldexp(amin_p, -ls), adual = nostep ? 0.0 : amin_d;
  Partials q = {0, 0, 0, 0, 0, 0, 0, 0, 1e300, 0};
  AP_STAMP(5);
  {
    SweepIO<gdouble, ldouble> io;
    io.zc = Pw.zc; io.tc = Pw.tc; io.lc = Pw.lc; io.nc = Pw.nc;
    io.zn = Pw.zn; io.tn = Pw.tn; io.ln = Pw.ln; io.nn = Pw.nn;
    io.pp = Pw.pp; io.gro = Pw.gc; io.jqo = Pw.jc; io.grn = Pw.gn; io.jqn = Pw.jn;
    io.gfa = Pw.pgf;
    io.SS = S; io.loff = (unsigned)k; io.kstride = 1u;
    io.rec = (gdouble *)F.R + bu * S * C::RS + (unsigned)k * (unsigned)C::RS;   // (the record array has 32 stage slots per instance)
    io.dzp = lstep; io.nup = lstep + NV;
    io.SSd = 1; io.loffd = (unsigned)(ks * SW); io.kstrided = live && k < N - 1 ? (unsigned)SW : 0u;
    io.wl = Pw.pwl; io.wn = Pw.pwn; io.warm = warm;
    const SweepK sk = {N, blk->M.dt, blk->M.use_curv};
    sweep_part<C, P, gdouble, ldouble, GView, FIRSTC>(sk, v, io, k, p, nostep, alpha, adual, mu, q
#ifdef RMPC_STAMPS
                                                      , ap_acc, ap_t0
#endif
                                                      );
  }
  {
    // (lanes without a stage contribute the neutral elements)
    double rs5[5] = {live ? q.f : 0.0, live ? q.th : 0.0, live ? q.logs : 0.0, live ? q.sumc : 0.0, live ? q.bad : 0.0};
    double rm4[4] = {live ? q.rstat : 0.0, live ? q.req : 0.0, live ? q.rineq : 0.0, live ? q.rcomp : 0.0};
    double rn1[1] = {live ? q.minc : 1e300};
    wave_reduce_many<64>(rs5, rm4, rn1);
    out->f = rs5[0]; out->th = rs5[1]; out->lgs = rs5[2]; out->sumc = rs5[3]; out->badf = rs5[4];
    out->rstat = rm4[0]; out->req = rm4[1]; out->rineq = rm4[2]; out->rcomp = rm4[3]; out->minc = rn1[0];
    out->amin_p = amin_p; out->amin_d = amin_d; out->gphi = gphi;
  }
#ifdef RMPC_STAMPS
  AP_STAMP(6);
  if (lane == 0) {
    for (int i = 0; i < 7; i++) atomicAdd((unsigned long long *)&g_sst[i], (unsigned long long)ap_acc[i]);
    atomicAdd((unsigned long long *)&g_sst[7], 1ull);
  }
#endif
}

template <class C>
__device__ __noinline__ bool arm_recursion_call(const int N, const double dt, const double mu, const bool usec, const int lane,
                                                ldouble *const work, const gdouble *const grec, gdouble *const kpb, const int kps,
                                                ldouble *const lstep, ldouble *const limg, const int lcap) {
  StepOut<ldouble> so;
  so.dz = lstep; so.nunew = lstep + C::NV; so.SS = 1; so.KS = ArmLds<C>::SW;
  return riccati_recursion<C, 64, false, gdouble, false, ldouble>(N, dt, mu, usec ? 1.0 : 0.0, lane, work, grec, kpb, kps, so, nullptr, limg, lcap);
}

template <class C, int P>
__global__ __launch_bounds__(64, 1) __attribute__((amdgpu_waves_per_eu(1, 1), disable_tail_calls))
void k_fused_arm(const DevModel M, const DevTables *__restrict__ Tp, const FusedWs F, const int B,
                 const double *__restrict__ xinit, const double *__restrict__ x0, const double *__restrict__ params,
                 double *__restrict__ zout, int *__restrict__ exitflag, int *__restrict__ iters_out,
                 double *__restrict__ kkt, double *__restrict__ obj, const int max_passes, const int warm_mode,
                 const int use_order) {
  constexpr int NX = C::NX, NV = C::NV, SW = ArmLds<C>::SW;
  const int lane = threadIdx.x;
  const int N = M.N;
  extern __shared__ double lds_dyn[];
  ldouble *const work = (ldouble *)lds_dyn;
  ldouble *const lstep = work + ArmLds<C>::step_off();
  ldouble *const hand = work + ArmLds<C>::hand_off(N);
  ldouble *const limg = work + ArmLds<C>::img_off(N);
  const int lcap = ArmLds<C>::img_slots(N);
  __attribute__((address_space(3))) ArmSweepOut *const sres = (__attribute__((address_space(3))) ArmSweepOut *)hand;
  typedef __attribute__((address_space(3))) unsigned long long lword;
  lword *const sinst = (lword *)(hand + 16);
  constexpr int IW = (int)((sizeof(Inst) + 7) / 8);   // solver words of the instance, as 8-byte words
  static_assert(sizeof(ArmSweepOut) <= 16 * 8 && IW <= ArmLds<C>::HAND - 16, "hand-over area");
  const ArmBlock *const blkp = (const ArmBlock *)(Tp + 1);
  const size_t S = kFusedStages;
  // lanes as (stage, half) for the copies of the prologue and the epilogue
  const int ck = lane & 31, ch = lane >> 5;
  const bool cstage = ck < N;

  union InstWords { Inst s; unsigned long long w[IW]; };
  Inst s;
  const bool warm = warm_mode != 0;
  // (every lane holds the same words: lane 0 parks them around a phase call, all lanes take them back)
  auto park = [&]() __attribute__((always_inline)) {
    InstWords u;
    u.s = s;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < IW; i++) sinst[i] = u.w[i];
    }
  };
  auto unpark = [&]() __attribute__((always_inline)) {
    WSYNC();
    InstWords u;
#pragma unroll
    for (int i = 0; i < IW; i++) u.w[i] = sinst[i];
    s = u.s;
  };
  inst_init(s, M.mu0);
  s.status = 0;
  size_t b = (size_t)(B - 1);
  bool valid = false, retired = false, first = true;
  int ipass = 0;
  int nextslot = blockIdx.x;
  int *const qhead = F.passes + 1;
  double gphi_sum = 0.0;
#ifdef RMPC_STAMPS
  long long st_sweep = 0, st_dec = 0, st_ric = 0, st_pro = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_a = st_t0, st_b;
  int st_pass = 0;
#endif
  for (;;) {
    // ---- a finished instance leaves, the next one of the queue comes in -----------------------------------------
    {
      const bool over = valid && (s.status == ST_ACTIVE) && ipass >= max_passes;
      const bool done = valid && (s.status != ST_ACTIVE || over);
      if (done || (!valid && !retired)) {   // (uniform: every lane holds the same words)
        if (done) {
          const FusedPtrs Pe = fused_ptrs(F, b);
          const bool okd = (s.status == ST_ACTIVE || s.status >= 0) && isfinite(s.mu) && s.mu > 0.0;
          if (cstage) {
            const gdouble *zf = Pe.pz[s.cur];
            double *zr = zout + (b * N + ck) * NV;
            for (int j = ch; j < NV; j += 2) zr[j] = zf[j * S + ck];
            const gdouble *lf = Pe.pl[s.cur], *nf = Pe.pn[s.cur];
            {
              int i = ch;
              for (; i + 14 < F.m; i += 16) {   // (eight requests in flight)
                double lv8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) lv8[u] = lf[(i + 2 * u) * S + ck];
#pragma unroll
                for (int u = 0; u < 8; u++) Pe.pwl[(i + 2 * u) * S + ck] = okd ? lv8[u] : 0.0;
              }
              for (; i < F.m; i += 2) Pe.pwl[i * S + ck] = okd ? lf[i * S + ck] : 0.0;
            }
            for (int j = ch; j < NX; j += 2) Pe.pwn[j * S + ck] = okd ? nf[j * S + ck] : 0.0;
          }
          if (lane == 0) {
            exitflag[b] = (s.status == ST_ACTIVE) ? 0 : s.status;
            iters_out[b] = s.iters;
            kkt[b] = fmax(fmax(s.res_stat, s.res_eq), fmax(s.res_ineq, s.res_comp));
            obj[b] = s.obj;
            F.wmu[b] = okd ? s.mu : M.mu0;
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
            if (lane == 0) t = atomicAdd(qhead, 1);
            pos = (int)gridDim.x + __shfl(t, 0, 64);
          }
          if (pos < B) {
            b = (size_t)(use_order ? F.order[pos] : pos);
            valid = true;
            const FusedPtrs P0 = fused_ptrs(F, b);
            if (cstage) {
              const double *zr = x0 + (b * N + ck) * NV;
              for (int j = ch; j < NV; j += 2) {
                double vz = zr[j];
                if (ck == 0 && j < NX) vz = xinit[b * NX + j];
                P0.pz[0][j * S + ck] = vz;
              }
              if (params) {
                const double *pr = params + (b * N + ck) * M.npar;
                int j = ch;
                for (; j + 14 < M.npar; j += 16) {
                  double pv8[8];
#pragma unroll
                  for (int u = 0; u < 8; u++) pv8[u] = pr[j + 2 * u];
#pragma unroll
                  for (int u = 0; u < 8; u++) P0.pp[(j + 2 * u) * S + ck] = pv8[u];
                }
                for (; j < M.npar; j += 2) P0.pp[j * S + ck] = pr[j];
              }
            }
            // the step is read (and discarded) by the first sweep: keep it finite
            for (int e = lane; e < N * SW; e += 64) lstep[e] = 0.0;
            inst_init(s, warm ? warm_mu(F.wmu[b], M.mu0) : M.mu0);
            first = true;
            ipass = 0;
            gphi_sum = 0.0;
          } else {
            retired = true;
            b = (size_t)(B - 1);
          }
        }
        GSYNC();
      }
    }
    const bool act = valid && (s.status == ST_ACTIVE);
    if (!act) break;   // (uniform) the queue is empty
    ipass++;
#ifdef RMPC_STAMPS
    st_pass++;
    STAMP_B(st_pro);
#endif
    // ---- step lengths of a fresh step, sweep at the trial point --------------------------------------------------------
    const bool nostep = first || (s.redo != 0);
    const bool fresh = !nostep && (s.newstep != 0);
    park();
    if (first) arm_sweep_call<C, P, 1>(sres, blkp, Tp, b, s.cur, lstep, nostep, fresh, s.ls, s.amin_p, s.amin_d, gphi_sum, s.mu, warm ? 1 : 0);
    else arm_sweep_call<C, P, 0>(sres, blkp, Tp, b, s.cur, lstep, nostep, fresh, s.ls, s.amin_p, s.amin_d, gphi_sum, s.mu, warm ? 1 : 0);
    unpark();
    Reduced r;
    {
      if (fresh) { s.amin_p = sres->amin_p; s.amin_d = sres->amin_d; gphi_sum = sres->gphi; }
      r.f = sres->f; r.th = sres->th; r.lgs = sres->lgs; r.sumc = sres->sumc; r.badf = sres->badf;
      r.rstat = sres->rstat; r.req = sres->req; r.rineq = sres->rineq; r.rcomp = sres->rcomp; r.minc = sres->minc;
    }
    r.gphi = first ? 0.0 : gphi_sum;
    GSYNC();   // trial point and records are complete before any lane reads another lane's part
#ifdef RMPC_STAMPS
    STAMP_B(st_sweep);
#endif
    // ---- decisions, then a new step when the trial was accepted ----------------------------------------------------------
    bool usec = false;
    const bool recurse = inst_decide<C>(M, s, r, first, usec);
    first = false;
#ifdef RMPC_STAMPS
    STAMP_B(st_dec);
#endif
    park();
    bool rec_ok = true;
    if (recurse)
      rec_ok = arm_recursion_call<C>(M.N, M.dt, s.mu, usec, lane, work, (gdouble *)F.R + b * S * C::RS,
                                     (gdouble *)F.KP + b * (size_t)N * F.kps, F.kps, lstep, limg, lcap);
    unpark();
    if (recurse) inst_after_recursion(s, rec_ok, usec);
    GSYNC();   // dz, nu+
#ifdef RMPC_STAMPS
    STAMP_B(st_ric);
#endif
  }
#ifdef RMPC_STAMPS
  if (threadIdx.x == 0) {
    long long *o = F.stamps + (size_t)blockIdx.x * 8;
    o[0] = st_sweep; o[1] = st_dec; o[2] = st_ric; o[3] = st_pro; o[4] = __builtin_amdgcn_s_memtime() - st_t0; o[5] = st_pass;
    o[6] = st_t0; o[7] = st_pass;
  }
#endif
}

