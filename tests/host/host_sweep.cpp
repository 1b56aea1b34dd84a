// Host harness for the sanitizers: the single-lane device functions of csrc/rmpc_kernels.hip -- sweep_body and
// step_body, one call per (instance, stage) -- compiled for x86 (tests/host/host_prelude.h) and run over the pass kernels'
// workspace layout with every array allocated on its own, so that AddressSanitizer sees an access behind any of them and
// UBSan any undefined operation in the row, kinematics and dynamics code.  Test infrastructure only: nothing in the
// product includes or links this file.  Build and run: tests/host/run_asan.sh.
#include "host_prelude.h"
#include "../../robot_mpcs_amd/csrc/rmpc_kernels.hip"

#include <vector>

namespace {
using namespace rmpc;

template <class C>
int run_sweeps(const rmpc_desc &d, const DevModel &M, const DevTables &T, int B, const double *xinit, const double *x0,
               const double *params, double *part_out, double *step_out) {
  const int N = M.N, Bp = (B + 63) / 64 * 64;
  const size_t S = (size_t)N * Bp;
  auto arr = [&](size_t slots) { return std::vector<double>(S * (slots > 0 ? slots : 1), 0.0); };
  std::vector<double> p = arr(M.npar), z[2] = {arr(M.nv), arr(M.nv)}, t[2] = {arr(M.m), arr(M.m)}, lam[2] = {arr(M.m), arr(M.m)},
                      nu[2] = {arr(M.nx), arr(M.nx)}, dz = arr(M.nv), nunew = arr(M.nx), gfa = arr(M.nv),
                      grow[2] = {arr(M.nh), arr(M.nh)}, Jq[2] = {arr(M.nfk * M.n), arr(M.nfk * M.n)}, wl = arr(M.m), wn = arr(M.nx);
  std::vector<double> R((size_t)B * N * C::RS, 0.0);
  std::vector<double> lds((size_t)(2 * C::NQ2 + C::NX + C::NQ + 1) * kSweepBlock, 0.0);
  // k_pack's layout: [slot][stage][instance]; x_1 := xinit (mpcModel.py:108)
  for (int b = 0; b < B; b++)
    for (int k = 0; k < N; k++) {
      for (int j = 0; j < M.nv; j++) {
        double v = x0[((size_t)b * N + k) * M.nv + j];
        if (k == 0 && j < M.nx) v = xinit[(size_t)b * M.nx + j];
        z[0][(size_t)j * S + (size_t)k * Bp + b] = v;
      }
      for (int j = 0; j < M.npar; j++) p[(size_t)j * S + (size_t)k * Bp + b] = params[((size_t)b * N + k) * M.npar + j];
    }
  const RtView v(M, T);
  const SweepK sk = {M.N, M.dt, M.use_curv};
  const double mu = M.mu0;
  for (int pass = 0; pass < 3; pass++) {
    const int cur = pass == 0 ? 0 : 1 - (pass & 1) ? 1 : 0;   // pass 0: buffer 0 -> 1; pass 1: 1 -> 0; pass 2: 0 -> 1
    const int c = pass == 0 ? 0 : (pass == 1 ? 1 : 0), nx2 = c ^ 1;
    (void)cur;
    for (int b = 0; b < B; b++)
      for (int k = 0; k < N; k++) {
        SweepIO<gdouble> io;
        io.zc = (gdouble *)z[c].data(); io.tc = (gdouble *)t[c].data(); io.lc = (gdouble *)lam[c].data(); io.nc = (gdouble *)nu[c].data();
        io.zn = (gdouble *)z[nx2].data(); io.tn = (gdouble *)t[nx2].data(); io.ln = (gdouble *)lam[nx2].data(); io.nn = (gdouble *)nu[nx2].data();
        io.pp = (gdouble *)p.data(); io.dzp = (gdouble *)dz.data(); io.gro = (gdouble *)grow[c].data(); io.jqo = (gdouble *)Jq[c].data();
        io.grn = (gdouble *)grow[nx2].data(); io.jqn = (gdouble *)Jq[nx2].data();
        io.nup = (gdouble *)nunew.data(); io.gfa = (gdouble *)gfa.data();
        io.rec = (gdouble *)(R.data() + ((size_t)b * N + k) * C::RS);
        io.loff = (unsigned)k * (unsigned)Bp + (unsigned)b; io.kstride = (unsigned)Bp; io.SS = S;
        io.SSd = io.SS; io.loffd = io.loff; io.kstrided = io.kstride;
        io.wl = (gdouble *)wl.data(); io.wn = (gdouble *)wn.data(); io.warm = 0;
        Partials pt;
        ldouble *const qacc = (ldouble *)lds.data();
        // pass 0: first sweep (slacks and multipliers are initialised); pass 1: null pass at the same point; pass 2: a trial
        // with step length 0.5 along a zero step (the arithmetic of a real trial at an unchanged point)
        if (pass == 0) sweep_body<C, -1, gdouble, RtView, 1>(sk, v, io, k, true, true, 0.0, 0.0, mu, pt, qacc);
        else if (pass == 1) sweep_body<C, -1, gdouble, RtView, 0>(sk, v, io, k, false, true, 0.0, 0.0, mu, pt, qacc);
        else sweep_body<C, -1, gdouble, RtView, 0>(sk, v, io, k, false, false, 0.5, 0.5, mu, pt, qacc);
        double *o = part_out + (((size_t)pass * B + b) * N + k) * 10;
        o[0] = pt.f; o[1] = pt.th; o[2] = pt.logs; o[3] = pt.rstat; o[4] = pt.req; o[5] = pt.rineq; o[6] = pt.rcomp;
        o[7] = pt.sumc; o[8] = pt.minc; o[9] = pt.bad;
      }
    if (pass == 1) nunew = nu[0];   // (the zero step of pass 2: nu+ = nu)
  }
  // the step phase on the current point (buffer 1 after pass 2) with the zero step: fraction-to-the-boundary ratios and merit slope
  for (int b = 0; b < B; b++)
    for (int k = 0; k < N; k++) {
      StepIO<gdouble> io;
      io.zc = (gdouble *)z[1].data(); io.tc = (gdouble *)t[1].data(); io.lc = (gdouble *)lam[1].data();
      io.grow = (gdouble *)grow[1].data(); io.Jq = (gdouble *)Jq[1].data(); io.gfa = (gdouble *)gfa.data();
      io.SS = S; io.loff = (unsigned)k * (unsigned)Bp + (unsigned)b;
      io.dz = (gdouble *)dz.data(); io.SSd = S; io.loffd = io.loff;
      double ap = 1.0, ad = 1.0, gp = 0.0;
      step_body<C, gdouble, RtView>(v, io, k, mu, ap, ad, gp);
      double *o = step_out + ((size_t)b * N + k) * 3;
      o[0] = ap; o[1] = ad; o[2] = gp;
    }
  (void)d;
  return 0;
}
}  // namespace

extern "C" int host_sweeps(const rmpc_desc *desc, int B, const double *xinit, const double *x0, const double *params,
                           double *part_out, double *step_out) {
  using namespace rmpc;
  DevModel M;
  DevTables T;
  std::string err;
  if (build_model(*desc, M, err) != 0 || build_tables(*desc, M, T, err) != 0) return -1;
  if (desc->robot == RMPC_ROBOT_CHAIN && desc->n == 3 && desc->ns == 0) return run_sweeps<Cfg<RMPC_ROBOT_CHAIN, 3, 0>>(*desc, M, T, B, xinit, x0, params, part_out, step_out);
  if (desc->robot == RMPC_ROBOT_CHAIN && desc->n == 7 && desc->ns == 0) return run_sweeps<Cfg<RMPC_ROBOT_CHAIN, 7, 0>>(*desc, M, T, B, xinit, x0, params, part_out, step_out);
  if (desc->robot == RMPC_ROBOT_DIFFDRIVE && desc->ns == 1) return run_sweeps<Cfg<RMPC_ROBOT_DIFFDRIVE, 3, 1>>(*desc, M, T, B, xinit, x0, params, part_out, step_out);
  return -2;
}

// The host pass of a HIP file registers its device code with the HIP runtime when the library is loaded; there is no
// device code here (--cuda-host-only) and no GPU is touched: the registration entry points are no-ops of this library
// (bound locally: -Wl,-Bsymbolic), and the fat-binary symbol they would be handed is defined by run_asan.sh.
extern "C" {
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void *, void *, const char *, size_t, unsigned) {}
}
