// Host build of the device functions for the sanitizers (tests/host/README in DESIGN.md 6): every __device__ function
// of csrc/rmpc_kernels.hip becomes __host__ __device__ and the gfx950 builtins get scalar stand-ins, so that the
// single-lane functions (sweep_body, step_body: one lane = one stage) can be called from a host harness under
// AddressSanitizer / UBSan.  Compiled with --cuda-host-only: no device pass sees these macros.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#undef __device__
#define __device__ __attribute__((device)) __attribute__((host))
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_memtime() (0ll)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) (src)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_ds_bpermute(a, v) (v)
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) (c)
// (HIP's cmath wrapper declares isfinite / isnan for the device only; the host versions by name)
// (declared in std as well: the host code of the file spells std::isfinite)
namespace std {
__host__ __attribute__((device)) inline bool isfinite_host(double x) { return __builtin_isfinite(x); }
__host__ __attribute__((device)) inline bool isnan_host(double x) { return __builtin_isnan(x); }
}  // namespace std
using std::isfinite_host;
using std::isnan_host;
#define isfinite(x) isfinite_host(x)
#define isnan(x) isnan_host(x)
