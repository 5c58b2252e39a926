// C-ABI layer (include/pnode.h): handle management, host-side slot bookkeeping, launches.
// No torch types, no Python: plain pointers and sizes. One handle = one GPU + one HIP stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/pnode.h"
#include "pnode_device.hpp"

using namespace pnode;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(PNODE_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));         \
  } while (0)

template <class T>
int dalloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) return PNODE_OK;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  HIP_TRY(hipMemset(*p, 0, count * sizeof(T)));
  return PNODE_OK;
}

void fill_iwp(int nu, IwpConst* K) {
  const int n = nu + 1;
  std::memset(K, 0, sizeof(*K));
  auto binom = [](int a, int b) {
    double r = 1.0;
    for (int i = 1; i <= b; ++i) r = r * double(a - b + i) / double(i);
    return r;
  };
  double q[kMaxN][kMaxN], l[kMaxN][kMaxN];
  for (int i = 0; i < n; ++i) {
    double f = 1.0;
    for (int m = 2; m <= nu - i; ++m) f *= double(m);
    K->fact[i] = f;
    for (int j = 0; j < n; ++j) {
      K->a[i][j] = (j >= i) ? std::round(binom(nu - i, nu - j)) : 0.0;
      q[i][j] = 1.0 / double(2 * nu + 1 - i - j);
      l[i][j] = 0.0;
    }
  }
  // lower Cholesky of the (flipped) Hilbert matrix
  for (int j = 0; j < n; ++j) {
    double s = q[j][j];
    for (int k = 0; k < j; ++k) s -= l[j][k] * l[j][k];
    l[j][j] = std::sqrt(s);
    for (int i = j + 1; i < n; ++i) {
      double t = q[i][j];
      for (int k = 0; k < j; ++k) t -= l[i][k] * l[j][k];
      l[i][j] = t / l[j][j];
    }
  }
  for (int i = 0; i < n; ++i) {
    K->qdiag_sqrt[i] = std::sqrt(q[i][i]);
    for (int j = 0; j < n; ++j) K->qsT[j][i] = l[i][j];
  }
}

// m^-[i][j] = (P Abar P^-1 m)[i][j] for i < k: what a host-evaluated vector field needs before an attempt.
__global__ void user_extrapolate_kernel(const double* __restrict__ mean, const StepArg* __restrict__ args,
                                        const IwpConst* __restrict__ Kc, double* __restrict__ y, int B, int n, int d, int k) {
  const int b = blockIdx.x;
  const StepArg a = args[b];
  if (!a.active) return;
  const IwpConst& K = *Kc;
  const double adt = fabs(a.dt);
  double p[kMaxN], pinv[kMaxN];
  double pw = sqrt(adt);
  for (int i = n - 1; i >= 0; --i) { p[i] = pw / K.fact[i]; pinv[i] = K.fact[i] / pw; pw *= adt; }
  const double* m = mean + (size_t(a.src) * B + b) * n * d;
  for (int j = threadIdx.x; j < d; j += blockDim.x)
    for (int i = 0; i < k; ++i) {
      double acc = 0.0;
      for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * pinv[kk] * m[kk * d + j];
      y[(size_t(b) * k + i) * d + j] = p[i] * acc;
    }
}

__global__ void gather_u_kernel(const double* __restrict__ mean, const StepArg* __restrict__ args, double* __restrict__ out,
                                int B, int n, int d, int G, int g) {
  const int b = blockIdx.x;
  const double* m = mean + (size_t(args[b].dst) * B + b) * n * d;
  for (int j = threadIdx.x; j < d; j += blockDim.x) out[(size_t(b) * G + g) * d + j] = m[j];
}

// Device-resident controller, multi-kernel form (a dense attempt is several launches): runs between two attempts, one thread per
// trajectory. Takes the error norm of the attempt just made (if any), accepts / rejects, applies the PI law, rotates the state
// slots and writes the arguments of the next attempt to device memory -- same arithmetic and order as the host loop of
// pnode_solve_adaptive_save_at (control_proportional_integral + adaptive, /root/reference/src/odecheckpts/ivpsolvers.py:52-53).
__global__ void dense_ctl_kernel(TrajCtl* __restrict__ ctl, StepArg* __restrict__ args, const double* __restrict__ err,
                                 ControlParams cp, int B, int first) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  TrajCtl c = ctl[b];
  StepArg a = args[b];
  const bool was_active = a.active != 0;
  if (c.status != 0 && !was_active) return;
  controller_decision(c, a, first ? 0.0 : err[b], cp, first != 0);
  args[b] = a;
  ctl[b] = c;
}

// Dense fixed-point smoother under the device-resident controller: the backward-transition kernels that follow the step take
// their arguments (FpArg) from the StepArg the controller kernel has just written.
__global__ void fp_args_from_step_args_kernel(const StepArg* __restrict__ args, FpArg* __restrict__ fp, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const StepArg a = args[b];
  fp[b] = FpArg{a.dt, a.src, a.dst, 1, a.active, 0, 0};
}

}  // namespace

struct pnode_batch {
  pnode_config cfg{};
  int n = 0, D = 0;
  bool dense = false;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  PhaseEvents phase{};
  bool phase_timing = false;
  std::vector<PhaseEvents> phase_pool;  // one set per round of pnode_advance_device, so that every attempt's phases are timed
  double phase_sum[4] = {0.0, 0.0, 0.0, 0.0};
  int64_t phase_count = 0;
  int phase_last = -1;  // events of the most recent attempt: index into phase_pool, -1 = the handle's own set (an index: the pool may grow)
  IsoView iso{};
  DenseView dn{};
  IwpConst* dK = nullptr;
  TrajCtl* dCtl = nullptr;      // device-resident controller state (multi-kernel form)
  double* dPlanSave = nullptr;  // isotropic model, checkpoints inside the controller kernel (CkptPlan): save_at [B][K + 1],
  int32_t* dPlanCur = nullptr;  //   cursor [B],
  int64_t* dPlanNum = nullptr;  //   accepted steps at the checkpoints [B][K + 1]
  FpArg* dFp = nullptr;         // arguments of the backward-transition kernels in that form (dense fixed-point smoother)
  StepArg* dArgs = nullptr;
  InterpArg* dIargs = nullptr;
  int32_t* dFilled = nullptr;
  double *dAtol = nullptr, *dRtol = nullptr, *dPrm = nullptr, *dErr = nullptr;
  double *dUserF = nullptr, *dUserJ = nullptr, *dUserY = nullptr;  // PNODE_VF_USER exchange buffers
  FpArg *hFp0 = nullptr, *hFp1 = nullptr, *zFp0 = nullptr, *zFp1 = nullptr;  // pinned + device views (dense fixedpoint)
  InterpArg* zIargs = nullptr;
  StepArg* zArgs = nullptr;     // device view of hArgs (zero-copy: kernels read the per-step arguments over PCIe)
  double* zErr = nullptr;       // device view of hErr  (kernels write the error norms straight to pinned host memory)
  bool step_timing = false;
  StepArg* hArgs = nullptr;     // pinned
  InterpArg* hIargs = nullptr;  // pinned
  double* hErr = nullptr;       // pinned
  int32_t* hFault = nullptr;    // pinned [4]: hand-shake failure of the one-launch LQ (pnode_dense_lq4.hip)
  // host bookkeeping, per trajectory
  std::vector<double> t_sf, t_if, dt_pending;
  std::vector<int8_t> s_sf, s_if, s_pr;
  std::vector<int64_t> steps;
  std::vector<uint8_t> pending;
  bool initialised = false;
  bool stepped = false;  // a step has run since set_initial: the layout of the state factor is fixed by the path that wrote it
  std::vector<void*> allocations;
};

extern "C" {

const char* pnode_last_error(void) { return g_err.c_str(); }
int pnode_version(void) { return 100; }

int pnode_device_count(void) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) return fail(PNODE_EDEVICE, hipGetErrorString(e));
  return c;
}

int pnode_destroy(pnode_batch* h) {
  if (!h) return PNODE_OK;
  (void)hipSetDevice(h->cfg.device);
  for (void* p : h->allocations) (void)hipFree(p);
  if (h->hArgs) (void)hipHostFree(h->hArgs);
  if (h->hIargs) (void)hipHostFree(h->hIargs);
  if (h->hErr) (void)hipHostFree(h->hErr);
  if (h->hFault) (void)hipHostFree(h->hFault);
  if (h->hFp0) (void)hipHostFree(h->hFp0);
  if (h->hFp1) (void)hipHostFree(h->hFp1);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (auto& e : h->phase.ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& pe : h->phase_pool)
    for (auto& e : pe.ev)
      if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PNODE_OK;
}

#define ALLOC(ptr, count)                                 \
  do {                                                    \
    int rc_ = dalloc(&(ptr), (count));                    \
    if (rc_ != PNODE_OK) {                                \
      pnode_destroy(h);                                   \
      return rc_;                                         \
    }                                                     \
    if (ptr) h->allocations.push_back((void*)(ptr));      \
  } while (0)

// After `new pnode_batch`: every failure releases the handle (stream, events, pinned and device memory) before returning.
#define HIP_TRY_H(expr)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      pnode_destroy(h);                                                                         \
      return fail(PNODE_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    }                                                                                           \
  } while (0)

int pnode_create(const pnode_config* cfg, pnode_batch** out) {
  if (!cfg || !out) return fail(PNODE_EINVAL, "null argument");
  *out = nullptr;
  if (cfg->batch <= 0 || cfg->d <= 0) return fail(PNODE_EINVAL, "batch and d must be positive");
  if (cfg->nu < 1 || cfg->nu > 8) return fail(PNODE_EINVAL, "nu must be in 1..8");
  if (cfg->ode_order < 1 || cfg->ode_order > 2 || cfg->ode_order > cfg->nu)
    return fail(PNODE_EINVAL, "ode_order must be 1 or 2 and <= nu");
  if (cfg->model != PNODE_MODEL_ISOTROPIC && cfg->model != PNODE_MODEL_DENSE) return fail(PNODE_EINVAL, "bad model");
  if (cfg->correction != PNODE_CORR_TS0 && cfg->correction != PNODE_CORR_TS1) return fail(PNODE_EINVAL, "bad correction");
  if (cfg->model == PNODE_MODEL_ISOTROPIC && cfg->correction == PNODE_CORR_TS1)
    return fail(PNODE_EINVAL, "correction_ts1 needs the dense model");
  if (cfg->strategy != PNODE_STRATEGY_FILTER && cfg->strategy != PNODE_STRATEGY_FIXEDPOINT)
    return fail(PNODE_EINVAL, "bad strategy");
  if (cfg->calibration != PNODE_CALIB_NONE && cfg->calibration != PNODE_CALIB_DYNAMIC)
    return fail(PNODE_EINVAL, "bad calibration");
  if (cfg->num_vf_params < 0 || cfg->num_vf_params > PNODE_MAX_VF_PARAMS) return fail(PNODE_EINVAL, "bad num_vf_params");
  if (cfg->num_checkpoints < 0) return fail(PNODE_EINVAL, "bad num_checkpoints");
  // vector-field shape checks
  const int d = cfg->d, k = cfg->ode_order;
  auto need = [&](bool ok, const char* what) { return ok ? PNODE_OK : fail(PNODE_EINVAL, what); };
  int rc = PNODE_OK;
  switch (cfg->vf_id) {
    case PNODE_VF_BRUSSELATOR: rc = need(d % 2 == 0 && k == 1 && cfg->num_vf_params <= 1, "brusselator: d even, order 1"); break;
    case PNODE_VF_VDP_ORDER2: rc = need(d == 1 && k == 2 && cfg->num_vf_params == 1, "vdp (order 2): d=1, params [mu]"); break;
    case PNODE_VF_VDP_ORDER1: rc = need(d == 2 && k == 1 && cfg->num_vf_params == 1, "vdp (order 1): d=2, params [mu]"); break;
    case PNODE_VF_RIGID_BODY: rc = need(d == 3 && k == 1 && cfg->num_vf_params == 3, "rigid body: d=3, params [a,b,c]"); break;
    case PNODE_VF_PLEIADES_ORDER2: rc = need(d == 14 && k == 2, "pleiades (order 2): d=14"); break;
    case PNODE_VF_PLEIADES_ORDER1: rc = need(d == 28 && k == 1, "pleiades (order 1): d=28"); break;
    case PNODE_VF_LOGISTIC: rc = need(k == 1 && cfg->num_vf_params == 2, "logistic: order 1, params [a,b]"); break;
    case PNODE_VF_LINEAR: rc = need(k == 1 && cfg->num_vf_params == 1, "linear: order 1, params [lam]"); break;
    case PNODE_VF_THREE_BODY: rc = need(d == 2 && k == 2 && cfg->num_vf_params == 1, "restricted three-body problem: d=2, order 2, params [mu]"); break;
    case PNODE_VF_USER: rc = need(cfg->num_vf_params == 0, "user vector field: no device parameters"); break;
    default: rc = fail(PNODE_EINVAL, "unknown vf_id");
  }
  if (rc != PNODE_OK) return rc;
  if (cfg->num_vf_params > 0 && !cfg->vf_params) return fail(PNODE_EINVAL, "vf_params is null");

  HIP_TRY(hipSetDevice(cfg->device));
  pnode_batch* h = new (std::nothrow) pnode_batch();
  if (!h) return fail(PNODE_EDEVICE, "host allocation failed");
  h->cfg = *cfg;
  h->cfg.vf_params = nullptr;
  h->n = cfg->nu + 1;
  h->D = h->n * d;
  h->dense = cfg->model == PNODE_MODEL_DENSE;
  const size_t B = cfg->batch, n = h->n, D = h->D, Kc = cfg->num_checkpoints;
  HIP_TRY_H(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIP_TRY_H(hipEventCreate(&h->ev0));
  HIP_TRY_H(hipEventCreate(&h->ev1));
  for (auto& e : h->phase.ev) HIP_TRY_H(hipEventCreate(&e));
  HIP_TRY_H(hipHostMalloc(reinterpret_cast<void**>(&h->hArgs), B * sizeof(StepArg)));
  HIP_TRY_H(hipHostMalloc(reinterpret_cast<void**>(&h->hIargs), B * sizeof(InterpArg)));
  HIP_TRY_H(hipHostMalloc(reinterpret_cast<void**>(&h->hErr), B * sizeof(double)));
  HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zArgs), h->hArgs, 0));
  HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zErr), h->hErr, 0));
  HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zIargs), h->hIargs, 0));
  HIP_TRY_H(hipHostMalloc(reinterpret_cast<void**>(&h->hFault), 4 * sizeof(int32_t)));
  std::memset(h->hFault, 0, 4 * sizeof(int32_t));
  HIP_TRY_H(hipHostMalloc(reinterpret_cast<void**>(&h->hFp0), B * sizeof(FpArg)));
  HIP_TRY_H(hipHostMalloc(reinterpret_cast<void**>(&h->hFp1), B * sizeof(FpArg)));
  HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zFp0), h->hFp0, 0));
  HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zFp1), h->hFp1, 0));

  ALLOC(h->dK, 1);
  ALLOC(h->dArgs, B);
  ALLOC(h->dCtl, B);
  if (cfg->model == PNODE_MODEL_ISOTROPIC) {
    ALLOC(h->dPlanSave, B * (size_t(cfg->num_checkpoints) + 1));
    ALLOC(h->dPlanCur, B);
    ALLOC(h->dPlanNum, B * (size_t(cfg->num_checkpoints) + 1));
  }
  ALLOC(h->dFp, B);
  ALLOC(h->dIargs, B);
  ALLOC(h->dFilled, B);
  ALLOC(h->dAtol, B);
  ALLOC(h->dRtol, B);
  ALLOC(h->dErr, B);
  const size_t nprm = size_t(cfg->num_vf_params) * (cfg->vf_params_per_trajectory ? B : 1);
  ALLOC(h->dPrm, nprm);
  if (nprm) HIP_TRY_H(hipMemcpy(h->dPrm, cfg->vf_params, nprm * sizeof(double), hipMemcpyHostToDevice));
  if (cfg->vf_id == PNODE_VF_USER) {
    ALLOC(h->dUserF, B * size_t(d));
    ALLOC(h->dUserJ, B * size_t(d) * k * d);
    ALLOC(h->dUserY, B * size_t(k) * d);
  }
  IwpConst hk;
  fill_iwp(cfg->nu, &hk);
  HIP_TRY_H(hipMemcpy(h->dK, &hk, sizeof(hk), hipMemcpyHostToDevice));

  if (!h->dense) {
    IsoView& v = h->iso;
    v.B = int(B); v.d = d; v.k = k; v.n = int(n); v.K = int(Kc);
    {
      int ncu = 0;
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || ncu <= 0) ncu = 256;
      v.cus = ncu;
      v.fp_pipeline = 0;
      v.fp_spin_limit = 0;
      HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void**>(&v.fault), h->hFault, 0));
    }
    v.fixedpoint = cfg->strategy == PNODE_STRATEGY_FIXEDPOINT;
    v.dynamic = cfg->calibration == PNODE_CALIB_DYNAMIC;
    v.vf_id = cfg->vf_id; v.P = cfg->num_vf_params; v.prm_per_traj = cfg->vf_params_per_trajectory;
    ALLOC(v.mean, 3 * B * n * d);
    ALLOC(v.fac, 3 * B * n * n);
    ALLOC(v.sigma, 3 * B);
    ALLOC(v.ck_mean, B * Kc * n * d);
    ALLOC(v.ck_fac, B * Kc * n * n);
    ALLOC(v.ck_steps, B * Kc);
    ALLOC(v.ck_sigma, B * Kc);
    if (v.fixedpoint) {
      ALLOC(v.cG, 3 * B * n * n);
      ALLOC(v.cxi, 3 * B * n * d);
      ALLOC(v.cLam, 3 * B * n * n);
      ALLOC(v.ck_G, B * Kc * n * n);
      ALLOC(v.ck_xi, B * Kc * n * d);
      ALLOC(v.ck_Lam, B * Kc * n * n);
    }
    v.atol = h->dAtol; v.rtol = h->dRtol; v.prm = nprm ? h->dPrm : nullptr; v.errnorm = h->zErr;
    v.user_f = h->dUserF; v.user_J = h->dUserJ;
  } else {
    DenseView& v = h->dn;
    v.B = int(B); v.d = d; v.k = k; v.n = int(n); v.D = int(D); v.K = int(Kc);
    v.ts1 = cfg->correction == PNODE_CORR_TS1;
    v.dynamic = cfg->calibration == PNODE_CALIB_DYNAMIC;
    v.vf_id = cfg->vf_id; v.P = cfg->num_vf_params; v.prm_per_traj = cfg->vf_params_per_trajectory;
    v.fixedpoint = cfg->strategy == PNODE_STRATEGY_FIXEDPOINT;
    v.nslots = v.fixedpoint ? 5 : 3;
    v.mr_rows = v.fixedpoint ? 2 : 1;
    if (v.fixedpoint) v.force_generic = 1;  // the extra-row LQ of the smoother runs on the generic kernels
    const size_t NS = v.nslots;
    ALLOC(v.mean, NS * B * n * d);
    ALLOC(v.U, NS * B * D * D);
    ALLOC(v.sigma, NS * B);
    ALLOC(v.ck_mean, B * Kc * n * d);
    ALLOC(v.ck_steps, B * Kc);
    if ((cfg->reserved & PNODE_FLAG_CHECKPOINT_FACTORS) && !v.fixedpoint) ALLOC(v.ck_fac, B * Kc * D * D);
    ALLOC(v.M, B * v.mr_rows * D * D);
    ALLOC(v.R, B * v.mr_rows * D * D);
    ALLOC(v.act, B);
    if (v.fixedpoint) {
      ALLOC(v.mext, B * D);
      ALLOC(v.cG, NS * B * D * D);
      ALLOC(v.cxi, NS * B * D);
      ALLOC(v.cLam, NS * B * D * D);
      ALLOC(v.ck_G, B * Kc * D * D);
      ALLOC(v.ck_xi, B * Kc * D);
      ALLOC(v.ck_Lam, B * Kc * D * D);
    }
    ALLOC(v.J, B * d * size_t(k) * d);
    ALLOC(v.z, B * d);
    ALLOC(v.err, B * d);
    ALLOC(v.Ft, B * d * D);
    ALLOC(v.X, B * D * d);
    ALLOC(v.S, B * d * d);
    ALLOC(v.sig_new, B);
    {
      const int DT = int(D) / 16;
      (void)DT;
      if (v.ts1 && d % 32 == 0 && D % 64 == 0 && D >= 64 && D <= 1024) {
        ALLOC(v.lgG, B * v.mr_rows * D * 128);
        ALLOC(v.lgW, B * v.mr_rows * D * 128);
        ALLOC(v.lgTf, B * 4 * 32 * 32);
        ALLOC(v.lgGram, B * 128 * 128);
        ALLOC(v.lgTfo, B * 128 * 128);
        ALLOC(v.lgOff, B);
        if (v.fixedpoint) {
          ALLOC(v.lgMaskMerge, B);
          ALLOC(v.lgOffS, B);
          ALLOC(v.lgOffD, B);
          ALLOC(v.lgXi, B * D);
        }
      }
    }
    v.atol = h->dAtol; v.rtol = h->dRtol; v.prm = nprm ? h->dPrm : nullptr; v.errnorm = h->zErr;
    v.user_f = h->dUserF; v.user_J = h->dUserJ;
    HIP_TRY_H(hipHostGetDevicePointer(reinterpret_cast<void**>(&v.fault), h->hFault, 0));
  }
  h->t_sf.assign(B, 0.0);
  h->t_if.assign(B, 0.0);
  h->dt_pending.assign(B, 0.0);
  h->s_sf.assign(B, 0);
  h->s_if.assign(B, 0);
  h->s_pr.assign(B, 1);
  h->steps.assign(B, 0);
  h->pending.assign(B, 0);
  // default tolerances (ivpsolve.adaptive defaults: atol 1e-4, rtol 1e-2)
  std::vector<double> a(B, 1e-4), r(B, 1e-2);
  HIP_TRY_H(hipMemcpy(h->dAtol, a.data(), B * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY_H(hipMemcpy(h->dRtol, r.data(), B * sizeof(double), hipMemcpyHostToDevice));
  *out = h;
  return PNODE_OK;
}

static double* mean_base(pnode_batch* h) { return h->dense ? h->dn.mean : h->iso.mean; }
static double* sigma_base(pnode_batch* h) { return h->dense ? h->dn.sigma : h->iso.sigma; }

int pnode_set_initial(pnode_batch* h, const double* tcoeffs, const double* t0, const double* output_scale) {
  if (!h || !tcoeffs || !t0 || !output_scale) return fail(PNODE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, n = h->n, d = h->cfg.d, D = h->D;
  double* mean = mean_base(h);
  double* sig = sigma_base(h);
  HIP_TRY(hipMemcpyAsync(mean, tcoeffs, B * n * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(sig, output_scale, B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (!h->dense) {
    HIP_TRY(hipMemsetAsync(h->iso.fac, 0, 3 * B * n * n * sizeof(double), h->stream));
    if (h->iso.fixedpoint) {
      std::vector<double> eye(B * n * n, 0.0);
      for (size_t b = 0; b < B; ++b)
        for (size_t i = 0; i < n; ++i) eye[(b * n + i) * n + i] = 1.0;
      HIP_TRY(hipMemcpyAsync(h->iso.cG, eye.data(), B * n * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIP_TRY(hipMemsetAsync(h->iso.cxi, 0, 3 * B * n * d * sizeof(double), h->stream));
      HIP_TRY(hipMemsetAsync(h->iso.cLam, 0, 3 * B * n * n * sizeof(double), h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
  } else {
    // every slot: the paths store the factor in different layouts, and a handle may change path after set_initial
    HIP_TRY(hipMemsetAsync(h->dn.U, 0, size_t(h->dn.nslots) * B * D * D * sizeof(double), h->stream));
    if (h->dn.fixedpoint) {
      std::vector<double> eye(B * D * D, 0.0);
      for (size_t b = 0; b < B; ++b)
        for (size_t i = 0; i < D; ++i) eye[(b * D + i) * D + i] = 1.0;
      HIP_TRY(hipMemcpyAsync(h->dn.cG, eye.data(), B * D * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIP_TRY(hipMemsetAsync(h->dn.cxi, 0, size_t(h->dn.nslots) * B * D * sizeof(double), h->stream));
      HIP_TRY(hipMemsetAsync(h->dn.cLam, 0, size_t(h->dn.nslots) * B * D * D * sizeof(double), h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t b = 0; b < B; ++b) {
    h->t_sf[b] = h->t_if[b] = t0[b];
    h->s_sf[b] = h->s_if[b] = 0;
    h->s_pr[b] = 1;
    h->steps[b] = 0;
    h->pending[b] = 0;
  }
  h->initialised = true;
  h->stepped = false;
  return PNODE_OK;
}

int pnode_set_tolerances(pnode_batch* h, const double* atol, const double* rtol) {
  if (!h || !atol || !rtol) return fail(PNODE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch;
  for (size_t b = 0; b < B; ++b)
    if (!(atol[b] >= 0.0) || !(rtol[b] >= 0.0) || (atol[b] == 0.0 && rtol[b] == 0.0))
      return fail(PNODE_EINVAL, "tolerances must be non-negative and not both zero");
  HIP_TRY(hipMemcpy(h->dAtol, atol, B * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->dRtol, rtol, B * sizeof(double), hipMemcpyHostToDevice));
  return PNODE_OK;
}

// After a stream synchronisation: a hand-shake of the one-launch LQ that ran into its spin limit is a device error, never a
// numerical result (the kernel drains and fills the outputs with NaN, which would otherwise read as a failed ODE).
static int check_fault(pnode_batch* h) {
  if (!h->hFault || !h->hFault[0]) return PNODE_OK;
  if (h->hFault[0] == 2) {  // pipelined controller of the isotropic fixed-point smoother (csrc/pnode_iso.hip)
    static const char* wait[3] = {"ring full", "accepted step not handed over", "composition of the step before"};
    const int w = h->hFault[3];
    const std::string msg = "hand-shake of the pipelined smoother controller timed out (" + std::string(wait[w >= 0 && w < 3 ? w : 0]) +
                            "), trajectory " + std::to_string(h->hFault[1]) + ": the state of the batch is undefined, call pnode_set_initial";
    std::memset(h->hFault, 0, 4 * sizeof(int32_t));
    h->initialised = false;
    return fail(PNODE_EDEVICE, msg);
  }
  static const char* kind[3] = {"panel / stored-strip flag", "LDS slot hand-over", "coefficient tables"};
  const int what = h->hFault[3] / 8, wave = h->hFault[3] % 8;
  const std::string msg = "LQ hand-shake timed out (" + std::string(kind[what < 3 && what >= 0 ? what : 0]) + "), trajectory " +
                          std::to_string(h->hFault[1]) + ", awaited value " + std::to_string(h->hFault[2]) + ", wavefront " +
                          std::to_string(wave) + ": the state of the batch is undefined, call pnode_set_initial";
  std::memset(h->hFault, 0, 4 * sizeof(int32_t));
  h->initialised = false;
  return fail(PNODE_EDEVICE, msg);
}

static void enqueue_step(pnode_batch* h, const StepArg* dargs, double* errnorm = nullptr, const PhaseEvents* events = nullptr,
                         const FusedControl* fc = nullptr) {
  h->stepped = true;
  if (h->dense) {
    const PhaseEvents* ev = h->phase_timing ? (events ? events : &h->phase) : nullptr;
    if (errnorm) {  // error norms to device memory (device-resident controller) instead of the pinned host buffer
      DenseView v = h->dn;
      v.errnorm = errnorm;
      dense_launch_step(v, dargs, h->dK, h->stream, ev, fc);
    } else {
      dense_launch_step(h->dn, dargs, h->dK, h->stream, ev, fc);
    }
  } else if (errnorm) {
    IsoView v = h->iso;
    v.errnorm = errnorm;
    iso_launch_step(v, dargs, h->dK, h->stream);
  } else {
    iso_launch_step(h->iso, dargs, h->dK, h->stream);
  }
}

// `rounds` batched attempts enqueued back to back, each followed by the controller kernel; no host synchronisation inside.
// hCtl[b].status: in 0 = take part, < 0 = skip; out 0 = still running, 1 = reached t_stop, 2 = failed.
static int advance_rounds_device(pnode_batch* h, TrajCtl* hCtl, const ControlParams& cp, int rounds) {
  const size_t B = h->cfg.batch;
  const unsigned blocks = unsigned((B + 255) / 256);
  HIP_TRY(hipMemcpyAsync(h->dCtl, hCtl, B * sizeof(TrajCtl), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(dense_ctl_kernel, dim3(blocks), dim3(256), 0, h->stream, h->dCtl, h->dArgs, h->dErr, cp, int(B), 1);
  if (h->step_timing) HIP_TRY(hipEventRecord(h->ev0, h->stream));
  const bool per_round = h->dense && h->phase_timing;
  if (per_round)
    while (h->phase_pool.size() < size_t(rounds)) {
      PhaseEvents pe{};
      for (auto& e : pe.ev) HIP_TRY(hipEventCreate(&e));
      h->phase_pool.push_back(pe);
    }
  // MFMA path of the dense EKF1: the measurement-update kernel carries the controller's decision (three launches per attempt:
  // begin, LQ, measurement update); every other path: controller kernel between the attempts
  const bool fused = h->dense && dense_fused_control_supported(h->dn);
  for (int r = 0; r < rounds; ++r) {
    if (fused) {
      const FusedControl fc{h->dCtl, h->dArgs, cp};
      enqueue_step(h, h->dArgs, h->dErr, per_round ? &h->phase_pool[size_t(r)] : nullptr, &fc);
    } else {
      enqueue_step(h, h->dArgs, h->dErr, per_round ? &h->phase_pool[size_t(r)] : nullptr);
      if (h->dense && h->dn.fixedpoint) {  // backward transition of the attempt, merged with the running one (as pnode_attempt_step)
        hipLaunchKernelGGL(fp_args_from_step_args_kernel, dim3(blocks), dim3(256), 0, h->stream, h->dArgs, h->dFp, int(B));
        dense_fp_launch_step_tail(h->dn, h->dFp, h->dK, h->stream);
      }
      hipLaunchKernelGGL(dense_ctl_kernel, dim3(blocks), dim3(256), 0, h->stream, h->dCtl, h->dArgs, h->dErr, cp, int(B), 0);
    }
  }
  if (h->step_timing) HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(hCtl, h->dCtl, B * sizeof(TrajCtl), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (per_round) {  // phase durations of EVERY attempt of this call (pnode_phase_ms_mean)
    for (double& x : h->phase_sum) x = 0.0;
    h->phase_count = rounds;
    for (int r = 0; r < rounds; ++r)
      for (int i = 0; i < 4; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, h->phase_pool[size_t(r)].ev[i], h->phase_pool[size_t(r)].ev[i + 1]));
        h->phase_sum[i] += ms;
      }
    h->phase_last = rounds - 1;  // pnode_last_phase_ms keeps meaning "the last attempt"
  }
  return check_fault(h);
}

int pnode_attempt_step(pnode_batch* h, const double* dt, const uint8_t* active, double* err_norm_out) {
  if (!h || !dt || !active || !err_norm_out) return fail(PNODE_EINVAL, "null argument");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch;
  for (size_t b = 0; b < B; ++b) {
    StepArg& a = h->hArgs[b];
    a.dt = dt[b];
    a.t = h->t_sf[b];
    a.src = h->s_sf[b];
    a.dst = h->s_pr[b];
    a.active = active[b] ? 1 : 0;
    a.pad = 0;
    if (active[b] && !(dt[b] > 0.0)) return fail(PNODE_EINVAL, "dt must be positive for active trajectories");
    h->pending[b] = active[b] ? 1 : 0;
    h->dt_pending[b] = dt[b];
  }
  // zero-copy hand-over: the kernels read StepArg[B] from and write err_norm[B] to pinned host memory, so a step
  // costs the launches plus one stream synchronisation (no hipMemcpy calls on the critical path)
  if (h->step_timing) HIP_TRY(hipEventRecord(h->ev0, h->stream));
  h->phase_last = -1;
  enqueue_step(h, h->zArgs);
  if (h->dense && h->dn.fixedpoint) {
    for (size_t b = 0; b < B; ++b) {
      const StepArg& a = h->hArgs[b];
      h->hFp0[b] = FpArg{a.dt, a.src, a.dst, 1, a.active, 0, 0};
    }
    dense_fp_launch_step_tail(h->dn, h->zFp0, h->dK, h->stream);
  }
  if (h->step_timing) HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (int rc = check_fault(h)) return rc;
  for (size_t b = 0; b < B; ++b)
    if (active[b]) err_norm_out[b] = h->hErr[b];
  return PNODE_OK;
}

int pnode_user_extrapolate(pnode_batch* h, const double* dt, const uint8_t* active, double* y_out) {
  if (!h || !dt || !active || !y_out) return fail(PNODE_EINVAL, "null argument");
  if (h->cfg.vf_id != PNODE_VF_USER) return fail(PNODE_ESTATE, "pnode_user_extrapolate needs vf_id = PNODE_VF_USER");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, d = h->cfg.d, k = h->cfg.ode_order;
  for (size_t b = 0; b < B; ++b) {
    StepArg& a = h->hArgs[b];
    a.dt = dt[b]; a.t = h->t_sf[b]; a.src = h->s_sf[b]; a.dst = h->s_pr[b]; a.active = active[b] ? 1 : 0; a.pad = 0;
    if (active[b] && !(dt[b] > 0.0)) return fail(PNODE_EINVAL, "dt must be positive for active trajectories");
  }
  hipLaunchKernelGGL(user_extrapolate_kernel, dim3(unsigned(B)), dim3(64), 0, h->stream, mean_base(h), h->zArgs, h->dK, h->dUserY,
                     int(B), h->n, int(d), int(k));
  HIP_TRY(hipGetLastError());
  std::vector<double> tmp(B * k * d);
  HIP_TRY(hipMemcpyAsync(tmp.data(), h->dUserY, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t b = 0; b < B; ++b)
    if (active[b]) std::memcpy(y_out + b * k * d, tmp.data() + b * k * d, k * d * sizeof(double));
  return PNODE_OK;
}

int pnode_user_set_vf(pnode_batch* h, const double* f, const double* jac) {
  if (!h || !f) return fail(PNODE_EINVAL, "null argument");
  if (h->cfg.vf_id != PNODE_VF_USER) return fail(PNODE_ESTATE, "pnode_user_set_vf needs vf_id = PNODE_VF_USER");
  if (!jac && h->cfg.correction == PNODE_CORR_TS1) return fail(PNODE_EINVAL, "correction_ts1 needs the Jacobian");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, d = h->cfg.d, k = h->cfg.ode_order;
  HIP_TRY(hipMemcpyAsync(h->dUserF, f, B * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (jac) HIP_TRY(hipMemcpyAsync(h->dUserJ, jac, B * d * k * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return PNODE_OK;
}

int pnode_commit(pnode_batch* h, const uint8_t* accept) {
  if (!h || !accept) return fail(PNODE_EINVAL, "null argument");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  const size_t B = h->cfg.batch;
  for (size_t b = 0; b < B; ++b) {
    if (!accept[b]) continue;
    if (!h->pending[b]) return fail(PNODE_ESTATE, "commit without a pending attempt");
    h->pending[b] = 0;
    h->s_if[b] = h->s_sf[b];
    h->t_if[b] = h->t_sf[b];
    h->s_sf[b] = h->s_pr[b];
    h->t_sf[b] += h->dt_pending[b];
    h->steps[b] += 1;
    int8_t fr = 0;
    while (fr == h->s_if[b] || fr == h->s_sf[b]) ++fr;
    h->s_pr[b] = fr;
  }
  return PNODE_OK;
}

static int run_interp(pnode_batch* h, const double* t, const int32_t* slot, bool at_step) {
  if (!h || !slot || (!at_step && !t)) return fail(PNODE_EINVAL, "null argument");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch;
  for (size_t b = 0; b < B; ++b) {
    InterpArg& a = h->hIargs[b];
    a.ckpt = slot[b];
    if (slot[b] >= h->cfg.num_checkpoints) return fail(PNODE_EINVAL, "checkpoint slot out of range");
    a.t = at_step ? h->t_sf[b] : t[b];
    a.t0 = h->t_if[b];
    a.t1 = h->t_sf[b];
    a.s0 = h->s_if[b];
    a.s1 = h->s_sf[b];
    a.at_step = at_step ? 1 : 0;
    a.steps = h->steps[b];
    if (slot[b] >= 0 && !at_step) {
      if (!(t[b] > h->t_if[b]) || !(t[b] < h->t_sf[b]))
        return fail(PNODE_EINVAL, "interpolation time outside (INTERP_FROM.t, STEP_FROM.t)");
      if (h->s_if[b] == h->s_sf[b]) return fail(PNODE_ESTATE, "interpolate needs an accepted step first");
    }
  }
  HIP_TRY(hipMemcpyAsync(h->dIargs, h->hIargs, B * sizeof(InterpArg), hipMemcpyHostToDevice, h->stream));
  if (h->dense && h->dn.fixedpoint) {
    for (size_t b = 0; b < B; ++b) {
      const InterpArg& a = h->hIargs[b];
      const int32_t act = (a.ckpt >= 0 && !a.at_step) ? 1 : 0;
      h->hFp0[b] = FpArg{a.t - a.t0, a.s0, 3, 1, act, 0, 0};  // t0 -> t, merged with the running transition
      h->hFp1[b] = FpArg{a.t1 - a.t, 3, 4, 0, act, 0, 0};     // t -> t1 from the identity
    }
    dense_fp_launch_interp(h->dn, h->dIargs, h->zFp0, h->zFp1, h->dK, h->stream);
  } else if (h->dense) {
    dense_launch_interp(h->dn, h->dIargs, h->dK, h->stream);
    if (h->dn.ck_fac) {  // covariance factor at the checkpoint: a time update without a measurement, on the generic kernels
      for (size_t b = 0; b < B; ++b) {
        const InterpArg& a = h->hIargs[b];
        h->hArgs[b] = StepArg{at_step ? 1.0 : a.t - a.t0, a.t0, a.s0, a.s0, (a.ckpt >= 0 && !a.at_step) ? 1 : 0, 0};
      }
      HIP_TRY(hipMemcpyAsync(h->dArgs, h->hArgs, B * sizeof(StepArg), hipMemcpyHostToDevice, h->stream));
      dense_launch_checkpoint_factors(h->dn, h->dIargs, h->dArgs, h->dK, h->stream);
    }
  } else
    iso_launch_interp(h->iso, h->dIargs, h->dK, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t b = 0; b < B; ++b) {
    if (slot[b] < 0) continue;
    if (at_step) {
      h->s_if[b] = h->s_sf[b];
      h->t_if[b] = h->t_sf[b];
      int8_t fr = 0;
      while (fr == h->s_sf[b]) ++fr;
      h->s_pr[b] = fr;
    } else if (!h->dense || h->dn.fixedpoint) {
      h->t_if[b] = t[b];  // INTERP_FROM was advanced to the checkpoint in place
    }
  }
  return PNODE_OK;
}

int pnode_interpolate(pnode_batch* h, const double* t, const int32_t* slot) { return run_interp(h, t, slot, false); }
int pnode_checkpoint_at_step(pnode_batch* h, const int32_t* slot) { return run_interp(h, nullptr, slot, true); }

int pnode_solve_adaptive_save_at(pnode_batch* h, const double* save_at, int32_t num_save, int32_t per_traj,
                                 const double* dt0, const pnode_control* ctrl, double nz, int64_t max_attempts,
                                 int64_t* num_steps_out, int64_t* num_attempts_out, uint8_t* failed_out) {
  if (!h || !save_at || !dt0 || !ctrl || !num_steps_out || !num_attempts_out || !failed_out)
    return fail(PNODE_EINVAL, "null argument");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  if (h->cfg.vf_id == PNODE_VF_USER)
    return fail(PNODE_EINVAL, "host-evaluated vector fields are driven attempt by attempt (pnode_user_extrapolate / pnode_user_set_vf)");
  const size_t B = h->cfg.batch;
  const int K = num_save - 1;
  if (K < 1 || K > h->cfg.num_checkpoints) return fail(PNODE_EINVAL, "num_save - 1 must be in 1..num_checkpoints");
  auto sa = [&](size_t b, int i) { return per_traj ? save_at[b * num_save + i] : save_at[i]; };
  for (size_t b = 0; b < B; ++b) {
    for (int i = 1; i < num_save; ++i)
      if (!(sa(b, i) > sa(b, i - 1))) return fail(PNODE_EINVAL, "save_at must be strictly increasing");
    if (!(dt0[b] > 0.0)) return fail(PNODE_EINVAL, "dt0 must be positive");
  }
  const double rate = double(h->cfg.nu + 1);
  const double n1 = ctrl->power_integral_unscaled / rate, n2 = ctrl->power_proportional_unscaled / rate;
  // factor = safety * (1/e)^n1 * (e_prev/e)^n2 = safety * exp(n2*log(e_prev) - (n1+n2)*log(e)); log(e_prev) is cached
  std::vector<double> dt(dt0, dt0 + B), log_prev(B, 0.0), err(B, 0.0), dt_used(B, 1.0), tq(B, 0.0);
  std::vector<int32_t> cursor(B, 1), slot(B, -1);
  std::vector<uint8_t> active(B, 0), accept(B, 0);
  std::vector<int64_t> attempts(B, 0);
  for (size_t b = 0; b < B; ++b) {
    failed_out[b] = 0;
    for (int i = 0; i < num_save; ++i) num_steps_out[b * num_save + i] = 0;
  }
  int64_t rounds = 0;
  for (;;) {
    // ---- checkpoints reached by STEP_FROM (possibly several per accepted step)
    for (;;) {
      bool any_over = false, any_corner = false;
      for (size_t b = 0; b < B; ++b) {
        slot[b] = -1;
        if (cursor[b] > K || failed_out[b]) continue;
        const double nxt = sa(b, cursor[b]);
        if (h->t_sf[b] + nz >= nxt && h->t_sf[b] > nxt + nz) {
          slot[b] = cursor[b] - 1;
          tq[b] = nxt;
          any_over = true;
        }
      }
      if (any_over) {
        int rc = pnode_interpolate(h, tq.data(), slot.data());
        if (rc != PNODE_OK) return rc;
        for (size_t b = 0; b < B; ++b)
          if (slot[b] >= 0) {
            num_steps_out[b * num_save + cursor[b]] = h->steps[b];
            cursor[b] += 1;
          }
      }
      for (size_t b = 0; b < B; ++b) {
        slot[b] = -1;
        if (cursor[b] > K || failed_out[b]) continue;
        const double nxt = sa(b, cursor[b]);
        if (h->t_sf[b] + nz >= nxt && !(h->t_sf[b] > nxt + nz)) {
          slot[b] = cursor[b] - 1;
          any_corner = true;
        }
      }
      if (any_corner) {
        int rc = pnode_checkpoint_at_step(h, slot.data());
        if (rc != PNODE_OK) return rc;
        for (size_t b = 0; b < B; ++b)
          if (slot[b] >= 0) {
            num_steps_out[b * num_save + cursor[b]] = h->steps[b];
            cursor[b] += 1;
          }
      }
      if (!any_over && !any_corner) break;
    }
    // ---- one batched attempt for every trajectory that still has a checkpoint ahead
    bool any = false;
    for (size_t b = 0; b < B; ++b) {
      active[b] = (cursor[b] <= K && !failed_out[b]) ? 1 : 0;
      double d = dt[b];
      if (active[b] && ctrl->clip) d = std::fmin(d, sa(b, cursor[b]) - h->t_sf[b]);
      dt_used[b] = active[b] ? d : 1.0;
      any = any || active[b];
    }
    if (!any) break;
    if (max_attempts > 0 && rounds >= max_attempts) return fail(PNODE_ESTATE, "max_attempts reached");
    ++rounds;
    int rc = pnode_attempt_step(h, dt_used.data(), active.data(), err.data());
    if (rc != PNODE_OK) return rc;
    for (size_t b = 0; b < B; ++b) {
      accept[b] = 0;
      if (!active[b]) continue;
      attempts[b] += 1;
      const double e = err[b];
      if (e != e) {  // NaN error estimate: the reference's controller turns dt into NaN; stop the trajectory
        failed_out[b] = 1;
        accept[b] = 1;  // ... after leaving the `while error_norm > 1` loop, i.e. with the NaN state accepted
        continue;
      }
      const double le = std::log(e);
      const double raw = ctrl->safety * std::exp(n2 * log_prev[b] - (n1 + n2) * le);
      const double factor = std::fmax(ctrl->factor_min, std::fmin(raw, ctrl->factor_max));
      if (e <= 1.0) log_prev[b] = le;
      dt[b] = factor * dt_used[b];
      accept[b] = !(e > 1.0);
    }
    rc = pnode_commit(h, accept.data());
    if (rc != PNODE_OK) return rc;
    for (size_t b = 0; b < B; ++b)
      if (active[b] && !(std::isfinite(dt[b]) && std::isfinite(h->t_sf[b]) && dt[b] > 0.0)) failed_out[b] = 1;
  }
  for (size_t b = 0; b < B; ++b) num_attempts_out[b] = attempts[b];
  return PNODE_OK;
}

int pnode_solve_adaptive_save_at_device(pnode_batch* h, const double* save_at, int32_t num_save, int32_t per_traj,
                                        const double* dt0, const pnode_control* ctrl, double nz, int32_t iters_per_launch,
                                        int64_t* num_steps_out, int64_t* num_attempts_out, uint8_t* failed_out) {
  if (!h || !save_at || !dt0 || !ctrl || !num_steps_out || !num_attempts_out || !failed_out)
    return fail(PNODE_EINVAL, "null argument");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  // dense model beyond the fused one-wavefront kernel: multi-kernel form (controller kernel between the attempts), filter only
  const bool multi = h->dense && !(dense_small_supported(h->dn) && !h->dn.force_generic);
  if (h->cfg.vf_id == PNODE_VF_USER) return fail(PNODE_EINVAL, "the device-resident controller needs a built-in vector field");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch;
  const int K = num_save - 1;
  if (K < 1 || K > h->cfg.num_checkpoints) return fail(PNODE_EINVAL, "num_save - 1 must be in 1..num_checkpoints");
  auto sa = [&](size_t b, int i) { return per_traj ? save_at[b * num_save + i] : save_at[i]; };
  for (size_t b = 0; b < B; ++b) {
    for (int i = 1; i < num_save; ++i)
      if (!(sa(b, i) > sa(b, i - 1))) return fail(PNODE_EINVAL, "save_at must be strictly increasing");
    if (!(dt0[b] > 0.0)) return fail(PNODE_EINVAL, "dt0 must be positive");
  }
  const double rate = double(h->cfg.nu + 1);
  ControlParams cp{ctrl->safety, ctrl->factor_min, ctrl->factor_max, ctrl->power_integral_unscaled / rate,
                   ctrl->power_proportional_unscaled / rate, nz, ctrl->clip, iters_per_launch > 0 ? iters_per_launch : 4096};
  const int rounds_per_sync = iters_per_launch > 0 ? (iters_per_launch < 64 ? iters_per_launch : 64) : 8;
  TrajCtl* hCtl = nullptr;
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&hCtl), B * sizeof(TrajCtl)));
  TrajCtl* zCtl = nullptr;
  hipError_t e0 = hipHostGetDevicePointer(reinterpret_cast<void**>(&zCtl), hCtl, 0);
  if (e0 != hipSuccess) {
    (void)hipHostFree(hCtl);
    return fail(PNODE_EDEVICE, hipGetErrorString(e0));
  }
  std::vector<double> dt(dt0, dt0 + B), log_prev(B, 0.0), tq(B, 0.0);
  std::vector<int32_t> cursor(B, 1), slot(B, -1);
  std::vector<int64_t> attempts(B, 0);
  for (size_t b = 0; b < B; ++b) {
    failed_out[b] = 0;
    for (int i = 0; i < num_save; ++i) num_steps_out[b * num_save + i] = 0;
  }
  if (!h->dense && iso_advance_walks_checkpoints(h->iso)) {
    // Isotropic model: the launch walks over the save_at points itself (interpolation / right corner inside the controller kernel);
    // the host only relaunches trajectories whose iteration budget ran out. One synchronisation per launch.
    double* dSave = h->dPlanSave;  // (allocated with the handle for its checkpoint capacity: num_save - 1 <= num_checkpoints)
    int32_t* dCur = h->dPlanCur;
    int64_t* dNum = h->dPlanNum;
    const size_t nsave = per_traj ? B * size_t(num_save) : size_t(num_save);
    auto release = [&] { (void)hipHostFree(hCtl); };
    std::vector<int32_t> cur(B, 1);
    std::vector<int64_t> num(B * size_t(num_save), 0);
    hipError_t e = hipMemcpyAsync(dSave, save_at, nsave * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dCur, cur.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dNum, num.data(), B * size_t(num_save) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) {
      release();
      return fail(PNODE_EDEVICE, hipGetErrorString(e));
    }
    for (size_t b = 0; b < B; ++b) {
      TrajCtl& c = hCtl[b];
      c.status = 0;
      c.t_sf = h->t_sf[b]; c.t_if = h->t_if[b]; c.dt = dt[b]; c.log_prev = 0.0; c.t_stop = sa(b, 1);
      c.steps = h->steps[b]; c.attempts = 0;
      c.s_sf = h->s_sf[b]; c.s_if = h->s_if[b]; c.s_pr = h->s_pr[b];
    }
    const CkptPlan plan{dSave, per_traj ? 1 : 0, K, dCur, dNum};
    int rc2 = PNODE_OK;
    for (;;) {
      iso_launch_advance(h->iso, zCtl, h->dK, cp, h->stream, plan);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) {
        rc2 = fail(PNODE_EDEVICE, hipGetErrorString(e));
        break;
      }
      rc2 = check_fault(h);
      if (rc2 != PNODE_OK) break;
      bool again = false;
      for (size_t b = 0; b < B; ++b) {
        TrajCtl& c = hCtl[b];
        if (c.status < 0) continue;
        if (c.status == 3) {  // iteration budget spent: goes on in the next launch
          c.status = 0;
          again = true;
          continue;
        }
        if (c.status == 2) failed_out[b] = 1;
        h->t_sf[b] = c.t_sf; h->t_if[b] = c.t_if; h->steps[b] = c.steps;
        h->s_sf[b] = int8_t(c.s_sf); h->s_if[b] = int8_t(c.s_if); h->s_pr[b] = int8_t(c.s_pr);
        h->pending[b] = 0;
        num_attempts_out[b] = c.attempts;
        c.status = -1;  // done (every checkpoint stored) or failed
      }
      if (!again) break;
    }
    if (rc2 == PNODE_OK) {
      e = hipMemcpy(num.data(), dNum, B * size_t(num_save) * sizeof(int64_t), hipMemcpyDeviceToHost);
      if (e != hipSuccess) rc2 = fail(PNODE_EDEVICE, hipGetErrorString(e));
    }
    release();
    if (rc2 != PNODE_OK) return rc2;
    for (size_t i = 0; i < B * size_t(num_save); ++i) num_steps_out[i] = num[i];
    h->stepped = true;
    return PNODE_OK;
  }
  int rc = PNODE_OK;
  for (;;) {
    // ---- checkpoints reached by STEP_FROM (host bookkeeping mirrors the device controller state)
    for (;;) {
      bool any_over = false, any_corner = false;
      for (size_t b = 0; b < B; ++b) {
        slot[b] = -1;
        if (cursor[b] > K || failed_out[b]) continue;
        const double nxt = sa(b, cursor[b]);
        if (h->t_sf[b] + nz >= nxt && h->t_sf[b] > nxt + nz) {
          slot[b] = cursor[b] - 1;
          tq[b] = nxt;
          any_over = true;
        }
      }
      if (any_over) {
        rc = pnode_interpolate(h, tq.data(), slot.data());
        if (rc != PNODE_OK) break;
        for (size_t b = 0; b < B; ++b)
          if (slot[b] >= 0) {
            num_steps_out[b * num_save + cursor[b]] = h->steps[b];
            cursor[b] += 1;
          }
      }
      for (size_t b = 0; b < B; ++b) {
        slot[b] = -1;
        if (cursor[b] > K || failed_out[b]) continue;
        const double nxt = sa(b, cursor[b]);
        if (h->t_sf[b] + nz >= nxt && !(h->t_sf[b] > nxt + nz)) {
          slot[b] = cursor[b] - 1;
          any_corner = true;
        }
      }
      if (any_corner) {
        rc = pnode_checkpoint_at_step(h, slot.data());
        if (rc != PNODE_OK) break;
        for (size_t b = 0; b < B; ++b)
          if (slot[b] >= 0) {
            num_steps_out[b * num_save + cursor[b]] = h->steps[b];
            cursor[b] += 1;
          }
      }
      if (!any_over && !any_corner) break;
    }
    if (rc != PNODE_OK) break;
    // ---- hand the trajectories that still have a checkpoint ahead to the device controller
    bool any = false;
    for (size_t b = 0; b < B; ++b) {
      TrajCtl& c = hCtl[b];
      const bool act = cursor[b] <= K && !failed_out[b];
      c.status = act ? 0 : -1;
      if (!act) continue;
      any = true;
      c.t_sf = h->t_sf[b]; c.t_if = h->t_if[b]; c.dt = dt[b]; c.log_prev = log_prev[b]; c.t_stop = sa(b, cursor[b]);
      c.steps = h->steps[b]; c.attempts = attempts[b];
      c.s_sf = h->s_sf[b]; c.s_if = h->s_if[b]; c.s_pr = h->s_pr[b];
    }
    if (!any) break;
    if (multi) {
      // until every trajectory handed over has reached its checkpoint (or failed): rounds of attempts without host involvement
      for (;;) {
        rc = advance_rounds_device(h, hCtl, cp, rounds_per_sync);
        if (rc != PNODE_OK) break;
        bool running = false;
        for (size_t b = 0; b < B; ++b) running = running || hCtl[b].status == 0;
        if (!running) break;
      }
      if (rc != PNODE_OK) break;
    } else {
      if (h->dense)
        dense_small_launch_advance(h->dn, zCtl, h->dK, cp, h->stream);
      else
        iso_launch_advance(h->iso, zCtl, h->dK, cp, h->stream);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) {
        rc = fail(PNODE_EDEVICE, hipGetErrorString(e));
        break;
      }
      rc = check_fault(h);
      if (rc != PNODE_OK) break;
    }
    for (size_t b = 0; b < B; ++b) {
      const TrajCtl& c = hCtl[b];
      if (c.status < 0) continue;
      h->t_sf[b] = c.t_sf; h->t_if[b] = c.t_if; h->steps[b] = c.steps;
      h->s_sf[b] = int8_t(c.s_sf); h->s_if[b] = int8_t(c.s_if); h->s_pr[b] = int8_t(c.s_pr);
      h->pending[b] = 0;
      dt[b] = c.dt; log_prev[b] = c.log_prev; attempts[b] = c.attempts;
      if (c.status == 2) failed_out[b] = 1;
    }
  }
  (void)hipHostFree(hCtl);
  if (rc != PNODE_OK) return rc;
  for (size_t b = 0; b < B; ++b) num_attempts_out[b] = attempts[b];
  return PNODE_OK;
}

int pnode_advance_device(pnode_batch* h, const double* t_stop, double* dt_inout, double* log_err_prev_inout, const pnode_control* ctrl,
                         double nz, int32_t rounds, int32_t* status_out, int64_t* attempts_out) {
  if (!h || !t_stop || !dt_inout || !log_err_prev_inout || !ctrl || !status_out || !attempts_out || rounds < 1)
    return fail(PNODE_EINVAL, "bad argument");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  if (h->cfg.vf_id == PNODE_VF_USER) return fail(PNODE_EINVAL, "the device-resident controller needs a built-in vector field");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch;
  const double rate = double(h->cfg.nu + 1);
  ControlParams cp{ctrl->safety, ctrl->factor_min, ctrl->factor_max, ctrl->power_integral_unscaled / rate,
                   ctrl->power_proportional_unscaled / rate, nz, ctrl->clip, 0};
  std::vector<TrajCtl> c(B);
  for (size_t b = 0; b < B; ++b) {
    TrajCtl& x = c[b];
    x.status = status_out[b] < 0 ? -1 : 0;
    x.t_sf = h->t_sf[b]; x.t_if = h->t_if[b]; x.dt = dt_inout[b]; x.log_prev = log_err_prev_inout[b]; x.t_stop = t_stop[b];
    x.steps = h->steps[b]; x.attempts = 0;
    x.s_sf = h->s_sf[b]; x.s_if = h->s_if[b]; x.s_pr = h->s_pr[b];
    if (x.status == 0 && !(x.dt > 0.0)) return fail(PNODE_EINVAL, "dt must be positive for the trajectories taking part");
  }
  int rc = advance_rounds_device(h, c.data(), cp, rounds);
  if (rc != PNODE_OK) return rc;
  for (size_t b = 0; b < B; ++b) {
    const TrajCtl& x = c[b];
    status_out[b] = x.status;
    attempts_out[b] = x.attempts;
    if (x.status < 0) continue;
    h->t_sf[b] = x.t_sf; h->t_if[b] = x.t_if; h->steps[b] = x.steps;
    h->s_sf[b] = int8_t(x.s_sf); h->s_if[b] = int8_t(x.s_if); h->s_pr[b] = int8_t(x.s_pr);
    h->pending[b] = 0;
    dt_inout[b] = x.dt; log_err_prev_inout[b] = x.log_prev;
  }
  return PNODE_OK;
}

int pnode_solve_fixed_grid(pnode_batch* h, const double* grid, int32_t G, int32_t per_traj, double* u_out) {
  if (!h || !grid || !u_out || G < 1) return fail(PNODE_EINVAL, "bad argument");
  if (!h->initialised) return fail(PNODE_ESTATE, "set_initial has not been called");
  if (h->dense && h->dn.fixedpoint) return fail(PNODE_EINVAL, "solve_fixed_grid: use strategy_filter with the dense model");
  if (h->cfg.vf_id == PNODE_VF_USER) return fail(PNODE_EINVAL, "solve_fixed_grid replays on the device: it needs a built-in vector field");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, n = h->n, d = h->cfg.d;
  std::vector<StepArg> all(size_t(G) * B);
  // step g (1-based) goes from grid[g-1] to grid[g]; entry 0 only carries the initial slot
  std::vector<int8_t> sf = h->s_sf, si = h->s_if, sp = h->s_pr;
  for (size_t b = 0; b < B; ++b) {
    all[b] = StepArg{0.0, 0.0, sf[b], sf[b], 0, 0};
  }
  for (int g = 1; g < G; ++g)
    for (size_t b = 0; b < B; ++b) {
      const double* gr = per_traj ? grid + b * G : grid;
      StepArg& a = all[size_t(g) * B + b];
      a.dt = gr[g] - gr[g - 1];
      a.t = gr[g - 1];
      if (!(a.dt > 0.0)) return fail(PNODE_EINVAL, "grid must be strictly increasing");
      a.src = sf[b];
      a.dst = sp[b];
      a.active = 1;
      a.pad = 0;
      si[b] = sf[b];
      sf[b] = sp[b];
      int8_t fr = 0;
      while (fr == si[b] || fr == sf[b]) ++fr;
      sp[b] = fr;
    }
  StepArg* dAll = nullptr;
  double* dU = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dAll), all.size() * sizeof(StepArg)));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&dU), B * size_t(G) * d * sizeof(double));
  if (e != hipSuccess) {
    (void)hipFree(dAll);
    return fail(PNODE_EDEVICE, hipGetErrorString(e));
  }
  auto cleanup = [&]() {
    (void)hipFree(dAll);
    (void)hipFree(dU);
  };
  e = hipMemcpyAsync(dAll, all.data(), all.size() * sizeof(StepArg), hipMemcpyHostToDevice, h->stream);
  if (e != hipSuccess) { cleanup(); return fail(PNODE_EDEVICE, hipGetErrorString(e)); }
  double* mean = mean_base(h);
  e = hipEventRecord(h->ev0, h->stream);
  if (e != hipSuccess) { cleanup(); return fail(PNODE_EDEVICE, hipGetErrorString(e)); }
  if (h->dense && dense_d1_fixed_grid_supported(h->dn)) {  // scalar ODE: the whole grid in one launch (csrc/pnode_dense_small.hip)
    h->stepped = true;
    dense_d1_launch_fixed_grid(h->dn, dAll, h->dK, dU, G, h->stream);
  } else {
    for (int g = 0; g < G; ++g) {
      if (g > 0) enqueue_step(h, dAll + size_t(g) * B);
      hipLaunchKernelGGL(gather_u_kernel, dim3(unsigned(B)), dim3(64), 0, h->stream, mean, dAll + size_t(g) * B, dU, int(B),
                         int(n), int(d), G, g);
    }
  }
  e = hipEventRecord(h->ev1, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(u_out, dU, B * size_t(G) * d * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipGetLastError();
  cleanup();
  if (e != hipSuccess) return fail(PNODE_EDEVICE, hipGetErrorString(e));
  if (int rc = check_fault(h)) return rc;
  for (size_t b = 0; b < B; ++b) {
    const double* gr = per_traj ? grid + b * G : grid;
    h->s_sf[b] = sf[b];
    h->s_if[b] = si[b];
    h->s_pr[b] = sp[b];
    h->t_if[b] = G > 1 ? gr[G - 2] : h->t_sf[b];
    h->t_sf[b] = gr[G - 1];
    h->steps[b] += G - 1;
    h->pending[b] = 0;
  }
  return PNODE_OK;
}

static int8_t slot_of(pnode_batch* h, int which, size_t b) {
  return which == 0 ? h->s_sf[b] : (which == 1 ? h->s_if[b] : h->s_pr[b]);
}

int pnode_read_mean(pnode_batch* h, int32_t which, double* out) {
  if (!h || !out || which < 0 || which > 2) return fail(PNODE_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, nd = size_t(h->n) * h->cfg.d;
  std::vector<double> tmp(3 * B * nd);
  HIP_TRY(hipMemcpy(tmp.data(), mean_base(h), tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (size_t b = 0; b < B; ++b)
    std::memcpy(out + b * nd, tmp.data() + (size_t(slot_of(h, which, b)) * B + b) * nd, nd * sizeof(double));
  return PNODE_OK;
}

int pnode_read_time(pnode_batch* h, double* t_out, int64_t* steps_out) {
  if (!h) return fail(PNODE_EINVAL, "null handle");
  const size_t B = h->cfg.batch;
  for (size_t b = 0; b < B; ++b) {
    if (t_out) t_out[b] = h->t_sf[b];
    if (steps_out) steps_out[b] = h->steps[b];
  }
  return PNODE_OK;
}

int pnode_read_output_scale(pnode_batch* h, int32_t which, double* out) {
  if (!h || !out || which < 0 || which > 2) return fail(PNODE_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch;
  std::vector<double> tmp(3 * B);
  HIP_TRY(hipMemcpy(tmp.data(), sigma_base(h), tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (size_t b = 0; b < B; ++b) out[b] = tmp[size_t(slot_of(h, which, b)) * B + b];
  return PNODE_OK;
}

int pnode_read_factor(pnode_batch* h, int32_t which, double* out) {
  if (!h || !out || which < 0 || which > 2) return fail(PNODE_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch;
  const size_t m = h->dense ? size_t(h->D) : size_t(h->n);
  const double* base = h->dense ? h->dn.U : h->iso.fac;
  std::vector<double> tmp(m * m);
  for (size_t b = 0; b < B; ++b) {
    HIP_TRY(hipMemcpy(tmp.data(), base + (size_t(slot_of(h, which, b)) * B + b) * m * m, m * m * sizeof(double),
                      hipMemcpyDeviceToHost));
    if (h->dense && dense_state_factor_tiled(h->dn)) {  // Ls (D x (D - d)) in the register order of the 16-row strips: hand out R = [Ls | 0]^T
      for (size_t i = 0; i < m; ++i)      // row of L
        for (size_t j = 0; j < m; ++j)    // column of L
          out[b * m * m + j * m + i] = tiled_factor_entry(tmp.data(), int(i), int(j), int(m), h->cfg.d);
    } else if (h->dense) {  // stored as L (Sigma = L L^T): hand out R = L^T
      for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; j < m; ++j) out[b * m * m + i * m + j] = tmp[j * m + i];
    } else {
      std::memcpy(out + b * m * m, tmp.data(), m * m * sizeof(double));
    }
  }
  return PNODE_OK;
}

int pnode_read_checkpoint_means(pnode_batch* h, double* out) {
  if (!h || !out) return fail(PNODE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t cnt = size_t(h->cfg.batch) * h->cfg.num_checkpoints * h->n * h->cfg.d;
  if (cnt == 0) return PNODE_OK;
  HIP_TRY(hipMemcpy(out, h->dense ? h->dn.ck_mean : h->iso.ck_mean, cnt * sizeof(double), hipMemcpyDeviceToHost));
  return PNODE_OK;
}

int pnode_read_checkpoint_factors(pnode_batch* h, double* out) {
  if (!h || !out) return fail(PNODE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, K = h->cfg.num_checkpoints;
  const size_t m = h->dense ? size_t(h->D) : size_t(h->n);
  const double* src = h->dense ? h->dn.ck_fac : h->iso.ck_fac;
  if (!src) return fail(PNODE_ESTATE, "checkpoint factors were not requested at creation (PNODE_FLAG_CHECKPOINT_FACTORS) or the "
                                     "strategy keeps backward transitions instead (dense fixedpoint)");
  const size_t cnt = B * K * m * m;
  if (cnt == 0) return PNODE_OK;
  if (!h->dense) {
    HIP_TRY(hipMemcpy(out, src, cnt * sizeof(double), hipMemcpyDeviceToHost));
    return PNODE_OK;
  }
  std::vector<double> tmp(m * m);
  for (size_t i = 0; i < B * K; ++i) {  // stored as L (Sigma = L L^T): hand out R = L^T like pnode_read_factor
    HIP_TRY(hipMemcpy(tmp.data(), src + i * m * m, m * m * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < m; ++r)
      for (size_t c = 0; c < m; ++c) out[i * m * m + r * m + c] = tmp[c * m + r];
  }
  return PNODE_OK;
}

int pnode_read_checkpoint_transitions(pnode_batch* h, double* G, double* xi, double* Lam) {
  if (!h || !G || !xi || !Lam) return fail(PNODE_EINVAL, "null argument");
  if (h->dense ? !h->dn.fixedpoint : !h->iso.fixedpoint)
    return fail(PNODE_EINVAL, "checkpoint transitions are stored with a backward model only (strategy fixedpoint / smoother)");
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const size_t B = h->cfg.batch, K = h->cfg.num_checkpoints, n = h->n, d = h->cfg.d;
  if (B * K == 0) return PNODE_OK;
  const size_t m = h->dense ? n * d : n, nx = n * d;  // dense: D x D matrices over the flattened state (index i d + j)
  const double* sG = h->dense ? h->dn.ck_G : h->iso.ck_G;
  const double* sx = h->dense ? h->dn.ck_xi : h->iso.ck_xi;
  const double* sL = h->dense ? h->dn.ck_Lam : h->iso.ck_Lam;
  HIP_TRY(hipMemcpy(G, sG, B * K * m * m * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(xi, sx, B * K * nx * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(Lam, sL, B * K * m * m * sizeof(double), hipMemcpyDeviceToHost));
  return PNODE_OK;
}

int pnode_smooth_checkpoint_means(pnode_batch* h, const int32_t* num_filled, double* out) {
  if (!h || !num_filled || !out) return fail(PNODE_EINVAL, "null argument");
  if ((h->dense && !h->dn.fixedpoint) || (!h->dense && !h->iso.fixedpoint))
    return fail(PNODE_EINVAL, "smoothing needs the fixedpoint strategy");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, K = h->cfg.num_checkpoints;
  for (size_t b = 0; b < B; ++b)
    if (num_filled[b] < 0 || size_t(num_filled[b]) > K) return fail(PNODE_EINVAL, "num_filled out of range");
  const size_t cnt = B * (K + 1) * h->n * h->cfg.d;
  double* dOut = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dOut), cnt * sizeof(double)));
  hipError_t e = hipMemsetAsync(dOut, 0, cnt * sizeof(double), h->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(h->dFilled, num_filled, B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    if (h->dense)
      dense_fp_launch_smooth(h->dn, h->dFilled, dOut, h->stream);
    else
      iso_launch_smooth(h->iso, h->dFilled, dOut, h->stream);
    e = hipMemcpyAsync(out, dOut, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(dOut);
  if (e != hipSuccess) return fail(PNODE_EDEVICE, hipGetErrorString(e));
  return PNODE_OK;
}

int pnode_offgrid_means(pnode_batch* h, const int32_t* num_filled, const double* init_means, int32_t num_queries, const double* t,
                        const double* t_lo, const double* t_hi, const int32_t* idx, const uint8_t* at_grid, double* out) {
  if (!h || !num_filled || !init_means || !t || !t_lo || !t_hi || !idx || !at_grid || !out || num_queries < 1)
    return fail(PNODE_EINVAL, "bad argument");
  if (h->dense || !h->iso.fixedpoint)
    return fail(PNODE_EINVAL, "off-grid marginals need the isotropic model with a stored backward model (smoother / fixedpoint)");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, K = h->cfg.num_checkpoints, Q = size_t(num_queries), nd = size_t(h->n) * h->cfg.d;
  std::vector<OffgridArg> qa(B * Q);
  for (size_t b = 0; b < B; ++b) {
    if (num_filled[b] < 0 || size_t(num_filled[b]) > K) return fail(PNODE_EINVAL, "num_filled out of range");
    for (size_t q = 0; q < Q; ++q) {
      const size_t i = b * Q + q;
      OffgridArg& a = qa[i];
      a.t = t[i]; a.t_lo = t_lo[i]; a.t_hi = t_hi[i]; a.idx = idx[i]; a.at_grid = at_grid[i] ? 1 : 0;
      if (a.idx >= num_filled[b]) return fail(PNODE_EINVAL, "query beyond the stored posterior");
      if (a.idx >= 0 && !a.at_grid && !(a.t > a.t_lo && a.t < a.t_hi)) return fail(PNODE_EINVAL, "query outside its interval");
    }
  }
  double *dSm = nullptr, *dInit = nullptr, *dOut = nullptr;
  OffgridArg* dQ = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(dSm); (void)hipFree(dInit); (void)hipFree(dOut); (void)hipFree(dQ);
  };
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&dSm), B * (K + 1) * nd * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dInit), B * nd * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dOut), B * Q * nd * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dQ), B * Q * sizeof(OffgridArg));
  if (e == hipSuccess) e = hipMemsetAsync(dSm, 0, B * (K + 1) * nd * sizeof(double), h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(dOut, 0, B * Q * nd * sizeof(double), h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h->dFilled, num_filled, B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dInit, init_means, B * nd * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dQ, qa.data(), B * Q * sizeof(OffgridArg), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    iso_launch_smooth(h->iso, h->dFilled, dSm, h->stream);
    iso_launch_offgrid(h->iso, dQ, int(Q), dInit, dSm, h->dK, dOut, h->stream);
    e = hipMemcpyAsync(out, dOut, B * Q * nd * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipGetLastError();
  cleanup();
  if (e != hipSuccess) return fail(PNODE_EDEVICE, hipGetErrorString(e));
  return PNODE_OK;
}

int pnode_sample_checkpoints_n(pnode_batch* h, const int32_t* num_filled, int32_t num_points, int32_t num_samples, const double* eps,
                               double* out) {
  if (!h || !num_filled || !eps || !out || num_samples < 1 || num_points < 1) return fail(PNODE_EINVAL, "bad argument");
  if (h->dense || !h->iso.fixedpoint)
    return fail(PNODE_EINVAL, "posterior sampling needs the isotropic model with a stored backward model (smoother / fixedpoint)");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t B = h->cfg.batch, K = h->cfg.num_checkpoints, S = size_t(num_samples), P = size_t(num_points), nd = size_t(h->n) * h->cfg.d;
  for (size_t b = 0; b < B; ++b)
    if (num_filled[b] < 0 || size_t(num_filled[b]) > K || size_t(num_filled[b]) + 1 > P) return fail(PNODE_EINVAL, "num_filled out of range");
  const size_t cnt = B * S * P * nd;
  double *dEps = nullptr, *dOut = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&dEps), cnt * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dOut), cnt * sizeof(double));
  if (e == hipSuccess) e = hipMemsetAsync(dOut, 0, cnt * sizeof(double), h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dEps, eps, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h->dFilled, num_filled, B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    iso_launch_sample(h->iso, h->dFilled, int(S), int(P), dEps, dOut, h->stream);
    e = hipMemcpyAsync(out, dOut, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipGetLastError();
  (void)hipFree(dEps);
  (void)hipFree(dOut);
  if (e != hipSuccess) return fail(PNODE_EDEVICE, hipGetErrorString(e));
  return PNODE_OK;
}

int pnode_sample_checkpoints(pnode_batch* h, const int32_t* num_filled, int32_t num_samples, const double* eps, double* out) {
  if (!h) return fail(PNODE_EINVAL, "bad argument");
  return pnode_sample_checkpoints_n(h, num_filled, h->cfg.num_checkpoints + 1, num_samples, eps, out);
}

int pnode_last_step_ms(pnode_batch* h, float* ms_out) {
  if (!h || !ms_out) return fail(PNODE_EINVAL, "null argument");
  if (!h->step_timing) return fail(PNODE_ESTATE, "step timing is off: call pnode_set_phase_timing(h, 1) first");
  HIP_TRY(hipEventElapsedTime(ms_out, h->ev0, h->ev1));
  return PNODE_OK;
}

int pnode_set_generic_path(pnode_batch* h, int32_t enabled) {
  if (!h) return fail(PNODE_EINVAL, "null handle");
  if ((enabled == 2 || enabled == 4) && !dense_large_supported(h->dn)) return fail(PNODE_EINVAL, "the large-D path does not support this shape");
  if (h->stepped)
    return fail(PNODE_ESTATE, "the device path is chosen before the first step (the paths store the state factor in different layouts)");
  h->dn.force_generic = enabled == 1 || h->dn.fixedpoint;  // the register-resident MFMA kernels have no extra-row LQ
  h->dn.user_generic = enabled == 1;
  h->dn.force_large = enabled == 2 || enabled == 4;
  h->dn.gemm_tile64 = enabled == 4;
  h->dn.lq_two_launch = enabled == 3;
  return PNODE_OK;
}

int pnode_set_smoother_pipeline(pnode_batch* h, int32_t mode) {
  if (!h || mode < 0 || mode > 1) return fail(PNODE_EINVAL, "bad argument");
  if (h->dense) return fail(PNODE_EINVAL, "the pipelined device controller belongs to the isotropic fixed-point smoother");
  h->iso.fp_pipeline = mode;
  return PNODE_OK;
}

int pnode_debug_set_lq_spin_limit(pnode_batch* h, int32_t polls) {
  if (!h || polls < 0) return fail(PNODE_EINVAL, "bad argument");
  if (!h->dense) {  // isotropic model: the waits of the pipelined smoother controller (pnode_set_smoother_pipeline)
    h->iso.fp_spin_limit = polls;
    return PNODE_OK;
  }
  h->dn.lq_spin_limit = polls;
  return PNODE_OK;
}

int pnode_set_phase_timing(pnode_batch* h, int32_t enabled) {
  if (!h) return fail(PNODE_EINVAL, "null handle");
  h->phase_timing = enabled != 0;
  h->step_timing = enabled != 0;
  return PNODE_OK;
}

int pnode_last_phase_ms(pnode_batch* h, float* ms_out) {
  if (!h || !ms_out) return fail(PNODE_EINVAL, "null argument");
  if (!h->dense || !h->phase_timing) return fail(PNODE_ESTATE, "phase timing is available for the dense model after set_phase_timing(1)");
  const PhaseEvents& pe = h->phase_last >= 0 && size_t(h->phase_last) < h->phase_pool.size() ? h->phase_pool[size_t(h->phase_last)] : h->phase;
  for (int i = 0; i < 4; ++i) HIP_TRY(hipEventElapsedTime(ms_out + i, pe.ev[i], pe.ev[i + 1]));
  return PNODE_OK;
}

int pnode_phase_ms_mean(pnode_batch* h, float* ms_out, int64_t* attempts_out) {
  if (!h || !ms_out) return fail(PNODE_EINVAL, "null argument");
  if (!h->dense || !h->phase_timing || h->phase_count <= 0)
    return fail(PNODE_ESTATE, "needs the dense model, pnode_set_phase_timing(h, 1) and a pnode_advance_device call since");
  for (int i = 0; i < 4; ++i) ms_out[i] = float(h->phase_sum[i] / double(h->phase_count));
  if (attempts_out) *attempts_out = h->phase_count;
  return PNODE_OK;
}

void* pnode_stream(pnode_batch* h) { return h ? (void*)h->stream : nullptr; }

#ifdef PNODE_TIMING_BUILD
// Diagnostic build only (make timing -> libpnode_hip_timing.so): 64-bit words of trajectory b's S scratch (in-kernel cycle stamps).
int pnode_debug_read_scratch(pnode_batch* h, int32_t b, int32_t offset_words, int32_t count, uint64_t* out) {
  if (!h || !out || !h->dense) return fail(PNODE_EINVAL, "bad argument");
  const size_t d = h->cfg.d;
  HIP_TRY(hipMemcpy(out, reinterpret_cast<uint64_t*>(h->dn.S + size_t(b) * d * d) + offset_words, size_t(count) * 8,
                    hipMemcpyDeviceToHost));
  return PNODE_OK;
}
#endif


}  // extern "C"
