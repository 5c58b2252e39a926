// Device views + launcher prototypes shared by the kernel translation units and the C-ABI layer.
#pragma once
#include "pnode_common.hpp"

namespace pnode {

// Isotropic model: device pointers for one batch (see pnode_api.hip for the allocation).
struct IsoView {
  int B, d, k, n, K;  // batch, ODE dim, ode order, nu+1, checkpoint capacity
  int fixedpoint, dynamic, vf_id, P, prm_per_traj;
  int cus;          // compute units of the device
  int fp_pipeline;  // device-resident controller of the fixed-point smoother: 0 one wavefront per trajectory, 1 pipelined (workgroup per trajectory)
  int fp_spin_limit;  // > 0: bound (in polls) of the waits of the pipelined kernel instead of its default (pnode_debug_set_lq_spin_limit)
  int32_t* fault;   // pinned host words [4]: a hand-shake of the pipelined kernel that ran into its bound (fault[0] = 2, fault[1] = trajectory, fault[3] = wait)
  double* mean;   // [3][B][n][d]
  double* fac;    // [3][B][n][n]   Sigma = fac^T fac
  double* cG;     // [3][B][n][n]   running backward transition (fixedpoint)
  double* cxi;    // [3][B][n][d]
  double* cLam;   // [3][B][n][n]   upper triangular, Lambda = cLam^T cLam
  double* sigma;  // [3][B]
  const double* atol;  // [B]
  const double* rtol;  // [B]
  const double* prm;   // [P] or [B][P]
  const double* user_f;  // PNODE_VF_USER: [B][d]
  const double* user_J;  // PNODE_VF_USER: [B][d][k*d]
  double* errnorm;     // [B]
  double* ck_mean;     // [B][K][n][d]
  double* ck_fac;      // [B][K][n][n]
  double* ck_G;        // [B][K][n][n]
  double* ck_xi;       // [B][K][n][d]
  double* ck_Lam;      // [B][K][n][n]
  int64_t* ck_steps;   // [B][K]
  double* ck_sigma;    // [B][K]   output scale the stored state was computed with
};

void iso_launch_step(const IsoView& v, const StepArg* args, const IwpConst* K, hipStream_t s);
struct DenseView;
bool dense_d1_fixed_grid_supported(const DenseView& v);
void dense_d1_launch_fixed_grid(const DenseView& v, const StepArg* args, const IwpConst* K, double* out, int G, hipStream_t s);
void iso_launch_interp(const IsoView& v, const InterpArg* args, const IwpConst* K, hipStream_t s);
void iso_launch_advance(const IsoView& v, TrajCtl* ctl, const IwpConst* K, const ControlParams& cp, hipStream_t s,
                        const CkptPlan& plan = CkptPlan{nullptr, 0, 0, nullptr, nullptr});
bool iso_advance_walks_checkpoints(const IsoView& v);  // the kernel that would run takes a CkptPlan (all but the pipelined smoother kernel)
void iso_launch_smooth(const IsoView& v, const int32_t* num_filled, double* out, hipStream_t s);
void iso_launch_offgrid(const IsoView& v, const OffgridArg* qs, int Q, const double* init, const double* smoothed, const IwpConst* K,
                        double* out, hipStream_t s);
void iso_launch_sample(const IsoView& v, const int32_t* num_filled, int S, int P, const double* eps, double* out, hipStream_t s);

// Dense model (EKF1 / EKF0 with a full D x D factor), D = n*d.
struct DenseView {
  int B, d, k, n, D, K;
  int ts1, dynamic, vf_id, P, prm_per_traj;
  int force_generic;  // 1: never take the MFMA fast path (A/B testing)
  int lq_spin_limit;  // > 0: bound (in polls) of every wait of the one-launch LQ, instead of its default (pnode_debug_set_lq_spin_limit)
  int32_t* fault;     // [4] pinned host memory: {raised, trajectory, awaited value, 8 * kind + wavefront} of a hand-shake that timed out
  int lq_two_launch;  // 1: LQ of the MFMA fast path by the two-launch, 32-row-pass kernels (pnode_dense_mfma.hip) even where the
                      // one-launch kernel with overlapping 64-row passes (pnode_dense_lq4.hip) supports the shape (A/B testing)
  double* mean;   // [3][B][n][d]
  double* U;      // [3][B][D][D]   Sigma = U^T U (row-major)
  double* sigma;  // [3][B]
  const double* atol;
  const double* rtol;
  const double* prm;
  const double* user_f;  // PNODE_VF_USER: [B][d]
  const double* user_J;  // PNODE_VF_USER: [B][d][k*d]
  double* errnorm;  // [B]
  double* ck_mean;  // [B][K][n][d]
  double* ck_fac;   // [B][K][D][D] L at the save_at points of the filter (row-major, Sigma = L L^T); null unless requested
  int u_tiled_now;  // set by the checkpoint-factor launcher only: v.U is in the strip order of the MFMA path
  int64_t* ck_steps;
  // per-attempt scratch
  double* M;     // [B][D][D]  QR input, (A L)^T in preconditioned coordinates
  double* R;     // [B][D][D]  QR output (upper), preconditioned coordinates
  double* J;     // [B][d][k*d] Jacobian blocks
  double* z;     // [B][d]     residual
  double* err;   // [B][d]     dt * sigma_hat * std (unnormalised error estimate)
  double* Ft;    // [B][D][d]  F^T = U^- H^T, later W^T
  double* X;     // [B][d][D]  W U^-
  double* S;     // [B][d][d]
  double* sig_new;  // [B]
  // fixed-point smoother (dense): slots 0..2 rotate, 3..4 are interpolation temporaries
  int fixedpoint, nslots, mr_rows;  // mr_rows: M and R hold mr_rows * D rows per trajectory (2 with fixedpoint)
  int32_t* act;     // [B] activity flags of the running batched operation (written by the input kernels)
  double* mext;     // [B][D] extrapolated mean of the running operation
  double* cG;       // [nslots][B][D][D] backward transition  x_prev | x ~ N(G x + xi, Lam Lam^T)
  double* cxi;      // [nslots][B][D]
  double* cLam;     // [nslots][B][D][D] lower triangular
  double* ck_G;     // [B][K][D][D]
  double* ck_xi;    // [B][K][D]
  double* ck_Lam;   // [B][K][D][D]
  // large-D path (pnode_dense_large.hip): panel scratch of the blocked LQ and per-trajectory output offsets
  int force_large;  // 1: take the large-D path whenever it supports the shape (A/B testing at small D)
  int gemm_tile64;  // 1: the batched GEMMs of the large-D path use the 64 x 64 tile only (pnode_set_generic_path(h, 4): A/B partner of the 128 x 128 tile)
  double* lgG;      // [B][mr_rows*D][128]
  double* lgW;      // [B][mr_rows*D][128]
  double* lgTf;     // [B][4][32][32]  panel factors of the current outer block
  double* lgGram;   // [B][128][128]
  double* lgTfo;    // [B][128][128]   aggregated factor of the outer block
  long* lgOff;      // [B]
  int user_generic;      // pnode_set_generic_path(h, 1): the caller asked for the generic kernels
  int32_t* lgMaskMerge;  // [B] fixed-point: trajectories whose running transition is merged in this operation
  long* lgOffS;          // [B] element offsets of the source / destination slots of the running operation
  long* lgOffD;          // [B]
  double* lgXi;          // [B][D]
};

// One-launch LQ of the MFMA fast path (pnode_dense_lq4.hip)
bool dense_lq4_supported(const DenseView& v);
void dense_launch_lq4(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s);

// Optional per-phase timing of one dense step: events[0..4] bracket begin | lq_input | lq | correct.
struct PhaseEvents {
  hipEvent_t ev[5];
};
// Device-resident controller fused into the step (MFMA path): the measurement-update kernel takes the controller's decision.
struct FusedControl {
  TrajCtl* ctl;
  StepArg* args;  // device memory, read and written
  ControlParams cp;
};
bool dense_fused_control_supported(const DenseView& v);
void dense_launch_step(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s,
                       const PhaseEvents* ev = nullptr, const FusedControl* fc = nullptr);
bool dense_mfma_supported(const DenseView& v);
bool dense_state_factor_tiled(const DenseView& v);  // L in the register order of the strips (MFMA path) instead of row-major
void dense_launch_mfma_begin(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s);
void dense_launch_mfma_tail(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s,
                            const PhaseEvents* ev, const FusedControl* fc = nullptr);
void dense_launch_interp(const DenseView& v, const InterpArg* args, const IwpConst* K, hipStream_t s);
void dense_launch_checkpoint_factors(const DenseView& v, const InterpArg* args, const StepArg* pargs, const IwpConst* K, hipStream_t s);
// dense fixed-point smoother (generic kernels, pnode_dense_fp.hip)
void dense_fp_launch_step_tail(const DenseView& v, const FpArg* fargs, const IwpConst* K, hipStream_t s);
void dense_fp_launch_interp(const DenseView& v, const InterpArg* iargs, const FpArg* op1, const FpArg* op2, const IwpConst* K,
                            hipStream_t s);
void dense_fp_launch_smooth(const DenseView& v, const int32_t* num_filled, double* out, hipStream_t s);
void dense_launch_lq_generic(const DenseView& v, int rows_total, hipStream_t s);
void dense_launch_lq_input(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s);
// large-D path (pnode_dense_large.hip): batched MFMA GEMMs + blocked LQ / Cholesky / triangular solve through HBM
bool dense_large_supported(const DenseView& v);
bool dense_use_large(const DenseView& v);  // the large-D path is the one that runs for this handle
void dense_large_lq(const DenseView& v, int rows_total, hipStream_t s);
void dense_large_fp_backward(const DenseView& v, const FpArg* fa, const IwpConst* K, hipStream_t s);
void dense_large_launch_step(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s, const PhaseEvents* ev);
// small-D fused path (pnode_dense_small.hip): one wavefront per trajectory, one kernel per attempt
bool dense_small_supported(const DenseView& v);
void dense_small_launch_step(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s);
void dense_small_launch_advance(const DenseView& v, TrajCtl* ctl, const IwpConst* K, const ControlParams& cp, hipStream_t s);

}  // namespace pnode
