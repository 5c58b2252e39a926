// Dense fixed-point smoother (strategy_fixedpoint with impl.select("dense")): generic kernels.
//
// The reference composes the backward transition of every predict step in place so that only one
// transition per checkpoint interval is kept (/root/reference/src/odecheckpts/ivpsolvers.py:43,
// experiments/4_brusselator/run.py:59; BASELINE config 5 asks for it with the dense model). In
// square-root form everything comes out of ONE extra-row LQ: the time-update stack [T0 | Mt] is
// extended by the never-pivoted rows [0 | P^-1 L]; after the row-Householder sweep
//      rows 0..D-1   : [L^-_p | 0]                 (the predicted factor, as in the filter)
//      rows D..2D-1  : [C21_p | Lbw_p]             cross factor and backward-noise factor
// so that  G_p = C21_p (L^-_p)^-1,  Lambda_p = Lbw_p Lbw_p^T  (oracle: `_extrapolate_cov`, revert_conditional).
// Merging with the running transition (G0, xi0, Lam0) is  G0 G,  G0 xi + xi0,  lq([Lam0 | G0 Lam]).
// One workgroup (256 threads) per trajectory, matrices in global memory: this is the functional path
// for the fixed-point dense model (parity with the oracle); it is not performance-tuned.
#include "pnode_device.hpp"

namespace pnode {

constexpr int kT = 256;

__device__ __forceinline__ void fp_precond(int n, double dt, const IwpConst& K, double* p, double* pinv) {
  const double adt = fabs(dt);
  double pw = sqrt(adt);
  for (int i = n - 1; i >= 0; --i) {
    p[i] = pw / K.fact[i];
    pinv[i] = K.fact[i] / pw;
    pw *= adt;
  }
}

// Build the 2D x D stacks for one predict operation (interpolation path; the step path uses
// dense_lq_input_kernel). Optionally extrapolates the mean into slot dst and v.mext.
__global__ __launch_bounds__(kT) void dfp_prep_kernel(DenseView v, const FpArg* __restrict__ fa, const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.y;
  const FpArg a = fa[b];
  if (!a.active) {
    if (blockIdx.x == 0 && threadIdx.x == 0) v.act[b] = 0;
    return;
  }
  const IwpConst& K = *Kc;
  const int d = v.d, n = v.n, D = v.D;
  double p[kMaxN], pinv[kMaxN];
  fp_precond(n, a.dt, K, p, pinv);
  const double sigma = v.sig_new[b];
  const size_t sB = size_t(v.B);
  const double* L = v.U + (size_t(a.src) * sB + b) * D * D;
  double* Mt = v.M + size_t(b) * 2 * D * D;
  double* T = v.R + size_t(b) * 2 * D * D;
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) v.act[b] = 1;
    const double* ms = v.mean + (size_t(a.src) * sB + b) * D;
    double* md = v.mean + (size_t(a.dst) * sB + b) * D;
    for (int j = threadIdx.x; j < d; j += kT)
      for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * pinv[kk] * ms[kk * d + j];
        const double e = p[i] * acc;
        md[i * d + j] = e;
        v.mext[size_t(b) * D + i * d + j] = e;
      }
  }
  for (int r = blockIdx.x; r < D; r += gridDim.x) {
    const int i = r / d, q = r % d;
    for (int c = threadIdx.x; c < D; c += kT) {
      double acc = 0.0;
      for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * pinv[kk] * L[size_t(kk * d + q) * D + c];
      Mt[size_t(r) * D + c] = acc;
      const int j = c / d, q2 = c % d;
      T[size_t(r) * D + c] = (q2 == q && j <= i) ? sigma * K.qsT[j][i] : 0.0;
      Mt[size_t(D + r) * D + c] = pinv[i] * L[size_t(r) * D + c];
      T[size_t(D + r) * D + c] = 0.0;
    }
  }
}

// U[dst] = diag(p) L^-_p  (interpolation path: the predicted state is not corrected)
__global__ __launch_bounds__(kT) void dfp_state_kernel(DenseView v, const FpArg* __restrict__ fa, const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.y;
  const FpArg a = fa[b];
  if (!a.active) return;
  const int d = v.d, n = v.n, D = v.D;
  double p[kMaxN], pinv[kMaxN];
  fp_precond(n, a.dt, *Kc, p, pinv);
  const double* T = v.R + size_t(b) * 2 * D * D;
  double* U = v.U + (size_t(a.dst) * v.B + b) * D * D;
  for (int r = blockIdx.x; r < D; r += gridDim.x)
    for (int c = threadIdx.x; c < D; c += kT) U[size_t(r) * D + c] = (c <= r) ? p[r / d] * T[size_t(r) * D + c] : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) v.sigma[size_t(a.dst) * v.B + b] = v.sig_new[b];
}

// Backward transition of the predict just factorised + merge with the running one. Leaves the merge
// stack [T := Lam0 | Mt := G0 Lam_raw] in rows 0..D-1 of R / M for the following D-row LQ.
__global__ __launch_bounds__(kT) void dfp_backward_kernel(DenseView v, const FpArg* __restrict__ fa, const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.x;
  const FpArg a = fa[b];
  if (!a.active) return;
  const int d = v.d, n = v.n, D = v.D;
  double p[kMaxN], pinv[kMaxN];
  fp_precond(n, a.dt, *Kc, p, pinv);
  const size_t sB = size_t(v.B);
  double* M = v.M + size_t(b) * 2 * D * D;
  double* R = v.R + size_t(b) * 2 * D * D;
  const double* Lm = R;                    // L^-_p (lower), rows 0..D-1
  double* G = R + size_t(D) * D;           // C21_p -> G_p -> G (plain coordinates), rows D..2D-1
  double* Lam = M + size_t(D) * D;         // Lbw_p -> Lam_raw (plain coordinates)
  // (a) G_p = C21_p (L^-_p)^-1 : one thread per row, backward over the columns (pseudo-inverse on a zero pivot)
  for (int r = threadIdx.x; r < D; r += kT) {
    double* x = G + size_t(r) * D;
    for (int j = D - 1; j >= 0; --j) {
      double acc = x[j];
      for (int k = j + 1; k < D; ++k) acc -= x[k] * Lm[size_t(k) * D + j];
      const double piv = Lm[size_t(j) * D + j];
      x[j] = (piv != 0.0) ? acc / piv : 0.0;
    }
  }
  __syncthreads();
  // (b) back to plain coordinates
  for (int idx = threadIdx.x; idx < D * D; idx += kT) {
    const int r = idx / D, c = idx % D;
    G[idx] *= p[r / d] * pinv[c / d];
    Lam[idx] *= p[r / d];
  }
  __syncthreads();
  // (c) xi = m_prev - G m^-   (kept in the z scratch? no: in mext's slot of this trajectory after use)
  const double* mprev = v.mean + (size_t(a.src) * sB + b) * D;
  const double* mext = v.mext + size_t(b) * D;
  double* xi_dst = v.cxi + (size_t(a.dst) * sB + b) * D;
  double* G_dst = v.cG + (size_t(a.dst) * sB + b) * D * D;
  extern __shared__ double xi_s[];  // [D]
  for (int r = threadIdx.x; r < D; r += kT) {
    double acc = mprev[r];
    for (int c = 0; c < D; ++c) acc -= G[size_t(r) * D + c] * mext[c];
    xi_s[r] = acc;
  }
  __syncthreads();
  // (d) merge
  if (a.merge) {
    const double* G0 = v.cG + (size_t(a.src) * sB + b) * D * D;
    const double* xi0 = v.cxi + (size_t(a.src) * sB + b) * D;
    const double* Lam0 = v.cLam + (size_t(a.src) * sB + b) * D * D;
    for (int idx = threadIdx.x; idx < D * D; idx += kT) {
      const int r = idx / D, c = idx % D;
      double g = 0.0, l = 0.0;
      for (int k = 0; k < D; ++k) {
        const double g0 = G0[size_t(r) * D + k];
        g += g0 * G[size_t(k) * D + c];
        l += g0 * Lam[size_t(k) * D + c];
      }
      G_dst[idx] = g;
      M[idx] = l;             // Mt := G0 Lam_raw
      R[idx] = Lam0[idx];     // T  := Lam0 (lower triangular)
    }
    for (int r = threadIdx.x; r < D; r += kT) {
      double acc = xi0[r];
      for (int k = 0; k < D; ++k) acc += G0[size_t(r) * D + k] * xi_s[k];
      xi_dst[r] = acc;
    }
  } else {
    for (int idx = threadIdx.x; idx < D * D; idx += kT) {
      G_dst[idx] = G[idx];
      M[idx] = Lam[idx];
      R[idx] = 0.0;
    }
    for (int r = threadIdx.x; r < D; r += kT) xi_dst[r] = xi_s[r];
  }
}

__global__ __launch_bounds__(kT) void dfp_store_lam_kernel(DenseView v, const FpArg* __restrict__ fa) {
  const int b = blockIdx.y;
  const FpArg a = fa[b];
  if (!a.active) return;
  const int D = v.D;
  const double* R = v.R + size_t(b) * 2 * D * D;
  double* Lam = v.cLam + (size_t(a.dst) * v.B + b) * D * D;
  for (int r = blockIdx.x; r < D; r += gridDim.x)
    for (int c = threadIdx.x; c < D; c += kT) Lam[size_t(r) * D + c] = (c <= r) ? R[size_t(r) * D + c] : 0.0;
}

__global__ void dfp_set_sigma_kernel(DenseView v, const InterpArg* __restrict__ ia) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= v.B) return;
  if (ia[b].ckpt >= 0 && !ia[b].at_step) v.sig_new[b] = v.sigma[size_t(ia[b].s1) * v.B + b];
}

// Checkpoint bookkeeping after the two predicts of an interpolation (tmpA = slot 3: state at t with the merged
// transition; tmpB = slot 4: transition t1 -> t), or for a checkpoint that coincides with STEP_FROM.
// Grid (G, B): the D x D copies are spread over the G workgroups of a trajectory; block 0 also moves the vectors.
__global__ __launch_bounds__(kT) void dfp_interp_finish_kernel(DenseView v, const InterpArg* __restrict__ ia) {
  const int b = blockIdx.y;
  const InterpArg a = ia[b];
  if (a.ckpt < 0) return;
  const int D = v.D;
  const size_t sB = size_t(v.B), DD = size_t(D) * D;
  const size_t ck = size_t(b) * v.K + a.ckpt;
  double* cm = v.ck_mean + ck * D;
  double* cG = v.ck_G + ck * DD;
  double* cx = v.ck_xi + ck * D;
  double* cL = v.ck_Lam + ck * DD;
  auto slotG = [&](int s) { return v.cG + (size_t(s) * sB + b) * DD; };
  auto slotL = [&](int s) { return v.cLam + (size_t(s) * sB + b) * DD; };
  auto slotX = [&](int s) { return v.cxi + (size_t(s) * sB + b) * D; };
  auto slotM = [&](int s) { return v.mean + (size_t(s) * sB + b) * D; };
  auto slotU = [&](int s) { return v.U + (size_t(s) * sB + b) * DD; };
  const size_t gtid = size_t(blockIdx.x) * kT + threadIdx.x, gstride = size_t(gridDim.x) * kT;
  const int lane = threadIdx.x & 63, gw = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), nw = gridDim.x * (kT / 64);
  if (a.at_step) {
    for (size_t r = gtid; r < size_t(D); r += gstride) {
      cm[r] = slotM(a.s1)[r];
      cx[r] = slotX(a.s1)[r];
      slotX(a.s1)[r] = 0.0;
    }
    for (size_t idx = gtid; idx < DD; idx += gstride) {
      cG[idx] = slotG(a.s1)[idx];
      cL[idx] = slotL(a.s1)[idx];
      slotG(a.s1)[idx] = (idx / D == idx % D) ? 1.0 : 0.0;
      slotL(a.s1)[idx] = 0.0;
    }
  } else {
    const int tA = 3, tB = 4;
    const double* G1 = slotG(tB);
    const double* m1 = slotM(a.s1);
    for (int r = gw; r < D; r += nw) {  // solution at the checkpoint: G1 m(t1) + xi1 (one wavefront per row)
      double acc = 0.0;
      for (int c = lane; c < D; c += 64) acc = fma(G1[size_t(r) * D + c], m1[c], acc);
      acc = wave_sum(acc);
      if (lane == 0) cm[r] = acc + slotX(tB)[r];
    }
    for (size_t r = gtid; r < size_t(D); r += gstride) {
      cx[r] = slotX(tA)[r];
      slotX(a.s1)[r] = slotX(tB)[r];     // STEP_FROM remembers how to get back to t
      slotM(a.s0)[r] = slotM(tA)[r];     // INTERP_FROM <- filtering state at t ...
      slotX(a.s0)[r] = 0.0;              //   ... with an identity transition
    }
    for (size_t idx = gtid; idx < DD; idx += gstride) {
      cG[idx] = slotG(tA)[idx];
      cL[idx] = slotL(tA)[idx];
      slotG(a.s1)[idx] = slotG(tB)[idx];
      slotL(a.s1)[idx] = slotL(tB)[idx];
      slotU(a.s0)[idx] = slotU(tA)[idx];
      slotG(a.s0)[idx] = (idx / D == idx % D) ? 1.0 : 0.0;
      slotL(a.s0)[idx] = 0.0;
    }
    if (gtid == 0) v.sigma[size_t(a.s0) * sB + b] = v.sig_new[b];
  }
  if (gtid == 0) v.ck_steps[ck] = a.steps;
}

// Backward pass over the stored checkpoint transitions: out [B][K+1][D]. One wavefront per row of G.
__global__ __launch_bounds__(kT) void dfp_smooth_kernel(DenseView v, const int32_t* __restrict__ num_filled, double* __restrict__ out) {
  extern __shared__ double mbuf[];  // [2][D]
  const int b = blockIdx.x;
  const int nf = num_filled[b];
  if (nf <= 0) return;
  const int D = v.D;
  double* cur = mbuf;
  double* nxt = mbuf + D;
  double* ob = out + size_t(b) * (v.K + 1) * D;
  const double* last = v.ck_mean + (size_t(b) * v.K + (nf - 1)) * D;
  for (int r = threadIdx.x; r < D; r += kT) {
    cur[r] = last[r];
    ob[size_t(nf) * D + r] = last[r];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = nf - 1; c >= 0; --c) {
    const double* G = v.ck_G + (size_t(b) * v.K + c) * D * D;
    const double* xi = v.ck_xi + (size_t(b) * v.K + c) * D;
    for (int r = wv; r < D; r += kT / 64) {
      double acc = 0.0;
      for (int k = lane; k < D; k += 64) acc = fma(G[size_t(r) * D + k], cur[k], acc);
      acc = wave_sum(acc);
      if (lane == 0) {
        nxt[r] = acc + xi[r];
        ob[size_t(c) * D + r] = acc + xi[r];
      }
    }
    __syncthreads();
    double* t = cur;
    cur = nxt;
    nxt = t;
  }
}

static void fp_backward_chain(const DenseView& v, const FpArg* fa, const IwpConst* K, hipStream_t s) {
  if (dense_use_large(v)) {
    dense_large_fp_backward(v, fa, K, s);  // batched GEMM / blocked TRSM + blocked LQ (pnode_dense_large.hip)
  } else {
    const size_t shm = size_t(v.D) * sizeof(double);
    hipLaunchKernelGGL(dfp_backward_kernel, dim3(v.B), dim3(kT), shm, s, v, fa, K);
    dense_launch_lq_generic(v, v.D, s);  // lq([Lam0 | G0 Lam_raw]) -> merged noise factor
  }
  hipLaunchKernelGGL(dfp_store_lam_kernel, dim3(v.D < 64 ? v.D : 64, v.B), dim3(kT), 0, s, v, fa);
}

// Step path: dense_launch_step (begin, 2D-row LQ, correct) has run; add the backward transition + merge.
void dense_fp_launch_step_tail(const DenseView& v, const FpArg* fargs, const IwpConst* K, hipStream_t s) {
  fp_backward_chain(v, fargs, K, s);
}

void dense_fp_launch_interp(const DenseView& v, const InterpArg* iargs, const FpArg* op1, const FpArg* op2, const IwpConst* K,
                            hipStream_t s) {
  const int rows = v.D < 64 ? v.D : 64;
  hipLaunchKernelGGL(dfp_set_sigma_kernel, dim3((v.B + 63) / 64), dim3(64), 0, s, v, iargs);
  for (const FpArg* op : {op1, op2}) {
    hipLaunchKernelGGL(dfp_prep_kernel, dim3(rows, v.B), dim3(kT), 0, s, v, op, K);
    if (dense_use_large(v))
      dense_large_lq(v, 2 * v.D, s);
    else
      dense_launch_lq_generic(v, 2 * v.D, s);
    hipLaunchKernelGGL(dfp_state_kernel, dim3(rows, v.B), dim3(kT), 0, s, v, op, K);
    fp_backward_chain(v, op, K, s);
  }
  hipLaunchKernelGGL(dfp_interp_finish_kernel, dim3(v.D >= 256 ? 32 : 4, v.B), dim3(kT), 0, s, v, iargs);
}

void dense_fp_launch_smooth(const DenseView& v, const int32_t* num_filled, double* out, hipStream_t s) {
  hipLaunchKernelGGL(dfp_smooth_kernel, dim3(v.B), dim3(kT), size_t(2) * v.D * sizeof(double), s, v, num_filled, out);
}

}  // namespace pnode
