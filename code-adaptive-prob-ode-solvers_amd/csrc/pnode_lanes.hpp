// n x n algebra of the isotropic model (and of the dense model of a scalar ODE, pnode_dense_small.hip) SPREAD OVER THE LANES of a
// wavefront (r04; replaces the per-lane-redundant form of rounds 1-3, `git show a465765:code-adaptive-prob-ode-solvers_amd/csrc/pnode_nxn.hpp`,
// which needed n^2 registers per matrix in every lane: 256 VGPRs + AGPR spills from nu = 4 on,
// scratch from nu = 7 on -- and one of those spills, placed by the compiler in front of the exec restore of a `lane < d` region,
// was the memory access fault of the looped smoother kernel).
//
// Layout: l = min(lane & 15, n - 1) (col_index) is a matrix index: the lanes of a 16-lane DPP row beyond n mirror lane n - 1, and the
// four DPP rows of a wavefront hold four identical copies, so a row broadcast is a broadcast to the whole wavefront, every lane
// holds valid data, and a matrix is stored by ALL lanes without a predicate (the copies write the same value to the same address).
//   "col-lane"  A[r] of lane l = element (r, l): lane l holds COLUMN l    (triangular factors: R, Lambda-factor)
//   "row-lane"  A[c] of lane l = element (l, c): lane l holds ROW l       (backward gains G)
// One matrix costs n doubles per lane. A uniform scalar is a register whose lanes agree; lane j's value becomes uniform by
// bcast<j>: ONE instruction, v_mov_b64_dpp row_newbcast:j (the only 64-bit DPP control of gfx90a and later; emitted and
// hazard-checked by the compiler from __builtin_amdgcn_update_dpp).
//
// RULE: a DPP broadcast reads its source lane's register only when that lane is enabled. Every function here must be called with all 64
// lanes enabled; the kernels keep every lane enabled throughout (indices are clamped, duplicate lanes repeat their neighbour's
// work and its stores) instead of looping over `lane < d`: no exec-masked region is left in which the compiler could misplace a
// spill or a copy.
// The copies must stay IDENTICAL bit for bit: whatever enters the n x n algebra from the ODE dimensions (sums over the wavefront)
// has to be the same in every lane (wave_sum of pnode_common.hpp; ln::uniform).
//
// Orthogonal triangularisations are Householder reflections (one reciprocal square root per COLUMN, the Givens rotations of rounds 1-3
// took one per ELEMENT), rows negated where needed so that the triangular factor has a non-negative diagonal like the one the
// rotations gave. Follows the reference call sites below (arithmetic re-derived, oracle/pn_oracle.py is the
// CPU restatement): /root/reference/src/odecheckpts/ivpsolvers.py:37-50.
#pragma once
#include <utility>

#include "pnode_device.hpp"

namespace pnode {
namespace ln {

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    sfor<B + 1, E>(f);
  }
}
#define PNODE_LAMBDA_INLINE __attribute__((always_inline))

// value of lane J of my 16-lane row, in every lane of the row
template <int J>
__device__ __forceinline__ double bcast(double x) {
  static_assert(J >= 0 && J < 16, "row broadcast");
  const long long v = __builtin_bit_cast(long long, x);
  // (through a 64-bit variable: inside a template the builtin's type is only settled at instantiation)
  const long long r = __builtin_amdgcn_update_dpp(0LL, v, 0x150 + J, 0xf, 0xf, true);
  return __builtin_bit_cast(double, r);
}
// value of lane `src` (wave-uniform, run-time) as a scalar
__device__ __forceinline__ double lane_value(double x, int src) {
  const long long v = __builtin_bit_cast(long long, x);
  const unsigned lo = unsigned(__builtin_amdgcn_readlane(int(v), src));
  const unsigned hi = unsigned(__builtin_amdgcn_readlane(int(v >> 32), src));
  return __builtin_bit_cast(double, (long long)((unsigned long long)lo | ((unsigned long long)hi << 32)));
}
// a value all lanes agree on, as a scalar the compiler knows to be uniform (branches on it are scalar branches)
__device__ __forceinline__ double uniform(double x) {
  const long long v = __builtin_bit_cast(long long, x);
  const unsigned lo = unsigned(__builtin_amdgcn_readfirstlane(int(v)));
  const unsigned hi = unsigned(__builtin_amdgcn_readfirstlane(int(v >> 32)));
  return __builtin_bit_cast(double, (long long)((unsigned long long)lo | ((unsigned long long)hi << 32)));
}

// ---------------------------------------------------------------------------------------------
// Loads / stores of row-major n x n matrices (leading dimension LD), l = col_index<N>(lane)
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ int col_index(int lane) {
  const int l = lane & 15;
  return l < N ? l : N - 1;
}
template <int N, int LD = N>
__device__ __forceinline__ void load_cols(const double* __restrict__ src, double (&A)[N], int l) {
#pragma unroll
  for (int r = 0; r < N; ++r) A[r] = src[r * LD + l];
}
template <int N, int LD = N>
__device__ __forceinline__ void load_rows(const double* __restrict__ src, double (&A)[N], int l) {
#pragma unroll
  for (int c = 0; c < N; ++c) A[c] = src[l * LD + c];
}
template <int N>
__device__ __forceinline__ void store_cols(double* __restrict__ dst, const double (&A)[N], int l) {
#pragma unroll
  for (int r = 0; r < N; ++r) dst[r * N + l] = A[r];
}
template <int N>
__device__ __forceinline__ void store_rows(double* __restrict__ dst, const double (&A)[N], int l) {
#pragma unroll
  for (int c = 0; c < N; ++c) dst[l * N + c] = A[c];
}

// Ties the order of two computations without an instruction: what is derived from `b` afterwards cannot be scheduled ahead of
// `a`. Used between the rows of the small matrix products, whose n^2 independent broadcasts the scheduler would otherwise issue
// up front and keep in registers.
__device__ __forceinline__ void after(double& a, double& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// Inside a loop over the ODE dimensions: hides a loop-invariant matrix from the optimiser, which would otherwise hoist the n^2
// row broadcasts of a matrix-vector product out of the loop and keep them all in registers (n = 9: 162 VGPRs per matrix --
// the register pressure this layout exists to avoid). No instruction is emitted.
template <int N>
__device__ __forceinline__ void opaque(double (&A)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(A[i]));
}

// Per-lane constants of the prior: arow[k] = a[l][k] (row l of Abar), qcol[r] = qsT[r][l] (column l of chol(Qbar)^T).
template <int N>
__device__ __forceinline__ void load_prior(const IwpConst* __restrict__ K, double (&arow)[N], double (&qcol)[N], int l) {
  load_rows<N, kMaxN>(&K->a[0][0], arow, l);
  load_cols<N, kMaxN>(&K->qsT[0][0], qcol, l);
}

// ---------------------------------------------------------------------------------------------
// Householder step on column J of a col-lane stack: `top` is row J of the (upper triangular) top block, B[C0 .. NB - 1] the
// active bottom rows. Afterwards `top` is row J of R (zero left of the diagonal, diagonal >= 0), column J of the bottom rows is
// annihilated (what is left there is rounding-level and never read again), the payload (ptop, PB) has gone through the same
// reflection.
// ---------------------------------------------------------------------------------------------
template <int J, int C0, int NB, bool PAY>
__device__ __forceinline__ void hh_step(double& top, double (&B)[NB], double& ptop, double (&PB)[NB], int l) {
  double vb[NB];
  double s2 = 0.0;
#pragma unroll
  for (int c = C0; c < NB; ++c) {
    vb[c] = bcast<J>(B[c]);
    s2 = fma(vb[c], vb[c], s2);
  }
  const double tb = bcast<J>(top);
  const bool nz = s2 != 0.0;  // nothing below the diagonal: no reflection (the row only gets a non-negative diagonal)
  const double h2 = fma(tb, tb, s2);
  const double inv = fast_rsqrt(nz ? h2 : 1.0);
  const double beta = h2 * inv;           // || (tb, vb) ||
  const double sg = copysign(1.0, tb);
  const double s = sg * beta;
  const double v0 = tb + s;               // v = (v0, vb), H = I - v v^T / (s v0), H (tb, vb) = (-s, 0)
  const double tinv = nz ? fast_rcp(s * v0) : 0.0;
  const double fl = nz ? -sg : sg;        // the row is negated where the reflection leaves a negative diagonal
  const double fv = fl * v0;
  {
    double w = v0 * top;
#pragma unroll
    for (int c = C0; c < NB; ++c) w = fma(vb[c], B[c], w);
    w *= tinv;
    double row = fma(-fv, w, fl * top);
    row = (l < J) ? 0.0 : row;
    row = (l == J && nz) ? beta : row;
    top = row;
#pragma unroll
    for (int c = C0; c < NB; ++c) B[c] = fma(-vb[c], w, B[c]);
  }
  if constexpr (PAY) {
    double w = v0 * ptop;
#pragma unroll
    for (int c = C0; c < NB; ++c) w = fma(vb[c], PB[c], w);
    w *= tinv;
    ptop = fma(-fv, w, fl * ptop);
#pragma unroll
    for (int c = C0; c < NB; ++c) PB[c] = fma(-vb[c], w, PB[c]);
  }
}

// R of the stack [T ; B] (T upper triangular n x n, B n x n, col-lane), in T; with PAY the payload [PT ; PB] becomes Q^T [PT ; PB].
template <int N, bool PAY>
__device__ __forceinline__ void qr_stack(double (&T)[N], double (&B)[N], double (&PT)[N], double (&PB)[N], int l) {
  sfor<0, N>([&](auto jc) PNODE_LAMBDA_INLINE {
    constexpr int j = decltype(jc)::value;
    hh_step<j, 0, N, PAY>(T[j], B, PT[j], PB, l);
  });
}
// R of the square Y (col-lane), in place (upper triangular, diagonal >= 0)
template <int N>
__device__ __forceinline__ void qr_square(double (&Y)[N], int l) {
  sfor<0, N>([&](auto jc) PNODE_LAMBDA_INLINE {
    constexpr int j = decltype(jc)::value;
    double unused = 0.0;
    hh_step<j, j + 1, N, false>(Y[j], Y, unused, Y, l);
  });
}

// ---------------------------------------------------------------------------------------------
// Time update in preconditioned coordinates.
//   in : Rprev (col-lane; Sigma_prev = Rprev^T Rprev), uniform p / pinv for this dt and my lane's entries pl / pil, sigma
//   out: Rm (col-lane, upper triangular, Sigma^- = Rm^T Rm); with COND the backward transition  x_prev | x_new ~ N(G x_new + xi,
//        RL^T RL): G row-lane, RL col-lane (upper triangular)
// ---------------------------------------------------------------------------------------------
template <int N, bool COND>
__device__ __forceinline__ void time_update(const double (&arow)[N], const double (&qcol)[N], const double (&Rprev)[N],
                                            const double (&pinv)[N], double pl, double pil, double sigma, int l, double (&Rm)[N],
                                            double (&G)[N], double (&RL)[N]) {
  double Bm[N], M[N], T[N];  // Bm = Rprev diag(pinv); M = Bm Abar^T = (Abar Ltilde)^T
#pragma unroll
  for (int c = 0; c < N; ++c) Bm[c] = Rprev[c] * pil;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    double acc = 0.0;
    sfor<0, N>([&](auto kc) PNODE_LAMBDA_INLINE {
      constexpr int k = decltype(kc)::value;
      acc += bcast<k>(Bm[c]) * arow[k];
    });
    M[c] = acc;
    if (c + 1 < N) after(M[c], Bm[c + 1]);
  }
#pragma unroll
  for (int r = 0; r < N; ++r) T[r] = sigma * qcol[r];
  if constexpr (!COND) {
    double PT[N], PB[N];
    qr_stack<N, false>(T, M, PT, PB, l);
  } else {
    // ONE triangularisation of the stack [sigma chol(Qbar)^T 0 ; M Bm] = Q [R R12 ; 0 Y], as the reference's revert_conditional:
    // R^T R = Sigma^-, R^T R12 = cov(x_new, x_prev), R22 = qr_r(Y): the backward noise (Schur complement), gain Gp = (R^-1 R12)^T.
    double R12[N];
#pragma unroll
    for (int r = 0; r < N; ++r) R12[r] = 0.0;
    qr_stack<N, true>(T, M, R12, Bm, l);
    qr_square<N>(Bm, l);  // Bm is R22 now
    // Gp^T = R^-1 R12 by back substitution, one column of R12 per lane: lane q gets row q of Gp (reciprocal pivots, 0 on an
    // exactly singular factor)
    double z[N];
    sfor<0, N>([&](auto jr) PNODE_LAMBDA_INLINE {
      constexpr int j = N - 1 - decltype(jr)::value;
      const double dj = T[j];
      const double rinv = bcast<j>((dj != 0.0) ? fast_rcp(dj) : 0.0);
      double acc = R12[j];
      sfor<j + 1, N>([&](auto kc) PNODE_LAMBDA_INLINE {
        constexpr int kk = decltype(kc)::value;
        acc -= bcast<kk>(T[j]) * z[kk];
      });
      z[j] = acc * rinv;
    });
    // back to plain coordinates
#pragma unroll
    for (int j = 0; j < N; ++j) {
      G[j] = pl * z[j] * pinv[j];
      RL[j] = Bm[j] * pl;
    }
  }
#pragma unroll
  for (int c = 0; c < N; ++c) Rm[c] = T[c] * pl;
}

// Compose  x_ckpt | x_prev ~ (G0, xi0, RL0)  with  x_prev | x_new ~ (G, xi, RL):  G <- G0 G ;  RL <- qr_r([RL G0^T ; RL0])
// (G, G0 row-lane; RL, RL0 col-lane; xi is handled per ODE dimension by the caller)
template <int N>
__device__ __forceinline__ void merge(const double (&G0)[N], const double (&RL0)[N], double (&G)[N], double (&RL)[N], int l) {
  double X[N], T[N], Gn[N];
#pragma unroll
  for (int c = 0; c < N; ++c) {  // row c of RL G0^T, col-lane: x_c[i] = sum_k RL[c][k] G0[i][k]
    double acc = 0.0;
    sfor<0, N>([&](auto kc) PNODE_LAMBDA_INLINE {
      constexpr int k = decltype(kc)::value;
      acc += bcast<k>(RL[c]) * G0[k];
    });
    X[c] = acc;
    if (c + 1 < N) after(X[c], RL[c + 1]);
  }
#pragma unroll
  for (int r = 0; r < N; ++r) T[r] = (l >= r) ? RL0[r] : 0.0;
  {
    double PT[N], PB[N];
    qr_stack<N, false>(T, X, PT, PB, l);
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {  // row-lane: Gn[i][j] = sum_k G0[i][k] G[k][j]
    double acc = 0.0;
    sfor<0, N>([&](auto kc) PNODE_LAMBDA_INLINE {
      constexpr int k = decltype(kc)::value;
      acc += G0[k] * bcast<k>(G[j]);
    });
    Gn[j] = acc;
    if (j + 1 < N) after(Gn[j], G[j + 1]);
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    G[j] = Gn[j];
    RL[j] = T[j];
  }
}

// y_i = sum_k A[i][k] x_k for a row-lane A and a per-lane vector x (x_k, y_i: one value per lane = per ODE dimension):
// the entries of A become uniform by row broadcasts. y supplies the start values; SIGN = -1 subtracts.
template <int N, int SIGN>
__device__ __forceinline__ void matvec_rows(double (&A)[N], const double (&x)[N], double (&y)[N]) {
  sfor<0, N>([&](auto ic) PNODE_LAMBDA_INLINE {
    constexpr int i = decltype(ic)::value;
    double acc = y[i];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double a = bcast<i>(A[k]);
      acc = SIGN > 0 ? acc + a * x[k] : acc - a * x[k];
    }
    y[i] = acc;
#pragma unroll
    for (int k = 0; k < N; ++k) after(y[i], A[k]);  // (row by row: at most n broadcasts in flight)
  });
}

}  // namespace ln
}  // namespace pnode
