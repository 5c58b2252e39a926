// n x n algebra of the isotropic model (and of the dense model with d = 1, where the two coincide): wave-uniform, in registers,
// computed by every lane. Shared by pnode_iso.hip and pnode_dense_small.hip.
#pragma once
#include "pnode_device.hpp"

namespace pnode {

// ---------------------------------------------------------------------------------------------
// Small (n x n) pieces, all wave-uniform
// ---------------------------------------------------------------------------------------------

// Time update in preconditioned coordinates.
//   in : Rprev (Sigma_prev = Rprev^T Rprev, plain coordinates), p/pinv for this dt, sigma
//   out: Rm   (Sigma^- = Rm^T Rm, plain coordinates, upper triangular)
//        if COND: G (n x n, plain coordinates) and RL (upper, Lambda = RL^T RL) of the backward
//        transition  x_prev | x_new ~ N(G x_new + xi, Lambda)
template <int N, bool COND>
__device__ __forceinline__ void iso_time_update(const IwpConst& K, const double (&Rprev)[N][N], const double (&p)[N],
                                                const double (&pinv)[N], double sigma, double (&Rm)[N][N],
                                                double (&G)[N][N], double (&RL)[N][N]) {
  double Bm[N][N];  // Rprev * diag(pinv)
  double M[N][N];   // Bm * Abar^T  = (Abar Ltilde)^T
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) Bm[c][i] = Rprev[c][i] * pinv[i];
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int k = i; k < N; ++k) acc += Bm[c][k] * K.a[i][k];
      M[c][i] = acc;
    }
  double R[N][N];
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) R[c][i] = (i >= c) ? sigma * K.qsT[c][i] : 0.0;
  if constexpr (!COND) {
#pragma unroll
    for (int c = 0; c < N; ++c) {
      double x[N];
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = M[c][i];
      givens_rotate_in<N>(R, x);
    }
  } else {
    // ONE triangularisation of the stack [sigma chol(Qbar)^T 0 ; M Bm] (2n x 2n) = Q [R R12 ; 0 R22], as the reference's
    // revert_conditional does it: R^T R = Sigma^-, R^T R12 = cov(x_new, x_prev), R22^T R22 = the backward noise Lambda (Schur
    // complement), gain Gp = (R^-1 R12)^T. The rotations that build R are the ones of the filter (same order, same
    // arithmetic: R comes out bit for bit as without COND); the columns of R12 ride along, and what is left of a row goes into R22.
    double R12[N][N], R22[N][N];
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) R12[c][i] = R22[c][i] = 0.0;
#pragma unroll
    for (int c = 0; c < N; ++c) {
      double x[N], y[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        x[i] = M[c][i];
        y[i] = Bm[c][i];
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const double xj = x[j];
        if (xj != 0.0) {
          const double rjj = R[j][j];
          const double h2 = fma(rjj, rjj, xj * xj);
          const double inv = fast_rsqrt(h2);
          const double cs = rjj * inv, sn = xj * inv;
          R[j][j] = h2 * inv;
#pragma unroll
          for (int l = j + 1; l < N; ++l) {
            const double rl = R[j][l], xl = x[l];
            R[j][l] = cs * rl + sn * xl;
            x[l] = cs * xl - sn * rl;
          }
#pragma unroll
          for (int l = 0; l < N; ++l) {
            const double rl = R12[j][l], yl = y[l];
            R12[j][l] = cs * rl + sn * yl;
            y[l] = cs * yl - sn * rl;
          }
        }
      }
      givens_rotate_in<N>(R22, y);
    }
    // Gp^T = R^-1 R12 (back substitution; reciprocal pivots, 0 on an exactly singular factor)
    double rinv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) rinv[j] = (R[j][j] != 0.0) ? fast_rcp(R[j][j]) : 0.0;
    double Gp[N][N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      double zc[N];
#pragma unroll
      for (int j = N - 1; j >= 0; --j) {
        double acc = R12[j][q];
#pragma unroll
        for (int kk = j + 1; kk < N; ++kk) acc -= R[j][kk] * zc[kk];
        zc[j] = acc * rinv[j];
      }
#pragma unroll
      for (int j = 0; j < N; ++j) Gp[q][j] = zc[j];
    }
    // back to plain coordinates
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) {
        G[i][j] = p[i] * Gp[i][j] * pinv[j];
        RL[i][j] = R22[i][j] * p[j];
      }
  }
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) Rm[c][i] = R[c][i] * p[i];
}

// Compose  x_ckpt | x_prev ~ (G0, xi0, RL0)  with  x_prev | x_new ~ (G, xi, RL):
//   G <- G0 G ;  RL <- qr_r([RL G0^T ; RL0])   (xi is handled per lane by the caller)
template <int N>
__device__ __forceinline__ void iso_merge_small(const double (&G0)[N][N], const double (&RL0)[N][N], double (&G)[N][N],
                                                double (&RL)[N][N]) {
  double Rn[N][N];
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) Rn[c][i] = (i >= c) ? RL0[c][i] : 0.0;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    double x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) acc += RL[c][k] * G0[i][k];
      x[i] = acc;
    }
    givens_rotate_in<N>(Rn, x);
  }
  double Gn[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) acc += G0[i][k] * G[k][j];
      Gn[i][j] = acc;
    }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      G[i][j] = Gn[i][j];
      RL[i][j] = Rn[i][j];
    }
}

template <int N>
__device__ __forceinline__ void load_small(const double* __restrict__ src, double (&dst)[N][N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) dst[i][j] = src[i * N + j];
}
template <int N>
__device__ __forceinline__ void store_small(double* __restrict__ dst, const double (&src)[N][N], int lane) {
  // wave-uniform data: let lane (i*N + j) % 64 write element (i, j)
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (((i * N + j) & 63) == lane) dst[i * N + j] = src[i][j];
}

}  // namespace pnode
