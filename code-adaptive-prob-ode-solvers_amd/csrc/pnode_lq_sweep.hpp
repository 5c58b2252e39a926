// Register-resident factorisation of a 16-row panel held by ONE wavefront in the strip layout of pnode_mfma_common.hpp
// (shared by the one-launch LQ kernels pnode_dense_lq.hip and pnode_dense_lq4.hip).
#pragma once
#include "pnode_mfma_common.hpp"

namespace pnode {

// ---------------------------------------------------------------------------------------------------------------------
// Register-resident panel factorisation in the strip layout.
// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ double bcast16(double v) {  // lane C of every 16-lane DPP row -> all lanes of that row
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + C, 0xf, 0xf, false);
}

// acc_i += x_e(lane C of the row) * x_e(this lane): the dot product of every row with row C, partial over this lane's columns.
// (s_nop 1: a VGPR written by a VALU instruction needs two wait states before a DPP read; nothing pads inside asm.)
template <int C>
__device__ __forceinline__ void dot16(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3, double x4, double x5, double x6, double x7, double x8, double x9, double x10, double x11, double x12, double x13, double x14, double x15) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %4, %4 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %5, %5 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %6, %6 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %7, %7 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %8, %8 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %9, %9 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %10, %10 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %11, %11 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %12, %12 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %13, %13 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %14, %14 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %15, %15 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %16, %16 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %17, %17 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %18, %18 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %19, %19 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(x8), "v"(x9), "v"(x10), "v"(x11), "v"(x12), "v"(x13), "v"(x14), "v"(x15), "i"(C));
}
// x_e += x_e(lane C of the row) * nws  (nws = 0 on the lanes whose row is not updated, lane C included)
template <int C>
__device__ __forceinline__ void upd16(double& x0, double& x1, double& x2, double& x3, double& x4, double& x5, double& x6, double& x7, double& x8, double& x9, double& x10, double& x11, double& x12, double& x13, double& x14, double& x15, double nws) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %0, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %1, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %2, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %3, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %4, %4, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %5, %5, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %6, %6, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %7, %7, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %8, %8, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %9, %9, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %10, %10, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %11, %11, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %12, %12, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %13, %13, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %14, %14, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %15, %15, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9), "+v"(x10), "+v"(x11), "+v"(x12), "+v"(x13), "+v"(x14), "+v"(x15)
      : "v"(nws), "i"(C));
}

// Sum of a value over the four 16-lane DPP rows of the wavefront (the four column groups of the strip layout); every lane
// receives the total, formed in the same order. v_permlane16_swap exchanges rows 1 <-> 0 and 3 <-> 2 between its two operands,
// v_permlane32_swap the upper half of one with the lower half of the other (gfx950).
__device__ __forceinline__ double sum_over_groups(double v) {
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(v);
  hi = __double2hiint(v);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto e = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(e[0], c[0]) + __hiloint2double(e[1], c[1]);
}

// Lane predicates of the sweep as LITERAL lane masks (lane l holds row l & 15): a select costs two v_cndmask_b32 with the mask
// in an SGPR pair (s_mov_b64 on the scalar unit) instead of v_cmp + wait states + v_cndmask on the vector unit. The sweep is
// bound by the number of vector instructions a wavefront issues between the FMAs (scripts/dpp_probe.hip: 5.6 cycles per DPP FMA,
// 4 to 5 for everything else, one at a time), not by their latencies.
constexpr unsigned long long rep4(unsigned v) {
  const unsigned long long w = v & 0xFFFFu;
  return w | (w << 16) | (w << 32) | (w << 48);
}
template <int C> constexpr unsigned long long kRowsBelow = rep4(0xFFFFu << (C + 1));  // r > C
template <int C> constexpr unsigned long long kRowIs = rep4(1u << C);                   // r == C
template <unsigned long long MASK>
__device__ __forceinline__ double sel_mask(double a, double b) {  // lanes of MASK: a, others: b
  const unsigned long long m = MASK;
  unsigned lo, hi;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(b)), "v"(__double2loint(a)), "s"(m));
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(b)), "v"(__double2hiint(a)), "s"(m));
  return __hiloint2double(hi, lo);
}
template <unsigned long long MASK>
__device__ __forceinline__ double sel_mask_or_zero(double a) {  // lanes of MASK: a, others: 0
  const unsigned long long m = MASK;
  unsigned lo, hi;
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(lo) : "v"(__double2loint(a)), "s"(m));
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(hi) : "v"(__double2hiint(a)), "s"(m));
  return __hiloint2double(hi, lo);
}

// Terms 4 Q .. 4 Q + 3 (those below C) of the T-factor recurrence acc_r += Tf[r][M] u_M: one DPP FMA per term, FOUR accumulators
// so that the terms do not wait for each other (as a single chain every term cost 16 cycles: two wait states + the latency).
// HAZARD (gfx950): a VGPR that a DPP instruction uses -- any operand -- must have been written at least two wait states earlier
// when its writer is a VALU instruction; the compiler does not look into inline asm and may place a register copy right in
// front of it (it does: scripts/sweep_probe.hip), so every asm block of DPP instructions starts with s_nop 1.
#define PNODE_TF(a, m) "v_fmac_f64_dpp " a ", %[u], " m " row_newbcast:%[b" m "] row_mask:0xf bank_mask:0xf\n\t"
template <int Q, int C>
__device__ __forceinline__ void tf_block(double (&acc)[4], double u, const double (&tfrow)[kNB]) {
  constexpr int M0 = 4 * Q, N = (C - M0 >= 4) ? 4 : C - M0;
  if constexpr (N == 1)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %[a0], %[u], %[t0] row_newbcast:%[m0] row_mask:0xf bank_mask:0xf"
                 : [a0] "+v"(acc[0]) : [u] "v"(u), [t0] "v"(tfrow[M0]), [m0] "i"(M0));
  else if constexpr (N == 2)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %[a0], %[u], %[t0] row_newbcast:%[m0] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %[a1], %[u], %[t1] row_newbcast:%[m1] row_mask:0xf bank_mask:0xf"
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]) : [u] "v"(u), [t0] "v"(tfrow[M0]), [t1] "v"(tfrow[M0 + 1]), [m0] "i"(M0), [m1] "i"(M0 + 1));
  else if constexpr (N == 3)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %[a0], %[u], %[t0] row_newbcast:%[m0] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %[a1], %[u], %[t1] row_newbcast:%[m1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %[a2], %[u], %[t2] row_newbcast:%[m2] row_mask:0xf bank_mask:0xf"
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2])
                 : [u] "v"(u), [t0] "v"(tfrow[M0]), [t1] "v"(tfrow[M0 + 1]), [t2] "v"(tfrow[M0 + 2]), [m0] "i"(M0), [m1] "i"(M0 + 1), [m2] "i"(M0 + 2));
  else if constexpr (N == 4)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %[a0], %[u], %[t0] row_newbcast:%[m0] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %[a1], %[u], %[t1] row_newbcast:%[m1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %[a2], %[u], %[t2] row_newbcast:%[m2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %[a3], %[u], %[t3] row_newbcast:%[m3] row_mask:0xf bank_mask:0xf"
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                 : [u] "v"(u), [t0] "v"(tfrow[M0]), [t1] "v"(tfrow[M0 + 1]), [t2] "v"(tfrow[M0 + 2]), [t3] "v"(tfrow[M0 + 3]), [m0] "i"(M0),
                   [m1] "i"(M0 + 1), [m2] "i"(M0 + 2), [m3] "i"(M0 + 3));
  if constexpr (M0 + 4 < C) tf_block<Q + 1, C>(acc, u, tfrow);
}
#undef PNODE_TF

// Pivot C of the sweep. On entry rows 0 .. C are final (row C received its last update at pivot C - 1).
//   v_C = [e_C ; scl_C x_C]  (x_C stays unscaled in the registers),  H_C = I - tau_C v_C^T v_C.
// tpp_row[C] <- T[row r][column C of the panel] for r >= C (beta on the diagonal, -w below; the entries of the rows r < C, zeros
//              of the triangle, are NOT written as such: the reader masks them)
// tfrow[C] <- Tf[r][C] of the compact-WY factor, H_0 ... H_C = I - V^T Tf V  (forward recurrence, dlarft)
// A pivot row without a single non-zero entry is reflected like any other (v = e_C, tau = 2: the sign of column C flips);
// 1e-200 keeps the reciprocal square root finite when the T0 pivot is zero too (output scale 0 and a zero factor).
template <int C, int DT>
__device__ __forceinline__ void pivot_step(double (&X)[4 * DT], double (&tfrow)[kNB], double* __restrict__ tpp_row, double& myscl,
                                           double alpha) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
  for (int e = 0; e < 4 * DT; e += 16)
    dot16<C>(a0, a1, a2, a3, X[e], X[e + 1], X[e + 2], X[e + 3], X[e + 4], X[e + 5], X[e + 6], X[e + 7], X[e + 8], X[e + 9],
             X[e + 10], X[e + 11], X[e + 12], X[e + 13], X[e + 14], X[e + 15]);
  const double part = (a0 + a1) + (a2 + a3);
  const double dotv = sum_over_groups(part);  // every lane of row r holds x_r . x_C
  const double xx = bcast16<C>(dotv);  // |x_C|^2   (NaN: a NaN row propagates, like the reference's QR)
  const double nrm2 = fma(alpha, alpha, xx) + 1e-200;
  const double rs = fast_rsqrt(nrm2);
  const double beta = -copysign(nrm2 * rs, alpha);
  const double tauc = fma(fabs(alpha), rs, 1.0);
  const double scale = copysign(rs, alpha) * fast_rcp(tauc);
  const double sa = scale * dotv;
  const double w = tauc * sa;  // (the rows of one panel have no T0 entry in each other's columns)
  const double nws = sel_mask_or_zero<kRowsBelow<C>>(-(w * scale));
  tpp_row[C] = sel_mask<kRowIs<C>>(beta, -w);  // (the four lanes of a row store the same value)
  double tfc;
  if constexpr (C > 0) {
    const double u = myscl * sa;  // v_r . v_C for the finished reflectors r < C; zero on the other rows (their myscl still is)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    tf_block<0, C>(acc, u, tfrow);
    const double ind = sel_mask_or_zero<kRowIs<C>>(1.0);
    tfc = tauc * (ind - ((acc[0] + acc[1]) + (acc[2] + acc[3])));  // -tau acc on the rows r < C (acc is zero elsewhere: Tf is upper triangular)
  } else {
    tfc = sel_mask_or_zero<kRowIs<C>>(tauc);
  }
  tfrow[C] = tfc;
  myscl = sel_mask<kRowIs<C>>(scale, myscl);
  // pin the recurrence inside its pivot: without this the scheduler sinks all 120 broadcasts below the last asm block of the
  // sweep (they only feed the publication), spilling their operands on the way
  asm volatile("" : "+v"(tfrow[C]), "+v"(myscl));
  if constexpr (C + 1 < kNB) {  // (the last pivot has no row left to update)
#pragma unroll
    for (int e = 0; e < 4 * DT; e += 16)
      upd16<C>(X[e], X[e + 1], X[e + 2], X[e + 3], X[e + 4], X[e + 5], X[e + 6], X[e + 7], X[e + 8], X[e + 9], X[e + 10], X[e + 11],
               X[e + 12], X[e + 13], X[e + 14], X[e + 15], nws);
  }
}

template <int C, int DT>
__device__ __forceinline__ void pivot_sweep(double (&X)[4 * DT], double (&tfrow)[kNB], double* __restrict__ tpp_row, double& myscl,
                                            double alpha) {
  if constexpr (C < kNB) {
    pivot_step<C, DT>(X, tfrow, tpp_row, myscl, alpha);
    pivot_sweep<C + 1, DT>(X, tfrow, tpp_row, myscl, alpha);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// In-place variant: plain row-Householder LQ of a strip whose pivot columns lie INSIDE the strip (the measurement update's
// F = [C 0] Q, pnode_dense_mfma.hip). Panel `gp` (= the 16-row strip number, < 4) has its pivots in the columns 16 gp + C,
// i.e. in register X[C] of the lanes of column group gp. On entry to pivot C the columns before 16 gp + C are zero in the rows
// >= C (earlier pivot columns are cleared as they are extracted, the columns before the panel by the caller).
//   reflector C:  v_C = scl_C * xhat_C,  xhat_C = [0 .. 0, 1 / scl_C (pivot column), x_C (columns beyond)],  H_C = I - tau_C v_C^T v_C
// The rows stay unscaled in the registers (the pivot entry 1 / scl_C = -beta tau is stored in place), so that the panel
// applies to other strips as a plain product with no separate unit part.
// tpp_row[C] <- C-factor entry [row r][column C of the panel] for r >= C (the reader masks the upper triangle);
// tfrow[C]: compact-WY factor as in pivot_step.
// ---------------------------------------------------------------------------------------------------------------------
template <int C> constexpr unsigned long long kRowsFrom = rep4(0xFFFFu << C);  // r >= C
template <int C, int DT>
__device__ __forceinline__ void pivot_step_inplace(double (&X)[4 * DT], double (&tfrow)[kNB], double* __restrict__ tpp_row,
                                                   double& myscl, int gp, int g_) {
  int g = g_;
  asm volatile("" : "+v"(g));
  const bool pivcol = g == gp;
  // the strip's entries in the pivot column, delivered to every lane of their row; cleared in the rows >= C
  const double t_r = sum_over_groups(pivcol ? X[C] : 0.0);
  const double xc_cleared = sel_mask_or_zero<~kRowsFrom<C>>(X[C]);
  X[C] = pivcol ? xc_cleared : X[C];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
  for (int e = 0; e < 4 * DT; e += 16)
    dot16<C>(a0, a1, a2, a3, X[e], X[e + 1], X[e + 2], X[e + 3], X[e + 4], X[e + 5], X[e + 6], X[e + 7], X[e + 8], X[e + 9],
             X[e + 10], X[e + 11], X[e + 12], X[e + 13], X[e + 14], X[e + 15]);
  const double dotv = sum_over_groups((a0 + a1) + (a2 + a3));  // x_r . x_C over the columns beyond the pivot
  const double xx = bcast16<C>(dotv);
  const double alpha = bcast16<C>(t_r);
  // (a pivot row without an entry beyond the pivot is reflected like any other: v = e_C, tau = 2; see pivot_step)
  const double nrm2 = fma(alpha, alpha, xx) + 1e-200;
  const double rs = fast_rsqrt(nrm2);
  const double beta = -copysign(nrm2 * rs, alpha);
  const double tauc = fma(fabs(alpha), rs, 1.0);
  const double scale = copysign(rs, alpha) * fast_rcp(tauc);
  const double gr = fma(scale, dotv, t_r);  // (row r) . v_C
  const double w = tauc * gr;
  const double nws = sel_mask_or_zero<kRowsBelow<C>>(-(w * scale));
  tpp_row[C] = sel_mask<kRowIs<C>>(beta, t_r - w);  // (rows r < C: not meaningful, the reader masks the upper triangle)
  double tfc;
  if constexpr (C > 0) {
    const double u = myscl * gr;  // v_r . v_C for the finished reflectors r < C (only their lanes are read)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    tf_block<0, C>(acc, u, tfrow);
    const double ind = sel_mask_or_zero<kRowIs<C>>(1.0);
    tfc = tauc * (ind - ((acc[0] + acc[1]) + (acc[2] + acc[3])));
  } else {
    tfc = sel_mask_or_zero<kRowIs<C>>(tauc);
  }
  tfrow[C] = tfc;
  myscl = sel_mask<kRowIs<C>>(scale, myscl);
  asm volatile("" : "+v"(tfrow[C]), "+v"(myscl));
  if constexpr (C + 1 < kNB) {
#pragma unroll
    for (int e = 0; e < 4 * DT; e += 16)
      upd16<C>(X[e], X[e + 1], X[e + 2], X[e + 3], X[e + 4], X[e + 5], X[e + 6], X[e + 7], X[e + 8], X[e + 9], X[e + 10], X[e + 11],
               X[e + 12], X[e + 13], X[e + 14], X[e + 15], nws);
  }
  // 1 / scl_C = alpha - beta = -beta tau in place, on the pivot lane of the pivot column
  const double piv = sel_mask<kRowIs<C>>(-(beta * tauc), X[C]);
  X[C] = pivcol ? piv : X[C];
}

template <int C, int DT>
__device__ __forceinline__ void pivot_sweep_inplace(double (&X)[4 * DT], double (&tfrow)[kNB], double* __restrict__ tpp_row,
                                                    double& myscl, int gp, int g) {
  if constexpr (C < kNB) {
    pivot_step_inplace<C, DT>(X, tfrow, tpp_row, myscl, gp, g);
    pivot_sweep_inplace<C + 1, DT>(X, tfrow, tpp_row, myscl, gp, g);
  }
}

// Forward substitution with a 16 x 16 lower-triangular block held one row per lane (crow = row r of the block, every 16-lane
// DPP row alike): on exit xsol(lane r) = (block^-1 rhs)[r].
template <int C>
__device__ __forceinline__ void tri_forward16(double& rhs, double& xsol, const double (&crow)[kNB], int r) {
  if constexpr (C < kNB) {
    const double cand = rhs * fast_rcp(crow[C]);  // meaningful on row C
    const double xc = bcast16<C>(cand);
    xsol = (r == C) ? xc : xsol;
    rhs = (r > C) ? fma(-crow[C], xc, rhs) : rhs;
    tri_forward16<C + 1>(rhs, xsol, crow, r);
  }
}

}  // namespace pnode
