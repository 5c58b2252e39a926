// Device helpers shared by the register-resident MFMA kernels (pnode_dense_mfma.hip, pnode_dense_lq.hip).
#pragma once
#include "pnode_device.hpp"

namespace pnode {

typedef double double4_t __attribute__((ext_vector_type(4)));

#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

constexpr int kNB = 16;  // panel height

__device__ __forceinline__ double group16_sum(double v) {
  v += __shfl_xor(v, 1, kWave);
  v += __shfl_xor(v, 2, kWave);
  v += __shfl_xor(v, 4, kWave);
  v += __shfl_xor(v, 8, kWave);
  return v;
}

__device__ __forceinline__ void precond_rt(int n, double dt, const IwpConst& K, double* p, double* pinv) {
  const double adt = fabs(dt);
  double pw = sqrt(adt);
  for (int i = n - 1; i >= 0; --i) {
    p[i] = pw / K.fact[i];
    pinv[i] = K.fact[i] / pw;
    pw *= adt;
  }
}

// ---------------------------------------------------------------------------------------------
// Blocked row-Householder LQ:  [T0 | Mt] -> T (lower triangular, written to v.R); 32-row outer blocks (two 16-row panels A, B per pass over the
// trailing rows, halving the trailing-matrix traffic), DPP reductions and ONE barrier per panel row.
// Strip register layout for HBM rows (128 contiguous bytes per lane per 64-column super-tile):
//      X[16 T + 4 j + q] = M[row0 + r][64 T + 16 g + 4 j + q]
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum over the 16 lanes of a DPP row; every lane of the row receives the total.
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

// ---------------------------------------------------------------------------------------------
// State factor of the handles that run the register-resident MFMA kernels: Ls (D rows, 16 DC columns: the D - d columns of the
// rank-(D - d) factor, pnode_common.hpp, padded with zero columns to whole super-tiles) is stored in HBM in the REGISTER ORDER
// of the 16-row strips ("tiled"): lane l = r + 16 g of strip s holds, as its pair pp = 8 T + e / 2, the two elements
//      Ls[16 s + r][64 T + 16 g + e], Ls[16 s + r][64 T + 16 g + e + 1]        (e even)
// at double2 index (s * 2 DC + pp) * 64 + l. Every load / store instruction of a wavefront then covers 1 KB of contiguous
// memory; in a row-major strip every lane touches a cache line of its own and the address unit serialises the 64 requests
// (measured: 2 TB/s against 4.5 TB/s). pnode_read_factor de-tiles on the host.
// ---------------------------------------------------------------------------------------------
constexpr int mfma_col_tiles(int DT, int DD) { return ((16 * (DT - DD) + 63) / 64) * 4; }
template <int DC>
__device__ __forceinline__ size_t tiled_index(int row, int col) {  // element index (in doubles) of Ls[row][col]
  const int s = row >> 4, r = row & 15, T = col >> 6, g = (col >> 4) & 3, e = col & 15;
  return ((size_t(s) * (2 * DC) + 8 * T + (e >> 1)) * 64 + r + 16 * g) * 2 + (e & 1);
}

// The LQ's output for the columns < 2d ("T scratch", v.R; read by the measurement update only) is stored in strip order too, with
// the columns of every 64-column super-tile PERMUTED so that the four columns one v_mfma_f64_16x16x4 contracts come from ONE
// 16-column tile: register e = 4 a + q of lane group g holds column 64 T + 16 a + 4 q + g (a = tile, q = 0..3). That is the C/D
// fragment of a 16 x 16 tile as the matrix core delivers it, so the LQ stores its tiles without a transpose, 1 KB per
// instruction, and the measurement update can skip the 16-column tiles in which a reflector panel of F is structurally zero.
// DCT = 2d / 16 column tiles.
template <int DCT>
__device__ __forceinline__ size_t tiledpi_index(int row, int col) {
  const int s = row >> 4, r = row & 15, T = col >> 6, a = (col >> 4) & 3, q = (col >> 2) & 3, g = col & 3, e = 4 * a + q;
  return ((size_t(s) * (2 * DCT) + 8 * T + (e >> 1)) * 64 + r + 16 * g) * 2 + (e & 1);
}

// 4 x 4 transpose of a C/D fragment among the four 16-lane rows of the wavefront: in, lane (r, g) holds columns g + 4 q of
// row r (q = 0..3); out, columns 4 g + q -- four CONSECUTIVE columns per lane, so that a tile of T leaves the wavefront in
// whole 128-byte rows instead of scattered 8-byte pieces. Two rounds of v_permlane32_swap / v_permlane16_swap (gfx950).
__device__ __forceinline__ void swap32(double& a, double& b) {
  auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
  auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double4_t transpose_groups(double4_t t) {
  double a0 = t[0], a1 = t[1], a2 = t[2], a3 = t[3];
  swap32(a0, a2);  // rows {2, 3} of register 0 <-> rows {0, 1} of register 2
  swap32(a1, a3);
  swap16(a0, a1);  // rows {1, 3} of register 0 <-> rows {0, 2} of register 1
  swap16(a2, a3);
  return double4_t{a0, a1, a2, a3};
}

// One panel's reflectors applied to a register strip X (HBM layout). Returns T[J][P] = T0 - W in tvals (C/D fragment:
// lane (r, g) holds columns g + 4 q of row r).
template <int DT, int LDV>
__device__ __forceinline__ void apply_panel_to_strip(double (&X)[4 * DT], const double* __restrict__ Vp,
                                                     const double* __restrict__ Tfp, const double* __restrict__ sclp,
                                                     double4_t t0v, double4_t& tvals, int r, int g) {
  double4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
  double4_t sc;
#pragma unroll
  for (int q = 0; q < 4; ++q) sc[q] = sclp[g + 4 * q];
#pragma unroll
  for (int T = 0; T < DT / 4; ++T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* vrow = Vp + r * LDV + 64 * T + 16 * g + q;
      a0 = MFMA64(vrow[0], X[16 * T + q], a0);
      a1 = MFMA64(vrow[4], X[16 * T + 4 + q], a1);
      a2 = MFMA64(vrow[8], X[16 * T + 8 + q], a2);
      a3 = MFMA64(vrow[12], X[16 * T + 12 + q], a3);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the LDS operand loads of later super-tiles from being hoisted
  }
  const double4_t acc = t0v + ((a0 + a1) + (a2 + a3)) * sc;  // G = T0 + (X raw^T) diag(scl)
  double4_t w4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w4 = MFMA64(Tfp[(4 * kk + g) * kNB + r], acc[kk], w4);
  tvals = t0v - w4;
  w4 = w4 * sc;  // the update below multiplies by the unscaled rows
  const int coff = 16 * (r & 3) + (r >> 2);
#pragma unroll
  for (int T = 0; T < DT / 4; ++T) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double4_t c4 = {X[16 * T + 4 * j], X[16 * T + 4 * j + 1], X[16 * T + 4 * j + 2], X[16 * T + 4 * j + 3]};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) c4 = MFMA64(-Vp[(4 * kk + g) * LDV + 64 * T + 4 * j + coff], w4[kk], c4);
      X[16 * T + 4 * j] = c4[0];
      X[16 * T + 4 * j + 1] = c4[1];
      X[16 * T + 4 * j + 2] = c4[2];
      X[16 * T + 4 * j + 3] = c4[3];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same, with the LDS operand reads software-pipelined in groups of eight: the reads of group h + 1 are in flight while the
// matrix core works on group h (8 x 64 cycles, longer than an LDS round trip under load, 300-400 cycles). In the version above
// every 64-column super-tile starts with its 16 reads exposed (the scheduling barrier that bounds the register pressure also
// stops the prefetch): eight exposed round trips per application, a quarter of its run time when the wavefront has the SIMD to itself.
template <int DT, int LDV>
__device__ __forceinline__ void apply_panel_to_strip_pipelined(double (&X)[4 * DT], const double* __restrict__ Vp,
                                                               const double* __restrict__ Tfp, const double* __restrict__ sclp,
                                                               double4_t t0v, double4_t& tvals, int r, int g) {
  constexpr int NG = DT / 2;  // groups of 8 operands: half a super-tile
  double4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
  double va[8], vb[8];
  // ---- G = X V^T: group h = (T, half): operands V[r][64 T + 16 g + 8 half + e], e = 0..7
  const double* vbase = Vp + r * LDV + 16 * g;
  auto load_g = [&](double (&dst)[8], int h) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = vbase[64 * (h >> 1) + 8 * (h & 1) + e];
  };
  auto mfma_g = [&](const double (&src)[8], int h) {
    const int xo = 16 * (h >> 1) + 8 * (h & 1);
    // (column 4 i + q of the half goes to accumulator i' = 2 (h & 1) + i: the same four chains as the plain version)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (h & 1) {
        a2 = MFMA64(src[q], X[xo + q], a2);
        a3 = MFMA64(src[4 + q], X[xo + 4 + q], a3);
      } else {
        a0 = MFMA64(src[q], X[xo + q], a0);
        a1 = MFMA64(src[4 + q], X[xo + 4 + q], a1);
      }
    }
  };
  load_g(va, 0);
#pragma unroll
  for (int h = 0; h < NG; h += 2) {
    load_g(vb, h + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_g(va, h);
    __builtin_amdgcn_sched_barrier(0);
    if (h + 2 < NG) load_g(va, h + 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_g(vb, h + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  double4_t sc;
#pragma unroll
  for (int q = 0; q < 4; ++q) sc[q] = sclp[g + 4 * q];
  double tf4[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) tf4[kk] = Tfp[(4 * kk + g) * kNB + r];
  // ---- update operands of the first group are requested before the small W = G Tf product
  const int coff = 16 * (r & 3) + (r >> 2);
  const double* ubase = Vp + g * LDV + coff;
  auto load_u = [&](double (&dst)[8], int h) {  // group h = tiles (T, j) = (h / 2, 2 (h & 1) + {0, 1}), four reflector groups kk each
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dst[4 * t + kk] = ubase[4 * kk * LDV + 64 * (h >> 1) + 4 * (2 * (h & 1) + t)];
  };
  load_u(va, 0);
  const double4_t acc = t0v + ((a0 + a1) + (a2 + a3)) * sc;  // G = T0 + (X raw^T) diag(scl)
  double4_t w4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w4 = MFMA64(tf4[kk], acc[kk], w4);
  tvals = t0v - w4;
  w4 = w4 * sc;  // the update below multiplies by the unscaled rows
  auto mfma_u = [&](const double (&src)[8], int h) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int xo = 16 * (h >> 1) + 4 * (2 * (h & 1) + t);
      double4_t c4 = {X[xo], X[xo + 1], X[xo + 2], X[xo + 3]};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) c4 = MFMA64(-src[4 * t + kk], w4[kk], c4);
      X[xo] = c4[0];
      X[xo + 1] = c4[1];
      X[xo + 2] = c4[2];
      X[xo + 3] = c4[3];
    }
  };
#pragma unroll
  for (int h = 0; h < NG; h += 2) {
    load_u(vb, h + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_u(va, h);
    __builtin_amdgcn_sched_barrier(0);
    if (h + 2 < NG) load_u(va, h + 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_u(vb, h + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace pnode
