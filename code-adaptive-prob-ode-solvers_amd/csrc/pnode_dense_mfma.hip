// Dense EKF1 fast path for gfx950: MFMA-tiled square-root time and measurement updates.
// Used when ode_order == 1, d % 16 == 0 and D/16 is one of the instantiated tile counts.
//
// Data layout (per trajectory, HBM): L is D x D row-major with Sigma = L L^T. The time update is a
// blocked ROW-Householder LQ of the D x 2D stack [T0 | Mt], T0 = sigma chol(Qbar) (x) I (never
// materialised: it is block-diagonal per 16-row panel), Mt = (Abar P^-1 (x) I) L. Panels of 16 rows
// are factorised in LDS; the trailing rows are updated by v_mfma_f64_16x16x4_f64 in 16-row strips
// that live entirely in registers in the MFMA operand layout
//      lane (r = lane & 15, g = lane >> 4) holds X[s] = M[row0 + r][4 s + g],  s = 0 .. D/4 - 1,
// which is at once the B operand of  G^T = Vx X^T  and the C/D fragment of the transposed update
// X^T -= Vx^T W^T, so a strip is read once and written once per panel with no LDS staging.
// Pass 0 reads the state factor L directly (Mt is formed on the fly). The measurement update (projector form)
// uses the same strip scheme and the fact that L^- is lower
// triangular: F = H L^- has only 2d non-zero columns, so Y = L^- W^T and L = L^- - Y W touch D x 2d.
//
// Reference anchors: strategy_filter + correction_ts1 + solver_dynamic
// (/root/reference/experiments/1_van_der_pol/vdp.py:62-66); arithmetic restated in oracle/pn_oracle.py.
#include "pnode_lq_sweep.hpp"

namespace pnode {

struct PanelScratch {
  double* Tpp;    // [16][16] diagonal block of T (lower)
  double* Tf;     // [16][16] compact-WY factor
  double* tau;    // [16]
  double* scl;    // [16]
  double* beta;   // [16]
};

// Row-Householder factorisation of the 16 LDS rows starting at `rows` (stride LDV); one barrier per
// row. Groups above the pivot row (idle in a textbook sweep) compute the Gram entries the T-factor
// needs. The rows are left UNSCALED: reflector c is v_c = [1 ; scl[c] * rows[c]]; consumers scale
// their 16-wide G / W fragments by scl[] instead. On exit ps.Tpp holds the final diagonal block of T
// and ps.Tf the compact-WY factor.
template <int DT, int LDV>
__device__ __forceinline__ void factor_panel(double* rows, const PanelScratch& ps, double* Gm, double* Nm, double t0diag,
                                             int tid) {
  const int rr = tid >> 4, ch = tid & 15;
  // the sweep is a chain of dependent short instructions: give it issue priority over the co-resident
  // workgroup's MFMA stream (which only needs one issue slot per 64 cycles)
  __builtin_amdgcn_s_setprio(3);
  ps.Tpp[tid] = (rr == ch) ? t0diag : 0.0;
  Gm[tid] = 0.0;
  // each 16-lane group keeps ITS row in registers for the whole sweep; a row is published to LDS
  // once, when it has received its last update (it is then the next pivot row)
  double xr[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) xr[i] = rows[rr * LDV + ch + 16 * i];
  // Householder scalars of pivot row c, computed ONCE by the group that owns the row (right after the row's last
  // update) and published with it: beta = -sign(alpha) sqrt(alpha^2 + xx), tau = 1 + |alpha| rs, scale = 1/(alpha - beta).
  // alpha = T0[c][c] = t0diag for every c (column c of the diagonal block is untouched before step c).
  auto publish_scalars = [&](int c) {
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int i = 0; i < DT; i += 2) {
      p0 = fma(xr[i], xr[i], p0);
      p1 = fma(xr[i + 1], xr[i + 1], p1);
    }
    const double xx = row16_sum(p0 + p1);
    const double alpha = t0diag;
    double tauc = 0.0, scale = 0.0, beta = alpha;
    if (xx != 0.0) {
      const double nrm2 = fma(alpha, alpha, xx);
      const double rs = fast_rsqrt(nrm2);
      const double sgn = copysign(1.0, alpha);
      beta = -sgn * nrm2 * rs;
      tauc = fma(fabs(alpha), rs, 1.0);
      scale = sgn * rs * fast_rcp(tauc);
    }
    if (ch == 0) {
      ps.beta[c] = beta;
      ps.tau[c] = tauc;
      ps.scl[c] = scale;
    }
  };
  if (rr == 0) publish_scalars(0);
  __syncthreads();
  for (int c = 0; c < kNB; ++c) {
    double xc[DT];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int i = 0; i < DT; i += 2) {
      xc[i] = rows[c * LDV + ch + 16 * i];
      xc[i + 1] = rows[c * LDV + ch + 16 * (i + 1)];
      d0 = fma(xr[i], xc[i], d0);
      d1 = fma(xr[i + 1], xc[i + 1], d1);
    }
    const double dot = row16_sum(d0 + d1);
    const double tauc = ps.tau[c], scale = ps.scl[c];
    if (rr > c) {
      if (tauc != 0.0) {
        const double tjc = ps.Tpp[rr * kNB + c];
        const double w = tauc * fma(scale, dot, tjc);
        const double ws = w * scale;
#pragma unroll
        for (int i = 0; i < DT; ++i) xr[i] = fma(-ws, xc[i], xr[i]);
        if (ch == 0) ps.Tpp[rr * kNB + c] = tjc - w;
      }
      if (rr == c + 1) {
#pragma unroll
        for (int i = 0; i < DT; ++i) rows[rr * LDV + ch + 16 * i] = xr[i];
        publish_scalars(c + 1);
      }
    } else if (rr < c) {
      if (ch == 0) Gm[rr * kNB + c] = dot;  // raw rows[rr] . rows[c]  (row rr is final)
    }
    __syncthreads();
  }
  // T-factor: Tf = (D^-1 + striu(V V^T))^-1 = (I + N)^-1 D,  N = D striu(G), via the nilpotent product
  // (I + N)^-1 = (I - N)(I + N^2)(I + N^4)(I + N^8); every thread owns one entry (i, j) = (rr, ch).
  {
    const int i = rr, j = ch;
    double* A = ps.Tf;  // accumulates (I - N)(I + N^2)...
    double* Pm = Nm;    // current power of N
    const double si = ps.scl[i], sj = ps.scl[j];
    const double nij = (j > i) ? ps.tau[i] * si * sj * Gm[i * kNB + j] : 0.0;
    Pm[tid] = nij;
    A[tid] = ((i == j) ? 1.0 : 0.0) - nij;
    if (tid < kNB) ps.Tpp[tid * kNB + tid] = ps.beta[tid];
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
      double p2 = 0.0;
#pragma unroll
      for (int m = 0; m < kNB; ++m) p2 = fma(Pm[i * kNB + m], Pm[m * kNB + j], p2);
      __syncthreads();
      Pm[tid] = p2;
      __syncthreads();
      double ap = A[tid];
#pragma unroll
      for (int m = 0; m < kNB; ++m) ap = fma(A[i * kNB + m], Pm[m * kNB + j], ap);
      __syncthreads();
      A[tid] = ap;
      __syncthreads();
    }
    const double tf = A[tid] * ps.tau[j];
    __syncthreads();
    A[tid] = tf;
  }
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();
}

// FIRST = true : pass 0 only, reading the state factor L (Mt formed on the fly), writing Mt rows 32.. and T
// FIRST = false: passes 1 .. DT/2 - 1 from Mt.   (FIRST = false with first_pass = 0: all passes from Mt)
template <int DT, int DD, bool FIRST>
__global__ __launch_bounds__(256, 2) void dense_lq2_mfma_kernel(DenseView v, const StepArg* __restrict__ args,
                                                             const IwpConst* __restrict__ Kc, int first_pass) {
  static_assert(DT % 4 == 0, "the HBM strip layout needs D % 64 == 0");
  constexpr int D = 16 * DT;
  constexpr int DC = mfma_col_tiles(DT, DD), WC = 16 * DC;  // columns of Mt: the rank-deficient state factor (pnode_common.hpp)
  constexpr int LDV = WC + 1;  // 2*LDV = 2 (mod 64) dwords: conflict-free A-operand reads in the G phase
  constexpr int DCT = 2 * DD;  // 16-column tiles of the T scratch (columns < 2d), tile-aligned strip order (tiledpi_index)
  __shared__ double Vx[2 * kNB * LDV];
  __shared__ double TppS[2 * kNB * kNB];
  __shared__ double TfS[2 * kNB * kNB];
  __shared__ double Gm[kNB * kNB];
  __shared__ double Nm[kNB * kNB];
  __shared__ double small[6 * kNB];

  const int b = blockIdx.x;
  if (!args[b].active) return;
  const IwpConst& K = *Kc;
  const int d = v.d;
  const double sigma = v.sig_new[b];
  double* __restrict__ Mt = v.M + size_t(b) * D * D;
  double* __restrict__ T = v.R + size_t(b) * D * D;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  // pass 0 reads the state factor L directly: Mt = (Abar P^-1 (x) I) L is formed on the fly (no separate
  // kernel, no extra HBM round trip); coef[i][k] = Abar[i][k] / p_k
  __shared__ double coef[kMaxN * kMaxN];
  constexpr bool fused = FIRST;
  const StepArg sarg = args[b];
  const double* __restrict__ Lsrc = v.U + (size_t(sarg.src) * v.B + b) * D * D;
  const int nn = v.n;
  if (FIRST && tid == 0) {
    double pl[kMaxN], pil[kMaxN];
    precond_rt(nn, sarg.dt, K, pl, pil);
    for (int i = 0; i < nn; ++i)
      for (int k = 0; k < nn; ++k) coef[i * kMaxN + k] = (k >= i) ? K.a[i][k] * pil[k] : 0.0;
  }
  __syncthreads();
  PanelScratch psA{TppS, TfS, small, small + 16, small + 32};
  PanelScratch psB{TppS + 256, TfS + 256, small + 48, small + 64, small + 80};
  double* VA = Vx;
  double* VB = Vx + kNB * LDV;

  auto t0_tile = [&](int row0, int col0) {
    // T0[row0 + r][col0 + g + 4q]: sigma * chol(Qbar)[iblk_row][iblk_col] where the intra-block indices agree
    double4_t t;
    const int ib = row0 / d, jb = col0 / d;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      t[q] = (((row0 + r) % d) == ((col0 + g + 4 * q) % d) && jb <= ib) ? sigma * K.qsT[jb][ib] : 0.0;
    return t;
  };

  // Columns >= 2d of L^- are not touched by the measurement update (F = H L^- has 2d non-zero columns): the passes
  // that produce them write p_row * T straight into the destination slot of the state factor, so the strip-update
  // kernel neither reads nor copies them.
  double* __restrict__ Ldst = v.U + (size_t(sarg.dst) * v.B + b) * D * D;
  __shared__ double pS[kMaxN];
  if (tid == 0) {
    double pl[kMaxN], pil[kMaxN];
    precond_rt(nn, sarg.dt, K, pl, pil);
    for (int i = 0; i < kMaxN; ++i) pS[i] = (i < nn) ? pl[i] : 0.0;
  }
  __syncthreads();
  bool carried = false;  // the 32 rows of this pass's panels were left in Vx by the previous pass (no HBM round trip)
  for (int P = FIRST ? 0 : first_pass; P < (FIRST ? 1 : DT / 2); ++P) {
    const int c0 = 2 * kNB * P;
    const bool direct = c0 >= 2 * d;  // (2d is a multiple of 32: a pass is either all below or all beyond)
    if (!carried) {
      // the 32 rows of the pass's two panels, read in the register order of their strips (coalesced) and scattered into Vx:
      // pair pp of lane ln of strip sl holds M[16 sl + (ln & 15)][64 (pp / 8) + 16 (ln >> 4) + 2 (pp % 8) .. + 1]
      constexpr int NL2 = 2 * kNB * WC / 512;
      double2 t2[NL2];
      if constexpr (fused) {
        // pass 0: rows 0 .. 31 lie in derivative block 0 (d >= 32): Mt = sum_k coef[0][k] L[k d + row]; k outermost, so the
        // loads of one k are all in flight together
#pragma unroll
        for (int e = 0; e < NL2; ++e) t2[e].x = t2[e].y = 0.0;
        for (int k = 0; k < nn; ++k) {
          const double2* __restrict__ src = reinterpret_cast<const double2*>(Lsrc) + size_t(k * d / kNB) * (2 * DC) * 64;
          const double ck = coef[k];
          double2 lv[NL2];
#pragma unroll
          for (int e = 0; e < NL2; ++e) lv[e] = src[tid + 256 * e];
#pragma unroll
          for (int e = 0; e < NL2; ++e) {
            t2[e].x = fma(ck, lv[e].x, t2[e].x);
            t2[e].y = fma(ck, lv[e].y, t2[e].y);
          }
        }
      } else {
        const double2* __restrict__ src = reinterpret_cast<const double2*>(Mt) + size_t(c0 / kNB) * (2 * DC) * 64;
#pragma unroll
        for (int e = 0; e < NL2; ++e) t2[e] = src[tid + 256 * e];
      }
#pragma unroll
      for (int e = 0; e < NL2; ++e) {
        const int idx = tid + 256 * e;
        const int sl = idx / (2 * DC * 64), rem = idx % (2 * DC * 64), pp = rem / 64, ln = rem % 64;
        const int col = 64 * (pp / 8) + 16 * (ln >> 4) + 2 * (pp % 8);
        Vx[(kNB * sl + (ln & 15)) * LDV + col] = t2[e].x;
        Vx[(kNB * sl + (ln & 15)) * LDV + col + 1] = t2[e].y;
      }
    }
    __syncthreads();
    // ---- panel A
    factor_panel<DC, LDV>(VA, psA, Gm, Nm, sigma * K.qsT[c0 / d][c0 / d], tid);
    // ---- panel B rows <- reflectors of A (rows live in LDS; natural column order 16 tt + m)
    {
      const double4_t t0v = t0_tile(c0 + kNB, c0);
      // G^T = V_A V_B^T: each wavefront contracts a quarter of the columns; the four partial tiles are summed in a
      // fixed order (bit-reproducible) through LDS that is idle between the two panel factorisations
      double4_t ar0 = {0.0, 0.0, 0.0, 0.0}, ar1 = ar0;
      constexpr int SQ = WC / 16;  // k-steps (of 4 columns) per wavefront
#pragma unroll 8
      for (int s = SQ * wv; s < SQ * (wv + 1); s += 2) {
        ar0 = MFMA64(VA[r * LDV + 4 * s + g], VB[r * LDV + 4 * s + g], ar0);
        ar1 = MFMA64(VA[r * LDV + 4 * s + 4 + g], VB[r * LDV + 4 * s + 4 + g], ar1);
      }
      {
        double* slot[4] = {Gm, Nm, psB.Tpp, psB.Tf};
        const double4_t part = ar0 + ar1;
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[wv][4 * lane + q] = part[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q)
          ar0[q] = ((slot[0][4 * lane + q] + slot[1][4 * lane + q]) + slot[2][4 * lane + q]) + slot[3][4 * lane + q];
        ar1 = {0.0, 0.0, 0.0, 0.0};
      }
      double4_t sc;
#pragma unroll
      for (int q = 0; q < 4; ++q) sc[q] = psA.scl[g + 4 * q];
      const double4_t acc = t0v + (ar0 + ar1) * sc;
      double4_t w4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w4 = MFMA64(psA.Tf[(4 * kk + g) * kNB + r], acc[kk], w4);
      if (wv == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = c0 + kNB + r, col = c0 + g + 4 * q;
          if (direct) Ldst[tiled_index<DC>(row, col - d)] = pS[row / d] * (t0v[q] - w4[q]);
          else T[tiledpi_index<DCT>(row, col)] = t0v[q] - w4[q];
        }
      }
      w4 = w4 * sc;
      __syncthreads();  // every wave has read all of B before anyone overwrites its tiles
      for (int tt = wv; tt < DC; tt += 4) {
        double4_t c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) c4[q] = VB[r * LDV + 16 * tt + g + 4 * q];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) c4 = MFMA64(-VA[(4 * kk + g) * LDV + 16 * tt + r], w4[kk], c4);
#pragma unroll
        for (int q = 0; q < 4; ++q) VB[r * LDV + 16 * tt + g + 4 * q] = c4[q];
      }
      __syncthreads();
    }
    // ---- panel B
    factor_panel<DC, LDV>(VB, psB, Gm, Nm, sigma * K.qsT[(c0 + kNB) / d][(c0 + kNB) / d], tid);
    // ---- diagonal blocks of T
    {
      const int i = tid >> 4, j = tid & 15;
      if (j <= i) {
          if (direct) {
          Ldst[tiled_index<DC>(c0 + i, c0 + j - d)] = pS[(c0 + i) / d] * psA.Tpp[i * kNB + j];
          Ldst[tiled_index<DC>(c0 + kNB + i, c0 + kNB + j - d)] = pS[(c0 + kNB + i) / d] * psB.Tpp[i * kNB + j];
        } else {
          T[tiledpi_index<DCT>(c0 + i, c0 + j)] = psA.Tpp[i * kNB + j];
          T[tiledpi_index<DCT>(c0 + kNB + i, c0 + kNB + j)] = psB.Tpp[i * kNB + j];
        }
      }
    }
    // ---- trailing strips: both panels in one pass over the rows
    const int nstrips = DT - 2 * (P + 1);
    // Mt (scratch private to the LQ) is stored in the register order of the strips: every load / store instruction of a
    // wavefront covers 1 KB of contiguous memory (a row-major strip makes each lane touch a cache line of its own)
    auto mt_pair = [&](int j0, int pp) { return reinterpret_cast<double2*>(Mt) + (size_t(j0 / kNB) * (2 * DC) + pp) * 64 + lane; };
    auto load_strip = [&](int st, double (&X)[4 * DC]) {
      const int j0 = c0 + 2 * kNB + kNB * st;
      if constexpr (fused) {
        const int ib = j0 / d;
#pragma unroll
        for (int T4 = 0; T4 < DC / 4; ++T4) {
#pragma unroll
          for (int e = 0; e < 16; ++e) X[16 * T4 + e] = 0.0;
#pragma unroll 1
          for (int k = ib; k < nn; ++k) {
            const double ck = coef[ib * kMaxN + k];
            // (rows k d + q of the state factor form whole strips: 16 | d)
            const double2* __restrict__ lrow = reinterpret_cast<const double2*>(Lsrc) + (size_t((k * d + j0 % d) / kNB) * (2 * DC) + 8 * T4) * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              const double2 t2 = lrow[(e / 2) * 64];
              X[16 * T4 + e] = fma(ck, t2.x, X[16 * T4 + e]);
              X[16 * T4 + e + 1] = fma(ck, t2.y, X[16 * T4 + e + 1]);
            }
          }
        }
      } else
#pragma unroll
      for (int pp = 0; pp < 2 * DC; ++pp) {
        const double2 t2 = *mt_pair(j0, pp);
        X[2 * pp] = t2.x;
        X[2 * pp + 1] = t2.y;
      }
    };
    auto apply_both = [&](int st, double (&X)[4 * DC]) {
      const int j0 = c0 + 2 * kNB + kNB * st;
      double4_t ta, tb;
      apply_panel_to_strip<DC, LDV>(X, VA, psA.Tf, psA.scl, t0_tile(j0, c0), ta, r, g);
      apply_panel_to_strip<DC, LDV>(X, VB, psB.Tf, psB.scl, t0_tile(j0, c0 + kNB), tb, r, g);
      if (direct) {
        const double osc = pS[(j0 + r) / d];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Ldst[tiled_index<DC>(j0 + r, c0 + g + 4 * q - d)] = osc * ta[q];
          Ldst[tiled_index<DC>(j0 + r, c0 + kNB + g + 4 * q - d)] = osc * tb[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          T[tiledpi_index<DCT>(j0 + r, c0 + g + 4 * q)] = ta[q];
          T[tiledpi_index<DCT>(j0 + r, c0 + kNB + g + 4 * q)] = tb[q];
        }
      }
    };
    auto store_strip = [&](int st, const double (&X)[4 * DC]) {
      const int j0 = c0 + 2 * kNB + kNB * st;
#pragma unroll
      for (int pp = 0; pp < 2 * DC; ++pp) {
        double2 t2;
        t2.x = X[2 * pp];
        t2.y = X[2 * pp + 1];
        *mt_pair(j0, pp) = t2;
      }
    };
    // Strips 0 and 1 are the panels of the NEXT pass: wavefronts 0 and 1 process them last, keep them in registers
    // over the end-of-pass barrier and drop them straight into Vx, instead of a write to and a re-read from HBM.
    const bool carry = !FIRST && nstrips >= 2;
    const bool keeper = carry && wv < 2;
    const int nmine = nstrips > wv ? (nstrips - wv + 3) / 4 : 0;
    double X[4 * DC];
    for (int it = 0; it < nmine; ++it) {
      const bool last = it + 1 == nmine;
      const int st = keeper ? (last ? wv : wv + 4 * (it + 1)) : wv + 4 * it;
      load_strip(st, X);
      apply_both(st, X);
      if (!(keeper && last)) store_strip(st, X);
    }
    __syncthreads();  // every wavefront is done with the reflectors in Vx
    if (keeper) {
#pragma unroll
      for (int T4 = 0; T4 < DC / 4; ++T4)
#pragma unroll
        for (int e = 0; e < 16; ++e) Vx[(kNB * wv + r) * LDV + 64 * T4 + 16 * g + e] = X[16 * T4 + e];
    }
    carried = carry;  // (the barrier at the top of the next pass publishes the rows)
  }
}

// ---------------------------------------------------------------------------------------------
// begin / measurement-update kernels: small LDS footprints (several workgroups per
// CU hide the latency of the sequential d x d Cholesky), F/W kept in global memory (L2), and the
// D x 2d strip update spread over B * DT/4 workgroups.
// ---------------------------------------------------------------------------------------------

// Blocked lower Cholesky of a d x d matrix in LDS (leading dimension LD, d = 16*DD), 256 threads.
// Per 16-column block: (1) one wave factors the 16 x 16 diagonal block in registers (lane = row,
// pivots broadcast with v_readlane), (2) the rows below are solved against it (lane = row),
// (3) the trailing tiles are updated with MFMA. Three barriers per block instead of three per column.
template <int DD, int LD>
__device__ __forceinline__ void lds_cholesky_blocked(double* Sm, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
#pragma unroll 1
  for (int kb = 0; kb < DD; ++kb) {
    const int o = 16 * kb;
    if (wv == 0) {
      double a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = Sm[(o + r) * LD + o + j];  // lanes 16..63 duplicate rows 0..15
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[c]), c),
                                            __builtin_amdgcn_readlane(__double2loint(a[c]), c));
        const double rs = fast_rsqrt(piv);  // NaN for a non-positive pivot, like sqrt
        a[c] = (r == c) ? piv * rs : a[c] * rs;
#pragma unroll
        for (int q = c + 1; q < 16; ++q) {
          const double lq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[c]), q),
                                             __builtin_amdgcn_readlane(__double2loint(a[c]), q));
          a[q] -= a[c] * lq;
        }
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j <= r) Sm[(o + r) * LD + o + j] = a[j];
      }
    }
    __syncthreads();
    // rows below the diagonal block: x L11^T = a_row
    for (int row = o + 16 + tid; row < 16 * DD; row += 256) {
      double x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = Sm[row * LD + o + j];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double acc = x[c];
#pragma unroll
        for (int m = 0; m < c; ++m) acc -= x[m] * Sm[(o + c) * LD + o + m];
        x[c] = acc * fast_rcp(Sm[(o + c) * LD + o + c]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) Sm[row * LD + o + j] = x[j];
    }
    __syncthreads();
    // trailing update A22 -= L21 L21^T (lower tiles)
    const int nt = DD - kb - 1;
    for (int tile = wv; tile < nt * (nt + 1) / 2; tile += 4) {
      int ta = 0, rem = tile;
      while (rem > ta) { rem -= ta + 1; ++ta; }
      const int ra = o + 16 * (ta + 1), rb = o + 16 * (rem + 1);
      double4_t acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = Sm[(ra + g + 4 * q) * LD + rb + r];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = MFMA64(-Sm[(ra + r) * LD + o + 4 * s + g], Sm[(rb + r) * LD + o + 4 * s + g], acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) Sm[(ra + g + 4 * q) * LD + rb + r] = acc[q];
    }
    __syncthreads();
  }
}

// x <- C^-1 x for lower-triangular C (d x d in LDS), x of length d <= 64 in LDS: one wave, lane = row.
template <int d, int LD>
__device__ __forceinline__ void lds_forward_vec_wave(const double* Sm, double* x, int tid) {
  if (tid < 64) {
    const int row = tid < d ? tid : d - 1;
    double xr = x[row];
    const double inv = 1.0 / Sm[row * LD + row];
#pragma unroll 8
    for (int a = 0; a < d; ++a) {
      const double cand = xr * inv;  // meaningful on lane a
      const double xa = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cand), a),
                                         __builtin_amdgcn_readlane(__double2loint(cand), a));
      if (row == a) xr = xa;
      else if (row > a) xr -= Sm[row * LD + a] * xa;
    }
    if (tid < d) x[tid] = xr;
  }
  __syncthreads();
}

// First phase of an attempt: mean extrapolation, vector field and Jacobian at the extrapolated mean, unit-noise innovation S1 and its Cholesky
// factor, local calibration sigma_hat, error estimate. Sm: [d][d + 1], ybuf, zbuf: [d], red: [4] doubles of LDS.
template <int DD>
__device__ __forceinline__ void begin_body(const DenseView& v, const StepArg& arg, const IwpConst& K, int b, double* Sm, double* ybuf,
                                           double* zbuf, double* red) {
  constexpr int d = 16 * DD, LD = d + 1;
  const int n = v.n;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  double p[kMaxN], pinv[kMaxN];
  precond_rt(n, arg.dt, K, p, pinv);
  const double lq00 = p[0] * K.qsT[0][0], lq10 = p[1] * K.qsT[0][1], lq11 = p[1] * K.qsT[1][1];
  const size_t sB = size_t(v.B);
  const double* msrc = v.mean + (size_t(arg.src) * sB + b) * n * d;
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * n * d;
  for (int j = tid; j < d; j += 256) {
    for (int i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * pinv[kk] * msrc[kk * d + j];
      const double e = p[i] * acc;
      mdst[i * d + j] = e;
      if (i == 0) ybuf[j] = e;
    }
  }
  __syncthreads();
  VfCtx ctx{v.vf_id, d, 1, v.prm ? v.prm + (v.prm_per_traj ? size_t(b) * v.P : 0) : nullptr,
            v.user_f ? v.user_f + size_t(b) * d : nullptr, v.user_J ? v.user_J + size_t(b) * d * 1 * d : nullptr};
  const double tnew = arg.t + arg.dt;
  double* zg = v.z + size_t(b) * d;
  for (int j = tid; j < d; j += 256) {
    const double z = mdst[d + j] - vf_f(ctx, ybuf, j, tnew);
    zbuf[j] = z;
    zg[j] = z;
  }
  // B0 = lq10 I - lq00 J, kept in LDS (Sm doubles as its buffer until the Gram is formed in registers)
  double* Jg = v.J + size_t(b) * d * d;
  for (int idx = tid; idx < d * d; idx += 256) {
    const int a = idx / d, q = idx % d;
    const double jv = vf_jac(ctx, ybuf, a, 0, q, tnew);
    Jg[idx] = jv;
    Sm[a * LD + q] = ((a == q) ? lq10 : 0.0) - lq00 * jv;
  }
  __syncthreads();
  // S1 = B0 B0^T + lq11^2 I  (lower tiles), accumulated in registers, then written over B0
  constexpr int NT = DD * (DD + 1) / 2;
  double4_t accs[(NT + 3) / 4];
  int cnt = 0;
  for (int tile = wv; tile < NT; tile += 4, ++cnt) {
    int ta = 0, rem = tile;
    while (rem > ta) { rem -= ta + 1; ++ta; }
    const int tb = rem;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int s = 0; s < d / 4; ++s)
      acc = MFMA64(Sm[(16 * ta + r) * LD + 4 * s + g], Sm[(16 * tb + r) * LD + 4 * s + g], acc);
    accs[cnt] = acc;
  }
  __syncthreads();
  cnt = 0;
  double* errg = v.err + size_t(b) * d;
  for (int tile = wv; tile < NT; tile += 4, ++cnt) {
    int ta = 0, rem = tile;
    while (rem > ta) { rem -= ta + 1; ++ta; }
    const int tb = rem;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * ta + g + 4 * q, col = 16 * tb + r;
      const double val = accs[cnt][q] + ((row == col) ? lq11 * lq11 : 0.0);
      Sm[row * LD + col] = val;
      if (row == col) errg[row] = val;
    }
  }
  __syncthreads();
  lds_cholesky_blocked<DD, LD>(Sm, tid);
  lds_forward_vec_wave<d, LD>(Sm, zbuf, tid);  // w = C^-1 z
  double part = 0.0;
  for (int j = tid; j < d; j += 256) part += zbuf[j] * zbuf[j];
  part = wave_sum(part);
  if (lane == 0) red[wv] = part;
  __syncthreads();
  const double sigma_hat = sqrt((red[0] + red[1] + red[2] + red[3]) / double(d));
  for (int j = tid; j < d; j += 256) errg[j] = arg.dt * sigma_hat * sqrt(errg[j]);
  if (tid == 0) v.sig_new[b] = v.dynamic ? sigma_hat : v.sigma[size_t(arg.src) * sB + b];
}

template <int DD>
__global__ __launch_bounds__(256) void dense_begin_mfma_kernel(DenseView v, const StepArg* __restrict__ args,
                                                               const IwpConst* __restrict__ Kc) {
  constexpr int d = 16 * DD, LD = d + 1;
  __shared__ double Sm[d * LD];
  __shared__ double ybuf[d];
  __shared__ double zbuf[d];
  __shared__ double red[4];
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  if (!arg.active) return;
  begin_body<DD>(v, arg, *Kc, b, Sm, ybuf, zbuf, red);
}

// ---------------------------------------------------------------------------------------------------------------------
// Measurement update of the MFMA path, by orthogonal transformations only (the reference's square-root correction: QR of
// (H L^-)^T, strategy_filter + correction_ts1, /root/reference/experiments/1_van_der_pol/vdp.py:64-65):
//   F = H L^- (d x 2d: L^- is lower triangular),  F = [C 0] Q^T by a blocked row-Householder LQ (S = C C^T is never formed),
//   L^- Q = [Y1 | Y2 | columns >= 2d]:  gain K = Y1 C^-1, mean m = m^- - Y1 (C^-1 z), and [Y2 | columns >= 2d] is a factor of the
//   updated covariance with D - d columns -- the observation is noise-free, the d directions of Y1 are known exactly afterwards.
// ONE kernel per trajectory, two workgroups per CU: the first half (F, its LQ, C^-1 z) is a chain of d pivots, latency; the second
// (every 16-row strip of L^- rotated by the reflector panels, which stay in LDS) is matrix-core streaming and fills the SIMDs
// while the other workgroup of the CU is in its first half. (r03 also measured the two halves as two kernels -- the first with
// 22 KB of LDS, the second with one or two workgroups per trajectory: 0.142 + 0.142 ms at B = 1024 against 0.31 for the r02
// kernel; the chain kernel leaves the CUs idle, three of its workgroups do not fit the register file: git show 7a66d4d.)
// The 16-column tiles in which a panel is structurally zero are skipped (176 instead of 272 MFMAs per strip).
// ---------------------------------------------------------------------------------------------------------------------

// Panel p of the LQ of F = [p1 T21 - p0 J T11 | p1 T22] (T22 lower triangular) is non-zero in the 16-column tiles p .. DD - 1 of
// the first half and DD .. DD + p of the second: the reflectors of row block p start at column 16 p, and a row of F ends at its
// diagonal entry in T22 (no fill-in beyond: the earlier reflectors end earlier).
template <int DD>
constexpr bool panel_has_tile(int p, int t) { return (t >= p && t < DD) || (t >= DD && t <= DD + p); }

// apply_panel_to_strip (pnode_mfma_common.hpp) for a strip and a panel whose columns are in the tile-aligned order of
// tiledpi_index (register e = 4 a + q of lane group g: column 64 T + 16 a + 4 q + g), without a T0 part, restricted to the
// tiles of panel P. LDS image of the panel: Vp[row][64 T + 16 g + e] = that register of lane (row, g).
template <int DF, int LDV, int DD, int P>
__device__ __forceinline__ void apply_panel_sparse(double (&X)[4 * DF], const double* __restrict__ Vp, const double* __restrict__ Tfp,
                                                   const double* __restrict__ sclp, int r, int g) {
  double4_t acc4[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) acc4[a] = double4_t{0.0, 0.0, 0.0, 0.0};
  double4_t sc;
#pragma unroll
  for (int q = 0; q < 4; ++q) sc[q] = sclp[g + 4 * q];
#pragma unroll
  for (int T = 0; T < DF / 4; ++T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* vrow = Vp + r * LDV + 64 * T + 16 * g + q;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (panel_has_tile<DD>(P, 4 * T + a)) acc4[a] = MFMA64(vrow[4 * a], X[16 * T + 4 * a + q], acc4[a]);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the LDS operand loads of later super-tiles from being hoisted
  }
  const double4_t gsum = ((acc4[0] + acc4[1]) + (acc4[2] + acc4[3])) * sc;  // G = (X raw^T) diag(scl)
  double4_t w4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w4 = MFMA64(Tfp[(4 * kk + g) * kNB + r], gsum[kk], w4);
  w4 = w4 * sc;  // the update below multiplies by the unscaled rows
  const int coff = 16 * (r & 3) + (r >> 2);
#pragma unroll
  for (int T = 0; T < DF / 4; ++T) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (panel_has_tile<DD>(P, 4 * T + j)) {
        double4_t c4 = {X[16 * T + 4 * j], X[16 * T + 4 * j + 1], X[16 * T + 4 * j + 2], X[16 * T + 4 * j + 3]};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) c4 = MFMA64(-Vp[(4 * kk + g) * LDV + 64 * T + 4 * j + coff], w4[kk], c4);
        X[16 * T + 4 * j] = c4[0];
        X[16 * T + 4 * j + 1] = c4[1];
        X[16 * T + 4 * j + 2] = c4[2];
        X[16 * T + 4 * j + 3] = c4[3];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int DD, int P, int DF, int LDV>
__device__ __forceinline__ void apply_all_panels(double (&X)[4 * DF], const double* __restrict__ Vp, const double* __restrict__ Tfp,
                                                 const double* __restrict__ sclp, int r, int g) {
  if constexpr (P < DD) {
    apply_panel_sparse<DF, LDV, DD, P>(X, Vp + P * kNB * LDV, Tfp + P * kNB * kNB, sclp + P * kNB, r, g);
    apply_all_panels<DD, P + 1, DF, LDV>(X, Vp, Tfp, sclp, r, g);
  }
}

// CTL (device-resident controller, pnode_advance_device): the tail of the kernel also takes the controller's decision for its
// trajectory (controller_decision, pnode_common.hpp): no controller kernel between the attempts. (Running the first phase of the
// NEXT attempt here as well -- two launches per attempt -- was measured and dropped: its serial Cholesky at two workgroups per CU
// costs more than the begin kernel at four, 0.340 against 0.261 + 0.049 ms at B = 1024.)
// Cycle stamps of the diagnostic build (make timing; scripts/correct_phase_stamps.py); empty otherwise.
struct CorrectStamps {
#ifdef PNODE_TIMING_BUILD
  unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __device__ __forceinline__ void at(int i) { t[i] = __builtin_amdgcn_s_memtime(); }
#else
  __device__ __forceinline__ void at(int) {}
#endif
};

template <int DT, int DD, bool CTL>
__global__ __launch_bounds__(256, 2) void dense_correct_mfma_kernel(DenseView v, StepArg* __restrict__ args, const IwpConst* __restrict__ Kc,
                                                                    TrajCtl* __restrict__ ctl, ControlParams cp) {
  constexpr int D = 16 * DT, d = 16 * DD, W2 = 2 * d, LDF = W2 + 1, DF = W2 / 16, DCT = DF, DC = mfma_col_tiles(DT, DD);
  static_assert(DD <= 4, "one wavefront per 16-row strip of F; the pivot columns of a strip lie in the first super-tile");
  static_assert(DT % 4 == 0, "whole strips per wavefront");
  extern __shared__ double gbuf[];
  double* Vf = gbuf;                      // [d][LDF]  reflector rows (unscaled, pivot entries 1 / scl): plain column order during the
                                          //           LQ of F, the tile-aligned order of the strips (tiledpi_index) afterwards
  double* TfF = Vf + d * LDF;             // [DD][16][16]
  double* sclF = TfF + DD * kNB * kNB;    // [d]
  double* tpp = sclF + d;                 // [16][16]  diagonal block of C of the sweeping wavefront
  double* wz = tpp + kNB * kNB;           // [2d] z, then C^-1 z block by block; zeros beyond d
  double* tv = wz + W2;                   // [2d] C^-1 z in the column order of the strips
  double* red = tv + W2;                  // [4]
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  const IwpConst& K = *Kc;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  if (!arg.active) {
    if constexpr (CTL) {  // (a trajectory that sat this attempt out may still have to be started: the first round of a call)
      if (tid == 0) {
        TrajCtl c = ctl[b];
        StepArg a = arg;
        if (c.status == 0) {
          controller_decision(c, a, 0.0, cp, true);
          args[b] = a;
          ctl[b] = c;
        }
      }
    }
    return;
  }
  CorrectStamps cs;
  cs.at(0);
  double p[kMaxN], pinv[kMaxN];
  precond_rt(v.n, arg.dt, K, p, pinv);
  const double p0 = p[0], p1 = p[1];
  const double* __restrict__ T = v.R + size_t(b) * D * D;  // tile-aligned strip order, columns < 2d (tiledpi_index)
  const double* __restrict__ Jg = v.J + size_t(b) * d * d;
  for (int j = tid; j < W2; j += 256) wz[j] = j < d ? v.z[size_t(b) * d + j] : 0.0;
  const bool owner = wv < DD;
  // ---- my strip of F in registers, plain strip order: X[16 T + e] = F[16 wv + r][64 T + 16 g + e]
  //      F = p1 T[d + ., 0 : 2d] - p0 [J T11 | 0]
  double X[4 * DF];
  if (owner) {
    // (addresses as ONE lane-dependent base + compile-time offsets: through tiledpi_index the compiler keeps a 64-bit address
    // per load alive, 200 registers of addresses)
    // plain column 64 T + 16 g + e of row 16 s + r sits at double index
    //   (s 2 DCT + 8 T) 128 + 256 g + 2 r   +   (e >> 3) 128 + 32 (e & 3) + ((e >> 2) & 1)
    const double* __restrict__ fbase = T + size_t(DD + wv) * (2 * DCT) * 128 + 256 * g + 2 * r;
#pragma unroll
    for (int T4 = 0; T4 < DF / 4; ++T4)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        // (the tiles beyond the diagonal tile of these rows are never written by the LQ: not read here)
        const bool beyond = 4 * T4 + g > DD + wv;
        X[16 * T4 + e] = beyond ? 0.0 : p1 * fbase[T4 * 1024 + (e >> 3) * 128 + 32 * (e & 3) + ((e >> 2) & 1)];
      }
    // - p0 (J T11)[16 wv + r][col]: the transposed product, so that the rows come out on the lanes: for the output registers
    // 4 j .. 4 j + 3 (columns 16 g + 4 j + i; register i of the C/D fragment of lane group g is row m = g + 4 i of the product)
    // lane r contributes, as row m = r of the A operand, column 16 (r & 3) + (r >> 2) + 4 j of T11; the B operand is my rows of J.
    // T11[k = 4 s4 + g][that column] sits at   256 (r & 3) + 32 (r >> 2) + 2 g   +   ((s4 >> 2) 2 DCT + (j >> 1)) 128 + 8 (s4 & 3) + (j & 1)
    double jrow[d / 4];
#pragma unroll
    for (int s4 = 0; s4 < d / 4; ++s4) jrow[s4] = -p0 * Jg[size_t(kNB * wv + r) * d + 4 * s4 + g];
    const bool live = 16 * (r & 3) < d;  // (d = 32: the second half of the super-tile has no J T11 part)
    const double* __restrict__ abase = T + 256 * (r & 3) + 32 * (r >> 2) + 2 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double aop[d / 4];
#pragma unroll
      for (int s4 = 0; s4 < d / 4; ++s4)  // all loads of a tile in flight together (zero above the diagonal of T11: the diagonal tiles store their zeros, the tiles beyond are never written)
        aop[s4] = (live && (r & 3) <= (s4 >> 2)) ? abase[((s4 >> 2) * (2 * DCT) + (j >> 1)) * 128 + 8 * (s4 & 3) + (j & 1)] : 0.0;  // (tile r & 3 of row strip s4 >> 2)
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < d / 4; ++s4) acc = MFMA64(aop[s4], jrow[s4], acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) X[4 * j + i] += acc[i];
      __builtin_amdgcn_sched_barrier(0);  // (one tile's operands at a time)
    }
  }
  __syncthreads();  // wz
  cs.at(1);
  // ---- blocked row-Householder LQ of F: wavefront w sweeps strip w in registers (pnode_lq_sweep.hpp); its panel stays in LDS
#pragma unroll 1
  for (int pnl = 0; pnl < DD; ++pnl) {
    if (owner && wv == pnl) {
      // my columns before the panel are final entries of C (row block wv): their share of the forward substitution
      //   rhs = z - C[rows, 0 : 16 wv] (C^-1 z)[0 : 16 wv]
      // is taken here, then they leave the strip (the reflectors of this panel act on the columns >= 16 wv only)
      double part = 0.0;
      if (g < wv) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          part = fma(X[e], wz[16 * g + e], part);
          X[e] = 0.0;
        }
      }
      double rhs = wz[kNB * wv + r] - sum_over_groups(part);
      double tfrow[kNB], myscl = 0.0;
#pragma unroll
      for (int m = 0; m < kNB; ++m) tfrow[m] = 0.0;
      pivot_sweep_inplace<0, DF>(X, tfrow, tpp + r * kNB, myscl, wv, g);
#pragma unroll
      for (int T4 = 0; T4 < DF / 4; ++T4)
#pragma unroll
        for (int e = 0; e < 16; ++e) Vf[(kNB * wv + r) * LDF + 64 * T4 + 16 * g + e] = X[16 * T4 + e];
      if (g == 0) {
        sclF[kNB * wv + r] = myscl;
#pragma unroll
        for (int m = 0; m < kNB; ++m) TfF[wv * kNB * kNB + r * kNB + m] = tfrow[m];
      }
      // forward substitution with the diagonal block (lower triangular, row r in tpp): lane r of every group alike
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double crow[kNB];
#pragma unroll
      for (int m = 0; m < kNB; ++m) crow[m] = (m <= r) ? tpp[r * kNB + m] : 0.0;  // (the sweep leaves the upper triangle unwritten)
      double xsol = 0.0;
      tri_forward16<0>(rhs, xsol, crow, r);
      if (g == 0) wz[kNB * wv + r] = xsol;
    }
    __syncthreads();
    if (owner && wv > pnl) {
      double4_t unused;
      apply_panel_to_strip<DF, LDF>(X, Vf + pnl * kNB * LDF, TfF + pnl * kNB * kNB, sclF + pnl * kNB, double4_t{0.0, 0.0, 0.0, 0.0},
                                    unused, r, g);
    }
  }
  __syncthreads();
  cs.at(2);
  // ---- the panels from the plain column order to the order of the strips, in place: slot 16 g + e of a super-tile (plain:
  //      column 16 g + e) moves to slot 16 (e & 3) + 4 g + (e >> 2) (there, register e' = 4 a + q of group g' is column
  //      16 a + 4 q + g'); every thread takes half a super-tile of one row
  {
    const int prow = tid >> 2, pT = (tid >> 1) & 1, half = tid & 1;
    const bool act = prow < d && pT < DF / 4;
    double tmp[32];
    double* rowp = Vf + prow * LDF + 64 * pT;
    if (act) {
#pragma unroll
      for (int i = 0; i < 32; ++i) tmp[i] = rowp[32 * half + i];
    }
    __syncthreads();
    if (act) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int k = 32 * half + i, gs = k >> 4, es = k & 15;
        rowp[16 * (es & 3) + 4 * gs + (es >> 2)] = tmp[i];
      }
    }
    for (int idx = tid; idx < W2; idx += 256) {  // slot idx = 64 T + 16 g' + e'
      const int T4 = idx >> 6, gp = (idx >> 4) & 3, ep = idx & 15;
      const int col = 64 * T4 + 16 * (ep >> 2) + 4 * (ep & 3) + gp;
      tv[idx] = col < d ? wz[col] : 0.0;
    }
  }
  __syncthreads();
  cs.at(3);
  // ---- every 16-row strip of L^- = diag(p) T (first 2d columns) rotated by the panels: X Q = [Y1 | Y2]. Y1 = L^- W^T feeds the
  //      mean, Y2 = L^- Q2 becomes the dense part of the new rank-(D - d) state factor (its columns 0 .. d-1, in the order the
  //      registers hold them: the column order of a factor is immaterial); the columns >= 2d of L^- were written by the LQ kernel
  //      as the columns >= d of Ls, already scaled.
  const size_t sB = size_t(v.B);
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * D;
  double* __restrict__ Ln = v.U + (size_t(arg.dst) * sB + b) * D * D;
#pragma unroll 1
  for (int st = wv; st < DT; st += 4) {
    const int row = 16 * st + r;
    const int iblk = (16 * st) / d;
    double pr = p[0];
#pragma unroll
    for (int i = 1; i < kMaxN; ++i) pr = (i == iblk) ? p[i] : pr;
    // the strip of T in the order it was stored: 1 KB per load instruction
    const double2* __restrict__ trow = reinterpret_cast<const double2*>(T) + size_t(st) * (2 * DCT) * 64 + lane;
#pragma unroll
    for (int pp = 0; pp < 2 * DF; ++pp) {
      const double2 t2 = trow[pp * 64];
      const bool beyond = (pp >> 1) > st;  // tile 4 T + a = pp / 2 of a strip of the first 2d rows: never written by the LQ
      X[2 * pp] = beyond ? 0.0 : pr * t2.x;
      X[2 * pp + 1] = beyond ? 0.0 : pr * t2.y;
    }
    apply_all_panels<DD, 0, DF, LDF>(X, Vf, TfF, sclF, r, g);
    // mean: m -= Y1 (C^-1 z)
    double mpart = 0.0;
#pragma unroll
    for (int T4 = 0; T4 < DF / 4; ++T4)
#pragma unroll
      for (int e = 0; e < 16; ++e) mpart = fma(X[16 * T4 + e], tv[64 * T4 + 16 * g + e], mpart);
    mpart += __shfl_xor(mpart, 16, kWave);
    mpart += __shfl_xor(mpart, 32, kWave);
    if (g == 0) mdst[row] -= mpart;
    // Y2: the columns d .. 2d-1 -> columns 0 .. d-1 of the state factor (any order)
    if constexpr (DD == 4) {  // the second super-tile as it stands
      double2* __restrict__ lrow = reinterpret_cast<double2*>(Ln) + size_t(st) * (2 * DC) * 64 + lane;
#pragma unroll
      for (int e = 0; e < 16; e += 2) lrow[(e / 2) * 64] = double2{X[16 + e], X[16 + e + 1]};
    } else {  // d = 32: the tiles 2, 3 of the only super-tile (registers 8 .. 15 of every group) -> the 32 columns of the groups 0, 1
#pragma unroll
      for (int e = 8; e < 16; ++e) Ln[tiled_index<DC>(row, 16 * (g & 1) + (e - 8) + 8 * (g >> 1))] = X[e];
    }
  }
  {
    // error norm over u = m[0] (rows 0 .. d-1 were updated above)
    cs.at(4);
    __syncthreads();
    cs.at(5);
    const double atol = v.atol[b], rtol = v.rtol[b];
    double part = 0.0;
    for (int j = tid; j < d; j += 256) {
      const double rel = v.err[size_t(b) * d + j] / (atol + rtol * fabs(mdst[j]));
      part += rel * rel;
    }
    part = wave_sum(part);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    const double enorm = sqrt((red[0] + red[1] + red[2] + red[3]) / double(d));
    if (tid == 0) {
      v.errnorm[b] = enorm;
      v.sigma[size_t(arg.dst) * sB + b] = v.sig_new[b];
    }
    if constexpr (CTL) {
      if (tid == 0) {  // the controller's decision for this trajectory
        TrajCtl c = ctl[b];
        StepArg a = arg;
        controller_decision(c, a, enorm, cp, false);
        args[b] = a;
        ctl[b] = c;
      }
    }
    cs.at(6);
#ifdef PNODE_TIMING_BUILD
    if (lane == 0) {  // (behind the 128 words of the LQ kernel's stamps)
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(v.S + size_t(b) * d * d) + 128 + wv * 8;
      for (int i = 0; i < 7; ++i) dbg[i] = cs.t[i];
    }
#endif
  }
}

bool dense_mfma_supported(const DenseView& v) {
  if (v.k != 1 || v.d % 16 != 0) return false;
  const int DT = v.D / 16, DD = v.d / 16;
  return (DT == 8 && DD == 2) || (DT == 12 && DD == 2) || (DT == 16 && DD == 4) || (DT == 20 && DD == 4);
}

template <int DT, int DD>
static void launch_mfma(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s, const PhaseEvents* ev,
                        const FusedControl* fc) {
  auto mark = [&](int i) {
    if (ev) (void)hipEventRecord(ev->ev[i], s);
  };
  mark(2);
  if (!v.lq_two_launch && dense_lq4_supported(v)) {
    dense_launch_lq4(v, args, K, s);  // one launch, overlapping 64-row passes (pnode_dense_lq4.hip)
  } else {
    hipLaunchKernelGGL((dense_lq2_mfma_kernel<DT, DD, true>), dim3(v.B), dim3(256), 0, s, v, args, K, 0);
    hipLaunchKernelGGL((dense_lq2_mfma_kernel<DT, DD, false>), dim3(v.B), dim3(256), 0, s, v, args, K, 1);
  }
  mark(3);
  constexpr int d = 16 * DD;
  const size_t shm1 = (size_t(d) * (2 * d + 1) + size_t(DD) * 256 + d + 256 + 4 * d + 8) * sizeof(double);  // Vf, TfF, sclF, tpp, wz, tv, red
  int dev = 0;
  (void)hipGetDevice(&dev);
  static bool attr1[64] = {};  // the opt-in to > 64 KB of dynamic LDS is per function AND per device
  if (!attr1[dev & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_correct_mfma_kernel<DT, DD, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(shm1));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_correct_mfma_kernel<DT, DD, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(shm1));
    attr1[dev & 63] = true;
  }
  if (fc)
    hipLaunchKernelGGL((dense_correct_mfma_kernel<DT, DD, true>), dim3(v.B), dim3(256), shm1, s, v, fc->args, K, fc->ctl, fc->cp);
  else
    hipLaunchKernelGGL((dense_correct_mfma_kernel<DT, DD, false>), dim3(v.B), dim3(256), shm1, s, v, const_cast<StepArg*>(args), K,
                       static_cast<TrajCtl*>(nullptr), ControlParams{});
  mark(4);
}

// Launches lq-input + LQ + correct (the caller has already launched the begin kernel and marked 0, 1).
void dense_launch_mfma_begin(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  const int DD = v.d / 16;
  if (DD == 2) hipLaunchKernelGGL((dense_begin_mfma_kernel<2>), dim3(v.B), dim3(256), 0, s, v, args, K);
  else hipLaunchKernelGGL((dense_begin_mfma_kernel<4>), dim3(v.B), dim3(256), 0, s, v, args, K);
}

void dense_launch_mfma_tail(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s,
                            const PhaseEvents* ev, const FusedControl* fc) {
  const int DT = v.D / 16, DD = v.d / 16;
  if (DT == 8 && DD == 2) launch_mfma<8, 2>(v, args, K, s, ev, fc);
  else if (DT == 12 && DD == 2) launch_mfma<12, 2>(v, args, K, s, ev, fc);
  else if (DT == 16 && DD == 4) launch_mfma<16, 4>(v, args, K, s, ev, fc);
  else if (DT == 20 && DD == 4) launch_mfma<20, 4>(v, args, K, s, ev, fc);
}

}  // namespace pnode
