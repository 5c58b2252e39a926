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
          else T[size_t(row) * D + col] = t0v[q] - w4[q];
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
        const size_t oa = size_t(c0 + i) * D + c0 + j, ob = size_t(c0 + kNB + i) * D + c0 + kNB + j;
        if (direct) {
          Ldst[tiled_index<DC>(c0 + i, c0 + j - d)] = pS[(c0 + i) / d] * psA.Tpp[i * kNB + j];
          Ldst[tiled_index<DC>(c0 + kNB + i, c0 + kNB + j - d)] = pS[(c0 + kNB + i) / d] * psB.Tpp[i * kNB + j];
        } else {
          T[oa] = psA.Tpp[i * kNB + j];
          T[ob] = psB.Tpp[i * kNB + j];
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
        double* Trow = T + size_t(j0 + r) * D + c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Trow[g + 4 * q] = ta[q];
          Trow[kNB + g + 4 * q] = tb[q];
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
  const IwpConst& K = *Kc;
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

// Second half of the measurement update: every 16-row strip of L^- = diag(p) T, first 2d columns, is rotated by the
// reflectors of F (d / 16 panels, in LDS): X Q = [Y1 | Y2]. Y1 = L^- W^T feeds the mean, Y2 = L^- Q2 is the dense part of the
// new rank-(D - d) state factor (columns 0 .. d-1 of Ls; the columns >= 2d of L^- were written by the LQ kernel as the columns
// >= d of Ls, already scaled). Then (strip block 0: the rows of the zeroth derivative) the normalised error norm.
template <int DT, int DD>
__device__ __forceinline__ void strip_update_body(const DenseView& v, const StepArg& arg, const IwpConst& K, int b,
                                                  const double* __restrict__ Vf, const double* __restrict__ TfF,
                                                  const double* __restrict__ sclF, const double* __restrict__ tv, double* red) {
  constexpr int D = 16 * DT, d = 16 * DD, W2 = 2 * d, LDF = W2 + 1, DF = W2 / 16, DC = mfma_col_tiles(DT, DD);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const double* __restrict__ T = v.R + size_t(b) * D * D;
  const size_t sB = size_t(v.B);
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * D;
  double* __restrict__ Ln = v.U + (size_t(arg.dst) * sB + b) * D * D;
  for (int st = wv; st < DT; st += 4) {
    const int row = 16 * st + r;
    const int iblk = (16 * st) / d;
    double pr;
    {
      double p[kMaxN], pinv[kMaxN];
      precond_rt(v.n, arg.dt, K, p, pinv);
      pr = p[0];
#pragma unroll
      for (int i = 1; i < kMaxN; ++i) pr = (i == iblk) ? p[i] : pr;
    }
    // register strip in the HBM layout  X[16 T + 4 j + q] = p_r T[row][64 T + 16 g + 4 j + q]  (128 B per lane
    // and super-tile); T's upper triangle is stored as zeros, so no predicate is needed
    double X[W2 / 4];
    const double* __restrict__ trow = T + size_t(row) * D + 16 * g;
#pragma unroll
    for (int T4 = 0; T4 < W2 / 64; ++T4)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const double2 t2 = *reinterpret_cast<const double2*>(trow + 64 * T4 + e);
        X[16 * T4 + e] = pr * t2.x;
        X[16 * T4 + e + 1] = pr * t2.y;
      }
#pragma unroll
    for (int pnl = 0; pnl < DD; ++pnl) {
      double4_t unused;
      apply_panel_to_strip<DF, LDF>(X, Vf + pnl * kNB * LDF, TfF + pnl * kNB * kNB, sclF + pnl * kNB, double4_t{0.0, 0.0, 0.0, 0.0},
                                    unused, r, g);
    }
    // mean: m -= Y1 (C^-1 z); tv holds C^-1 z in its first d entries and zeros beyond
    double mpart = 0.0;
#pragma unroll
    for (int T4 = 0; T4 < W2 / 64; ++T4)
#pragma unroll
      for (int e = 0; e < 16; ++e) mpart = fma(X[16 * T4 + e], tv[64 * T4 + 16 * g + e], mpart);
    mpart += __shfl_xor(mpart, 16, kWave);
    mpart += __shfl_xor(mpart, 32, kWave);
    if (g == 0) mdst[row] -= mpart;
    {
      // Y2 = columns d .. 2d-1 of the rotated strip -> columns 0 .. d-1 of the state factor, in the register order of its strips
#pragma unroll
      for (int T4 = 0; T4 < W2 / 64; ++T4)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const int col = 64 * T4 + 16 * g + e - d;
          if (col >= 0) *reinterpret_cast<double2*>(Ln + tiled_index<DC>(row, col)) = double2{X[16 * T4 + e], X[16 * T4 + e + 1]};
        }
    }
  }
  {
    // error norm over u = m[0] (rows 0 .. d-1 were updated above)
    __syncthreads();
    const double atol = v.atol[b], rtol = v.rtol[b];
    double part = 0.0;
    for (int j = tid; j < d; j += 256) {
      const double rel = v.err[size_t(b) * d + j] / (atol + rtol * fabs(mdst[j]));
      part += rel * rel;
    }
    part = wave_sum(part);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (tid == 0) {
      v.errnorm[b] = sqrt((red[0] + red[1] + red[2] + red[3]) / double(d));
      v.sigma[size_t(arg.dst) * sB + b] = v.sig_new[b];
    }
  }
}

// Measurement update, ONE kernel per trajectory, by orthogonal transformations only (the reference's square-root correction:
// QR of (H L^-)^T, strategy_filter + correction_ts1, /root/reference/experiments/1_van_der_pol/vdp.py:64-65):
//   F = H L^- (d x 2d: L^- is lower triangular),  F = [C 0] Q^T by a blocked row-Householder LQ (S = C C^T is never formed),
//   L^- Q = [Y1 | Y2 | columns >= 2d]:  gain K = Y1 C^-1, mean m = m^- - Y1 (C^-1 z), and [Y2 | columns >= 2d] is a factor of the
//   updated covariance with D - d columns -- the observation is noise-free, the d directions of Y1 are known exactly afterwards.
// The panels of the LQ (wavefront p sweeps strip p of F in registers, pnode_lq_sweep.hpp) stay in LDS for the strips of L^-.
template <int DT, int DD>
__global__ __launch_bounds__(256, 2) void dense_correct_mfma_kernel(DenseView v, const StepArg* __restrict__ args,
                                                              const IwpConst* __restrict__ Kc) {
  constexpr int D = 16 * DT, d = 16 * DD, W2 = 2 * d, LD = d + 1, LDF = W2 + 1, DF = W2 / 16;
  static_assert(DD <= 4, "one wavefront per 16-row strip of F; the pivot columns of a strip lie in the first super-tile");
  // One LDS buffer, two lives: (1) J and T[0:d, 0:d] staged for the MFMA that forms F (2) F itself, d x 2d, whose strips become
  // the reflector panels in place.
  extern __shared__ double gbuf[];
  double* Js = gbuf;            // [d][LD]      life 1
  double* Ts = gbuf + d * LD;   // [d][LD]      life 1
  double* Fl = gbuf;            // [d][LDF]     life 2: F, then the reflector rows (unscaled, pivot entries 1 / scl)
  double* TfF = gbuf + 2 * d * LD;     // [DD][16][16]  compact-WY factors of the panels   (beyond both lives)
  double* sclF = TfF + DD * kNB * kNB; // [d]
  double* tpp = sclF + d;              // [16][16]  diagonal block of C of the sweeping wavefront
  double* wz = tpp + kNB * kNB;        // [2d]  C^-1 z, zeros beyond d
  double* red = wz + W2;               // [4]
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const IwpConst& K = *Kc;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
#ifdef PNODE_TIMING_BUILD
  // diagnostic build only: cycle sums by phase of wavefront 0, written behind the LQ's stamps in the S scratch
  unsigned long long ck_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ck_prev = __builtin_amdgcn_s_memtime();
#define CK_STAMP(slot)                                             \
  do {                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    ck_acc[slot] += now_ - ck_prev;                                \
    ck_prev = now_;                                                \
  } while (0)
#else
#define CK_STAMP(slot) \
  do {                 \
  } while (0)
#endif
  double p[kMaxN], pinv[kMaxN];
  precond_rt(v.n, arg.dt, K, p, pinv);
  const double p0 = p[0], p1 = p[1];
  const double* __restrict__ T = v.R + size_t(b) * D * D;
  const double* __restrict__ Jg = v.J + size_t(b) * d * d;
  // ---- life 1: stage J and the lower-triangular T[0:d, 0:d] (coalesced), form F[:, 0:d] = p1 T[d:2d, 0:d] - p0 J T[0:d, 0:d]
  for (int idx = tid; idx < d * d; idx += 256) {
    const int a = idx / d, c = idx % d;
    Js[a * LD + c] = -p0 * Jg[idx];
    Ts[a * LD + c] = T[size_t(a) * D + c];  // (the upper part is stored as zeros)
  }
  for (int j = tid; j < W2; j += 256) wz[j] = j < d ? v.z[size_t(b) * d + j] : 0.0;
  constexpr int NT = (DD * DD + 3) / 4;  // F1 tiles per wavefront
  double4_t f1[NT];
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int tile = wv + 4 * it;
    if (tile < DD * DD) {
      const int ta = tile / DD, tc = tile % DD;
#pragma unroll
      for (int q = 0; q < 4; ++q) f1[it][q] = p1 * T[size_t(d + 16 * ta + g + 4 * q) * D + 16 * tc + r];
    }
  }
  __syncthreads();
  CK_STAMP(0);
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int tile = wv + 4 * it;
    if (tile < DD * DD) {
      const int ta = tile / DD, tc = tile % DD;
      double4_t acc = f1[it];
#pragma unroll
      for (int s = 0; s < d / 4; ++s) acc = MFMA64(Js[(16 * ta + r) * LD + 4 * s + g], Ts[(4 * s + g) * LD + 16 * tc + r], acc);
      f1[it] = acc;
    }
  }
  __syncthreads();  // everyone is done with J and T in LDS
  CK_STAMP(1);
  // ---- life 2: F in LDS; F[:, d:2d] = p1 T[d:2d, d:2d] (lower triangular block)
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int tile = wv + 4 * it;
    if (tile < DD * DD) {
      const int ta = tile / DD, tc = tile % DD;
#pragma unroll
      for (int q = 0; q < 4; ++q) Fl[(16 * ta + g + 4 * q) * LDF + 16 * tc + r] = f1[it][q];
    }
  }
  for (int idx = tid; idx < d * d; idx += 256) {
    const int a = idx / d, c = idx % d;
    Fl[a * LDF + d + c] = (c <= a) ? p1 * T[size_t(d + a) * D + d + c] : 0.0;
  }
  __syncthreads();
  CK_STAMP(2);
  // ---- blocked row-Householder LQ of F: wavefront w owns strip w (16 rows, all 2d columns) in registers
  double X[4 * DF];
  const bool owner = wv < DD;
  if (owner) {
#pragma unroll
    for (int T4 = 0; T4 < DF / 4; ++T4)
#pragma unroll
      for (int e = 0; e < 16; ++e) X[16 * T4 + e] = Fl[(kNB * wv + r) * LDF + 64 * T4 + 16 * g + e];
  }
  CK_STAMP(3);
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
      double* tpp_row = tpp + r * kNB;
      pivot_sweep_inplace<0, DF>(X, tfrow, tpp_row, myscl, wv, r, g);
#pragma unroll
      for (int T4 = 0; T4 < DF / 4; ++T4)
#pragma unroll
        for (int e = 0; e < 16; ++e) Fl[(kNB * wv + r) * LDF + 64 * T4 + 16 * g + e] = X[16 * T4 + e];
      if (g == 0) {
        sclF[kNB * wv + r] = myscl;
#pragma unroll
        for (int m = 0; m < kNB; ++m) TfF[wv * kNB * kNB + r * kNB + m] = tfrow[m];
      }
      // forward substitution with the diagonal block (lower triangular, row r in tpp_row): lane r of every group alike
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double crow[kNB];
#pragma unroll
      for (int m = 0; m < kNB; ++m) crow[m] = tpp_row[m];
      double xsol = 0.0;
      tri_forward16<0>(rhs, xsol, crow, r);
      if (g == 0) wz[kNB * wv + r] = xsol;
    }
    __syncthreads();
    if (owner && wv > pnl) {
      double4_t unused;
      apply_panel_to_strip<DF, LDF>(X, Fl + pnl * kNB * LDF, TfF + pnl * kNB * kNB, sclF + pnl * kNB, double4_t{0.0, 0.0, 0.0, 0.0},
                                    unused, r, g);
    }
  }
  __syncthreads();
  CK_STAMP(6);
  strip_update_body<DT, DD>(v, arg, K, b, Fl, TfF, sclF, wz, red);
  CK_STAMP(7);
#ifdef PNODE_TIMING_BUILD
  if (tid == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(v.S + size_t(b) * d * d) + 64;
    for (int i = 0; i < 8; ++i) dbg[i] = ck_acc[i];
  }
#endif
}

bool dense_mfma_supported(const DenseView& v) {
  if (v.k != 1 || v.d % 16 != 0) return false;
  const int DT = v.D / 16, DD = v.d / 16;
  return (DT == 8 && DD == 2) || (DT == 12 && DD == 2) || (DT == 16 && DD == 4) || (DT == 20 && DD == 4);
}

template <int DT, int DD>
static void launch_mfma(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s, const PhaseEvents* ev) {
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
  static_assert(DD <= 4, "the error norm is computed by strip block 0, which must cover the d rows of u");
  // 2 d (d + 1) >= d (2d + 1) for the two lives of the big buffer, then TfF, sclF, tpp, wz, red: see the kernel
  const size_t shm1 = (size_t(2) * d * (d + 1) + size_t(DD) * 256 + d + 256 + 2 * d + 8) * sizeof(double);
  int dev = 0;
  (void)hipGetDevice(&dev);
  static bool attr1[64] = {};  // the opt-in to > 64 KB of dynamic LDS is per function AND per device
  if (!attr1[dev & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_correct_mfma_kernel<DT, DD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(shm1));
    attr1[dev & 63] = true;
  }
  hipLaunchKernelGGL((dense_correct_mfma_kernel<DT, DD>), dim3(v.B), dim3(256), shm1, s, v, args, K);
  mark(4);
}

// Launches lq-input + LQ + correct (the caller has already launched the begin kernel and marked 0, 1).
void dense_launch_mfma_begin(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  const int DD = v.d / 16;
  if (DD == 2) hipLaunchKernelGGL((dense_begin_mfma_kernel<2>), dim3(v.B), dim3(256), 0, s, v, args, K);
  else hipLaunchKernelGGL((dense_begin_mfma_kernel<4>), dim3(v.B), dim3(256), 0, s, v, args, K);
}

void dense_launch_mfma_tail(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s,
                            const PhaseEvents* ev) {
  const int DT = v.D / 16, DD = v.d / 16;
  if (DT == 8 && DD == 2) launch_mfma<8, 2>(v, args, K, s, ev);
  else if (DT == 12 && DD == 2) launch_mfma<12, 2>(v, args, K, s, ev);
  else if (DT == 16 && DD == 4) launch_mfma<16, 4>(v, args, K, s, ev);
  else if (DT == 20 && DD == 4) launch_mfma<20, 4>(v, args, K, s, ev);
}

}  // namespace pnode
