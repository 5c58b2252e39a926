// Time update of the dense EKF1 fast path, ONE launch, column-split: blocked row-Householder LQ of the D x 2D stack
// [T0 | Mt], one workgroup of 8 wavefronts per trajectory, two groups of four wavefronts (A = 0..3, B = 4..7).
//
//   * A group holds ONE 64-row block of Mt in registers, split by COLUMNS: wavefront j of the group owns a quarter of the
//     columns of all 64 rows (4 strips x D/16 doubles per lane, in the MFMA operand layout of pnode_mfma_common.hpp, so the
//     HBM image of a strip is read and written in 1 KB pieces exactly as before). Every contraction over the columns
//     (pivot dot products, G = X V^T) is a sum of four per-wavefront partials, exchanged through LDS and added in a fixed
//     order by every wavefront (all four hold bit-identical totals and take identical decisions).
//   * The groups alternate as the PANEL group of the 64-row passes. In pass P the panel group holds block P and factorises
//     its four 16-row panels (sweep in registers with DPP-operand FMAs; each wavefront sweeps its own columns, so a panel costs
//     a quarter of the one-wavefront sweep plus 16 small exchanges); the other group holds block P + 1 and applies every panel
//     as it is published, so that it can start the chain of pass P + 1 the moment panel 3 is applied, while the former panel
//     group streams the remaining blocks of pass P (last of them block P + 2, which it keeps for the pass after next).
//     The chain of the 4 D/64 sweeps therefore runs beside the trailing updates of the previous pass instead of after them,
//     and at D = 256 only ONE 64-row block makes one round trip through HBM (none at D <= 192).
//   * Reflectors live in LDS (one 16 x D slot per panel of the running pass, V = [e ; scl x] with x unscaled); a slot is
//     rewritten by the next pass only after the four consumers of the previous pass have counted themselves out.
//   * No workgroup barrier after the prologue: every wait is a spin on an LDS word with an iteration limit (a wavefront that
//     runs into it raises the abort word, every other wait then falls through and the trajectory's output is NaN).
//
// Reference anchors: strategy_filter (/root/reference/experiments/1_van_der_pol/vdp.py:65) as probdiffeq runs it (sqrt
// time update by QR of the stacked factor); arithmetic restated in oracle/pn_oracle.py (predict_sqrt).
#include "pnode_mfma_common.hpp"

namespace pnode {
namespace lqc {

constexpr long long kSent = static_cast<long long>(0xFFF4DEADBEEF5EEDull);  // signalling NaN pattern no arithmetic produces
constexpr int kSpinLimit = 1 << 21;

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// The words the wavefronts talk through are accessed through pointers that CARRY the LDS address space: a volatile access
// through a generic pointer is not rewritten by the address-space inference and becomes a flat_load / flat_store (the slow
// path to LDS, several hundred cycles per round trip -- measured: 1000 cycles per pivot exchange instead of 250).
using lds_ll = __attribute__((address_space(3))) long long;
using lds_int = __attribute__((address_space(3))) int;

template <int C>
__device__ __forceinline__ double bcast16(double v) {  // lane C of every 16-lane DPP row -> all lanes of that row
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + C, 0xf, 0xf, false);
}

// acc_i += x_e(lane C of the row) * x_e(this lane), four elements: the dot product of every row with row C, partial over
// this lane's columns. (s_nop 1: a VGPR written by a VALU instruction needs two wait states before a DPP read.)
template <int C>
__device__ __forceinline__ void dot4(double& a0, double& a1, double& a2, double& a3, double x0, double x1, double x2, double x3) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %4, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %5, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %6, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %7, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "i"(C));
}
// x_e += x_e(lane C of the row) * nws  (nws = 0 on the lanes whose row is not updated, lane C included)
template <int C>
__device__ __forceinline__ void upd4(double& x0, double& x1, double& x2, double& x3, double nws) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
      : "v"(nws), "i"(C));
}

// Sum over the four 16-lane DPP rows of the wavefront; every lane receives the total, formed in the same order.
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

template <int M, int C>
__device__ __forceinline__ void tf_accumulate(double& acc, double u, const double (&tfrow)[kNB]) {
  if constexpr (M < C) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(tfrow[M]), "i"(M));
    tf_accumulate<M + 1, C>(acc, u, tfrow);
  }
}

// Cycle stamps of the diagnostic build (make timing); empty otherwise.
struct Stamps {
#ifdef PNODE_TIMING_BUILD
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prev = 0;
  __device__ __forceinline__ void start() { prev = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void count(int slot, int n) { acc[slot] += n; }
  __device__ __forceinline__ void mark(int slot) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    acc[slot] += now - prev;
    prev = now;
  }
#else
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void count(int, int) {}
#endif
};

// What a wavefront needs to talk to the three others of its group.
struct GroupLink {
  volatile lds_ll* ring;  // [3][4][16] partial dot products of the running pivot, kSent = not there yet
  volatile lds_int* abort_word;
  int slot;                  // ring slot of the group's next pivot
  int j;                     // my column quarter
};

// Pivot C of the sweep of the panel strip X[0 .. E). On entry rows 0 .. C are final.
//   v_C = [e_C ; scl_C x_C]  (x_C stays unscaled in the registers),  H_C = I - tau_C v_C^T v_C.
// tpp_row[C] <- T[row r][column C of the panel]  (beta on the diagonal, -w below, 0 above), staged in LDS by quarter 0
// tfrow[C] <- Tf[r][C] of the compact-WY factor, H_0 ... H_C = I - V^T Tf V  (forward recurrence, dlarft)
template <int C, int E>
__device__ __forceinline__ void pivot_step(double (&X)[4 * E], double (&tfrow)[kNB], double* __restrict__ tpp_row, double& myscl,
                                           double alpha, int r_, GroupLink& gl, Stamps& st) {
  int r = r_;
  asm volatile("" : "+v"(r));  // opaque: the per-pivot lane predicates are formed here, not hoisted out of the pass loop
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
  for (int e = 0; e < E; e += 4) dot4<C>(a0, a1, a2, a3, X[e], X[e + 1], X[e + 2], X[e + 3]);
  const double part = sum_over_groups((a0 + a1) + (a2 + a3));  // every lane of row r: x_r . x_C over my columns
  // ---- exchange with the three other quarters: the datum is its own flag
  st.mark(1);
  gl.ring[gl.slot * 64 + gl.j * 16 + r] = __double_as_longlong(part);
  long long b0, b1, b2, b3;
  int spins = 0;
  while (true) {
    b0 = gl.ring[gl.slot * 64 + r];
    b1 = gl.ring[gl.slot * 64 + 16 + r];
    b2 = gl.ring[gl.slot * 64 + 32 + r];
    b3 = gl.ring[gl.slot * 64 + 48 + r];
    const bool missing = (b0 == kSent) | (b1 == kSent) | (b2 == kSent) | (b3 == kSent);
    if (!__any(missing)) break;
    if ((++spins & 255) == 0 && (spins > kSpinLimit || __builtin_amdgcn_readfirstlane(*gl.abort_word))) {
      *gl.abort_word = 1;
      break;
    }
  }
  st.count(0, spins + 1);
  // every wavefront has read the previous pivot's slot before it wrote this one: my entry there can be re-armed
  const int prev = (gl.slot == 0) ? 2 : gl.slot - 1;
  gl.ring[prev * 64 + gl.j * 16 + r] = kSent;
  gl.slot = (gl.slot == 2) ? 0 : gl.slot + 1;
  const double dotv = (__longlong_as_double(b0) + __longlong_as_double(b1)) + (__longlong_as_double(b2) + __longlong_as_double(b3));
  st.mark(2);
  const double xx = bcast16<C>(dotv);  // |x_C|^2
  double tauc = 0.0, scale = 0.0, beta = alpha;
  {
    const double nrm2 = fma(alpha, alpha, xx);
    const double rs = fast_rsqrt(nrm2);
    const double sgn = copysign(1.0, alpha);
    const double t1 = fma(fabs(alpha), rs, 1.0);
    const bool refl = xx != 0.0;  // (NaN != 0: a NaN row propagates, like the reference's QR)
    beta = refl ? -sgn * nrm2 * rs : alpha;
    tauc = refl ? t1 : 0.0;
    scale = refl ? sgn * rs * fast_rcp(t1) : 0.0;
  }
  const double w = tauc * scale * dotv;  // (the rows of one panel have no T0 entry in each other's columns)
  const double nws = (r > C) ? -(w * scale) : 0.0;
  if (gl.j == 0) tpp_row[C] = (r > C) ? -w : ((r == C) ? beta : 0.0);  // (the four lanes of a row store the same value)
  {
    const double u = (r < C) ? myscl * scale * dotv : 0.0;  // v_r . v_C for the finished reflectors r < C
    double acc = 0.0;
    tf_accumulate<0, C>(acc, u, tfrow);
    tfrow[C] = (r < C) ? -tauc * acc : ((r == C) ? tauc : 0.0);
  }
  myscl = (r == C) ? scale : myscl;
  asm volatile("" : "+v"(tfrow[C]), "+v"(myscl));  // pin the recurrence inside its pivot
  st.mark(3);
#pragma unroll
  for (int e = 0; e < E; e += 4) upd4<C>(X[e], X[e + 1], X[e + 2], X[e + 3], nws);
  st.mark(4);
}

template <int C, int E>
__device__ __forceinline__ void pivot_sweep(double (&X)[4 * E], double (&tfrow)[kNB], double* __restrict__ tpp_row, double& myscl,
                                            double alpha, int r, GroupLink& gl, Stamps& st) {
  if constexpr (C < kNB) {
    pivot_step<C, E>(X, tfrow, tpp_row, myscl, alpha, r, gl, st);
    pivot_sweep<C + 1, E>(X, tfrow, tpp_row, myscl, alpha, r, gl, st);
  }
}

}  // namespace lqc

template <int DT, int DD>
__global__ __launch_bounds__(512) void dense_lqc_mfma_kernel(const DenseView v, const StepArg* __restrict__ args,
                                                              const IwpConst* __restrict__ Kc) {
  using namespace lqc;
  static_assert(DT % 4 == 0, "whole 64-row blocks");
  static_assert((2 * 16 * DD) % 64 == 0, "a pass is all below or all beyond column 2d");
  constexpr int D = 16 * DT, d = 16 * DD, E = DT, NBLK = DT / 4;
  constexpr int LDV = D + 1;  // odd: the A-operand reads of the G phase (16 rows x 4 column groups) spread over the banks
  __shared__ double Vx[4 * kNB * LDV];
  __shared__ double TfS[4 * kNB * kNB];
  __shared__ double sclS[4 * kNB];
  __shared__ double TppS[kNB * kNB];          // diagonal block of T of the running panel, staged for a coalesced write
  __shared__ double GxS[2 * 4 * 64 * 4];      // [group][quarter][lane][4] partial G of the running strip
  __shared__ long long ringS[2 * 3 * 64];     // [group][slot][quarter][row]
  __shared__ double coef[kMaxN * kMaxN];
  __shared__ double pS[kMaxN];
  __shared__ int pubS[4];     // += 1 by every quarter that has published panel k (V columns, Tf, scl): 4 (P + 1) = pass P complete
  __shared__ int freeS[4];    // += 1 by every wavefront of the panel group of pass P after its last read of slot k
  __shared__ int arrS[2], rdS[2];  // G exchange of a group: partials written / totals read
  __shared__ int storedS[4];  // quarter j: number of passes whose trailing blocks are back in HBM
  __shared__ int abortS;

  const int b = blockIdx.x;
  const StepArg sarg = args[b];
  if (!sarg.active) return;
  const IwpConst& K = *Kc;
  const int nn = v.n;
  const double sigma = v.sig_new[b];
  double* __restrict__ Mt = v.M + size_t(b) * D * D;
  double* __restrict__ T = v.R + size_t(b) * D * D;
  const double* __restrict__ Lsrc = v.U + (size_t(sarg.src) * v.B + b) * D * D;
  double* __restrict__ Ldst = v.U + (size_t(sarg.dst) * v.B + b) * D * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gi = wv >> 2, j = wv & 3;
  const int r = lane & 15;
  if (tid == 0) {
    // Mt = (Abar P^-1 (x) I) L is formed while the strips are read: coef[i][k] = Abar[i][k] / p_k
    double pl[kMaxN], pil[kMaxN];
    precond_rt(nn, sarg.dt, K, pl, pil);
    for (int i = 0; i < nn; ++i)
      for (int k = 0; k < nn; ++k) coef[i * kMaxN + k] = (k >= i) ? K.a[i][k] * pil[k] : 0.0;
    for (int i = 0; i < kMaxN; ++i) pS[i] = (i < nn) ? pl[i] : 0.0;
    for (int k = 0; k < 4; ++k) pubS[k] = freeS[k] = storedS[k] = 0;
    arrS[0] = arrS[1] = rdS[0] = rdS[1] = 0;
    abortS = 0;
  }
  if (tid < 2 * 3 * 64) ringS[tid] = kSent;
  __syncthreads();  // the only workgroup barrier

  // chunk ci (four consecutive columns per lane) of my quarter: columns col0[ci] + 16 g + q, pair (E/2) j + 2 ci + q/2 of the
  // strip's HBM image
  int col0[E / 4];
#pragma unroll
  for (int ci = 0; ci < E / 4; ++ci) {
    const int c = (E / 4) * j + ci;
    col0[ci] = 64 * (c >> 2) + 4 * (c & 3);
  }
  volatile lds_int* abort_word = (volatile lds_int*)&abortS;
  auto spin_ge = [&](int* f_, int target) {
    volatile lds_int* f = (volatile lds_int*)f_;
    int n = 0;
    while (__builtin_amdgcn_readfirstlane(*f) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++n & 255) == 0 && (n > kSpinLimit || __builtin_amdgcn_readfirstlane(*abort_word))) {
        *abort_word = 1;
        break;
      }
    }
    asm volatile("" ::: "memory");
  };
  auto bump = [&](int* f) {  // everything this wavefront sent to LDS before is visible to whoever sees the increment
    lds_fence();
    if (lane == 0) __hip_atomic_fetch_add((lds_int*)f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  GroupLink gl;
  gl.ring = (volatile lds_ll*)(ringS + gi * 3 * 64);
  gl.abort_word = abort_word;
  gl.slot = 0;
  gl.j = j;
  int gseq = 0;  // G exchanges of my group so far
  Stamps st;
  double* Gx = GxS + gi * 4 * 64 * 4;

  // sum over the four quarters of a C/D fragment (fixed order: every quarter obtains the same bits)
  auto exchange_g = [&](double4_t part) {
    spin_ge(&rdS[gi], 4 * gseq);  // the previous totals have been read by all four
    double2* mine = reinterpret_cast<double2*>(Gx + (j * 64 + lane) * 4);
    mine[0] = double2{part[0], part[1]};
    mine[1] = double2{part[2], part[3]};
    bump(&arrS[gi]);
    spin_ge(&arrS[gi], 4 * (gseq + 1));
    double4_t tot;
    {
      const double* p0 = Gx + lane * 4;  // (ordinary loads: the spin above ends in a compiler barrier)
      double4_t q0, q1, q2, q3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q0[i] = p0[i];
        q1[i] = p0[256 + i];
        q2[i] = p0[512 + i];
        q3[i] = p0[768 + i];
      }
      tot = (q0 + q1) + (q2 + q3);
    }
    bump(&rdS[gi]);
    ++gseq;
    return tot;
  };

  // (lane coordinates made opaque per call site: otherwise every LDS / HBM address of every call is hoisted out of the pass
  // loop as a loop invariant and the hoisted addresses alone overflow the register file)
  auto opaque = [](int x) {
    asm volatile("" : "+v"(x));
    return x;
  };
  auto t0_tile = [&](int row0, int colp, int r, int g) {
    // T0[row0 + r][colp + g + 4q] = sigma chol(Qbar)[ib][jb] where the intra-block indices agree
    double4_t t;
    const int ib = row0 / d, jb = colp / d;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      t[q] = (((row0 + r) % d) == ((colp + g + 4 * q) % d) && jb <= ib) ? sigma * K.qsT[jb][ib] : 0.0;
    return t;
  };

  double X[4 * E];  // the block this wavefront's group holds: strip s, element e <-> row 16 s + r, column col0[e / 4] + 16 g + e % 4

  auto load_block = [&](int blk, bool from_state) {
    const int lane = opaque(tid & 63);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int st = 4 * blk + s;
      if (from_state) {
        // Mt rows of derivative block ib: sum_{k >= ib} coef[ib][k] L[k d + q]; k runs DOWNWARDS so that the wavefronts of both
        // groups, which start together, ask for the same rows of L at the same time
        const int ib = (kNB * st) / d;
#pragma unroll
        for (int e = 0; e < E; ++e) X[s * E + e] = 0.0;
#pragma unroll 1
        for (int k = nn - 1; k >= ib; --k) {
          const double ck = coef[ib * kMaxN + k];
          const double2* __restrict__ lrow = reinterpret_cast<const double2*>(Lsrc) + (size_t((k * d + (kNB * st) % d) / kNB) * (2 * DT) + (E / 2) * j) * 64 + lane;
          double2 tmp[E / 2];
#pragma unroll
          for (int i = 0; i < E / 2; ++i) tmp[i] = lrow[i * 64];
#pragma unroll
          for (int i = 0; i < E / 2; ++i) {
            X[s * E + 2 * i] = fma(ck, tmp[i].x, X[s * E + 2 * i]);
            X[s * E + 2 * i + 1] = fma(ck, tmp[i].y, X[s * E + 2 * i + 1]);
          }
        }
      } else {
        const double2* __restrict__ src = reinterpret_cast<const double2*>(Mt) + (size_t(st) * (2 * DT) + (E / 2) * j) * 64 + lane;
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
          const double2 t2 = src[i * 64];
          X[s * E + 2 * i] = t2.x;
          X[s * E + 2 * i + 1] = t2.y;
        }
      }
    }
  };
  auto store_block = [&](int blk) {
    const int lane = opaque(tid & 63);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double2* __restrict__ dst = reinterpret_cast<double2*>(Mt) + (size_t(4 * blk + s) * (2 * DT) + (E / 2) * j) * 64 + lane;
#pragma unroll
      for (int i = 0; i < E / 2; ++i) dst[i * 64] = double2{X[s * E + 2 * i], X[s * E + 2 * i + 1]};
    }
  };

  // Columns >= 2d of L^- are not touched by the measurement update (F = H L^- has 2d non-zero columns): the passes that
  // produce them write p_row * T straight into the destination slot of the state factor.
  auto is_direct = [&](int pass) { return 64 * pass >= 2 * d; };
  auto write_tile = [&](double4_t tv, int j0, int cp, bool direct, int r, int g) {  // T[j0 + r][cp + g + 4q] = tv[q]
    tv = transpose_groups(tv);  // lane (r, g): columns cp + 4 g .. cp + 4 g + 3 of row j0 + r
    const double osc = direct ? pS[(j0 + r) / d] : 1.0;
    const double2 lo2{osc * tv[0], osc * tv[1]}, hi2{osc * tv[2], osc * tv[3]};
    if (direct) {  // strip order of the state factor: two pairs per lane, 16 lanes of a pair contiguous
      double* base = Ldst + tiled_index<DT>(j0 + r, cp + 4 * g);
      *reinterpret_cast<double2*>(base) = lo2;
      *reinterpret_cast<double2*>(base + 128) = hi2;  // next pair: + 64 lanes x 2 doubles
    } else {
      double* Trow = T + size_t(j0 + r) * D + cp + 4 * g;
      *reinterpret_cast<double2*>(Trow) = lo2;
      *reinterpret_cast<double2*>(Trow + 2) = hi2;
    }
  };
  // Panel k of pass `pass` applied to the strip Xs (E registers, rows j0 ..): partial G over my columns, exchange, W = G Tf,
  // update of my columns. panel_regs: the panel strip is my own registers Xp (the A operand of G is the lane's own value).
  auto apply_strip = [&](double* Xs, const double* Xp, bool panel_regs, int pass, int k, int j0, const double (&tf4)[4], double4_t sc, bool writer) {
    const int r = opaque(lane & 15), g = opaque(lane >> 4);
    const int coff = 16 * (r & 3) + (r >> 2);
    const double* Vp = Vx + k * kNB * LDV;
    const int cp = 64 * pass + kNB * k;
    double4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
#pragma unroll
    for (int ci = 0; ci < E / 4; ++ci) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double av = panel_regs ? Xp[4 * ci + q] : Vp[r * LDV + col0[ci] + 16 * g + q];
        if (ci & 1)
          a1 = MFMA64(av, Xs[4 * ci + q], a1);
        else
          a0 = MFMA64(av, Xs[4 * ci + q], a0);
      }
    }
    st.mark(5);
    const double4_t tot = exchange_g(a0 + a1);
    st.mark(6);
    const double4_t t0v = t0_tile(j0, cp, r, g);
    const double4_t acc = t0v + tot * sc;  // G = T0 + (X raw^T) diag(scl)
    double4_t w4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w4 = MFMA64(tf4[kk], acc[kk], w4);
    if (writer) write_tile(t0v - w4, j0, cp, is_direct(pass), r, g);
    w4 = w4 * sc;  // the update below multiplies by the unscaled rows
#pragma unroll
    for (int ci = 0; ci < E / 4; ++ci) {
      double4_t c4 = {Xs[4 * ci], Xs[4 * ci + 1], Xs[4 * ci + 2], Xs[4 * ci + 3]};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) c4 = MFMA64(-Vp[(4 * kk + g) * LDV + col0[ci] + coff], w4[kk], c4);
      Xs[4 * ci] = c4[0];
      Xs[4 * ci + 1] = c4[1];
      Xs[4 * ci + 2] = c4[2];
      Xs[4 * ci + 3] = c4[3];
    }
    st.mark(7);
  };
  auto read_factor = [&](int k, double (&tf4)[4], double4_t& sc) {
    const int r = opaque(lane & 15), g = opaque(lane >> 4);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) tf4[kk] = TfS[k * kNB * kNB + (4 * kk + g) * kNB + r];
#pragma unroll
    for (int q = 0; q < 4; ++q) sc[q] = sclS[k * kNB + g + 4 * q];
  };
  // all four panels of pass `pass` applied to the block I hold (rows 64 blk ..)
  auto apply_pass_to_block = [&](int blk, int pass, bool wait_pub, bool last_use) {
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      if (wait_pub) spin_ge(&pubS[k], 4 * (pass + 1));
      double tf4[4];
      double4_t sc;
      read_factor(k, tf4, sc);
#pragma unroll
      for (int s = 0; s < 4; ++s) apply_strip(X + s * E, nullptr, false, pass, k, 64 * blk + kNB * s, tf4, sc, j == s);
      if (last_use) bump(&freeS[k]);
    }
  };

  st.start();
  // Prologue: group A starts with block 0, group B with block 1, to which it applies the panels of pass 0 as they come. After
  // that both groups run the same straight loop over THEIR passes P = gi, gi + 2, ..: chain of pass P on the block held,
  // the other blocks of pass P (block P + 2 kept), then the panels of pass P + 1 (the other group's) applied to block P + 2.
  // (One straight body: with the two roles as the arms of a branch inside the loop the register allocator keeps part of
  // the 128-register block in scratch around the join.)
  load_block(gi, true);
  
  if (gi == 1) apply_pass_to_block(1, 0, true, false);
  
#pragma unroll 1
  for (int P = gi; P < NBLK; P += 2) {
    const int c0 = 64 * P;
    const bool direct = is_direct(P);
    const bool has_trailing = P + 2 < NBLK;
    // ================= chain of pass P: the four panels of block P
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const int cp = c0 + kNB * k;
      {
        const double alpha = sigma * K.qsT[cp / d][cp / d];
        double tfrow[kNB], myscl = 0.0;
#pragma unroll
        for (int m = 0; m < kNB; ++m) tfrow[m] = 0.0;
        double* tpp_row = TppS + r * kNB;
        __builtin_amdgcn_s_setprio(3);
        pivot_sweep<0, E>(X, tfrow, tpp_row, myscl, alpha, r, gl, st);
        __builtin_amdgcn_s_setprio(0);
        
        // ---- publish: my columns of V, and (identically from every quarter) Tf and scl
        if (P > 0) spin_ge(&freeS[k], 4 * P);  // the previous pass is done with this slot
        
        const int r = opaque(lane & 15), g = opaque(lane >> 4);
        double* Vp = Vx + k * kNB * LDV;
#pragma unroll
        for (int ci = 0; ci < E / 4; ++ci)
#pragma unroll
          for (int q = 0; q < 4; ++q) Vp[r * LDV + col0[ci] + 16 * g + q] = X[4 * ci + q];
        if (g == 0) {
          sclS[k * kNB + r] = myscl;
#pragma unroll
          for (int m = 0; m < kNB; ++m) TfS[k * kNB * kNB + r * kNB + m] = tfrow[m];
        }
        // diagonal block of T (quarter 0 staged it in TppS): lane (r, g) writes columns 4g .. 4g+3 of row r. Read BEFORE the
        // publication counter moves: once panel 3 is public the other group may start its own sweep and restage TppS.
        double o4[4] = {0.0, 0.0, 0.0, 0.0};
        if (j == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o4[q] = TppS[r * kNB + 4 * g + q];
        }
        bump(&pubS[k]);
        if (j == 0) {
          const int row = cp + r;
          const double osc = direct ? pS[row / d] : 1.0;
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            const double2 t2{osc * o4[q], osc * o4[q + 1]};
            double* dstp = direct ? Ldst + tiled_index<DT>(row, cp + 4 * g + q) : T + size_t(row) * D + cp + 4 * g + q;
            *reinterpret_cast<double2*>(dstp) = t2;  // (4 g + q is even: the pair is contiguous in both layouts)
          }
        }
      }
      
      if (k < 3) {
        // ---- the rest of my block: strips k + 1 .. 3 sit in X[E ..] (the block is shifted down after every panel)
        double tf4[4];
        double4_t sc;
        read_factor(k, tf4, sc);
#pragma unroll
        for (int s = 1; s < 4; ++s)
          if (s <= 3 - k) apply_strip(X + s * E, X, true, P, k, c0 + kNB * (k + s), tf4, sc, j == s);
      }
      if (!has_trailing) bump(&freeS[k]);
#pragma unroll
      for (int e = 0; e < 3 * E; ++e) X[e] = X[e + E];
      
    }
    // ================= the other blocks of pass P: blocks P + 3 .. go back to HBM, block P + 2 stays with me
#pragma unroll 1
    for (int blk = NBLK - 1; blk >= P + 2; --blk) {
      if (P > 0) {
        spin_ge(&storedS[j], P);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      load_block(blk, P == 0);
      
      apply_pass_to_block(blk, P, false, blk == P + 2);
      
      if (blk != P + 2) store_block(blk);
      
    }
    if (NBLK - 1 >= P + 3) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // my stores have left before the flag moves
      if (lane == 0) *(volatile lds_int*)&storedS[j] = P + 1;
    }
    // ================= the panels of pass P + 1 (the other group's chain) on block P + 2, my next panel block
    if (has_trailing) apply_pass_to_block(P + 2, P + 1, true, false);
    
  }
#ifdef PNODE_TIMING_BUILD
  if (lane == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(v.S + size_t(b) * d * d) + wv * 8;
    for (int i = 0; i < 8; ++i) dbg[i] = st.acc[i];
  }
#endif
  if (abortS) {
    // a wait ran into its limit: make the failure visible in every output of the step
    for (int i = lane; i < D * D; i += 64) {
      T[i] = __builtin_nan("");
      Ldst[i] = __builtin_nan("");
    }
  }
}

bool dense_lqc_supported(const DenseView& v) {
  const int DT = v.D / 16, DD = v.d / 16;
  return v.d % 16 == 0 && ((DT == 8 && DD == 2) || (DT == 12 && DD == 2) || (DT == 16 && DD == 4));
}

void dense_launch_lqc(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  switch (v.D / 16) {
    case 8: hipLaunchKernelGGL((dense_lqc_mfma_kernel<8, 2>), dim3(v.B), dim3(512), 0, s, v, args, K); break;
    case 12: hipLaunchKernelGGL((dense_lqc_mfma_kernel<12, 2>), dim3(v.B), dim3(512), 0, s, v, args, K); break;
    case 16: hipLaunchKernelGGL((dense_lqc_mfma_kernel<16, 4>), dim3(v.B), dim3(512), 0, s, v, args, K); break;
    default: break;
  }
}

}  // namespace pnode
