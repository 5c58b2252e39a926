// Dense EKF1 for LARGE state dimensions (D = (nu+1) d a multiple of 64 up to 1024, d % 32 == 0): BASELINE config 5
// (Brusselator N = 128, d = 256, nu = 3, D = 1024 — "MFMA Jacobian x cov path"). One trajectory's factor (8 MB at
// D = 1024) fits neither registers nor LDS, so the step is a sequence of batched, HBM/L2-tiled kernels:
//   * gemm_kernel        : C = beta C0 + alpha A op(B), fp64 MFMA 16x16x4, 64x64 or 128x32 tiles staged through LDS,
//                          register prefetch of the next k-tile; one launch covers all trajectories (blockIdx.z)
//   * lq_panel_kernel    : row-Householder factorisation of a 32-row panel of [T | Mt] (rows in registers, pivot row
//                          broadcast through LDS, DPP reductions) + compact-WY factor by the nilpotent product
//   * potrf_panel_kernel / trsm_block_kernel : blocked Cholesky and triangular solve of the d x d innovation
// The blocked LQ applies each panel's block reflector I - Y Tf Y^T to the trailing rows with four GEMMs:
//   G = T[:, panel] + Mt V^T ;  W = G Tf ;  T[:, panel] -= W ;  Mt -= W V.
// Same algorithm and the same numbers (to rounding) as pnode_dense.hip / pnode_dense_mfma.hip; reference anchors:
// strategy_filter + correction_ts1 + solver_dynamic (/root/reference/experiments/1_van_der_pol/vdp.py:62-66) on the
// Brusselator of /root/reference/src/odecheckpts/ivps.py:124-156; arithmetic restated in oracle/pn_oracle.py.
#include "pnode_device.hpp"

namespace pnode {
namespace lg {

typedef double double4_t __attribute__((ext_vector_type(4)));
#define LG_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

constexpr int PB = 32;  // panel height of the blocked LQ / Cholesky / triangular solve

// ---------------------------------------------------------------------------------------------
// batched GEMM (row-major): C[b] = beta * C0[b] + alpha * A[b] * op(B[b]);  op(B) = B (K x N) or B^T (B is N x K)
// M, N, K multiples of 16; all leading dimensions and offsets multiples of 4 doubles.
// ---------------------------------------------------------------------------------------------
struct Gemm {
  const double* A;
  const double* B;
  const double* C0;
  double* C;
  long sA, sB, sC0, sC;
  int lda, ldb, ldc0, ldc;
  int M, N, K;
  double alpha, beta;
  const int32_t* act;  // per-trajectory activity flags (nullptr: all)
  const long* offC;    // optional per-trajectory element offsets of C (replaces b * sC)
  const long* offA;    // same for A
  const long* offB;    // same for B
  int ksplit;          // > 1 (N <= 32 only): blockIdx.x selects a K-slice; slice kx writes its partial product to
  long sPart;          //   C + kx * sPart (C0 is added by slice 0 only); sum_partials_kernel adds the slices up
  double* E;           // optional second output of the SAME product, updated in place: E[b] = E[b] + ealpha * A op(B)  (no K-slices)
  long sE;             //   -- the block reflector's W = G Tf and T[:, blk] -= G Tf are one product, not two
  int lde;
  double ealpha;
};

// PD = register prefetch depth in k-tiles: the skinny shapes (N <= 32: a handful of workgroups per trajectory, K up to D)
// are bound by the global-load latency of their k-loop, so they keep four tiles in flight; the square shapes hide it
// with occupancy and keep one.
template <bool TB, int WMs, int WNs, int PD>
__global__ __launch_bounds__(256) void gemm_kernel(Gemm g) {
  static_assert(WMs * WNs == 4, "four waves per workgroup");
  constexpr int BM = 32 * WMs, BN = 32 * WNs, KS = 16, LDAS = KS + 1;
  constexpr int LDBS = TB ? KS + 1 : BN + 16;
  constexpr int APASS = BM / 64;
  __shared__ double As[BM * LDAS];
  __shared__ double Bs[TB ? BN * LDBS : KS * LDBS];
  const int b = blockIdx.z;
  if (g.act && !g.act[b]) return;
  const bool split = g.ksplit > 1;
  const int kx = split ? blockIdx.x : 0;
  const int kbeg = split ? kx * (g.K / g.ksplit) : 0, kend = split ? kbeg + g.K / g.ksplit : g.K;
  const int m0 = blockIdx.y * BM, n0 = split ? 0 : blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, gq = lane >> 4;
  const int wm = w / WNs, wn = w % WNs;
  const double* A = g.A + (g.offA ? g.offA[b] : long(b) * g.sA);
  const double* Bm = g.B + (g.offB ? g.offB[b] : long(b) * g.sB);
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  double4_t acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
  double4_t ra[PD][APASS], rb[PD];
  const int lrow = tid >> 2, lc4 = (tid & 3) * 4;               // A tile / transposed B tile: 64 rows x 16 per pass
  const int nkr = tid / (BN / 4), nc4 = (tid % (BN / 4)) * 4;   // B tile (not transposed): 16 rows x BN
  // Loads are unconditional (clamped addresses, values masked afterwards) so that the compiler can count the loads in
  // flight exactly (s_waitcnt vmcnt(n) instead of vmcnt(0)) and the prefetch depth is real.
  auto fetch = [&](int kk, double4_t (&qa)[APASS], double4_t& qb) {
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const int gr = m0 + p * 64 + lrow;
      const int grc = gr < g.M ? gr : g.M - 1;
      const double4_t t = *reinterpret_cast<const double4_t*>(A + long(grc) * g.lda + kk + lc4);
      qa[p] = (gr < g.M) ? t : zero4;
    }
    if (TB) {
      const int gn = n0 + lrow;
      const int gnc = gn < g.N ? gn : g.N - 1;
      const double4_t t = *reinterpret_cast<const double4_t*>(Bm + long(gnc) * g.ldb + kk + lc4);
      qb = (lrow < BN && gn < g.N) ? t : zero4;
    } else {
      const int kr = nkr < KS ? nkr : KS - 1;
      const int cc = n0 + nc4 < g.N ? n0 + nc4 : g.N - 4;
      const double4_t t = *reinterpret_cast<const double4_t*>(Bm + long(kk + kr) * g.ldb + cc);
      qb = (nkr < KS && n0 + nc4 < g.N) ? t : zero4;
    }
  };
  auto stash = [&](const double4_t (&qa)[APASS], const double4_t& qb) {
#pragma unroll
    for (int p = 0; p < APASS; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) As[(p * 64 + lrow) * LDAS + lc4 + q] = qa[p][q];
    if (TB) {
      if (lrow < BN)
#pragma unroll
        for (int q = 0; q < 4; ++q) Bs[lrow * LDBS + lc4 + q] = qb[q];
    } else if (nkr < KS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Bs[nkr * LDBS + nc4 + q] = qb[q];
    }
  };
  const int klast = kend - KS;
#pragma unroll
  for (int u = 0; u < PD; ++u) fetch(kbeg + u * KS < klast ? kbeg + u * KS : klast, ra[u], rb[u]);
  for (int kk0 = kbeg; kk0 < kend; kk0 += PD * KS) {
#pragma unroll
   for (int u = 0; u < PD; ++u) {
    const int kk = kk0 + u * KS;  // (the launcher guarantees that K and every K-slice are multiples of PD * KS)
    stash(ra[u], rb[u]);
    __syncthreads();
    fetch(kk + PD * KS < klast ? kk + PD * KS : klast, ra[u], rb[u]);  // (past the end: a harmless re-read of the last tile)
    __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMAs (they would be sunk next to the next stash)
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      double a[2], bb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[(wm * 32 + t * 16 + r) * LDAS + k4 * 4 + gq];
        bb[t] = TB ? Bs[(wn * 32 + t * 16 + r) * LDBS + k4 * 4 + gq] : Bs[(k4 * 4 + gq) * LDBS + wn * 32 + t * 16 + r];
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = LG_MFMA(a[tm], bb[tn], acc[tm][tn]);
    }
    __syncthreads();
   }
  }
  const double* C0 = (g.C0 && kx == 0) ? g.C0 + long(b) * g.sC0 : nullptr;
  double* C = g.C + (g.offC ? g.offC[b] : long(b) * g.sC) + long(kx) * g.sPart;
  // epilogue in two phases (all C0 loads in flight, then the stores): C0 may alias C element by element (in-place
  // updates), which would otherwise serialise every load behind the previous store
  double4_t c0v[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * 32 + tm * 16 + 4 * q + gq, col = n0 + wn * 32 + tn * 16 + r;
        c0v[tm][tn][q] = (C0 && row < g.M && col < g.N) ? C0[long(row) * g.ldc0 + col] : 0.0;
      }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * 32 + tm * 16 + 4 * q + gq, col = n0 + wn * 32 + tn * 16 + r;
        if (row < g.M && col < g.N) C[long(row) * g.ldc + col] = fma(g.beta, c0v[tm][tn][q], g.alpha * acc[tm][tn][q]);
      }
  if (g.E) {  // (the same expression a separate launch with C0 = C = E, beta = 1 would evaluate)
    double* E = g.E + long(b) * g.sE;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + wm * 32 + tm * 16 + 4 * q + gq, col = n0 + wn * 32 + tn * 16 + r;
          c0v[tm][tn][q] = (row < g.M && col < g.N) ? E[long(row) * g.lde + col] : 0.0;
        }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + wm * 32 + tm * 16 + 4 * q + gq, col = n0 + wn * 32 + tn * 16 + r;
          if (row < g.M && col < g.N) E[long(row) * g.lde + col] = fma(1.0, c0v[tm][tn][q], g.ealpha * acc[tm][tn][q]);
        }
  }
}

// The same product for the square shapes of the blocked LQ and of the measurement update (M, N >= 128, K >= 64): 128 x 128 tile per
// workgroup, 64 x 64 (4 x 4 MFMA tiles, 16 accumulators) per wavefront -- one LDS operand read per two MFMAs instead of one per
// MFMA, half the global traffic per flop of the 64 x 64 tile -- and the k-tiles double-buffered in LDS: ONE barrier per k-tile, 64
// MFMAs (4 096 matrix-pipe cycles) per wavefront between two barriers instead of 16. Tile it + 1 is written to the other LDS buffer
// and the global loads of tile it + 2 are issued in front of the MFMAs of tile it. Element by element the same sums in the same order as
// gemm_kernel (k ascending, four per MFMA): the two kernels give identical bits.
template <bool TB>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(Gemm g) {
  constexpr int BM = 128, BN = 128, KS = 16, LDAS = KS + 1;
  constexpr int LDBS = TB ? KS + 1 : BN + 16;
  constexpr int ASZ = BM * LDAS, BSZ = TB ? BN * LDBS : KS * LDBS;
  __shared__ double As[2 * ASZ];
  __shared__ double Bs[2 * BSZ];
  // Workgroup -> (trajectory, tile): consecutive workgroup ids go to consecutive XCDs (eight, each with its own L2), so with the plain
  // (x, y, z) order the eight column tiles that read the same 128 rows of A would sit in eight different L2s. All tiles of a trajectory
  // go to ONE XCD instead (id % 8 picks the XCD, trajectories are dealt round-robin to the XCDs): A and B of a trajectory are fetched
  // into one L2 once per pass over k.
  const int tx = (g.N + BN - 1) / BN, ty = (g.M + BM - 1) / BM, tiles = tx * ty;
  int b, tile;
  {
    const int id = blockIdx.x, nb = gridDim.x / tiles;
    if (nb % 8 == 0) {
      const int slot = id >> 3;
      b = (slot / tiles) * 8 + (id & 7);
      tile = slot % tiles;
    } else {
      b = id / tiles;
      tile = id % tiles;
    }
  }
  if (g.act && !g.act[b]) return;
  const int m0 = (tile / tx) * BM, n0 = (tile % tx) * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, gq = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  const double* A = g.A + (g.offA ? g.offA[b] : long(b) * g.sA);
  const double* Bm = g.B + (g.offB ? g.offB[b] : long(b) * g.sB);
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  double4_t acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = zero4;
  double4_t ra[2], rb[2];
  const int lrow = tid >> 2, lc4 = (tid & 3) * 4;  // A tile / transposed B tile: 64 rows x 16 per pass, two passes
  const int nkr = tid >> 5, nc4 = (tid & 31) * 4;  // B tile (not transposed): 8 rows x 128 per pass, two passes
  // Clamped addresses, unconditional loads; the values are masked when they are WRITTEN to LDS, a k-tile later -- a mask next to
  // the load would make the wavefront wait for it in front of the MFMAs it is meant to hide behind.
  bool oka[2], okb[2];
  const double* pa[2];
  const double* pb[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int gr = m0 + p * 64 + lrow;
    oka[p] = gr < g.M;
    pa[p] = A + long(oka[p] ? gr : g.M - 1) * g.lda + lc4;
    if (TB) {
      const int gn = n0 + p * 64 + lrow;
      okb[p] = gn < g.N;
      pb[p] = Bm + long(okb[p] ? gn : g.N - 1) * g.ldb + lc4;
    } else {
      okb[p] = n0 + nc4 < g.N;
      pb[p] = Bm + long(p * 8 + nkr) * g.ldb + (okb[p] ? n0 + nc4 : g.N - 4);
    }
  }
  auto fetch = [&](int kk) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ra[p] = *reinterpret_cast<const double4_t*>(pa[p] + kk);
      rb[p] = *reinterpret_cast<const double4_t*>(pb[p] + (TB ? long(kk) : long(kk) * g.ldb));
    }
  };
  auto stash = [&](int buf) {
    double* as = As + buf * ASZ;
    double* bs = Bs + buf * BSZ;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const double4_t va = oka[p] ? ra[p] : zero4, vb = okb[p] ? rb[p] : zero4;
#pragma unroll
      for (int q = 0; q < 4; ++q) as[(p * 64 + lrow) * LDAS + lc4 + q] = va[q];
      if (TB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bs[(p * 64 + lrow) * LDBS + lc4 + q] = vb[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) bs[(p * 8 + nkr) * LDBS + nc4 + q] = vb[q];
      }
    }
  };
  const int nk = g.K / KS;
  fetch(0);
  stash(0);
  __syncthreads();
  if (nk > 1) fetch(KS);
  for (int it = 0; it < nk; ++it) {
    // tile it + 1 (in registers since the previous iteration) goes to the other buffer -- every wavefront finished reading that
    // buffer before the barrier that ended the previous iteration -- and the loads of tile it + 2 start: both in front of the MFMAs
    if (it + 1 < nk) {
      stash((it + 1) & 1);
      if (it + 2 < nk) fetch((it + 2) * KS);
    }
    __builtin_amdgcn_sched_barrier(0);
    const double* as = As + (it & 1) * ASZ;
    const double* bs = Bs + (it & 1) * BSZ;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      double a[4], bb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = as[(wm * 64 + t * 16 + r) * LDAS + k4 * 4 + gq];
        bb[t] = TB ? bs[(wn * 64 + t * 16 + r) * LDBS + k4 * 4 + gq] : bs[(k4 * 4 + gq) * LDBS + wn * 64 + t * 16 + r];
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = LG_MFMA(a[tm], bb[tn], acc[tm][tn]);
    }
    __syncthreads();
  }
  const double* C0 = g.C0 ? g.C0 + long(b) * g.sC0 : nullptr;
  double* C = g.C + (g.offC ? g.offC[b] : long(b) * g.sC);
  // epilogue by tile rows, loads of a tile row before its stores (C0 may alias C element by element)
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    double4_t c0v[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * 64 + tm * 16 + 4 * q + gq, col = n0 + wn * 64 + tn * 16 + r;
        c0v[tn][q] = (C0 && row < g.M && col < g.N) ? C0[long(row) * g.ldc0 + col] : 0.0;
      }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * 64 + tm * 16 + 4 * q + gq, col = n0 + wn * 64 + tn * 16 + r;
        if (row < g.M && col < g.N) C[long(row) * g.ldc + col] = fma(g.beta, c0v[tn][q], g.alpha * acc[tm][tn][q]);
      }
  }
}

// P0 += P1 + ... (fixed order): the K-slices of a split skinny GEMM
__global__ __launch_bounds__(256) void sum_partials_kernel(double* __restrict__ C, long sC, long sPart, int count, int nparts,
                                                           const int32_t* __restrict__ act) {
  const int b = blockIdx.y;
  if (act && !act[b]) return;
  double* c = C + long(b) * sC;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < count; e += gridDim.x * 256) {
    double acc = c[e];
    for (int k = 1; k < nparts; ++k) acc += c[e + long(k) * sPart];
    c[e] = acc;
  }
}

// Inside an outer block, after G = T[:, panel] + Mt V^T of a panel (rows <= 96, 32 columns, possibly in K-slices): the slices summed, W = G Tf
// and T[:, panel] -= G Tf in ONE launch, one workgroup per 32 rows (it replaces sum_partials_kernel and two 32-wide gemm_kernel launches;
// the sums and products in the same order, element by element: identical bits).
__global__ __launch_bounds__(256) void panel_gw_kernel(const double* __restrict__ G, long sG, long sPart, int nparts, const double* __restrict__ Tf,
                                                       long sTf, double* __restrict__ W, double* __restrict__ T, long sT, int ldt, int rows,
                                                       const int32_t* __restrict__ act) {
  __shared__ double Gs[32 * 33];
  __shared__ double Ts[32 * 33];
  const int b = blockIdx.y;
  if (act && !act[b]) return;
  const int r0 = blockIdx.x * 32, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
  const double* g = G + long(b) * sG;
  const double* tf = Tf + long(b) * sTf;
  // four elements per thread, every load of every slice in flight at once (nparts <= 4)
  double part[4][4], tfv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u, row = e >> 5, col = e & 31;
    const long idx = long(r0 + row < rows ? r0 + row : rows - 1) * 32 + col;
#pragma unroll
    for (int k = 0; k < 4; ++k) part[u][k] = g[idx + long(k < nparts ? k : 0) * sPart];
    tfv[u] = tf[e];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u, row = e >> 5, col = e & 31;
    double acc = part[u][0];
#pragma unroll
    for (int k = 1; k < 4; ++k) acc = (k < nparts) ? acc + part[u][k] : acc;
    Gs[row * 33 + col] = (r0 + row < rows) ? acc : 0.0;
    Ts[row * 33 + col] = tfv[u];
  }
  __syncthreads();
  // wavefront wv: the 16 x 16 tile (tm, tn) = (wv >> 1, wv & 1) of the 32 x 32 product
  const int tm = wv >> 1, tn = wv & 1;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) acc = LG_MFMA(Gs[(tm * 16 + r) * 33 + k4 * 4 + gq], Ts[(k4 * 4 + gq) * 33 + tn * 16 + r], acc);
  double* w = W + long(b) * sG;
  double* t = T + long(b) * sT;
  double told[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = r0 + tm * 16 + 4 * q + gq, col = tn * 16 + r;
    told[q] = t[long(row < rows ? row : rows - 1) * ldt + col];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = r0 + tm * 16 + 4 * q + gq, col = tn * 16 + r;
    if (row < rows) {
      w[long(row) * 32 + col] = fma(0.0, 0.0, 1.0 * acc[q]);
      t[long(row) * ldt + col] = fma(1.0, told[q], -1.0 * acc[q]);
    }
  }
}

// (v.gemm_tile64: pnode_set_generic_path(h, 4) pins the 64 x 64 tile, the A/B partner of gemm128_kernel)
static void gemm(const DenseView& v, hipStream_t s, int batch, bool transB, const Gemm& g) {
  const bool gemm_tile128 = !v.gemm_tile64;
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return;
#ifdef PNODE_TIMING_BUILD
  fprintf(stderr, "gemm M %d N %d K %d batch %d transB %d ksplit %d beta %g\n", g.M, g.N, g.K, batch, int(transB), g.ksplit, g.beta);
#endif
  if (g.N <= 32) {
    dim3 grid(g.ksplit > 1 ? g.ksplit : (g.N + 31) / 32, (g.M + 127) / 128, batch);
    if (transB && g.K % 64 == 0)
      hipLaunchKernelGGL((gemm_kernel<true, 4, 1, 4>), grid, dim3(256), 0, s, g);
    else if (transB)
      hipLaunchKernelGGL((gemm_kernel<true, 4, 1, 1>), grid, dim3(256), 0, s, g);
    else
      hipLaunchKernelGGL((gemm_kernel<false, 4, 1, 1>), grid, dim3(256), 0, s, g);
  } else if (gemm_tile128 && !g.E && g.ksplit <= 1 && g.M >= 128 && g.N >= 128 && g.K >= 512 &&
             long((g.M + 127) / 128) * ((g.N + 127) / 128) * batch >= 256) {
    // Long inner dimension only (measured at D = 1024, B = 64, profiles/r04_gemm_tiles.txt): 1024 x 256 x 1024 goes from 55 to 62 TFLOP/s,
    // while the rank-128 / rank-256 updates of the blocked LQ (K <= 256, bound by the traffic of C: 16 flops per byte) lose 12 % on the
    // large tile, whose two workgroups per CU overlap the epilogue of one tile with the MFMAs of the next less well than eight do.
    // (The choice depends on the shape and on the batch size, the bits do not: see gemm128_kernel.)
    dim3 grid(((g.N + 127) / 128) * ((g.M + 127) / 128) * batch);
    if (transB)
      hipLaunchKernelGGL((gemm128_kernel<true>), grid, dim3(256), 0, s, g);
    else
      hipLaunchKernelGGL((gemm128_kernel<false>), grid, dim3(256), 0, s, g);
  } else {
    dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, batch);
    if (transB)
      hipLaunchKernelGGL((gemm_kernel<true, 2, 2, 1>), grid, dim3(256), 0, s, g);
    else
      hipLaunchKernelGGL((gemm_kernel<false, 2, 2, 1>), grid, dim3(256), 0, s, g);
  }
}

// ---------------------------------------------------------------------------------------------
// LQ panel: rows p0 .. p0+31 of [T | Mt]. 512 threads: 16-lane group rr owns row rr in registers.
// On exit: Mt rows hold the SCALED reflector rows V (reflector c = [e_c | V_c]), T[p0.., p0..] the final lower block,
// Tf the 32 x 32 compact-WY factor (row-major) such that the block reflector is I - Y Tf Y^T.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// Sum over the wavefront, result uniform: DPP within the 16-lane rows, v_readlane across them.
__device__ __forceinline__ double wave_total(double v) {
  v = row16_sum(v);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double s = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  s += __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  s += __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  s += __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return s;
}

// 1024 threads: wavefront w owns rows 2w, 2w+1 (lane <-> columns lane + 64 i), so the pivot row is read from LDS once
// per step and serves the norm, both dot products and both updates out of registers.
template <int DT>
__global__ __launch_bounds__(1024) void lq_panel_kernel(DenseView v, int p0, double* __restrict__ Tf_out, int tf_stride) {
  constexpr int D = 16 * DT, NC = DT / 4;
  __shared__ double piv[2][D];
  __shared__ double Tpp[PB * PB], Gm[PB * PB], Nm[PB * PB], Af[PB * PB];
  __shared__ double s_tau[PB], s_scl[PB], s_beta[PB];
  const int b = blockIdx.x;
  if (!v.act[b]) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r0 = 2 * w, r1 = 2 * w + 1;
  double* Mt = v.M + size_t(b) * v.mr_rows * D * D + size_t(p0) * D;
  double* T = v.R + size_t(b) * v.mr_rows * D * D + size_t(p0) * D + p0;
  double x0[NC], x1[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    x0[i] = Mt[size_t(r0) * D + lane + 64 * i];
    x1[i] = Mt[size_t(r1) * D + lane + 64 * i];
  }
  {
    const int i = tid >> 5, j = tid & 31;
    Tpp[tid] = (j <= i) ? T[size_t(i) * D + j] : 0.0;
    Gm[tid] = 0.0;
  }
  // The owner of the next pivot row computes its Householder scalars and publishes them with the row, so the other
  // fifteen wavefronts spend their issue slots on the two dot products and the two row updates only.
  auto publish = [&](int c, const double (&x)[NC]) {
    double* pn = piv[c & 1];
    double pp = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      pn[lane + 64 * i] = x[i];
      pp = fma(x[i], x[i], pp);
    }
    const double xx = wave_total(pp);
    const double alpha = Tpp[c * PB + c];
    double tauc = 0.0, scale = 0.0, beta = alpha;
    if (xx != 0.0) {
      const double nrm2 = fma(alpha, alpha, xx);
      const double rs = fast_rsqrt(nrm2);
      const double sgn = copysign(1.0, alpha);
      beta = -sgn * nrm2 * rs;
      tauc = fma(fabs(alpha), rs, 1.0);
      scale = sgn * rs * fast_rcp(tauc);
    }
    if (lane == 0) {
      s_beta[c] = beta;
      s_tau[c] = tauc;
      s_scl[c] = scale;
    }
  };
  __syncthreads();
  if (w == 0) publish(0, x0);
  __syncthreads();
  for (int c = 0; c < PB; ++c) {
    const double* pc = piv[c & 1];
    double xc[NC];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      xc[i] = pc[lane + 64 * i];
      d0 = fma(x0[i], xc[i], d0);
      d1 = fma(x1[i], xc[i], d1);
    }
    const double dot0 = wave_total(d0);
    const double dot1 = wave_total(d1);
    const double tauc = s_tau[c], scale = s_scl[c];
    if (r0 > c) {
      if (tauc != 0.0) {
        const double tjc = Tpp[r0 * PB + c];
        const double wv = tauc * fma(scale, dot0, tjc);
        const double ws = wv * scale;
#pragma unroll
        for (int i = 0; i < NC; ++i) x0[i] = fma(-ws, xc[i], x0[i]);
        if (lane == 0) Tpp[r0 * PB + c] = tjc - wv;
      }
    } else if (r0 < c) {
      if (lane == 0) Gm[r0 * PB + c] = dot0;  // raw rows[r0] . rows[c], both final
    }
    if (r1 > c) {
      if (tauc != 0.0) {
        const double tjc = Tpp[r1 * PB + c];
        const double wv = tauc * fma(scale, dot1, tjc);
        const double ws = wv * scale;
#pragma unroll
        for (int i = 0; i < NC; ++i) x1[i] = fma(-ws, xc[i], x1[i]);
        if (lane == 0) Tpp[r1 * PB + c] = tjc - wv;
      }
    } else if (r1 < c) {
      if (lane == 0) Gm[r1 * PB + c] = dot1;
    }
    if (c + 1 < PB && ((c + 1) >> 1) == w) {  // the next pivot row has received its last update
      if ((c + 1) & 1) publish(c + 1, x1);
      else publish(c + 1, x0);
    }
    __syncthreads();
  }
  // scaled reflector rows back to HBM; final diagonal block of T
  {
    const double sc0 = s_scl[r0], sc1 = s_scl[r1];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      Mt[size_t(r0) * D + lane + 64 * i] = sc0 * x0[i];
      Mt[size_t(r1) * D + lane + 64 * i] = sc1 * x1[i];
    }
  }
  const int ei = tid >> 5, ej = tid & 31;
  if (ej < ei) T[size_t(ei) * D + ej] = Tpp[tid];
  else if (ej == ei) T[size_t(ei) * D + ej] = s_beta[ei];
  // Tf = (diag(tau)^-1 + striu(Y^T Y))^-1 = (I + N)^-1 diag(tau), N = diag(tau) striu(Y^T Y), nilpotent:
  // (I + N)^-1 = (I - N)(I + N^2)(I + N^4)(I + N^8)(I + N^16); one entry per thread
  {
    const double nij = (ej > ei) ? s_tau[ei] * s_scl[ei] * s_scl[ej] * Gm[tid] : 0.0;
    Nm[tid] = nij;
    Af[tid] = ((ei == ej) ? 1.0 : 0.0) - nij;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      double q = 0.0;
#pragma unroll 8
      for (int m = 0; m < PB; ++m) q = fma(Nm[ei * PB + m], Nm[m * PB + ej], q);
      __syncthreads();
      Nm[tid] = q;
      __syncthreads();
      double a = Af[tid];
#pragma unroll 8
      for (int m = 0; m < PB; ++m) a = fma(Af[ei * PB + m], Nm[m * PB + ej], a);
      __syncthreads();
      Af[tid] = a;
      __syncthreads();
    }
    Tf_out[size_t(b) * tf_stride + tid] = Af[tid] * s_tau[ej];
  }
}

// Compact-WY factor of an outer block of nb 32-row panels from the panel factors (diagonal blocks) and the Gram matrix
// of the scaled reflector rows: Tf^-1 = diag-blocks(Tf_p^-1) + striu-blocks(V V^T), inverted by block back-substitution
//      X_ii = Tf_i ,   X_ij = -Tf_i sum_{l=i+1..j} Gram_il X_lj   (j > i).
// r04: everything the back-substitution touches stays in LDS (the upper blocks of X and of the Gram matrix: 17 blocks of 32 x 33 doubles,
// 144 KB of dynamic LDS) -- one pass over global memory in, one out, instead of a global round trip per 32 x 32 product -- and the
// products run on the matrix cores (one 16 x 16 output tile per wavefront; the scalar version read two LDS operands per multiply-add).
constexpr int kTfBlock = PB * (PB + 1);
constexpr size_t kTfOuterShm = 17 * size_t(kTfBlock) * sizeof(double);
__global__ __launch_bounds__(256) void tf_outer_kernel(const double* __restrict__ Tfp, const double* __restrict__ Gram,
                                                       double* __restrict__ Tfo, int nb, const int32_t* __restrict__ act) {
  constexpr int LS = PB + 1;
  extern __shared__ double tf_sm[];
  double* Xb = tf_sm;                  // block (i, j), i <= j, at index j (j + 1) / 2 + i
  double* Gb = tf_sm + 10 * kTfBlock;  // block (i, l), i < l, at index l (l - 1) / 2 + i
  double* tmp = tf_sm + 16 * kTfBlock;
  const int b = blockIdx.x;
  if (act && !act[b]) return;
  const int OBs = nb * PB, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
  const int tm = wv >> 1, tn = wv & 1;  // this wavefront's 16 x 16 tile of every 32 x 32 product
  const double* tp = Tfp + size_t(b) * 4 * PB * PB;  // fixed per-trajectory strides (at most four panels per outer block)
  const double* G = Gram + size_t(b) * 128 * 128;
  double* X = Tfo + size_t(b) * 128 * 128;
  auto xb = [&](int i, int j) { return Xb + (j * (j + 1) / 2 + i) * kTfBlock; };
  auto gb = [&](int i, int l) { return Gb + (l * (l - 1) / 2 + i) * kTfBlock; };
  for (int i = 0; i < nb; ++i)
    for (int j = i; j < nb; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, rr = e >> 5, cc = e & 31;
        if (j == i) xb(i, i)[rr * LS + cc] = tp[i * PB * PB + e];
        else gb(i, j)[rr * LS + cc] = G[size_t(i * PB + rr) * OBs + j * PB + cc];
      }
  __syncthreads();
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  for (int dist = 1; dist < nb; ++dist)
    for (int i = 0; i + dist < nb; ++i) {
      const int j = i + dist;
      double4_t acc = zero4;
      for (int l = i + 1; l <= j; ++l) {
        const double* A = gb(i, l);
        const double* Bm = xb(l, j);
#pragma unroll
        for (int k4 = 0; k4 < PB / 4; ++k4) acc = LG_MFMA(A[(tm * 16 + r) * LS + k4 * 4 + gq], Bm[(k4 * 4 + gq) * LS + tn * 16 + r], acc);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) tmp[(tm * 16 + 4 * q + gq) * LS + tn * 16 + r] = acc[q];
      __syncthreads();
      const double* Ti = xb(i, i);
      double4_t o = zero4;
#pragma unroll
      for (int k4 = 0; k4 < PB / 4; ++k4) o = LG_MFMA(Ti[(tm * 16 + r) * LS + k4 * 4 + gq], tmp[(k4 * 4 + gq) * LS + tn * 16 + r], o);
      double* Xo = xb(i, j);
#pragma unroll
      for (int q = 0; q < 4; ++q) Xo[(tm * 16 + 4 * q + gq) * LS + tn * 16 + r] = -o[q];
      __syncthreads();
    }
  for (int e = tid; e < OBs * OBs; e += 256) {
    const int rr = e / OBs, cc = e % OBs, bi = rr / PB, bj = cc / PB;
    X[e] = (bi <= bj) ? xb(bi, bj)[(rr % PB) * LS + cc % PB] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// Blocked Cholesky of the d x d matrices S (lower, in place): one panel step = this kernel + one GEMM.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void potrf_panel_kernel(double* __restrict__ S, int d, int j0, const int32_t* __restrict__ act) {
  __shared__ double Ld[PB * (PB + 1)];
  const int b = blockIdx.x;
  if (act && !act[b]) return;
  double* A = S + size_t(b) * d * d;
  const int tid = threadIdx.x;
  // The 32 x 32 diagonal block by ONE wavefront, lane i <-> row i in registers: the pivot and the column under it travel by v_readlane, no
  // barrier inside the 32 pivots (r04; before: all four wavefronts on the block in LDS, three barriers per pivot -- 16 of the kernel's 27 us).
  if (tid < 64) {
    const int i = tid & 31;  // (lanes 32 .. 63 repeat lanes 0 .. 31: same values, same addresses)
    double x[PB];
    const double* ai = A + size_t(j0 + i) * d + j0;
#pragma unroll
    for (int j = 0; j < PB; ++j) x[j] = (j <= i) ? ai[j <= i ? j : 0] : 0.0;
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      // column c of the rows below: x[c] of lane c is the pivot
      const double dc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[c]), c), __builtin_amdgcn_readlane(__double2loint(x[c]), c));
      const double piv = sqrt(dc);
      x[c] = (i == c) ? piv : ((i > c) ? x[c] / piv : x[c]);
#pragma unroll
      for (int j = c + 1; j < PB; ++j) {
        // L[j][c], held by lane j
        const double ljc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[c]), j), __builtin_amdgcn_readlane(__double2loint(x[c]), j));
        x[j] = (i >= j) ? fma(-x[c], ljc, x[j]) : x[j];
      }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) Ld[i * (PB + 1) + j] = (j <= i) ? x[j] : 0.0;
  }
  __syncthreads();
  for (int e = tid; e < PB * PB; e += 256) {
    const int i = e >> 5, j = e & 31;
    A[size_t(j0 + i) * d + j0 + j] = Ld[i * (PB + 1) + j];  // upper part of the block zeroed
  }
  // rows below the block: x <- x Ld^-T
  for (int row = j0 + PB + tid; row < d; row += 256) {
    double x[PB];
    double* ar = A + size_t(row) * d + j0;
#pragma unroll
    for (int c = 0; c < PB; ++c) x[c] = ar[c];
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      double acc = x[c];
#pragma unroll
      for (int q = 0; q < c; ++q) acc = fma(-x[q], Ld[c * (PB + 1) + q], acc);
      x[c] = acc / Ld[c * (PB + 1) + c];
      __builtin_amdgcn_sched_barrier(0);  // keep the LDS operand loads of later columns from being hoisted (spills)
    }
#pragma unroll
    for (int c = 0; c < PB; ++c) ar[c] = x[c];
  }
}

// W_ib = C_ii^-1 F_ib for the 32-row block ib of the d x ncols right-hand side F (leading dimension ldf); thread per column
__global__ __launch_bounds__(256) void trsm_block_kernel(const double* __restrict__ S, double* __restrict__ F, long sF, int ldf,
                                                         int ncols, int d, int ib, const int32_t* __restrict__ act) {
  __shared__ double Ld[PB * (PB + 1)];
  const int b = blockIdx.y;
  if (act && !act[b]) return;
  const double* C = S + size_t(b) * d * d;
  const int tid = threadIdx.x, r0 = ib * PB;
  for (int e = tid; e < PB * PB; e += 256) {
    const int i = e >> 5, j = e & 31;
    Ld[i * (PB + 1) + j] = (j <= i) ? C[size_t(r0 + i) * d + r0 + j] : 0.0;
  }
  __syncthreads();
  const int col = blockIdx.x * 256 + tid;
  if (col >= ncols) return;
  double* f = F + long(b) * sF + long(r0) * ldf + col;
  double x[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) x[i] = f[long(i) * ldf];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    double acc = x[i];
#pragma unroll
    for (int q = 0; q < i; ++q) acc = fma(-Ld[i * (PB + 1) + q], x[q], acc);
    x[i] = acc / Ld[i * (PB + 1) + i];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) f[long(i) * ldf] = x[i];
}

__device__ __forceinline__ void precond(int n, double dt, const IwpConst& K, double* p, double* pinv) {
  const double adt = fabs(dt);
  double pw = sqrt(adt);
  for (int i = n - 1; i >= 0; --i) {
    p[i] = pw / K.fact[i];
    pinv[i] = K.fact[i] / pw;
    pw *= adt;
  }
}

__device__ __forceinline__ double block_sum256(double val, double* red) {
  val = wave_sum(val);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// x <- C^-1 x (lower C, d x d, global; d % 32 == 0), whole workgroup (256 threads). Blocked by 32: wavefront 0 solves
// the diagonal block in registers (lane = row, pivots broadcast with v_readlane: no barriers), then every thread
// subtracts the block's contribution from one of the rows below. Two barriers per 32 rows instead of two per row.
__device__ void forward_vec(const double* C, int d, double* x) {
  __shared__ double xs[PB];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int a0 = 0; a0 < d; a0 += PB) {
    __syncthreads();
    if (tid < 64) {
      double xr = (lane < PB) ? x[a0 + lane] : 0.0;
      for (int a = 0; a < PB; ++a) {
        const double piv = C[size_t(a0 + a) * d + a0 + a];
        const int lo = __builtin_amdgcn_readlane(__double2loint(xr), a), hi = __builtin_amdgcn_readlane(__double2hiint(xr), a);
        const double xa = __hiloint2double(hi, lo) / piv;
        if (lane == a) xr = xa;
        else if (lane > a && lane < PB) xr = fma(-C[size_t(a0 + lane) * d + a0 + a], xa, xr);
      }
      if (lane < PB) {
        x[a0 + lane] = xr;
        xs[lane] = xr;
      }
    }
    __syncthreads();
    for (int r = a0 + PB + tid; r < d; r += 256) {
      const double* cr = C + size_t(r) * d + a0;
      double acc = x[r];
#pragma unroll 8
      for (int a = 0; a < PB; ++a) acc = fma(-cr[a], xs[a], acc);
      x[r] = acc;
    }
  }
  __syncthreads();
}

// begin, part 1: activity flags, m^-, f, J, z, and F1 = [B_0 .. B_k] (d x (k+1)d, leading dimension D, into v.X) with
// B_j = lq[k][j] I - sum_{i<k} lq[i][j] J_i, so that S1 = F1 F1^T; diag(S1) -> v.err
__global__ __launch_bounds__(256) void begin_a_kernel(DenseView v, const StepArg* __restrict__ args, const IwpConst* __restrict__ Kc) {
  // grid (G, B): every workgroup of a trajectory extrapolates the (small) mean redundantly; the rows of J, F1 and
  // diag(S1) are split over the G workgroups; workgroup 0 owns the per-trajectory outputs (flags, m^-, z)
  extern __shared__ double smem[];
  const int b = blockIdx.y;
  const bool first = blockIdx.x == 0;
  const StepArg arg = args[b];
  if (first && threadIdx.x == 0) {
    v.act[b] = arg.active ? 1 : 0;
    v.lgOff[b] = (long(arg.dst) * v.B + b) * long(v.D) * v.D;
  }
  if (!arg.active) return;
  const IwpConst& K = *Kc;
  const int d = v.d, k = v.k, n = v.n, kd = k * d, D = v.D;
  double* ybuf = smem;      // [k+1][d]  (row k: the derivative the residual is taken against)
  double* lq = ybuf + kd + d;  // [n][n]
  double p[kMaxN], pinv[kMaxN];
  precond(n, arg.dt, K, p, pinv);
  for (int idx = threadIdx.x; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    lq[idx] = (j <= i) ? p[i] * K.qsT[j][i] : 0.0;
  }
  const size_t sB = size_t(v.B);
  const double* msrc = v.mean + (size_t(arg.src) * sB + b) * n * d;
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * n * d;
  for (int j = threadIdx.x; j < d; j += 256)
    for (int i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * pinv[kk] * msrc[kk * d + j];
      const double e = p[i] * acc;
      if (first) {
        mdst[i * d + j] = e;
        if (v.fixedpoint) v.mext[size_t(b) * n * d + i * d + j] = e;
      }
      if (i <= k) ybuf[i * d + j] = e;
    }
  __syncthreads();
  VfCtx ctx{v.vf_id, d, k, v.prm ? v.prm + (v.prm_per_traj ? size_t(b) * v.P : 0) : nullptr,
            v.user_f ? v.user_f + size_t(b) * d : nullptr, v.user_J ? v.user_J + size_t(b) * d * k * d : nullptr};
  const double tnew = arg.t + arg.dt;
  if (first)
    for (int j = threadIdx.x; j < d; j += 256) v.z[size_t(b) * d + j] = ybuf[k * d + j] - vf_f(ctx, ybuf, j, tnew);
  const int G = gridDim.x, rows_per = (d + G - 1) / G;
  const int a0 = blockIdx.x * rows_per, a1 = (a0 + rows_per < d) ? a0 + rows_per : d;
  double* Jg = v.J + size_t(b) * d * kd;
  for (int idx = a0 * kd + threadIdx.x; idx < a1 * kd; idx += 256) {
    const int a = idx / kd, rest = idx % kd;
    Jg[idx] = vf_jac(ctx, ybuf, a, rest / d, rest % d, tnew);
  }
  __syncthreads();
  double* F1 = v.X + size_t(b) * d * D;
  const int w1 = (k + 1) * d;
  for (int idx = a0 * w1 + threadIdx.x; idx < a1 * w1; idx += 256) {
    const int a = idx / w1, rest = idx % w1;
    const int j = rest / d, c = rest % d;
    double val = (a == c) ? lq[k * n + j] : 0.0;
    for (int i = 0; i < k; ++i) val -= lq[i * n + j] * Jg[a * kd + i * d + c];
    F1[size_t(a) * D + rest] = val;
  }
  __syncthreads();
  {  // diag(S1) = squared row norms of F1: one wavefront per row (coalesced)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int a = a0 + wv; a < a1; a += 4) {
      double acc = 0.0;
      for (int c = lane; c < w1; c += kWave) acc = fma(F1[size_t(a) * D + c], F1[size_t(a) * D + c], acc);
      acc = wave_sum(acc);
      if (lane == 0) v.err[size_t(b) * d + a] = acc;
    }
  }
}

// begin, part 2 (after S1 = C C^T): sigma_hat^2 = |C^-1 z|^2 / d, err = dt sigma_hat sqrt(diag S1)
__global__ __launch_bounds__(256) void begin_c_kernel(DenseView v, const StepArg* __restrict__ args, double* __restrict__ wbuf) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const int d = v.d;
  double* x = wbuf + size_t(b) * d;
  for (int j = threadIdx.x; j < d; j += 256) x[j] = v.z[size_t(b) * d + j];
  forward_vec(v.S + size_t(b) * d * d, d, x);
  double part = 0.0;
  for (int j = threadIdx.x; j < d; j += 256) part += x[j] * x[j];
  const double ww = block_sum256(part, red);
  const double sigma_hat = sqrt(ww / double(d));
  double* errg = v.err + size_t(b) * d;
  for (int j = threadIdx.x; j < d; j += 256) errg[j] = arg.dt * sigma_hat * sqrt(errg[j]);
  if (threadIdx.x == 0) v.sig_new[b] = v.dynamic ? sigma_hat : v.sigma[size_t(arg.src) * v.B + b];
}

// L^- = diag(p) T (lower; zeros above the diagonal) into v.M (Mt is dead after the LQ): regular batch stride for the GEMMs
__global__ __launch_bounds__(256) void scale_t_kernel(DenseView v, const StepArg* __restrict__ args, const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.y;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const int D = v.D, d = v.d;
  double p[kMaxN], pinv[kMaxN];
  precond(v.n, arg.dt, *Kc, p, pinv);
  const double* T = v.R + size_t(b) * v.mr_rows * D * D;
  double* L = v.M + size_t(b) * v.mr_rows * D * D;
  for (int r = blockIdx.x; r < D; r += gridDim.x) {
    const double pr = p[r / d];
    for (int c = threadIdx.x; c < D; c += 256) L[size_t(r) * D + c] = (c <= r) ? pr * T[size_t(r) * D + c] : 0.0;
  }
}

// wz = C^-1 z (in place in v.z)
__global__ __launch_bounds__(256) void solve_wz_kernel(DenseView v, const StepArg* __restrict__ args) {
  const int b = blockIdx.x;
  if (!args[b].active) return;
  forward_vec(v.S + size_t(b) * v.d * v.d, v.d, v.z + size_t(b) * v.d);
}

// m = m^- - Y wz
__global__ __launch_bounds__(256) void finish_mean_kernel(DenseView v, const StepArg* __restrict__ args) {
  const int b = blockIdx.y;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const int d = v.d, D = v.D;
  double* mdst = v.mean + (size_t(arg.dst) * v.B + b) * D;
  const double* Y = v.Ft + size_t(b) * D * d;
  const double* wz = v.z + size_t(b) * d;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wv; r < D; r += 4 * gridDim.x) {
    double acc = 0.0;
    for (int a = lane; a < d; a += kWave) acc = fma(Y[size_t(r) * d + a], wz[a], acc);
    acc = wave_sum(acc);
    if (lane == 0) mdst[r] -= acc;
  }
}

// normalised error norm ; output scale of the new state
__global__ __launch_bounds__(256) void finish_kernel(DenseView v, const StepArg* __restrict__ args) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const int d = v.d, D = v.D;
  const size_t sB = size_t(v.B);
  const double* mdst = v.mean + (size_t(arg.dst) * sB + b) * D;
  const double atol = v.atol[b], rtol = v.rtol[b];
  double part = 0.0;
  for (int j = threadIdx.x; j < d; j += 256) {
    const double rel = v.err[size_t(b) * d + j] / (atol + rtol * fabs(mdst[j]));
    part += rel * rel;
  }
  const double tot = block_sum256(part, red);
  if (threadIdx.x == 0) {
    v.errnorm[b] = sqrt(tot / double(d));
    v.sigma[size_t(arg.dst) * sB + b] = v.sig_new[b];
  }
}

static void potrf(hipStream_t s, const DenseView& v) {
  const int d = v.d;
  for (int j0 = 0; j0 < d; j0 += PB) {
    hipLaunchKernelGGL(potrf_panel_kernel, dim3(v.B), dim3(256), 0, s, v.S, d, j0, v.act);
    const int rem = d - j0 - PB;
    if (rem > 0) {
      double* sub = v.S + size_t(j0 + PB) * d + j0;  // rows below, panel columns
      Gemm g{sub, sub, v.S + size_t(j0 + PB) * d + j0 + PB, v.S + size_t(j0 + PB) * d + j0 + PB,
             long(d) * d, long(d) * d, long(d) * d, long(d) * d, d, d, d, d, rem, rem, PB, -1.0, 1.0, v.act, nullptr};
      gemm(v, s, v.B, true, g);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Fixed-point smoother on the large-D path: backward transition of the predict just factorised + merge with the
// running one (same algebra as dfp_backward_kernel in pnode_dense_fp.hip, as batched GEMM / blocked TRSM).
// ---------------------------------------------------------------------------------------------
__global__ void fp_setup_kernel(DenseView v, const FpArg* __restrict__ fa) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= v.B) return;
  const FpArg a = fa[b];
  const long DD = long(v.D) * v.D;
  v.act[b] = a.active ? 1 : 0;
  v.lgMaskMerge[b] = (a.active && a.merge) ? 1 : 0;
  v.lgOffS[b] = (long(a.src) * v.B + b) * DD;
  v.lgOffD[b] = (long(a.dst) * v.B + b) * DD;
}

// X[:, jb] = C21[:, jb] Lm[jb, jb]^-1 for the 32-column block jb (thread per row; zero pivot -> pseudo-inverse)
__global__ __launch_bounds__(256) void trsm_right_block_kernel(DenseView v, int jb) {
  __shared__ double Ld[PB * (PB + 1)];
  const int b = blockIdx.y;
  if (!v.act[b]) return;
  const int D = v.D, tid = threadIdx.x, c0 = jb * PB;
  double* R = v.R + size_t(b) * 2 * D * D;
  const double* Lm = R;
  double* C = R + size_t(D) * D;
  for (int e = tid; e < PB * PB; e += 256) {
    const int i = e >> 5, j = e & 31;
    Ld[i * (PB + 1) + j] = (j <= i) ? Lm[size_t(c0 + i) * D + c0 + j] : 0.0;
  }
  __syncthreads();
  const int row = blockIdx.x * 256 + tid;
  if (row >= D) return;
  double* cr = C + size_t(row) * D + c0;
  double x[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) x[j] = cr[j];
#pragma unroll
  for (int j = PB - 1; j >= 0; --j) {
    double acc = x[j];
#pragma unroll
    for (int k = j + 1; k < PB; ++k) acc = fma(-x[k], Ld[k * (PB + 1) + j], acc);
    const double piv = Ld[j * (PB + 1) + j];
    x[j] = (piv != 0.0) ? acc / piv : 0.0;
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) cr[j] = x[j];
}

// back to plain coordinates: G *= p_r / p_c, Lam *= p_r
__global__ __launch_bounds__(256) void fp_scale_kernel(DenseView v, const FpArg* __restrict__ fa, const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.y;
  const FpArg a = fa[b];
  if (!a.active) return;
  const int D = v.D, d = v.d;
  double p[kMaxN], pinv[kMaxN];
  precond(v.n, a.dt, *Kc, p, pinv);
  double* G = v.R + size_t(b) * 2 * D * D + size_t(D) * D;
  double* Lam = v.M + size_t(b) * 2 * D * D + size_t(D) * D;
  for (int r = blockIdx.x; r < D; r += gridDim.x) {
    const double pr = p[r / d];
    for (int c = threadIdx.x; c < D; c += 256) {
      G[size_t(r) * D + c] *= pr * pinv[c / d];
      Lam[size_t(r) * D + c] *= pr;
    }
  }
}

// xi = m_prev - G m^-   (into v.lgXi)
__global__ __launch_bounds__(256) void fp_xi_kernel(DenseView v, const FpArg* __restrict__ fa) {
  const int b = blockIdx.y;
  const FpArg a = fa[b];
  if (!a.active) return;
  const int D = v.D;
  const double* G = v.R + size_t(b) * 2 * D * D + size_t(D) * D;
  const double* mprev = v.mean + (size_t(a.src) * v.B + b) * D;
  const double* mext = v.mext + size_t(b) * D;
  double* xi = v.lgXi + size_t(b) * D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wv; r < D; r += 4 * gridDim.x) {
    double acc = 0.0;
    for (int c = lane; c < D; c += kWave) acc = fma(G[size_t(r) * D + c], mext[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) xi[r] = mprev[r] - acc;
  }
}

// merge: T := Lam0 ; xi_dst = xi0 + G0 xi.   no merge: G_dst = G, Mt := Lam, T := 0, xi_dst = xi.
__global__ __launch_bounds__(256) void fp_merge_misc_kernel(DenseView v, const FpArg* __restrict__ fa) {
  const int b = blockIdx.y;
  const FpArg a = fa[b];
  if (!a.active) return;
  const int D = v.D;
  const size_t sB = size_t(v.B), DD = size_t(D) * D;
  double* M = v.M + size_t(b) * 2 * DD;
  double* R = v.R + size_t(b) * 2 * DD;
  const double* G = R + DD;
  const double* Lam = M + DD;
  const double* xi = v.lgXi + size_t(b) * D;
  double* xi_dst = v.cxi + (size_t(a.dst) * sB + b) * D;
  if (a.merge) {
    const double* Lam0 = v.cLam + (size_t(a.src) * sB + b) * DD;
    const double* G0 = v.cG + (size_t(a.src) * sB + b) * DD;
    const double* xi0 = v.cxi + (size_t(a.src) * sB + b) * D;
    for (int r = blockIdx.x; r < D; r += gridDim.x)
      for (int c = threadIdx.x; c < D; c += 256) R[size_t(r) * D + c] = Lam0[size_t(r) * D + c];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wv; r < D; r += 4 * gridDim.x) {
      double acc = 0.0;
      for (int k = lane; k < D; k += kWave) acc = fma(G0[size_t(r) * D + k], xi[k], acc);
      acc = wave_sum(acc);
      if (lane == 0) xi_dst[r] = xi0[r] + acc;
    }
  } else {
    double* G_dst = v.cG + (size_t(a.dst) * sB + b) * DD;
    for (int r = blockIdx.x; r < D; r += gridDim.x)
      for (int c = threadIdx.x; c < D; c += 256) {
        const size_t idx = size_t(r) * D + c;
        G_dst[idx] = G[idx];
        M[idx] = Lam[idx];
        R[idx] = 0.0;
      }
    if (blockIdx.x == 0)
      for (int r = threadIdx.x; r < D; r += 256) xi_dst[r] = xi[r];
  }
}

template <int DT>
static void launch_panel(const DenseView& v, int p0, double* Tf, int tf_stride, hipStream_t s) {
  hipLaunchKernelGGL(lq_panel_kernel<DT>, dim3(v.B), dim3(1024), 0, s, v, p0, Tf, tf_stride);
}

// Blocked row-Householder LQ of the rows_total x 2D stack [T | Mt] in v.R / v.M (rows >= D are updated, never pivoted).
// Two levels: 32-row panels are applied inside their 128-row outer block only; the outer block's aggregated reflector
// (Y = 128 scaled rows, 128 x 128 factor) then updates all rows below in one pass: G = T[:, blk] + Mt V^T, W = G Tf,
// T[:, blk] -= W, Mt -= W V  -- a quarter of the trailing-matrix traffic of panel-by-panel updates.
static void blocked_lq(const DenseView& v, int rows_total, hipStream_t s) {
  const int B = v.B, D = v.D;
  const long sDD = long(v.mr_rows) * D * D;
  const long sG = long(rows_total) * 128;
  const int tfs = 4 * PB * PB;  // panel factors of one outer block (at most four panels)
  for (int o0 = 0; o0 < D; o0 += 128) {
    const int OB = D - o0 < 128 ? D - o0 : 128, nb = OB / PB;  // (the last outer block may be shorter: D % 128 in {0, 64})
    for (int pi = 0; pi < nb; ++pi) {
      const int p0 = o0 + pi * PB;
      double* Tf = v.lgTf + size_t(pi) * PB * PB;
      switch (D / 64) {
        case 1: launch_panel<4>(v, p0, Tf, tfs, s); break;
        case 2: launch_panel<8>(v, p0, Tf, tfs, s); break;
        case 3: launch_panel<12>(v, p0, Tf, tfs, s); break;
        case 4: launch_panel<16>(v, p0, Tf, tfs, s); break;
        case 5: launch_panel<20>(v, p0, Tf, tfs, s); break;
        case 6: launch_panel<24>(v, p0, Tf, tfs, s); break;
        case 7: launch_panel<28>(v, p0, Tf, tfs, s); break;
        case 8: launch_panel<32>(v, p0, Tf, tfs, s); break;
        case 9: launch_panel<36>(v, p0, Tf, tfs, s); break;
        case 10: launch_panel<40>(v, p0, Tf, tfs, s); break;
        case 11: launch_panel<44>(v, p0, Tf, tfs, s); break;
        case 12: launch_panel<48>(v, p0, Tf, tfs, s); break;
        case 13: launch_panel<52>(v, p0, Tf, tfs, s); break;
        case 14: launch_panel<56>(v, p0, Tf, tfs, s); break;
        case 15: launch_panel<60>(v, p0, Tf, tfs, s); break;
        default: launch_panel<64>(v, p0, Tf, tfs, s); break;
      }
      const int j0 = p0 + PB;
      const int rows = o0 + OB - j0;
      if (rows <= 0) continue;
      double* Mb = v.M + size_t(j0) * D;       // trailing rows of Mt (inside the outer block)
      double* Vp = v.M + size_t(p0) * D;       // scaled reflector rows (32 x D)
      double* Tb = v.R + size_t(j0) * D + p0;  // trailing rows of T, panel columns
      Gemm g1{Mb, Vp, Tb, v.lgG, sDD, sDD, sDD, sG, D, D, D, PB, rows, PB, D, 1.0, 1.0, v.act, nullptr};
      // one workgroup per trajectory would walk all of K alone (latency-bound, a quarter of the CUs busy): K-slices of 256
      // A slice must be a whole number of PD * KS = 64-wide k-steps (D = 576, 704, 832, 896 ... would otherwise
      // accumulate a clamped tail tile more than once or drop columns): among 2, 3, 4 slices take the one whose
      // slice length is closest to 256, none if D / 64 has no such divisor.
      g1.ksplit = 1;
      if (D >= 512) {
        int best = 1 << 30;
        for (int ks = 2; ks <= 4; ++ks)
          if (D % (64 * ks) == 0 && abs(D / ks - 256) < best) {
            best = abs(D / ks - 256);
            g1.ksplit = ks;
          }
      }
      g1.sPart = long(rows) * PB;
      gemm(v, s, B, true, g1);  // G = T[:, panel] + Mt V^T
      // K-slices summed, W = G Tf, T[:, panel] -= G Tf
      hipLaunchKernelGGL(panel_gw_kernel, dim3((rows + 31) / 32, B), dim3(256), 0, s, v.lgG, sG, g1.sPart, g1.ksplit, Tf, long(tfs), v.lgW, Tb, sDD, D,
                         rows, v.act);
      Gemm g4{v.lgW, Vp, Mb, Mb, sG, sDD, sDD, sDD, PB, D, D, D, rows, D, PB, -1.0, 1.0, v.act, nullptr};
      gemm(v, s, B, false, g4);  // Mt -= W V
    }
    const int j0 = o0 + OB, rows = rows_total - j0;
    if (rows <= 0) continue;
    double* Vo = v.M + size_t(o0) * D;        // OB scaled reflector rows
    double* Mb = v.M + size_t(j0) * D;
    double* Tb = v.R + size_t(j0) * D + o0;
    const long sT = 128 * 128;
    Gemm gg{Vo, Vo, nullptr, v.lgGram, sDD, sDD, 0, sT, D, D, 0, OB, OB, OB, D, 1.0, 0.0, v.act, nullptr};
    gemm(v, s, B, true, gg);  // Gram = V V^T
    {
      int dev = 0;
      (void)hipGetDevice(&dev);
      static bool attr[64] = {};  // the opt-in to > 64 KB of dynamic LDS is per function AND per device
      if (!attr[dev & 63]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tf_outer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(kTfOuterShm));
        attr[dev & 63] = true;
      }
    }
    hipLaunchKernelGGL(tf_outer_kernel, dim3(B), dim3(256), kTfOuterShm, s, v.lgTf, v.lgGram, v.lgTfo, nb, v.act);
    Gemm g1{Mb, Vo, Tb, v.lgG, sDD, sDD, sDD, sG, D, D, D, OB, rows, OB, D, 1.0, 1.0, v.act, nullptr};
    gemm(v, s, B, true, g1);
    Gemm g2{v.lgG, v.lgTfo, nullptr, v.lgW, sG, sT, 0, sG, OB, OB, 0, OB, rows, OB, OB, 1.0, 0.0, v.act, nullptr};
    g2.E = Tb; g2.sE = sDD; g2.lde = D; g2.ealpha = -1.0;  // W = G Tf and T[:, blk] -= G Tf: one product, two outputs
    gemm(v, s, B, false, g2);
    Gemm g4{v.lgW, Vo, Mb, Mb, sG, sDD, sDD, sDD, OB, D, D, D, rows, D, OB, -1.0, 1.0, v.act, nullptr};
    gemm(v, s, B, false, g4);
  }
}

}  // namespace lg

void dense_large_lq(const DenseView& v, int rows_total, hipStream_t s) {
  lg::blocked_lq(v, rows_total, s);
}

// After the 2D-row LQ of a predict: G_p = C21_p (L^-_p)^-1 (blocked right triangular solve), plain coordinates,
// xi, merge with the running transition, and the LQ of the merged noise stack [Lam0 | G0 Lam] (rows 0..D-1 of R / M).
void dense_large_fp_backward(const DenseView& v, const FpArg* fa, const IwpConst* K, hipStream_t s) {
  using namespace lg;
  const int B = v.B, D = v.D;
  const long DD = long(D) * D, sDD = 2 * DD;
  hipLaunchKernelGGL(fp_setup_kernel, dim3((B + 63) / 64), dim3(64), 0, s, v, fa);
  double* C21 = v.R + DD;  // rows D..2D-1 of R
  for (int jb = D / PB - 1; jb >= 0; --jb) {
    hipLaunchKernelGGL(trsm_right_block_kernel, dim3((D + 255) / 256, B), dim3(256), 0, s, v, jb);
    if (jb > 0) {
      Gemm g{C21 + jb * PB, v.R + size_t(jb) * PB * D, C21, C21, sDD, sDD, sDD, sDD, D, D, D, D, D, jb * PB, PB, -1.0, 1.0, v.act,
             nullptr};
      gemm(v, s, B, false, g);  // C21[:, 0:jb*32] -= X[:, jb] Lm[jb, 0:jb*32]
    }
  }
  hipLaunchKernelGGL(fp_scale_kernel, dim3(64, B), dim3(256), 0, s, v, fa, K);
  hipLaunchKernelGGL(fp_xi_kernel, dim3(32, B), dim3(256), 0, s, v, fa);
  {
    Gemm gd{v.cG, C21, nullptr, v.cG, 0, sDD, 0, 0, D, D, 0, D, D, D, D, 1.0, 0.0, v.lgMaskMerge, v.lgOffD, v.lgOffS, nullptr};
    gemm(v, s, B, false, gd);  // G_dst = G0 G
    Gemm gl{v.cG, v.M + DD, nullptr, v.M, 0, sDD, 0, sDD, D, D, 0, D, D, D, D, 1.0, 0.0, v.lgMaskMerge, nullptr, v.lgOffS, nullptr};
    gemm(v, s, B, false, gl);  // Mt := G0 Lam
  }
  hipLaunchKernelGGL(fp_merge_misc_kernel, dim3(64, B), dim3(256), 0, s, v, fa);
  blocked_lq(v, D, s);
}

bool dense_large_supported(const DenseView& v) {
  return v.ts1 && v.d % 32 == 0 && v.D % 64 == 0 && v.D >= 64 && v.D <= 1024 && v.lgG != nullptr;
}

// Filter: D beyond the register-resident MFMA path. Fixed-point smoother: every supported shape (the register-resident
// kernels have no extra-row LQ; the alternative is the generic kernels). pnode_set_generic_path(h, 2) forces the path,
// (h, 1) forbids it.
bool dense_use_large(const DenseView& v) {
  if (!dense_large_supported(v)) return false;
  if (v.force_large) return true;
  if (v.user_generic) return false;
  return v.fixedpoint ? true : !dense_mfma_supported(v);
}

void dense_large_launch_step(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s, const PhaseEvents* ev) {
  using namespace lg;
  auto mark = [&](int i) {
    if (ev) (void)hipEventRecord(ev->ev[i], s);
  };
  const int B = v.B, d = v.d, k = v.k, D = v.D, kd = k * d;
  const long sDD = long(v.mr_rows) * D * D;
  // ---- begin
  const size_t shm_a = (size_t(kd) + d + size_t(v.n) * v.n) * sizeof(double);
  hipLaunchKernelGGL(begin_a_kernel, dim3(d >= 128 ? 8 : 2, B), dim3(256), shm_a, s, v, args, K);
  {
    Gemm g{v.X, v.X, nullptr, v.S, long(d) * D, long(d) * D, 0, long(d) * d, D, D, 0, d, d, d, (k + 1) * d, 1.0, 0.0, v.act, nullptr};
    gemm(v, s, B, true, g);
  }
  potrf(s, v);
  hipLaunchKernelGGL(begin_c_kernel, dim3(B), dim3(256), 0, s, v, args, v.lgW);
  mark(1);
  // ---- time update: blocked row-Householder LQ of [T | Mt]
  dense_launch_lq_input(v, args, K, s);
  mark(2);
  blocked_lq(v, v.mr_rows * D, s);
  mark(3);
  // ---- measurement update (projector form)
  hipLaunchKernelGGL(scale_t_kernel, dim3(64, B), dim3(256), 0, s, v, args, K);
  {  // F = L^-[k-block rows] - J L^-[0:kd rows]      (d x D)
    Gemm f{v.J, v.M, v.M + size_t(kd) * D, v.X, long(d) * kd, sDD, sDD, long(d) * D, kd, D, D, D, d, D, kd, -1.0, 1.0, v.act, nullptr};
    gemm(v, s, B, false, f);
  }
  {
    Gemm g{v.X, v.X, nullptr, v.S, long(d) * D, long(d) * D, 0, long(d) * d, D, D, 0, d, d, d, D, 1.0, 0.0, v.act, nullptr};
    gemm(v, s, B, true, g);  // S = F F^T
  }
  potrf(s, v);
  hipLaunchKernelGGL(solve_wz_kernel, dim3(B), dim3(256), 0, s, v, args);
  for (int ib = 0; ib < d / PB; ++ib) {  // W = C^-1 F, right-looking blocked forward substitution
    hipLaunchKernelGGL(trsm_block_kernel, dim3((D + 255) / 256, B), dim3(256), 0, s, v.S, v.X, long(d) * D, D, D, d, ib, v.act);
    const int rem = d - (ib + 1) * PB;
    if (rem > 0) {
      Gemm g{v.S + size_t(ib + 1) * PB * d + ib * PB, v.X + size_t(ib) * PB * D, v.X + size_t(ib + 1) * PB * D,
             v.X + size_t(ib + 1) * PB * D, long(d) * d, long(d) * D, long(d) * D, long(d) * D, d, D, D, D, rem, D, PB, -1.0, 1.0,
             v.act, nullptr};
      gemm(v, s, B, false, g);
    }
  }
  {
    Gemm y{v.M, v.X, nullptr, v.Ft, sDD, long(d) * D, 0, long(D) * d, D, D, 0, d, D, d, D, 1.0, 0.0, v.act, nullptr};
    gemm(v, s, B, true, y);  // Y = L^- W^T   (D x d)
    // L = L^- - Y W, written to each trajectory's destination slot (per-trajectory offsets)
    Gemm u{v.Ft, v.X, v.M, v.U, long(D) * d, long(d) * D, sDD, 0, d, D, D, D, D, D, d, -1.0, 1.0, v.act, v.lgOff};
    gemm(v, s, B, false, u);
  }
  hipLaunchKernelGGL(finish_mean_kernel, dim3(32, B), dim3(256), 0, s, v, args);
  hipLaunchKernelGGL(finish_kernel, dim3(B), dim3(256), 0, s, v, args);
  mark(4);
}

}  // namespace pnode
