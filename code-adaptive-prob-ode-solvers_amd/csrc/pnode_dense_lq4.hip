// Time update of the dense EKF1 fast path, ONE launch, overlapping passes: blocked row-Householder LQ of the D x 2D stack
// [T0 | Mt] with 64-row passes, one workgroup of 8 wavefronts per trajectory, two groups of four wavefronts (A = 0..3,
// B = 4..7) that ALTERNATE as the panel group of the passes.
//
//   * A wavefront holds one 16-row strip of Mt (full width) in registers, a group one 64-row block. In pass P the panel group
//     holds block P: wavefront j applies the panels 0 .. j-1 of the pass as they are published, factorises its own strip in
//     registers (pnode_lq_sweep.hpp: DPP-operand FMAs, no LDS, no barrier) and publishes it as panel j (V = [e ; scl x], the
//     compact-WY factor and the scales, in an LDS slot). The other group holds block P + 1 and applies every panel the moment it
//     is published, so that ITS chain (pass P + 1) starts as soon as panel 3 is applied -- beside the remaining updates of pass P,
//     which the wavefronts of the former panel group stream as they come free (strip j of the blocks P + 3 .. back to HBM, strip j
//     of block P + 2 last: it is kept and is that wavefront's panel in pass P + 2).
//   * The serial chain of the 4 D/64 x 4 sweeps, which bounds the kernel, therefore never waits for trailing updates, and at
//     D = 256 only one 64-row block makes one round trip through HBM (none at D <= 192).
//   * Persistent workgroups, one per CU: each walks over the trajectories blockIdx.x, blockIdx.x + gridDim.x, .. and the passes
//     of consecutive trajectories form ONE sequence (global pass G = NBLK it + P): group A loads and factorises block 0 of the
//     next trajectory while group B is still in the last pass of the current one -- one panel BEHIND it: the wavefronts j of the
//     two groups share SIMD j, two sweeps side by side on one SIMD take twice as long, so the chain of the first pass works on
//     SIMD j only once the chain of the last pass has left it (`follows` below).
//   * FIVE LDS slots for the four panels of a pass: panel j of global pass G has the sequence number n = 4 G + j and lives in slot
//     n % 5, so that the first panel of a pass does not wait for the last trailing update of the pass before.
//   * No workgroup barrier after the prologue. pubS[s] = n + 1: panel n is in slot s; nextS[w] = n: wavefront w will not read a
//     panel with a sequence number < n again (the writer of panel n waits until all eight are beyond n - 5; per-wavefront words,
//     not a shared counter); storedS[j]: strip j of the blocks parked in HBM by pass G. Every wait is a bounded spin (a wavefront
//     that runs into the limit reports it in the handle's fault words and raises the abort word, all waits then fall through,
//     the outputs of the trajectory are NaN and the API call returns PNODE_EDEVICE).
//   * scripts/lq_timeline.py (diagnostic build) prints when every panel of a trajectory arrived at its sweep, was swept, had its
//     slot and was published: what is left beside the chain are collisions of the chain's wavefront with the trailing MFMAs of
//     the other wavefront of its SIMD (profiles/r03_lq_timeline.txt).
//   * (A column-split variant -- four wavefronts sweeping a quarter of the columns each -- was measured and dropped: an LDS
//     round trip costs 300-400 cycles under load, so the per-pivot exchange of the partial dot products, 850 cycles, outweighs
//     the four times shorter DPP sweep.)
//
// Reference anchors: strategy_filter (/root/reference/experiments/1_van_der_pol/vdp.py:65) as probdiffeq runs it (sqrt
// time update by QR of the stacked factor); arithmetic restated in oracle/pn_oracle.py (predict_sqrt).
#include "pnode_lq_sweep.hpp"

namespace pnode {
namespace lq4 {

constexpr int kSpinLimit = 1 << 21;  // default bound of every wait (polls); DenseView::lq_spin_limit overrides it (diagnostics)
// The words the wavefronts talk through are accessed through pointers that CARRY the LDS address space: a volatile access
// through a generic pointer is not rewritten by the address-space inference and becomes a flat_load / flat_store.
using lds_int = __attribute__((address_space(3))) int;

// Cycle stamps of the diagnostic build (make timing); empty otherwise.
struct Stamps {
#ifdef PNODE_TIMING_BUILD
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prev = 0;
  unsigned long long when[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // absolute times of the events of this wavefront's (up to two) passes
  __device__ __forceinline__ void event(int slot) { when[slot] = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void start() { prev = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void mark(int slot) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    acc[slot] += now - prev;
    prev = now;
  }
#else
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void event(int) {}
#endif
};

}  // namespace lq4

template <int DT, int DD>
__global__ __launch_bounds__(512) void dense_lq4_mfma_kernel(const DenseView v, const StepArg* __restrict__ args,
                                                              const IwpConst* __restrict__ Kc) {
  using namespace lq4;
  static_assert(DT % 4 == 0, "whole 64-row blocks");
  static_assert((2 * 16 * DD) % 64 == 0, "a pass is all below or all beyond column 2d");
  constexpr int D = 16 * DT, d = 16 * DD, NBLK = DT / 4;
  // Mt = (Abar P^-1 (x) I) Ls has the D - d columns of the rank-deficient state factor (pnode_common.hpp), padded to DC tiles
  constexpr int DC = mfma_col_tiles(DT, DD), WC = 16 * DC;
  constexpr int LDV = WC + 1;
  constexpr int DCT = 2 * DD;  // 16-column tiles of the T scratch (columns < 2d)  // odd: the A-operand reads of the G phase (16 rows x 4 column groups) spread over the banks
  // FIVE panel slots for the four panels of a pass: panel j of global pass G has the sequence number n = 4 G + j and lives in
  // slot n % 5, so a panel may be published while the panel FOUR places before it is still being read -- the first panel of a
  // pass does not wait for the last trailing update of the pass before to let go of "its" slot (scripts/lq_timeline.py: that
  // wait was 30-50 k cycles of every trajectory at D = 256).
  constexpr int NS = 5;
  __shared__ double Vx[NS * kNB * LDV];
  __shared__ double TfS[NS * kNB * kNB];
  __shared__ double sclS[NS * kNB];
  __shared__ double TppS[8 * kNB * kNB];  // diagonal blocks of T, staged by the sweeping wavefront for a coalesced write (one region per
                                          // wavefront: the sweeps of wavefronts j of the two groups overlap across trajectories)
  __shared__ double coefS[2][kMaxN * kMaxN];  // per trajectory, double-buffered by the parity of the workgroup's trajectory count
  __shared__ double pSS[2][kMaxN];
  __shared__ int pubS[NS], storedS[4];  // pubS[s] = n + 1: panel n is in slot s = n % NS
  __shared__ int nextS[8];  // nextS[w] = n: wavefront w will not read a panel with a sequence number < n again
  __shared__ int prepS;  // number of trajectories whose coefS / pSS are written
  __shared__ int trajS[8];  // trajS[w]: trajectories wavefront w has finished
  __shared__ int abortS;

  // PERSISTENT workgroups: the grid is one workgroup per CU (at most), each walks over the trajectories b = blockIdx.x,
  // blockIdx.x + gridDim.x, .. WITHOUT a barrier in between: the passes of consecutive trajectories form one sequence (global
  // pass number base + P), so group A loads and factorises block 0 of the next trajectory while group B is still in the last
  // pass of the current one -- the tail of a trajectory (one group idle) overlaps the head of the next (the other group idle).
  const IwpConst& K = *Kc;
  const int nn = v.n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gi = wv >> 2, j = wv & 3;  // (wavefront w runs on SIMD w & 3: strip j of both groups on SIMD j)
  if (tid == 0) {
    for (int k = 0; k < 4; ++k) storedS[k] = 0;
    for (int k = 0; k < NS; ++k) pubS[k] = 0;
    for (int k = 0; k < 8; ++k) trajS[k] = nextS[k] = 0;
    prepS = 0;
    abortS = 0;
  }
  __syncthreads();  // the only workgroup barrier
  // per trajectory (set at the top of the trajectory loop; the lambdas below see them by reference)
  int b = 0, base = 0;
  StepArg sarg = args[blockIdx.x];
  double sigma = 0.0;
  double* __restrict__ Mt = nullptr;
  double* __restrict__ T = nullptr;
  const double* __restrict__ Lsrc = nullptr;
  double* __restrict__ Ldst = nullptr;
  const double* coef = coefS[0];
  const double* pS = pSS[0];

  volatile lds_int* abort_word = (volatile lds_int*)&abortS;
  const int spin_limit = v.lq_spin_limit > 0 ? v.lq_spin_limit : kSpinLimit;
  // A wait that runs into its limit is a synchronisation FAILURE, not a numerical result: it is recorded in the handle's fault
  // words (pinned host memory; the API returns PNODE_EDEVICE after the stream synchronisation), every wait of the launch then
  // falls through so that the grid drains, and the outputs of the trajectories the workgroup still takes are NaN.
  auto timed_out = [&](int n, int what, int target) {
    if (n <= spin_limit && !__builtin_amdgcn_readfirstlane(*abort_word)) return false;
    if (n > spin_limit && (tid & 63) == 0 && v.fault) {
      v.fault[1] = b;
      v.fault[2] = target;
      v.fault[3] = 8 * what + wv;
      __atomic_store_n(v.fault, 1, __ATOMIC_RELEASE);
    }
    *abort_word = 1;
    return true;
  };
  auto spin_ge = [&](int* f_, int target) {
    volatile lds_int* f = (volatile lds_int*)f_;
    int n = 0;
    while (__builtin_amdgcn_readfirstlane(*f) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (((++n & 255) == 0 || n > spin_limit) && timed_out(n, 0, target)) break;
    }
    asm volatile("" ::: "memory");
  };
  auto lds_done = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  // Slot bookkeeping by per-wavefront progress words: a wavefront reads the panels in the order of their sequence numbers, except
  // that the trailing updates of a pass go through its four panels once per block -- its word then moves on the last block only.
  auto seq = [&](int pass, int k) { return 4 * (base + pass) + k; };
  auto wait_published = [&](int pass, int k) {
    const int n = seq(pass, k);
    spin_ge(&pubS[n % NS], n + 1);
  };
  auto done_before = [&](int n) {  // my LDS reads issued so far have completed before the word moves
    lds_done();
    if (lane == 0) *(volatile lds_int*)&nextS[wv] = n;
  };
  auto done_through = [&](int pass, int k) { done_before(seq(pass, k) + 1); };
  auto wait_slot_free = [&](int n) {  // every wavefront is done with the panel this slot held before panel n
    int polls = 0;
    while (true) {
      const int mine = (lane < 8) ? *(volatile lds_int*)&nextS[lane] : 0x7fffffff;
      if (!__any(mine < n - NS + 1)) break;
      __builtin_amdgcn_s_sleep(2);
      if (((++polls & 255) == 0 || polls > spin_limit) && timed_out(polls, 1, n)) break;
    }
    asm volatile("" ::: "memory");
  };
  // (lane coordinates made opaque per call site: otherwise every LDS / HBM address of every call is hoisted out of the pass
  // loop as a loop invariant and the hoisted addresses alone overflow the register file)
  auto opaque = [](int x) {
    asm volatile("" : "+v"(x));
    return x;
  };
  Stamps st;

  auto t0_tile = [&](int row0, int colp, int r, int g) {
    // T0[row0 + r][colp + g + 4q] = sigma chol(Qbar)[ib][jb] where the intra-block indices agree
    double4_t t;
    const int ib = row0 / d, jb = colp / d;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      t[q] = (((row0 + r) % d) == ((colp + g + 4 * q) % d) && jb <= ib) ? sigma * K.qsT[jb][ib] : 0.0;
    return t;
  };

  double X[4 * DC];  // my strip: X[16 T + e] = M[row0 + r][64 T + 16 g + e]

  auto load_strip = [&](int st_, bool from_state) {
    const int ln = opaque(tid & 63);
    const int j0 = kNB * st_;
    if (from_state) {
      // Mt rows of derivative block ib: sum_{k >= ib} coef[ib][k] L[k d + q]: one round trip to memory per 64-column group and
      // k, eight 16-byte loads per lane in flight (sixteen, two terms per round trip, measured slower); k runs downwards so that wavefronts which start together ask for the same
      // rows of L at the same time
      const int ib = j0 / d;
#pragma unroll
      for (int e = 0; e < 4 * DC; ++e) X[e] = 0.0;
#pragma unroll
      for (int T4 = 0; T4 < DC / 4; ++T4) {
#pragma unroll 1
        for (int k = nn - 1; k >= ib; --k) {
          const double ck = coef[ib * kMaxN + k];
          const double2* __restrict__ lrow = reinterpret_cast<const double2*>(Lsrc) + (size_t((k * d + j0 % d) / kNB) * (2 * DC) + 8 * T4) * 64 + ln;
          double2 tmp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) tmp[e] = lrow[e * 64];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            X[16 * T4 + 2 * e] = fma(ck, tmp[e].x, X[16 * T4 + 2 * e]);
            X[16 * T4 + 2 * e + 1] = fma(ck, tmp[e].y, X[16 * T4 + 2 * e + 1]);
          }
        }
      }
    } else {
      const double2* __restrict__ src = reinterpret_cast<const double2*>(Mt) + size_t(st_) * (2 * DC) * 64 + ln;
#pragma unroll
      for (int pp = 0; pp < 2 * DC; ++pp) {
        const double2 t2 = src[pp * 64];
        X[2 * pp] = t2.x;
        X[2 * pp + 1] = t2.y;
      }
    }
  };
  auto store_strip = [&](int st_) {
    const int ln = opaque(tid & 63);
    double2* __restrict__ dst = reinterpret_cast<double2*>(Mt) + size_t(st_) * (2 * DC) * 64 + ln;
#pragma unroll
    for (int pp = 0; pp < 2 * DC; ++pp) dst[pp * 64] = double2{X[2 * pp], X[2 * pp + 1]};
  };
  // Columns >= 2d of L^- are not touched by the measurement update (F = H L^- has 2d non-zero columns): the passes that
  // produce them write p_row * T straight into the destination slot of the state factor.
  auto is_direct = [&](int pass) { return 64 * pass >= 2 * d; };
  // panel k of pass `pass` applied to my strip (global strip index st_), once it is published. (Raising the issue priority of
  // the applications the chain is waiting for changed nothing measurable.)
  auto apply_panel = [&](int pass, int k, int st_) {
    const int sl = seq(pass, k) % NS;
    spin_ge(&pubS[sl], seq(pass, k) + 1);
    st.mark(1);
    const int r = opaque(lane & 15), g = opaque(lane >> 4);
    const int j0 = kNB * st_, cp = 64 * pass + kNB * k;
    double4_t tv;
    apply_panel_to_strip_pipelined<DC, LDV>(X, Vx + sl * kNB * LDV, TfS + sl * kNB * kNB, sclS + sl * kNB, t0_tile(j0, cp, r, g), tv, r, g);
    const bool direct = is_direct(pass);
    if (direct) {  // strip order of the state factor (column c of L^- is column c - d of Ls): two pairs per lane, 16 lanes of a pair contiguous
      tv = transpose_groups(tv);  // lane (r, g): columns cp + 4 g .. cp + 4 g + 3 of row j0 + r
      const double osc = pS[(j0 + r) / d];
      double* base = Ldst + tiled_index<DC>(j0 + r, cp + 4 * g - d);
      *reinterpret_cast<double2*>(base) = double2{osc * tv[0], osc * tv[1]};
      *reinterpret_cast<double2*>(base + 128) = double2{osc * tv[2], osc * tv[3]};  // next pair: + 64 lanes x 2 doubles
    } else {  // T scratch: the C/D fragment as it is (lane (r, g): columns cp + g + 4 q), tile-aligned strip order (tiledpi_index)
      double* base = T + tiledpi_index<DCT>(j0 + r, cp + g);
      *reinterpret_cast<double2*>(base) = double2{tv[0], tv[1]};
      *reinterpret_cast<double2*>(base + 128) = double2{tv[2], tv[3]};
    }
    st.mark(2);
  };

  st.start();
  int it = 0;  // trajectories this workgroup has taken so far
#pragma unroll 1
  for (b = blockIdx.x; b < v.B; b += gridDim.x) {
    sarg = args[b];
    if (!sarg.active) continue;  // (uniform over the workgroup)
    base = NBLK * it;
    sigma = v.sig_new[b];
    Mt = v.M + size_t(b) * D * D;
    T = v.R + size_t(b) * D * D;
    Lsrc = v.U + (size_t(sarg.src) * v.B + b) * D * D;
    Ldst = v.U + (size_t(sarg.dst) * v.B + b) * D * D;
    coef = coefS[it & 1];
    pS = pSS[it & 1];
    if (wv == 0) {
      // Mt = (Abar P^-1 (x) I) L is formed while the strips are read: coef[i][k] = Abar[i][k] / p_k. The tables of trajectory
      // it - 2 are overwritten: wait until every wavefront has finished that trajectory (wavefront 0 runs ahead of the others:
      // at D = 128 it is done with a trajectory after ONE sweep, while the last pass still scales its output with pS).
      if (it >= 2) {
        int n = 0;
        while (true) {
          const int mine = (lane < 8) ? *(volatile lds_int*)&trajS[lane] : 0x7fffffff;
          if (!__any(mine < it - 1)) break;
          __builtin_amdgcn_s_sleep(4);
          if (((++n & 255) == 0 || n > spin_limit) && timed_out(n, 2, it - 1)) break;
        }
        asm volatile("" ::: "memory");
      }
      {
        // one table entry per lane, in registers (a single lane filling the tables through local arrays took 25 k cycles
        // at the head of every trajectory); the powers are formed by the multiplication sequence of precond_rt: same bits
        const double adt = fabs(sarg.dt);
        double* cw = coefS[it & 1];
        double* pw_out = pSS[it & 1];
#pragma unroll
        for (int e0 = 0; e0 < kMaxN * kMaxN; e0 += 64) {
          const int e = e0 + lane;
          if (e < kMaxN * kMaxN) {
            const int i = e / kMaxN, k = e % kMaxN;
            double val = 0.0;
            if (i < nn && k < nn && k >= i) {
              double pw = sqrt(adt);
              for (int m = nn - 1; m > k; --m) pw *= adt;
              val = K.a[i][k] * (K.fact[k] / pw);
            }
            cw[e] = val;
          }
        }
        if (lane < kMaxN) {
          double val = 0.0;
          if (lane < nn) {
            double pw = sqrt(adt);
            for (int m = nn - 1; m > lane; --m) pw *= adt;
            val = pw / K.fact[lane];
          }
          pw_out[lane] = val;
        }
      }
      lds_done();
      if (lane == 0) *(volatile lds_int*)&prepS = it + 1;
    }
    spin_ge(&prepS, it + 1);
    // Prologue: the wavefronts of group A start with the strips of block 0, those of group B with block 1, to which they apply
    // the panels of pass 0 as they come. After that both groups run the same straight loop over THEIR passes P = gi, gi + 2, ..
    // (one straight body: with the two roles as the arms of a branch inside the loop the register allocator keeps part of the
    // 128-register strip in scratch around the join).
    load_strip(4 * gi + j, true);
    st.mark(0);
    if (gi == 1) {
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        // The two wavefronts j of the two groups share a SIMD, and v_mfma_f64 shares the SIMD's FP64 pipe with the DPP FMAs of the sweep.
        // While the other group runs a chain, its wavefront j is on the critical path from the publication of panel j - 1 until it has
        // published panel j (8.4 k cycles of MFMA + 16.4 k of DPP FMAs that nothing should share): I apply panel j - 1 only once
        // panel j is out. (No cycle: its publication waits for marks I set before I entered this phase. Measured: 662 k -> 594 k
        // cycles per trajectory. s_setprio does not arbitrate the matrix core; a flag raised and polled around the critical
        // application only moved the collision into the sweep.)
        if (k == j - 1) wait_published(0, j);
        apply_panel(0, k, 4 + j);
        done_through(0, k);
      }
    }
#pragma unroll 1
    for (int P = gi; P < NBLK; P += 2) {
      const bool has_trailing = P + 2 < NBLK;
      const int mystrip = 4 * P + j;
      // The first pass of a trajectory (group A) runs beside the last pass of the one before when that is group B's, and the slot
      // hand-over ties panel j of the one to panel j of the other: the two wavefronts j share SIMD j, and two sweeps side by side
      // on one SIMD take 31 k cycles instead of 15 k each (scripts/lq_timeline.py: 44 k per panel of both passes). So the chain of
      // the first pass FOLLOWS the other by one panel: its work on SIMD j -- the application of panel j - 1 to strip j, sweep j --
      // starts when the other chain has left that SIMD (its panel j is out).
      const bool follows = P == 0 && it > 0 && ((NBLK - 1) & 1) == 1;
      // ================= chain of pass P: the panels before mine, then my strip becomes panel j
      // (unrolled with guards: as a loop with a run-time trip count, with the strip live into and out of it, the register allocator
      // parks the strip in scratch)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        if (kk < j) {
          if (kk == j - 1 && follows) wait_published(-1, j);
          apply_panel(P, kk, mystrip);
          if (!has_trailing) done_through(P, kk);  // (with trailing blocks the panels of the pass are read again)
        }
      }
      {
        const int cp = 64 * P + kNB * j;
        const int r = opaque(lane & 15), g = opaque(lane >> 4);
        const double alpha = sigma * K.qsT[cp / d][cp / d];
        double tfrow[kNB], myscl = 0.0;
#pragma unroll
        for (int m = 0; m < kNB; ++m) tfrow[m] = 0.0;
        double* tpp_row = TppS + wv * kNB * kNB + r * kNB;
        if (j == 0 && follows) {
          wait_published(-1, 0);
          st.mark(1);
        }
        st.event(4 * (P >> 1));  // arrived at the sweep
        __builtin_amdgcn_s_setprio(3);
        pivot_sweep<0, DC>(X, tfrow, tpp_row, myscl, alpha);
        __builtin_amdgcn_s_setprio(0);
        st.mark(3);
        st.event(4 * (P >> 1) + 1);  // swept
        const int myseq = seq(P, j), sl = myseq % NS;
        wait_slot_free(myseq);  // every wavefront is done with the panel the slot held before
        st.mark(4);
        st.event(4 * (P >> 1) + 2);  // slot free
        double* Vp = Vx + sl * kNB * LDV;
#pragma unroll
        for (int T4 = 0; T4 < DC / 4; ++T4)
#pragma unroll
          for (int e = 0; e < 16; ++e) Vp[r * LDV + 64 * T4 + 16 * g + e] = X[16 * T4 + e];
        if (g == 0) {
          sclS[sl * kNB + r] = myscl;
#pragma unroll
          for (int m = 0; m < kNB; ++m) TfS[sl * kNB * kNB + r * kNB + m] = tfrow[m];
        }
        lds_done();
        if (lane == 0) *(volatile lds_int*)&pubS[sl] = myseq + 1;
        st.event(4 * (P >> 1) + 3);  // published
        {
          // diagonal block of T (zeros above the diagonal; the sweep leaves the upper triangle of tpp unwritten)
          const int row = cp + r;
          if (is_direct(P)) {  // state factor: lane (r, g) writes columns 4g .. 4g+3 of row r
            const double osc = pS[row / d];
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const double2 t2{osc * ((4 * g + q <= r) ? tpp_row[4 * g + q] : 0.0), osc * ((4 * g + q + 1 <= r) ? tpp_row[4 * g + q + 1] : 0.0)};
              *reinterpret_cast<double2*>(Ldst + tiled_index<DC>(row, cp + 4 * g + q - d)) = t2;  // (4 g + q is even: the pair is contiguous)
            }
          } else {  // T scratch: lane (r, g) writes columns g + 4q
            double o4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o4[q] = (g + 4 * q <= r) ? tpp_row[g + 4 * q] : 0.0;
            double* base = T + tiledpi_index<DCT>(row, cp + g);
            *reinterpret_cast<double2*>(base) = double2{o4[0], o4[1]};
            *reinterpret_cast<double2*>(base + 128) = double2{o4[2], o4[3]};
          }
        }
        st.mark(5);
      }
      // (no further role in this trajectory after a pass without trailing blocks: the next panel I read is one of the next trajectory)
      if (!has_trailing) done_before(seq(NBLK, 0));
      // ================= the other blocks of pass P: strip j of the blocks P + 3 .. goes back to HBM, strip j of block P + 2 stays
#pragma unroll 1
      for (int blk = NBLK - 1; blk >= P + 2; --blk) {
        if (P > 0) {
          spin_ge(&storedS[j], base + P);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        load_strip(4 * blk + j, P == 0);
        st.mark(0);
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
          apply_panel(P, k, 4 * blk + j);
          if (blk == P + 2) done_through(P, k);
        }
        if (blk != P + 2) store_strip(4 * blk + j);
        st.mark(6);
      }
      if (NBLK - 1 >= P + 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // my stores have left before the flag moves
        if (lane == 0) *(volatile lds_int*)&storedS[j] = base + P + 1;
      }
      // ================= the panels of pass P + 1 (the other group's chain) on strip j of block P + 2, my next panel
      if (has_trailing) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
          if (k == j - 1) wait_published(P + 1, j);  // (stay off the SIMD while its other wavefront is the chain's, see the prologue)
          apply_panel(P + 1, k, 4 * (P + 2) + j);
          done_through(P + 1, k);
        }
      }
    }
    // the group that has no role in the last pass (its panels are applied inside the panel group only) signs off for it
    if (gi == (NBLK & 1)) done_before(seq(NBLK, 0));
    ++it;
    lds_done();
    if (lane == 0) *(volatile lds_int*)&trajS[wv] = it;
#ifdef PNODE_TIMING_BUILD
    if (lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(v.S + size_t(b) * d * d) + wv * 8;
      for (int i = 0; i < 8; ++i) dbg[i] = st.acc[i];
      for (int i = 0; i < 8; ++i) dbg[64 + i] = st.when[i];
    }
#endif
    if (*abort_word) {
      // a wait ran into its limit: make the failure visible in every output of the step
      for (int i = lane; i < D * D; i += 64) {
        T[i] = __builtin_nan("");
        Ldst[i] = __builtin_nan("");
      }
    }
  }
}

bool dense_lq4_supported(const DenseView& v) {
  const int DT = v.D / 16, DD = v.d / 16;
  return v.d % 16 == 0 && ((DT == 8 && DD == 2) || (DT == 12 && DD == 2) || (DT == 16 && DD == 4));
}

void dense_launch_lq4(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  // one persistent workgroup per CU (a workgroup needs a CU's LDS to itself), each taking every gridDim.x-th trajectory
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& ncu = cus[dev & 63];
  if (ncu == 0) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  const dim3 grid(v.B < ncu ? v.B : ncu);
  switch (v.D / 16) {
    case 8: hipLaunchKernelGGL((dense_lq4_mfma_kernel<8, 2>), grid, dim3(512), 0, s, v, args, K); break;
    case 12: hipLaunchKernelGGL((dense_lq4_mfma_kernel<12, 2>), grid, dim3(512), 0, s, v, args, K); break;
    case 16: hipLaunchKernelGGL((dense_lq4_mfma_kernel<16, 4>), grid, dim3(512), 0, s, v, args, K); break;
    default: break;
  }
}

}  // namespace pnode
