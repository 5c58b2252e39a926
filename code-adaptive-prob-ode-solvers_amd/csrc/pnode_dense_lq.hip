// Time update of the dense EKF1 fast path, ONE launch: blocked row-Householder LQ of the D x 2D stack [T0 | Mt] with
// 64-row passes (NP = 4 panels of 16 rows), one workgroup of 8 wavefronts per trajectory.
//
//   * Every wavefront owns at most one 16-row strip of Mt at a time, resident in registers in the MFMA operand layout
//     (pnode_mfma_common.hpp). A strip is read from HBM once per PASS (64 rows of reflectors are applied to it while it
//     stays in registers) and the strips that become the panels of the next pass never leave the chip: the wavefront
//     that updated them last keeps them and factorises them in place ("carried" strips). Per trajectory the trailing
//     matrix moves (D/64 - 1)(D/64 - 2)/2 * 64 rows in and out instead of 8 x that with 32-row passes and no carry.
//   * A panel is factorised by ONE wavefront, in registers, in the strip layout itself: lane (r, g) holds row r, the
//     pivot row is broadcast inside each 16-lane DPP row by the DPP operand of v_fmac_f64 (row_newbcast: one instruction
//     per element for the dot products and one for the update), the four column groups are summed by one
//     v_mfma_f64_16x16x4 against a matrix of ones. No LDS round trip, no barrier inside a panel; the compact-WY factor is
//     built by the forward recurrence of dlarft during the sweep. The other wavefronts meanwhile apply the panels that
//     are already published (LDS) to the strips they hold: the chain of the 4 panels of a pass runs beside MFMA work.
//   * Which wavefront holds which strip when is a small table built on the host (lq_build_schedule, checked by a CPU test).
//
// Reference anchors: strategy_filter (/root/reference/experiments/1_van_der_pol/vdp.py:65) as probdiffeq runs it (sqrt
// time update by QR of the stacked factor); arithmetic restated in oracle/pn_oracle.py (predict_sqrt).
#include <cstring>

#include "pnode_lq_sweep.hpp"

namespace pnode {

bool lq_build_schedule(int DT, int NP, LqSchedule* S) {
  if (NP < 1 || NP > kLqMaxNP || DT % NP != 0 || DT / NP > kLqMaxPass || DT / NP < 1) return false;
  std::memset(S, -1, sizeof(*S));
  S->npass = int8_t(DT / NP);
  S->np = int8_t(NP);
  S->dt = int8_t(DT);
  S->pad = 0;
  for (int k = 0; k < NP; ++k) S->pw[0][k] = int8_t(k);
  for (int P = 0; P < S->npass; ++P) {
    bool panel_wave[kLqWaves] = {}, holding[kLqWaves] = {};
    for (int k = 0; k < NP; ++k) {
      if (S->pw[P][k] < 0 || S->pw[P][k] >= kLqWaves || panel_wave[S->pw[P][k]]) return false;
      panel_wave[S->pw[P][k]] = true;
    }
    auto carried = [&](int s) { return P + 1 < S->npass && s >= NP * (P + 1) && s < NP * (P + 2); };
    int lst[64], n = 0;
    for (int s = NP * (P + 2); s < DT; ++s) lst[n++] = s;                                    // plain trailing strips first
    for (int s = NP * (P + 1); s < NP * (P + 2) && s < DT; ++s) lst[n++] = s;                  // the next pass's panels last
    int pos = 0;
    auto give = [&](int w, int s) {
      if (carried(s)) {
        holding[w] = true;
        S->pw[P + 1][s - NP * (P + 1)] = int8_t(w);
      }
    };
    for (int w = 0; w < kLqWaves && pos < n; ++w)
      if (!panel_wave[w]) {
        S->shadow[P][w] = int8_t(lst[pos]);
        give(w, lst[pos++]);
      }
    int rd = 0;
    while (pos < n) {
      if (rd >= kLqMaxRounds) return false;
      bool any = false;
      for (int w = 0; w < kLqWaves && pos < n; ++w)
        if (!holding[w]) {
          S->round[P][rd][w] = int8_t(lst[pos]);
          give(w, lst[pos++]);
          any = true;
        }
      if (!any) return false;
      ++rd;
    }
    S->nrounds[P] = int8_t(rd);
  }
  return true;
}

struct Lq3Args {  // ONE by-value kernel argument: the schedule is read through the kernarg segment pointer (scalar loads with
  DenseView v;    // computed offsets) instead of being copied into SGPRs
  const StepArg* args;
  const IwpConst* Kc;
  LqSchedule S;
};

template <int DT, int NP, int DD>
__global__ __launch_bounds__(64 * kLqWaves) void dense_lq3_mfma_kernel(const Lq3Args ka_) {
  static_assert(DT % 4 == 0, "the HBM strip layout needs D % 64 == 0");
  static_assert(DT % NP == 0, "whole passes");
  static_assert((2 * 16 * DD) % (kNB * NP) == 0, "a pass is all below or all beyond column 2d");
  constexpr int D = 16 * DT, d = 16 * DD;
  const DenseView& v = ka_.v;
  const StepArg* __restrict__ args = ka_.args;
  const IwpConst* __restrict__ Kc = ka_.Kc;
  // the schedule is indexed with run-time (wave-uniform) indices: read it through the kernarg segment (scalar loads)
#if defined(__HIP_DEVICE_COMPILE__)
  const auto* S = &((const __attribute__((address_space(4))) Lq3Args*)__builtin_amdgcn_kernarg_segment_ptr())->S;
#else
  const LqSchedule* S = &ka_.S;
#endif
  constexpr int LDV = D + 1;  // odd: the A-operand reads of the G phase (16 rows x 2 column groups per half wave) hit 32 bank pairs
  __shared__ double Vx[NP * kNB * LDV];
  __shared__ double TfS[NP * kNB * kNB];
  __shared__ double sclS[NP * kNB];
  __shared__ double TppS[NP * kNB * kNB];  // diagonal blocks of T, staged for a coalesced write
  __shared__ double coef[kMaxN * kMaxN];
  __shared__ double pS[kMaxN];
  __shared__ int pubS[NP];  // pubS[k] = P + 1 once panel k of pass P is published (V, Tf, scl in LDS); monotone over the passes

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
  const int r = lane & 15, g = lane >> 4;
  if (tid == 0) {
    // Mt = (Abar P^-1 (x) I) L is formed while the strips are read: coef[i][k] = Abar[i][k] / p_k
    double pl[kMaxN], pil[kMaxN];
    precond_rt(nn, sarg.dt, K, pl, pil);
    for (int i = 0; i < nn; ++i)
      for (int k = 0; k < nn; ++k) coef[i * kMaxN + k] = (k >= i) ? K.a[i][k] * pil[k] : 0.0;
    for (int i = 0; i < kMaxN; ++i) pS[i] = (i < nn) ? pl[i] : 0.0;
    for (int k = 0; k < NP; ++k) pubS[k] = 0;
  }
  __syncthreads();
  // Publication flags instead of workgroup barriers inside a pass: a wavefront waits only for the panel it is about to apply,
  // so the loads of the other wavefronts, their applications and the chain of the four sweeps run decoupled. (LDS operations of
  // one wavefront complete in order; the release fence drains them before the flag store.)
  auto wait_published = [&](int k, int P) {
    volatile int* f = pubS + k;
    while (__builtin_amdgcn_readfirstlane(*f) < P + 1) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };

  auto t0_tile = [&](int row0, int col0) {
    // T0[row0 + r][col0 + g + 4q] = sigma chol(Qbar)[ib][jb] where the intra-block indices agree
    double4_t t;
    const int ib = row0 / d, jb = col0 / d;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      t[q] = (((row0 + r) % d) == ((col0 + g + 4 * q) % d) && jb <= ib) ? sigma * K.qsT[jb][ib] : 0.0;
    return t;
  };
  auto load_strip = [&](int st, double (&X)[4 * DT], bool from_state) {
    const int j0 = kNB * st;
    if (from_state) {
      // Mt rows of derivative block ib: sum_{k >= ib} coef[ib][k] L[k d + q]: one round trip to memory per 64-column group and
      // k, eight 16-byte loads per lane in flight (a larger staging buffer does not fit next to the 128-register strip).
      const int ib = j0 / d;
#pragma unroll
      for (int e = 0; e < 4 * DT; ++e) X[e] = 0.0;
#pragma unroll
      for (int T4 = 0; T4 < DT / 4; ++T4) {
#pragma unroll 1
        for (int k = ib; k < nn; ++k) {
          const double ck = coef[ib * kMaxN + k];
          // (rows k d + q of the state factor form whole strips, stored in register order: coalesced 1 KB per instruction)
          const double2* __restrict__ lrow = reinterpret_cast<const double2*>(Lsrc) + (size_t((k * d + j0 % d) / kNB) * (2 * DT) + 8 * T4) * 64 + lane;
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
      // Mt (scratch private to the LQ) lives in the register order of the strips: pair pp of lane l of strip st at
      // ((st * 2 DT + pp) * 64 + l) * 16 bytes, so every load / store instruction of a wavefront covers 1 KB of contiguous
      // memory (a row-major strip makes each lane touch a cache line of its own and the address unit serialises them)
      const double2* __restrict__ src = reinterpret_cast<const double2*>(Mt) + size_t(st) * (2 * DT) * 64 + lane;
#pragma unroll
      for (int pp = 0; pp < 2 * DT; ++pp) {
        const double2 t2 = src[pp * 64];
        X[2 * pp] = t2.x;
        X[2 * pp + 1] = t2.y;
      }
    }
  };
  auto store_strip = [&](int st, const double (&X)[4 * DT]) {
    double2* __restrict__ dst = reinterpret_cast<double2*>(Mt) + size_t(st) * (2 * DT) * 64 + lane;
#pragma unroll
    for (int pp = 0; pp < 2 * DT; ++pp) {
      double2 t2;
      t2.x = X[2 * pp];
      t2.y = X[2 * pp + 1];
      dst[pp * 64] = t2;
    }
  };

#ifdef PNODE_TIMING_BUILD
  // separate diagnostic build only (make timing): per-wavefront cycle sums by phase, written to the S scratch (unused by the MFMA path)
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define LQ_STAMP(slot)                                              \
  do {                                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    tacc[slot] += now_ - tprev;                                     \
    tprev = now_;                                                   \
  } while (0)
#else
#define LQ_STAMP(slot) \
  do {                 \
  } while (0)
#endif
  const int npass = S->npass;
#pragma unroll 1
  for (int P = 0; P < npass; ++P) {
    // The strip a wavefront holds is NOT carried across this loop in registers: a strip kept for the next pass is parked in the
    // (then free) LDS slot of the panel it will be and read back by the same wavefront at the top of the next pass. (With the
    // 128 registers live across the back edge the register allocator routed them through scratch at every pass: 1.4 MB of
    // HBM reads and as many writes per trajectory.)
    double X[4 * DT];
    bool have = false;
    int mystrip = -1;
    const int c0 = kNB * NP * P;
    // Columns >= 2d of L^- are not touched by the measurement update (F = H L^- has 2d non-zero columns): the passes that
    // produce them write p_row * T straight into the destination slot of the state factor.
    const bool direct = c0 >= 2 * d;
    auto carried = [&](int s) { return P + 1 < npass && s >= NP * (P + 1) && s < NP * (P + 2); };
    auto apply = [&](int k, int st, double (&Xs)[4 * DT]) {
      wait_published(k, P);
      const int j0 = kNB * st, cp = c0 + kNB * k;
      double4_t tv;
      apply_panel_to_strip<DT, LDV>(Xs, Vx + k * kNB * LDV, TfS + k * kNB * kNB, sclS + k * kNB, t0_tile(j0, cp), tv, r, g);
      tv = transpose_groups(tv);  // lane (r, g): columns cp + 4 g .. cp + 4 g + 3 of row j0 + r
      const double osc = direct ? pS[(j0 + r) / d] : 1.0;
      double2 lo2, hi2;
      lo2.x = osc * tv[0]; lo2.y = osc * tv[1]; hi2.x = osc * tv[2]; hi2.y = osc * tv[3];
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
    int mypanel = -1;
#pragma unroll
    for (int k = 0; k < NP; ++k) mypanel = (S->pw[P][k] == wv) ? k : mypanel;
    // task 0: the strip this wavefront holds while the panels of the pass are factorised (a panel of this pass, carried over
    // from the previous one, or a trailing strip); tasks 1 ..: the strips it processes once all panels are published
    const int ntasks = 1 + S->nrounds[P];
#pragma unroll 1
    for (int task = 0; task < ntasks; ++task) {
      if (task == 0) {
        if (mypanel >= 0) {
          mystrip = NP * P + mypanel;
          have = true;
          if (P == 0) {
            load_strip(mystrip, X, true);
          } else {
            const double* Vp = Vx + mypanel * kNB * LDV;  // parked here by this wavefront at the end of the previous pass
#pragma unroll
            for (int T4 = 0; T4 < DT / 4; ++T4)
#pragma unroll
              for (int e = 0; e < 16; ++e) X[16 * T4 + e] = Vp[r * LDV + 64 * T4 + 16 * g + e];
          }
        } else {
          const int s = S->shadow[P][wv];
          if (s >= 0) {
            mystrip = s;
            load_strip(s, X, P == 0);
            have = true;
          }
        }
        LQ_STAMP(0);
#pragma unroll 1
        for (int k = 0; k < NP; ++k) {
          if (have && k > 0) apply(k - 1, mystrip, X);
          LQ_STAMP(1);
          if (k == mypanel) {
            // ---- factorise my strip in place and publish it as panel k
            const int cp = c0 + kNB * k;
            const double alpha = sigma * K.qsT[cp / d][cp / d];
            double tfrow[kNB], myscl = 0.0;
#pragma unroll
            for (int m = 0; m < kNB; ++m) tfrow[m] = 0.0;
            double* tpp_row = TppS + k * kNB * kNB + r * kNB;
            __builtin_amdgcn_s_setprio(3);
            pivot_sweep<0, DT>(X, tfrow, tpp_row, myscl, alpha, r);
            __builtin_amdgcn_s_setprio(0);
            LQ_STAMP(2);
            double* Vp = Vx + k * kNB * LDV;
#pragma unroll
            for (int T4 = 0; T4 < DT / 4; ++T4)
#pragma unroll
              for (int e = 0; e < 16; ++e) Vp[r * LDV + 64 * T4 + 16 * g + e] = X[16 * T4 + e];
            if (g == 0) {
              sclS[k * kNB + r] = myscl;
#pragma unroll
              for (int m = 0; m < kNB; ++m) TfS[k * kNB * kNB + r * kNB + m] = tfrow[m];
            }
            {
              // diagonal block of T: lane (r, g) writes columns 4g .. 4g+3 of row r (zeros above the diagonal)
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (same wavefront wrote the block: program order suffices)
              double o4[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) o4[q] = tpp_row[4 * g + q];
              const int row = cp + r;
              const double osc = direct ? pS[row / d] : 1.0;
#pragma unroll
              for (int q = 0; q < 4; q += 2) {
                double2 t2;
                t2.x = osc * o4[q];
                t2.y = osc * o4[q + 1];
                double* dstp = direct ? Ldst + tiled_index<DT>(row, cp + 4 * g + q) : T + size_t(row) * D + cp + 4 * g + q;
                *reinterpret_cast<double2*>(dstp) = t2;  // (4 g + q is even: the pair is contiguous in both layouts)
              }
            }
            have = false;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) pubS[k] = P + 1;
            LQ_STAMP(3);
          }
        }
      } else {
        const int s = S->round[P][task - 1][wv];
        if (s >= 0) {
          mystrip = s;
          load_strip(s, X, P == 0);
          have = true;
        }
        LQ_STAMP(0);
      }
      if (have) {
        // the panels not applied yet: the last one for the strip held during the chain, all of them otherwise
#pragma unroll 1
        for (int k = (task == 0) ? NP - 1 : 0; k < NP; ++k) apply(k, mystrip, X);
        LQ_STAMP(5);
        if (!carried(mystrip)) {
          store_strip(mystrip, X);
          have = false;
        }
        LQ_STAMP(6);
        if (have) break;  // a strip kept for the next pass: its wavefront takes no further task in this one
      }
    }
    __syncthreads();  // every wavefront is done with this pass's reflectors before their LDS slots are reused
    if (have) {
      double* Vp = Vx + (mystrip - NP * (P + 1)) * kNB * LDV;  // the slot of the panel this strip is in the next pass
#pragma unroll
      for (int T4 = 0; T4 < DT / 4; ++T4)
#pragma unroll
        for (int e = 0; e < 16; ++e) Vp[r * LDV + 64 * T4 + 16 * g + e] = X[16 * T4 + e];
    }
    LQ_STAMP(7);
  }
#ifdef PNODE_TIMING_BUILD
  if (lane == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(v.S + size_t(b) * d * d) + wv * 8;
    for (int i = 0; i < 8; ++i) dbg[i] = tacc[i];
  }
#endif
}

template <int DT, int NP, int DD>
static void launch_lq3(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  static const LqSchedule sched = [] {  // one table per instantiation (thread-safe static initialisation)
    LqSchedule t;
    (void)lq_build_schedule(DT, NP, &t);
    return t;
  }();
  Lq3Args ka{v, args, K, sched};
  hipLaunchKernelGGL((dense_lq3_mfma_kernel<DT, NP, DD>), dim3(v.B), dim3(64 * kLqWaves), 0, s, ka);
}

bool dense_lq3_supported(const DenseView& v) {
  const int DT = v.D / 16, DD = v.d / 16;
  return v.d % 16 == 0 && ((DT == 8 && DD == 2) || (DT == 12 && DD == 2) || (DT == 16 && DD == 4));
}

void dense_launch_lq3(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  switch (v.D / 16) {
    case 8: launch_lq3<8, 4, 2>(v, args, K, s); break;
    case 12: launch_lq3<12, 4, 2>(v, args, K, s); break;
    case 16: launch_lq3<16, 4, 4>(v, args, K, s); break;
    default: break;
  }
}

}  // namespace pnode
