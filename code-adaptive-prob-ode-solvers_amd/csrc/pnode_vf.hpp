// Built-in vector fields and analytic Jacobians (device code).
// Replaces the JAX callables + autodiff of /root/reference/src/odecheckpts/ivps.py:
//   brusselator :124-156, van_der_pol :159-167 (and experiments/1_van_der_pol/vdp.py:52-54),
//   rigid_body :20-29 (diffeqzoo defaults), _pleiades :66-99, pleiades_1st :44-56, logistic :8-17,
//   first-order Van der Pol: experiments/old/7_node/node.py:23-37, three_body_restricted :32-41 (diffeqzoo's Arenstorf orbit).
// Contract: y points at k*d doubles laid out [order][d] (y[i*d + b] = u^(i)_b), any address space.
//   vf_f(c, y, a, t)          -> component a of f(u, ..., u^(k-1), t)
//   vf_jac(c, y, a, i, b, t)  -> d f_a / d u^(i)_b
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pnode.h"

namespace pnode {

struct VfCtx {
  int id;
  int d;
  int k;              // ode order
  const double* prm;  // per-trajectory parameter block (device memory), may be nullptr
  const double* uf;   // PNODE_VF_USER: host-evaluated f of this trajectory, [d]
  const double* uj;   // PNODE_VF_USER: host-evaluated Jacobian, [d][k*d]
};

__device__ __forceinline__ double pleiades_acc(const double* y, int a, bool want_y) {
  // y[0:7] = x, y[7:14] = y positions; masses m_j = j+1
  const int ia = a;
  double acc = 0.0;
  const double xa = y[ia], ya = y[7 + ia];
  for (int j = 0; j < 7; ++j) {
    if (j == ia) continue;  // reference: nan_to_num of 0/0 -> 0
    const double dx = y[j] - xa, dy = y[7 + j] - ya;
    const double r2 = dx * dx + dy * dy;
    const double r3 = r2 * sqrt(r2);
    acc += double(j + 1) * (want_y ? dy : dx) / r3;
  }
  return acc;
}

__device__ __forceinline__ double pleiades_jac(const double* y, int a, int b) {
  // d acc_a / d pos_b for the 14-vector pos = [x, y]; a, b in [0, 14)
  const bool ay = a >= 7, by = b >= 7;
  const int ia = ay ? a - 7 : a, ib = by ? b - 7 : b;
  const double xa = y[ia], ya = y[7 + ia];
  auto pair_term = [&](int j) {
    // derivative of m_j * (c_j - c_a) / r^3 with respect to coordinate (by ? y_j : x_j)
    const double dx = y[j] - xa, dy = y[7 + j] - ya;
    const double r2 = dx * dx + dy * dy;
    const double r = sqrt(r2);
    const double r3 = r2 * r, r5 = r3 * r2;
    const double num = ay ? dy : dx;  // numerator coordinate difference
    const double dd = by ? dy : dx;   // differentiation coordinate difference
    const double same = (ay == by) ? 1.0 / r3 : 0.0;
    return double(j + 1) * (same - 3.0 * num * dd / r5);
  };
  if (ib != ia) return pair_term(ib);
  double acc = 0.0;
  for (int j = 0; j < 7; ++j)
    if (j != ia) acc -= pair_term(j);
  return acc;
}

__device__ __forceinline__ double vf_f(const VfCtx& c, const double* y, int a, double t) {
  (void)t;
  switch (c.id) {
    case PNODE_VF_BRUSSELATOR: {
      const int N = c.d / 2;
      const double alpha = c.prm ? c.prm[0] : 1.0 / 50.0;
      const double cc = alpha * double(N + 1) * double(N + 1);
      const bool is_u = a < N;
      const int j = is_u ? a : a - N;
      const double u = y[j], v = y[N + j];
      const double* w = is_u ? y : y + N;
      const double pad = is_u ? 1.0 : 3.0;
      const double left = j > 0 ? w[j - 1] : pad;
      const double right = j < N - 1 ? w[j + 1] : pad;
      const double conv = left - 2.0 * w[j] + right;
      return is_u ? 1.0 + u * u * v - 4.0 * u + cc * conv : 3.0 * u - u * u * v + cc * conv;
    }
    case PNODE_VF_VDP_ORDER2: {
      const double mu = c.prm[0];
      const double u = y[0], du = y[1];
      return mu * (du * (1.0 - u * u) - u);
    }
    case PNODE_VF_VDP_ORDER1: {
      const double mu = c.prm[0];
      return a == 0 ? y[1] : mu * (1.0 - y[0] * y[0]) * y[1] - y[0];
    }
    case PNODE_VF_RIGID_BODY: {
      const double pa = c.prm[0], pb = c.prm[1], pc = c.prm[2];
      return a == 0 ? pa * y[1] * y[2] : (a == 1 ? pb * y[0] * y[2] : pc * y[0] * y[1]);
    }
    case PNODE_VF_PLEIADES_ORDER2:
      return pleiades_acc(y, a < 7 ? a : a - 7, a >= 7);
    case PNODE_VF_PLEIADES_ORDER1:
      if (a < 14) return y[14 + a];
      return pleiades_acc(y, (a - 14) < 7 ? a - 14 : a - 21, (a - 14) >= 7);
    case PNODE_VF_LOGISTIC:
      return c.prm[0] * y[a] * (1.0 - c.prm[1] * y[a]);
    case PNODE_VF_LINEAR:
      return c.prm[0] * y[a];
    case PNODE_VF_THREE_BODY: {
      // rotating frame, y = [Y0, Y1, dY0, dY1]: Y'' = Y + 2 J dY - (1 - mu) (Y - P1) / |Y - P1|^3 - mu (Y - P2) / |Y - P2|^3
      const double mu = c.prm[0], mp = 1.0 - mu;
      const double x1 = y[0] + mu, x2 = y[0] - mp, yy = y[1];
      const double r1 = x1 * x1 + yy * yy, r2 = x2 * x2 + yy * yy;
      const double d1 = r1 * sqrt(r1), d2 = r2 * sqrt(r2);
      return a == 0 ? y[0] + 2.0 * y[3] - mp * x1 / d1 - mu * x2 / d2 : yy - 2.0 * y[2] - mp * yy / d1 - mu * yy / d2;
    }
    case PNODE_VF_USER:
      return c.uf[a];
    default:
      return 0.0;
  }
}

__device__ __forceinline__ double vf_jac(const VfCtx& c, const double* y, int a, int i, int b, double t) {
  (void)t;
  switch (c.id) {
    case PNODE_VF_BRUSSELATOR: {
      const int N = c.d / 2;
      const double alpha = c.prm ? c.prm[0] : 1.0 / 50.0;
      const double cc = alpha * double(N + 1) * double(N + 1);
      const bool au = a < N, bu = b < N;
      const int ja = au ? a : a - N, jb = bu ? b : b - N;
      const double u = y[ja], v = y[N + ja];
      double out = 0.0;
      if (au == bu) {
        if (ja == jb) out = -2.0 * cc;
        else if (ja == jb + 1 || ja + 1 == jb) out = cc;
      }
      if (ja == jb) {
        if (au && bu) out += 2.0 * u * v - 4.0;
        else if (au && !bu) out += u * u;
        else if (!au && bu) out += 3.0 - 2.0 * u * v;
        else out += -u * u;
      }
      return out;
    }
    case PNODE_VF_VDP_ORDER2: {
      const double mu = c.prm[0];
      const double u = y[0], du = y[1];
      return i == 0 ? mu * (-2.0 * u * du - 1.0) : mu * (1.0 - u * u);
    }
    case PNODE_VF_VDP_ORDER1: {
      const double mu = c.prm[0];
      if (a == 0) return b == 1 ? 1.0 : 0.0;
      return b == 0 ? -2.0 * mu * y[0] * y[1] - 1.0 : mu * (1.0 - y[0] * y[0]);
    }
    case PNODE_VF_RIGID_BODY: {
      const double pa = c.prm[0], pb = c.prm[1], pc = c.prm[2];
      if (a == b) return 0.0;
      if (a == 0) return pa * (b == 1 ? y[2] : y[1]);
      if (a == 1) return pb * (b == 0 ? y[2] : y[0]);
      return pc * (b == 0 ? y[1] : y[0]);
    }
    case PNODE_VF_PLEIADES_ORDER2:
      return i == 0 ? pleiades_jac(y, a, b) : 0.0;
    case PNODE_VF_PLEIADES_ORDER1:
      if (a < 14) return b == 14 + a ? 1.0 : 0.0;
      return b < 14 ? pleiades_jac(y, a - 14, b) : 0.0;
    case PNODE_VF_LOGISTIC:
      return a == b ? c.prm[0] * (1.0 - 2.0 * c.prm[1] * y[a]) : 0.0;
    case PNODE_VF_LINEAR:
      return a == b ? c.prm[0] : 0.0;
    case PNODE_VF_THREE_BODY: {
      if (i == 1) return a == b ? 0.0 : (a == 0 ? 2.0 : -2.0);  // Coriolis term
      const double mu = c.prm[0], mp = 1.0 - mu;
      const double x1 = y[0] + mu, x2 = y[0] - mp, yy = y[1];
      const double r1 = x1 * x1 + yy * yy, r2 = x2 * x2 + yy * yy;
      const double i13 = 1.0 / (r1 * sqrt(r1)), i23 = 1.0 / (r2 * sqrt(r2));
      const double i15 = i13 / r1, i25 = i23 / r2;
      if (a != b) return 3.0 * yy * (mp * x1 * i15 + mu * x2 * i25);
      const double n1 = a == 0 ? x1 * x1 : yy * yy, n2 = a == 0 ? x2 * x2 : yy * yy;
      return 1.0 - mp * (i13 - 3.0 * n1 * i15) - mu * (i23 - 3.0 * n2 * i25);
    }
    case PNODE_VF_USER:
      return c.uj[size_t(a) * c.k * c.d + i * c.d + b];
    default:
      return 0.0;
  }
}

}  // namespace pnode
