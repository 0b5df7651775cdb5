// Device-side P1/P2 triangle element: geometry, Lagrange bases through
// barycentric coordinates, quadrature tables.  Conventions are those of
// flow_amd/fem/reference.py: local P2 dofs [v0 v1 v2 e0 e1 e2], edge dof e_i on
// the edge opposite vertex i.
#pragma once
#include "common.h"

namespace flow {

template <int DEG>
struct Elem;
template <>
struct Elem<1> {
  static constexpr int NL = 3;
  static constexpr int NQ = 3;   // degree-2 rule
};
template <>
struct Elem<2> {
  static constexpr int NL = 6;
  static constexpr int NQ = 7;   // degree-5 Radon rule (skew convection: 1+2+2)
};

// barycentric quadrature points and weights (weights sum to 1)
__device__ constexpr double kQ3L[3][3] = {
    {2.0 / 3.0, 1.0 / 6.0, 1.0 / 6.0},
    {1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0},
    {1.0 / 6.0, 1.0 / 6.0, 2.0 / 3.0}};
__device__ constexpr double kQ3W[3] = {1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0};

// a = (6 -+ sqrt(15))/21, w = (155 -+ sqrt(15))/1200
#define FLOW_RA 0.10128650732345633880
#define FLOW_RB 0.47014206410511508977
#define FLOW_WA 0.12593918054482715260
#define FLOW_WB 0.13239415278850618074
__device__ constexpr double kQ7L[7][3] = {
    {1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0},
    {1.0 - 2.0 * FLOW_RA, FLOW_RA, FLOW_RA},
    {FLOW_RA, 1.0 - 2.0 * FLOW_RA, FLOW_RA},
    {FLOW_RA, FLOW_RA, 1.0 - 2.0 * FLOW_RA},
    {1.0 - 2.0 * FLOW_RB, FLOW_RB, FLOW_RB},
    {FLOW_RB, 1.0 - 2.0 * FLOW_RB, FLOW_RB},
    {FLOW_RB, FLOW_RB, 1.0 - 2.0 * FLOW_RB}};
__device__ constexpr double kQ7W[7] = {0.225,   FLOW_WA, FLOW_WA, FLOW_WA,
                                       FLOW_WB, FLOW_WB, FLOW_WB};

// 16-point collapsed Gauss-Jacobi rule, exact for degree 7 (SUPG terms:
// conv(2) * grad u(1) * tau(1) * conv(2) * grad v(1)); generated from
// flow_amd/fem/reference.py triangle_rule(7); weights sum to 1
__device__ constexpr double kQ16L[16][3] = {
    {0.87742880933046774, 0.057104196114517725, 0.065466994555014452},
    {0.6317312516411252, 0.057104196114517725, 0.31116455224435702},
    {0.31116455224435702, 0.057104196114517725, 0.6317312516411252},
    {0.065466994555014479, 0.057104196114517725, 0.87742880933046774},
    {0.67294686315050645, 0.2768430136381238, 0.050210123211369778},
    {0.48450832663043331, 0.2768430136381238, 0.23864865973144292},
    {0.23864865973144295, 0.2768430136381238, 0.48450832663043325},
    {0.050210123211369861, 0.2768430136381238, 0.67294686315050634},
    {0.38749748340669415, 0.58359043236891683, 0.028912084224389012},
    {0.2789904634965088, 0.58359043236891683, 0.13741910413457437},
    {0.13741910413457437, 0.58359043236891683, 0.2789904634965088},
    {0.028912084224389012, 0.58359043236891683, 0.38749748340669415},
    {0.1300560792168344, 0.86024013565621948, 0.0097037851269461094},
    {0.093637784437328481, 0.86024013565621948, 0.046122079906452035},
    {0.046122079906452035, 0.86024013565621948, 0.093637784437328481},
    {0.0097037851269461128, 0.86024013565621948, 0.1300560792168344},
};
__device__ constexpr double kQ16W[16] = {
    0.047136736386764778, 0.088370177044723719, 0.088370177044723719, 0.047136736386764778,
    0.070776135796171771, 0.13268843221409932, 0.13268843221409932, 0.070776135796171771,
    0.0451680985647398, 0.084679449043492519, 0.084679449043492519, 0.0451680985647398,
    0.010846451821050504, 0.020334519128957576, 0.020334519128957576, 0.010846451821050504};

template <int DEG>
__device__ __forceinline__ double qpoint(int q, int k) {
  if constexpr (DEG == 1) return kQ3L[q][k];
  else return kQ7L[q][k];
}
template <int DEG>
__device__ __forceinline__ double qweight(int q) {
  if constexpr (DEG == 1) return kQ3W[q];
  else return kQ7W[q];
}

// 2-point Gauss rule on [0,1]
#define FLOW_G2A 0.21132486540518711775
#define FLOW_G2B 0.78867513459481288225

// the two vertices of local facet lf (opposite vertex lf)
__device__ __forceinline__ int facet_v0(int lf) { return lf == 0 ? 1 : 0; }
__device__ __forceinline__ int facet_v1(int lf) { return lf == 2 ? 1 : 2; }

struct Geom {
  double gl[3][2];   // physical gradients of the barycentric coordinates
  double adet;       // |det J| = 2 * area
};

__device__ __forceinline__ Geom load_geom(const double* __restrict__ xy, int nc,
                                          int c) {
  const double x0 = xy[0 * nc + c], x1 = xy[1 * nc + c], x2 = xy[2 * nc + c];
  const double y0 = xy[3 * nc + c], y1 = xy[4 * nc + c], y2 = xy[5 * nc + c];
  const double j00 = x1 - x0, j01 = x2 - x0, j10 = y1 - y0, j11 = y2 - y0;
  const double det = j00 * j11 - j01 * j10;
  const double inv = 1.0 / det;
  Geom g;
  g.gl[1][0] = j11 * inv;
  g.gl[1][1] = -j01 * inv;
  g.gl[2][0] = -j10 * inv;
  g.gl[2][1] = j00 * inv;
  g.gl[0][0] = -g.gl[1][0] - g.gl[2][0];
  g.gl[0][1] = -g.gl[1][1] - g.gl[2][1];
  g.adet = fabs(det);
  return g;
}

// values and d/d(lambda_k) of the basis at barycentric point L
template <int DEG>
__device__ __forceinline__ void basis(const double L[3],
                                      double phi[Elem<DEG>::NL],
                                      double dphi[Elem<DEG>::NL][3]) {
  if constexpr (DEG == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      phi[i] = L[i];
#pragma unroll
      for (int k = 0; k < 3; ++k) dphi[i][k] = (i == k) ? 1.0 : 0.0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      phi[i] = L[i] * (2.0 * L[i] - 1.0);
#pragma unroll
      for (int k = 0; k < 3; ++k) dphi[i][k] = (i == k) ? 4.0 * L[i] - 1.0 : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int j = (e == 0) ? 1 : 0;
      const int k2 = (e == 2) ? 1 : 2;
      phi[3 + e] = 4.0 * L[j] * L[k2];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        dphi[3 + e][k] = (k == j) ? 4.0 * L[k2] : ((k == k2) ? 4.0 * L[j] : 0.0);
    }
  }
}

template <int NL>
__device__ __forceinline__ void phys_grad(const Geom& g, const double dphi[NL][3],
                                          double gphi[NL][2]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int d = 0; d < 2; ++d)
      gphi[i][d] = dphi[i][0] * g.gl[0][d] + dphi[i][1] * g.gl[1][d] +
                   dphi[i][2] * g.gl[2][d];
  }
}

// ---------------------------------------------------------------------------
// The same basis through its STRUCTURE, for the flop-bound cell kernels: the
// derivative of basis function i with respect to lambda_k is zero for most
// (i, k) -- vertex function i only depends on lambda_i, edge function e on the
// two barycentric coordinates of its end points -- so gradients are summed in
// barycentric ("reference") coordinates first and mapped with the 3 x 2 matrix
// gl once per quadrature point, instead of mapping every basis gradient:
//   grad u_a = sum_k gur[a][k] grad(lambda_k),  gur[a][k] = sum_j U_aj dphi_j/dlambda_k
//   sum_d S[d] dphi_i/dx_d = sum_k dphi_i/dlambda_k T[k],  T[k] = sum_d gl[k][d] S[d]
// ---------------------------------------------------------------------------
// The callers pass quadrature points that are compile-time constants (unrolled
// loops over constexpr tables): every coefficient below -- written as ONE
// parenthesised product of L's -- folds to a literal, and each term costs a
// single FMA.  (Written the other way round, U * L * (2 L - 1), the compiler
// must not re-associate and spends two to three fp64 instructions per term:
// the interior path of the P2 Jacobian action went from ~1300 to 1073 fp64
// instructions per cell.)
// gur[k] = sum_j U[j] dphi_j / dlambda_k
template <int DEG>
__device__ __forceinline__ void ref_gradient(const double U[Elem<DEG>::NL],
                                             const double L[3], double gur[3]) {
  if constexpr (DEG == 1) {
    gur[0] = U[0];
    gur[1] = U[1];
    gur[2] = U[2];
  } else {
    // edges: e = 0: (1, 2), e = 1: (0, 2), e = 2: (0, 1)
    gur[0] = U[0] * (4.0 * L[0] - 1.0) + U[4] * (4.0 * L[2]) + U[5] * (4.0 * L[1]);
    gur[1] = U[1] * (4.0 * L[1] - 1.0) + U[3] * (4.0 * L[2]) + U[5] * (4.0 * L[0]);
    gur[2] = U[2] * (4.0 * L[2] - 1.0) + U[3] * (4.0 * L[1]) + U[4] * (4.0 * L[0]);
  }
}

// acc[i] += phi_i * s0 + sum_k (dphi_i / dlambda_k) T[k]
template <int DEG>
__device__ __forceinline__ void test_accumulate(const double L[3], double s0,
                                                const double T[3],
                                                double acc[Elem<DEG>::NL]) {
  if constexpr (DEG == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] += L[i] * s0 + T[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      acc[i] += (L[i] * (2.0 * L[i] - 1.0)) * s0;
      acc[i] += (4.0 * L[i] - 1.0) * T[i];
    }
    acc[3] += (4.0 * L[1] * L[2]) * s0;
    acc[3] += (4.0 * L[2]) * T[1];
    acc[3] += (4.0 * L[1]) * T[2];
    acc[4] += (4.0 * L[0] * L[2]) * s0;
    acc[4] += (4.0 * L[2]) * T[0];
    acc[4] += (4.0 * L[0]) * T[2];
    acc[5] += (4.0 * L[0] * L[1]) * s0;
    acc[5] += (4.0 * L[1]) * T[0];
    acc[5] += (4.0 * L[0]) * T[1];
  }
}

// value at L: sum_j U[j] phi_j
template <int DEG>
__device__ __forceinline__ double eval_at(const double U[Elem<DEG>::NL],
                                          const double L[3]) {
  if constexpr (DEG == 1) {
    return U[0] * L[0] + U[1] * L[1] + U[2] * L[2];
  } else {
    double s = U[0] * (L[0] * (2.0 * L[0] - 1.0));
    s += U[1] * (L[1] * (2.0 * L[1] - 1.0));
    s += U[2] * (L[2] * (2.0 * L[2] - 1.0));
    s += U[3] * (4.0 * L[1] * L[2]);
    s += U[4] * (4.0 * L[0] * L[2]);
    s += U[5] * (4.0 * L[0] * L[1]);
    return s;
  }
}

// div(u) at the three cell vertices (P1 per cell for P2 u, constant for P1 u)
template <int DEG>
__device__ __forceinline__ void div_at_vertices(
    const Geom& g, const double U[2][Elem<DEG>::NL], double d[3]) {
  constexpr int NL = Elem<DEG>::NL;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const double L[3] = {m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0,
                         m == 2 ? 1.0 : 0.0};
    double phi[NL], dphi[NL][3], gphi[NL][2];
    basis<DEG>(L, phi, dphi);
    phys_grad<NL>(g, dphi, gphi);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += U[0][i] * gphi[i][0] + U[1][i] * gphi[i][1];
    d[m] = s;
  }
}

}  // namespace flow
