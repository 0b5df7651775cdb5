// Finite-element assembly of the hot path on gfx950 (K1-K7, K13, K14).
//
// Replaces FFC-generated tabulate_tensor + DOLFIN Assembler + DirichletBC for
// the forms of flow/navier_stokes/pressure_correction.py (:135-144, :169-202,
// :317-323, :442-449) and flow/heat.py (:39-88).
//
// Two-phase, atomic-free and therefore bitwise reproducible:
//   phase 1  cell kernels: coalesced reads of the SoA cell->dof / coordinate
//            arrays, local element tensor in registers (the Jacobian kernel
//            stages u(x_q), grad u(x_q) of 64 cells in LDS and gives every
//            wavefront a wave-uniform list of (i,j) entry pairs), coalesced
//            stores to scratch[entry][cell];
//   phase 2  gather kernels: one lane per CSR nonzero / per dof sums its
//            contributions through the host-built contribution map, fixed
//            order, coalesced stores of the CSR values.
// Everything here is HBM-bound integer/fp64 streaming; no MFMA.
#include <cstdlib>

#include "fem_device.h"

namespace flow {

// ---------------------------------------------------------------------------
// phase 2
// ---------------------------------------------------------------------------
// out[p][k] = sum of scratch[p][src[t]], t in [ptr[k], ptr[k+1]), in that order.
// The contribution lists are short (<= ~8 cells around a dof), so the cost is
// the chain ptr -> src -> scratch of dependent loads: four list entries are in
// flight per step, and all planes share one read of the indices.
template <int NP>
__global__ __launch_bounds__(kBlock) void gather_kernel(
    int nout, const int* __restrict__ ptr, const int* __restrict__ src,
    const double* __restrict__ scratch, size_t plane_stride, size_t out_stride,
    double* __restrict__ out, const double* __restrict__ stop,
    const unsigned char* __restrict__ idrow = nullptr, size_t id_stride = 0,
    const double* __restrict__ v = nullptr, size_t v_stride = 0) {
  // idrow != nullptr: identity rows -- where idrow[p * id_stride + k] is set,
  // out = v[p * v_stride + k] instead of the sum (the Dirichlet rows of the
  // matrix-free Jacobian: one launch less than copying them afterwards)
  if (stopped(stop)) return;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nout;
       k += gridDim.x * blockDim.x) {
    const int a = ptr[k];
    const int b = ptr[k + 1];
    double s[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = 0.0;
    for (int t = a; t < b; t += 4) {
      int idx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) idx[j] = t + j < b ? src[t + j] : -1;
      double w[NP][4];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          w[p][j] = idx[j] >= 0 ? scratch[p * plane_stride + idx[j]] : 0.0;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[p] += w[p][j];
    }
    if (idrow) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (idrow[p * id_stride + k]) s[p] = v[p * v_stride + k];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) out[static_cast<size_t>(p) * out_stride + k] = s[p];
  }
}

// rows [r0, r1) of the nout (r1 = 0: all); out_stride = 0: nout
static int gather(int nout, int nplanes, const int* ptr, const int* src,
                  const double* scratch, size_t plane_stride, double* out,
                  hipStream_t st, size_t out_stride = 0, int r0 = 0, int r1 = 0,
                  const double* stop = nullptr,
                  const unsigned char* idrow = nullptr, const double* v = nullptr,
                  size_t v_stride = 0) {
  const size_t os = out_stride ? out_stride : static_cast<size_t>(nout);
  const size_t ids = static_cast<size_t>(nout);     // mask: component stride n
  if (r1 > 0) {
    ptr += r0;
    out += r0;
    if (idrow) idrow += r0;
    if (v) v += r0;
    nout = r1 - r0;
  }
  const dim3 grid(grid_for(nout, kBlock, 1 << 20));
  FLOW_REQUIRE(nplanes == 1 || nplanes == 2, "gather: 1 or 2 planes");
  if (nplanes == 1)
    hipLaunchKernelGGL(gather_kernel<1>, grid, dim3(kBlock), 0, st, nout, ptr,
                       src, scratch, plane_stride, os, out, stop, idrow, ids, v,
                       v_stride);
  else
    hipLaunchKernelGGL(gather_kernel<2>, grid, dim3(kBlock), 0, st, nout, ptr,
                       src, scratch, plane_stride, os, out, stop, idrow, ids, v,
                       v_stride);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
template <int NL>
__device__ __forceinline__ void load_local(const double* __restrict__ u, int n,
                                           const int* __restrict__ cd, int nc,
                                           int c, int /*ncomp = 2*/,
                                           double U[2][NL]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int d = cd[i * nc + c];
    U[0][i] = u[d];
    U[1][i] = u[static_cast<size_t>(n) + d];
  }
}

// ---------------------------------------------------------------------------
// K1 / K3 / lumped mass: constant-coefficient scalar matrices
// ---------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(kBlock) void scalar_matrix_kernel(
    int kind, int nc, const double* __restrict__ xy,
    double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = Elem<2>::NQ;   // 7-point rule: exact for P2 mass (deg 4)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const Geom g = load_geom(xy, nc, c);
  double Ke[NL][NL];
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int j = 0; j < NL; ++j) Ke[i][j] = 0.0;
  if (kind == 2) {
    // 'vertex' quadrature scheme: points = cell vertices, weights |T|/3
    // (flow/heat.py:39-45); only the vertex basis functions are non-zero.
#pragma unroll
    for (int i = 0; i < 3; ++i) Ke[i][i] = g.adet / 6.0;
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const double L[3] = {kQ7L[q][0], kQ7L[q][1], kQ7L[q][2]};
      const double w = 0.5 * kQ7W[q] * g.adet;
      double phi[NL], dphi[NL][3], gphi[NL][2];
      basis<DEG>(L, phi, dphi);
      phys_grad<NL>(g, dphi, gphi);
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int j = 0; j < NL; ++j)
          Ke[i][j] += (kind == 0)
                          ? w * (gphi[i][0] * gphi[j][0] + gphi[i][1] * gphi[j][1])
                          : w * phi[i] * phi[j];
    }
  }
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int j = 0; j < NL; ++j)
      scratch[static_cast<size_t>(i * NL + j) * nc + c] = Ke[i][j];
}

// ---------------------------------------------------------------------------
// K2: pressure right-hand side
// ---------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(kBlock) void pressure_rhs_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy, const int* __restrict__ cdu, int nu,
    const int* __restrict__ cdp, const double* __restrict__ u,
    const double* __restrict__ p0, double alpha_rho_dt, double mu,
    int rotational, double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  const int c = cb + xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (c >= ce) return;
  const Geom g = load_geom(xy, nc, c);
  double U[2][NL];
  load_local<NL>(u, nu, cdu, nc, c, 2, U);
  double d[3];
  div_at_vertices<DEG>(g, U, d);
  const double area = 0.5 * g.adet;
  const double dsum = d[0] + d[1] + d[2];
  double gp[2] = {0.0, 0.0};   // grad p0 [- mu grad div u]
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double coef = p0[cdp[k * nc + c]];
    if (rotational) coef -= mu * d[k];
    gp[0] += coef * g.gl[k][0];
    gp[1] += coef * g.gl[k][1];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double b = -alpha_rho_dt * (area / 12.0) * (dsum + d[i]) +
                     area * (gp[0] * g.gl[i][0] + gp[1] * g.gl[i][1]);
    scratch[static_cast<size_t>(i) * nc + c] = b;
  }
}

// ---------------------------------------------------------------------------
// K4: velocity-correction right-hand side
// ---------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(kBlock) void correction_rhs_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy, const int* __restrict__ cdu, int nu,
    const int* __restrict__ cdp, const double* __restrict__ u,
    const double* __restrict__ p1, const double* __restrict__ p0, double dt_rho,
    double mu, int rotational, double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = Elem<2>::NQ;
  const int c = cb + xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (c >= ce) return;
  const Geom g = load_geom(xy, nc, c);
  double U[2][NL];
  load_local<NL>(u, nu, cdu, nc, c, 2, U);
  double d[3] = {0.0, 0.0, 0.0};
  if (rotational & 1) div_at_vertices<DEG>(g, U, d);
  // (bit 1: the increment's right-hand side, no (u, v) term)
  const double mass = (rotational & 2) ? 0.0 : 1.0;
  double gphi_f[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int dof = cdp[k * nc + c];
    const double coef = p1[dof] - p0[dof] + mu * d[k];
    gphi_f[0] += coef * g.gl[k][0];
    gphi_f[1] += coef * g.gl[k][1];
  }
  double acc[2][NL];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[a][i] = 0.0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double L[3] = {kQ7L[q][0], kQ7L[q][1], kQ7L[q][2]};
    const double w = 0.5 * kQ7W[q] * g.adet;
    double phi[NL], dphi[NL][3];
    basis<DEG>(L, phi, dphi);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double uq = 0.0;
#pragma unroll
      for (int j = 0; j < NL; ++j) uq += U[a][j] * phi[j];
      const double val = w * (mass * uq - dt_rho * gphi_f[a]);
#pragma unroll
      for (int i = 0; i < NL; ++i) acc[a][i] += val * phi[i];
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i)
      scratch[static_cast<size_t>(a * NL + i) * nc + c] = acc[a][i];
}

// ---------------------------------------------------------------------------
// source term (f, v) from per-cell lattice values
// ---------------------------------------------------------------------------
template <int NL>
__device__ __forceinline__ void add_source(const flow_coef& f, int nc, int c,
                                           int dim, double scale,
                                           double acc[][NL]) {
  const size_t nce = f.cell_stride ? static_cast<size_t>(nc) : 1;
  const size_t cc = f.cell_stride ? static_cast<size_t>(c) : 0;
  for (int a = 0; a < dim; ++a) {
    for (int l = 0; l < f.nl; ++l) {
      const double F = scale * f.values[(static_cast<size_t>(a) * f.nl + l) * nce + cc];
#pragma unroll
      for (int i = 0; i < NL; ++i) acc[a][i] += F * f.G[l * NL + i];
    }
  }
}

template <int DEG>
__global__ __launch_bounds__(kBlock) void source_kernel(
    int nc, const double* __restrict__ xy, int dim, flow_coef f,
    double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const Geom g = load_geom(xy, nc, c);
  double acc[2][NL];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[a][i] = 0.0;
  add_source<NL>(f, nc, c, dim, g.adet, acc);
  for (int a = 0; a < dim; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i)
      scratch[static_cast<size_t>(a * NL + i) * nc + c] = acc[a][i];
}

// ---------------------------------------------------------------------------
// K5 + K6: momentum residual
// ---------------------------------------------------------------------------
// acc += scale * R(u; v) of _rhs_weak (pressure_correction.py:135-144) without
// the source term; volume part at one quadrature point
template <int NL>
__device__ __forceinline__ void add_rhs_weak_point(
    double w, double rho, double mu, const double U[2][NL], const double P[3],
    const double L[3], const double phi[NL], const double gphi[NL][2],
    double acc[2][NL]) {
  double uq[2] = {0.0, 0.0};
  double gu[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // gu[a][b] = d_b u_a
#pragma unroll
  for (int j = 0; j < NL; ++j) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      uq[a] += U[a][j] * phi[j];
      gu[a][0] += U[a][j] * gphi[j][0];
      gu[a][1] += U[a][j] * gphi[j][1];
    }
  }
  const double pq = P[0] * L[0] + P[1] * L[1] + P[2] * L[2];
  const double conv[2] = {gu[0][0] * uq[0] + gu[0][1] * uq[1],
                          gu[1][0] * uq[0] + gu[1][1] * uq[1]};
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const double ugp = uq[0] * gphi[i][0] + uq[1] * gphi[i][1];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double r = -0.5 * rho * (conv[a] * phi[i] - ugp * uq[a]);
      r -= mu * ((gu[a][0] + gu[0][a]) * gphi[i][0] +
                 (gu[a][1] + gu[1][a]) * gphi[i][1]);
      r += pq * gphi[i][a];
      acc[a][i] += w * r;
    }
  }
}

// exterior facets: - p0 n.v ds + mu ((grad u)^T n).v ds  (:142-143)
template <int DEG>
__device__ __forceinline__ void add_rhs_weak_facets(
    int mask, double scale, double mu, const Geom& g,
    const double U[2][Elem<DEG>::NL], const double P[3],
    double acc[2][Elem<DEG>::NL]) {
  constexpr int NL = Elem<DEG>::NL;
  for (int lf = 0; lf < 3; ++lf) {
    if (!((mask >> lf) & 1)) continue;
    const int v0 = facet_v0(lf), v1 = facet_v1(lf);
    // outward normal times facet length: -grad(lambda_lf) * |detJ|
    const double nL[2] = {-g.gl[lf][0] * g.adet, -g.gl[lf][1] * g.adet};
    for (int gq = 0; gq < 2; ++gq) {
      const double s = gq == 0 ? FLOW_G2A : FLOW_G2B;
      double L[3] = {0.0, 0.0, 0.0};
      L[v0] = 1.0 - s;
      L[v1] = s;
      double phi[NL], dphi[NL][3], gphi[NL][2];
      basis<DEG>(L, phi, dphi);
      phys_grad<NL>(g, dphi, gphi);
      double gu[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
      for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          gu[a][0] += U[a][j] * gphi[j][0];
          gu[a][1] += U[a][j] * gphi[j][1];
        }
      const double pq = P[0] * L[0] + P[1] * L[1] + P[2] * L[2];
      const double w = 0.5 * scale;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        // ((grad u)^T n)_a = sum_b d_a u_b n_b = sum_b gu[b][a] n_b
        const double t = -pq * nL[a] + mu * (gu[0][a] * nL[0] + gu[1][a] * nL[1]);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc[a][i] += w * t * phi[i];
      }
    }
  }
}

template <int DEG>
__global__ __launch_bounds__(kBlock) void momentum_residual_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy, const int* __restrict__ cdu, int nu,
    const int* __restrict__ cdp, const int* __restrict__ bfmask,
    const double* __restrict__ ui, const double* __restrict__ u0,
    const double* __restrict__ p0, flow_coef f0, flow_coef f1,
    flow_ns_params prm, double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = Elem<DEG>::NQ;
  const int c = cb + xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (c >= ce) return;
  const Geom g = load_geom(xy, nc, c);
  double Ui[2][NL], U0[2][NL], P[3];
  load_local<NL>(ui, nu, cdu, nc, c, 2, Ui);
  load_local<NL>(u0, nu, cdu, nc, c, 2, U0);
#pragma unroll
  for (int k = 0; k < 3; ++k) P[k] = p0[cdp[k * nc + c]];
  const int mask = bfmask[c];
  const double ci = -prm.dt / prm.rho * prm.theta_i;
  const double cexp = -prm.dt / prm.rho * prm.theta_e;
  double acc[2][NL];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[a][i] = 0.0;

#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double L[3] = {qpoint<DEG>(q, 0), qpoint<DEG>(q, 1), qpoint<DEG>(q, 2)};
    const double w = 0.5 * qweight<DEG>(q) * g.adet;
    double phi[NL], dphi[NL][3], gphi[NL][2];
    basis<DEG>(L, phi, dphi);
    phys_grad<NL>(g, dphi, gphi);
    // (ui - u0, v)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double du = 0.0;
#pragma unroll
      for (int j = 0; j < NL; ++j) du += (Ui[a][j] - U0[a][j]) * phi[j];
#pragma unroll
      for (int i = 0; i < NL; ++i) acc[a][i] += w * du * phi[i];
    }
    if (ci != 0.0)
      add_rhs_weak_point<NL>(w * ci, prm.rho, prm.mu, Ui, P, L, phi, gphi, acc);
    if (cexp != 0.0)
      add_rhs_weak_point<NL>(w * cexp, prm.rho, prm.mu, U0, P, L, phi, gphi, acc);
  }
  if (mask) {
    if (ci != 0.0) add_rhs_weak_facets<DEG>(mask, ci, prm.mu, g, Ui, P, acc);
    if (cexp != 0.0) add_rhs_weak_facets<DEG>(mask, cexp, prm.mu, g, U0, P, acc);
  }
  if (ci != 0.0) add_source<NL>(f1, nc, c, 2, ci * g.adet, acc);
  if (cexp != 0.0) add_source<NL>(f0, nc, c, 2, cexp * g.adet, acc);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i)
      scratch[static_cast<size_t>(a * NL + i) * nc + c] = acc[a][i];
}

// ---------------------------------------------------------------------------
// Matrix-free Jacobian action  J(ui) v = dF/dui [v]: the directional derivative
// of momentum_residual_kernel's integrand, one thread per cell.
//   d/dU [ -rho/2 ((u.grad)u . phi - (u.grad phi) . u) ] [v]
//     = -rho/2 ( ((v.grad)u + (u.grad)v) . phi - (v.grad phi) u - (u.grad phi) v )
// The viscous and facet terms are linear in u (add_rhs_weak_facets with p = 0).
// ---------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(kBlock) void momentum_jvp_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy,
    const int* __restrict__ cdu, int nu, int nv,
    const int* __restrict__ bfmask, const double* __restrict__ ui,
    const double* __restrict__ v, flow_ns_params prm,
    const double* __restrict__ prm_dev,
    double* __restrict__ scratch, const double* __restrict__ stop) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = Elem<DEG>::NQ;
  if (stopped(stop)) return;
  const int c = cb + xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (c >= ce) return;
  const Geom g = load_geom(xy, nc, c);
  double U[2][NL], V[2][NL];
  load_local<NL>(ui, nu, cdu, nc, c, 2, U);
  load_local<NL>(v, nv, cdu, nc, c, 2, V);
  const int mask = bfmask[c];
  // (prm_dev: the three numbers below in device memory, written by
  // jvp_params_kernel -- a replayed graph cannot carry the step size by value)
  const double ci = prm_dev ? load_scalar(prm_dev)
                            : -prm.dt / prm.rho * prm.theta_i;
  const double hr = prm_dev ? load_scalar(prm_dev + 1) : 0.5 * prm.rho;
  const double mu = prm_dev ? load_scalar(prm_dev + 2) : prm.mu;
  double acc[2][NL];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[a][i] = 0.0;

#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double L[3] = {qpoint<DEG>(q, 0), qpoint<DEG>(q, 1), qpoint<DEG>(q, 2)};
    const double w = 0.5 * qweight<DEG>(q) * g.adet;
    // fields and their gradients at the point, gradients via barycentric sums
    // (fem_device.h: ref_gradient / test_accumulate)
    double uq[2], vq[2], gu[2][2], gv[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      uq[a] = eval_at<DEG>(U[a], L);
      vq[a] = eval_at<DEG>(V[a], L);
      double gur[3], gvr[3];
      ref_gradient<DEG>(U[a], L, gur);
      ref_gradient<DEG>(V[a], L, gvr);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        gu[a][d] = gur[0] * g.gl[0][d] + gur[1] * g.gl[1][d] + gur[2] * g.gl[2][d];
        gv[a][d] = gvr[0] * g.gl[0][d] + gvr[1] * g.gl[1][d] + gvr[2] * g.gl[2][d];
      }
    }
    // acc[a][i] += phi_i S0[a] + sum_d d_d(phi_i) S1[a][d]  with
    //   S0 = w (v - ci rho/2 dconv),  dconv = (grad v) u + (grad u) v,
    //   S1[a][d] = w ci (rho/2 (v_d u_a + u_d v_a) - mu (d_d v_a + d_a v_d))
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const double dconv = gv[a][0] * uq[0] + gv[a][1] * uq[1] +
                           gu[a][0] * vq[0] + gu[a][1] * vq[1];
      const double s0 = w * (vq[a] - ci * hr * dconv);
      double S1[2];
#pragma unroll
      for (int d = 0; d < 2; ++d)
        S1[d] = w * ci * (hr * (vq[d] * uq[a] + uq[d] * vq[a]) -
                          mu * (gv[a][d] + gv[d][a]));
      double T[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) T[k] = g.gl[k][0] * S1[0] + g.gl[k][1] * S1[1];
      test_accumulate<DEG>(L, s0, T, acc[a]);
    }
  }
  if (mask && ci != 0.0) {
    const double P0[3] = {0.0, 0.0, 0.0};
    add_rhs_weak_facets<DEG>(mask, ci, mu, g, V, P0, acc);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i)
      scratch[static_cast<size_t>(a * NL + i) * nc + c] = acc[a][i];
}

// The same action with TWO LANES PER CELL (P2): lane 2k + a of a wavefront
// works on velocity component a of cell k of its tile.  Each lane holds ONE
// component of the linearisation state and of the direction (12 doubles
// instead of 24) and accumulates the six test functions of ITS component (6
// doubles instead of 12); per quadrature point the two lanes swap the three
// numbers the other needs -- u_b, v_b and the cross derivative d_a v_b --
// through DPP (__shfl_xor 1).  No work is done twice except the geometry.  What
// it is for: the one-lane kernel runs at 148 VGPRs = 3 waves per SIMD, its
// counters read 31 % of the wave cycles waiting for the dof gathers
// (profiles/counters_r04.md); with the state split over two lanes the kernel
// fits 5 waves per SIMD (96 VGPRs).  Measured: slower (123 against 108 us),
// see momentum_jvp_apply -- kept behind FLOW_AMD_JVP_PAIR=1 for the record and
// the counters.
template <int DEG>
__global__ __launch_bounds__(kBlock) void momentum_jvp_pair_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy,
    const int* __restrict__ cdu, int nu, int nv,
    const int* __restrict__ bfmask, const double* __restrict__ ui,
    const double* __restrict__ v, flow_ns_params prm,
    const double* __restrict__ prm_dev,
    double* __restrict__ scratch, const double* __restrict__ stop) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = Elem<DEG>::NQ;
  if (stopped(stop)) return;
  const int t = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const int c = cb + (t >> 1);
  const int a = t & 1;                 // this lane's velocity component
  if (c >= ce) return;                 // (both lanes of a pair leave together)
  const Geom g = load_geom(xy, nc, c);
  double U[NL], V[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int d = cdu[i * nc + c];
    U[i] = ui[static_cast<size_t>(a) * nu + d];
    V[i] = v[static_cast<size_t>(a) * nv + d];
  }
  const int mask = bfmask[c];
  const double ci = prm_dev ? load_scalar(prm_dev)
                            : -prm.dt / prm.rho * prm.theta_i;
  const double hr = prm_dev ? load_scalar(prm_dev + 1) : 0.5 * prm.rho;
  const double mu = prm_dev ? load_scalar(prm_dev + 2) : prm.mu;
  double acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.0;

#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double L[3] = {qpoint<DEG>(q, 0), qpoint<DEG>(q, 1), qpoint<DEG>(q, 2)};
    const double w = 0.5 * qweight<DEG>(q) * g.adet;
    const double uq = eval_at<DEG>(U, L);
    const double vq = eval_at<DEG>(V, L);
    double gur[3], gvr[3], gu[2], gv[2];
    ref_gradient<DEG>(U, L, gur);
    ref_gradient<DEG>(V, L, gvr);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      gu[d] = gur[0] * g.gl[0][d] + gur[1] * g.gl[1][d] + gur[2] * g.gl[2][d];
      gv[d] = gvr[0] * g.gl[0][d] + gvr[1] * g.gl[1][d] + gvr[2] * g.gl[2][d];
    }
    // the partner's component b = 1 - a: u_b, v_b, d_a v_b
    const double uo = __shfl_xor(uq, 1, 64);
    const double vo = __shfl_xor(vq, 1, 64);
    const double cross = __shfl_xor(a == 0 ? gv[1] : gv[0], 1, 64);
    const double u0 = a == 0 ? uq : uo, u1 = a == 0 ? uo : uq;
    const double v0 = a == 0 ? vq : vo, v1 = a == 0 ? vo : vq;
    // as momentum_jvp_kernel, for component a:
    //   S0 = w (v_a - ci rho/2 dconv),  dconv = (grad v_a).u + (grad u_a).v
    //   S1[d] = w ci (rho/2 (v_d u_a + u_d v_a) - mu (d_d v_a + d_a v_d))
    const double dconv = gv[0] * u0 + gv[1] * u1 + gu[0] * v0 + gu[1] * v1;
    const double s0 = w * (vq - ci * hr * dconv);
    // d_a v_d: d = a is this lane's own gv[a], d = b the partner's
    const double t0 = a == 0 ? gv[0] : cross;     // d_a v_0
    const double t1 = a == 0 ? cross : gv[1];     // d_a v_1
    double S1[2];
    S1[0] = w * ci * (hr * (v0 * uq + u0 * vq) - mu * (gv[0] + t0));
    S1[1] = w * ci * (hr * (v1 * uq + u1 * vq) - mu * (gv[1] + t1));
    double T[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) T[k] = g.gl[k][0] * S1[0] + g.gl[k][1] * S1[1];
    test_accumulate<DEG>(L, s0, T, acc);
  }
  if (mask && ci != 0.0) {
    // exterior facets, p = 0:  mu ((grad v)^T n)_a = mu sum_b d_a v_b n_b
    // (add_rhs_weak_facets; both lanes of the pair are here: same cell)
    for (int lf = 0; lf < 3; ++lf) {
      if (!((mask >> lf) & 1)) continue;
      const int f0 = facet_v0(lf), f1 = facet_v1(lf);
      const double nL[2] = {-g.gl[lf][0] * g.adet, -g.gl[lf][1] * g.adet};
      for (int gq = 0; gq < 2; ++gq) {
        const double sg = gq == 0 ? FLOW_G2A : FLOW_G2B;
        double L[3] = {0.0, 0.0, 0.0};
        L[f0] = 1.0 - sg;
        L[f1] = sg;
        double gvr[3], gv[2];
        ref_gradient<DEG>(V, L, gvr);
#pragma unroll
        for (int d = 0; d < 2; ++d)
          gv[d] = gvr[0] * g.gl[0][d] + gvr[1] * g.gl[1][d] + gvr[2] * g.gl[2][d];
        const double cross = __shfl_xor(a == 0 ? gv[1] : gv[0], 1, 64);
        // d_a v_0 n_0 + d_a v_1 n_1
        const double tn = a == 0 ? gv[0] * nL[0] + cross * nL[1]
                                 : cross * nL[0] + gv[1] * nL[1];
        const double wt = 0.5 * ci * mu * tn;
        double phi[NL], dphi[NL][3];
        basis<DEG>(L, phi, dphi);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc[i] += wt * phi[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NL; ++i)
    scratch[static_cast<size_t>(a * NL + i) * nc + c] = acc[i];
}

// out[d] = v[d] on the Dirichlet dofs (identity rows) whose row lies in
// [r0, r1); d = a*n + row, the vectors have component strides vs / os
__global__ void bc_copy_kernel(int nbc, const int* __restrict__ dofs, int n,
                               int r0, int r1, const double* __restrict__ v,
                               int vs, double* __restrict__ out, int os,
                               const double* __restrict__ stop) {
  if (stopped(stop)) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nbc) return;
  const int d = dofs[k];
  const int a = d / n;
  const int row = d - a * n;
  if (row >= r0 && row < r1)
    out[static_cast<size_t>(a) * os + row] = v[static_cast<size_t>(a) * vs + row];
}

// ---------------------------------------------------------------------------
// K5 + K6: Jacobian  J = dF/dui
// ---------------------------------------------------------------------------
// basis tables at the quadrature points (wave-uniform index -> scalar loads)
template <int DEG>
struct BasisTab {
  double phi[Elem<DEG>::NQ][Elem<DEG>::NL];
  double dphi[Elem<DEG>::NQ][Elem<DEG>::NL][3];
};

template <int DEG>
__global__ __launch_bounds__(kBlock) void momentum_jacobian_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy, const int* __restrict__ cdu, int nu,
    const int* __restrict__ bfmask, const double* __restrict__ ui,
    flow_ns_params prm, double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = Elem<DEG>::NQ;
  constexpr int NP = NL * NL;
  // LDS staging of the per-cell fields at the quadrature points, [item][lane]
  __shared__ double s_uq[NQ][2][64];
  __shared__ double s_gu[NQ][4][64];
  __shared__ BasisTab<DEG> tab;

  const int lane = threadIdx.x & 63;
  const int grp = threadIdx.x >> 6;          // wavefront id: wave-uniform
  const int c = cb + blockIdx.x * 64 + lane;
  const bool active = c < ce;
  const int cc = active ? c : ce - 1;

  // basis tables: one (q, i) entry per thread
  for (int t = threadIdx.x; t < NQ * NL; t += kBlock) {
    const int q = t / NL, i = t % NL;
    const double L[3] = {qpoint<DEG>(q, 0), qpoint<DEG>(q, 1), qpoint<DEG>(q, 2)};
    double phi[NL], dphi[NL][3];
    basis<DEG>(L, phi, dphi);
    tab.phi[q][i] = phi[i];
    tab.dphi[q][i][0] = dphi[i][0];
    tab.dphi[q][i][1] = dphi[i][1];
    tab.dphi[q][i][2] = dphi[i][2];
  }

  const Geom g = load_geom(xy, nc, cc);
  const int mask = bfmask[cc];
  {
    double U[2][NL];
    load_local<NL>(ui, nu, cdu, nc, cc, 2, U);
    // wave `grp` stages quadrature points grp, grp+4, ...
    for (int q = grp; q < NQ; q += 4) {
      const double L[3] = {qpoint<DEG>(q, 0), qpoint<DEG>(q, 1),
                           qpoint<DEG>(q, 2)};
      double phi[NL], dphi[NL][3], gphi[NL][2];
      basis<DEG>(L, phi, dphi);
      phys_grad<NL>(g, dphi, gphi);
      double uq[2] = {0.0, 0.0}, gu[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        uq[0] += U[0][j] * phi[j];
        uq[1] += U[1][j] * phi[j];
        gu[0] += U[0][j] * gphi[j][0];
        gu[1] += U[0][j] * gphi[j][1];
        gu[2] += U[1][j] * gphi[j][0];
        gu[3] += U[1][j] * gphi[j][1];
      }
      s_uq[q][0][lane] = uq[0];
      s_uq[q][1][lane] = uq[1];
#pragma unroll
      for (int k = 0; k < 4; ++k) s_gu[q][k][lane] = gu[k];
    }
  }
  __syncthreads();

  const double c1 = -prm.dt / prm.rho * prm.theta_i;
  const double hr = 0.5 * prm.rho;
  const double mu = prm.mu;

  for (int pr = grp; pr < NP; pr += 4) {
    const int i = pr / NL, j = pr % NL;       // row local dof i, column j
    double B[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const double w = 0.5 * qweight<DEG>(q) * g.adet;
      const double pi = tab.phi[q][i], pj = tab.phi[q][j];
      double gi[2], gj[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        gi[d] = tab.dphi[q][i][0] * g.gl[0][d] + tab.dphi[q][i][1] * g.gl[1][d] +
                tab.dphi[q][i][2] * g.gl[2][d];
        gj[d] = tab.dphi[q][j][0] * g.gl[0][d] + tab.dphi[q][j][1] * g.gl[1][d] +
                tab.dphi[q][j][2] * g.gl[2][d];
      }
      const double u0 = s_uq[q][0][lane], u1 = s_uq[q][1][lane];
      const double gu[2][2] = {{s_gu[q][0][lane], s_gu[q][1][lane]},
                               {s_gu[q][2][lane], s_gu[q][3][lane]}};
      const double uq[2] = {u0, u1};
      const double ugj = u0 * gj[0] + u1 * gj[1];
      const double ugi = u0 * gi[0] + u1 * gi[1];
      const double kij = gi[0] * gj[0] + gi[1] * gj[1];
      // diagonal (a == c) part
      const double diag =
          pi * pj + c1 * (-hr * (ugj * pi - ugi * pj) - mu * kij);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          double v = c1 * (-hr * (pj * pi * gu[a][cb] - pj * gi[cb] * uq[a]) -
                           mu * gj[a] * gi[cb]);
          if (a == cb) v += diag;
          B[a][cb] += w * v;
        }
    }
    if (mask) {
      // + dt/rho-scaled  mu int_Gamma d_a phi_j n_c phi_i ds
      for (int lf = 0; lf < 3; ++lf) {
        if (!((mask >> lf) & 1)) continue;
        const int v0 = facet_v0(lf), v1 = facet_v1(lf);
        const double nL[2] = {-g.gl[lf][0] * g.adet, -g.gl[lf][1] * g.adet};
        for (int gq = 0; gq < 2; ++gq) {
          const double s = gq == 0 ? FLOW_G2A : FLOW_G2B;
          double L[3] = {0.0, 0.0, 0.0};
          L[v0] = 1.0 - s;
          L[v1] = s;
          double phi[NL], dphi[NL][3], gphi[NL][2];
          basis<DEG>(L, phi, dphi);
          phys_grad<NL>(g, dphi, gphi);
          double pi = 0.0, gj0 = 0.0, gj1 = 0.0;
#pragma unroll
          for (int t = 0; t < NL; ++t) {
            if (t == i) pi = phi[t];
            if (t == j) {
              gj0 = gphi[t][0];
              gj1 = gphi[t][1];
            }
          }
          const double w = 0.5 * c1 * mu * pi;
          B[0][0] += w * gj0 * nL[0];
          B[0][1] += w * gj0 * nL[1];
          B[1][0] += w * gj1 * nL[0];
          B[1][1] += w * gj1 * nL[1];
        }
      }
    }
    if (active) {
      // cell-major scratch [cell][ij][plane]: the 2x2 block of one (cell, ij)
      // is one aligned 32-byte sector, and the gather below finds all entries
      // of the few cells around a row close together
      double2* out = reinterpret_cast<double2*>(
          scratch + (static_cast<size_t>(c) * NP + pr) * 4);
      out[0] = make_double2(B[0][0], B[0][1]);
      out[1] = make_double2(B[1][0], B[1][1]);
    }
  }
}

// phase 2 for the Jacobian: per CSR nonzero, sum the 32-byte 2x2 blocks of its
// contributions (src[t] = ij*nc + cell, the shared contribution map) and write
// the four value planes
__global__ void gather_blocks_kernel(int k0, int nnz, int nc, int np,
                                     const int* __restrict__ ptr,
                                     const int* __restrict__ src,
                                     const double* __restrict__ scratch,
                                     size_t out_stride,
                                     double* __restrict__ out) {
  for (int k = k0 + blockIdx.x * blockDim.x + threadIdx.x; k < nnz;
       k += gridDim.x * blockDim.x) {
    const int a = ptr[k];
    const int b = ptr[k + 1];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int t = a; t < b; ++t) {
      const int s = src[t];
      const int ij = s / nc;
      const int cell = s - ij * nc;
      const double2* blk = reinterpret_cast<const double2*>(
          scratch + (static_cast<size_t>(cell) * np + ij) * 4);
      const double2 v0 = blk[0], v1 = blk[1];
      s0 += v0.x;
      s1 += v0.y;
      s2 += v1.x;
      s3 += v1.y;
    }
    out[k] = s0;
    out[out_stride + k] = s1;
    out[2 * out_stride + k] = s2;
    out[3 * out_stride + k] = s3;
  }
}

// ---------------------------------------------------------------------------
// K7: Dirichlet conditions
// ---------------------------------------------------------------------------
__global__ void bc_identity_rows_kernel(int kind, int n, size_t nnz,
                                        const int* __restrict__ rowptr,
                                        const int* __restrict__ diag_idx,
                                        int nbc, const int* __restrict__ dofs,
                                        double* __restrict__ vals) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nbc;
       t += gridDim.x * blockDim.x) {
    const int d = dofs[t];
    const int a = d / n;
    const int i = d - a * n;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    if (kind == 2) {
      double* p0 = vals + static_cast<size_t>(2 * a) * nnz;
      double* p1 = vals + static_cast<size_t>(2 * a + 1) * nnz;
      for (int k = k0; k < k1; ++k) {
        p0[k] = 0.0;
        p1[k] = 0.0;
      }
      vals[static_cast<size_t>(3 * a) * nnz + diag_idx[i]] = 1.0;
    } else {
      double* p = vals + static_cast<size_t>(a) * nnz;
      for (int k = k0; k < k1; ++k) p[k] = 0.0;
      p[diag_idx[i]] = 1.0;
    }
  }
}

__global__ void bc_residual_kernel(int nbc, const int* __restrict__ dofs,
                                   const double* __restrict__ g,
                                   const double* __restrict__ u,
                                   double* __restrict__ F) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nbc;
       t += gridDim.x * blockDim.x) {
    const int d = dofs[t];
    F[d] = u ? u[d] - g[t] : g[t];
  }
}

__global__ void bc_symmetric_kernel(int n, const int* __restrict__ rowptr,
                                    const int* __restrict__ cols,
                                    const double* __restrict__ vin,
                                    const unsigned char* __restrict__ isbc,
                                    double* __restrict__ vout) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n;
       r += gridDim.x * blockDim.x) {
    const bool rb = isbc[r] != 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int c = cols[k];
      vout[k] = (rb || isbc[c]) ? (r == c ? 1.0 : 0.0) : vin[k];
    }
  }
}

// ---------------------------------------------------------------------------
// K13 / K14: heat operator and SUPG tau
// ---------------------------------------------------------------------------
// SupgStab::eval (flow/stabilization.py:50-143)
__device__ __forceinline__ double supg_tau(const double px[3], const double py[3],
                                           double area, double bx, double by,
                                           double eps, int p, int* status) {
  const double nb = sqrt(bx * bx + by * by);
  if (nb < 1.0e-10) return 0.0;
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i + 1; j < 3; ++j) {
      const double e0 = px[i] - px[j];
      const double e1 = py[i] - py[j];
      sum += fabs(e1 * bx - e0 * by);
    }
  const double h = 4.0 * nb * area / sum;
  const double Pe = 0.5 * nb * h / (p * eps);
  const double xi = Pe > 1.0e-5
                        ? (1.0 / tanh(Pe) - 1.0 / Pe) / Pe
                        : 1.0 / 3.0 - Pe * Pe / 45.0 +
                              2.0 / 945.0 * Pe * Pe * Pe * Pe;
  const double tau = h * h / 4.0 / eps / p * xi;
  if (tau > 1.0e3) *status = 1;
  return tau;
}

template <int DEGQ, int DEGW, bool SUPG>
__global__ __launch_bounds__(kBlock) void heat_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy,
    const int* __restrict__ cdw, int nw,
    const double* __restrict__ conv, double kappa, double rho_cp,
    double* __restrict__ scratch, double* __restrict__ tau_out,
    int* __restrict__ status) {
  constexpr int NL = Elem<DEGQ>::NL;
  constexpr int NW = Elem<DEGW>::NL;
  // plain operator: conv(2) grad u(1) v(2) = degree 5 -> 7-point rule;
  // SUPG terms reach degree 7 -> 16-point rule (FFC picks the exact degree)
  constexpr int NQ = SUPG ? 16 : 7;
  const int c = cb + blockIdx.x * blockDim.x + threadIdx.x;   // [cb, ce): the
  if (c >= ce) return;                                        // rank's cells
  const Geom g = load_geom(xy, nc, c);
  double Cv[2][NW];
  load_local<NW>(conv, nw, cdw, nc, c, 2, Cv);
  double tau_v[3] = {0.0, 0.0, 0.0};
  double lap[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) lap[i] = 0.0;
  if constexpr (SUPG) {
    const double px[3] = {xy[0 * nc + c], xy[1 * nc + c], xy[2 * nc + c]};
    const double py[3] = {xy[3 * nc + c], xy[4 * nc + c], xy[5 * nc + c]};
    // tau is Expression(degree=1): evaluated at the cell vertices
    // (stabilization.py:147); conv at vertex m = its vertex dof value
#pragma unroll
    for (int m = 0; m < 3; ++m)
      tau_v[m] = supg_tau(px, py, 0.5 * g.adet, Cv[0][m], Cv[1][m], kappa, DEGQ,
                          status);
    if (tau_out) {
#pragma unroll
      for (int m = 0; m < 3; ++m) tau_out[static_cast<size_t>(m) * nc + c] = tau_v[m];
    }
    if constexpr (DEGQ == 2) {
      // Laplacians of the P2 basis (constant per cell)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        lap[i] = 4.0 * (g.gl[i][0] * g.gl[i][0] + g.gl[i][1] * g.gl[i][1]);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int j = facet_v0(e), k = facet_v1(e);
        lap[3 + e] = 8.0 * (g.gl[j][0] * g.gl[k][0] + g.gl[j][1] * g.gl[k][1]);
      }
    }
  }
  const size_t plane = static_cast<size_t>(NL * NL) * nc;
  // rows one at a time to bound register use: Ke[i][:] and Me[i][:]
  for (int i = 0; i < NL; ++i) {
    double Ka[NL], Ma[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) Ka[j] = Ma[j] = 0.0;
    for (int q = 0; q < NQ; ++q) {
      const double L[3] = {SUPG ? kQ16L[q][0] : kQ7L[q][0],
                           SUPG ? kQ16L[q][1] : kQ7L[q][1],
                           SUPG ? kQ16L[q][2] : kQ7L[q][2]};
      const double w = 0.5 * (SUPG ? kQ16W[q] : kQ7W[q]) * g.adet;
      double phi[NL], dphi[NL][3], gphi[NL][2];
      basis<DEGQ>(L, phi, dphi);
      phys_grad<NL>(g, dphi, gphi);
      double wphi[NW], wd[NW][3];
      basis<DEGW>(L, wphi, wd);
      double b[2] = {0.0, 0.0};
#pragma unroll
      for (int t = 0; t < NW; ++t) {
        b[0] += Cv[0][t] * wphi[t];
        b[1] += Cv[1][t] * wphi[t];
      }
      double pi = 0.0, gi0 = 0.0, gi1 = 0.0;
#pragma unroll
      for (int t = 0; t < NL; ++t)
        if (t == i) {
          pi = phi[t];
          gi0 = gphi[t][0];
          gi1 = gphi[t][1];
        }
      const double tq = tau_v[0] * L[0] + tau_v[1] * L[1] + tau_v[2] * L[2];
      const double cgi = b[0] * gi0 + b[1] * gi1;   // conv . grad v_i
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const double cgj = b[0] * gphi[j][0] + b[1] * gphi[j][1];
        double k = -kappa / rho_cp * (gphi[j][0] * gi0 + gphi[j][1] * gi1) -
                   cgj * pi;
        if constexpr (SUPG) {
          k += (kappa / rho_cp * lap[j] - cgj) * tq * cgi;
          Ma[j] += w * phi[j] * tq * cgi;
        }
        Ka[j] += w * k;
      }
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const size_t idx = static_cast<size_t>(i * NL + j) * nc + c;
      scratch[idx] = Ka[j];
      if constexpr (SUPG) scratch[plane + idx] = Ma[j];
    }
  }
}

// SUPG part of the heat load vector (flow/heat.py:79-86, the `source / rho_cp`
// term of R2):  b_i = int (source / rho_cp) tau (conv . grad v_i).  The source
// is a P_k interpolant per cell, k = 0, 1, 2 (f.nl = 1, 3, 6 lattice values in
// the local dof order); degree k + 1 + 2 + 1 <= 6: the 16-point rule is exact.
template <int DEGQ, int DEGW>
__global__ __launch_bounds__(kBlock) void heat_supg_source_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy,
    const int* __restrict__ cdw, int nw,
    const double* __restrict__ conv, double kappa, double rho_cp, flow_coef f,
    double* __restrict__ scratch, int* __restrict__ status) {
  constexpr int NL = Elem<DEGQ>::NL;
  constexpr int NW = Elem<DEGW>::NL;
  const int c = cb + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ce) return;
  const Geom g = load_geom(xy, nc, c);
  double Cv[2][NW];
  load_local<NW>(conv, nw, cdw, nc, c, 2, Cv);
  const double px[3] = {xy[0 * nc + c], xy[1 * nc + c], xy[2 * nc + c]};
  const double py[3] = {xy[3 * nc + c], xy[4 * nc + c], xy[5 * nc + c]};
  double tau_v[3];
#pragma unroll
  for (int m = 0; m < 3; ++m)
    tau_v[m] = supg_tau(px, py, 0.5 * g.adet, Cv[0][m], Cv[1][m], kappa, DEGQ,
                        status);
  const size_t nce = f.cell_stride ? static_cast<size_t>(nc) : 1;
  const size_t cc = f.cell_stride ? static_cast<size_t>(c) : 0;
  double Sv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int l = 0; l < f.nl; ++l) Sv[l] = f.values[static_cast<size_t>(l) * nce + cc];
  double acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.0;
  for (int q = 0; q < 16; ++q) {
    const double L[3] = {kQ16L[q][0], kQ16L[q][1], kQ16L[q][2]};
    const double w = 0.5 * kQ16W[q] * g.adet;
    double phi[NL], dphi[NL][3], gphi[NL][2];
    basis<DEGQ>(L, phi, dphi);
    phys_grad<NL>(g, dphi, gphi);
    double wphi[NW], wd[NW][3];
    basis<DEGW>(L, wphi, wd);
    double b[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      b[0] += Cv[0][t] * wphi[t];
      b[1] += Cv[1][t] * wphi[t];
    }
    double sq;
    if (f.nl == 1) {
      sq = Sv[0];
    } else if (f.nl == 3) {
      sq = Sv[0] * L[0] + Sv[1] * L[1] + Sv[2] * L[2];
    } else {
      double p2[6], d2[6][3];
      basis<2>(L, p2, d2);
      sq = 0.0;
#pragma unroll
      for (int l = 0; l < 6; ++l) sq += Sv[l] * p2[l];
    }
    const double tq = tau_v[0] * L[0] + tau_v[1] * L[1] + tau_v[2] * L[2];
    const double fac = w * sq / rho_cp * tq;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      acc[i] += fac * (b[0] * gphi[i][0] + b[1] * gphi[i][1]);
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) scratch[static_cast<size_t>(i) * nc + c] = acc[i];
}

// ---------------------------------------------------------------------------
// Stokes (flow/stokes.py:40-42): adjoint of the divergence coupling,
// out_(a,i) = - int p d_a phi_i   (the term  - p * div(v) * dx)
// ---------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(kBlock) void div_adjoint_kernel(
    int nc, const double* __restrict__ xy, const int* __restrict__ cdp,
    const double* __restrict__ p, double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = 7;   // p (1) * grad phi (1): degree 2, the 7-pt rule is exact
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const Geom g = load_geom(xy, nc, c);
  double P[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) P[k] = p[cdp[k * nc + c]];
  double acc[2][NL];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[a][i] = 0.0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double L[3] = {kQ7L[q][0], kQ7L[q][1], kQ7L[q][2]};
    const double w = 0.5 * kQ7W[q] * g.adet;
    double phi[NL], dphi[NL][3], gphi[NL][2];
    basis<DEG>(L, phi, dphi);
    phys_grad<NL>(g, dphi, gphi);
    const double pq = P[0] * L[0] + P[1] * L[1] + P[2] * L[2];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      acc[0][i] -= w * pq * gphi[i][0];
      acc[1][i] -= w * pq * gphi[i][1];
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NL; ++i)
      scratch[static_cast<size_t>(a * NL + i) * nc + c] = acc[a][i];
}

// ---------------------------------------------------------------------------
// K16: load vector of |u| for the callers' step-size control
// ---------------------------------------------------------------------------
// b_i = int m(u) phi_i, m = sqrt(ux^2+uy^2) (mode 0) or |ux|+|uy| (mode 1)
template <int DEG>
__global__ __launch_bounds__(kBlock) void magnitude_kernel(
    int nc, int cb, int ce, const double* __restrict__ xy, const int* __restrict__ cdu, int nu,
    const double* __restrict__ u, int mode, double* __restrict__ scratch) {
  constexpr int NL = Elem<DEG>::NL;
  constexpr int NQ = 7;
  const int c = cb + xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (c >= ce) return;
  const Geom g = load_geom(xy, nc, c);
  double U[2][NL];
  load_local<NL>(u, nu, cdu, nc, c, 2, U);
  double acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double L[3] = {kQ7L[q][0], kQ7L[q][1], kQ7L[q][2]};
    const double w = 0.5 * kQ7W[q] * g.adet;
    double phi[NL], dphi[NL][3];
    basis<DEG>(L, phi, dphi);
    double ux = 0.0, uy = 0.0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      ux += U[0][j] * phi[j];
      uy += U[1][j] * phi[j];
    }
    const double m = mode == 0 ? sqrt(ux * ux + uy * uy) : fabs(ux) + fabs(uy);
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[i] += w * m * phi[i];
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) scratch[static_cast<size_t>(i) * nc + c] = acc[i];
}

static int check_mesh_space(const flow_mesh* mesh, const flow_space* V) {
  FLOW_REQUIRE(mesh && mesh->nc > 0 && mesh->xy, "mesh");
  FLOW_REQUIRE(mesh->c1 == 0 ||
                   (0 <= mesh->c0 && mesh->c0 < mesh->c1 && mesh->c1 <= mesh->nc),
               "mesh cell range");
  FLOW_REQUIRE(V && (V->deg == 1 || V->deg == 2) && V->n > 0 && V->cell_dofs,
               "space");
  FLOW_REQUIRE(V->r1 == 0 || (0 <= V->r0 && V->r0 < V->r1 && V->r1 <= V->n),
               "space row range");
  return FLOW_OK;
}

// the cells the kernels of this call run over
struct CellRange {
  int cb, ce;
  explicit CellRange(const flow_mesh* m)
      : cb(m->c1 > 0 ? m->c0 : 0), ce(m->c1 > 0 ? m->c1 : m->nc) {}
  int count() const { return ce - cb; }
};

}  // namespace flow

using namespace flow;

#define FLOW_DISPATCH_DEG(deg, KERNEL, grid, st, ...)                          \
  do {                                                                         \
    if ((deg) == 1)                                                            \
      hipLaunchKernelGGL(KERNEL<1>, grid, dim3(kBlock), 0, st, __VA_ARGS__);   \
    else                                                                       \
      hipLaunchKernelGGL(KERNEL<2>, grid, dim3(kBlock), 0, st, __VA_ARGS__);   \
    FLOW_CHECK_LAUNCH();                                                       \
  } while (0)

static inline dim3 cell_grid(int nc) { return dim3((nc + kBlock - 1) / kBlock); }


extern "C" int flow_assemble_scalar_matrix(int kind, const flow_mesh* mesh,
                                           const flow_space* V, double* scratch,
                                           double* vals, void* stream) {
  int rc = check_mesh_space(mesh, V);
  if (rc) return rc;
  FLOW_REQUIRE(kind >= 0 && kind <= 2, "matrix kind");
  FLOW_REQUIRE(scratch && vals && V->cptr && V->csrc && V->nnz > 0, "matrix maps");
  hipStream_t st = as_stream(stream);
  FLOW_DISPATCH_DEG(V->deg, scalar_matrix_kernel, cell_grid(mesh->nc), st, kind,
                    mesh->nc, mesh->xy, scratch);
  return gather(V->nnz, 1, V->cptr, V->csrc, scratch, 0, vals, st);
}

extern "C" int flow_assemble_pressure_rhs(const flow_mesh* mesh,
                                          const flow_space* W,
                                          const flow_space* P, const double* u,
                                          const double* p0, double alpha_rho_dt,
                                          double mu, int rotational,
                                          double* scratch, double* b,
                                          void* stream) {
  int rc = check_mesh_space(mesh, W);
  if (rc) return rc;
  if ((rc = check_mesh_space(mesh, P))) return rc;
  FLOW_REQUIRE(P->deg == 1 && P->vptr && P->vsrc, "pressure space must be P1");
  FLOW_REQUIRE(u && p0 && scratch && b, "pointers");
  hipStream_t st = as_stream(stream);
  const CellRange cr(mesh);
  FLOW_DISPATCH_DEG(W->deg, pressure_rhs_kernel, cell_grid(cr.count()), st,
                    mesh->nc, cr.cb, cr.ce, mesh->xy, W->cell_dofs, W->n,
                    P->cell_dofs, u, p0, alpha_rho_dt, mu, rotational, scratch);
  return gather(P->n, 1, P->vptr, P->vsrc, scratch, 0, b, st, 0, P->r0, P->r1);
}

extern "C" int flow_assemble_correction_rhs(
    const flow_mesh* mesh, const flow_space* W, const flow_space* P,
    const double* u, const double* p1, const double* p0, double dt_rho,
    double mu, int rotational, double* scratch, double* b, void* stream) {
  int rc = check_mesh_space(mesh, W);
  if (rc) return rc;
  if ((rc = check_mesh_space(mesh, P))) return rc;
  FLOW_REQUIRE(P->deg == 1 && W->vptr && W->vsrc, "spaces");
  FLOW_REQUIRE(u && p1 && p0 && scratch && b, "pointers");
  hipStream_t st = as_stream(stream);
  const CellRange cr(mesh);
  FLOW_DISPATCH_DEG(W->deg, correction_rhs_kernel, cell_grid(cr.count()), st,
                    mesh->nc, cr.cb, cr.ce, mesh->xy, W->cell_dofs, W->n,
                    P->cell_dofs, u, p1, p0, dt_rho, mu, rotational, scratch);
  const int nl = W->deg == 1 ? 3 : 6;
  return gather(W->n, 2, W->vptr, W->vsrc, scratch,
                static_cast<size_t>(nl) * mesh->nc, b, st, 0, W->r0, W->r1);
}

static int check_coef(const flow_coef* f) {
  FLOW_REQUIRE(f && f->nl >= 1 && f->nl <= 21 && f->values && f->G, "coefficient");
  FLOW_REQUIRE(f->cell_stride == 0 || f->cell_stride == 1, "coefficient stride");
  return FLOW_OK;
}

extern "C" int flow_assemble_source(const flow_mesh* mesh, const flow_space* V,
                                    int dim, const flow_coef* f, double* scratch,
                                    double* b, void* stream) {
  int rc = check_mesh_space(mesh, V);
  if (rc) return rc;
  if ((rc = check_coef(f))) return rc;
  FLOW_REQUIRE(dim == 1 || dim == 2, "dim");
  FLOW_REQUIRE(scratch && b && V->vptr && V->vsrc, "pointers");
  hipStream_t st = as_stream(stream);
  FLOW_DISPATCH_DEG(V->deg, source_kernel, cell_grid(mesh->nc), st, mesh->nc,
                    mesh->xy, dim, *f, scratch);
  const int nl = V->deg == 1 ? 3 : 6;
  return gather(V->n, dim, V->vptr, V->vsrc, scratch,
                static_cast<size_t>(nl) * mesh->nc, b, st);
}

extern "C" int flow_assemble_momentum(
    const flow_mesh* mesh, const flow_space* W, const flow_space* P,
    const int* bfmask, const double* ui, const double* u0, const double* p0,
    const flow_coef* f0, const flow_coef* f1, const flow_ns_params* prm,
    double* scratch, double* F, double* Jvals, size_t j_plane_stride,
    void* stream) {
  int rc = check_mesh_space(mesh, W);
  if (rc) return rc;
  if ((rc = check_mesh_space(mesh, P))) return rc;
  if ((rc = check_coef(f0))) return rc;
  if ((rc = check_coef(f1))) return rc;
  FLOW_REQUIRE(P->deg == 1, "pressure space must be P1");
  FLOW_REQUIRE(bfmask && ui && u0 && p0 && prm && scratch, "pointers");
  FLOW_REQUIRE(prm->dt > 0.0 && prm->rho > 0.0 && prm->mu > 0.0, "parameters");
  hipStream_t st = as_stream(stream);
  const int nl = W->deg == 1 ? 3 : 6;
  const CellRange cr(mesh);
  if (F) {
    FLOW_REQUIRE(W->vptr && W->vsrc, "vector map");
    FLOW_DISPATCH_DEG(W->deg, momentum_residual_kernel, cell_grid(cr.count()),
                      st, mesh->nc, cr.cb, cr.ce, mesh->xy, W->cell_dofs, W->n,
                      P->cell_dofs, bfmask, ui, u0, p0, *f0, *f1, *prm, scratch);
    if ((rc = gather(W->n, 2, W->vptr, W->vsrc, scratch,
                     static_cast<size_t>(nl) * mesh->nc, F, st, 0, W->r0,
                     W->r1)))
      return rc;
  }
  if (Jvals) {
    FLOW_REQUIRE(W->cptr && W->csrc && W->nnz > 0, "matrix map");
    FLOW_REQUIRE(j_plane_stride >= (size_t)W->nnz, "Jacobian plane stride");
    FLOW_DISPATCH_DEG(W->deg, momentum_jacobian_kernel,
                      dim3((cr.count() + 63) / 64), st, mesh->nc, cr.cb, cr.ce,
                      mesh->xy, W->cell_dofs, W->n, bfmask, ui, *prm, scratch);
    // the nonzeros of the rows [r0, r1) (a host-known range: the caller
    // passes the row pointer values through the space's row range)
    int k0 = 0, k1 = W->nnz;
    if (W->r1 > 0) {
      FLOW_REQUIRE(W->nnz0 >= 0 && W->nnz0 < W->nnz1 && W->nnz1 <= W->nnz,
                   "space nonzero range");
      k0 = W->nnz0;
      k1 = W->nnz1;
    }
    hipLaunchKernelGGL(gather_blocks_kernel,
                       dim3(grid_for(k1 - k0, kBlock, 1 << 20)), dim3(kBlock), 0,
                       st, k0, k1, mesh->nc, nl * nl, W->cptr, W->csrc, scratch,
                       j_plane_stride, Jvals);
    FLOW_CHECK_LAUNCH();
  }
  return FLOW_OK;
}

int flow::momentum_jvp_check(const flow_momentum_jvp* J) {
  FLOW_REQUIRE(J != nullptr, "matrix-free Jacobian is NULL");
  int rc = check_mesh_space(J->mesh, J->W);
  if (rc) return rc;
  FLOW_REQUIRE(J->W->vptr && J->W->vsrc, "vector map");
  FLOW_REQUIRE(J->bfmask && J->ui && J->scratch, "matrix-free Jacobian pointers");
  FLOW_REQUIRE(J->prm.dt > 0.0 && J->prm.rho > 0.0 && J->prm.mu > 0.0,
               "parameters");
  FLOW_REQUIRE(J->nbc >= 0 && (J->nbc == 0 || J->bc_dofs), "Dirichlet dofs");
  return FLOW_OK;
}

// v_stride / out_stride: component strides of v and out (0: W->n); both
// pointers are indexed by GLOBAL row (callers with compact vectors shift them)
__global__ void jvp_params_kernel(flow_ns_params prm, double* __restrict__ dst) {
  if (threadIdx.x == 0) {
    store_scalar(dst, -prm.dt / prm.rho * prm.theta_i);
    store_scalar(dst + 1, 0.5 * prm.rho);
    store_scalar(dst + 2, prm.mu);
  }
}

int flow::momentum_jvp_params(const flow_momentum_jvp* J, double* dst,
                              hipStream_t st) {
  hipLaunchKernelGGL(jvp_params_kernel, dim3(1), dim3(64), 0, st, J->prm, dst);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

int flow::momentum_jvp_apply(const flow_momentum_jvp* J, const double* v,
                             double* out, hipStream_t st, int v_stride,
                             int out_stride, const double* stop,
                             const double* prm_dev) {
  const flow_mesh* mesh = J->mesh;
  const flow_space* W = J->W;
  const int nl = W->deg == 1 ? 3 : 6;
  const int vs = v_stride ? v_stride : W->n;
  const int os = out_stride ? out_stride : W->n;
  const CellRange cr(mesh);
  int rc;
  // P2, FLOW_AMD_JVP_PAIR=1 in the environment: two lanes per cell
  // (momentum_jvp_pair_kernel).  Measured on the 9.87 M-DoF workload and NOT
  // the default: 96 VGPRs / 5 waves per SIMD instead of 148 / 3, bit-identical
  // results -- and 123 us per launch instead of 108: the kernel is bound by
  // fp64 issue, and the swaps, selects and the second copy of the geometry
  // and index loads add instructions where the extra waves only hide latency
  // (profiles/NOTES.md section 5, round 5)
  static const bool pair = [] {
    const char* e = getenv("FLOW_AMD_JVP_PAIR");
    return e && e[0] == '1';
  }();
  if (W->deg == 2 && pair) {
    hipLaunchKernelGGL(momentum_jvp_pair_kernel<2>, cell_grid(2 * cr.count()),
                       dim3(kBlock), 0, st, mesh->nc, cr.cb, cr.ce, mesh->xy,
                       W->cell_dofs, W->n, vs, J->bfmask, J->ui, v, J->prm,
                       prm_dev, J->scratch, stop);
    FLOW_CHECK_LAUNCH();
  } else {
    FLOW_DISPATCH_DEG(W->deg, momentum_jvp_kernel, cell_grid(cr.count()), st,
                      mesh->nc, cr.cb, cr.ce, mesh->xy, W->cell_dofs, W->n, vs,
                      J->bfmask, J->ui, v, J->prm, prm_dev, J->scratch, stop);
  }
  // (Dirichlet rows are identity rows: with the byte mask the gather writes
  // them itself; without it a copy kernel follows)
  if ((rc = gather(W->n, 2, W->vptr, W->vsrc, J->scratch,
                   static_cast<size_t>(nl) * mesh->nc, out, st, os, W->r0,
                   W->r1, stop, J->nbc > 0 ? J->bc_mask : nullptr, v, vs)))
    return rc;
  if (J->nbc > 0 && !J->bc_mask) {
    const int r0 = W->r1 > 0 ? W->r0 : 0, r1 = W->r1 > 0 ? W->r1 : W->n;
    hipLaunchKernelGGL(bc_copy_kernel, dim3(grid_for(J->nbc)), dim3(kBlock), 0,
                       st, J->nbc, J->bc_dofs, W->n, r0, r1, v, vs, out, os, stop);
    FLOW_CHECK_LAUNCH();
  }
  return FLOW_OK;
}

extern "C" int flow_momentum_jvp_apply(const flow_momentum_jvp* J,
                                       const double* v, double* out,
                                       void* stream) {
  int rc = momentum_jvp_check(J);
  if (rc) return rc;
  FLOW_REQUIRE(v && out && v != out, "jvp vectors");
  return momentum_jvp_apply(J, v, out, as_stream(stream), 0, 0, nullptr);
}

extern "C" int flow_assemble_magnitude(const flow_mesh* mesh, const flow_space* W,
                                       int mode, const double* u,
                                       double* scratch, double* b,
                                       void* stream) {
  int rc = check_mesh_space(mesh, W);
  if (rc) return rc;
  FLOW_REQUIRE(mode == 0 || mode == 1, "mode");
  FLOW_REQUIRE(u && scratch && b && W->vptr && W->vsrc, "pointers");
  hipStream_t st = as_stream(stream);
  const CellRange cr(mesh);
  FLOW_DISPATCH_DEG(W->deg, magnitude_kernel, cell_grid(cr.count()), st, mesh->nc,
                    cr.cb, cr.ce, mesh->xy, W->cell_dofs, W->n, u, mode, scratch);
  return gather(W->n, 1, W->vptr, W->vsrc, scratch, 0, b, st, 0, W->r0, W->r1);
}

extern "C" int flow_assemble_div_adjoint(const flow_mesh* mesh,
                                         const flow_space* W, const flow_space* P,
                                         const double* p, double* scratch,
                                         double* out, void* stream) {
  int rc = check_mesh_space(mesh, W);
  if (rc) return rc;
  if ((rc = check_mesh_space(mesh, P))) return rc;
  FLOW_REQUIRE(P->deg == 1, "pressure space must be P1");
  FLOW_REQUIRE(p && scratch && out && W->vptr && W->vsrc, "pointers");
  hipStream_t st = as_stream(stream);
  FLOW_DISPATCH_DEG(W->deg, div_adjoint_kernel, cell_grid(mesh->nc), st, mesh->nc,
                    mesh->xy, P->cell_dofs, p, scratch);
  const int nl = W->deg == 1 ? 3 : 6;
  return gather(W->n, 2, W->vptr, W->vsrc, scratch,
                static_cast<size_t>(nl) * mesh->nc, out, st);
}

extern "C" int flow_bc_identity_rows(const flow_operator* A, double* vals_planes,
                                     const int* diag_idx, int nbc,
                                     const int* dofs, void* stream) {
  FLOW_REQUIRE(A && A->n > 0 && A->nnz > 0 && A->rowptr, "operator");
  FLOW_REQUIRE(A->kind >= 0 && A->kind <= 2, "operator kind");
  FLOW_REQUIRE(vals_planes && diag_idx && nbc >= 0, "pointers");
  if (nbc == 0) return FLOW_OK;
  FLOW_REQUIRE(dofs != nullptr, "dofs");
  FLOW_REQUIRE(vals_planes == A->vals[0], "vals_planes must be A->vals[0]");
  // planes are equally spaced (stride >= nnz, even: 16-B aligned planes)
  const long long stride =
      A->kind == 0 ? A->nnz : static_cast<long long>(A->vals[1] - A->vals[0]);
  FLOW_REQUIRE(stride >= A->nnz, "plane stride");
  if (A->kind == 2)
    FLOW_REQUIRE(A->vals[2] - A->vals[0] == 2 * stride &&
                     A->vals[3] - A->vals[0] == 3 * stride,
                 "equally spaced planes");
  hipLaunchKernelGGL(bc_identity_rows_kernel, dim3(grid_for(nbc)), dim3(kBlock),
                     0, as_stream(stream), A->kind, A->n,
                     static_cast<size_t>(stride), A->rowptr, diag_idx, nbc, dofs,
                     vals_planes);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_bc_residual(int nbc, const int* dofs, const double* g,
                                const double* u, double* F, void* stream) {
  FLOW_REQUIRE(nbc >= 0, "nbc");
  if (nbc == 0) return FLOW_OK;
  FLOW_REQUIRE(dofs && g && u && F, "pointers");
  hipLaunchKernelGGL(bc_residual_kernel, dim3(grid_for(nbc)), dim3(kBlock), 0,
                     as_stream(stream), nbc, dofs, g, u, F);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_bc_set_values(int nbc, const int* dofs, const double* g,
                                  double* x, void* stream) {
  FLOW_REQUIRE(nbc >= 0, "nbc");
  if (nbc == 0) return FLOW_OK;
  FLOW_REQUIRE(dofs && g && x, "pointers");
  hipLaunchKernelGGL(bc_residual_kernel, dim3(grid_for(nbc)), dim3(kBlock), 0,
                     as_stream(stream), nbc, dofs, g,
                     static_cast<const double*>(nullptr), x);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_bc_symmetric_matrix(int n, const int* rowptr,
                                        const int* cols, const double* vals_in,
                                        const unsigned char* isbc,
                                        double* vals_out, void* stream) {
  FLOW_REQUIRE(n > 0 && rowptr && cols && vals_in && isbc && vals_out, "pointers");
  hipLaunchKernelGGL(bc_symmetric_kernel, dim3(grid_for(n)), dim3(kBlock), 0,
                     as_stream(stream), n, rowptr, cols, vals_in, isbc, vals_out);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_assemble_heat(const flow_mesh* mesh, const flow_space* Q,
                                  const flow_space* W, const double* conv,
                                  double kappa, double rho_cp, int supg,
                                  double* scratch, double* Avals,
                                  double* Msupg_vals, double* tau_out,
                                  int* status_dev, void* stream) {
  int rc = check_mesh_space(mesh, Q);
  if (rc) return rc;
  if ((rc = check_mesh_space(mesh, W))) return rc;
  FLOW_REQUIRE(conv && scratch && Avals && status_dev, "pointers");
  FLOW_REQUIRE(Q->cptr && Q->csrc && Q->nnz > 0, "matrix map");
  FLOW_REQUIRE(!supg || Msupg_vals, "Msupg_vals required with supg");
  FLOW_REQUIRE(kappa > 0.0 && rho_cp > 0.0, "coefficients");
  hipStream_t st = as_stream(stream);
  // K15: the rank's cells (mesh->c0 / c1) and the nonzeros of its rows
  // (Q->nnz0 / nnz1), everything by default
  const CellRange cr(mesh);
  const dim3 grid = cell_grid(cr.count());
#define FLOW_HEAT(DQ, DW)                                                      \
  do {                                                                         \
    if (supg)                                                                  \
      hipLaunchKernelGGL((heat_kernel<DQ, DW, true>), grid, dim3(kBlock), 0, st, \
                         mesh->nc, cr.cb, cr.ce, mesh->xy, W->cell_dofs, W->n, \
                         conv, kappa, rho_cp, scratch, tau_out, status_dev);   \
    else                                                                       \
      hipLaunchKernelGGL((heat_kernel<DQ, DW, false>), grid, dim3(kBlock), 0,  \
                         st, mesh->nc, cr.cb, cr.ce, mesh->xy, W->cell_dofs,   \
                         W->n, conv, kappa, rho_cp, scratch, tau_out,          \
                         status_dev);                                          \
  } while (0)
  if (Q->deg == 1 && W->deg == 1) FLOW_HEAT(1, 1);
  else if (Q->deg == 1 && W->deg == 2) FLOW_HEAT(1, 2);
  else if (Q->deg == 2 && W->deg == 1) FLOW_HEAT(2, 1);
  else FLOW_HEAT(2, 2);
#undef FLOW_HEAT
  FLOW_CHECK_LAUNCH();
  const int nl = Q->deg == 1 ? 3 : 6;
  const size_t plane = static_cast<size_t>(nl) * nl * mesh->nc;
  int k0 = 0, k1 = 0;          // (0, 0): every nonzero
  if (Q->r1 > 0) {
    FLOW_REQUIRE(Q->nnz0 >= 0 && Q->nnz0 < Q->nnz1 && Q->nnz1 <= Q->nnz,
                 "space nonzero range");
    k0 = Q->nnz0;
    k1 = Q->nnz1;
  }
  if ((rc = gather(Q->nnz, 1, Q->cptr, Q->csrc, scratch, 0, Avals, st, 0, k0, k1)))
    return rc;
  if (supg)
    return gather(Q->nnz, 1, Q->cptr, Q->csrc, scratch + plane, 0, Msupg_vals, st,
                  0, k0, k1);
  return FLOW_OK;
}

extern "C" int flow_assemble_heat_supg_source(
    const flow_mesh* mesh, const flow_space* Q, const flow_space* W,
    const double* conv, double kappa, double rho_cp, const flow_coef* source,
    double* scratch, double* b, int* status_dev, void* stream) {
  int rc = check_mesh_space(mesh, Q);
  if (rc) return rc;
  if ((rc = check_mesh_space(mesh, W))) return rc;
  FLOW_REQUIRE(source && source->values &&
                   (source->nl == 1 || source->nl == 3 || source->nl == 6),
               "SUPG source: a P0, P1 or P2 interpolant per cell");
  FLOW_REQUIRE(source->cell_stride == 0 || source->cell_stride == 1,
               "coefficient stride");
  FLOW_REQUIRE(conv && scratch && b && status_dev && Q->vptr && Q->vsrc,
               "pointers");
  FLOW_REQUIRE(kappa > 0.0 && rho_cp > 0.0, "coefficients");
  hipStream_t st = as_stream(stream);
  const CellRange cr(mesh);
  const dim3 grid = cell_grid(cr.count());
#define FLOW_HEAT(DQ, DW)                                                        \
  hipLaunchKernelGGL((heat_supg_source_kernel<DQ, DW>), grid, dim3(kBlock), 0,   \
                     st, mesh->nc, cr.cb, cr.ce, mesh->xy, W->cell_dofs, W->n,   \
                     conv, kappa, rho_cp, *source, scratch, status_dev)
  if (Q->deg == 1 && W->deg == 1) FLOW_HEAT(1, 1);
  else if (Q->deg == 1 && W->deg == 2) FLOW_HEAT(1, 2);
  else if (Q->deg == 2 && W->deg == 1) FLOW_HEAT(2, 1);
  else FLOW_HEAT(2, 2);
#undef FLOW_HEAT
  FLOW_CHECK_LAUNCH();
  return gather(Q->n, 1, Q->vptr, Q->vsrc, scratch, 0, b, st, 0, Q->r0, Q->r1);
}
