// Analytic Hessian-vector products of the symmetry-function + MLP models (round 3).
//
// Replaces `tf.hessians(energy, positions)` (reference nn/basic.py:411-421) and the cell derivative of
// the virial behind the elastic constants (nn/constraint/elastic.py:24-44) for the descriptor models;
// the EAM family has its own kernels (ta_eam.hip::eam_hvp). With E = sum_i MLP(G_i(D)) and
// g[p] = dE/dD_p = sum_c w_ic dG_ic/dD_p (w = dE/dG, what the backward kernels compute), the directional
// derivative along D-dot is
//     g-dot[p] = sum_c ( w-dot_ic dG_ic/dD_p  +  w_ic d/d eps (dG_ic/dD_p) ),   w-dot_i = H_mlp,i G-dot_i.
// G-dot comes from the per-channel pair Jacobians of the training path (descriptor_jvp_kernel), w-dot
// from the second-order MLP pass (mlp_grad2_kernel's input adjoint), and both terms of g-dot from ONE
// evaluation of the first-generation backward expression (ta_kernels.hip::backward_kernel: every ordered
// pair of neighbours of a centre, own-side gradient only, no atomics) in dual arithmetic with
// w = (w, w-dot) and D = (D, D-dot). Not a throughput kernel: Hessians are for cells of tens to hundreds
// of atoms; it reads its partners' vectors from global memory and visits every partner of every pair.
#include <hip/hip_runtime.h>

#include "ta_device.h"
#include "ta_dual.h"
#include "ta_math.h"
#include "ta_reduce.h"

namespace ta {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ void cutoff_dual(int kind, Dual u, Dual &f, Dual &dfdu) {
  double fv, d1, d2;
  cutoff_u2(kind, u.v, fv, d1, d2);
  f = make_dual(fv, d1 * u.d);
  dfdu = make_dual(d1, d2 * u.d);
}

// base^(zi - 1), zi >= 1
__device__ __forceinline__ Dual pow_int_m1_dual(Dual base, int zi) {
  Dual r = make_dual(1.0), b = base;
  int e = zi - 1;
  while (e) {
    if (e & 1) r = r * b;
    e >>= 1;
    if (e) b = b * b;
  }
  return r;
}

// D_p = Rj - Ri + S.h as [P][4] (universal.py:463-468), the value part of the dual pair vectors
__global__ __launch_bounds__(kBlock) void pair_vec_kernel(DeviceBatch b, double *Dv) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n_pairs) return;
  const int i = b.pair_i[p], j = b.pair_j[p];
  const double *h = b.cells + 9 * (size_t)b.frame_of_atom[i];
  const double sx = (double)b.pair_shift[3 * p], sy = (double)b.pair_shift[3 * p + 1], sz = (double)b.pair_shift[3 * p + 2];
  const double *ri = b.pos + 3 * (size_t)i, *rj = b.pos + 3 * (size_t)j;
  double *o = Dv + 4 * (size_t)p;
  o[0] = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
  o[1] = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
  o[2] = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
  o[3] = 0.0;
}

// one lane per directed pair a = (i -> j): g[a] = dE/dD_a and its tangent, channels of one chunk
__global__ __launch_bounds__(kBlock) void backward_hvp_kernel(SFParams sf, AngChunk ch, int nb, int ng, int nz,
                                                              DeviceBatch b, const double *__restrict__ Dv,
                                                              const double *__restrict__ Dd,
                                                              const double *__restrict__ wdot, int first,
                                                              int angular, double *gv, double *gd) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n_pairs) return;
  const int i = b.pair_i[p];
  const int sa = b.species[b.pair_j[p]];
  const int nel = sf.n_elements;
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  auto vec = [&](int64_t q, Dual (&x)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = make_dual(Dv[4 * (size_t)q + c], Dd[4 * (size_t)q + c]);
  };
  Dual a[3];
  vec(p, a);
  const Dual ra2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + sf.eps;
  const Dual ra = t_sqrt(ra2);
  const Dual inv_ra = 1.0 / ra;
  Dual g[3] = {make_dual(0.0), make_dual(0.0), make_dual(0.0)};
  const double *wv = b.dEdG + (size_t)i * sf.ndim, *wd = wdot + (size_t)i * sf.ndim;

  if (angular) {
    const Dual ua = ra2 * sf.inv_ac2;
    if (ua.v < 1.0) {
      Dual fa, dfa;
      cutoff_dual(sf.cutoff, ua, fa, dfa);
      Dual Ha[2], Ga[2];
      for (int ib = 0; ib < nb; ++ib) {
        const Dual ea = t_exp(-(ch.beta[ib] * ua));
        Ha[ib] = ea * fa;
        Ga[ib] = ea * (2.0 * sf.inv_ac2) * (dfa - ch.beta[ib] * fa);
      }
      Dual sumAQ = make_dual(0.0), v[3] = {make_dual(0.0), make_dual(0.0), make_dual(0.0)};
      const Dual inv_ra2 = inv_ra * inv_ra;
      for (int sb = 0; sb < nel; ++sb) {
        // dE/dG of the channels of this (a-species, b-species) term, times 2^(1 - zeta), value and tangent
        Dual w[2][2][2];
        const int off = sf.n_radial_dim + angular_term_of(sa, sb, nel) * sf.n_ang;
        for (int ib = 0; ib < nb; ++ib)
          for (int ig = 0; ig < ng; ++ig)
            for (int iz = 0; iz < nz; ++iz) {
              const int c = off + ch.chan[(ib * ng + ig) * nz + iz];
              w[ib][ig][iz] = make_dual(wv[c] * ch.kz[iz], wd[c] * ch.kz[iz]);
            }
        for (int q = seg[sb]; q < seg[sb + 1]; ++q) {
          if (q == p) continue;
          Dual bv[3];
          vec(q, bv);
          const Dual e0 = bv[0] - a[0], e1 = bv[1] - a[1], e2 = bv[2] - a[2];
          const Dual d2 = e0 * e0 + e1 * e1 + e2 * e2 + sf.eps;
          const Dual u = d2 * sf.inv_ac2;
          if (!(u.v < 1.0)) continue;
          const Dual rb2 = bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2] + sf.eps;
          const Dual ub = rb2 * sf.inv_ac2;
          if (!(ub.v < 1.0)) continue;
          const Dual inv_rb = 1.0 / t_sqrt(rb2);
          Dual fb, dfb, fd, dfd;
          cutoff_dual(sf.cutoff, ub, fb, dfb);
          cutoff_dual(sf.cutoff, u, fd, dfd);
          const Dual inv_ab = inv_ra * inv_rb;
          const Dual cth = (ra2 + rb2 - d2) * 0.5 * inv_ab;
          Dual A = make_dual(0.0), Q = make_dual(0.0);
          for (int ib = 0; ib < nb; ++ib) {
            const Dual Hb = t_exp(-(ch.beta[ib] * ub)) * fb;
            const Dual ed = t_exp(-(ch.beta[ib] * u));
            const Dual Hd = ed * fd;
            const Dual Hd2 = (2.0 * sf.inv_ac2) * ed * (dfd - ch.beta[ib] * fd);
            Dual S0 = make_dual(0.0), S1 = make_dual(0.0);
            for (int ig = 0; ig < ng; ++ig) {
              const Dual base = ch.gamma[ig] * cth + 1.0;
              for (int iz = 0; iz < nz; ++iz) {
                const Dual pm1 = pow_int_m1_dual(base, ch.zeta_int[iz]);
                S0 = S0 + w[ib][ig][iz] * (pm1 * base);
                S1 = S1 + w[ib][ig][iz] * (ch.zeta[iz] * ch.gamma[ig]) * pm1;
              }
            }
            A = A + Hb * Hd * (S1 * Ha[ib] * (inv_ab - cth * inv_ra2) + S0 * Ga[ib]);
            Q = Q + Ha[ib] * Hb * (S0 * Hd2 - S1 * inv_ab * Hd);
          }
          sumAQ = sumAQ + A + Q;
#pragma unroll
          for (int c = 0; c < 3; ++c) v[c] = v[c] + Q * bv[c];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = a[c] * sumAQ - v[c];
    }
  }
  if (first) {  // the radial part rides on the first launch (ta_kernels.hip::radial_backward)
    const Dual u = ra2 * sf.inv_rc2;
    if (u.v < 1.0) {
      Dual f, dfdu;
      cutoff_dual(sf.cutoff, u, f, dfdu);
      const Dual dfdr = dfdu * (2.0 * sf.inv_rc2) * ra;
      const int tr = radial_term_of(b.species[i], sa);
      Dual s = make_dual(0.0);
      for (int c = 0; c < sf.n_rad; ++c) {
        const Dual w = make_dual(wv[tr * sf.n_rad + c], wd[tr * sf.n_rad + c]);
        const Dual dr = ra - sf.omega[c];
        const Dual e = t_exp(-(sf.eta[c] * sf.inv_rc2 * (dr * dr)));
        s = s + w * e * (dfdr - (2.0 * sf.eta[c] * sf.inv_rc2) * dr * f);
      }
      s = s * inv_ra;
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = g[c] + s * a[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gv[4 * (size_t)p + c] = g[c].v;
      gd[4 * (size_t)p + c] = g[c].d;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gv[4 * (size_t)p + c] += g[c].v;
      gd[4 * (size_t)p + c] += g[c].d;
    }
  }
}

// F-dot_i = sum_p (g-dot[p] - g-dot[rev p]),  W-dot_i = sum_p (g-dot[p] (x) D_p + g[p] (x) D-dot_p); one
// wavefront per atom
__global__ __launch_bounds__(kBlock) void hvp_gather_kernel(DeviceBatch b, const double *__restrict__ Dv,
                                                            const double *__restrict__ Dd,
                                                            const double *__restrict__ gv,
                                                            const double *__restrict__ gd, double *fdot,
                                                            double *wdot_at) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  double f[3] = {0, 0, 0}, w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = b.pair_start[i] + lane; p < b.pair_start[i + 1]; p += 64) {
    const int r = b.pair_rev[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double gq = gd[4 * (size_t)p + c];
      f[c] += gq - gd[4 * (size_t)r + c];
      if (wdot_at) {
#pragma unroll
        for (int e = 0; e < 3; ++e)
          w[3 * c + e] += gq * Dv[4 * (size_t)p + e] + gv[4 * (size_t)p + c] * Dd[4 * (size_t)p + e];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) f[c] = wave_sum(f[c]);
  if (wdot_at) {
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = wave_sum(w[k]);
  }
  if (lane == 0) {
    for (int c = 0; c < 3; ++c) fdot[3 * (size_t)i + c] = f[c];
    if (wdot_at)
      for (int k = 0; k < 9; ++k) wdot_at[9 * (size_t)i + k] = w[k];
  }
}

inline unsigned nblk(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

void launch_pair_vec(const DeviceBatch &b, double *Dv, hipStream_t s) {
  if (b.n_pairs > 0) hipLaunchKernelGGL(pair_vec_kernel, dim3(nblk(b.n_pairs, kBlock)), dim3(kBlock), 0, s, b, Dv);
}

// one chunk of angular channels (nb x ng x nz, at most 2 each; integer zetas); `angular` = 0: radial part only
void launch_backward_hvp(const SFParams &sf, const AngChunk &ch, int nb, int ng, int nz, bool first, bool angular,
                         const DeviceBatch &b, const double *Dv, const double *Dd, const double *wdot, double *gv,
                         double *gd, hipStream_t s) {
  if (b.n_pairs > 0)
    hipLaunchKernelGGL(backward_hvp_kernel, dim3(nblk(b.n_pairs, kBlock)), dim3(kBlock), 0, s, sf, ch, nb, ng, nz, b,
                       Dv, Dd, wdot, first ? 1 : 0, angular ? 1 : 0, gv, gd);
}

void launch_hvp_gather(const DeviceBatch &b, const double *Dv, const double *Dd, const double *gv, const double *gd,
                       double *fdot, double *wdot_at, hipStream_t s) {
  if (b.n_atoms > 0)
    hipLaunchKernelGGL(hvp_gather_kernel, dim3(nblk(b.n_atoms * 64, kBlock)), dim3(kBlock), 0, s, b, Dv, Dd, gv, gd,
                       fdot, wdot_at);
}

}  // namespace ta
