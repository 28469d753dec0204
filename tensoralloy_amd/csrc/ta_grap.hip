// GRAP (GenericRadialAtomicPotential) descriptor kernels on the fp64 matrix cores.
//
// Replaces reference tensoralloy/nn/atomic/grap.py:
//   radial filters (sf / morse / density / pexp)   :124-219 (+ potentials/generic.py:15-30, :87-99, :120-176)
//   apply_legacy_pairwise_descriptor_functions      :378-468
//   apply_model, moment / multiplicity tensors      :470-529, :592-683
// and the tf.gradients through them (nn/basic.py:277-331).
//
// For centre i and neighbour-species block b the descriptor needs
//     P[k][d] = sum_j H_k(r_ij) M_d(u_ij),   H_k = v_k(r) fc(r),  M_d = ux^nx uy^ny uz^nz
// for K radial filters and the 1 / 4 / 10 / 20 packed moment components: a (K x n) . (n x nd)
// matrix product, GEMM-shaped work, so it runs as v_mfma_f64_16x16x4_f64 with both operands
// computed in registers (lane l supplies H_{l&15}(r_j) and M_{l&15}(u_j) for neighbour
// j = 4 step + (l >> 4)): no LDS, no cross-lane reduction for the sum over neighbours.
// Q[k][m] = sum_d T[d][m] P[k][d]^2 is a 16-lane DPP row sum of the accumulator tile.
//
// Backward: A[k][d] = dE/dP[k][d] (grap_dp_kernel), then per 16-pair tile
//     a[j][d] = sum_k H_k(r_j) A[k][d],  b[j][d] = sum_k H'_k(r_j) A[k][d]     (MFMA again)
//     dE/dD_j = sum_d b_d M_d u + a_d (dM_d/du - deg_d M_d u) / r               (row sum over d)
// Forces / virial / energy then use force_gather and frame_reduce like every other model.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "ta_device.h"
#include "ta_math.h"

namespace ta {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxFilters = 32;
constexpr int kMaxComp = 20;
typedef double f64x4 __attribute__((ext_vector_type(4)));

enum { GRAP_SF = 0, GRAP_MORSE = 1, GRAP_DENSITY = 2, GRAP_PEXP = 3 };

struct GrapParams {
  int nel, K, max_moment, nd;  // nd = 1, 4, 10, 20 packed components
  int nf;                      // features per (block, filter)
  int legacy, algo, cutoff;
  int col_of_m[4];             // feature column of moment m, -1 = not emitted
  double rcut, inv_rc2;
  double T[kMaxComp][4];       // multiplicity tensor (grap.py:470-492)
  const double *fp;            // device [K][4] filter constants
};

// exponents (nx, ny, nz) packed 2 bits each, in the reference's component order (grap.py:501-511)
__device__ __forceinline__ int comp_code(int d) {
  constexpr unsigned char tab[kMaxComp] = {
      0x00, 0x01, 0x04, 0x10, 0x02, 0x05, 0x11, 0x08, 0x14, 0x20,
      0x03, 0x06, 0x12, 0x09, 0x15, 0x21, 0x0c, 0x18, 0x24, 0x30};
  // select chain instead of a memory table: d is lane-dependent
  int c = 0;
#pragma unroll
  for (int k = 0; k < kMaxComp; ++k) c = (d == k) ? tab[k] : c;
  return c;
}

__device__ __forceinline__ double pow3(double u, int n) {
  const double u2 = u * u;
  return n == 0 ? 1.0 : (n == 1 ? u : (n == 2 ? u2 : u2 * u));
}
// d/du u^n
__device__ __forceinline__ double dpow3(double u, int n) {
  return n == 0 ? 0.0 : (n == 1 ? 1.0 : (n == 2 ? 2.0 * u : 3.0 * u * u));
}

// v(r) and dv/dr of one radial filter (without the cutoff)
__device__ __forceinline__ void filter_fn(int algo, double p0, double p1, double p2, double r,
                                          double inv_rc2, double &v, double &dv) {
  switch (algo) {
    case GRAP_SF: {  // exp(-eta (r - omega)^2 / rc^2); p0 = eta, p1 = omega
      const double t = r - p1;
      v = ta_exp(-p0 * t * t * inv_rc2);
      dv = v * (-2.0 * p0 * t * inv_rc2);
      break;
    }
    case GRAP_MORSE: {  // D [exp(-2 g (r - r0)) - 2 exp(-g (r - r0))]; p0 = D, p1 = gamma, p2 = r0
      const double e1 = ta_exp(-p1 * (r - p2));
      const double e2 = e1 * e1;
      v = p0 * (e2 - 2.0 * e1);
      dv = p0 * p1 * (2.0 * e1 - 2.0 * e2);
      break;
    }
    case GRAP_DENSITY: {  // A exp(-beta (r / re - 1)); p0 = A, p1 = beta, p2 = re
      v = p0 * ta_exp(-p1 * (r / p2 - 1.0));
      dv = v * (-p1 / p2);
      break;
    }
    default: {  // exp(-(r / rl)^pl); p0 = rl, p1 = pl
      const double x = r / p0;
      double xp;
      if (p1 == 1.0) xp = x;
      else if (p1 == 2.0) xp = x * x;
      else if (p1 == 3.0) xp = x * x * x;
      else xp = pow(x, p1);
      v = ta_exp(-xp);
      dv = v * (-p1 * xp / r);
      break;
    }
  }
}

// H = v fc, dH/dr
__device__ __forceinline__ void filter_cut(const GrapParams &g, double p0, double p1, double p2,
                                           double r, double r2, double &H, double &dH) {
  double v, dv, f, dfdu;
  filter_fn(g.algo, p0, p1, p2, r, g.inv_rc2, v, dv);
  const double u = r2 * g.inv_rc2;
  if (u < 1.0) {
    cutoff_u(g.cutoff, u, f, dfdu);
  } else {
    f = 0.0;
    dfdu = 0.0;
  }
  H = v * f;
  dH = dv * f + v * dfdu * 2.0 * r * g.inv_rc2;
}

// block index of neighbour species sb for centre species sA: [AA, AB (B != A sorted)]
__device__ __forceinline__ int term_block(int sA, int sb) { return sb == sA ? 0 : (sb < sA ? sb + 1 : sb); }

// One wavefront per atom: P (kept for the backward pass) and the features.
__global__ __launch_bounds__(kBlock) void grap_forward_kernel(GrapParams g, DeviceBatch b, double *Pbuf,
                                                              int ndim) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int nel = g.nel, K = g.K, nd = g.nd;
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  const int c0 = comp_code(m16), c1 = comp_code(16 + m16 < kMaxComp ? 16 + m16 : 0);
  const bool d0_ok = m16 < nd, d1_ok = 16 + m16 < nd;
  double T0[4], T1[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    T0[m] = d0_ok ? g.T[m16][m] : 0.0;
    T1[m] = d1_ok ? g.T[16 + m16 < kMaxComp ? 16 + m16 : 0][m] : 0.0;
  }
  for (int sb = 0; sb < nel; ++sb) {
    const int lo = seg[sb], hi = seg[sb + 1];
    const int tb = term_block(sA, sb);
    for (int kt = 0; kt * 16 < K; ++kt) {
      const int k = kt * 16 + m16;
      const bool k_ok = k < K;
      const double p0 = k_ok ? g.fp[4 * k] : 1.0, p1 = k_ok ? g.fp[4 * k + 1] : 1.0,
                   p2 = k_ok ? g.fp[4 * k + 2] : 1.0;
      f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      for (int base = lo; base < hi; base += 4) {
        const int p = base + q4;
        double h = 0.0, ma = 0.0, mb = 0.0;
        if (p < hi) {
          const double *rec = b.rec + kRecDoubles * (size_t)p;
          const double r2 = rec[3], inv_r = rec[4];
          const double ux = rec[0] * inv_r, uy = rec[1] * inv_r, uz = rec[2] * inv_r;
          if (k_ok) {
            double dH;
            filter_cut(g, p0, p1, p2, sqrt(r2), r2, h, dH);
          }
          if (d0_ok) ma = pow3(ux, c0 & 3) * pow3(uy, (c0 >> 2) & 3) * pow3(uz, (c0 >> 4) & 3);
          if (d1_ok) mb = pow3(ux, c1 & 3) * pow3(uy, (c1 >> 2) & 3) * pow3(uz, (c1 >> 4) & 3);
        }
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(h, ma, acc0, 0, 0, 0);
        if (nd > 16) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(h, mb, acc1, 0, 0, 0);
      }
      // accumulator register r holds P[k' = 16 kt + q4 + 4 r][d = m16 (+ 16)]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kt * 16 + q4 + 4 * r;
        const bool kk_ok = kk < K;
        double *Prow = Pbuf + (((size_t)i * nel + tb) * K + (kk_ok ? kk : 0)) * nd;
        if (kk_ok && d0_ok) Prow[m16] = acc0[r];
        if (kk_ok && d1_ok) Prow[16 + m16] = acc1[r];
        const double s0 = acc0[r] * acc0[r], s1 = acc1[r] * acc1[r];
        const double p_lin = row16_sum(m16 == 0 ? acc0[r] : 0.0);  // P[k'][0] in every lane of the row
        double feat = 0.0;
        int col = -1;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (m > g.max_moment) break;
          double q = row16_sum(T0[m] * s0 + T1[m] * s1);
          if (m == 0) {
            // legacy: the raw sum (grap.py:425-428); new: sign(P0) sqrt(Q0 + 1e-16) (:663-672)
            const double sgn = p_lin > 0.0 ? 1.0 : (p_lin < 0.0 ? -1.0 : 0.0);
            q = g.legacy ? p_lin : sgn * sqrt(q + 1e-16);
          }
          if (m16 == m) {  // spread the stores over the lanes of the row
            feat = q;
            col = g.col_of_m[m];
          }
        }
        if (kk_ok && col >= 0) b.G[(size_t)i * ndim + ((size_t)tb * K + kk) * g.nf + col] = feat;
      }
    }
  }
}

// A[k][d] = dE/dP[k][d] = 2 P[k][d] sum_m c[k][m] T[d][m]   (+ dE/dG0 for d = 0 in legacy mode)
__global__ __launch_bounds__(kBlock) void grap_dp_kernel(GrapParams g, DeviceBatch b, const double *Pbuf,
                                                         double *Abuf, int ndim) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  const int nel = g.nel, K = g.K, nd = g.nd;
  const int total = nel * K * nd;
  for (int idx = lane; idx < total; idx += 64) {
    const int d = idx % nd, k = (idx / nd) % K, tb = idx / (nd * K);
    const size_t row = (((size_t)i * nel + tb) * K + k) * nd;
    const double P = Pbuf[row + d];
    const double *w = b.dEdG + (size_t)i * ndim + ((size_t)tb * K + k) * g.nf;
    double s = 0.0, lin = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m > g.max_moment) break;
      const int col = g.col_of_m[m];
      if (col < 0) continue;
      double c = w[col];
      if (m == 0) {
        if (g.legacy) {
          lin = (d == 0) ? c : 0.0;
          c = 0.0;
        } else {
          const double P0 = Pbuf[row];
          const double sgn = P0 > 0.0 ? 1.0 : (P0 < 0.0 ? -1.0 : 0.0);
          c = c * sgn / (2.0 * sqrt(P0 * P0 + 1e-16));
        }
      }
      s = fma(c, g.T[d][m], s);
    }
    Abuf[row + d] = 2.0 * P * s + lin;
  }
}

// One wavefront per atom: dE/dD of its directed pairs.
__global__ __launch_bounds__(kBlock) void grap_backward_kernel(GrapParams g, DeviceBatch b,
                                                               const double *Abuf) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int nel = g.nel, K = g.K, nd = g.nd;
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  const int d1 = 16 + m16 < kMaxComp ? 16 + m16 : 0;
  const int c0 = comp_code(m16), c1 = comp_code(d1);
  const bool d0_ok = m16 < nd, d1_ok = 16 + m16 < nd;
  const int nx0 = c0 & 3, ny0 = (c0 >> 2) & 3, nz0 = (c0 >> 4) & 3;
  const int nx1 = c1 & 3, ny1 = (c1 >> 2) & 3, nz1 = (c1 >> 4) & 3;
  const double deg0 = nx0 + ny0 + nz0, deg1 = nx1 + ny1 + nz1;
  for (int sb = 0; sb < nel; ++sb) {
    const int lo = seg[sb], hi = seg[sb + 1];
    const int tb = term_block(sA, sb);
    const double *A = Abuf + ((size_t)i * nel + tb) * K * nd;
    for (int j0 = lo; j0 < hi; j0 += 16) {
      // A operand rows: this lane's pair
      const int pa = j0 + m16;
      double r = 1.0, r2 = 1.0;
      const bool pa_ok = pa < hi;
      if (pa_ok) {
        r2 = b.rec[kRecDoubles * (size_t)pa + 3];
        r = sqrt(r2);
      }
      f64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, b0 = a0, b1 = a0;
      for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + q4;
        double H = 0.0, dH = 0.0, B0 = 0.0, B1 = 0.0;
        if (k < K) {
          if (pa_ok) filter_cut(g, g.fp[4 * k], g.fp[4 * k + 1], g.fp[4 * k + 2], r, r2, H, dH);
          if (d0_ok) B0 = A[(size_t)k * nd + m16];
          if (d1_ok) B1 = A[(size_t)k * nd + 16 + m16];
        }
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(H, B0, a0, 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(dH, B0, b0, 0, 0, 0);
        if (nd > 16) {
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(H, B1, a1, 0, 0, 0);
          b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(dH, B1, b1, 0, 0, 0);
        }
      }
      // register rr: pair j0 + q4 + 4 rr, component d = m16 (+ 16)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int p = j0 + q4 + 4 * rr;
        const bool p_ok = p < hi;
        double gx = 0.0, gy = 0.0, gz = 0.0;
        if (p_ok) {
          const double *rec = b.rec + kRecDoubles * (size_t)p;
          const double inv_r = rec[4];
          const double ux = rec[0] * inv_r, uy = rec[1] * inv_r, uz = rec[2] * inv_r;
          if (d0_ok) {
            const double px = pow3(ux, nx0), py = pow3(uy, ny0), pz = pow3(uz, nz0);
            const double M = px * py * pz;
            const double rad = (b0[rr] - deg0 * a0[rr] * inv_r) * M;  // multiplies u
            const double t = a0[rr] * inv_r;
            gx = fma(rad, ux, t * dpow3(ux, nx0) * py * pz);
            gy = fma(rad, uy, t * px * dpow3(uy, ny0) * pz);
            gz = fma(rad, uz, t * px * py * dpow3(uz, nz0));
          }
          if (d1_ok) {
            const double px = pow3(ux, nx1), py = pow3(uy, ny1), pz = pow3(uz, nz1);
            const double M = px * py * pz;
            const double rad = (b1[rr] - deg1 * a1[rr] * inv_r) * M;
            const double t = a1[rr] * inv_r;
            gx += fma(rad, ux, t * dpow3(ux, nx1) * py * pz);
            gy += fma(rad, uy, t * px * dpow3(uy, ny1) * pz);
            gz += fma(rad, uz, t * px * py * dpow3(uz, nz1));
          }
        }
        gx = row16_sum(gx);
        gy = row16_sum(gy);
        gz = row16_sum(gz);
        if (p_ok && m16 == 0) {
          b.g[4 * (size_t)p] = gx;
          b.g[4 * (size_t)p + 1] = gy;
          b.g[4 * (size_t)p + 2] = gz;
        }
      }
    }
  }
}

}  // namespace

struct GrapModel {
  GrapParams p;
  double *fp = nullptr;                  // device filter constants
  double *Pbuf = nullptr, *Abuf = nullptr;
  size_t cap_atoms = 0;
  int ndim = 0;
};

// grap_desc: [algorithm, K, max_moment, legacy, symmetric, moment mask (bit m = emitted),
//             then K x 3 filter constants]
GrapModel *grap_create(const ta_model_desc *m, std::string &err) {
  if (!m->grap_params || m->n_grap_params < 6) {
    err = "grap_params is missing";
    return nullptr;
  }
  const double *q = m->grap_params;
  const int algo = (int)q[0], K = (int)q[1], mm = (int)q[2], legacy = (int)q[3], symmetric = (int)q[4];
  const int mask = (int)q[5];
  if (algo < 0 || algo > 3) {
    err = "unknown GRAP algorithm";
    return nullptr;
  }
  if (K < 1 || K > kMaxFilters) {
    err = "GRAP supports 1.." + std::to_string(kMaxFilters) + " radial filters";
    return nullptr;
  }
  if (mm < 0 || mm > 3) {
    err = "GRAP moment tensors above rank 3 are not implemented";
    return nullptr;
  }
  if (legacy && mm > 2) {
    err = "GRAP legacy mode has moments 0, 1, 2 only (grap.py:423-460)";
    return nullptr;
  }
  if (m->n_grap_params != 6 + 3 * K) {
    err = "grap_params must hold 6 + 3 K doubles";
    return nullptr;
  }
  if (!(m->rcut > 0.0)) {
    err = "rcut must be positive";
    return nullptr;
  }
  GrapModel *g = new GrapModel();
  std::memset(&g->p, 0, sizeof(g->p));
  GrapParams &p = g->p;
  p.nel = m->n_elements;
  p.K = K;
  p.max_moment = mm;
  p.nd = mm == 0 ? 1 : (mm == 1 ? 4 : (mm == 2 ? 10 : 20));
  p.legacy = legacy ? 1 : 0;
  p.algo = algo;
  p.cutoff = m->cutoff_function;
  p.rcut = m->rcut;
  p.inv_rc2 = 1.0 / (m->rcut * m->rcut);
  int col = 0;
  for (int k = 0; k < 4; ++k) {
    const bool on = legacy ? ((mask >> k) & 1) && k <= 2 : k <= mm;
    p.col_of_m[k] = (on && k <= mm) ? col++ : -1;
  }
  p.nf = col;
  if (col == 0) {
    delete g;
    err = "GRAP needs at least one moment tensor";
    return nullptr;
  }
  // multiplicity tensor, grap.py:470-492 (the legacy sums are the non-symmetric one)
  const bool sym = symmetric && !legacy;
  p.T[0][0] = 1.0;
  if (mm >= 1)
    for (int d = 1; d < 4; ++d) p.T[d][1] = 1.0;
  if (mm >= 2) {
    const double t2[6] = {1, 2, 2, 1, 2, 1};
    for (int d = 0; d < 6; ++d) p.T[4 + d][2] = t2[d];
    if (sym) p.T[0][2] = -1.0 / 3.0;
  }
  if (mm >= 3) {
    const double t3[10] = {1, 3, 3, 3, 6, 3, 1, 3, 3, 1};
    for (int d = 0; d < 10; ++d) p.T[10 + d][3] = t3[d];
    if (sym)
      for (int d = 1; d < 4; ++d) p.T[d][3] = -3.0 / 5.0;
  }
  double host[kMaxFilters * 4] = {0};
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < 3; ++c) host[4 * k + c] = q[6 + 3 * k + c];
  for (int k = 0; k < K; ++k) {
    const double *f = &host[4 * k];
    const bool bad = (algo == GRAP_DENSITY && !(f[2] != 0.0)) || (algo == GRAP_PEXP && !(f[0] > 0.0));
    if (bad) {
      delete g;
      err = "GRAP filter constants out of range (re != 0, rl > 0)";
      return nullptr;
    }
  }
  if (hipMalloc((void **)&g->fp, sizeof(host)) != hipSuccess ||
      hipMemcpy(g->fp, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) {
    delete g;
    err = "device allocation failed";
    return nullptr;
  }
  p.fp = g->fp;
  g->ndim = p.nf * K * p.nel;
  return g;
}

int grap_ndim(const GrapModel *g) { return g->ndim; }

void grap_destroy(GrapModel *g) {
  if (!g) return;
  if (g->fp) (void)hipFree(g->fp);
  if (g->Pbuf) (void)hipFree(g->Pbuf);
  if (g->Abuf) (void)hipFree(g->Abuf);
  delete g;
}

void grap_ensure(GrapModel *g, const DeviceBatch &b) {
  const size_t n = (size_t)b.n_atoms;
  if (n <= g->cap_atoms) return;
  if (g->Pbuf) (void)hipFree(g->Pbuf);
  if (g->Abuf) (void)hipFree(g->Abuf);
  g->Pbuf = g->Abuf = nullptr;
  const size_t cap = n + n / 8 + 64;
  const size_t per = (size_t)g->p.nel * g->p.K * g->p.nd * sizeof(double);
  if (hipMalloc((void **)&g->Pbuf, cap * per) != hipSuccess ||
      hipMalloc((void **)&g->Abuf, cap * per) != hipSuccess)
    throw std::bad_alloc();
  g->cap_atoms = cap;
}

void launch_grap_forward(GrapModel *g, const DeviceBatch &b, hipStream_t s) {
  if (b.n_atoms == 0) return;
  hipLaunchKernelGGL(grap_forward_kernel, dim3((unsigned)((b.n_atoms * 64 + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, s, g->p, b, g->Pbuf, g->ndim);
}

void launch_grap_backward(GrapModel *g, const DeviceBatch &b, hipStream_t s) {
  if (b.n_atoms == 0) return;
  const dim3 grid((unsigned)((b.n_atoms * 64 + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(grap_dp_kernel, grid, dim3(kBlock), 0, s, g->p, b, g->Pbuf, g->Abuf, g->ndim);
  hipLaunchKernelGGL(grap_backward_kernel, grid, dim3(kBlock), 0, s, g->p, b, g->Abuf);
}

}  // namespace ta
