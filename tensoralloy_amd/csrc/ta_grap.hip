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
// Backward: A[k][d] = dE/dP[k][d] (formed while staging it in LDS), then per 16-pair tile
//     a[j][d] = sum_k H_k(r_j) A[k][d],  b[j][d] = sum_k H'_k(r_j) A[k][d]     (MFMA again)
//     dE/dD_j = sum_d b_d M_d u + a_d (dM_d/du - deg_d M_d u) / r               (row sum over d)
// Both kernels run one wavefront (= workgroup) per centre and stage its pairs in LDS (unit
// vector, r, fc, dfc/dr: the cutoff is evaluated once per pair, not once per filter).
// Forces / virial / energy then use force_gather and frame_reduce like every other model.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ta_device.h"
#include "ta_dual.h"
#include "ta_math.h"

namespace ta {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxFilters = 32;
constexpr int kMaxComp = 56;   // packed components of the moments 0..5: 1 + 3 + 6 + 10 + 15 + 21
constexpr int kSmallComp = 20; // ... of the moments 0..3 (the usual case: smaller LDS tables)
constexpr int kMaxMom = 6;
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double mlp_f64x4 __attribute__((ext_vector_type(4)));
constexpr int kMlpTileRows = 16;

enum { GRAP_SF = 0, GRAP_MORSE = 1, GRAP_DENSITY = 2, GRAP_PEXP = 3, GRAP_NN = 4 };

struct GrapParams {
  int nel, K, max_moment, nd;  // nd = 1, 4, 10, 20, 35, 56 packed components
  int nf;                      // features per (block, filter)
  int legacy, algo, cutoff;
  int col_of_m[kMaxMom];       // feature column of moment m, -1 = not emitted
  double rcut, inv_rc2;
  const double *T;             // device [nd][kMaxMom] multiplicity tensor (grap.py:470-492, :577-600)
  // device [nd] component words: bits 0-5 own index, 6-11 / 12-17 / 18-23 index of the component with
  // one power of ux / uy / uz less (0 where that exponent is 0), 24-26 / 27-29 / 30-32 the exponents
  const unsigned long long *cw;
  const double *fp;            // device [K][4] filter constants
  // algorithm `nn`: per pair the K filter values and their r-derivatives, written by
  // grap_nn_filter_kernel: Hbuf[p][0 .. Ks) = v_k(r_p), Hbuf[p][Ks .. 2 Ks) = dv_k/dr
  const double *Hbuf;
  int Ks;                      // K rounded up to 16
};

// The packed components M_d = ux^nx uy^ny uz^nz come in the reference's order (grap.py:501-511):
// degree after degree, inside a degree with nx descending, then ny (1 | x y z | xx xy xz yy yz zz |
// xxx xxy ...); their exponents and the indices of the components with one power less (what
// dM_d/du needs) are tabulated on the host (component words, see GrapParams::cw).

// v(r) and dv/dr of one radial filter (without the cutoff), from constants preprocessed on the
// host (grap_create) so that no division and no logarithm is left per (pair, filter):
//   sf      exp(-eta (r - omega)^2 / rc^2)                 c0 = eta / rc^2, c1 = omega
//   morse   D [exp(-2 g (r - r0)) - 2 exp(-g (r - r0))]    c0 = D, c1 = gamma, c2 = r0
//   density A exp(-beta (r / re - 1))                      c0 = A exp(beta), c1 = beta / re
//   pexp    exp(-(r / rl)^pl) = exp(-exp(pl (ln r - ln rl)))   c0 = pl, c1 = ln rl
// `logr` = ln r and `inv_r` are per-pair values staged in LDS.
__device__ __forceinline__ void filter_fn(int algo, double c0, double c1, double c2, double r,
                                          double logr, double inv_r, double &v, double &dv) {
  switch (algo) {
    case GRAP_SF: {
      const double t = r - c1;
      v = ta_exp(-c0 * t * t);
      dv = v * (-2.0 * c0 * t);
      break;
    }
    case GRAP_MORSE: {
      const double e1 = ta_exp(-c1 * (r - c2));
      const double e2 = e1 * e1;
      v = c0 * (e2 - 2.0 * e1);
      dv = c0 * c1 * (2.0 * e1 - 2.0 * e2);
      break;
    }
    case GRAP_DENSITY: {
      v = c0 * ta_exp(-c1 * r);
      dv = -c1 * v;
      break;
    }
    default: {
      const double xp = ta_exp(c0 * (logr - c1));
      v = ta_exp(-xp);
      dv = v * (-c0 * xp * inv_r);
      break;
    }
  }
}

// block index of neighbour species sb for centre species sA: [AA, AB (B != A sorted)]
__device__ __forceinline__ int term_block(int sA, int sb) { return sb == sA ? 0 : (sb < sA ? sb + 1 : sb); }

constexpr int kWave = 64;  // one wavefront per workgroup = per centre atom
constexpr int kFwdChunk = 64;  // pairs of one (centre, species) segment staged in LDS at a time
constexpr int kBwdChunk = 64;

// Per-pair fields staged in LDS: r, ln r, 1/r, fc(r), dfc/dr, the unit vector and the table of
// all packed monomials M[t][d] (every lane of a 16-lane row needs a different component of the
// same pair: one ds_read instead of a per-lane select chain). Entries n .. round_up(n, 16) are
// zero-filled so that the MFMA loops need no tail predicates.
template <int CH, int MC>
struct PairLds {
  double r[CH], logr[CH], inv_r[CH], f[CH], df[CH], ux[CH], uy[CH], uz[CH];
  double M[CH][MC];
};

// GEOM: the pair geometry D = Rj - Ri + S.h, r^2 = D.D + eps (universal.py:448-474) is computed here
// (forward pass) and left in the pair record for the backward kernel and the force gather.
// `cwl`: the component words in LDS.
template <int CH, int MC, bool GEOM>
__device__ __forceinline__ void stage_pairs(const GrapParams &g, const DeviceBatch &b, int first, int n,
                                            PairLds<CH, MC> &L, const unsigned long long *cwl, int lane,
                                            int64_t centre = 0, double eps = 0.0) {
  const int npad = min(CH, (n + 15) & ~15);
  for (int t = lane; t < npad; t += kWave) {
    double ux = 0.0, uy = 0.0, uz = 0.0, r = 1.0, inv_r = 1.0, f = 0.0, df = 0.0, one = 0.0;
    if (t < n) {
      double dx, dy, dz, r2;
      if constexpr (GEOM) {
        const size_t q = (size_t)(first + t);
        const int j = b.pair_j[q];
        const double *h = b.cells + 9 * (size_t)b.frame_of_atom[centre];
        const double sx = (double)b.pair_shift[3 * q], sy = (double)b.pair_shift[3 * q + 1],
                     sz = (double)b.pair_shift[3 * q + 2];
        const double *ri = b.pos + 3 * (size_t)centre, *rj = b.pos + 3 * (size_t)j;
        dx = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
        dy = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
        dz = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
        r2 = dx * dx + dy * dy + dz * dz + eps;
        inv_r = 1.0 / sqrt(r2);
        if (b.rec4) {  // compact 32-byte record {D, r^2}: the readers recompute 1 / r
          double2 *dst = reinterpret_cast<double2 *>(b.rec4 + 4 * q);
          dst[0] = make_double2(dx, dy);
          dst[1] = make_double2(dz, r2);
        } else {
          double2 *dst = reinterpret_cast<double2 *>(b.rec + kRecDoubles * q);
          dst[0] = make_double2(dx, dy);
          dst[1] = make_double2(dz, r2);
          dst[2] = make_double2(inv_r, 0.0);
        }
      } else {
        const double2 *rec = pair_geom(b, (size_t)(first + t));
        const double2 v0 = rec[0], v1 = rec[1];
        dx = v0.x;
        dy = v0.y;
        dz = v1.x;
        r2 = v1.y;
        inv_r = b.rec4 ? 1.0 / sqrt(r2) : b.rec[kRecDoubles * (size_t)(first + t) + 4];
      }
      r = r2 * inv_r;  // sqrt(r2)
      ux = dx * inv_r;
      uy = dy * inv_r;
      uz = dz * inv_r;
      const double u = r2 * g.inv_rc2;
      double dfdu = 0.0;
      if (u < 1.0) cutoff_u(g.cutoff, u, f, dfdu);
      df = dfdu * 2.0 * r * g.inv_rc2;
      one = 1.0;
    }
    L.r[t] = r;
    L.logr[t] = g.algo == GRAP_PEXP ? log(r) : 0.0;
    L.inv_r[t] = inv_r;
    L.f[t] = f;
    L.df[t] = df;
    L.ux[t] = ux;
    L.uy[t] = uy;
    L.uz[t] = uz;
    // every monomial is its parent (one power of one axis less, always an earlier component) times
    // that axis: one multiplication per component
    double *M = L.M[t];
    M[0] = one;
    if constexpr (MC <= kSmallComp) {  // moments 0..3: straight-line code
      if (g.nd > 1) {
        M[1] = ux;
        M[2] = uy;
        M[3] = uz;
      }
      if (g.nd > 4) {
        const double xx = ux * ux, xy = ux * uy, xz = ux * uz, yy = uy * uy, yz = uy * uz, zz = uz * uz;
        M[4] = xx;
        M[5] = xy;
        M[6] = xz;
        M[7] = yy;
        M[8] = yz;
        M[9] = zz;
        if (g.nd > 10) {
          M[10] = xx * ux;
          M[11] = xx * uy;
          M[12] = xx * uz;
          M[13] = xy * uy;
          M[14] = xy * uz;
          M[15] = xz * uz;
          M[16] = yy * uy;
          M[17] = yy * uz;
          M[18] = yz * uz;
          M[19] = zz * uz;
        }
      }
    } else {
      for (int d = 1; d < g.nd; ++d) {
        const unsigned long long w = cwl[d];
        const int ex = (int)((w >> 24) & 7), ey = (int)((w >> 27) & 7);
        const int parent = ex ? (int)((w >> 6) & 63) : (ey ? (int)((w >> 12) & 63) : (int)((w >> 18) & 63));
        M[d] = M[parent] * (ex ? ux : (ey ? uy : uz));
      }
    }
  }
}

// One wavefront (= workgroup) per atom: P (kept for the backward pass) and the features.
// MC = size of the monomial table: 20 (moments 0..3) or 56 (moments 4, 5): NT = 2 or 4 column tiles.
template <int MC>
__global__ __launch_bounds__(kWave) void grap_forward_kernel(GrapParams g, DeviceBatch b, double *Pbuf,
                                                            int ndim, double eps) {
  constexpr int NT = (MC + 15) / 16;
  __shared__ PairLds<kFwdChunk, MC> L;
  __shared__ unsigned long long cwl[MC];
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int nel = g.nel, K = g.K, nd = g.nd;
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  for (int d = lane; d < MC; d += kWave) cwl[d] = d < nd ? g.cw[d] : 0ull;
  bool d_ok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) d_ok[t] = 16 * t + m16 < nd;
  // multiplicities of this lane's components (one load per kernel, not per block and filter tile)
  constexpr int kMomHere = MC <= kSmallComp ? 4 : kMaxMom;
  double Tm[NT][kMomHere];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int m = 0; m < kMomHere; ++m)
      Tm[ct][m] = (d_ok[ct] && m <= g.max_moment) ? g.T[(16 * ct + m16) * kMaxMom + m] : 0.0;
  for (int sb = 0; sb < nel; ++sb) {
    const int lo = seg[sb], hi = seg[sb + 1];
    const int tb = term_block(sA, sb);
    for (int kt = 0; kt * 16 < K; ++kt) {
      const int k = kt * 16 + m16;
      const bool k_ok = k < K;
      const double fp0 = k_ok ? g.fp[4 * k] : 0.0, fp1 = k_ok ? g.fp[4 * k + 1] : 0.0,
                   fp2 = k_ok ? g.fp[4 * k + 2] : 0.0;
      f64x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = {0.0, 0.0, 0.0, 0.0};
      for (int first = lo; first < hi; first += kFwdChunk) {
        const int n = min(kFwdChunk, hi - first);
        __syncthreads();
        if (kt == 0)
          stage_pairs<kFwdChunk, MC, true>(g, b, first, n, L, cwl, lane, i, eps);
        else
          stage_pairs<kFwdChunk, MC, false>(g, b, first, n, L, cwl, lane);
        __syncthreads();
        for (int base = 0; base < n; base += 4) {
          const int t = base + q4;  // < round_up(n, 16): staged (zero beyond n)
          double v, dv;
          if (g.algo == GRAP_NN)  // rows beyond n belong to later pairs (or the padding): f = 0 there
            v = g.Hbuf[(size_t)(first + t) * (2 * g.Ks) + (k_ok ? k : 0)];
          else
            filter_fn(g.algo, fp0, fp1, fp2, L.r[t], L.logr[t], L.inv_r[t], v, dv);
          const double h = k_ok ? v * L.f[t] : 0.0;
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
            if (16 * ct < nd)
              acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(h, d_ok[ct] ? L.M[t][16 * ct + m16] : 0.0, acc[ct], 0, 0, 0);
        }
      }
      // accumulator register r of tile ct holds P[k' = 16 kt + q4 + 4 r][d = 16 ct + m16]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kt * 16 + q4 + 4 * r;
        const bool kk_ok = kk < K;
        double *Prow = Pbuf + (((size_t)i * nel + tb) * K + (kk_ok ? kk : 0)) * nd;
        double sq[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const double pv = acc[ct][r];
          if (kk_ok && d_ok[ct]) Prow[16 * ct + m16] = pv;
          sq[ct] = pv * pv;
        }
        const double p_lin = row16_sum(m16 == 0 ? acc[0][r] : 0.0);  // P[k'][0] in every lane of the row
        double feat = 0.0;
        int col = -1;
#pragma unroll
        for (int m = 0; m < kMomHere; ++m) {
          if (m > g.max_moment) break;
          double part = 0.0;
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) part = fma(Tm[ct][m], sq[ct], part);
          double q = row16_sum(part);
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

// One wavefront (= workgroup) per atom: dE/dD of its directed pairs.
//   A[k][d] = dE/dP[k][d] = 2 P[k][d] sum_m c[k][m] T[d][m]  (+ dE/dG0 for d = 0 in legacy mode)
// is formed while staging it in LDS. dM_d/du_c = n_c M_{d - e_c}: a second read of the monomial
// table at the index of the component with one power of u_c less.
template <int MC>
__global__ __launch_bounds__(kWave) void grap_backward_kernel(GrapParams g, DeviceBatch b,
                                                             const double *Pbuf, int ndim) {
  constexpr int NT = (MC + 15) / 16;
  __shared__ PairLds<kBwdChunk, MC> L;
  __shared__ unsigned long long cwl[MC];
  extern __shared__ double dyn[];  // A[Kp][nd], then FP[4 K]
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int nel = g.nel, K = g.K, nd = g.nd;
  const int Kp = (K + 3) & ~3;
  double *A = dyn, *FP = dyn + Kp * nd, *wrow = FP + 4 * K;  // wrow: this atom's dE/dG row
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  // The pair tiles are computed transposed (rows = components, columns = pairs): accumulator
  // register r of tile ct holds component d = 16 ct + q4 + 4 r of this lane's OWN pair j = m16, so
  // the sum over components is 4 NT terms in registers plus one exchange between the four 16-lane
  // rows, and the pair's unit vector is read once. The constants of those components (table indices
  // of M_{d - e_c}, exponents) are the component words in LDS.
  for (int d = lane; d < MC; d += kWave) cwl[d] = d < nd ? g.cw[d] : 0ull;
  __shared__ double Tl[MC * kMaxMom];  // multiplicities (read once per atom and term block below)
  for (int t = lane; t < nd * kMaxMom; t += kWave) Tl[t] = g.T[t];
  for (int t = lane; t < 4 * K; t += kWave) FP[t] = g.fp[t];
  for (int t = lane; t < ndim; t += kWave) wrow[t] = b.dEdG[(size_t)i * ndim + t];
  // own-side sums of this centre for force_gather (DeviceBatch::fown): after the row exchange below all
  // four 16-lane rows hold a pair's dE/dD, so row 0 sums g and row c + 1 sums g_c D (the virial row)
  double own[3] = {0.0, 0.0, 0.0};
  for (int sb = 0; sb < nel; ++sb) {
    const int lo = seg[sb], hi = seg[sb + 1];
    if (lo == hi) continue;
    const int tb = term_block(sA, sb);
    __syncthreads();
    for (int idx = lane; idx < Kp * nd; idx += kWave) {
      const int d = idx % nd, k = idx / nd;
      double val = 0.0;
      if (k < K) {
        const size_t row = (((size_t)i * nel + tb) * K + k) * nd;
        const double P = Pbuf[row + d];
        const double *w = wrow + ((size_t)tb * K + k) * g.nf;
        double s = 0.0, lin = 0.0;
#pragma unroll
        for (int m = 0; m < kMaxMom; ++m) {
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
          s = fma(c, Tl[d * kMaxMom + m], s);
        }
        val = 2.0 * P * s + lin;
      }
      A[k * nd + d] = val;
    }
    for (int first = lo; first < hi; first += kBwdChunk) {
      const int n = min(kBwdChunk, hi - first);
      __syncthreads();
      stage_pairs<kBwdChunk, MC, false>(g, b, first, n, L, cwl, lane);
      __syncthreads();
      for (int j0 = 0; j0 < n; j0 += 16) {
        // B-operand columns: this lane's pair (zero-filled beyond n: f = df = 0)
        const int ta = j0 + m16;
        const double r = L.r[ta], logr = L.logr[ta], inv_r = L.inv_r[ta], f = L.f[ta], df = L.df[ta];
        f64x4 av[NT], bv[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          av[ct] = {0.0, 0.0, 0.0, 0.0};
          bv[ct] = {0.0, 0.0, 0.0, 0.0};
        }
        for (int k0 = 0; k0 < Kp; k0 += 4) {
          const int k = k0 + q4;
          double H = 0.0, dH = 0.0;
          if (k < K) {
            double v, dv;
            if (g.algo == GRAP_NN) {
              const double *hp = g.Hbuf + (size_t)(first + ta) * (2 * g.Ks);
              v = hp[k];
              dv = hp[g.Ks + k];
            } else {
              filter_fn(g.algo, FP[4 * k], FP[4 * k + 1], FP[4 * k + 2], r, logr, inv_r, v, dv);
            }
            H = v * f;
            dH = dv * f + v * df;
          }
          // A operand = A^T[d = 16 ct + m16][k], B operand = H[k][pair = m16]
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
            if (16 * ct < nd) {
              const double At = (16 * ct + m16 < nd) ? A[k * nd + 16 * ct + m16] : 0.0;
              av[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(At, H, av[ct], 0, 0, 0);
              bv[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(At, dH, bv[ct], 0, 0, 0);
            }
        }
        // this lane: pair ta, components 16 ct + q4 + 4 rr
        const double ux = L.ux[ta], uy = L.uy[ta], uz = L.uz[ta];
        const double *M = L.M[ta];
        double gx = 0.0, gy = 0.0, gz = 0.0;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            if (16 * ct + 4 * rr >= MC) continue;  // compile time: no such component in this kernel
            const int d = 16 * ct + q4 + 4 * rr;
            if (d >= nd) continue;  // (accumulators of components beyond nd are zero anyway)
            // (component words from LDS: holding them in registers costs 54 VGPRs and a wave of occupancy)
            const unsigned long long w = cwl[d];
            const int px = (int)((w >> 6) & 63), py = (int)((w >> 12) & 63), pz = (int)((w >> 18) & 63);
            const int ex = (int)((w >> 24) & 7), ey = (int)((w >> 27) & 7), ez = (int)((w >> 30) & 7);
            const double ad = av[ct][rr], bd = bv[ct][rr];
            const double at = ad * inv_r;
            const double rad = (bd - (double)(ex + ey + ez) * at) * M[d];  // multiplies u
            gx = fma(rad, ux, fma(at * (double)ex, M[px], gx));
            gy = fma(rad, uy, fma(at * (double)ey, M[py], gy));
            gz = fma(rad, uz, fma(at * (double)ez, M[pz], gz));
          }
        // components live in the four 16-lane rows: lanes m16, m16 + 16, + 32, + 48
        gx += __shfl_xor(gx, 16);
        gy += __shfl_xor(gy, 16);
        gz += __shfl_xor(gz, 16);
        gx += __shfl_xor(gx, 32);
        gy += __shfl_xor(gy, 32);
        gz += __shfl_xor(gz, 32);
        if (ta < n && q4 == 0) {
          double *dst = b.g + 4 * (size_t)(first + ta);
          dst[0] = gx;
          dst[1] = gy;
          dst[2] = gz;
        }
        if (ta < n) {
          const double gc = q4 == 1 ? gx : (q4 == 2 ? gy : gz);
          own[0] += q4 == 0 ? gx : gc * (r * ux);
          own[1] += q4 == 0 ? gy : gc * (r * uy);
          own[2] += q4 == 0 ? gz : gc * (r * uz);
        }
      }
    }
  }
  if (b.own_sums) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const double v = row16_sum(own[e]);
      if (m16 == 0) b.fown[12 * (size_t)i + 3 * q4 + e] = v;
    }
  }
}

// ---- the `nn` algorithm: one shared filter network r -> K filter values (grap.py:220-270, :632-643) --
constexpr int kNetMaxLayers = 8;   // dense layers incl. the linear output layer
constexpr int kNetMaxWidth = 64;   // padded layer width (4 column tiles)

struct GrapNet {
  int L, act, resnet;
  int np[kNetMaxLayers];      // padded output width of every layer (multiple of 16)
  int res[kNetMaxLayers];     // skip connection on this layer (convolutional.py:272-273)
  const double *w[kNetMaxLayers];  // layer 0: [np0] (K = 1); else [kp = np[l-1]][np[l]] row-major, zero padded
  const double *b[kNetMaxLayers];  // [np[l]], zeros where the layer has no bias
  int xs;                     // row stride of the activation buffers (2 or 18 mod 32: conflict-free A reads)
  // input of the network (grap.py:620-631): 0: r; 1: r / rcov; 2: exp(-r / rcov), rcov = covalent
  // radius of the CENTRE's element
  int modifier;
  double inv_rcov[kMaxElements];
};

__host__ __device__ inline int net_wstride(int np) { return (np % 32 == 0) ? np + 16 : np; }

// Value and r-derivative of all K filters for every pair, forward mode: a wavefront owns 16 pairs;
// the activations h and dh/dr of the current layer sit in the wavefront's LDS rows as [pair][unit]
// (the A operand of v_mfma_f64_16x16x4_f64), [h; h'] . W shares the B operand read from the
// workgroup's LDS image of the weights; the accumulator layout puts a pair's value and derivative of
// one unit in the same lane, so the activation is applied in registers. Layer 0 has K = 1 and needs
// no GEMM. Pair geometry: r = sqrt(r^2) of the pair records (pair_geometry_kernel ran before).
// NTMAX = column tiles of the widest layer (1..4): sizes the accumulator arrays
template <int ACT, int NTMAX>
__global__ __launch_bounds__(kBlock) void grap_nn_filter_kernel(GrapNet net, DeviceBatch b, double *Hbuf,
                                                               int Ks) {
  extern __shared__ double lds[];
  const int act = ACT >= 0 ? ACT : net.act;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kq = lane >> 4;
  // LDS: w0, b0, then per GEMM layer W (padded stride) and bias, then the wavefronts' X buffers
  double *w0 = lds, *b0 = w0 + net.np[0];
  double *Wl[kNetMaxLayers], *Bl[kNetMaxLayers];
  double *cur = b0 + net.np[0];
  for (int l = 1; l < net.L; ++l) {
    Wl[l] = cur;
    cur += (size_t)net.np[l - 1] * net_wstride(net.np[l]);
    Bl[l] = cur;
    cur += net.np[l];
  }
  double *Xv = cur + (size_t)wave * 2 * kMlpTileRows * net.xs, *Xd = Xv + (size_t)kMlpTileRows * net.xs;
  for (int idx = tid; idx < net.np[0]; idx += kBlock) {
    w0[idx] = net.w[0][idx];
    b0[idx] = net.b[0][idx];
  }
  for (int l = 1; l < net.L; ++l) {
    const int kp = net.np[l - 1], np = net.np[l], ws = net_wstride(np);
    for (int idx = tid; idx < kp * np; idx += kBlock) {
      const int row = idx / np, col = idx - row * np;
      Wl[l][row * ws + col] = net.w[l][idx];
    }
    for (int idx = tid; idx < np; idx += kBlock) Bl[l][idx] = net.b[l][idx];
  }
  __syncthreads();
  const int64_t ntiles = (b.n_pairs + kMlpTileRows - 1) / kMlpTileRows;
  // every wavefront of the workgroup makes the same number of trips (block-level barriers inside)
  for (int64_t base = (int64_t)blockIdx.x * 4; base < ntiles; base += (int64_t)gridDim.x * 4) {
    const int64_t t = base + wave;
    const int64_t p = t * kMlpTileRows + m;
    const bool valid = t < ntiles && p < b.n_pairs;
    double x = valid ? sqrt(b.rec[kRecDoubles * (size_t)p + 3]) : 0.0;
    double dxdr = 1.0;  // d(input) / dr: seeds the forward-mode derivative
    if (net.modifier && valid) {
      const double ir = net.inv_rcov[b.species[b.pair_i[p]]];
      if (net.modifier == 1) {
        x *= ir;
        dxdr = ir;
      } else {
        x = exp(-x * ir);
        dxdr = -x * ir;
      }
    }
    // np[0] is a multiple of 16: four independent activation chains per trip
    for (int c0 = kq; c0 < net.np[0]; c0 += 16) {
      double h[4], dh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + 4 * j;
        const double wk = w0[c];
        activation_fn(act, fma(wk, x, b0[c]), h[j], dh[j]);
        dh[j] *= wk * dxdr;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Xv[m * net.xs + c0 + 4 * j] = h[j];
        Xd[m * net.xs + c0 + 4 * j] = dh[j];
      }
    }
    __syncthreads();
    for (int l = 1; l < net.L; ++l) {
      const int kp = net.np[l - 1], NT = net.np[l] / 16, ws = net_wstride(net.np[l]);
      mlp_f64x4 accv[NTMAX], accd[NTMAX];
#pragma unroll
      for (int nt = 0; nt < NTMAX; ++nt) {
        const double bb = nt < NT ? Bl[l][16 * nt + m] : 0.0;
        accv[nt] = {bb, bb, bb, bb};
        accd[nt] = {0.0, 0.0, 0.0, 0.0};
      }
      for (int kk = 0; kk < kp / 4; ++kk) {
        const int ki = 4 * kk + kq;
        const double av = Xv[m * net.xs + ki], ad = Xd[m * net.xs + ki];
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt)
          if (nt < NT) {
            const double B = Wl[l][ki * ws + 16 * nt + m];
            accv[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, B, accv[nt], 0, 0, 0);
            accd[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, B, accd[nt], 0, 0, 0);
          }
      }
      __syncthreads();  // every read of this layer's input is done
      if (l < net.L - 1) {
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt)
          if (nt < NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = kq + 4 * q, col = 16 * nt + m;
              double h, dh;
              activation_fn(act, accv[nt][q], h, dh);
              double hd = dh * accd[nt][q];
              if (net.res[l]) {  // x = act(w x + b) + x
                h += Xv[row * net.xs + col];
                hd += Xd[row * net.xs + col];
              }
              Xv[row * net.xs + col] = h;
              Xd[row * net.xs + col] = hd;
            }
          }
        __syncthreads();
      } else {
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt)
          if (nt < NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int64_t pp = t * kMlpTileRows + kq + 4 * q;
              const int col = 16 * nt + m;
              if (t < ntiles && pp < b.n_pairs && col < Ks) {
                Hbuf[(size_t)pp * (2 * Ks) + col] = accv[nt][q];
                Hbuf[(size_t)pp * (2 * Ks) + Ks + col] = accd[nt][q];
              }
            }
          }
      }
    }
  }
}

// ---- analytic Hessian-vector products (round 3) ------------------------------------------------------
// The tangent of g[p] = dE/dD_p along (D-dot, w-dot): this file's forward and backward expressions in dual
// arithmetic, w = dE/dG dual (w-dot = H_mlp G-dot from the second-order MLP pass), D dual. One wavefront per
// centre, a pair per lane: P[k][d] = sum_j H_k(r_j) M_d(u_j) (dual, LDS), A[k][d] = dE/dP[k][d] from it,
// then per pair dE/dD = sum_d b_d M_d u + a_d (dM_d/du - deg_d M_d u) / r with a = sum_k H_k A, b = sum_k
// H'_k A. Not a throughput kernel (no matrix cores, wavefront reductions per (k, d)): Hessians are for
// cells of tens to hundreds of atoms. The four analytic filter families in dual arithmetic; the `nn` filter
// network through a value / first / second derivative sweep per pair (grap_net_d2).
template <typename T>
__device__ __forceinline__ void filter_fn_t(int algo, double c0, double c1, double c2, T r, T &v, T &dv) {
  switch (algo) {
    case GRAP_SF: {
      const T t = r - c1;
      v = t_exp(-(c0 * (t * t)));
      dv = v * (-2.0 * c0 * t);
      break;
    }
    case GRAP_MORSE: {
      const T e1 = t_exp(-(c1 * (r - c2)));
      const T e2 = e1 * e1;
      v = c0 * (e2 - 2.0 * e1);
      dv = (c0 * c1) * (2.0 * e1 - 2.0 * e2);
      break;
    }
    case GRAP_DENSITY: {
      v = c0 * t_exp(-(c1 * r));
      dv = -c1 * v;
      break;
    }
    default: {
      const T xp = t_exp(c0 * (t_log(r) - c1));
      v = t_exp(-xp);
      dv = v * (-c0 * xp / r);
      break;
    }
  }
}

// The K filters of the `nn` algorithm with their first AND second r-derivative for one pair (second-order
// forward mode through the 1x1 CNN, grap_nn_filter_kernel's network evaluated by one lane; analytic
// Hessian-vector products only).
__device__ __forceinline__ void grap_net_d2(const GrapNet &net, double r, double inv_rcov, int K, double *v,
                                            double *v1, double *v2) {
  double x = r, x1 = 1.0, x2 = 0.0;
  if (net.modifier == 1) {
    x = r * inv_rcov;
    x1 = inv_rcov;
  } else if (net.modifier == 2) {
    x = exp(-r * inv_rcov);
    x1 = -x * inv_rcov;
    x2 = x * inv_rcov * inv_rcov;
  }
  double h[2][kNetMaxWidth], h1[2][kNetMaxWidth], h2[2][kNetMaxWidth];
  int cur = 0;
  for (int c = 0; c < net.np[0]; ++c) {  // layer 0: one input
    const double wk = net.w[0][c];
    const double z = fma(wk, x, net.b[0][c]), z1 = wk * x1, z2 = wk * x2;
    double a = z, da = 1.0, d2a = 0.0;
    if (net.L > 1) activation_fn2(net.act, z, a, da, d2a);
    h[0][c] = a;
    h1[0][c] = da * z1;
    h2[0][c] = d2a * z1 * z1 + da * z2;
  }
  for (int l = 1; l < net.L; ++l) {
    const int kp = net.np[l - 1], np = net.np[l], nxt = cur ^ 1;
    const bool last = l == net.L - 1;
    for (int n = 0; n < np; ++n) {
      double z = net.b[l][n], z1 = 0.0, z2 = 0.0;
      for (int k = 0; k < kp; ++k) {
        const double w = net.w[l][(size_t)k * np + n];
        z = fma(w, h[cur][k], z);
        z1 = fma(w, h1[cur][k], z1);
        z2 = fma(w, h2[cur][k], z2);
      }
      double a = z, da = 1.0, d2a = 0.0;
      if (!last) activation_fn2(net.act, z, a, da, d2a);
      double o = a, o1 = da * z1, o2 = d2a * z1 * z1 + da * z2;
      if (!last && net.res[l]) {  // x = act(w x + b) + x
        o += h[cur][n];
        o1 += h1[cur][n];
        o2 += h2[cur][n];
      }
      h[nxt][n] = o;
      h1[nxt][n] = o1;
      h2[nxt][n] = o2;
    }
    cur = nxt;
  }
  for (int k = 0; k < K; ++k) {
    v[k] = h[cur][k];
    v1[k] = h1[cur][k];
    v2[k] = h2[cur][k];
  }
}

__global__ __launch_bounds__(kWave) void grap_hvp_kernel(GrapParams g, GrapNet net, DeviceBatch b, int ndim,
                                                         double eps, const double *__restrict__ Dv,
                                                         const double *__restrict__ Dd,
                                                         const double *__restrict__ wdot, double *gv, double *gd) {
  __shared__ Dual PA[kMaxFilters * kMaxComp];  // P[k][d], then A[k][d] in place
  __shared__ Dual C0[kMaxFilters];             // new mode, moment 0: w0 sgn(P0) / (2 sqrt(P0^2 + 1e-16)) per filter
  __shared__ unsigned long long cwl[kMaxComp];
  __shared__ double Tl[kMaxComp * kMaxMom];
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const int nel = g.nel, K = g.K, nd = g.nd;
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  for (int d = lane; d < nd; d += kWave) cwl[d] = g.cw[d];
  for (int t = lane; t < nd * kMaxMom; t += kWave) Tl[t] = g.T[t];
  const double *wv = b.dEdG + (size_t)i * ndim, *wd = wdot + (size_t)i * ndim;
  // a pair of this lane: geometry, filters with the cutoff folded in (H, H'), monomials; all dual
  auto pair_fields = [&](int q, bool valid, Dual (&u)[3], Dual &inv_r, Dual (&H)[kMaxFilters], Dual (&Hp)[kMaxFilters],
                         Dual (&M)[kMaxComp]) {
    Dual D[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      D[c] = valid ? make_dual(Dv[4 * (size_t)q + c], Dd[4 * (size_t)q + c]) : make_dual(c == 0 ? 1.0 : 0.0);
    const Dual r2 = D[0] * D[0] + D[1] * D[1] + D[2] * D[2] + eps;
    const Dual r = t_sqrt(r2);
    inv_r = 1.0 / r;
#pragma unroll
    for (int c = 0; c < 3; ++c) u[c] = D[c] * inv_r;
    const Dual uu = r2 * g.inv_rc2;
    Dual f = make_dual(0.0), df = make_dual(0.0);
    if (valid && uu.v < 1.0) {
      const double x = uu.v;
      double fv, d1, d2;
      cutoff_u2(g.cutoff, x, fv, d1, d2);
      f = make_dual(fv, d1 * uu.d);
      const Dual dfdu = make_dual(d1, d2 * uu.d);
      df = dfdu * (2.0 * g.inv_rc2) * r;
    }
    if (g.algo == GRAP_NN) {
      double nv[kMaxFilters], n1[kMaxFilters], n2[kMaxFilters];
      grap_net_d2(net, r.v, net.inv_rcov[sA], K, nv, n1, n2);
      for (int k = 0; k < K; ++k) {
        const Dual v = make_dual(nv[k], n1[k] * r.d), dv = make_dual(n1[k], n2[k] * r.d);
        H[k] = v * f;
        Hp[k] = dv * f + v * df;
      }
    } else {
      for (int k = 0; k < K; ++k) {
        Dual v, dv;
        filter_fn_t<Dual>(g.algo, g.fp[4 * k], g.fp[4 * k + 1], g.fp[4 * k + 2], r, v, dv);
        H[k] = v * f;
        Hp[k] = dv * f + v * df;
      }
    }
    M[0] = make_dual(valid ? 1.0 : 0.0);
    for (int d = 1; d < nd; ++d) {
      const unsigned long long w = cwl[d];
      const int ex = (int)((w >> 24) & 7), ey = (int)((w >> 27) & 7);
      const int parent = ex ? (int)((w >> 6) & 63) : (ey ? (int)((w >> 12) & 63) : (int)((w >> 18) & 63));
      M[d] = M[parent] * (ex ? u[0] : (ey ? u[1] : u[2]));
    }
  };
  for (int sb = 0; sb < nel; ++sb) {
    const int lo = seg[sb], hi = seg[sb + 1];
    if (lo == hi) continue;
    const int tb = term_block(sA, sb);
    __syncthreads();
    for (int idx = lane; idx < K * nd; idx += kWave) PA[idx] = make_dual(0.0);
    __syncthreads();
    for (int first = lo; first < hi; first += kWave) {
      const int q = first + lane;
      const bool valid = q < hi;
      Dual u[3], inv_r, H[kMaxFilters], Hp[kMaxFilters], M[kMaxComp];
      pair_fields(q, valid, u, inv_r, H, Hp, M);
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < nd; ++d) {
          const Dual t = H[k] * M[d];
          const double sv = wave_sum(valid ? t.v : 0.0), sd = wave_sum(valid ? t.d : 0.0);
          if (lane == 0) PA[k * nd + d] += make_dual(sv, sd);
        }
    }
    __syncthreads();
    // A[k][d] = dE/dP[k][d] (grap_backward_kernel's expression), in place
    const int col0 = g.col_of_m[0];
    for (int k = lane; k < K; k += kWave) {
      Dual c = make_dual(0.0);
      if (col0 >= 0 && !g.legacy) {
        const int col = (tb * K + k) * g.nf + col0;
        const Dual w0 = make_dual(wv[col], wd[col]);
        const Dual P0 = PA[k * nd];
        const double sgn = P0.v > 0.0 ? 1.0 : (P0.v < 0.0 ? -1.0 : 0.0);
        c = w0 * sgn / (2.0 * t_sqrt(P0 * P0 + 1e-16));
      }
      C0[k] = c;
    }
    __syncthreads();
    for (int idx = lane; idx < K * nd; idx += kWave) {
      const int d = idx % nd, k = idx / nd;
      Dual s = make_dual(0.0), lin = make_dual(0.0);
      for (int m = 0; m < kMaxMom; ++m) {
        if (m > g.max_moment) break;
        const int cm = g.col_of_m[m];
        if (cm < 0) continue;
        const int col = (tb * K + k) * g.nf + cm;
        Dual c = make_dual(wv[col], wd[col]);
        if (m == 0) {
          if (g.legacy) {
            if (d == 0) lin = c;
            continue;
          }
          c = C0[k];
        }
        s = s + c * Tl[d * kMaxMom + m];
      }
      PA[idx] = 2.0 * PA[idx] * s + lin;
    }
    __syncthreads();
    for (int first = lo; first < hi; first += kWave) {
      const int q = first + lane;
      const bool valid = q < hi;
      Dual u[3], inv_r, H[kMaxFilters], Hp[kMaxFilters], M[kMaxComp];
      pair_fields(q, valid, u, inv_r, H, Hp, M);
      Dual gq[3] = {make_dual(0.0), make_dual(0.0), make_dual(0.0)};
      for (int d = 0; d < nd; ++d) {
        Dual a = make_dual(0.0), bb = make_dual(0.0);
        for (int k = 0; k < K; ++k) {
          const Dual A = PA[k * nd + d];
          a = a + A * H[k];
          bb = bb + A * Hp[k];
        }
        const unsigned long long w = cwl[d];
        const int e[3] = {(int)((w >> 24) & 7), (int)((w >> 27) & 7), (int)((w >> 30) & 7)};
        const int par[3] = {(int)((w >> 6) & 63), (int)((w >> 12) & 63), (int)((w >> 18) & 63)};
        const double deg = (double)(e[0] + e[1] + e[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const Dual dM = e[c] ? (double)e[c] * M[par[c]] : make_dual(0.0);
          gq[c] = gq[c] + bb * M[d] * u[c] + a * (dM - deg * M[d] * u[c]) * inv_r;
        }
      }
      if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          gv[4 * (size_t)q + c] = gq[c].v;
          gd[4 * (size_t)q + c] = gq[c].d;
        }
      }
    }
  }
}

}  // namespace

struct GrapModel {
  GrapParams p;
  double *fp = nullptr;                  // device filter constants
  double *T_dev = nullptr;               // device multiplicity tensor [nd][kMaxMom]
  unsigned long long *cw_dev = nullptr;  // device component words [nd]
  double *Pbuf = nullptr;
  size_t cap_atoms = 0;
  int ndim = 0;
  // algorithm `nn`
  GrapNet net;
  std::vector<double *> owned;           // device copies of the network's weights
  double *Hbuf = nullptr;
  size_t cap_pairs = 0;
  size_t net_lds = 0;
};

void grap_destroy(GrapModel *g);

namespace {

// [L, activation, use_resnet_dt, h_abck_modifier, sizes 1, h1, ..., K, then per layer W [in][out], b [out]]
void build_filter_net(GrapModel *g, const double *q, int n, int K, std::string &err) {
  auto fail = [&](const char *msg) { throw std::invalid_argument(msg); };
  if (n < 4) fail("grap_params: the filter network description is missing");
  const int L = (int)q[0], act = (int)q[1], resnet = (int)q[2], modifier = (int)q[3];
  if (L < 2 || L > kNetMaxLayers) fail("GRAP/nn: 2..8 dense layers (incl. the output layer)");
  if (act < 0 || act > TA_ACT_ELU) fail("GRAP/nn: unknown activation");
  if (modifier < 0 || modifier > 2) fail("GRAP/nn: unknown H(r) modifier");
  if (n < 4 + L + 1) fail("grap_params: the filter network's layer sizes are missing");
  int sizes[kNetMaxLayers + 1];
  size_t need = 4 + L + 1;
  for (int l = 0; l <= L; ++l) {
    sizes[l] = (int)q[4 + l];
    if (sizes[l] < 1 || sizes[l] > kNetMaxWidth) fail("GRAP/nn: layer widths 1..64");
    if (l > 0) need += (size_t)sizes[l - 1] * sizes[l] + sizes[l];
  }
  if (sizes[0] != 1 || sizes[L] != K) fail("GRAP/nn: the network maps 1 input to K filters");
  const int nel = g->p.nel;
  if (modifier) need += (size_t)nel;  // covalent radii of the elements close the block
  if ((size_t)n != need) fail("grap_params: wrong length for the filter network");
  GrapNet &net = g->net;
  std::memset(&net, 0, sizeof(net));
  net.L = L;
  net.act = act;
  net.resnet = resnet ? 1 : 0;
  net.modifier = modifier;
  for (int e = 0; e < kMaxElements; ++e) net.inv_rcov[e] = 1.0;
  if (modifier)
    for (int e = 0; e < nel; ++e) {
      const double rc = q[n - nel + e];
      if (!(rc > 0.0)) fail("GRAP/nn: covalent radius missing for an element");
      net.inv_rcov[e] = 1.0 / rc;
    }
  const double *src = q + 4 + L + 1;
  int maxw = 16;
  for (int l = 0; l < L; ++l) {
    const int k = sizes[l], nn = sizes[l + 1];
    const int kp = l == 0 ? 1 : (k + 15) / 16 * 16, np = (nn + 15) / 16 * 16;
    net.np[l] = np;
    net.res[l] = (resnet && l > 0 && l < L - 1 && k == nn) ? 1 : 0;
    maxw = std::max(maxw, np);
    std::vector<double> w((size_t)kp * np, 0.0), bb(np, 0.0);
    for (int a = 0; a < k; ++a)
      for (int c = 0; c < nn; ++c) w[(size_t)a * np + c] = src[(size_t)a * nn + c];
    src += (size_t)k * nn;
    for (int c = 0; c < nn; ++c) bb[c] = src[c];
    src += nn;
    double *dw = nullptr, *db = nullptr;
    if (hipMalloc((void **)&dw, w.size() * sizeof(double)) != hipSuccess) throw std::bad_alloc();
    g->owned.push_back(dw);
    if (hipMalloc((void **)&db, bb.size() * sizeof(double)) != hipSuccess) throw std::bad_alloc();
    g->owned.push_back(db);
    if (hipMemcpy(dw, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(db, bb.data(), bb.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      throw std::runtime_error("hipMemcpy of the GRAP filter network failed");
    net.w[l] = dw;
    net.b[l] = db;
  }
  net.xs = maxw + 2;
  size_t doubles = 2 * (size_t)net.np[0];
  for (int l = 1; l < L; ++l) doubles += (size_t)net.np[l - 1] * net_wstride(net.np[l]) + net.np[l];
  doubles += 4 * 2 * (size_t)kMlpTileRows * net.xs;
  g->net_lds = doubles * sizeof(double);
  if (g->net_lds > 150 * 1024) fail("GRAP/nn: the filter network does not fit the LDS image");
  (void)err;
}

}  // namespace

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
  if (algo < 0 || algo > GRAP_NN) {
    err = "unknown GRAP algorithm";
    return nullptr;
  }
  if (algo == GRAP_NN && legacy) {
    err = "GRAP/nn exists only in the non-legacy formulation (grap.py:620-643)";
    return nullptr;
  }
  if (K < 1 || K > kMaxFilters) {
    err = "GRAP supports 1.." + std::to_string(kMaxFilters) + " radial filters";
    return nullptr;
  }
  if (mm < 0 || mm > 5) {
    err = "The maximum angular moment should be <= 5";  // grap.py:580-581
    return nullptr;
  }
  if (legacy && mm > 2) {
    err = "GRAP legacy mode has moments 0, 1, 2 only (grap.py:423-460)";
    return nullptr;
  }
  if (algo != GRAP_NN && m->n_grap_params != 6 + 3 * K) {
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
  const int nd_of[6] = {1, 4, 10, 20, 35, 56};
  p.nd = nd_of[mm];
  p.legacy = legacy ? 1 : 0;
  p.algo = algo;
  p.cutoff = m->cutoff_function;
  p.rcut = m->rcut;
  p.inv_rc2 = 1.0 / (m->rcut * m->rcut);
  int col = 0;
  for (int k = 0; k < kMaxMom; ++k) {
    const bool on = legacy ? ((mask >> k) & 1) && k <= 2 : k <= mm;
    p.col_of_m[k] = (on && k <= mm) ? col++ : -1;
  }
  p.nf = col;
  if (col == 0) {
    delete g;
    err = "GRAP needs at least one moment tensor";
    return nullptr;
  }
  // Packed components in the reference's order (grap.py:501-511, continued for the ranks 4 and 5):
  // degree after degree, nx descending, then ny. Multiplicity tensor: grap.py:470-492 for moments up
  // to 3 (with the traceless corrections of the `symmetric` variant); for max_moment > 3 the
  // reference sums the FULL 3^m tensors with unit weights (get_moment_tensor / get_T_dm, :538-600),
  // i.e. every packed component counts with its multinomial coefficient, at every rank, and
  // `symmetric` plays no role.
  std::vector<int> ex, ey, ez;
  for (int deg = 0; deg <= mm; ++deg)
    for (int nx = deg; nx >= 0; --nx)
      for (int ny = deg - nx; ny >= 0; --ny) {
        ex.push_back(nx);
        ey.push_back(ny);
        ez.push_back(deg - nx - ny);
      }
  auto find = [&](int a, int b2, int c) {
    if (a < 0 || b2 < 0 || c < 0) return 0;
    for (int d = 0; d < p.nd; ++d)
      if (ex[d] == a && ey[d] == b2 && ez[d] == c) return d;
    return 0;
  };
  auto fact = [](int n) { double f = 1.0; for (int k = 2; k <= n; ++k) f *= k; return f; };
  std::vector<double> T((size_t)p.nd * kMaxMom, 0.0);
  std::vector<unsigned long long> cw(p.nd, 0ull);
  const bool sym = symmetric && !legacy && mm <= 3;
  for (int d = 0; d < p.nd; ++d) {
    const int deg = ex[d] + ey[d] + ez[d];
    T[(size_t)d * kMaxMom + deg] = fact(deg) / (fact(ex[d]) * fact(ey[d]) * fact(ez[d]));
    cw[d] = (unsigned long long)d | ((unsigned long long)find(ex[d] - 1, ey[d], ez[d]) << 6) |
            ((unsigned long long)find(ex[d], ey[d] - 1, ez[d]) << 12) |
            ((unsigned long long)find(ex[d], ey[d], ez[d] - 1) << 18) | ((unsigned long long)ex[d] << 24) |
            ((unsigned long long)ey[d] << 27) | ((unsigned long long)ez[d] << 30);
  }
  if (sym && mm >= 2) T[0 * kMaxMom + 2] = -1.0 / 3.0;
  if (sym && mm >= 3)
    for (int d = 1; d < 4; ++d) T[(size_t)d * kMaxMom + 3] = -3.0 / 5.0;
  if (hipMalloc((void **)&g->T_dev, T.size() * sizeof(double)) != hipSuccess ||
      hipMemcpy(g->T_dev, T.data(), T.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc((void **)&g->cw_dev, cw.size() * sizeof(unsigned long long)) != hipSuccess ||
      hipMemcpy(g->cw_dev, cw.data(), cw.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) {
    grap_destroy(g);
    err = "device allocation failed";
    return nullptr;
  }
  p.T = g->T_dev;
  p.cw = g->cw_dev;
  p.Ks = (K + 15) / 16 * 16;
  if (algo == GRAP_NN) {
    try {
      build_filter_net(g, q + 6, m->n_grap_params - 6, K, err);
    } catch (const std::exception &ex) {
      err = ex.what();
      grap_destroy(g);
      return nullptr;
    }
  }
  double host[kMaxFilters * 4] = {0};
  for (int k = 0; k < K && algo != GRAP_NN; ++k)
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
  // constants in the form filter_fn wants
  for (int k = 0; k < K; ++k) {
    double *f = &host[4 * k];
    if (algo == GRAP_SF) {
      f[0] = f[0] * p.inv_rc2;
    } else if (algo == GRAP_DENSITY) {
      const double A = f[0], beta = f[1], re = f[2];
      f[0] = A * std::exp(beta);
      f[1] = beta / re;
      f[2] = 0.0;
    } else if (algo == GRAP_PEXP) {
      const double rl = f[0], pl = f[1];
      f[0] = pl;
      f[1] = std::log(rl);
      f[2] = 0.0;
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
bool grap_uses_filter_net(const GrapModel *g) { return g->p.algo == GRAP_NN; }

void grap_destroy(GrapModel *g) {
  if (!g) return;
  if (g->fp) (void)hipFree(g->fp);
  if (g->T_dev) (void)hipFree(g->T_dev);
  if (g->cw_dev) (void)hipFree(g->cw_dev);
  if (g->Pbuf) (void)hipFree(g->Pbuf);
  if (g->Hbuf) (void)hipFree(g->Hbuf);
  for (double *d : g->owned) (void)hipFree(d);
  delete g;
}

void grap_ensure(GrapModel *g, const DeviceBatch &b) {
  const size_t n = (size_t)b.n_atoms;
  if (g->p.algo == GRAP_NN && (size_t)b.n_pairs > g->cap_pairs) {
    if (g->Hbuf) (void)hipFree(g->Hbuf);
    g->Hbuf = nullptr;
    g->cap_pairs = 0;
    // the forward kernel's zero-weighted padding rows read up to one chunk past the last pair
    const size_t cap = (size_t)b.n_pairs + (size_t)b.n_pairs / 8 + 2 * kFwdChunk;
    if (hipMalloc((void **)&g->Hbuf, cap * 2 * g->p.Ks * sizeof(double)) != hipSuccess) throw std::bad_alloc();
    if (hipMemset(g->Hbuf, 0, cap * 2 * g->p.Ks * sizeof(double)) != hipSuccess)
      throw std::runtime_error("hipMemset failed");
    g->cap_pairs = cap - 2 * kFwdChunk;
    g->p.Hbuf = g->Hbuf;
  }
  if (n <= g->cap_atoms) return;
  if (g->Pbuf) (void)hipFree(g->Pbuf);
  g->Pbuf = nullptr;
  const size_t cap = n + n / 8 + 64;
  const size_t per = (size_t)g->p.nel * g->p.K * g->p.nd * sizeof(double);
  if (hipMalloc((void **)&g->Pbuf, cap * per) != hipSuccess) throw std::bad_alloc();
  g->cap_atoms = cap;
}

void launch_grap_forward(GrapModel *g, const DeviceBatch &b, double eps, hipStream_t s) {
  if (b.n_atoms == 0) return;
  if (g->p.algo == GRAP_NN && b.n_pairs > 0) {
    // geometry first (the filter network needs r), then the K filters of every pair
    SFParams sf;
    std::memset(&sf, 0, sizeof(sf));
    sf.n_elements = g->p.nel;
    sf.eps = eps;
    launch_pair_geometry(sf, b, s);
    const unsigned tiles = (unsigned)((b.n_pairs + kMlpTileRows - 1) / kMlpTileRows);
    const dim3 grid(std::min((tiles + 3) / 4, 2048u));
    int ntmax = 1;
    for (int l = 1; l < g->net.L; ++l) ntmax = std::max(ntmax, g->net.np[l] / 16);
#define TA_GRAP_NET(ACT, NT)                                                                                \
  hipLaunchKernelGGL((grap_nn_filter_kernel<ACT, NT>), grid, dim3(kBlock), g->net_lds, s, g->net, b, g->Hbuf, \
                     g->p.Ks)
    if (g->net.act == TA_ACT_SOFTPLUS) {
      if (ntmax <= 2) TA_GRAP_NET(TA_ACT_SOFTPLUS, 2);
      else TA_GRAP_NET(TA_ACT_SOFTPLUS, 4);
    } else {
      if (ntmax <= 2) TA_GRAP_NET(-1, 2);
      else TA_GRAP_NET(-1, 4);
    }
#undef TA_GRAP_NET
  }
  if (g->p.nd <= kSmallComp)
    hipLaunchKernelGGL(grap_forward_kernel<kSmallComp>, dim3((unsigned)b.n_atoms), dim3(kWave), 0, s, g->p, b,
                       g->Pbuf, g->ndim, eps);
  else
    hipLaunchKernelGGL(grap_forward_kernel<kMaxComp>, dim3((unsigned)b.n_atoms), dim3(kWave), 0, s, g->p, b,
                       g->Pbuf, g->ndim, eps);
}

bool grap_hvp_supported(const GrapModel *g) { return g->p.algo != GRAP_NN || g->p.K <= kMaxFilters; }

// g and its tangent for one direction (ta_hessian_vectors): Dv / Dd = the pair vectors and their tangents
// [P][4], wdot = H_mlp G-dot [N][ndim]
void launch_grap_hvp(GrapModel *g, const DeviceBatch &b, double eps, const double *Dv, const double *Dd,
                     const double *wdot, double *gv, double *gd, hipStream_t s) {
  if (b.n_atoms == 0) return;
  hipLaunchKernelGGL(grap_hvp_kernel, dim3((unsigned)b.n_atoms), dim3(kWave), 0, s, g->p, g->net, b, g->ndim, eps, Dv,
                     Dd, wdot, gv, gd);
}

void launch_grap_backward(GrapModel *g, const DeviceBatch &b, hipStream_t s) {
  if (b.n_atoms == 0) return;
  const int Kp = (g->p.K + 3) & ~3;
  const size_t lds = ((size_t)Kp * g->p.nd + 4 * (size_t)g->p.K + (size_t)g->ndim) * sizeof(double);
  if (g->p.nd <= kSmallComp)
    hipLaunchKernelGGL(grap_backward_kernel<kSmallComp>, dim3((unsigned)b.n_atoms), dim3(kWave), lds, s, g->p, b,
                       g->Pbuf, g->ndim);
  else
    hipLaunchKernelGGL(grap_backward_kernel<kMaxComp>, dim3((unsigned)b.n_atoms), dim3(kWave), lds, s, g->p, b,
                       g->Pbuf, g->ndim);
}

}  // namespace ta
