// Hand-written gfx950 kernels for the symmetry-function + MLP hot path.
//
// What each kernel replaces in the reference (TensorFlow-1 graph ops):
//   pair_geometry_kernel      calculate_rij            transformer/universal.py:448-474
//   g4_forward_kernel         build_angular_graph + _apply_g4_functions
//                                                       universal.py:622-694, nn/atomic/sf.py:121-182
//   descriptor_reduce_kernel  build_radial_graph + _apply_g2_functions + concat
//                                                       universal.py:583-620, sf.py:79-119, :184-215
//   backward_kernel           tf.gradients(E, positions/cell) through all of the above
//                                                       nn/basic.py:277-331
//   force_gather_kernel       forces = -dE/dR ; virial  basic.py:277-331
//   frame_reduce_kernel       energy = sum(atomic)      nn/atomic/atomic.py:289-302
//
// Layout. Directed pairs are sorted by centre atom (and by neighbour species
// inside a centre). Each pair owns one 64-byte record {Dx,Dy,Dz,r^2,1/r,H0,H1,H2}
// with H_k = exp(-beta_k r^2/acut^2) fc(r; acut). The angular kernels give one
// lane to one directed pair (i, a); a 256-lane workgroup stages the records of
// the 3-5 centres it touches in LDS and every lane walks the other neighbours b
// of its centre, reading their records as LDS broadcasts. Triples are never
// materialised in HBM: r_jk comes from |D_b - D_a|^2 and the cosine cutoff is a
// polynomial in r_jk^2, so a triple costs no sqrt and no trig. Lane a sums the
// ordered pairs (a, b), b != a: every unordered triple is visited from both of
// its neighbours, which gives each lane the complete dE/dD_a in registers
// (no atomics, no cross-lane traffic); the descriptor sum takes half of it.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdlib>
#include <stdexcept>

#include "ta_device.h"
#include "ta_math.h"
#include "ta_reduce.h"

namespace ta {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int radial_term(int center, int other) {
  // [AA, AB (B != A, sorted)]  (reference utils.py:265-273)
  return other == center ? 0 : (other < center ? other + 1 : other);
}
__device__ __forceinline__ int angular_term(int s1, int s2, int nel) {
  // sorted pair (j <= k) in row-major upper-triangular order (utils.py:274-282)
  int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
  return a * nel - (a * (a - 1)) / 2 + (b - a);
}

// --------------------------------------------------------------------------
// K1: pair geometry
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pair_geometry_kernel(SFParams sf, DeviceBatch b) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n_pairs) return;
  const int i = b.pair_i[p], j = b.pair_j[p];
  const double *h = b.cells + 9 * (size_t)b.frame_of_atom[i];
  const double sx = (double)b.pair_shift[3 * p], sy = (double)b.pair_shift[3 * p + 1],
               sz = (double)b.pair_shift[3 * p + 2];
  const double *ri = b.pos + 3 * (size_t)i, *rj = b.pos + 3 * (size_t)j;
  // D = Rj - Ri + S.h  (universal.py:463-468)
  const double dx = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
  const double dy = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
  const double dz = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
  const double r2 = dx * dx + dy * dy + dz * dz + sf.eps;  // universal.py:470-472
  const double r = sqrt(r2);
  double rec[kRecDoubles];
  rec[0] = dx;
  rec[1] = dy;
  rec[2] = dz;
  rec[3] = r2;
  rec[4] = 1.0 / r;
  rec[5] = rec[6] = rec[7] = 0.0;
  if (sf.angular) {
    const double u = r2 * sf.inv_ac2;
    if (u < 1.0) {
      const double f = cutoff_u_value(sf.cutoff, u);
      for (int k = 0; k < sf.n_beta; ++k) rec[5 + k] = ta_exp(-sf.beta[k] * u) * f;
    }
  }
  double2 *dst = reinterpret_cast<double2 *>(b.rec + kRecDoubles * (size_t)p);
  dst[0] = make_double2(rec[0], rec[1]);
  dst[1] = make_double2(rec[2], rec[3]);
  dst[2] = make_double2(rec[4], rec[5]);
  dst[3] = make_double2(rec[6], rec[7]);
}

// stage the pair records of every centre touched by this workgroup into LDS
__device__ __forceinline__ int stage_records(const DeviceBatch &b, double *lds, int64_t p0) {
  const int64_t plast = (p0 + kBlock - 1 < b.n_pairs) ? p0 + kBlock - 1 : b.n_pairs - 1;
  const int i_lo = b.pair_i[p0], i_hi = b.pair_i[plast];
  const int s0 = b.pair_start[i_lo], s1 = b.pair_start[i_hi + 1];
  const double2 *src = reinterpret_cast<const double2 *>(b.rec + kRecDoubles * (size_t)s0);
  double2 *dst = reinterpret_cast<double2 *>(lds);
  const int n16 = (s1 - s0) * (kRecDoubles / 2);
  for (int k = threadIdx.x; k < n16; k += kBlock) dst[k] = src[k];
  __syncthreads();
  return s0;
}

// --------------------------------------------------------------------------
// K2: angular descriptors, forward. part4[(sb*n_ang + c)][p] = sum over
// neighbours b of species sb of centre(p) of the G4 summand of channel c for
// the ordered pair (a = p, b).
// --------------------------------------------------------------------------
template <int NB, int NG, int NZ>
__global__ __launch_bounds__(kBlock) void g4_forward_kernel(SFParams sf, AngChunk ch, DeviceBatch b) {
  extern __shared__ double lds[];
  const int64_t p0 = (int64_t)blockIdx.x * kBlock;
  const int s0 = stage_records(b, lds, p0);
  const int64_t p = p0 + threadIdx.x;
  if (p >= b.n_pairs) return;

  const double *ra = lds + kRecDoubles * (size_t)(p - s0);
  const double ax = ra[0], ay = ra[1], az = ra[2], ra2 = ra[3], inv_ra = ra[4];
  double Ha[NB];
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) Ha[ib] = ra[5 + ch.hslot[ib]];
  const int i = b.pair_i[p];
  const int nel = sf.n_elements;
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  const int lp = (int)(p - s0);

  for (int sb = 0; sb < nel; ++sb) {
    const int q0 = seg[sb] - s0, q1 = seg[sb + 1] - s0;
    double acc[NB][NG][NZ];
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) acc[ib][ig][iz] = 0.0;

    for (int q = q0; q < q1; ++q) {
      const double *rb = lds + kRecDoubles * (size_t)q;
      const double ex = rb[0] - ax, ey = rb[1] - ay, ez = rb[2] - az;
      const double rb2 = rb[3], inv_rb = rb[4];
      // r_jk^2 = |D_ik - D_ij|^2 + eps  (universal.py:213, :470-472)
      const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
      const double u = d2 * sf.inv_ac2;
      const bool ok = (q != lp) && (u < 1.0);
      // cos(theta) = (rij^2 + rik^2 - rjk^2) / (2 rij rik)  (sf.py:145-148)
      const double cth = (ra2 + rb2 - d2) * 0.5 * inv_ra * inv_rb;
      const double fd = cutoff_u_value(sf.cutoff, u);
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) {
        const double ed = ta_exp(-ch.beta[ib] * u);
        double common = Ha[ib] * rb[5 + ch.hslot[ib]] * ed * fd;
        common = ok ? common : 0.0;
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) {
          const double base = fma(ch.gamma[ig], cth, 1.0);
#pragma unroll
          for (int iz = 0; iz < NZ; ++iz) {
            double pw;
            if (ch.zeta_int[iz] > 0)
              pw = pow_int_m1(base, ch.zeta_int[iz]) * base;
            else
              pw = safe_pow_value(ch.safe_pow, base, ch.zeta[iz]);
            acc[ib][ig][iz] = fma(pw, common, acc[ib][ig][iz]);
          }
        }
      }
    }
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) {
          const int c = ch.chan[(ib * NG + ig) * NZ + iz];
          b.part4[(size_t)(sb * sf.n_ang + c) * b.n_pairs + p] = acc[ib][ig][iz] * ch.kz[iz];
        }
  }
}

// --------------------------------------------------------------------------
// K3a: per-atom descriptors: G2 straight from the pair records, G4 from the
// per-pair partial sums. One wavefront per atom, shuffle reduction.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void descriptor_reduce_kernel(SFParams sf, DeviceBatch b) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  double *Gi = b.G + (size_t)i * sf.ndim;
  atom_descriptors<64>(sf, b, i, lane, true, [&](int c, double v) { Gi[c] = v; });
}

// --------------------------------------------------------------------------
// K4: backward. g[p] = dE/dD_p for the directed pair p = (i, a):
//   G2:  s_p D_a / r_a,  s_p = sum_c dE/dG_c d g_c / d r
//   G4:  D_a sum_b (A_ab + Q_ab) - sum_b Q_ab D_b   
// --------------------------------------------------------------------------
__device__ __forceinline__ void radial_backward(const SFParams &sf, const DeviceBatch &b, int i,
                                                int sa, double ra2, double inv_ra, double &s) {
  s = 0.0;
  const double u = ra2 * sf.inv_rc2;
  if (u < 1.0) {
    double f, dfdu;
    cutoff_u(sf.cutoff, u, f, dfdu);
    const double r = sqrt(ra2);
    const double dfdr = dfdu * 2.0 * r * sf.inv_rc2;
    const int tr = radial_term(b.species[i], sa);
    const double *w = b.dEdG + (size_t)i * sf.ndim + tr * sf.n_rad;
    for (int c = 0; c < sf.n_rad; ++c) {
      const double dr = r - sf.omega[c];
      const double e = ta_exp(-sf.eta[c] * dr * dr * sf.inv_rc2);
      s = fma(w[c], e * (dfdr - 2.0 * sf.eta[c] * dr * f * sf.inv_rc2), s);
    }
  }
}

__global__ __launch_bounds__(kBlock) void g2_backward_kernel(SFParams sf, DeviceBatch b) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n_pairs) return;
  const double *ra = b.rec + kRecDoubles * (size_t)p;
  const int i = b.pair_i[p];
  const int sa = b.species[b.pair_j[p]];
  double s;
  radial_backward(sf, b, i, sa, ra[3], ra[4], s);
  s *= ra[4];
  b.g[4 * (size_t)p] = s * ra[0];
  b.g[4 * (size_t)p + 1] = s * ra[1];
  b.g[4 * (size_t)p + 2] = s * ra[2];
}

template <int NB, int NG, int NZ>
__global__ __launch_bounds__(kBlock) void backward_kernel(SFParams sf, AngChunk ch, DeviceBatch b,
                                                          int first) {
  extern __shared__ double lds[];
  const int64_t p0 = (int64_t)blockIdx.x * kBlock;
  const int s0 = stage_records(b, lds, p0);
  const int64_t p = p0 + threadIdx.x;
  if (p >= b.n_pairs) return;

  const double *ra = lds + kRecDoubles * (size_t)(p - s0);
  const double ax = ra[0], ay = ra[1], az = ra[2], ra2 = ra[3], inv_ra = ra[4];
  const int i = b.pair_i[p];
  const int sa = b.species[b.pair_j[p]];
  const int nel = sf.n_elements;
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  const int lp = (int)(p - s0);

  // own-pair factors: H_a = exp(-beta u_a) fc(u_a),  G_a = (dH_a/dr_a) / r_a
  double Ha[NB], Ga[NB];
  {
    const double ua = ra2 * sf.inv_ac2;
    double fa = 0.0, dfa = 0.0;
    const bool oka = ua < 1.0;
    if (oka) cutoff_u(sf.cutoff, ua, fa, dfa);
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
      const double ea = oka ? ta_exp(-ch.beta[ib] * ua) : 0.0;
      Ha[ib] = ea * fa;
      Ga[ib] = ea * 2.0 * sf.inv_ac2 * (dfa - ch.beta[ib] * fa);
    }
  }

  double sumAQ = 0.0, vx = 0.0, vy = 0.0, vz = 0.0;
  const double inv_ra2 = inv_ra * inv_ra;
  for (int sb = 0; sb < nel; ++sb) {
    const int q0 = seg[sb] - s0, q1 = seg[sb + 1] - s0;
    // dE/dG of the channels of this (a-species, b-species) term, times 2^(1-zeta)
    double w[NB][NG][NZ];
    {
      const double *wsrc = b.dEdG + (size_t)i * sf.ndim + sf.n_radial_dim +
                           angular_term(sa, sb, nel) * sf.n_ang;
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
#pragma unroll
          for (int iz = 0; iz < NZ; ++iz)
            w[ib][ig][iz] = wsrc[ch.chan[(ib * NG + ig) * NZ + iz]] * ch.kz[iz];
    }
    for (int q = q0; q < q1; ++q) {
      const double *rb = lds + kRecDoubles * (size_t)q;
      const double bx = rb[0], by = rb[1], bz = rb[2];
      const double ex = bx - ax, ey = by - ay, ez = bz - az;
      const double rb2 = rb[3], inv_rb = rb[4];
      const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
      const double u = d2 * sf.inv_ac2;
      const bool ok = (q != lp) && (u < 1.0);
      const double inv_ab = inv_ra * inv_rb;
      const double cth = (ra2 + rb2 - d2) * 0.5 * inv_ab;
      double fd, dfd;
      cutoff_u(sf.cutoff, u, fd, dfd);
      double A = 0.0, Q = 0.0;
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) {
        const double ed = ta_exp(-ch.beta[ib] * u);
        const double Hb = rb[5 + ch.hslot[ib]];
        const double Hd = ed * fd;
        const double Hd2 = 2.0 * sf.inv_ac2 * ed * (dfd - ch.beta[ib] * fd);
        double S0 = 0.0, S1 = 0.0;
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) {
          const double base = fma(ch.gamma[ig], cth, 1.0);
#pragma unroll
          for (int iz = 0; iz < NZ; ++iz) {
            double pm1;
            if (ch.zeta_int[iz] > 0)
              pm1 = pow_int_m1(base, ch.zeta_int[iz]);
            else
              pm1 = safe_pow_grad(ch.safe_pow, base, ch.zeta[iz] - 1.0);
            S0 = fma(w[ib][ig][iz], pm1 * base, S0);
            S1 = fma(w[ib][ig][iz] * ch.zeta[iz] * ch.gamma[ig], pm1, S1);
          }
        }
        A = fma(Hb * Hd, fma(S1 * Ha[ib], inv_ab - cth * inv_ra2, S0 * Ga[ib]), A);
        Q = fma(Ha[ib] * Hb, fma(-S1 * inv_ab, Hd, S0 * Hd2), Q);
      }
      A = ok ? A : 0.0;
      Q = ok ? Q : 0.0;
      sumAQ += A + Q;
      vx = fma(Q, bx, vx);
      vy = fma(Q, by, vy);
      vz = fma(Q, bz, vz);
    }
  }
  double gx = fma(ax, sumAQ, -vx), gy = fma(ay, sumAQ, -vy), gz = fma(az, sumAQ, -vz);
  if (first) {
    double s;
    radial_backward(sf, b, i, sa, ra2, inv_ra, s);
    s *= inv_ra;
    gx = fma(s, ax, gx);
    gy = fma(s, ay, gy);
    gz = fma(s, az, gz);
  } else {
    gx += b.g[4 * (size_t)p];
    gy += b.g[4 * (size_t)p + 1];
    gz += b.g[4 * (size_t)p + 2];
  }
  b.g[4 * (size_t)p] = gx;
  b.g[4 * (size_t)p + 1] = gy;
  b.g[4 * (size_t)p + 2] = gz;
}

// --------------------------------------------------------------------------
// K5: forces and per-atom virial. Full list => the reaction force on j of the
// pair (i -> j) is read from its reverse pair (j -> i): no atomics, fixed order.
//   F_i = sum_{p in N(i)} (g[p] - g[rev p])        (F = -dE/dR, basic.py:281-287)
//   W_i = sum_{p in N(i)} g[p] (x) D[p]            (== -F^T R + (dE/dh)^T h, basic.py:306-316)
// --------------------------------------------------------------------------
// W lanes per atom, 16 atoms per workgroup (one `bpart` record): 16 (one DPP row: the 12 sums are VALU row
// rotations, no LDS shuffles) for batches; 32 for a single small frame, whose ~1000 wavefronts would
// otherwise walk 90 pairs in two dependent batches of 4 x 16 (one batch of 4 x 32 instead).
template <int W>
__global__ __launch_bounds__(16 * W) void force_gather_kernel(DeviceBatch b) {
  const int64_t i = ((int64_t)blockIdx.x * (16 * W) + threadIdx.x) / W;
  const int lane = threadIdx.x & (W - 1);
  const bool active = i < b.n_atoms;
  double f[3] = {0, 0, 0}, w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int q0 = active ? b.pair_start[i] : 0, q1 = active ? pair_stop_of(b, i) : 0;
  if (b.own_sums) {
    // the backward kernel left sum_p g[p] and the virial rows per atom: only g[rev p] is gathered
    // (36 bytes per pair instead of 100)
    for (int qb = q0 + lane; qb < q1; qb += 4 * W) {
      int r[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = qb + W * k;
        ok[k] = q < q1;
        r[k] = ok[k] ? pair_rev_of(b, q) : 0;
      }
      double gr[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double2 *grp = reinterpret_cast<const double2 *>(b.g + 4 * (size_t)r[k]);
        const double2 c0 = grp[0], c1 = grp[1];
        gr[k][0] = c0.x;
        gr[k][1] = c0.y;
        gr[k][2] = c1.x;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[k]) {
          f[0] -= gr[k][0];
          f[1] -= gr[k][1];
          f[2] -= gr[k][2];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) f[k] = group_sum<W>(f[k]);
    if (lane == 0 && active) {
      const double *own = b.fown + 12 * (size_t)i;
      for (int k = 0; k < 3; ++k) b.forces[3 * (size_t)i + k] = f[k] + own[k];
      for (int k = 0; k < 9; ++k) w[k] = own[3 + k];
    }
    block_partials(b, blockIdx.x, i, active, lane == 0, w);
    return;
  }
  // batches of 4 strided pairs: the 4 reverse indices, then all 4 x 9 operands, are in flight
  // together, so a batch costs two memory latencies instead of eight
  for (int qb = q0 + lane; qb < q1; qb += 4 * W) {
    int r[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = qb + W * k;
      ok[k] = q < q1;
      r[k] = ok[k] ? pair_rev_of(b, q) : 0;
    }
    double gq[4][3], gr[4][3], d[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = ok[k] ? qb + W * k : q0;
      const double2 *rec = pair_geom(b, (size_t)q);
      const double2 v0 = rec[0], v1 = rec[1];
      d[k][0] = v0.x;
      d[k][1] = v0.y;
      d[k][2] = v1.x;
      // g is stored as [P][4] (x, y, z, pad): one 32-byte sector per reverse-pair access
      const double2 *gqp = reinterpret_cast<const double2 *>(b.g + 4 * (size_t)q);
      const double2 *grp = reinterpret_cast<const double2 *>(b.g + 4 * (size_t)r[k]);
      const double2 a0 = gqp[0], a1 = gqp[1], c0 = grp[0], c1 = grp[1];
      gq[k][0] = a0.x;
      gq[k][1] = a0.y;
      gq[k][2] = a1.x;
      gr[k][0] = c0.x;
      gr[k][1] = c0.y;
      gr[k][2] = c1.x;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!ok[k]) continue;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        f[c] += gq[k][c] - gr[k][c];
#pragma unroll
        for (int e = 0; e < 3; ++e) w[3 * c + e] = fma(gq[k][c], d[k][e], w[3 * c + e]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) f[k] = group_sum<W>(f[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = group_sum<W>(w[k]);
  if (lane == 0 && active) {
    for (int k = 0; k < 3; ++k) b.forces[3 * (size_t)i + k] = f[k];
  }
  block_partials(b, blockIdx.x, i, active, lane == 0, w);
}

// --------------------------------------------------------------------------
// K6: per-frame energy and virial (fixed summation order), then the batch energy.
// One 1024-lane workgroup per frame: every lane strides over the frame's atoms
// with 10 running sums, then shuffle + LDS reduction.
// --------------------------------------------------------------------------
constexpr int kRedBlock = 1024;

// `use_partials`: force_gather / the EAM force kernel ran and left the 16-atom group records
// `mirror` (MD step, ta_step): a page-locked host image of the packed results [energy F | virial 9F | atomic N |
// forces 3N]. The frame sums are written to both places, and workgroups beyond the frames copy the
// first `n_tail` per-atom values (finished by the kernels before this one) across, so the step needs no
// download launch behind this one.
__global__ __launch_bounds__(kRedBlock) void frame_reduce_kernel(DeviceBatch b, int want_virial, int use_partials,
                                                                 double *mirror, long long n_tail) {
  if ((int)blockIdx.x >= b.n_frames) {
    const double *src = b.energy + 10 * (size_t)b.n_frames;
    double *dst = mirror + 10 * (size_t)b.n_frames;
    const long long stride = (long long)(gridDim.x - b.n_frames) * kRedBlock;
    for (long long k = (long long)((int)blockIdx.x - b.n_frames) * kRedBlock + threadIdx.x; k < n_tail; k += stride)
      dst[k] = src[k];
    return;
  }
  __shared__ double red[10][kRedBlock / 64];
  const int f = blockIdx.x;
  const int a0 = b.atom_start[f], a1 = b.atom_start[f + 1];
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  if (use_partials) {
    // whole groups of the frame: [g0, g1); the atoms before 16 g0 and from 16 g1 on sit in groups
    // shared with a neighbouring frame and come one by one
    // (the last group of the batch may hold fewer than 16 atoms and still be whole)
    const int g0 = (a0 + 15) / 16, g1 = max(g0, (a1 == (int)b.n_atoms) ? (a1 + 15) / 16 : a1 / 16);
    for (int g = g0 + (int)threadIdx.x; g < g1; g += kRedBlock) {
      const double *p = b.bpart + 10 * (size_t)g;
#pragma unroll
      for (int k = 0; k < 10; ++k) acc[k] += p[k];
    }
    const int e0 = min(a1, 16 * g0), e1 = min(a1, max(e0, 16 * g1));  // edge atoms: [a0, e0) and [e1, a1)
    const int n_edge = (e0 - a0) + (a1 - e1);
    if (g1 == g0) {  // no whole group: every atom of the frame is an edge atom
      for (int a = a0 + (int)threadIdx.x; a < a1; a += kRedBlock) {
        acc[0] += b.eatom[a];
        for (int k = 0; k < 9; ++k) acc[1 + k] += b.wat[9 * (size_t)a + k];
      }
    } else if ((int)threadIdx.x < n_edge) {
      const int a = (int)threadIdx.x < e0 - a0 ? a0 + (int)threadIdx.x : e1 + ((int)threadIdx.x - (e0 - a0));
      acc[0] += b.eatom[a];
      for (int k = 0; k < 9; ++k) acc[1 + k] += b.wat[9 * (size_t)a + k];
    }
  } else {
    for (int a = a0 + threadIdx.x; a < a1; a += kRedBlock) acc[0] += b.eatom[a];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    double v = 0.0;
    for (int w = 0; w < kRedBlock / 64; ++w) v += red[threadIdx.x][w];
    if (threadIdx.x == 0) {
      b.energy[f] = v;
      if (mirror) mirror[f] = v;
      if (b.n_frames == 1) b.batch_energy[0] = v;
    } else if (want_virial) {
      b.virial[9 * (size_t)f + (threadIdx.x - 1)] = v;
      if (mirror) mirror[(size_t)b.n_frames + 9 * (size_t)f + (threadIdx.x - 1)] = v;
    }
  }
}

__global__ __launch_bounds__(kBlock) void batch_energy_kernel(DeviceBatch b) {
  __shared__ double red[kBlock / 64];
  double acc = 0.0;
  for (int f = threadIdx.x; f < b.n_frames; f += kBlock) acc += b.energy[f];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) v += red[w];
    b.batch_energy[0] = v;
  }
}

inline unsigned blocks_for(int64_t n, int per_block) {
  return (unsigned)((n + per_block - 1) / per_block);
}

template <int NB, int NG, int NZ>
void launch_fwd_t(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b, hipStream_t s) {
  hipLaunchKernelGGL((g4_forward_kernel<NB, NG, NZ>), dim3(blocks_for(b.n_pairs, kBlock)),
                     dim3(kBlock), g4_lds_bytes(b.nnl_max), s, sf, ch, b);
}
template <int NB, int NG, int NZ>
void launch_bwd_t(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b, int first,
                  hipStream_t s) {
  hipLaunchKernelGGL((backward_kernel<NB, NG, NZ>), dim3(blocks_for(b.n_pairs, kBlock)),
                     dim3(kBlock), g4_lds_bytes(b.nnl_max), s, sf, ch, b, first);
}

}  // namespace

size_t g4_lds_bytes(int nnl_max) {
  return (size_t)(kBlock + 2 * (size_t)nnl_max) * kRecDoubles * sizeof(double);
}

void launch_pair_geometry(const SFParams &sf, const DeviceBatch &b, hipStream_t s) {
  if (b.n_pairs == 0) return;
  hipLaunchKernelGGL(pair_geometry_kernel, dim3(blocks_for(b.n_pairs, kBlock)), dim3(kBlock), 0, s,
                     sf, b);
}

#define TA_DISPATCH(FN, ...)                                             \
  do {                                                                   \
    const int key = nb * 100 + ng * 10 + nz;                             \
    switch (key) {                                                       \
      case 111: FN<1, 1, 1>(__VA_ARGS__); break;                         \
      case 112: FN<1, 1, 2>(__VA_ARGS__); break;                         \
      case 121: FN<1, 2, 1>(__VA_ARGS__); break;                         \
      case 122: FN<1, 2, 2>(__VA_ARGS__); break;                         \
      case 211: FN<2, 1, 1>(__VA_ARGS__); break;                         \
      case 212: FN<2, 1, 2>(__VA_ARGS__); break;                         \
      case 221: FN<2, 2, 1>(__VA_ARGS__); break;                         \
      case 222: FN<2, 2, 2>(__VA_ARGS__); break;                         \
      default: throw std::domain_error("no angular kernel for this (beta, gamma, zeta) chunk shape"); \
    }                                                                    \
  } while (0)

void launch_g4_forward(const SFParams &sf, const AngChunk &ch, int nb, int ng, int nz,
                       const DeviceBatch &b, hipStream_t s) {
  if (b.n_pairs == 0) return;
  TA_DISPATCH(launch_fwd_t, sf, ch, b, s);
}

void launch_descriptor_reduce(const SFParams &sf, const DeviceBatch &b, hipStream_t s) {
  if (b.n_atoms == 0) return;
  hipLaunchKernelGGL(descriptor_reduce_kernel, dim3(blocks_for(b.n_atoms * 64, kBlock)),
                     dim3(kBlock), 0, s, sf, b);
}

void launch_backward(const SFParams &sf, const AngChunk &ch, int nb, int ng, int nz, bool first,
                     bool radial_only, const DeviceBatch &b, hipStream_t s) {
  if (b.n_pairs == 0) return;
  if (radial_only) {
    hipLaunchKernelGGL(g2_backward_kernel, dim3(blocks_for(b.n_pairs, kBlock)), dim3(kBlock), 0, s,
                       sf, b);
    return;
  }
  const int f = first ? 1 : 0;
  TA_DISPATCH(launch_bwd_t, sf, ch, b, f, s);
}

void launch_force_gather(const SFParams &, const DeviceBatch &b, hipStream_t s) {
  if (b.n_atoms == 0) return;
  static const int w_env = getenv("TA_GATHER_W") ? atoi(getenv("TA_GATHER_W")) : 0;  // A/B switch
  const int W = (w_env == 16 || w_env == 32) ? w_env : (b.n_atoms < 16384 ? 32 : 16);
  const dim3 grid((unsigned)((b.n_atoms + 15) / 16));
  if (W == 32) hipLaunchKernelGGL(force_gather_kernel<32>, grid, dim3(512), 0, s, b);
  else hipLaunchKernelGGL(force_gather_kernel<16>, grid, dim3(256), 0, s, b);
}

void launch_frame_reduce(const DeviceBatch &b, bool want_virial, double *mirror, int64_t n_tail, hipStream_t s) {
  if (b.n_frames == 0) return;
  // `want_virial` = the force path ran: its kernels left the 16-atom group records behind
  const unsigned extra = (mirror && n_tail > 0) ? (unsigned)std::min<int64_t>((n_tail + kRedBlock - 1) / kRedBlock, 32) : 0u;
  hipLaunchKernelGGL(frame_reduce_kernel, dim3((unsigned)b.n_frames + extra), dim3(kRedBlock), 0, s, b,
                     want_virial ? 1 : 0, want_virial ? 1 : 0, mirror, (long long)(mirror ? n_tail : 0));
  if (b.n_frames > 1) hipLaunchKernelGGL(batch_energy_kernel, dim3(1), dim3(kBlock), 0, s, b);
}

}  // namespace ta
