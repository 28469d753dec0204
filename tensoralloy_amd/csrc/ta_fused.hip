// Fused per-centre kernel: pair geometry -> G2/G4 descriptors -> per-atom MLP
// (forward + input gradient, fp64 MFMA) -> dE/dD of every pair, in ONE launch.
//
// Replaces, for one structure batch, the whole chain of reference ops between
// the feed dict and dE/dD: calculate_rij (transformer/universal.py:448-474),
// build_radial_graph / build_angular_graph (:583-694), _apply_g2_functions /
// _apply_g4_functions (nn/atomic/sf.py:79-182), _apply_minmax_normalization
// (nn/atomic/atomic.py:157-195), convolution1x1 (nn/convolutional.py:154-300),
// the energy ops (atomic.py:270-302) and tf.gradients through all of them
// (nn/basic.py:277-331). Forces / virial are then assembled by force_gather.
//
// A workgroup owns a run of WHOLE centre atoms (<= 16 centres, <= cap pairs, one
// lane per directed pair), so everything an atom's energy depends on is inside
// the workgroup:
//   phase 0  stage pair records in LDS (computing the geometry), as ta_kernels_v2.hip
//   phase 1  angular sweep (rotation schedule, fp32 candidate mask kept in registers,
//            Horner Hd(u)), radial terms of the own pair; per-lane partial sums -> LDS
//   phase 2  per-centre reduction of the partial sums -> descriptors G (LDS + HBM)
//   phase 3  MLP tile (ta_mlp_tile.h): rows = the workgroup's centres, per element;
//            atomic energies -> HBM, dE/dG -> LDS table
//   phase 4  backward sweep over the SAME candidate masks (no second scan, no second
//            staging); partner shares through ds_add_f64; epilogue adds the G2 term
// Conditions (else the multi-kernel path runs): one parameter chunk (one beta,
// <= 2 gamma, <= 2 zeta), <= 3 elements, <= 128 neighbours per atom, <= 8 radial
// channels per species, descriptor length <= 64, LDS plan <= 64 KB.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "ta_device.h"
#include "ta_math.h"
#include "ta_mlp_tile.h"

namespace ta {
namespace {

constexpr int kBlock = 256;
constexpr int kNF = 7;
constexpr int kRingPad = 64;
constexpr int kMaxRadFused = 8;
constexpr int kMaxDimFused = 64;

__device__ __forceinline__ int aterm(int s1, int s2, int nel) {
  int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
  return a * nel - (a * (a - 1)) / 2 + (b - a);
}
__device__ __forceinline__ int rterm(int center, int other) {
  return other == center ? 0 : (other < center ? other + 1 : other);
}

struct Lds {
  double *x, *y, *z, *r2, *inv, *H, *G;
  float *xf, *yf, *zf;
  unsigned char *sp, *icl;
  double *gtab, *dtab;
  int *segs;          // [16][NSPEC + 1] item offsets of the species segments of every centre
  double *P;          // [npart][cap]
  double *buf0, *buf1, *gacc;
};

__device__ __forceinline__ Lds carve(double *lds, const FusedPlan &pl, int ndim) {
  Lds f;
  const int cap = pl.cap;
  f.x = lds;
  f.y = f.x + cap;
  f.z = f.y + cap;
  f.r2 = f.z + cap;
  f.inv = f.r2 + cap;
  f.H = f.inv + cap;
  f.G = f.H + cap;
  f.xf = reinterpret_cast<float *>(f.G + cap);
  f.yf = f.xf + (2 * cap + kRingPad);
  f.zf = f.yf + (2 * cap + kRingPad);
  f.sp = reinterpret_cast<unsigned char *>(f.zf + (2 * cap + kRingPad));
  f.icl = f.sp + cap;
  f.gtab = lds + pl.off_tab;
  f.dtab = f.gtab + kMaxCentersPerBlock * ndim;
  f.segs = reinterpret_cast<int *>(f.dtab + kMaxCentersPerBlock * ndim);
  f.P = lds + pl.off_b;
  f.buf1 = lds + pl.off_b;
  f.gacc = lds + pl.off_b;
  f.buf0 = lds + pl.off_buf0;
  return f;
}

template <int HD>
__device__ __forceinline__ void hd_eval(const SFParams &sf, const AngChunk &ch, double beta,
                                        double u, double &h, double &dh) {
  if constexpr (HD > 0) {
    double p = ch.hd[HD - 1], d = 0.0;
#pragma unroll
    for (int k = HD - 2; k >= 0; --k) {
      d = fma(d, u, p);
      p = fma(p, u, ch.hd[k]);
    }
    h = p;
    dh = d;
  } else {
    double fd, dfd;
    cutoff_u(sf.cutoff, u, fd, dfd);
    const double ed = ta_exp(-beta * u);
    h = ed * fd;
    dh = ed * (dfd - beta * fd);
  }
}
template <int HD>
__device__ __forceinline__ double hd_value(const SFParams &sf, const AngChunk &ch, double beta,
                                           double u) {
  if constexpr (HD > 0) {
    double p = ch.hd[HD - 1];
#pragma unroll
    for (int k = HD - 2; k >= 0; --k) p = fma(p, u, ch.hd[k]);
    return p;
  } else {
    return ta_exp(-beta * u) * cutoff_u_value(sf.cutoff, u);
  }
}

// fp32 superset mask of the partners a + s, s = 1 .. smax (<= 64)  (see ta_kernels_v2.hip)
__device__ __forceinline__ unsigned long long partner_mask(const SFParams &sf, const Lds &f,
                                                           int base, int n, int a, int smax) {
  const int ring = 2 * base + a;
  const float ax = f.xf[ring], ay = f.yf[ring], az = f.zf[ring];
  const float lim = (float)(sf.acut * sf.acut) * 1.0001f;
  unsigned long long mask = 0ull;
  for (int g = 0; g < 4 && 16 * g < smax; ++g) {
    const float *px = f.xf + ring + 1 + 16 * g, *py = f.yf + ring + 1 + 16 * g,
                *pz = f.zf + ring + 1 + 16 * g;
    unsigned m = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float ex = px[k] - ax, ey = py[k] - ay, ez = pz[k] - az;
      const float d2 = fmaf(ex, ex, fmaf(ey, ey, ez * ez));
      m |= (d2 < lim) ? (1u << k) : 0u;
    }
    mask |= (unsigned long long)m << (16 * g);
  }
  if (smax < 64) mask &= (1ull << smax) - 1ull;
  const int half = n >> 1;
  if (!(n & 1) && a >= half && half >= 1) mask &= ~(1ull << (half - 1));
  return mask;
}

template <int NSPEC, int NG, int NZ, int HD, bool DEFZ>
__global__ __launch_bounds__(kBlock) void sf_fused_kernel(SFParams sf, AngChunk ch, DeviceBatch b,
                                                          FusedPlan pl, const MlpDev *mlps, int act,
                                                          int want_forces, double *scratch) {
  static_assert(!DEFZ || NZ == 2, "DEFZ needs the two-zeta grid");
  constexpr int NA = NG * NZ;
  extern __shared__ double lds[];
  const int cap = pl.cap, ndim = sf.ndim, nel = sf.n_elements, nrad = sf.n_rad;
  const Lds f = carve(lds, pl, ndim);
  const int c0 = b.blk_center[blockIdx.x], c1 = b.blk_center[blockIdx.x + 1];
  const int nc = c1 - c0;
  const int s0 = b.pair_start[c0];
  const int M = b.pair_start[c1] - s0;
  const double beta = ch.beta[0];
  const int tid = threadIdx.x;

  // ---------------- phase 0: stage ----------------
  if (tid < nc * (NSPEC + 1)) {
    const int cl = tid / (NSPEC + 1), k = tid - cl * (NSPEC + 1);
    f.segs[tid] = b.seg_start[(size_t)(c0 + cl) * (nel + 1) + k] - s0;
  }
  const int item = tid;
  const bool has_item = item < M;
  int base = 0, n = 0, a = 0, cl = 0, sa = 0;
  if (has_item) {
    const int64_t p = (int64_t)s0 + item;
    const int i = b.pair_i[p], j = b.pair_j[p];
    const double *h = b.cells + 9 * (size_t)b.frame_of_atom[i];
    const double sx = (double)b.pair_shift[3 * p], sy = (double)b.pair_shift[3 * p + 1],
                 sz = (double)b.pair_shift[3 * p + 2];
    const double *ri = b.pos + 3 * (size_t)i, *rj = b.pos + 3 * (size_t)j;
    const double dx = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
    const double dy = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
    const double dz = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
    const double r2 = dx * dx + dy * dy + dz * dz + sf.eps;  // universal.py:463-472
    const double inv_r = 1.0 / sqrt(r2);
    double2 *dst = reinterpret_cast<double2 *>(b.rec + kRecDoubles * (size_t)p);
    dst[0] = make_double2(dx, dy);
    dst[1] = make_double2(dz, r2);
    dst[2] = make_double2(inv_r, 0.0);
    dst[3] = make_double2(0.0, 0.0);
    f.x[item] = dx;
    f.y[item] = dy;
    f.z[item] = dz;
    f.r2[item] = r2;
    f.inv[item] = inv_r;
    base = b.pair_start[i] - s0;
    n = b.pair_start[i + 1] - b.pair_start[i];
    a = item - base;
    cl = i - c0;
    sa = b.species[j];
    const int k0 = 2 * base + a;
    f.xf[k0] = f.xf[k0 + n] = (float)dx;
    f.yf[k0] = f.yf[k0 + n] = (float)dy;
    f.zf[k0] = f.zf[k0 + n] = (float)dz;
    const double u = r2 * sf.inv_ac2;
    double H = 0.0, G = 0.0;
    if (u < 1.0) {
      double fc, dfdu;
      cutoff_u(sf.cutoff, u, fc, dfdu);
      const double e = ta_exp(-beta * u);
      H = e * fc;
      G = e * 2.0 * sf.inv_ac2 * (dfdu - beta * fc);
    }
    f.H[item] = H;
    f.G[item] = G;
    f.sp[item] = (unsigned char)sa;
    f.icl[item] = (unsigned char)cl;
  }
  __syncthreads();

  // ---------------- phase 1: forward sweep ----------------
  unsigned long long mask0 = 0ull;
  if (has_item) {
    const double ax = f.x[item], ay = f.y[item], az = f.z[item];
    const double ra2 = f.r2[item], inv_ra = f.inv[item], Ha = f.H[item];
    double acc[NSPEC][NG][NZ];
#pragma unroll
    for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) acc[sp][ig][iz] = 0.0;
    const int smax = (Ha != 0.0) ? (n >> 1) : 0;
    if (smax > 0) mask0 = partner_mask(sf, f, base, n, a, smax);
    unsigned long long mask = mask0;
    while (mask) {
      const int k = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      int bl = a + 1 + k;
      if (bl >= n) bl -= n;
      const int q = base + bl;
      const double ex = f.x[q] - ax, ey = f.y[q] - ay, ez = f.z[q] - az;
      const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
      const double u = d2 * sf.inv_ac2;
      if (!(u < 1.0)) continue;  // exact test (the mask is a superset)
      const double cth = (ra2 + f.r2[q] - d2) * 0.5 * inv_ra * f.inv[q];
      const double common = Ha * f.H[q] * hd_value<HD>(sf, ch, beta, u);
      const int sb = f.sp[q];
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) {
        const double basev = fma(ch.gamma[ig], cth, 1.0);
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) {
          double pw;
          if constexpr (DEFZ) {
            const double b2 = basev * basev;
            pw = (iz == 0) ? basev : b2 * b2;
          } else {
            if (ch.zeta_int[iz] > 0)
              pw = pow_int_m1(basev, ch.zeta_int[iz]) * basev;
            else
              pw = pow(basev, ch.zeta[iz]);
          }
          const double v = pw * common;
#pragma unroll
          for (int sp = 0; sp < NSPEC; ++sp)
            acc[sp][ig][iz] += (NSPEC == 1 || sb == sp) ? v : 0.0;
        }
      }
    }
#pragma unroll
    for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz)
          f.P[((sp * NG + ig) * NZ + iz) * cap + item] = acc[sp][ig][iz] * ch.kz[iz];
    // radial terms of the own pair (sf.py:101-108)
    {
      const double ur = ra2 * sf.inv_rc2;
      const double r = sqrt(ra2);
      const double fr = (ur < 1.0) ? cutoff_u_value(sf.cutoff, ur) : 0.0;
      for (int c = 0; c < nrad; ++c) {
        const double dr = r - sf.omega[c];
        f.P[(NSPEC * NA + c) * cap + item] = ta_exp(-sf.eta[c] * dr * dr * sf.inv_rc2) * fr;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 2: per-centre reduction -> G ----------------
  {
    const int group = tid >> 4, l = tid & 15, ngroups = blockDim.x >> 4;
    const int ntasks = nc * ndim;
    for (int t0 = 0; t0 < ntasks; t0 += ngroups) {
      const int task = t0 + group;
      const bool active = task < ntasks;
      const int tcl = active ? task / ndim : 0;
      const int k = active ? task - tcl * ndim : 0;
      const int sA = b.species[c0 + tcl];
      const int *seg = f.segs + tcl * (NSPEC + 1);
      double acc = 0.0;
      if (active) {
        if (k < sf.n_radial_dim) {
          const int tr = k / nrad, c = k - tr * nrad;
          // radial term tr of centre species sA collects neighbour species sb
          for (int sb = 0; sb < NSPEC; ++sb)
            if (rterm(sA, sb) == tr)
              for (int q = seg[sb] + l; q < seg[sb + 1]; q += 16) acc += f.P[(NSPEC * NA + c) * cap + q];
        } else {
          const int ka = k - sf.n_radial_dim;
          const int t = ka / sf.n_ang, cc = ka - t * sf.n_ang;
          // which (ig, iz) of this launch is channel cc (the chunk holds every angular channel)
          int slot = 0;
          for (int j = 0; j < NA; ++j)
            if (ch.chan[j] == cc) slot = j;
          for (int s1 = 0; s1 < NSPEC; ++s1)
            for (int s2 = s1; s2 < NSPEC; ++s2)
              if (aterm(s1, s2, nel) == t) {
                for (int q = seg[s1] + l; q < seg[s1 + 1]; q += 16) acc += f.P[(s2 * NA + slot) * cap + q];
                if (s1 != s2)
                  for (int q = seg[s2] + l; q < seg[s2 + 1]; q += 16)
                    acc += f.P[(s1 * NA + slot) * cap + q];
              }
        }
      }
      acc = row16_sum(acc);
      if (active && l == 0) {
        f.gtab[tcl * ndim + k] = acc;
        b.G[(size_t)(c0 + tcl) * ndim + k] = acc;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 3: MLP, per element ----------------
  {
    double *da = scratch + (size_t)blockIdx.x * kMaxLayers * kMlpRows * pl.stride;
    for (int e = 0; e < NSPEC; ++e) {
      bool any = false;
      for (int c = 0; c < nc; ++c) any |= (b.species[c0 + c] == e);
      if (!any) continue;  // uniform across the workgroup
      for (int idx = tid; idx < kMlpRows * ndim; idx += blockDim.x) {
        const int row = idx / ndim, k = idx - row * ndim;
        f.buf0[row * pl.stride + k] = row < nc ? f.gtab[row * ndim + k] : 0.0;
      }
      __syncthreads();
      mlp_tile(
          mlps[e], act, ndim, nc, f.buf0, f.buf1, pl.stride, da,
          [&](int row, double y) {
            if (b.species[c0 + row] == e) b.eatom[c0 + row] = y;
          },
          [&](int row, int k, double d) {
            if (b.species[c0 + row] == e) f.dtab[row * ndim + k] = d;
          });
    }
  }
  if (!want_forces) return;

  // ---------------- phase 4: backward sweep ----------------
  for (int k = tid; k < 3 * cap; k += blockDim.x) f.gacc[k] = 0.0;
  __syncthreads();
  if (has_item) {
    const double ax = f.x[item], ay = f.y[item], az = f.z[item];
    const double ra2 = f.r2[item], inv_ra = f.inv[item], Ha = f.H[item], Ga = f.G[item];
    const double inv_ra2 = inv_ra * inv_ra;
    const double *dE = f.dtab + cl * ndim;
    double w[NSPEC][NG][NZ], wd[NSPEC][NG][NZ];
#pragma unroll
    for (int sp = 0; sp < NSPEC; ++sp) {
      const double *wsrc = dE + sf.n_radial_dim + aterm(sa, sp, nel) * sf.n_ang;
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) {
          w[sp][ig][iz] = wsrc[ch.chan[ig * NZ + iz]] * ch.kz[iz];
          wd[sp][ig][iz] = w[sp][ig][iz] * ch.zeta[iz] * ch.gamma[ig];
        }
    }
    double gx = 0.0, gy = 0.0, gz = 0.0;
    unsigned long long mask = mask0;
    while (mask) {
      const int k = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      int bl = a + 1 + k;
      if (bl >= n) bl -= n;
      const int q = base + bl;
      const double bx = f.x[q], by = f.y[q], bz = f.z[q];
      const double ex = bx - ax, ey = by - ay, ez = bz - az;
      const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
      const double u = d2 * sf.inv_ac2;
      if (!(u < 1.0)) continue;
      const double inv_rb = f.inv[q];
      const double inv_ab = inv_ra * inv_rb;
      const double cth = (ra2 + f.r2[q] - d2) * 0.5 * inv_ab;
      double Hd, dHd;
      hd_eval<HD>(sf, ch, beta, u, Hd, dHd);
      const double Hd2 = 2.0 * sf.inv_ac2 * dHd;
      const double Hb = f.H[q], Gb = f.G[q];
      const int sb = f.sp[q];
      double S0 = 0.0, S1 = 0.0;
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) {
        const double basev = fma(ch.gamma[ig], cth, 1.0);
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) {
          double ws = w[0][ig][iz], wds = wd[0][ig][iz];
#pragma unroll
          for (int sp = 1; sp < NSPEC; ++sp) {
            ws = (sb == sp) ? w[sp][ig][iz] : ws;
            wds = (sb == sp) ? wd[sp][ig][iz] : wds;
          }
          if constexpr (DEFZ) {
            if (iz == 0) {
              S0 = fma(ws, basev, S0);
              S1 += wds;
            } else {
              const double b2 = basev * basev;
              S0 = fma(ws, b2 * b2, S0);
              S1 = fma(wds, b2 * basev, S1);
            }
          } else {
            double pm1;
            if (ch.zeta_int[iz] > 0)
              pm1 = pow_int_m1(basev, ch.zeta_int[iz]);
            else
              pm1 = pow(basev, ch.zeta[iz] - 1.0);
            S0 = fma(ws, pm1 * basev, S0);
            S1 = fma(wds, pm1, S1);
          }
        }
      }
      const double Aa = Hb * Hd * fma(S1 * Ha, inv_ab - cth * inv_ra2, S0 * Ga);
      const double Ab = Ha * Hd * fma(S1 * Hb, inv_ab - cth * inv_rb * inv_rb, S0 * Gb);
      const double Q = Ha * Hb * fma(-S1 * inv_ab, Hd, S0 * Hd2);
      const double ca = Aa + Q, cb = Ab + Q;
      gx = fma(ca, ax, fma(-Q, bx, gx));
      gy = fma(ca, ay, fma(-Q, by, gy));
      gz = fma(ca, az, fma(-Q, bz, gz));
      atomicAdd(&f.gacc[q], fma(cb, bx, -Q * ax));
      atomicAdd(&f.gacc[cap + q], fma(cb, by, -Q * ay));
      atomicAdd(&f.gacc[2 * cap + q], fma(cb, bz, -Q * az));
    }
    // radial (G2) share of the own pair: s D / r
    {
      const double ur = ra2 * sf.inv_rc2;
      double s = 0.0;
      if (ur < 1.0) {
        double fc, dfdu;
        cutoff_u(sf.cutoff, ur, fc, dfdu);
        const double r = sqrt(ra2);
        const double dfdr = dfdu * 2.0 * r * sf.inv_rc2;
        const double *wr = dE + rterm(b.species[c0 + cl], sa) * nrad;
        for (int c = 0; c < nrad; ++c) {
          const double dr = r - sf.omega[c];
          const double e = ta_exp(-sf.eta[c] * dr * dr * sf.inv_rc2);
          s = fma(wr[c], e * (dfdr - 2.0 * sf.eta[c] * dr * fc * sf.inv_rc2), s);
        }
      }
      s *= inv_ra;
      gx = fma(s, ax, gx);
      gy = fma(s, ay, gy);
      gz = fma(s, az, gz);
    }
    atomicAdd(&f.gacc[item], gx);
    atomicAdd(&f.gacc[cap + item], gy);
    atomicAdd(&f.gacc[2 * cap + item], gz);
  }
  __syncthreads();
  if (has_item) {
    const int64_t p = (int64_t)s0 + item;
    double2 *dst = reinterpret_cast<double2 *>(b.g + 4 * (size_t)p);
    dst[0] = make_double2(f.gacc[item], f.gacc[cap + item]);
    dst[1] = make_double2(f.gacc[2 * cap + item], 0.0);
  }
}

}  // namespace

// LDS plan; returns false when the fused kernel does not apply.
bool fused_plan(const SFParams &sf, int nspec, int ng, int nz, int cap, int mlp_stride_max,
                FusedPlan &pl) {
  if (cap > kBlock || sf.n_rad > kMaxRadFused || sf.ndim > kMaxDimFused) return false;
  pl.cap = cap;
  pl.npart = nspec * ng * nz + sf.n_rad;
  pl.stride = mlp_stride_max;
  int off = kNF * cap;                             // fields
  off += (3 * (2 * cap + kRingPad) + 1) / 2;       // float rings
  off += (2 * cap + 7) / 8;                        // species + centre bytes
  pl.off_tab = off;
  off += 2 * kMaxCentersPerBlock * sf.ndim;        // gtab, dtab
  off += (kMaxCentersPerBlock * (nspec + 1) + 1) / 2;  // segs (ints)
  pl.off_b = off;
  const int region = std::max(std::max(pl.npart * cap, kMlpRows * pl.stride), 3 * cap);
  off += region;
  pl.off_buf0 = off;
  off += kMlpRows * pl.stride;
  pl.total = off;
  return (size_t)off * sizeof(double) <= 64 * 1024;
}

size_t fused_scratch_doubles(const FusedPlan &pl, int n_blk) {
  return (size_t)n_blk * kMaxLayers * kMlpRows * pl.stride;
}

template <int NSPEC, int NG, int NZ>
static void fused_t(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b, const FusedPlan &pl,
                    const MlpDev *mlps, int act, int want_forces, double *scratch, hipStream_t s) {
  const dim3 grid((unsigned)b.n_blk), block((unsigned)(b.cap < kBlock ? b.cap : kBlock));
  const size_t lds = (size_t)pl.total * sizeof(double);
  if constexpr (NZ == 2) {
    if (ch.n_hd > 0 && ch.n_hd <= 16 && ch.zeta_int[0] == 1 && ch.zeta_int[1] == 4) {
      hipLaunchKernelGGL((sf_fused_kernel<NSPEC, NG, NZ, 16, true>), grid, block, lds, s, sf, ch, b, pl, mlps, act, want_forces, scratch);
      return;
    }
  }
  if (ch.n_hd > 0 && ch.n_hd <= 16)  // coefficients beyond n_hd are zero
    hipLaunchKernelGGL((sf_fused_kernel<NSPEC, NG, NZ, 16, false>), grid, block, lds, s, sf, ch, b, pl, mlps, act, want_forces, scratch);
  else if (ch.n_hd == 24)
    hipLaunchKernelGGL((sf_fused_kernel<NSPEC, NG, NZ, 24, false>), grid, block, lds, s, sf, ch, b, pl, mlps, act, want_forces, scratch);
  else
    hipLaunchKernelGGL((sf_fused_kernel<NSPEC, NG, NZ, 0, false>), grid, block, lds, s, sf, ch, b, pl, mlps, act, want_forces, scratch);
}

void launch_sf_fused(const SFParams &sf, const AngChunk &ch, int ng, int nz, const DeviceBatch &b,
                     const FusedPlan &pl, const MlpDev *mlps, int act, bool want_forces,
                     double *scratch, hipStream_t s) {
  if (b.n_blk == 0) return;
  const int key = sf.n_elements * 100 + ng * 10 + nz;
  const int wf = want_forces ? 1 : 0;
  switch (key) {
    case 111: fused_t<1, 1, 1>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 112: fused_t<1, 1, 2>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 121: fused_t<1, 2, 1>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 122: fused_t<1, 2, 2>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 211: fused_t<2, 1, 1>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 212: fused_t<2, 1, 2>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 221: fused_t<2, 2, 1>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 222: fused_t<2, 2, 2>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 311: fused_t<3, 1, 1>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 312: fused_t<3, 1, 2>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 321: fused_t<3, 2, 1>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    case 322: fused_t<3, 2, 2>(sf, ch, b, pl, mlps, act, wf, scratch, s); break;
    default: break;
  }
}

}  // namespace ta
