// Angular (G4) kernels, third generation: the second generation's evaluation
// (every unordered triple once, only if r_jk < acut; e^{-beta u} fc(u) as one
// Horner sweep; see ta_kernels_v2.hip) with the work of a wavefront REBALANCED.
//
// Why. With one lane per directed pair (i, a) the number of partners b that
// survive the r_jk < acut test differs a lot between lanes (mean 20, sigma 7.8,
// max 47 on the 4000-atom Ni frame), and a wavefront runs as long as its
// busiest lane: measured lane utilisation of the expensive body was 54 %.
//
// How. Per block of 32 candidate steps every lane builds its bit mask (fp32
// scan, as before); the wavefront then takes an exclusive prefix sum of the
// popcounts, and every lane writes its surviving (pair a, partner b) entries,
// 16 bits each, into a wavefront-private LDS list. Lane L then processes the
// CONTIGUOUS slice [L K, (L+1) K) of that list, K = ceil(total / 64): equal
// work for all 64 lanes, and because a slice is a run of the compacted list it
// covers only one to three different owners a, so the owner's fields and its
// partial sums stay in registers and are flushed with LDS atomics only when
// the owner changes. The partner's share goes through ds_add_f64 as before.
//
// Replaces the same reference ops as ta_kernels_v2.hip: build_angular_graph +
// _apply_g4_functions (transformer/universal.py:622-694, nn/atomic/sf.py:121-182)
// and their tf.gradients (nn/basic.py:277-331).
#include <hip/hip_runtime.h>

#include "ta_device.h"
#include "ta_math.h"

namespace ta {
namespace {

constexpr int kBlock = 256;   // upper bound; launched with min(cap, 256) lanes
constexpr int kNF = 7;        // x y z r2 inv_r H G
constexpr int kRingPad = 64;
constexpr int kChunk = 32;    // candidate steps per pass
constexpr int kList = 64 * kChunk;  // list entries per wavefront and pass

__device__ __forceinline__ int aterm(int s1, int s2, int nel) {
  int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
  return a * nel - (a * (a - 1)) / 2 + (b - a);
}
__device__ __forceinline__ int rterm(int center, int other) {
  return other == center ? 0 : (other < center ? other + 1 : other);
}

struct Lds {
  double *x, *y, *z, *r2, *inv, *H, *G;  // [cap]
  float *xf, *yf, *zf;                    // rings [2 cap + pad]
  unsigned char *sp, *ibase, *inum, *icl; // per item: species of j, first item of its centre,
                                          // neighbour count, centre index inside the block
  unsigned short *list;                   // [waves][kList]
  double *acc;                            // accumulators / tables (kernel specific)
};

// sizes in bytes of the common part (multiple of 8)
__host__ __device__ inline size_t common_bytes(int cap, int nwaves) {
  size_t b = (size_t)cap * kNF * 8 + 3 * (size_t)(2 * cap + kRingPad) * 4 + 4 * (size_t)cap +
             (size_t)nwaves * kList * 2;
  return (b + 7) & ~(size_t)7;
}

__device__ __forceinline__ Lds carve(double *lds, int cap, int nwaves) {
  Lds f;
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
  f.ibase = f.sp + cap;
  f.inum = f.ibase + cap;
  f.icl = f.inum + cap;
  f.list = reinterpret_cast<unsigned short *>(f.icl + cap);
  f.acc = reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + common_bytes(cap, nwaves));
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

// Stage the pair records of the workgroup's centres (see ta_kernels_v2.hip::stage).
__device__ __forceinline__ void stage(const SFParams &sf, double beta, const DeviceBatch &b,
                                      const Lds &f, int c0, int s0, int M, int geom) {
  for (int item = threadIdx.x; item < M; item += blockDim.x) {
    const int64_t p = (int64_t)s0 + item;
    const int i = b.pair_i[p];
    double2 v0, v1, v2;
    if (geom) {
      const int j = b.pair_j[p];
      const double *h = b.cells + 9 * (size_t)b.frame_of_atom[i];
      const double sx = (double)b.pair_shift[3 * p], sy = (double)b.pair_shift[3 * p + 1],
                   sz = (double)b.pair_shift[3 * p + 2];
      const double *ri = b.pos + 3 * (size_t)i, *rj = b.pos + 3 * (size_t)j;
      const double dx = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
      const double dy = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
      const double dz = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
      const double r2 = dx * dx + dy * dy + dz * dz + sf.eps;  // universal.py:463-472
      v0 = make_double2(dx, dy);
      v1 = make_double2(dz, r2);
      v2 = make_double2(1.0 / sqrt(r2), 0.0);
      double2 *dst = reinterpret_cast<double2 *>(b.rec + kRecDoubles * (size_t)p);
      dst[0] = v0;
      dst[1] = v1;
      dst[2] = v2;
      dst[3] = make_double2(0.0, 0.0);
    } else {
      const double2 *src = reinterpret_cast<const double2 *>(b.rec + kRecDoubles * (size_t)p);
      v0 = src[0];
      v1 = src[1];
      v2 = src[2];
    }
    f.x[item] = v0.x;
    f.y[item] = v0.y;
    f.z[item] = v1.x;
    f.r2[item] = v1.y;
    f.inv[item] = v2.x;
    const int cbase = b.pair_start[i] - s0, cn = b.pair_start[i + 1] - b.pair_start[i];
    const int k0 = 2 * cbase + (item - cbase);
    f.xf[k0] = f.xf[k0 + cn] = (float)v0.x;
    f.yf[k0] = f.yf[k0 + cn] = (float)v0.y;
    f.zf[k0] = f.zf[k0 + cn] = (float)v1.x;
    f.ibase[item] = (unsigned char)cbase;
    f.inum[item] = (unsigned char)cn;
    f.icl[item] = (unsigned char)(i - c0);
    const double u = v1.y * sf.inv_ac2;
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
    f.sp[item] = (unsigned char)b.species[b.pair_j[p]];
  }
}

// 32 candidate steps sc .. sc+31 of the lane's ring position (fp32, superset mask)
__device__ __forceinline__ unsigned scan32(const SFParams &sf, const Lds &f, int base, int n, int a,
                                           int sc, int smax) {
  const int count = smax - sc + 1;
  if (count <= 0) return 0u;
  const int ring = 2 * base + a;
  const float ax = f.xf[ring], ay = f.yf[ring], az = f.zf[ring];
  const float lim = (float)(sf.acut * sf.acut) * 1.0001f;
  unsigned mask = 0u;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (16 * g < count) {
      const float *px = f.xf + ring + sc + 16 * g, *py = f.yf + ring + sc + 16 * g,
                  *pz = f.zf + ring + sc + 16 * g;
      unsigned m = 0u;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float ex = px[k] - ax, ey = py[k] - ay, ez = pz[k] - az;
        const float d2 = fmaf(ex, ex, fmaf(ey, ey, ez * ez));
        m |= (d2 < lim) ? (1u << k) : 0u;
      }
      mask |= m << (16 * g);
    }
  }
  if (count < 32) mask &= (1u << count) - 1u;
  // even n: the antipodal partner is shared by two lanes, the lower one keeps it
  const int half = n >> 1;
  if (!(n & 1) && a >= half && half >= sc && half < sc + 32) mask &= ~(1u << (half - sc));
  return mask;
}

// wavefront exclusive prefix sum of a small non-negative integer; also returns the total
__device__ __forceinline__ int wave_exclusive_sum(int v, int lane, int &total) {
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  total = __shfl(incl, 63, 64);
  return incl - v;
}

// Build the wavefront's list for one pass; returns the number of entries.
__device__ __forceinline__ int build_list(const SFParams &sf, const Lds &f, unsigned short *list,
                                          int item, bool has_item, int lane, int sc) {
  unsigned mask = 0u;
  int base = 0, n = 0, a = 0;
  if (has_item) {
    base = f.ibase[item];
    n = f.inum[item];
    a = item - base;
    const int smax = (f.H[item] != 0.0) ? (n >> 1) : 0;
    mask = scan32(sf, f, base, n, a, sc, smax);
  }
  int total;
  int pos = wave_exclusive_sum(__popc(mask), lane, total);
  while (mask) {
    const int k = __ffs((int)mask) - 1;
    mask &= mask - 1u;
    int bl = a + sc + k;
    if (bl >= n) bl -= n;
    list[pos++] = (unsigned short)((item << 8) | (base + bl));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return total;
}

// --------------------------------------------------------------------------------------
// forward: per-pair partial sums of the G4 summands -> part4[(sp * n_ang + c)][p]
// --------------------------------------------------------------------------------------
template <int NSPEC, int NG, int NZ, int HD, bool DEFZ>
__global__ __launch_bounds__(kBlock) void g4_forward_v3_kernel(SFParams sf, AngChunk ch,
                                                               DeviceBatch b, int geom) {
  static_assert(!DEFZ || NZ == 2, "DEFZ needs the two-zeta grid");
  constexpr int NCH = NSPEC * NG * NZ;
  extern __shared__ double lds[];
  const int cap = b.cap;
  const int nwaves = blockDim.x >> 6;
  const Lds f = carve(lds, cap, nwaves);
  double *facc = f.acc;  // [NCH][cap] per-item sums
  const int c0 = b.blk_center[blockIdx.x], c1 = b.blk_center[blockIdx.x + 1];
  const int s0 = b.pair_start[c0];
  const int M = b.pair_start[c1] - s0;
  const double beta = ch.beta[0];
  for (int k = threadIdx.x; k < NCH * cap; k += blockDim.x) facc[k] = 0.0;
  stage(sf, beta, b, f, c0, s0, M, geom);
  // passes needed by the largest centre of the workgroup
  int nmax = 0;
  for (int c = c0; c < c1; ++c) nmax = max(nmax, b.pair_start[c + 1] - b.pair_start[c]);
  const int npass = ((nmax >> 1) + kChunk - 1) / kChunk;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned short *list = f.list + wave * kList;
  for (int base_item = 0; base_item < M; base_item += blockDim.x) {
    const int item = base_item + threadIdx.x;
    const bool has_item = item < M;
    for (int pass = 0; pass < npass; ++pass) {
      const int sc = 1 + pass * kChunk;
      const int total = build_list(sf, f, list, item, has_item, lane, sc);
      const int K = (total + 63) >> 6;
      const int lo = lane * K, hi = min(total, lo + K);
      // The owner's fields are re-read from LDS for every entry (no data-dependent reload
      // branch: owners change in almost every iteration for SOME lane of the wavefront);
      // its partial sums stay in registers and leave through a short predicated flush when
      // the next entry of the slice has another owner.
      double acc[NSPEC][NG][NZ];
#pragma unroll
      for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
#pragma unroll
          for (int iz = 0; iz < NZ; ++iz) acc[sp][ig][iz] = 0.0;
      int e = (lo < hi) ? list[lo] : -1;
      for (int k = 0; k < K; ++k) {
        const int enext = (lo + k + 1 < hi) ? list[lo + k + 1] : -1;
        if (e >= 0) {
          const int it = e >> 8, q = e & 255;
          const double ax = f.x[it], ay = f.y[it], az = f.z[it];
          const double ex = f.x[q] - ax, ey = f.y[q] - ay, ez = f.z[q] - az;
          const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
          const double u = d2 * sf.inv_ac2;
          if (u < 1.0) {  // exact test; the list is a superset
            const double cth = (f.r2[it] + f.r2[q] - d2) * 0.5 * f.inv[it] * f.inv[q];
            const double common = f.H[it] * f.H[q] * hd_value<HD>(sf, ch, beta, u);
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
          if ((enext >> 8) != it) {  // enext == -1 -> -1 != it: last entry of the slice
#pragma unroll
            for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
              for (int ig = 0; ig < NG; ++ig)
#pragma unroll
                for (int iz = 0; iz < NZ; ++iz) {
                  atomicAdd(&facc[((sp * NG + ig) * NZ + iz) * cap + it], acc[sp][ig][iz]);
                  acc[sp][ig][iz] = 0.0;
                }
          }
        }
        e = enext;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (has_item) {
      const int64_t p = (int64_t)s0 + item;
#pragma unroll
      for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
#pragma unroll
          for (int iz = 0; iz < NZ; ++iz) {
            const int c = ch.chan[ig * NZ + iz];
            double *slot = &facc[((sp * NG + ig) * NZ + iz) * cap + item];
            b.part4[(size_t)(sp * sf.n_ang + c) * b.n_pairs + p] = *slot * ch.kz[iz];
          }
    }
    // (a second sweep over the same LDS rows only happens when cap > 256; rows differ)
  }
}

// --------------------------------------------------------------------------------------
// backward: g[p] = dE/dD_p
// --------------------------------------------------------------------------------------
template <int NSPEC, int NG, int NZ, int HD, bool DEFZ>
__global__ __launch_bounds__(kBlock) void backward_v3_kernel(SFParams sf, AngChunk ch,
                                                             DeviceBatch b, int first) {
  static_assert(!DEFZ || NZ == 2, "DEFZ needs the two-zeta grid");
  constexpr int NW = NSPEC * NSPEC * NG * NZ;  // weights per centre
  extern __shared__ double lds[];
  const int cap = b.cap;
  const int nwaves = blockDim.x >> 6;
  const Lds f = carve(lds, cap, nwaves);
  double *gacc = f.acc;                // [3][cap]
  double *wtab = gacc + 3 * cap;       // [kMaxCentersPerBlock][NW]  w = dE/dG 2^(1-zeta)
  double *wdtab = wtab + kMaxCentersPerBlock * NW;  // w zeta gamma
  const int c0 = b.blk_center[blockIdx.x], c1 = b.blk_center[blockIdx.x + 1];
  const int s0 = b.pair_start[c0];
  const int M = b.pair_start[c1] - s0;
  const double beta = ch.beta[0];
  const int nel = sf.n_elements;
  for (int k = threadIdx.x; k < 3 * cap; k += blockDim.x) gacc[k] = 0.0;
  // per-centre weight tables: [centre][sa][sp][ig][iz]
  for (int idx = threadIdx.x; idx < (c1 - c0) * NW; idx += blockDim.x) {
    const int cl = idx / NW, r = idx - cl * NW;
    const int sa = r / (NSPEC * NG * NZ), r2 = r - sa * (NSPEC * NG * NZ);
    const int sp = r2 / (NG * NZ), r3 = r2 - sp * (NG * NZ);
    const int ig = r3 / NZ, iz = r3 - ig * NZ;
    const double *wsrc = b.dEdG + (size_t)(c0 + cl) * sf.ndim + sf.n_radial_dim +
                         aterm(sa, sp, nel) * sf.n_ang;
    const double w = wsrc[ch.chan[ig * NZ + iz]] * ch.kz[iz];
    wtab[idx] = w;
    wdtab[idx] = w * ch.zeta[iz] * ch.gamma[ig];
  }
  stage(sf, beta, b, f, c0, s0, M, 0);
  int nmax = 0;
  for (int c = c0; c < c1; ++c) nmax = max(nmax, b.pair_start[c + 1] - b.pair_start[c]);
  const int npass = ((nmax >> 1) + kChunk - 1) / kChunk;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned short *list = f.list + wave * kList;
  for (int base_item = 0; base_item < M; base_item += blockDim.x) {
    const int item = base_item + threadIdx.x;
    const bool has_item = item < M;
    for (int pass = 0; pass < npass; ++pass) {
      const int sc = 1 + pass * kChunk;
      const int total = build_list(sf, f, list, item, has_item, lane, sc);
      const int K = (total + 63) >> 6;
      const int lo = lane * K, hi = min(total, lo + K);
      // Owner fields are re-read per entry; the per-centre weights sit in registers and are
      // reloaded only when the slice crosses into another centre or owner species (rare: a
      // wavefront spans at most a few centres); the owner's dE/dD leaves through a short
      // predicated flush when the next entry has another owner.
      int curw = -1;
      double gx = 0, gy = 0, gz = 0;
      double w[NSPEC][NG][NZ], wd[NSPEC][NG][NZ];
#pragma unroll
      for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
#pragma unroll
          for (int iz = 0; iz < NZ; ++iz) w[sp][ig][iz] = wd[sp][ig][iz] = 0.0;
      int e = (lo < hi) ? list[lo] : -1;
      for (int k = 0; k < K; ++k) {
        const int enext = (lo + k + 1 < hi) ? list[lo + k + 1] : -1;
        if (e >= 0) {
          const int it = e >> 8, q = e & 255;
          const int wkey = (int)f.icl[it] * NSPEC + (int)f.sp[it];
          if (wkey != curw) {
            curw = wkey;
            const double *wt = wtab + wkey * (NSPEC * NG * NZ);
            const double *wdt = wdtab + wkey * (NSPEC * NG * NZ);
#pragma unroll
            for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
              for (int ig = 0; ig < NG; ++ig)
#pragma unroll
                for (int iz = 0; iz < NZ; ++iz) {
                  w[sp][ig][iz] = wt[(sp * NG + ig) * NZ + iz];
                  wd[sp][ig][iz] = wdt[(sp * NG + ig) * NZ + iz];
                }
          }
          const double ax = f.x[it], ay = f.y[it], az = f.z[it];
          const double bx = f.x[q], by = f.y[q], bz = f.z[q];
          const double ex = bx - ax, ey = by - ay, ez = bz - az;
          const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
          const double u = d2 * sf.inv_ac2;
          if (u < 1.0) {  // exact test; the list is a superset
            const double inv_ra = f.inv[it], inv_rb = f.inv[q];
            const double inv_ab = inv_ra * inv_rb;
            const double cth = (f.r2[it] + f.r2[q] - d2) * 0.5 * inv_ab;
            double Hd, dHd;
            hd_eval<HD>(sf, ch, beta, u, Hd, dHd);
            const double Hd2 = 2.0 * sf.inv_ac2 * dHd;
            const double Ha = f.H[it], Ga = f.G[it], Hb = f.H[q], Gb = f.G[q];
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
            const double Aa = Hb * Hd * fma(S1 * Ha, inv_ab - cth * inv_ra * inv_ra, S0 * Ga);
            const double Ab = Ha * Hd * fma(S1 * Hb, inv_ab - cth * inv_rb * inv_rb, S0 * Gb);
            const double Q = Ha * Hb * fma(-S1 * inv_ab, Hd, S0 * Hd2);
            const double ca = Aa + Q, cb = Ab + Q;
            gx = fma(ca, ax, fma(-Q, bx, gx));
            gy = fma(ca, ay, fma(-Q, by, gy));
            gz = fma(ca, az, fma(-Q, bz, gz));
            atomicAdd(&gacc[q], fma(cb, bx, -Q * ax));
            atomicAdd(&gacc[cap + q], fma(cb, by, -Q * ay));
            atomicAdd(&gacc[2 * cap + q], fma(cb, bz, -Q * az));
          }
          if ((enext >> 8) != it) {
            atomicAdd(&gacc[it], gx);
            atomicAdd(&gacc[cap + it], gy);
            atomicAdd(&gacc[2 * cap + it], gz);
            gx = gy = gz = 0.0;
          }
        }
        e = enext;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int item = threadIdx.x; item < M; item += blockDim.x) {
    const int64_t p = (int64_t)s0 + item;
    double gx = gacc[item], gy = gacc[cap + item], gz = gacc[2 * cap + item];
    if (first) {
      const int i = b.pair_i[p];
      const double r2 = f.r2[item], inv_r = f.inv[item];
      const double ur = r2 * sf.inv_rc2;
      double s = 0.0;
      if (ur < 1.0) {
        double fc, dfdu;
        cutoff_u(sf.cutoff, ur, fc, dfdu);
        const double r = sqrt(r2);
        const double dfdr = dfdu * 2.0 * r * sf.inv_rc2;
        const double *wr = b.dEdG + (size_t)i * sf.ndim + rterm(b.species[i], f.sp[item]) * sf.n_rad;
        for (int c = 0; c < sf.n_rad; ++c) {
          const double dr = r - sf.omega[c];
          const double e = ta_exp(-sf.eta[c] * dr * dr * sf.inv_rc2);
          s = fma(wr[c], e * (dfdr - 2.0 * sf.eta[c] * dr * fc * sf.inv_rc2), s);
        }
      }
      s *= inv_r;
      gx = fma(s, f.x[item], gx);
      gy = fma(s, f.y[item], gy);
      gz = fma(s, f.z[item], gz);
    } else {
      gx += b.g[4 * (size_t)p];
      gy += b.g[4 * (size_t)p + 1];
      gz += b.g[4 * (size_t)p + 2];
    }
    b.g[4 * (size_t)p] = gx;
    b.g[4 * (size_t)p + 1] = gy;
    b.g[4 * (size_t)p + 2] = gz;
  }
}

inline int v3_threads(const DeviceBatch &b) { return b.cap < kBlock ? b.cap : kBlock; }

template <int NSPEC, int NG, int NZ>
void fwd_t(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b, int geom, hipStream_t s) {
  const int threads = v3_threads(b);
  const dim3 grid((unsigned)b.n_blk), block((unsigned)threads);
  const size_t lds = common_bytes(b.cap, threads / 64) + (size_t)NSPEC * NG * NZ * b.cap * 8;
  if constexpr (NZ == 2) {
    if (ch.n_hd > 0 && ch.n_hd <= 16 && ch.zeta_int[0] == 1 && ch.zeta_int[1] == 4) {
      hipLaunchKernelGGL((g4_forward_v3_kernel<NSPEC, NG, NZ, 16, true>), grid, block, lds, s, sf, ch, b, geom);
      return;
    }
  }
  if (ch.n_hd > 0 && ch.n_hd <= 16)  // coefficients beyond n_hd are zero
    hipLaunchKernelGGL((g4_forward_v3_kernel<NSPEC, NG, NZ, 16, false>), grid, block, lds, s, sf, ch, b, geom);
  else if (ch.n_hd == 24)
    hipLaunchKernelGGL((g4_forward_v3_kernel<NSPEC, NG, NZ, 24, false>), grid, block, lds, s, sf, ch, b, geom);
  else
    hipLaunchKernelGGL((g4_forward_v3_kernel<NSPEC, NG, NZ, 0, false>), grid, block, lds, s, sf, ch, b, geom);
}
template <int NSPEC, int NG, int NZ>
void bwd_t(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b, int first, hipStream_t s) {
  const int threads = v3_threads(b);
  const dim3 grid((unsigned)b.n_blk), block((unsigned)threads);
  const size_t lds = common_bytes(b.cap, threads / 64) +
                     (3 * (size_t)b.cap + 2 * (size_t)kMaxCentersPerBlock * NSPEC * NSPEC * NG * NZ) * 8;
  if constexpr (NZ == 2) {
    if (ch.n_hd > 0 && ch.n_hd <= 16 && ch.zeta_int[0] == 1 && ch.zeta_int[1] == 4) {
      hipLaunchKernelGGL((backward_v3_kernel<NSPEC, NG, NZ, 16, true>), grid, block, lds, s, sf, ch, b, first);
      return;
    }
  }
  if (ch.n_hd > 0 && ch.n_hd <= 16)  // coefficients beyond n_hd are zero
    hipLaunchKernelGGL((backward_v3_kernel<NSPEC, NG, NZ, 16, false>), grid, block, lds, s, sf, ch, b, first);
  else if (ch.n_hd == 24)
    hipLaunchKernelGGL((backward_v3_kernel<NSPEC, NG, NZ, 24, false>), grid, block, lds, s, sf, ch, b, first);
  else
    hipLaunchKernelGGL((backward_v3_kernel<NSPEC, NG, NZ, 0, false>), grid, block, lds, s, sf, ch, b, first);
}

}  // namespace

#define TA_DISPATCH_V3(FN, ...)                                   \
  do {                                                            \
    const int key = nspec * 100 + ng * 10 + nz;                   \
    switch (key) {                                                \
      case 111: FN<1, 1, 1>(__VA_ARGS__); break;                  \
      case 112: FN<1, 1, 2>(__VA_ARGS__); break;                  \
      case 121: FN<1, 2, 1>(__VA_ARGS__); break;                  \
      case 122: FN<1, 2, 2>(__VA_ARGS__); break;                  \
      case 211: FN<2, 1, 1>(__VA_ARGS__); break;                  \
      case 212: FN<2, 1, 2>(__VA_ARGS__); break;                  \
      case 221: FN<2, 2, 1>(__VA_ARGS__); break;                  \
      case 222: FN<2, 2, 2>(__VA_ARGS__); break;                  \
      case 311: FN<3, 1, 1>(__VA_ARGS__); break;                  \
      case 312: FN<3, 1, 2>(__VA_ARGS__); break;                  \
      case 321: FN<3, 2, 1>(__VA_ARGS__); break;                  \
      case 322: FN<3, 2, 2>(__VA_ARGS__); break;                  \
      default: break;                                             \
    }                                                             \
  } while (0)

void launch_g4_forward_v3(const SFParams &sf, const AngChunk &ch, int ng, int nz, bool geometry,
                          const DeviceBatch &b, hipStream_t s) {
  if (b.n_blk == 0) return;
  const int nspec = sf.n_elements;
  const int geom = geometry ? 1 : 0;
  TA_DISPATCH_V3(fwd_t, sf, ch, b, geom, s);
}

void launch_backward_v3(const SFParams &sf, const AngChunk &ch, int ng, int nz, bool first,
                        const DeviceBatch &b, hipStream_t s) {
  if (b.n_blk == 0) return;
  const int nspec = sf.n_elements;
  const int f = first ? 1 : 0;
  TA_DISPATCH_V3(bwd_t, sf, ch, b, f, s);
}

}  // namespace ta
