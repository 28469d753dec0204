// Angular (G4) kernels, second generation: every unordered triple {j, k} of a
// centre is evaluated ONCE, and only when r_jk < acut.
//
// Replaces the same reference ops as ta_kernels.hip (build_angular_graph +
// _apply_g4_functions, transformer/universal.py:622-694, nn/atomic/sf.py:121-182,
// and their tf.gradients, nn/basic.py:277-331).
//
// Work decomposition. A workgroup (min(cap, 256) lanes) owns a run of WHOLE
// centre atoms whose neighbour counts sum to <= cap (packed on the host, cap =
// 192 or the largest neighbour count, so normally one lane per directed pair
// and 3 wavefronts per workgroup); their pair records are
// staged in LDS as structure-of-arrays {x, y, z, r^2, 1/r, H, G} (H = exp(-beta
// u) fc(u), G = (dH/dr)/r for this launch's beta, computed while staging) so a
// wavefront's per-lane reads of consecutive neighbours are bank-conflict free.
// One lane = one directed pair (i, a). Lane a is responsible for the partners
// b = a + s (mod n), s = 1..n/2 ("rotation" schedule): every unordered {a, b}
// has exactly one owner and the owners' loads are balanced. For each block of
// 64 candidate steps the lane first builds a 64-bit mask of the partners with
// r_ab < acut (cheap: 3 subtractions, 3 FMAs, a compare), then runs the
// expensive body only over the set bits, so the ~55 % of triples whose cutoff
// factor vanishes cost almost nothing. In the backward kernel the partner's
// share of dE/dD_b goes through ds_add_f64 into an LDS accumulator (3 adds per
// triple); the own share stays in registers.
//
// Cutoff x Gaussian of the third side. For the cosine cutoff the factor
// Hd(u) = exp(-beta u) fc(u), u = r_jk^2 / acut^2, is an entire function of u;
// the host expands it in powers of u (long double) and the kernels evaluate
// value and derivative with one Horner sweep whose coefficients sit in scalar
// registers (template parameter HD = number of coefficients; HD = 0 keeps the
// exact exp + cutoff evaluation, used for the polynomial cutoff or large beta).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <type_traits>

#include "ta_device.h"
#include "ta_math.h"
#include "ta_reduce.h"

namespace ta {
namespace {

constexpr int kBlock = 256;  // upper bound; launched with min(cap, 256) lanes
constexpr int kNF = 8;   // backward: {z r2} {x y} {inv_r H} {G species}: four 16-byte records per pair
constexpr int kNFf = 7;  // forward: the last record is the species alone (8 bytes): 1.5 KB of LDS less,
                         // which with the trimmed job counters lets 7 workgroups share a CU instead of 6
// control words of the forward kernel's job mode, as ints: hist[20] start[20] (make_jobs), cstart[20] (first
// pair of every centre of the workgroup), then segl[16][kSegW] (first pair of every (centre, neighbour species)
// segment, + the centre's end) and sslot[16][kSegW] (first partial-sum slot of the segment)
constexpr int kSegW = 6;  // up to 5 species + the end
constexpr int kJobCtlBytes = (60 + 2 * kMaxCentersPerBlock * kSegW) * 4;
// The per-pair partial G4 sums of the forward sweep are only ever summed over a (centre, neighbour species)
// segment, so they are kept per GROUP OF FOUR pairs of a segment: n_local x (cap / 4 + 16 nspec) doubles
// instead of n_local x cap (one element: 22.0 -> 18.0 KB of LDS per workgroup, 8 instead of 7 workgroups per
// CU -- with 6 wavefronts per SIMD all 2006 workgroups of the 4000-atom frame are resident at once; two
// elements: 28.7 -> 22.3 KB).
__host__ __device__ inline int v2_pcols(int cap, int nspec) {
  return nspec >= 1 ? cap / 4 + kMaxCentersPerBlock * nspec : cap;
}
constexpr int kRingPad = 64;  // floats readable past the last ring (masked candidates)
constexpr int kRTab = 8;      // backward: radial dE/dG values per centre kept in LDS (neighbour species x radial channels)

// LDS experiment switches (scripts/lds_probe.sh, -DTA_LDS_PROBE builds only; wrong results by
// construction): TA_DEBUG_SKIP bit 4 sends the partner atomics of the backward body to conflict-free
// addresses, bit 5 its late partner reads, bit 6 every partner read (and drops the exact u < 1 test)
#ifdef TA_LDS_PROBE
__device__ __forceinline__ bool kProbe(int flags, int bit) { return (flags >> bit) & 1; }
#else
__device__ __forceinline__ constexpr bool kProbe(int, int) { return false; }
#endif

#ifdef TA_PHASE_STAMPS
#define TA_STAMP(b, kernel, k)                                                                          \
  do {                                                                                                  \
    if ((b).stamps && threadIdx.x == 0)                                                                 \
      (b).stamps[((size_t)(kernel) * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define TA_STAMP(b, kernel, k) do {} while (0)
#endif

__device__ __forceinline__ int angular_term2(int s1, int s2, int nel) {
  int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
  return a * nel - (a * (a - 1)) / 2 + (b - a);
}
__device__ __forceinline__ int radial_term2(int center, int other) {
  return other == center ? 0 : (other < center ? other + 1 : other);
}

// The fields of a pair as four arrays of 16-byte records: the triple bodies read a random partner's
// geometry with four 128-bit LDS reads ({x y}, {z r2}, {1/r H}, {G species}) instead of eight 64-bit
// and one byte read (LDS instructions, not arithmetic, bound the backward body).
struct Fields {
  double2 *zr, *xy, *ih, *gs;  // {z, r^2}, {x, y}, {1/r, H}, {L = (dH/dr) / (r H), species of the neighbour as a double}
  double *sp1;                 // forward kernel: species alone (gs is null there)
  float *xf, *yf, *zf;  // single-precision ring copies for the candidate scan: the n neighbours of
                        // a centre are stored twice in a row (2 * base + k and + n), so partner
                        // a + s needs no wrap-around arithmetic
};

__device__ __forceinline__ Fields carve(double *lds, int cap, bool forward = false) {
  Fields f;
  // {z r2} first: xy .. gs / sp1 (6 or 5 cap doubles) and the rings behind them are one contiguous region that
  // is dead once the triple body is done (the descriptor assembly reuses it, see reduce_from_lds)
  f.zr = reinterpret_cast<double2 *>(lds);
  f.xy = f.zr + cap;
  f.ih = f.xy + cap;
  f.gs = forward ? nullptr : f.ih + cap;
  f.sp1 = forward ? reinterpret_cast<double *>(f.ih + cap) : nullptr;
  // the single-precision rings exist in the forward kernel only: the backward kernel never scans (it
  // takes the forward's job list, or its candidate masks from `b.masks`)
  f.xf = forward ? reinterpret_cast<float *>(f.sp1 + cap) : nullptr;
  f.yf = forward ? f.xf + (2 * cap + kRingPad) : nullptr;
  f.zf = forward ? f.yf + (2 * cap + kRingPad) : nullptr;
  return f;
}

// Hd(u) = exp(-beta u) fc(u) and dHd/du. HD > 0: Horner with derivative on the
// host-expanded coefficients; HD == 0: exact evaluation.
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

// `geom` != 0: the pair geometry D = Rj - Ri + S.h, r^2 = D.D + eps, 1/r
// (reference calculate_rij, transformer/universal.py:448-474) is computed here
// and the pair record written for the later kernels; otherwise it is read.
// `forward`: the forward kernel has no use for G = (dH/dr)/r (Fields::sp1 instead of Fields::gs).
// The pair's own factor H(u) = exp(-beta u) fc(u), u = r^2 / acut^2, is the SAME function as the
// third side's Hd(u): with the power series (HD > 0) value and derivative cost one Horner sweep (22
// fused multiply-adds; 11 in the forward kernel, which needs the value only) instead of the cutoff
// series + an exponential (about 50). The backward kernel stages the logarithmic derivative
// L = (dH/dr) / (r H) instead of G = (dH/dr) / r: its triple body then factors T = Ha Hb Hd out of all
// three derivatives (see there). H = 0 (pair beyond acut) gives L = 0; such a pair has no triples.
template <int HD>
__device__ __forceinline__ void stage(const SFParams &sf, const AngChunk &ch, double beta, const DeviceBatch &b,
                                      const Fields &f, int s0, int M, int geom = 0, bool forward = false) {
  for (int item = threadIdx.x; item < M; item += blockDim.x) {
    double2 v0, v1, v2;
    if (geom) {
      const int64_t p = (int64_t)s0 + item;
      const int i = b.pair_i[p], j = b.pair_j[p];
      const double *h = b.cells + 9 * (size_t)b.frame_of_atom[i];
      const double sx = (double)b.pair_shift[3 * p], sy = (double)b.pair_shift[3 * p + 1],
                   sz = (double)b.pair_shift[3 * p + 2];
      const double *ri = b.pos + 3 * (size_t)i, *rj = b.pos + 3 * (size_t)j;
      const double dx = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
      const double dy = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
      const double dz = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
      const double r2 = dx * dx + dy * dy + dz * dz + sf.eps;
      v0 = make_double2(dx, dy);
      v1 = make_double2(dz, r2);
      v2 = make_double2(1.0 / sqrt(r2), 0.0);
      if (b.rec4) {  // compact record: 1 / r is recomputed by the readers
        double2 *dst = reinterpret_cast<double2 *>(b.rec4 + 4 * (size_t)p);
        dst[0] = v0;
        dst[1] = v1;
      } else {
        double2 *dst = reinterpret_cast<double2 *>(b.rec + kRecDoubles * (size_t)p);
        dst[0] = v0;
        dst[1] = v1;
        dst[2] = v2;
        dst[3] = make_double2(0.0, 0.0);
      }
    } else if (b.rec4) {
      const double2 *src = reinterpret_cast<const double2 *>(b.rec4 + 4 * (size_t)(s0 + item));
      v0 = src[0];
      v1 = src[1];
      v2 = make_double2(1.0 / sqrt(v1.y), 0.0);  // the same expression the writer used
    } else {
      const double2 *src = reinterpret_cast<const double2 *>(b.rec + kRecDoubles * (size_t)(s0 + item));
      v0 = src[0];
      v1 = src[1];
      v2 = src[2];
    }
    f.xy[item] = v0;
    f.zr[item] = v1;
    if (forward) {
      const int ci = b.pair_i[s0 + item];
      const int cbase = b.pair_start[ci] - s0, cn = pair_stop_of(b, ci) - b.pair_start[ci];
      const int k0 = 2 * cbase + (item - cbase);
      f.xf[k0] = f.xf[k0 + cn] = (float)v0.x;
      f.yf[k0] = f.yf[k0 + cn] = (float)v0.y;
      f.zf[k0] = f.zf[k0 + cn] = (float)v1.x;
    }
    const double u = v1.y * sf.inv_ac2;
    double H = 0.0, L = 0.0;
    if (u < 1.0) {
      if (forward) {
        H = hd_value<HD>(sf, ch, beta, u);
      } else {
        double dH;
        hd_eval<HD>(sf, ch, beta, u, H, dH);
        L = (H != 0.0) ? sf.two_inv_ac2 * dH / H : 0.0;
      }
    }
    f.ih[item] = make_double2(v2.x, H);
    if (forward) f.sp1[item] = (double)b.species[b.pair_j[s0 + item]];
    else f.gs[item] = make_double2(L, (double)b.species[b.pair_j[s0 + item]]);
  }
  __syncthreads();
}

// Candidate mask of the partners a + s, s in [sc, sc + 63]. The scan runs in
// single precision (half the issue cost of fp64, 4-byte LDS reads at immediate
// offsets from the lane's ring position, fully unrolled in groups of 16) with
// a relative margin far above the fp32 rounding of r_jk^2; the double-precision
// test u < 1 is repeated on every candidate before it contributes, so the mask
// only has to be a superset.
__device__ __forceinline__ unsigned long long partner_mask(const SFParams &sf, const Fields &f,
                                                           int base, int n, int a, int sc, int smax) {
  const int ring = 2 * base + a;
  const float ax = f.xf[ring], ay = f.yf[ring], az = f.zf[ring];
  const float lim = (float)(sf.acut * sf.acut) * 1.0001f;
  unsigned long long mask = 0ull;
  const int count = smax - sc + 1;  // candidates wanted in this block of 64 (may exceed 64)
  // two candidates per instruction: gfx950 has packed fp32 add / mul / fma (v_pk_*_f32), so the
  // difference vectors and the squared distances of candidates k, k + 1 are formed together
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
  for (int g = 0; g < 4 && 16 * g < count; ++g) {
    const float *px = f.xf + ring + sc + 16 * g, *py = f.yf + ring + sc + 16 * g,
                *pz = f.zf + ring + sc + 16 * g;
    unsigned m = 0u;
    // candidates from the highest position down, m = 2 m + (d2 < lim): the comparison's lane mask goes
    // straight into the carry input of an add (v_cmp + v_addc_co, two instructions per candidate
    // instead of compare / select / or)
#pragma unroll
    for (int k = 14; k >= 0; k -= 2) {
      const f32x2 vx = {px[k], px[k + 1]}, vy = {py[k], py[k + 1]}, vz = {pz[k], pz[k + 1]};
      const f32x2 ex = vx - ax2, ey = vy - ay2, ez = vz - az2;
      const f32x2 d2 = __builtin_elementwise_fma(ex, ex, __builtin_elementwise_fma(ey, ey, ez * ez));
      const float dhi = d2.y, dlo = d2.x;
      asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(dhi), "v"(lim) : "vcc");
      asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(dlo), "v"(lim) : "vcc");
    }
    mask |= (unsigned long long)m << (16 * g);
  }
  if (count < 64) mask &= (1ull << count) - 1ull;
  // even n: the antipodal partner is shared by two lanes, the lower one keeps it
  const int half = n >> 1;
  if (!(n & 1) && a >= half && half >= sc && half < sc + 64) mask &= ~(1ull << (half - sc));
  return mask;
}

// Lane balance. The expensive triple body runs once per set bit of a lane's candidate mask, and
// a wavefront runs as long as its busiest lane: with lanes in pair order the mean lane has 20 set
// bits but the mean wavefront maximum is 37. When a workgroup holds one pair per lane and a single
// scan pass (n / 2 <= 64), the (pair, mask) jobs are therefore re-dealt to the lanes in
// descending popcount order (counting sort through LDS), so the lanes of a wavefront finish
// together. Which lane runs a pair does not change what is computed for it.
// Scratch aliases the single-precision rings, which are dead once the masks exist.
__device__ __forceinline__ void deal_by_popcount(const Fields &f, int cap, int M, int &item,
                                                 unsigned long long &mask) {
  int *hist = reinterpret_cast<int *>(f.xf);                      // [65] (+ pad)
  int *start = hist + 66;                                        // [65]
  unsigned long long *pmask = reinterpret_cast<unsigned long long *>(hist + 132);  // [cap]
  int *pitem = reinterpret_cast<int *>(pmask + cap);             // [cap]
  const int tid = threadIdx.x;
  const bool active = tid < M;
  __syncthreads();  // every lane is done with the rings
  if (tid < 66) hist[tid] = 0;
  __syncthreads();
  const int bucket = 64 - (active ? __popcll(mask) : 0);
  int rank = 0;
  if (active) rank = atomicAdd(&hist[bucket], 1);
  __syncthreads();
  if (tid < 65) {
    int s = 0;
    for (int k = 0; k < tid; ++k) s += hist[k];
    start[tid] = s;
  }
  __syncthreads();
  if (active) {
    const int slot = start[bucket] + rank;
    pitem[slot] = item;
    pmask[slot] = mask;
  }
  __syncthreads();
  if (active) {
    item = pitem[tid];
    mask = pmask[tid];
  }
}

// Finer lane balance than the re-dealing above: JOBS (see make_jobs). CPU simulation on the benchmark
// frame: 73 % (re-dealing) -> 91-94 % busy lanes in the triple loops. The list aliases the fp32 rings
// (dead once the masks exist), is written to HBM by the forward kernel and read back by the backward
// kernel: one construction serves both sweeps.
struct JobLists {
  uint32_t *word;            // [4 * lanes]: pair | window << 8 | centre of the pair in the workgroup << 10 | bits << 16
  int *hist, *start;         // [20], [20] (17 used: job sizes 16 .. 1, total)
};

__device__ __forceinline__ JobLists job_lists(const Fields &f) {
  JobLists j;
  j.word = reinterpret_cast<uint32_t *>(f.xf);  // the fp32 rings are dead once the masks exist
  return j;
}
__device__ __forceinline__ int job_item(uint32_t w) { return (int)(w & 255u); }
// a job names its centre (index in the workgroup's run, < kMaxCentersPerBlock = 16), so that the sweeps
// take the centre's first pair and neighbour count from a table in LDS (`cstart`) instead of the chain
// of dependent global loads pair_i[p] -> pair_start[i], pair_start[i + 1]
__device__ __forceinline__ int job_centre(uint32_t w) { return (int)((w >> 10) & 15u); }
static_assert(kMaxCentersPerBlock <= 16, "four bits of a job word name the centre");

// byte offset of the job counters (kJobCtlBytes) and, behind
// them, of the forward kernel's per-pair partial sums P[n_local][cap]
__host__ __device__ inline size_t v2_counter_offset(int cap) {
  const size_t end = (size_t)cap * kNFf * sizeof(double) + 3 * (size_t)(2 * cap + kRingPad) * sizeof(float);
  return (end + 15) & ~(size_t)15;
}
// four jobs per lane at most; the words fit the rings: 4 * 4 cap <= 3 * 4 (2 cap + 64) bytes
__host__ __device__ inline int v2_max_jobs(int cap) { return 4 * cap; }

// `j.hist` must be zero on entry (the kernel clears it before staging).
// Jobs by POSITION: a lane's 64-bit candidate mask is cut into four windows of 16 positions; every
// non-empty window is one job (a 32-bit word). Sorted by size, largest first (counting sort over
// the popcounts: one LDS atomic per job, one wavefront scans the 16 buckets); the sweep deals them
// to the lanes in serpentine rounds (job_slot), so a wavefront holds jobs of nearly equal size in
// every round and the wavefronts of a workgroup get large and small rounds alike. Replaces the cut
// into jobs of exactly K = 8 set bits, whose bit-peeling loops and prefix sums were 2.5 M of the
// forward kernel's 17.4 M VALU instructions, and its 10-byte (mask, code) records.
__device__ __forceinline__ int make_jobs(JobLists &j, bool active, int item, int centre, unsigned long long m0) {
  const int tid = threadIdx.x;
  int cnt[4] = {0, 0, 0, 0}, rank[4] = {0, 0, 0, 0};
  if (active) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = __popc((unsigned)((m0 >> (16 * w)) & 0xffffull));
      cnt[w] = c;
      if (c) rank[w] = atomicAdd(&j.hist[16 - c], 1);
    }
  }
  __syncthreads();  // also: every lane is done with the rings (its mask exists)
  if (tid < 64) {
    const int v = tid < 16 ? j.hist[tid] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int u = __shfl_up(incl, off);
      if (tid >= off) incl += u;
    }
    if (tid < 16) j.start[tid] = incl - v;
    if (tid == 15) j.start[16] = incl;
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (cnt[w]) {
        const int slot = j.start[16 - cnt[w]] + rank[w];
        j.word[slot] = (uint32_t)item | ((uint32_t)w << 8) | ((uint32_t)centre << 10) |
                       ((uint32_t)((m0 >> (16 * w)) & 0xffffull) << 16);
      }
  }
  const int n = j.start[16];
  __syncthreads();
  return n;
}

// job of lane `tid` in round `r` (T lanes): serpentine over the size-sorted list
__device__ __forceinline__ int job_slot(int r, int tid, int T) { return r * T + ((r & 1) ? T - 1 - tid : tid); }

// Descriptor vectors of the workgroup's centres from data that is still in LDS, one wavefront per
// centre, round robin. G2 from r^2 (sf.py:79-119):
__device__ __forceinline__ void reduce_radial_from_lds(const SFParams &sf, const DeviceBatch &b,
                                                       const Fields &f, int c0, int c1, int s0) {
  const int nel = sf.n_elements;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  // the usual grid has omega = 0 for every channel: exp(-eta r^2 / rc^2), no square root (uniform test)
  bool no_shift = true;
  for (int c = 0; c < sf.n_rad; ++c) no_shift = no_shift && sf.omega[c] == 0.0;
  for (int64_t i = c0 + w; i < c1; i += nwaves) {
    const int sA = b.species[i];
    const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
    double *Gi = b.G + (size_t)i * sf.ndim;
    for (int sb = 0; sb < nel; ++sb) {
      const int tr = radial_term2(sA, sb);
      const int q0 = seg[sb] - s0, q1 = seg[sb + 1] - s0;
      for (int cc = 0; cc < sf.n_rad; cc += 4) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int q = q0 + l; q < q1; q += 64) {
          const double r2 = f.zr[q].y;
          const double u = r2 * sf.inv_rc2;
          const double fc = (u < 1.0) ? cutoff_u_value(sf.cutoff, u) : 0.0;
          if (no_shift) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int c = (cc + k < sf.n_rad) ? cc + k : cc;
              acc[k] += ta_exp(-sf.eta[c] * u) * fc;  // sf.py:101-108 with omega = 0
            }
          } else {
            const double r = sqrt(r2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int c = (cc + k < sf.n_rad) ? cc + k : cc;
              const double dr = r - sf.omega[c];
              acc[k] += ta_exp(-sf.eta[c] * dr * dr * sf.inv_rc2) * fc;  // sf.py:101-108
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double v = wave_sum(acc[k]);
          if (l == 0 && cc + k < sf.n_rad) Gi[tr * sf.n_rad + cc + k] = v;
        }
      }
    }
  }
}

// G4 (sf.py:121-182, :184-215) from the per-pair partial sums the lanes left in LDS for the partner
// species [sp_lo, sp_hi): red[((sp - sp_lo) * NG + ig) * NZ + iz][item]. The term {A, B} (A < B) gets
// the pairs of species B paired with partners A and the pairs of species A paired with partners B;
// with the partner loop outermost and ascending, the first of the two always comes first (also
// across passes), so it stores and the second adds.
template <int NSPEC, int NG, int NZ>
__device__ __forceinline__ void reduce_angular_from_lds(const SFParams &sf, const AngChunk &ch,
                                                        const DeviceBatch &b, const double *red, int cap,
                                                        int c0, int c1, int s0, int sp_lo, int sp_hi) {
  const int nel = sf.n_elements;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  for (int64_t i = c0 + w; i < c1; i += nwaves) {
    const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
    double *Gi = b.G + (size_t)i * sf.ndim + sf.n_radial_dim;
    for (int sp = sp_lo; sp < sp_hi; ++sp)
      for (int sg = 0; sg < NSPEC; ++sg) {
        const int t = angular_term2(sg, sp, nel);
        const int q0 = seg[sg] - s0, q1 = seg[sg + 1] - s0;
#pragma unroll
        for (int gz = 0; gz < NG * NZ; ++gz) {
          double v = 0.0;
          const double *col = red + (size_t)((sp - sp_lo) * NG * NZ + gz) * cap;
          for (int q = q0 + l; q < q1; q += 64) v += col[q];
          v = wave_sum(v) * sf.ang_scale;
          if (l == 0) {
            double *dst = Gi + t * sf.n_ang + ch.chan[gz];
            *dst = (sg >= sp) ? v : *dst + v;
          }
        }
      }
  }
}

// Job mode: the same two assemblies with every lane busy. The G2 terms are evaluated one PAIR per
// lane (a wavefront per centre walks ~90 pairs in two passes of 64, 70 % of its lanes; the
// workgroup's ~180 pairs fill 94 % of three wavefronts) and left in LDS next to the G4 partial sums;
// then every 16-lane row sums one (centre, channel) column (DPP row rotations instead of cross-row
// shuffles): about 350 of the 950 wavefront instructions of the two assemblies for one element. The
// angular term {A, B}, A < B, is the column of partner species A over the pairs of species B plus the
// column of partner species B over the pairs of species A, summed by the same row (one store).
template <int NSPEC, int NG, int NZ>
__device__ __forceinline__ void assemble_flat(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b,
                                              const Fields &f, const double *P, const int *sslot, int c0, int c1,
                                              int s0, int item, bool active, bool skip_radial, int cap) {
  constexpr int kGZ = NG * NZ;
  constexpr int kTerms = NSPEC * (NSPEC + 1) / 2;
  double *R = reinterpret_cast<double *>(f.xy);  // xy, ih, sp1 (5 cap doubles) are dead after the sweep
  const int row = threadIdx.x >> 4, l = threadIdx.x & 15, nrows = blockDim.x >> 4;
  const int ncent = c1 - c0;
  bool no_shift = true;
  for (int c = 0; c < sf.n_rad; ++c) no_shift = no_shift && sf.omega[c] == 0.0;
  const double r2 = active ? f.zr[item].y : 0.0;
  const int n_rad = skip_radial ? 0 : sf.n_rad;
  for (int cc = 0; cc == 0 || cc < n_rad; cc += 4) {
    if (cc) __syncthreads();  // the previous chunk's columns have been summed
    if (active && cc < n_rad) {
      const double u = r2 * sf.inv_rc2;
      const double fc = (u < 1.0) ? cutoff_u_value(sf.cutoff, u) : 0.0;
      const double r = no_shift ? 0.0 : sqrt(r2);
      double ev[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = (cc + k < n_rad) ? cc + k : cc;
        const int pw = (cc + k < n_rad && k > 0) ? sf.eta_pow[c] : 0;  // wave-uniform
        if (pw > 0) {
          ev[k] = pow_int_m1(ev[k > 0 ? k - 1 : 0], pw) * ev[k > 0 ? k - 1 : 0];  // SFParams::eta_pow
        } else {
          double arg = sf.eta[c] * u;
          if (!no_shift) {
            const double dr = r - sf.omega[c];
            arg = sf.eta[c] * dr * dr * sf.inv_rc2;
          }
          ev[k] = ta_exp(-arg);
        }
        R[k * cap + item] = ev[k] * fc;  // sf.py:101-108
      }
    }
    __syncthreads();
    const int nrad_here = cc < n_rad ? (n_rad - cc < 4 ? n_rad - cc : 4) : 0;
    const int n_radial_tasks = NSPEC * nrad_here;
    const int per_centre = n_radial_tasks + (cc == 0 ? kTerms * kGZ : 0);
    for (int t = row; t < ncent * per_centre; t += nrows) {
      const int ci = t / per_centre, k = t - ci * per_centre;
      const int64_t i = c0 + ci;
      const int32_t *seg = b.seg_start + (size_t)i * (NSPEC + 1);
      double *Gi = b.G + (size_t)i * sf.ndim;
      double v = 0.0;
      if (k < n_radial_tasks) {
        const int sb = NSPEC == 1 ? 0 : k / nrad_here, kk = k - sb * nrad_here;
        const double *col = R + (size_t)kk * cap;
        for (int q = seg[sb] - s0 + l; q < seg[sb + 1] - s0; q += 16) v += col[q];
        v = row16_sum(v);
        if (l == 0) Gi[radial_term2(b.species[i], sb) * sf.n_rad + cc + kk] = v;
      } else {
        const int k2 = k - n_radial_tasks;
        const int term = k2 / kGZ, gz = k2 - term * kGZ;
        int sa = 0, sb = 0;  // term index -> (sa <= sb), row-major upper triangle
        for (int rem = term, len = NSPEC; rem >= len; rem -= len, --len) ++sa;
        sb = term - (sa * NSPEC - (sa * (sa - 1)) / 2) + sa;
        // one partial-sum slot per four pairs of a (centre, neighbour species) segment (v2_pcols, sslot)
        const int *ss = sslot + ci * kSegW;
        const double *colA = P + (size_t)(sa * kGZ + gz) * v2_pcols(cap, NSPEC);  // partner species sa, pairs of sb
        for (int q = ss[sb] + l; q < ss[sb + 1]; q += 16) v += colA[q];
        if (sa != sb) {
          const double *colB = P + (size_t)(sb * kGZ + gz) * v2_pcols(cap, NSPEC);
          for (int q = ss[sa] + l; q < ss[sa + 1]; q += 16) v += colB[q];
        }
        v = row16_sum(v);
        if (l == 0) Gi[sf.n_radial_dim + term * sf.n_ang + ch.chan[gz]] = v * sf.ang_scale;
      }
    }
  }
}

// Phase stagger (flags bits 8..15 = sleep count, bits 16..20 = shift): a launch whose workgroups
// are all resident at once runs its memory-bound staging and its VALU-bound triple loop in lock
// step; holding back every other group of workgroups for a few microseconds lets the two phases of
// different workgroups overlap on a CU, as they do by themselves in many-frame batches.
__device__ __forceinline__ void stagger(int flags) {
  const int n = (flags >> 8) & 0xff;
  if (n && ((blockIdx.x >> ((flags >> 16) & 31)) & 1))
    for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(127);
}

// DEFZ: zeta = {1, 4} known at compile time (the reference's default grid,
// nn/atomic/sf.py:37), so the powers are two multiplications, no scalar loops.
// CAP: the records a workgroup stages (b.cap) as a compile-time constant, 0 = take it from the batch.
// With CAP known the LDS arrays sit at constant offsets from each other, so the random partner reads
// and the accumulator updates of the triple bodies address them as one shifted index + immediate
// offsets instead of one address computation each (six VALU instructions per backward triple).
// WPE: wavefronts per SIMD asked of the register allocator (0 = the compiler's choice)
template <int NSPEC, int NG, int NZ, int HD, bool DEFZ, int CAP, int WPE = (DEFZ && NSPEC <= 2 ? 5 : 0)>
__global__ __launch_bounds__(kBlock)
    __attribute__((amdgpu_waves_per_eu(WPE > 0 ? WPE : 1, 8))) void g4_forward_v2_kernel(SFParams sf, AngChunk ch,
                                                               DeviceBatch b, int flags) {
  static_assert(!DEFZ || NZ == 2, "DEFZ needs the two-zeta grid");
  const int kCap = CAP > 0 ? CAP : b.cap;
  const int geom = flags & 1;
  if (b.n_blk_dev && (int)blockIdx.x >= *b.n_blk_dev) return;  // grid sized by an upper bound (MD loop)
  stagger(flags);
  extern __shared__ double lds[];
  const Fields f = carve(lds, kCap, true);
  const int slot = b.blk_groups > 0 ? 16 * ((int)blockIdx.x % b.blk_groups) + (int)blockIdx.x / b.blk_groups
                                    : (int)blockIdx.x;
  const int c0 = b.blk_center[slot], c1 = b.blk_center[slot + 1];
  if (c0 >= c1) return;  // an unused run slot of the MD step's packing (ta_nlist.hip::filter_group_kernel)
  const int s0 = b.pair_start[c0];
  const int M = pair_stop_of(b, c1 - 1) - s0;  // a workgroup's centres are contiguous in the list
  const double beta = ch.beta[0];
  if (b.job_count) {  // job counters and partial sums: cleared before the staging barrier
    char *raw = reinterpret_cast<char *>(lds);
    int *cnt = reinterpret_cast<int *>(raw + v2_counter_offset(kCap));
    if (threadIdx.x < 40) cnt[threadIdx.x] = 0;
    else if (threadIdx.x < 40 + 17 && (int)threadIdx.x - 40 <= c1 - c0)  // cstart[k]: first pair of centre c0 + k
      cnt[threadIdx.x] = ((int)threadIdx.x - 40 < c1 - c0 ? b.pair_start[c0 + threadIdx.x - 40] : s0 + M) - s0;
    else if (threadIdx.x >= 64 && (int)threadIdx.x - 64 < (c1 - c0) * (NSPEC + 1)) {  // segl[k][sg]
      const int k = ((int)threadIdx.x - 64) / (NSPEC + 1), sg = ((int)threadIdx.x - 64) % (NSPEC + 1);
      cnt[60 + k * kSegW + sg] = b.seg_start[(size_t)(c0 + k) * (NSPEC + 1) + sg] - s0;
    }
    double *P0 = reinterpret_cast<double *>(raw + v2_counter_offset(kCap) + kJobCtlBytes);
    for (int k = threadIdx.x; k < NSPEC * NG * NZ * v2_pcols(kCap, NSPEC); k += blockDim.x) P0[k] = 0.0;
  }
  TA_STAMP(b, 0, 0);
  stage<HD>(sf, ch, beta, b, f, s0, M, geom, true);
  TA_STAMP(b, 0, 1);
  if (b.job_count && threadIdx.x == 0) {
    // sslot[k][sg]: first partial-sum slot of segment (centre c0 + k, neighbour species sg), one slot per
    // four pairs (v2_pcols); read by the sweep and the assembly, both behind the barriers of make_jobs
    int *ctl = reinterpret_cast<int *>(reinterpret_cast<char *>(lds) + v2_counter_offset(kCap));
    const int *segl = ctl + 60;
    int *sslot = ctl + 60 + kMaxCentersPerBlock * kSegW;
    int acc = 0;
    for (int k = 0; k < c1 - c0; ++k)
      for (int sg = 0; sg <= NSPEC; ++sg) {
        sslot[k * kSegW + sg] = acc;
        if (sg < NSPEC) acc += (segl[k * kSegW + sg + 1] - segl[k * kSegW + sg] + 3) >> 2;
      }
  }

  // one job = one directed pair (i, a); `have_mask`: the single scan pass was done up front
  // `out`: null = store the partial sums in part4 (global), else hand them back to the caller
  // `pacc`: job mode, the sums of this (pair, partner subset) are ADDED to P[channel][item] in LDS
  // `is_job` (std::true_type): `mask0` is a job word (see make_jobs): 16 candidate bits and their window,
  // walked with 32-bit operations and without the loop over blocks of 64 candidate steps
  auto run_item = [&](int item, bool have_mask, unsigned long long mask0, double *out, double *pacc, auto is_job) {
    constexpr bool kJob = decltype(is_job)::value;
    const int64_t p = (int64_t)s0 + item;
    int base, n;
    int pslot = item;  // column index of this pair's partial sums in `pacc`
    if constexpr (kJob) {
      const int *cstart = reinterpret_cast<const int *>(reinterpret_cast<const char *>(lds) + v2_counter_offset(kCap)) + 40;
      const int ci = job_centre((uint32_t)mask0);
      base = cstart[ci];
      n = cstart[ci + 1] - base;
      {
        const int *segl = cstart + 20, *sslot = segl + kMaxCentersPerBlock * kSegW;
        const int sg = NSPEC == 1 ? 0 : (int)f.sp1[item];
        pslot = sslot[ci * kSegW + sg] + ((item - segl[ci * kSegW + sg]) >> 2);
      }
    } else {
      const int i = b.pair_i[p];
      base = b.pair_start[i] - s0;
      n = pair_stop_of(b, i) - b.pair_start[i];
    }
    const int a = item - base;
    const double2 axy = f.xy[item], azr = f.zr[item], aih = f.ih[item];
    const double ax = axy.x, ay = axy.y, az = azr.x;
    const double ra2 = azr.y, inv_ra = aih.x, Ha = aih.y;

    double acc[NSPEC][NG][NZ];
#pragma unroll
    for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) acc[sp][ig][iz] = 0.0;
    // DEFZ: zeta = {1, 4}: every channel is a polynomial of degree <= 4 in cos(theta),
    //   (1 + g c) and (1 + g c)^4 = sum_k C(4, k) g^k c^k,
    // so the lane accumulates the five moments m_k = sum common c^k per partner species (8 operations
    // per triple instead of 14) and expands them into the channels once, at the end
    double mom[NSPEC][5];
#pragma unroll
    for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
      for (int k = 0; k < 5; ++k) mom[sp][k] = 0.0;

    // one candidate partner: position a + step of the centre's ring
    auto triple = [&](int bl) {
        bl = (int)min((unsigned)bl, (unsigned)bl - (unsigned)n);  // bl < 2 n: bl mod n without a compare / select
        const int q = kProbe(flags, 30) ? (int)threadIdx.x : base + bl;
        const double2 bxy = f.xy[q], bzr = f.zr[q];
        const double ex = bxy.x - ax, ey = bxy.y - ay, ez = bzr.x - az;
        const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
        const double u = kProbe(flags, 30) ? 0.5 : d2 * sf.inv_ac2;
        if (!(u < 1.0)) return;  // exact test (the mask is a superset); H_b = 0 adds nothing
        const double2 bih = f.ih[kProbe(flags, 29) ? (int)threadIdx.x : q];
        const double cth = (ra2 + bzr.y - d2) * 0.5 * inv_ra * bih.x;
        const double common = Ha * bih.y * hd_value<HD>(sf, ch, beta, u);
        const int sb = NSPEC == 1 ? 0 : (int)f.sp1[q];
        if constexpr (DEFZ) {
          const double c2 = cth * cth;
          const double t1 = common * cth, t2 = common * c2, t3 = t1 * c2, t4 = t2 * c2;
#pragma unroll
          for (int sp = 0; sp < NSPEC; ++sp) {
            const bool on = NSPEC == 1 || sb == sp;
            mom[sp][0] += on ? common : 0.0;
            mom[sp][1] += on ? t1 : 0.0;
            mom[sp][2] += on ? t2 : 0.0;
            mom[sp][3] += on ? t3 : 0.0;
            mom[sp][4] += on ? t4 : 0.0;
          }
        } else {
#pragma unroll
          for (int ig = 0; ig < NG; ++ig) {
            const double basev = fma(ch.gamma[ig], cth, 1.0);
#pragma unroll
            for (int iz = 0; iz < NZ; ++iz) {
              double pw;
              if (ch.zeta_int[iz] > 0)
                pw = pow_int_m1(basev, ch.zeta_int[iz]) * basev;
              else
                pw = safe_pow_value(ch.safe_pow, basev, ch.zeta[iz]);
              const double v = pw * common;
#pragma unroll
              for (int sp = 0; sp < NSPEC; ++sp)
                acc[sp][ig][iz] += (NSPEC == 1 || sb == sp) ? v : 0.0;
            }
          }
        }
    };
    if constexpr (kJob) {
      const uint32_t jw = (uint32_t)mask0;
      unsigned m = (flags & (1 << 24)) ? 0u : jw >> 16;  // bit 24: measurement switch, triple bodies off
      const int k0 = a + 1 + 16 * (int)((jw >> 8) & 3u);
      while (m) {
        const int k = __ffs((int)m) - 1;
        m &= m - 1;
        triple(k0 + k);
      }
    } else {
    const int smax = (Ha != 0.0) ? n / 2 : 0;
    for (int sc = 1; sc <= smax; sc += 64) {
      unsigned long long mask;
      if (have_mask) {
        mask = (flags & (1 << 24)) ? 0ull : mask0;
      } else {
        mask = partner_mask(sf, f, base, n, a, sc, smax);
        // the candidate mask is geometry only: keep it for the backward kernel
        if (b.masks) b.masks[(size_t)(sc >> 6) * b.n_pairs + p] = mask;
      }
      while (mask) {
        const int k = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        triple(a + sc + k);
      }
    }
    }
    if constexpr (DEFZ) {
#pragma unroll
      for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) {
          const double g1 = ch.gamma[ig], g2 = g1 * g1;
          acc[sp][ig][0] = fma(g1, mom[sp][1], mom[sp][0]);
          acc[sp][ig][1] = fma(g2 * g2, mom[sp][4], fma(4.0 * g2 * g1, mom[sp][3],
                               fma(6.0 * g2, mom[sp][2], fma(4.0 * g1, mom[sp][1], mom[sp][0]))));
        }
    }
#pragma unroll
    for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) {
          const int c = ch.chan[ig * NZ + iz];
          const double v = acc[sp][ig][iz] * ch.kz[iz];
          if (pacc) {
            if (v != 0.0) atomicAdd(&pacc[(size_t)((sp * NG + ig) * NZ + iz) * v2_pcols(kCap, NSPEC) + pslot], v);
          } else if (out)
            out[(sp * NG + ig) * NZ + iz] = v;
          else
            b.part4[(size_t)(sp * sf.n_ang + c) * b.n_pairs + p] = v;
        }
  };

  // balanced path: one pair per lane, one scan pass for every centre of the workgroup
  int item = threadIdx.x;
  const bool active = item < M;
  int n_own = 0;
  if (active) {
    const int i = b.pair_i[s0 + item];
    n_own = pair_stop_of(b, i) - b.pair_start[i];
  }
  const bool one_pass = M <= (int)blockDim.x && !__syncthreads_or(active && (n_own >> 1) > 64);
  if (one_pass) {
    unsigned long long mask = 0ull;
    if (active) {
      const int64_t p = (int64_t)s0 + item;
      const int i = b.pair_i[p];
      const int base = b.pair_start[i] - s0;
      const int smax = (f.ih[item].y != 0.0) ? n_own / 2 : 0;
      if (smax > 0 && !(flags & (1 << 25))) mask = partner_mask(sf, f, base, n_own, item - base, 1, smax);
      if (b.masks && !b.job_count) b.masks[p] = mask;  // (with a job list the backward kernel reads that instead)
    }
    TA_STAMP(b, 0, 2);
    if (b.job_count) {
      // job mode (see make_jobs): build the list once, leave it for the backward kernel, sweep it
      char *raw = reinterpret_cast<char *>(lds);
      JobLists jl = job_lists(f);
      jl.hist = reinterpret_cast<int *>(raw + v2_counter_offset(kCap));
      jl.start = jl.hist + 20;
      double *P = reinterpret_cast<double *>(raw + v2_counter_offset(kCap) + kJobCtlBytes);
      const int n_jobs = make_jobs(jl, active, item, active ? b.pair_i[s0 + item] - c0 : 0, mask);
      TA_STAMP(b, 0, 3);
      const size_t jbase = (size_t)blockIdx.x * b.job_stride;
      if (threadIdx.x == 0) b.job_count[blockIdx.x] = n_jobs;
      for (int slot = threadIdx.x; slot < n_jobs; slot += blockDim.x) b.job_word[jbase + slot] = jl.word[slot];
      TA_STAMP(b, 0, 4);
      if (!(flags & (1 << 27)))
        for (int r = 0; r * (int)blockDim.x < n_jobs; ++r) {
          const int slot = job_slot(r, threadIdx.x, blockDim.x);
          if (slot < n_jobs) {
            const uint32_t jw = jl.word[slot];
            run_item(job_item(jw), true, jw, nullptr, P, std::true_type{});
          }
        }
      __syncthreads();
      TA_STAMP(b, 0, 5);
      if (flags & 4) {
        assemble_flat<NSPEC, NG, NZ>(sf, ch, b, f, P, jl.hist + 60 + kMaxCentersPerBlock * kSegW, c0, c1, s0, item, active,
                                     (flags & (1 << 26)) != 0, kCap);
        TA_STAMP(b, 0, 6);
        return;
      }
      // several forward launches (one per beta): the sums travel through part4 as before
      if (active) {
        const int64_t p = (int64_t)s0 + threadIdx.x;
        // the first pair of every group of four carries the group's sum, the others zero
        const int ci = b.pair_i[p] - c0, sg = NSPEC == 1 ? 0 : (int)f.sp1[threadIdx.x];
        const int *segl = jl.hist + 60, *sslot = segl + kMaxCentersPerBlock * kSegW;
        const int rel = (int)threadIdx.x - segl[ci * kSegW + sg];
        const int pidx = sslot[ci * kSegW + sg] + (rel >> 2);
        const bool holds = (rel & 3) == 0;
#pragma unroll
        for (int sp = 0; sp < NSPEC; ++sp)
#pragma unroll
          for (int gz = 0; gz < NG * NZ; ++gz)
            b.part4[(size_t)(sp * sf.n_ang + ch.chan[gz]) * b.n_pairs + p] =
                holds ? P[(size_t)(sp * NG * NZ + gz) * v2_pcols(kCap, NSPEC) + pidx] : 0.0;
      }
    } else {
    deal_by_popcount(f, kCap, M, item, mask);
    // flags & 4: this launch holds every angular channel of the model: the descriptors are
    // assembled from LDS, without a round trip through part4. The partial sums of up to 8 local
    // channels at a time go behind {z r2}: xy, ih, sp1 (5 cap doubles) and the rings behind them (3 cap
    // doubles) are dead by now; more partner species take more passes.
    constexpr int kLocal = NSPEC * NG * NZ;
    constexpr int kGZ = NG * NZ;
    constexpr int kSpPerPass = 8 / kGZ;
    if (flags & 4) {
      double mine[kLocal];
#pragma unroll
      for (int k = 0; k < kLocal; ++k) mine[k] = 0.0;
      if (active) run_item(item, true, mask, mine, nullptr, std::false_type{});
      __syncthreads();  // nobody reads x .. G or the sort scratch any more
      double *red = reinterpret_cast<double *>(f.xy);
#pragma unroll
      for (int sp_lo = 0; sp_lo < NSPEC; sp_lo += kSpPerPass) {
        const int sp_hi = sp_lo + kSpPerPass < NSPEC ? sp_lo + kSpPerPass : NSPEC;
        if (active) {
#pragma unroll
          for (int k = 0; k < kSpPerPass * kGZ; ++k)
            if (sp_lo * kGZ + k < kLocal) red[(size_t)k * kCap + item] = mine[sp_lo * kGZ + k];
        }
        __syncthreads();
        if (sp_lo == 0) reduce_radial_from_lds(sf, b, f, c0, c1, s0);
        reduce_angular_from_lds<NSPEC, NG, NZ>(sf, ch, b, red, kCap, c0, c1, s0, sp_lo, sp_hi);
        __syncthreads();
      }
      return;
    }
    if (active) run_item(item, true, mask, nullptr, nullptr, std::false_type{});
    }
  } else {
    if (b.job_count && threadIdx.x == 0) b.job_count[blockIdx.x] = -1;  // no list: the backward kernel scans itself
    for (int it = threadIdx.x; it < M; it += blockDim.x) run_item(it, false, 0ull, nullptr, nullptr, std::false_type{});
  }

  // Last forward launch of the evaluation (flags & 2): the workgroup owns whole centres, so it
  // also assembles their descriptor vectors (G2 from the pair records, G4 from the per-pair
  // partial sums of all launches) instead of leaving that to a separate kernel.
  if (flags & 2) {
    __syncthreads();  // this workgroup's rec / part4 stores are visible to all of its lanes
    const int ncent = c1 - c0;
    const int nwaves = blockDim.x >> 6;
    if (ncent <= nwaves) {  // one wavefront per centre
      const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
      const bool act = w < ncent;
      const int64_t i = c0 + (act ? w : 0);
      atom_descriptors<64>(sf, b, i, l, act, [&](int c, double v) { b.G[(size_t)i * sf.ndim + c] = v; });
    } else {                // one 16-lane row per centre, several rounds
      const int row = threadIdx.x >> 4, l = threadIdx.x & 15, nrows = blockDim.x >> 4;
      for (int k0 = 0; k0 < ncent; k0 += nrows) {
        const bool act = k0 + row < ncent;
        const int64_t i = c0 + (act ? k0 + row : 0);
        atom_descriptors<16>(sf, b, i, l, act, [&](int c, double v) { b.G[(size_t)i * sf.ndim + c] = v; });
      }
    }
  }
}

template <int NSPEC, int NG, int NZ, int HD, bool DEFZ, int CAP, int WPE = (DEFZ ? (NSPEC <= 2 ? 5 : 4) : 0)>
// Occupancy: the backward body is latency-bound at the 3 wavefronts per SIMD the compiler settles for
// (133 VGPRs); asking for 5 (96 VGPRs, 10 spilled) measured 72 -> 66 us on the benchmark frame and
// 51 -> 45 us per frame in batches (4: 69 / 47, 6: 68 / 45). Only the default-grid instantiations
// are constrained; the generic ones keep the compiler's choice.
__global__ __launch_bounds__(kBlock)
    __attribute__((amdgpu_waves_per_eu(WPE > 0 ? WPE : 1, 8))) void backward_v2_kernel(SFParams sf, AngChunk ch,
                                                             DeviceBatch b, int flags) {
  static_assert(!DEFZ || NZ == 2, "DEFZ needs the two-zeta grid");
  const int first = flags & 1;
  if (b.n_blk_dev && (int)blockIdx.x >= *b.n_blk_dev) return;  // grid sized by an upper bound (MD loop)
  stagger(flags);
  extern __shared__ double lds[];
  const int kCap = CAP > 0 ? CAP : b.cap;  // multiple of 64
  const Fields f = carve(lds, kCap);
  // partner accumulators behind the fields (kCap bytes in between hold the centres' first pairs, `cstart`)
  double *gacc = lds + kNF * kCap + kCap / 8;
  const int slot = b.blk_groups > 0 ? 16 * ((int)blockIdx.x % b.blk_groups) + (int)blockIdx.x / b.blk_groups
                                    : (int)blockIdx.x;
  const int c0 = b.blk_center[slot], c1 = b.blk_center[slot + 1];
  if (c0 >= c1) return;  // unused run slot
  const int s0 = b.pair_start[c0];
  const int M = pair_stop_of(b, c1 - 1) - s0;
  const double beta = ch.beta[0];
  for (int k = threadIdx.x; k < 3 * kCap; k += blockDim.x) gacc[k] = 0.0;
  // cstart[k]: first pair of centre c0 + k (see job_centre), in the kCap bytes in front of gacc
  int *cstart = reinterpret_cast<int *>(gacc) - 20;
  if ((int)threadIdx.x <= c1 - c0 && threadIdx.x < 17)
    cstart[threadIdx.x] = ((int)threadIdx.x < c1 - c0 ? b.pair_start[c0 + threadIdx.x] : s0 + M) - s0;
  // One-element default grid: the polynomial coefficients of dE/dG (see `pc` below) once per CENTRE,
  // by the last lanes of the workgroup while the others stage, instead of four dependent global loads
  // and twenty operations in the prologue of every job
  double *pctab = gacc + 3 * kCap;  // [kMaxCentersPerBlock][5]
  if constexpr (DEFZ && NSPEC == 1) {
    const int k = (int)blockDim.x - 1 - (int)threadIdx.x;
    if (k < c1 - c0) {
      const double *wsrc = b.dEdG + (size_t)(c0 + k) * sf.ndim + sf.n_radial_dim;
      double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) {
        const double g1 = ch.gamma[ig], g2 = g1 * g1;
        const double w1 = wsrc[ch.chan[ig * NZ]] * ch.kz[0], w4 = wsrc[ch.chan[ig * NZ + 1]] * ch.kz[1];
        p0 += w1 + w4;
        p1 += g1 * (w1 + 4.0 * w4);
        p2 += 6.0 * g2 * w4;
        p3 += 4.0 * g2 * g1 * w4;
        p4 += g2 * g2 * w4;
      }
      // in powers of w = 2 cos(theta) (what the triple body has: (r_a^2 + r_b^2 - r_ab^2) / (r_a r_b))
      pctab[5 * k] = p0;
      pctab[5 * k + 1] = 0.5 * p1;
      pctab[5 * k + 2] = 0.25 * p2;
      pctab[5 * k + 3] = 0.125 * p3;
      pctab[5 * k + 4] = 0.0625 * p4;
    }
  }
  // dE/dG of the RADIAL channels per (centre, neighbour species) for the epilogue's G2 share, fetched
  // here by the lanes next to those (the workgroup's last ones, mostly without a pair) so that the
  // epilogue walks no pair_i -> species / dE/dG chain of dependent global loads at the kernel's tail
  double *rtab = pctab + 5 * kMaxCentersPerBlock;  // [kMaxCentersPerBlock][kRTab]
  const bool use_rtab = first && NSPEC * sf.n_rad <= kRTab;
  if (use_rtab) {
    const int per = NSPEC * sf.n_rad;
    const int t = (int)blockDim.x - 1 - (int)threadIdx.x - kMaxCentersPerBlock;
    if (t >= 0 && t < (c1 - c0) * per) {
      const int ci = t / per, rem = t - ci * per;
      const int sb = rem / sf.n_rad, c = rem - sb * sf.n_rad;
      const int64_t i = c0 + ci;
      rtab[ci * kRTab + rem] = b.dEdG[(size_t)i * sf.ndim + radial_term2(b.species[i], sb) * sf.n_rad + c];
    }
  }
  TA_STAMP(b, 1, 0);
  stage<HD>(sf, ch, beta, b, f, s0, M);
  TA_STAMP(b, 1, 1);
  const int nel = sf.n_elements;

  auto run_item = [&](int item, bool have_mask, unsigned long long mask0, auto is_job) {
    constexpr bool kJob = decltype(is_job)::value;  // see the forward kernel
    const int64_t p = (int64_t)s0 + item;
    int i, base, n;
    if constexpr (kJob) {
      const int ci = job_centre((uint32_t)mask0);
      i = c0 + ci;
      base = cstart[ci];
      n = cstart[ci + 1] - base;
    } else {
      i = b.pair_i[p];
      base = b.pair_start[i] - s0;
      n = pair_stop_of(b, i) - b.pair_start[i];
    }
    const int a = item - base;
    const double2 axy = f.xy[item], azr = f.zr[item], aih = f.ih[item], ags = f.gs[item];
    const double ax = axy.x, ay = axy.y, az = azr.x;
    const double ra2 = azr.y, inv_ra = aih.x, Ha = aih.y, La = ags.x;
    const double inv_ra2 = inv_ra * inv_ra;
    const int sa = (int)ags.y;

    // dE/dG of the channels of term (sa, sp) for every partner species sp
    // w = dE/dG 2^(1-zeta); wd = w zeta gamma (factor of the derivative of (1 + gamma c)^zeta)
    constexpr bool kTable = kJob && DEFZ && NSPEC == 1;  // coefficients from `pctab`
    double w_[NSPEC][NG][NZ], wd[NSPEC][NG][NZ];
#pragma unroll
    for (int sp = 0; sp < (kTable ? 0 : NSPEC); ++sp) {
      const double *wsrc = b.dEdG + (size_t)i * sf.ndim + sf.n_radial_dim +
                           angular_term2(sa, sp, nel) * sf.n_ang;
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) {
          w_[sp][ig][iz] = wsrc[ch.chan[ig * NZ + iz]] * ch.kz[iz];
          wd[sp][ig][iz] = w_[sp][ig][iz] * ch.zeta[iz] * ch.gamma[ig];
        }
    }

    // DEFZ: S0 = sum_c w_c kz (1 + g c)^zeta is a polynomial P(c) of degree 4 in cos(theta) and
    // S1 = sum_c w_c kz zeta g (1 + g c)^(zeta - 1) its derivative: coefficients once per job, then
    // 4 + 3 fused multiply-adds per triple instead of 16 operations
    double pc[NSPEC][5];
    if constexpr (kTable) {
      const double *src = pctab + 5 * (i - c0);
#pragma unroll
      for (int k = 0; k < 5; ++k) pc[0][k] = src[k];
    } else if constexpr (DEFZ) {
#pragma unroll
      for (int sp = 0; sp < NSPEC; ++sp) {
#pragma unroll
        for (int k = 0; k < 5; ++k) pc[sp][k] = 0.0;
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) {
          const double g1 = ch.gamma[ig], g2 = g1 * g1;
          const double w1 = w_[sp][ig][0], w4 = w_[sp][ig][1];
          pc[sp][0] += w1 + w4;
          pc[sp][1] += 0.5 * g1 * (w1 + 4.0 * w4);  // powers of w = 2 cos(theta), as in `pctab`
          pc[sp][2] += 0.25 * 6.0 * g2 * w4;
          pc[sp][3] += 0.125 * 4.0 * g2 * g1 * w4;
          pc[sp][4] += 0.0625 * g2 * g2 * w4;
        }
      }
    }
    // Own share of dE/dD_a = sum_b (ca D_a + Q D_b): the scalar sum of ca and the vector sum of Q D_b
    // (1 + 3 operations per triple instead of 6), combined once at the end
    double csum = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    auto triple = [&](int bl) {
        bl = (int)min((unsigned)bl, (unsigned)bl - (unsigned)n);  // bl < 2 n: bl mod n without a compare / select
        const int q = kProbe(flags, 30) ? (int)threadIdx.x : base + bl;
        const double2 bxy = f.xy[q], bzr = f.zr[q];
        const double bx = bxy.x, by = bxy.y, bz = bzr.x;
        const double ex = bx - ax, ey = by - ay, ez = bz - az;
        const double d2 = fma(ex, ex, fma(ey, ey, fma(ez, ez, sf.eps)));
        const double u = kProbe(flags, 30) ? 0.5 : d2 * sf.inv_ac2;
        if (!(u < 1.0)) return;  // exact test (the mask is a superset)
        const int q2 = kProbe(flags, 29) ? (int)threadIdx.x : q;
        const double2 bih = f.ih[q2], bgs = f.gs[q2];
        const double inv_rb = bih.x, Hb = bih.y, Lb = bgs.x;
        const int sb = NSPEC == 1 ? 0 : (int)bgs.y;
        // V = S0(c) T with T = Ha Hb Hd, c = cos(theta) = w / 2, w = (ra^2 + rb^2 - d^2) / (ra rb):
        //   dV/dD_a = ca D_a + Q D_b,   dV/dD_b = cb D_b + Q D_a,
        //   ca = S0 T La - (S1 T) c / ra^2 + X,  cb = S0 T Lb - (S1 T) c / rb^2 + X,
        //   Q = (S1 T) / (ra rb) - X,   X = S0 Ha Hb Hd2,   S1 = dS0/dc,
        // La, Lb = (dH/dr) / (r H) of the two pairs (staged), Hd2 = (dHd/dr) / r of the third side.
        // Product form throughout: nothing is divided by a cutoff factor or by (1 + gamma c).
        const double inv_ab = inv_ra * inv_rb;
        const double w = (ra2 + bzr.y - d2) * inv_ab;
        double Hd, dHd;
        hd_eval<HD>(sf, ch, beta, u, Hd, dHd);
        const double Hd2 = sf.two_inv_ac2 * dHd;
        double S0 = 0.0, Yc, Y;  // Y = S1 T, Yc = Y c
        const double HH = Ha * Hb, T = HH * Hd;
        if constexpr (DEFZ) {
          double p0 = pc[0][0], p1 = pc[0][1], p2 = pc[0][2], p3 = pc[0][3], p4 = pc[0][4];
#pragma unroll
          for (int sp = 1; sp < NSPEC; ++sp) {
            p0 = (sb == sp) ? pc[sp][0] : p0;
            p1 = (sb == sp) ? pc[sp][1] : p1;
            p2 = (sb == sp) ? pc[sp][2] : p2;
            p3 = (sb == sp) ? pc[sp][3] : p3;
            p4 = (sb == sp) ? pc[sp][4] : p4;
          }
          S0 = fma(fma(fma(fma(p4, w, p3), w, p2), w, p1), w, p0);
          const double S1w = fma(fma(fma(4.0 * p4, w, 3.0 * p3), w, 2.0 * p2), w, p1);  // dS0/dw = S1 / 2
          const double Z = S1w * T;
          Yc = Z * w;  // (2 Z) (w / 2)
          Y = Z + Z;
        } else {
          const double cth = 0.5 * w;
          double S1 = 0.0;
#pragma unroll
          for (int ig = 0; ig < NG; ++ig) {
            const double basev = fma(ch.gamma[ig], cth, 1.0);
#pragma unroll
            for (int iz = 0; iz < NZ; ++iz) {
              double ws = w_[0][ig][iz], wds = wd[0][ig][iz];
#pragma unroll
              for (int sp = 1; sp < NSPEC; ++sp) {
                ws = (sb == sp) ? w_[sp][ig][iz] : ws;
                wds = (sb == sp) ? wd[sp][ig][iz] : wds;
              }
              double pm1;
              if (ch.zeta_int[iz] > 0)
                pm1 = pow_int_m1(basev, ch.zeta_int[iz]);
              else
                pm1 = safe_pow_grad(ch.safe_pow, basev, ch.zeta[iz] - 1.0);
              S0 = fma(ws, pm1 * basev, S0);
              S1 = fma(wds, pm1, S1);
            }
          }
          Y = S1 * T;
          Yc = Y * cth;
        }
        const double S0T = S0 * T;
        const double X = (S0 * HH) * Hd2;
        const double ca = fma(S0T, La, fma(-Yc, inv_ra2, X));
        const double cb = fma(S0T, Lb, fma(-Yc, inv_rb * inv_rb, X));
        const double Q = fma(Y, inv_ab, -X);
        csum += ca;
        gx = fma(Q, bx, gx);
        gy = fma(Q, by, gy);
        gz = fma(Q, bz, gz);
        const int q3 = kProbe(flags, 28) ? (int)threadIdx.x : q;
        atomicAdd(&gacc[q3], fma(cb, bx, Q * ax));
        atomicAdd(&gacc[kCap + q3], fma(cb, by, Q * ay));
        atomicAdd(&gacc[2 * kCap + q3], fma(cb, bz, Q * az));
    };
    if constexpr (kJob) {
      const uint32_t jw = (uint32_t)mask0;
      unsigned m = (flags & (1 << 24)) ? 0u : jw >> 16;  // bit 24: measurement switch, triple bodies off
      const int k0 = a + 1 + 16 * (int)((jw >> 8) & 3u);
      while (m) {
        const int k = __ffs((int)m) - 1;
        m &= m - 1;
        triple(k0 + k);
      }
    } else {
    const int smax = (Ha != 0.0) ? n / 2 : 0;
    for (int sc = 1; sc <= smax; sc += 64) {
      unsigned long long mask = have_mask ? mask0 : b.masks[(size_t)(sc >> 6) * b.n_pairs + p];
      if (flags & (1 << 24)) mask = 0ull;
      while (mask) {
        const int k = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        triple(a + sc + k);
      }
    }
    }
    atomicAdd(&gacc[item], fma(csum, ax, gx));
    atomicAdd(&gacc[kCap + item], fma(csum, ay, gy));
    atomicAdd(&gacc[2 * kCap + item], fma(csum, az, gz));
  };

  {
    int item = threadIdx.x;
    const bool active = item < M;
    int n_own = 0;
    if (active) {
      const int i = b.pair_i[s0 + item];
      n_own = pair_stop_of(b, i) - b.pair_start[i];
    }
    const bool one_pass = M <= (int)blockDim.x && !__syncthreads_or(active && (n_own >> 1) > 64);
    const int n_jobs = (one_pass && b.job_count) ? b.job_count[blockIdx.x] : -1;
    if (n_jobs >= 0) {  // the forward kernel's job list (see make_jobs): no scan, no sort here
      const size_t jbase = (size_t)blockIdx.x * b.job_stride;
      // at most four rounds (four jobs per pair, one pair per lane): the words of all of them are
      // fetched before the first job runs, not one dependent global load per round
      uint32_t jws[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int slot = job_slot(r, threadIdx.x, blockDim.x);
        jws[r] = slot < n_jobs ? b.job_word[jbase + slot] : 0u;
      }
      TA_STAMP(b, 1, 2);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r * (int)blockDim.x >= n_jobs) break;
        const int slot = job_slot(r, threadIdx.x, blockDim.x);
        if (slot < n_jobs) run_item(job_item(jws[r]), true, jws[r], std::true_type{});
      }
      TA_STAMP(b, 1, 3);
    } else if (one_pass) {  // see deal_by_popcount
      unsigned long long mask = 0ull;
      if (active) {
        const int64_t p = (int64_t)s0 + item;
        const int smax = (f.ih[item].y != 0.0) ? n_own / 2 : 0;
        if (smax > 0) mask = b.masks[p];  // the forward kernel's candidate mask (always stored)
      }
      // (no re-dealing by popcount here: its scratch lived in the rings, which this kernel no longer has;
      // this path runs only with TA_NO_JOBS=1)
      if (active) run_item(item, true, mask, std::false_type{});
    } else {
      for (int it = threadIdx.x; it < M; it += blockDim.x) run_item(it, false, 0ull, std::false_type{});
    }
  }
  __syncthreads();
  TA_STAMP(b, 1, 4);
  for (int item = threadIdx.x; item < M; item += blockDim.x) {
    const int64_t p = (int64_t)s0 + item;
    double gx = gacc[item], gy = gacc[kCap + item], gz = gacc[2 * kCap + item];
    if (first) {
      // radial (G2) share: s_p D / r  (sf.py:101-108 differentiated)
      const double *wr;
      if (use_rtab) {
        int ci = 0;
        for (int k = 1; k < c1 - c0; ++k) ci += (item >= cstart[k]) ? 1 : 0;
        wr = rtab + ci * kRTab + (NSPEC == 1 ? 0 : (int)f.gs[item].y * sf.n_rad);
      } else {
        const int i = b.pair_i[p];
        wr = b.dEdG + (size_t)i * sf.ndim + radial_term2(b.species[i], (int)f.gs[item].y) * sf.n_rad;
      }
      const double r2 = f.zr[item].y, inv_r = f.ih[item].x;
      const double ur = r2 * sf.inv_rc2;
      double s = 0.0;
      if (ur < 1.0) {
        double fc, dfdu;
        cutoff_u(sf.cutoff, ur, fc, dfdu);
        const double r = r2 * inv_r;  // no square root: 1 / r is staged
        const double dfdr = dfdu * 2.0 * r * sf.inv_rc2;
        double e = 0.0;
        for (int c = 0; c < sf.n_rad; ++c) {
          const double dr = r - sf.omega[c];
          const int pw = sf.eta_pow[c];  // wave-uniform: exp(-eta[c] x) = exp(-eta[c - 1] x)^pw
          e = pw > 0 ? pow_int_m1(e, pw) * e : ta_exp(-sf.eta[c] * dr * dr * sf.inv_rc2);
          s = fma(wr[c], e * (dfdr - 2.0 * sf.eta[c] * dr * fc * sf.inv_rc2), s);
        }
      }
      s *= inv_r;
      gx = fma(s, f.xy[item].x, gx);
      gy = fma(s, f.xy[item].y, gy);
      gz = fma(s, f.zr[item].x, gz);
    } else {
      gx += b.g[4 * (size_t)p];
      gy += b.g[4 * (size_t)p + 1];
      gz += b.g[4 * (size_t)p + 2];
    }
    b.g[4 * (size_t)p] = gx;
    b.g[4 * (size_t)p + 1] = gy;
    b.g[4 * (size_t)p + 2] = gz;
  }
  TA_STAMP(b, 1, 5);
}

template <int NSPEC, int NG, int NZ>
void fwd_t(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b, int geom, hipStream_t s) {
  const dim3 grid((unsigned)b.n_blk), block((unsigned)(b.cap < kBlock ? b.cap : kBlock));
  const size_t lds = v2_lds_bytes(false, b.cap, b.job_count ? NSPEC * NG * NZ : 0, NSPEC);
  if constexpr (NZ == 2) {
    if (ch.n_hd == 12 && ch.zeta_int[0] == 1 && ch.zeta_int[1] == 4) {
      if (b.cap == kCapMin) {
        // One-element models take the 80-register build (6 wavefronts per SIMD, 19 spilled dwords): with the
        // compact partial sums (18 KB of LDS) 8 workgroups share a CU and the 2006 workgroups of the
        // 4000-atom frame are resident at once (they enter within 1.7 us instead of 22 us: before, 1536 ran
        // and the rest followed as a second lock-step round) -- forward 45.3 -> 41.6 us for one frame, and
        // 26.9 -> 24.8 us per frame in 64-frame batches, where the extra wavefront per SIMD hides more of
        // the LDS latency of the sweep. TA_FWD_WPE=5 selects the 96-register build (A/B).
        static const int force = getenv("TA_FWD_WPE") ? atoi(getenv("TA_FWD_WPE")) : 0;
        if constexpr (NSPEC <= 2) {
          if (force != 5) {  // (two elements: 55.6 -> 54.2 us for one Ni-Mo frame, 34.9 -> 34.1 us per frame at 16)
            hipLaunchKernelGGL((g4_forward_v2_kernel<NSPEC, NG, NZ, 12, true, kCapMin, 6>), grid, block, lds, s, sf, ch,
                               b, geom);
            return;
          }
        }
        hipLaunchKernelGGL((g4_forward_v2_kernel<NSPEC, NG, NZ, 12, true, kCapMin>), grid, block, lds, s, sf, ch, b, geom);
      } else
        hipLaunchKernelGGL((g4_forward_v2_kernel<NSPEC, NG, NZ, 12, true, 0>), grid, block, lds, s, sf, ch, b, geom);
      return;
    }
    if (ch.n_hd == 16 && ch.zeta_int[0] == 1 && ch.zeta_int[1] == 4) {
      hipLaunchKernelGGL((g4_forward_v2_kernel<NSPEC, NG, NZ, 16, true, 0>), grid, block, lds, s, sf, ch, b, geom);
      return;
    }
  }
  if (ch.n_hd > 0 && ch.n_hd <= 16)  // coefficients beyond n_hd are zero
    hipLaunchKernelGGL((g4_forward_v2_kernel<NSPEC, NG, NZ, 16, false, 0>), grid, block, lds, s, sf, ch, b, geom);
  else if (ch.n_hd == 24)
    hipLaunchKernelGGL((g4_forward_v2_kernel<NSPEC, NG, NZ, 24, false, 0>), grid, block, lds, s, sf, ch, b, geom);
  else
    hipLaunchKernelGGL((g4_forward_v2_kernel<NSPEC, NG, NZ, 0, false, 0>), grid, block, lds, s, sf, ch, b, geom);
}
template <int NSPEC, int NG, int NZ>
void bwd_t(const SFParams &sf, const AngChunk &ch, const DeviceBatch &b, int first, hipStream_t s) {
  const dim3 grid((unsigned)b.n_blk), block((unsigned)(b.cap < kBlock ? b.cap : kBlock));
  const size_t lds = v2_lds_bytes(true, b.cap);
  if constexpr (NZ == 2) {
    if (ch.n_hd == 12 && ch.zeta_int[0] == 1 && ch.zeta_int[1] == 4) {
      if (b.cap == kCapMin) {
        // (the 80-register build of this kernel, which also lets every workgroup of one frame be resident at
        // once, measured the same as the 96-register one: 48.9-49.3 against 49.5 us for one frame, 33.1
        // against 33.1 us per frame in batches; TA_BWD_WPE=6 selects it)
        static const int force = getenv("TA_BWD_WPE") ? atoi(getenv("TA_BWD_WPE")) : 0;
        if constexpr (NSPEC == 1) {
          if (force == 6) {
            hipLaunchKernelGGL((backward_v2_kernel<NSPEC, NG, NZ, 12, true, kCapMin, 6>), grid, block, lds, s, sf, ch,
                               b, first);
            return;
          }
        }
        hipLaunchKernelGGL((backward_v2_kernel<NSPEC, NG, NZ, 12, true, kCapMin>), grid, block, lds, s, sf, ch, b, first);
      } else
        hipLaunchKernelGGL((backward_v2_kernel<NSPEC, NG, NZ, 12, true, 0>), grid, block, lds, s, sf, ch, b, first);
      return;
    }
    if (ch.n_hd == 16 && ch.zeta_int[0] == 1 && ch.zeta_int[1] == 4) {
      hipLaunchKernelGGL((backward_v2_kernel<NSPEC, NG, NZ, 16, true, 0>), grid, block, lds, s, sf, ch, b, first);
      return;
    }
  }
  if (ch.n_hd > 0 && ch.n_hd <= 16)  // coefficients beyond n_hd are zero
    hipLaunchKernelGGL((backward_v2_kernel<NSPEC, NG, NZ, 16, false, 0>), grid, block, lds, s, sf, ch, b, first);
  else if (ch.n_hd == 24)
    hipLaunchKernelGGL((backward_v2_kernel<NSPEC, NG, NZ, 24, false, 0>), grid, block, lds, s, sf, ch, b, first);
  else
    hipLaunchKernelGGL((backward_v2_kernel<NSPEC, NG, NZ, 0, false, 0>), grid, block, lds, s, sf, ch, b, first);
}

}  // namespace

// n_local > 0: forward launch in job mode (counters + per-pair partial sums behind the fields)
size_t v2_lds_bytes(bool backward, int cap, int n_local, int nspec) {
  if (!backward && n_local > 0)
    return v2_counter_offset(cap) + kJobCtlBytes + (size_t)n_local * v2_pcols(cap, nspec) * sizeof(double);
  if (backward)  // fields, cstart gap, three accumulator planes, the per-centre tables
    return (size_t)cap * kNF * sizeof(double) + cap + 3 * (size_t)cap * sizeof(double) +
           kMaxCentersPerBlock * (5 + kRTab) * sizeof(double);
  return (size_t)cap * kNF * sizeof(double) + 3 * (size_t)(2 * cap + kRingPad) * sizeof(float) + cap;
}
int v2_job_stride(int cap) { return v2_max_jobs(cap); }

#ifdef TA_V2_FEW  // experiment builds (ISA inspection, quick A/B): the benchmark shape only
#define TA_DISPATCH_V2(FN, ...)                                   \
  do {                                                            \
    const int key = nspec * 100 + ng * 10 + nz;                   \
    if (key == 122) FN<1, 2, 2>(__VA_ARGS__);                     \
    else if (key == 222) FN<2, 2, 2>(__VA_ARGS__);                \
    else throw std::domain_error("TA_V2_FEW build");              \
  } while (0)
#else
#define TA_DISPATCH_V2(FN, ...)                                   \
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
      case 422: FN<4, 2, 2>(__VA_ARGS__); break;                  \
      case 522: FN<5, 2, 2>(__VA_ARGS__); break;                  \
      default:                                                    \
        throw std::domain_error("no second-generation angular kernel for this (species, gamma, zeta) shape"); \
    }                                                             \
  } while (0)
#endif

// stagger flags for a launch: "<count>[,<shift>]" from the environment (experiment switch)
static int stagger_bits(const DeviceBatch &b, const char *var) {
  const char *e = getenv(var);
  if (getenv("TA_DEBUG_NO_TRIPLES")) return 1 << 24;  // instruction accounting (wrong results)
  // more accounting switches (wrong results): bit 0 triple bodies, 1 candidate scan, 2 G2 sums, 3 job sweep
  if (const char *d = getenv("TA_DEBUG_SKIP")) return (atoi(d) & 127) << 24;
  if (!e) return 0;
  int n = 0, shift = 3;
  if (sscanf(e, "%d,%d", &n, &shift) < 1) return 0;
  (void)b;
  return ((n & 0xff) << 8) | ((shift & 31) << 16);
}

// `ch` must describe ONE beta (nb == 1).
void launch_g4_forward_v2(const SFParams &sf, const AngChunk &ch, int ng, int nz, bool geometry,
                          bool reduce, const DeviceBatch &b, hipStream_t s) {
  if (b.n_blk == 0) return;
  const int nspec = sf.n_elements;
  // bit 0: compute the pair geometry; bit 1: assemble the descriptors at the end; bit 2: this is
  // the only forward launch (all angular channels are here)
  const int geom = (geometry ? 1 : 0) | (reduce ? 2 : 0) | ((geometry && reduce) ? 4 : 0) |
                   stagger_bits(b, "TA_STAGGER_FWD");
  TA_DISPATCH_V2(fwd_t, sf, ch, b, geom, s);
}

void launch_backward_v2(const SFParams &sf, const AngChunk &ch, int ng, int nz, bool first,
                        const DeviceBatch &b, hipStream_t s) {
  if (b.n_blk == 0) return;
  const int nspec = sf.n_elements;
  const int f = (first ? 1 : 0) | stagger_bits(b, "TA_STAGGER_BWD");
  TA_DISPATCH_V2(bwd_t, sf, ch, b, f, s);
}

}  // namespace ta
