// Device-side neighbour list: positions -> packed pair buffers, entirely on the GPU.
//
// Replaces the host part of reference transformer/universal.py:46-112
// (`get_radial_metadata`: ASE `neighbor_list('ijS', atoms, rc)` at :58 plus the
// per-pair Python loops), which the reference's own profile shows to be 97 % of
// its wall time (doc/papers/nn/figures/cpc_speed.py:13-16). Same semantics as
// the host builder in ta_neighbor.cpp: full list, strict |Rj - Ri + S.h| < rc on
// the positions as given, per-axis periodicity; output sorted by centre and
// neighbour species with the reverse-pair index.
//
// Linked cells in fractional coordinates (bin width >= rc along every axis). The 27 neighbouring
// (bin, image shift) combinations of a centre are distinct even when an axis has only one or
// two bins (the same bin then appears with different shifts, which is how self-images and
// multiple images of one neighbour arise), so the only requirement is a cell at least rc
// thick along every periodic axis; thinner cells are built on the host.
//
// Kernels: bin_atoms (wrap, bin id, histogram) -> scan -> fill_bins -> gather_bins (records
// in bin order, atoms of a bin by index: deterministic) -> count_pairs (one wavefront per
// atom, ballot + popcount per neighbour species) -> scan_counts -> fill_pairs
// (same traversal, slots from the running popcounts) -> reverse_pairs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <stdexcept>
#include <vector>

#include "ta_device.h"
#include "ta_internal.h"

namespace ta {
namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void bin_atoms_kernel(int n_atoms, const double *pos,
                                                           const int32_t *frame_of_atom,
                                                           const NlGrid *grids, int32_t *wrap,
                                                           int32_t *binid, int32_t *bin_count,
                                                           int32_t *slot_in_bin, unsigned long long *clear,
                                                           int n_clear) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  // one-pass builder: its statistics and look-back words of the PREVIOUS call are cleared here, before the
  // kernel that uses them (no memset launch in front of every list)
  for (int k = i; k < n_clear; k += gridDim.x * kBlock) clear[k] = 0ull;
  if (i >= n_atoms) return;
  const NlGrid &g = grids[frame_of_atom[i]];
  const double *r = pos + 3 * (size_t)i;
  int b[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double f = r[0] * g.hinv[0 * 3 + a] + r[1] * g.hinv[1 * 3 + a] + r[2] * g.hinv[2 * 3 + a];
    int w = 0;
    if (g.pbc[a]) {
      w = (int)floor(f);
      f -= w;
    }
    wrap[3 * (size_t)i + a] = w;
    int k = (int)floor((f - g.lo[a]) * g.inv_w[a]);
    k = k < 0 ? 0 : (k >= g.nb[a] ? g.nb[a] - 1 : k);
    b[a] = k;
  }
  const int id = g.bin_offset + (b[0] * g.nb[1] + b[1]) * g.nb[2] + b[2];
  binid[i] = id;
  const int sl = atomicAdd(&bin_count[id], 1);
  if (slot_in_bin) slot_in_bin[i] = sl;  // one-pass builder: place in the bin as the atomics arrive
}

// inclusive scan of one int per lane over the 1024 lanes of a workgroup: wavefront scans by
// cross-lane shuffles, the 16 wavefront totals through LDS (two barriers instead of thirty)
__device__ __forceinline__ int block_scan_1024(int v, int *wtot /* [16], shared */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(v, off);
    if (lane >= off) v += u;
  }
  if (lane == 63) wtot[wave] = v;
  __syncthreads();
  int add = 0;
  for (int w = 0; w < wave; ++w) add += wtot[w];
  __syncthreads();
  return v + add;
}

// exclusive scan of `n` ints by one workgroup (n is small: bins, or atoms x species)
__global__ __launch_bounds__(1024) void scan_kernel(int n, const int32_t *in, int32_t *out,
                                                    int32_t *total) {
  __shared__ int wtot[16];
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
  int s = 0;
  for (int k = lo; k < hi; ++k) s += in[k];
  const int incl = block_scan_1024(s, wtot);
  int run = incl - s;
  for (int k = lo; k < hi; ++k) {
    const int v = in[k];
    out[k] = run;
    run += v;
  }
  if (t == 1023) {
    out[n] = incl;
    if (total) *total = incl;
  }
}

__global__ __launch_bounds__(kBlock) void fill_bins_kernel(int n_atoms, const int32_t *binid,
                                                           const int32_t *bin_start,
                                                           int32_t *bin_cursor, int32_t *bin_atoms) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_atoms) return;
  const int id = binid[i];
  bin_atoms[bin_start[id] + atomicAdd(&bin_cursor[id], 1)] = i;
}

// Atoms of every bin in index order (rank = members with a smaller index: removes the
// atomics' ordering noise), gathered into one record each so that the pair kernels read
// their candidates contiguously.
__global__ __launch_bounds__(kBlock) void gather_bins_kernel(int n_atoms, const double *pos,
                                                             const int32_t *species,
                                                             const int32_t *wrap, const int32_t *binid,
                                                             const int32_t *bin_start,
                                                             const int32_t *bin_atoms, NlRec *recs) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_atoms) return;
  const int bin = binid[i];
  const int lo = bin_start[bin], hi = bin_start[bin + 1];
  int rank = 0;
  for (int c = lo; c < hi; ++c) rank += bin_atoms[c] < i;
  NlRec r;
  r.x = pos[3 * (size_t)i];
  r.y = pos[3 * (size_t)i + 1];
  r.z = pos[3 * (size_t)i + 2];
  r.wx = wrap[3 * (size_t)i];
  r.wy = wrap[3 * (size_t)i + 1];
  r.wz = wrap[3 * (size_t)i + 2];
  r.j = i;
  r.sp = species[i];
  r.pad_ = 0;
  recs[lo + rank] = r;
}

// One wavefront per centre atom. Lanes 0..26 look up the 27 neighbouring bins (range of
// records + image shift), the ranges are concatenated by a prefix sum, and the wavefront then
// sweeps the concatenation 64 candidates at a time. MODE 0 counts the neighbours per
// species, MODE 1 writes them (same traversal, slots from the running popcounts).
template <int MODE>
__global__ __launch_bounds__(kBlock) void pairs_kernel(int n_atoms, int nel, double rmax,
                                                       const double *pos, const int32_t *frame_of_atom,
                                                       const NlGrid *grids, const int32_t *wrap,
                                                       const int32_t *binid, const int32_t *bin_start,
                                                       const NlRec *recs, int32_t *counts,
                                                       const int32_t *seg_start, int32_t *pair_i,
                                                       int32_t *pair_j, int32_t *pair_shift) {
  const int i = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n_atoms) return;
  const NlGrid &g = grids[frame_of_atom[i]];
  const int local = binid[i] - g.bin_offset;
  const int bz = local % g.nb[2], by = (local / g.nb[2]) % g.nb[1], bx = local / (g.nb[2] * g.nb[1]);
  const double rix = pos[3 * (size_t)i], riy = pos[3 * (size_t)i + 1], riz = pos[3 * (size_t)i + 2];
  const int wix = wrap[3 * (size_t)i], wiy = wrap[3 * (size_t)i + 1], wiz = wrap[3 * (size_t)i + 2];

  int running[kMaxElements];
#pragma unroll
  for (int s = 0; s < kMaxElements; ++s) running[s] = 0;
  const int32_t *seg = seg_start ? seg_start + (size_t)i * (nel + 1) : nullptr;

  // (bin, image shift) combinations around the centre: offsets -m .. m per axis, 27 of them unless the
  // cell is thinner than the cutoff along a periodic axis (NlGrid::m), handled 27 at a time by lanes 0..26
  const int e0 = 2 * g.m[0] + 1, e1 = 2 * g.m[1] + 1, e2 = 2 * g.m[2] + 1;
  const int n_combo = e0 * e1 * e2;
  constexpr int kSelf = 0x888;  // code of the zero shift
  for (int c0 = 0; c0 < n_combo; c0 += 27) {
    int lo = 0, len = 0, code = kSelf;
    if (lane < 27 && c0 + lane < n_combo) {
      const int id = c0 + lane;
      int c[3] = {bx + id / (e1 * e2) - g.m[0], by + (id / e2) % e1 - g.m[1], bz + id % e2 - g.m[2]};
      int sh[3] = {0, 0, 0};
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (c[a] < 0 || c[a] >= g.nb[a]) {
          if (!g.pbc[a]) ok = false;
          // floor division: how many cells the offset leaves the grid by
          sh[a] = c[a] >= 0 ? c[a] / g.nb[a] : -((-c[a] + g.nb[a] - 1) / g.nb[a]);
          c[a] -= sh[a] * g.nb[a];
        }
      }
      if (ok) {
        const int bin = g.bin_offset + (c[0] * g.nb[1] + c[1]) * g.nb[2] + c[2];
        lo = bin_start[bin];
        len = bin_start[bin + 1] - lo;
        code = (sh[0] + 8) | ((sh[1] + 8) << 4) | ((sh[2] + 8) << 8);
      }
    }
    // exclusive prefix of len over the lanes
    int incl = len;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    const int excl = incl - len;
    const int total = __builtin_amdgcn_readlane(incl, 26);
    const int delta = lo - excl;  // record index = delta[k] + flat index

    for (int base = 0; base < total; base += 64) {
      const int f = base + lane;
      int k = 0;
#pragma unroll
      for (int q = 1; q < 27; ++q) k += f >= __builtin_amdgcn_readlane(excl, q);
      const int dl = __shfl(delta, k);
      const int cd = __shfl(code, k);
      bool valid = false;
      int j = 0, Sx = 0, Sy = 0, Sz = 0, sj = 0;
      if (f < total) {
        const NlRec r = recs[dl + f];
        const int sx = (cd & 15) - 8, sy = ((cd >> 4) & 15) - 8, sz = ((cd >> 8) & 15) - 8;
        j = r.j;
        sj = r.sp;
        // shift relative to the positions as given: S = s - w_j + w_i
        Sx = sx - r.wx + wix;
        Sy = sy - r.wy + wiy;
        Sz = sz - r.wz + wiz;
        const double Dx = r.x - rix + (Sx * g.h[0] + Sy * g.h[3] + Sz * g.h[6]);
        const double Dy = r.y - riy + (Sx * g.h[1] + Sy * g.h[4] + Sz * g.h[7]);
        const double Dz = r.z - riz + (Sx * g.h[2] + Sy * g.h[5] + Sz * g.h[8]);
        const double r2 = Dx * Dx + Dy * Dy + Dz * Dz;
        valid = (sqrt(r2) < rmax) && !(j == i && cd == kSelf);
      }
      for (int s = 0; s < nel; ++s) {
        const unsigned long long m = __ballot(valid && sj == s);
        if (MODE == 1 && valid && sj == s) {
          const int slot = seg[s] + running[s] + __popcll(m & ((1ull << lane) - 1ull));
          pair_i[slot] = i;
          pair_j[slot] = j;
          pair_shift[3 * (size_t)slot] = Sx;
          pair_shift[3 * (size_t)slot + 1] = Sy;
          pair_shift[3 * (size_t)slot + 2] = Sz;
        }
        running[s] += __popcll(m);
      }
    }
  }
  if (MODE == 0 && lane == 0) {
    for (int s = 0; s < nel; ++s) counts[(size_t)i * (nel + 1) + s] = running[s];
    counts[(size_t)i * (nel + 1) + nel] = 0;  // slot so that seg_start has nel + 1 entries per atom
  }
}

// seg_start (from the scan) -> pair_start, and per-atom statistics
__global__ __launch_bounds__(kBlock) void finish_starts_kernel(int n_atoms, int nel,
                                                               const int32_t *seg_start,
                                                               const int32_t *counts,
                                                               int32_t *pair_start,
                                                               unsigned long long *n_triples,
                                                               int32_t *nnl_max,
                                                               unsigned long long *n_pairs64) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned long long tri = 0, cnt = 0;
  if (i == n_atoms) pair_start[i] = seg_start[(size_t)n_atoms * (nel + 1)];
  if (i < n_atoms) {
    const int a = seg_start[(size_t)i * (nel + 1)], b = seg_start[(size_t)(i + 1) * (nel + 1)];
    pair_start[i] = a;
    const int n = b - a;
    tri = (unsigned long long)n * (unsigned long long)(n - 1) / 2ull;
    // 64-bit total of the per-atom counts (from the int32 counts array, not from the scan): a batch
    // whose 32-bit running sum wraps is detected by the host even when the wrapped value is positive
    for (int sp = 0; sp < nel; ++sp) cnt += (unsigned long long)(unsigned)counts[(size_t)i * (nel + 1) + sp];
  }
  // one atomic per wavefront and statistic, not one per atom (they all hit the same three words)
  int mx = (int)cnt;
  for (int off = 32; off; off >>= 1) {
    tri += __shfl_xor(tri, off);
    cnt += __shfl_xor(cnt, off);
    mx = max(mx, __shfl_xor(mx, off));
  }
  if ((threadIdx.x & 63) == 0) {
    if (tri) atomicAdd(n_triples, tri);
    if (cnt) atomicAdd(n_pairs64, cnt);
    if (mx) atomicMax(nnl_max, mx);
  }
}

// (i -> j, S) <-> (j -> i, -S): eight lanes search j's segment of species(i)
__global__ __launch_bounds__(kBlock) void reverse_pairs_kernel(int64_t n_pairs, int nel,
                                                               const int32_t *species,
                                                               const int32_t *seg_start,
                                                               const int32_t *pair_i,
                                                               const int32_t *pair_j,
                                                               const int32_t *pair_shift,
                                                               int32_t *pair_rev, int32_t *n_missing) {
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t p = tid >> 3;
  const int sub = (int)(tid & 7);
  int found = -1;
  if (p < n_pairs) {
    const int i = pair_i[p], j = pair_j[p];
    const int sx = pair_shift[3 * p], sy = pair_shift[3 * p + 1], sz = pair_shift[3 * p + 2];
    const int32_t *seg = seg_start + (size_t)j * (nel + 1);
    const int si = species[i];
    const int hi = seg[si + 1];
    // four independent loads in flight per lane: the search is latency-bound
    for (int q0 = seg[si] + sub; q0 < hi; q0 += 32) {
      int qq[4], jj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        qq[u] = min(q0 + 8 * u, hi - 1);
        jj[u] = pair_j[qq[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = qq[u];
        if (jj[u] == i && pair_shift[3 * (size_t)q] == -sx && pair_shift[3 * (size_t)q + 1] == -sy &&
            pair_shift[3 * (size_t)q + 2] == -sz)
          found = q;
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) found = max(found, __shfl_xor(found, off));
  if (p < n_pairs && sub == 0) {
    pair_rev[p] = found;
    if (found < 0) atomicAdd(n_missing, 1);
  }
}

// ---- One-pass builder (round 3) ------------------------------------------------------------------
// The two-pass builder above traverses the bins twice (count, prefix sum over the batch, fill) and then
// finds every reverse pair by a linear search: 108 us of a 4000-atom frame's 155 us. Here ONE traversal
// per centre keeps the neighbours it finds in LDS as 64-bit keys {species, j, S}, a workgroup of 16
// centres learns its offset in the pair arrays from the groups before it (decoupled look-back over one
// 64-bit word per group, a wavefront reading 64 predecessors at a time), and the neighbours leave in
// KEY ORDER: sorted by species, then j, then S. The order inside a (centre, species) segment is thereby
// canonical (independent of the binning and of the order atomics arrived in, so the bins need no
// ranking pass), and the reverse pair (j -> i, -S) is a binary search in j's segment.
//
// Limits of this path (the two-pass builder takes over beyond them, see `bad` below): at most
// kBuildStash neighbours per centre, |S| < kShiftBias per axis.
constexpr int kBuildGroup = 16;    // centres per workgroup, one wavefront each
constexpr int kBuildStash = 384;   // neighbours per centre kept in LDS
constexpr int kShiftBias = 512;    // S + bias in 10 bits per axis
constexpr unsigned long long kStateMask = (1ull << 62) - 1ull;
constexpr unsigned long long kStateSum = 1ull << 62;     // value = pairs of this group
constexpr unsigned long long kStatePrefix = 2ull << 62;  // value = pairs of groups 0 .. this one
constexpr int kSpinLimit = 1 << 19;  // look-back polls before giving up (a second or so)

__device__ __forceinline__ unsigned long long nl_key(int sp, int j, int sx, int sy, int sz) {
  return ((unsigned long long)(unsigned)sp << 61) | ((unsigned long long)(unsigned)j << 30) |
         ((unsigned long long)(unsigned)(sx + kShiftBias) << 20) |
         ((unsigned long long)(unsigned)(sy + kShiftBias) << 10) | (unsigned long long)(unsigned)(sz + kShiftBias);
}

// SCAN: every workgroup forms the bins' offsets itself in LDS (a few hundred bins for one frame) instead of
// a one-workgroup scan launch in front; workgroup 0 also leaves them in `bin_start` for the pair kernel
constexpr int kPlaceScanBins = 2048;

template <bool SCAN>
__global__ __launch_bounds__(kBlock) void place_recs_kernel(int n_atoms, int n_bins, const double *pos,
                                                            const int32_t *species, const int32_t *wrap,
                                                            const int32_t *binid, const int32_t *bin_count,
                                                            int32_t *bin_start, const int32_t *slot_in_bin,
                                                            NlRec *recs) {
  __shared__ int start[SCAN ? kPlaceScanBins + 1 : 1];
  __shared__ int wtot[kBlock / 64];
  if constexpr (SCAN) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int chunk = (n_bins + kBlock - 1) / kBlock;
    const int lo = min(n_bins, t * chunk), hi = min(n_bins, lo + chunk);
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += bin_count[k];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off);
      if (lane >= off) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wtot[w];
    for (int k = lo; k < hi; ++k) {
      start[k] = run;
      run += bin_count[k];
    }
    if (t == kBlock - 1) start[n_bins] = run;
    __syncthreads();
    if (blockIdx.x == 0)
      for (int k = t; k <= n_bins; k += kBlock) bin_start[k] = start[k];
  }
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_atoms) return;
  NlRec r;
  r.x = pos[3 * (size_t)i];
  r.y = pos[3 * (size_t)i + 1];
  r.z = pos[3 * (size_t)i + 2];
  r.wx = wrap[3 * (size_t)i];
  r.wy = wrap[3 * (size_t)i + 1];
  r.wz = wrap[3 * (size_t)i + 2];
  r.j = i;
  r.sp = species[i];
  r.pad_ = 0;
  const int first = SCAN ? start[binid[i]] : bin_start[binid[i]];
  recs[first + slot_in_bin[i]] = r;
}

// stats (8 x u64, zero before the launch): [0] triples; as int32: [2] nnl_max, [3] `bad` (a centre beyond
// the limits, or the look-back gave up: use the two-pass builder), [6] reverse pairs missing; [4] pairs.
__global__ __launch_bounds__(64 * kBuildGroup) void build_pairs_kernel(
    int n_atoms, int nel, double rmax, const double *pos, const int32_t *frame_of_atom, const NlGrid *grids,
    const int32_t *wrap, const int32_t *binid, const int32_t *bin_start, const NlRec *recs, int32_t *bin_count,
    int n_bins, unsigned long long *gstate, long long capacity, int32_t *seg_start, int32_t *pair_start,
    int32_t *host_pair_start, int32_t *pair_i, int32_t *pair_j, int32_t *pair_shift,
    unsigned long long *stats) {
  __shared__ unsigned long long stash[kBuildGroup][kBuildStash];
  __shared__ unsigned short rank16[kBuildGroup][kBuildStash];
  __shared__ int cnt[kBuildGroup], cstart[kBuildGroup];
  __shared__ long long s_base;
  __shared__ int s_write;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * kBuildGroup + w);
  const bool act = i < n_atoms;
  // the bin histogram has been consumed (place_recs_kernel): cleared here for the next list
  for (int k = blockIdx.x * 64 * kBuildGroup + threadIdx.x; k <= n_bins; k += gridDim.x * 64 * kBuildGroup)
    bin_count[k] = 0;
  unsigned long long *mine = stash[w];
  int n = 0;
  bool bad = false;
  if (act) {
    const NlGrid &g = grids[__builtin_amdgcn_readfirstlane(frame_of_atom[i])];
    const int local = binid[i] - g.bin_offset;
    const int bz = local % g.nb[2], by = (local / g.nb[2]) % g.nb[1], bx = local / (g.nb[2] * g.nb[1]);
    const double rix = pos[3 * (size_t)i], riy = pos[3 * (size_t)i + 1], riz = pos[3 * (size_t)i + 2];
    const int wix = wrap[3 * (size_t)i], wiy = wrap[3 * (size_t)i + 1], wiz = wrap[3 * (size_t)i + 2];
    const double rmax2 = rmax * rmax;
    const int e0 = 2 * g.m[0] + 1, e1 = 2 * g.m[1] + 1, e2 = 2 * g.m[2] + 1;
    const int n_combo = e0 * e1 * e2;
    constexpr int kSelf = 0x888;
    // where the centre sits inside its own bin (0 .. 1 per axis): a neighbouring bin whose nearest face is
    // farther than rmax holds no neighbour and is skipped (27 bins of width >= rmax hold 6.4 times the
    // sphere's volume; for an orthogonal cell about a third of them drop out)
    const int own[3] = {bx, by, bz};
    double inbin[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double fa = rix * g.hinv[0 * 3 + a] + riy * g.hinv[1 * 3 + a] + riz * g.hinv[2 * 3 + a];
      if (g.pbc[a]) fa -= floor(fa);
      const double t = (fa - g.lo[a]) * g.inv_w[a] - own[a];
      inbin[a] = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    }
    for (int c0 = 0; c0 < n_combo; c0 += 27) {
      int lo = 0, len = 0, code = kSelf;
      if (lane < 27 && c0 + lane < n_combo) {
        const int id = c0 + lane;
        const int d[3] = {id / (e1 * e2) - g.m[0], (id / e2) % e1 - g.m[1], id % e2 - g.m[2]};
        int c[3] = {bx + d[0], by + d[1], bz + d[2]};
        int sh[3] = {0, 0, 0};
        bool ok = true;
        {
          double far2 = 0.0, far1 = 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double gap = d[a] == 0 ? 0.0 : ((d[a] < 0 ? -d[a] - 1 + inbin[a] : d[a] - inbin[a]) * g.bw[a]);
            far2 = fma(gap, gap, far2);
            far1 = gap > far1 ? gap : far1;
          }
          const double lim = rmax * (1.0 + 1e-9);
          if (g.ortho ? far2 > lim * lim : far1 > lim) ok = false;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (c[a] < 0 || c[a] >= g.nb[a]) {
            if (!g.pbc[a]) ok = false;
            sh[a] = c[a] >= 0 ? c[a] / g.nb[a] : -((-c[a] + g.nb[a] - 1) / g.nb[a]);
            c[a] -= sh[a] * g.nb[a];
          }
        }
        if (ok) {
          const int bin = g.bin_offset + (c[0] * g.nb[1] + c[1]) * g.nb[2] + c[2];
          lo = bin_start[bin];
          len = bin_start[bin + 1] - lo;
          code = (sh[0] + 8) | ((sh[1] + 8) << 4) | ((sh[2] + 8) << 8);
        }
      }
      int incl = len;
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
      }
      const int excl = incl - len;
      const int total = __builtin_amdgcn_readlane(incl, 26);
      const int delta = lo - excl;
      for (int base = 0; base < total; base += 64) {
        const int f = base + lane;
        int k = 0;
#pragma unroll
        for (int q = 1; q < 27; ++q) k += f >= __builtin_amdgcn_readlane(excl, q);
        const int dl = __shfl(delta, k);
        const int cd = __shfl(code, k);
        bool valid = false;
        unsigned long long key = 0;
        if (f < total) {
          const NlRec r = recs[dl + f];
          const int Sx = (cd & 15) - 8 - r.wx + wix, Sy = ((cd >> 4) & 15) - 8 - r.wy + wiy,
                    Sz = ((cd >> 8) & 15) - 8 - r.wz + wiz;
          const double Dx = r.x - rix + (Sx * g.h[0] + Sy * g.h[3] + Sz * g.h[6]);
          const double Dy = r.y - riy + (Sx * g.h[1] + Sy * g.h[4] + Sz * g.h[7]);
          const double Dz = r.z - riz + (Sx * g.h[2] + Sy * g.h[5] + Sz * g.h[8]);
          const double r2 = Dx * Dx + Dy * Dy + Dz * Dz;
          // the two-pass builder's test is sqrt(r2) < rmax; away from the boundary r2 decides, next to it
          // (relative 1e-14) the square root does, so both builders keep exactly the same pairs
          valid = r2 < rmax2 * (1.0 - 1e-14) || (r2 <= rmax2 * (1.0 + 1e-14) && sqrt(r2) < rmax);
          valid = valid && !(r.j == i && cd == kSelf);
          if (valid) {
            if (abs(Sx) >= kShiftBias || abs(Sy) >= kShiftBias || abs(Sz) >= kShiftBias) bad = true;
            key = nl_key(r.sp, r.j, Sx, Sy, Sz);
          }
        }
        const unsigned long long m = __ballot(valid);
        const int at = n + __popcll(m & ((1ull << lane) - 1ull));
        if (valid && at < kBuildStash) mine[at] = key;
        n += __popcll(m);
      }
    }
    if (n > kBuildStash) bad = true;
    bad = __any(bad);
  }
  const int nn = bad ? 0 : n;  // a centre beyond the limits writes nothing; the host falls back
  if (lane == 0) cnt[w] = nn;
  __syncthreads();

  // offsets: the centres of the group by a scan over 16 lanes, the group by looking back
  if (w == 0) {
    const int c = lane < kBuildGroup ? cnt[lane] : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < kBuildGroup; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    if (lane < kBuildGroup) cstart[lane] = incl - c;
    {  // statistics: ONE set of atomics per group (they all hit the same two words)
      unsigned long long tri = (unsigned long long)c * (unsigned long long)(c > 0 ? c - 1 : 0) / 2ull;
      int mx = c;
#pragma unroll
      for (int off = 1; off < kBuildGroup; off <<= 1) {
        tri += __shfl_xor(tri, off);
        mx = max(mx, __shfl_xor(mx, off));
      }
      if (lane == 0) {
        if (tri) atomicAdd(stats, tri);
        if (mx) atomicMax(reinterpret_cast<int32_t *>(stats) + 2, mx);
      }
    }
    const unsigned long long total = (unsigned long long)__builtin_amdgcn_readlane(incl, kBuildGroup - 1);
    const int gi = blockIdx.x;
    if (lane == 0)
      __hip_atomic_store(&gstate[gi], (gi == 0 ? kStatePrefix : kStateSum) | total, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long before = 0;
    bool gave_up = false;
    int spins = 0;
    for (int hi = gi - 1; hi >= 0;) {
      const int k = hi - lane;
      // below group 0: a prefix of zero, so the walk always ends
      const unsigned long long v =
          k >= 0 ? __hip_atomic_load(&gstate[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatePrefix;
      const unsigned fl = (unsigned)(v >> 62);
      const unsigned long long m_prefix = __ballot(fl == 2), m_empty = __ballot(fl == 0);
      const int first = m_prefix ? __builtin_ctzll(m_prefix) : 63;  // lanes 0 .. first are what is needed
      const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
      if (m_empty & need) {
        if (++spins > kSpinLimit) {
          gave_up = true;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      unsigned long long part = lane <= first ? (v & kStateMask) : 0ull;
#pragma unroll
      for (int off = 32; off; off >>= 1) part += __shfl_xor(part, off);
      before += part;
      if (m_prefix) break;
      hi -= 64;
    }
    if (lane == 0) {
      // after giving up: still publish a prefix, so that the groups behind do not wait as well
      if (gi > 0)
        __hip_atomic_store(&gstate[gi], kStatePrefix | ((before + total) & kStateMask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_base = (long long)before;
      s_write = !gave_up && (long long)(before + total) <= capacity;
      if (gave_up) atomicOr(reinterpret_cast<int32_t *>(stats) + 3, 1);
      const int last = min(gi * kBuildGroup + kBuildGroup, n_atoms);
      if (last == n_atoms) {  // the group of the last centre closes the arrays
        const unsigned long long all = before + total;
        pair_start[n_atoms] = (int32_t)all;
        if (host_pair_start) host_pair_start[n_atoms] = (int32_t)all;
        seg_start[(size_t)n_atoms * (nel + 1)] = (int32_t)all;
        stats[4] = all;
      }
    }
  }
  // order of the neighbours inside the centre: rank of each key among the centre's keys (they are
  // distinct); n is about a hundred, so n^2 / 64 compares per lane beat a sorting network's bookkeeping
  for (int e = lane; e < nn; e += 64) {
    const unsigned long long key = mine[e];
    int r = 0;
#pragma unroll 4
    for (int k = 0; k < nn; ++k) r += mine[k] < key;
    rank16[w][e] = (unsigned short)r;
  }
  if (act && __any(bad) && lane == 0) atomicOr(reinterpret_cast<int32_t *>(stats) + 3, 1);
  __syncthreads();
  if (!act) return;
  const long long at = s_base + cstart[w];
  // segment offsets: lane s counts the keys of species s
  int segc = 0;
  for (int e0 = 0; e0 < nn; e0 += 64) {
    const int e = e0 + lane;
    const int sp = e < nn ? (int)(mine[e] >> 61) : -1;
    for (int s = 0; s < nel; ++s) {
      const int c = __popcll(__ballot(sp == s));
      if (lane == s) segc += c;
    }
  }
  int incl = segc;
#pragma unroll
  for (int off = 1; off <= kMaxElements; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane <= nel) seg_start[(size_t)i * (nel + 1) + lane] = (int32_t)(at + incl - segc);
  if (lane == 0) {
    pair_start[i] = (int32_t)at;
    if (host_pair_start) host_pair_start[i] = (int32_t)at;
  }
  if (!s_write) return;
  for (int e = lane; e < nn; e += 64) {
    const unsigned long long key = mine[e];
    const long long slot = at + rank16[w][e];
    pair_i[slot] = i;
    pair_j[slot] = (int32_t)((key >> 30) & 0x7fffffffull);
    pair_shift[3 * slot] = (int)((key >> 20) & 1023) - kShiftBias;
    pair_shift[3 * slot + 1] = (int)((key >> 10) & 1023) - kShiftBias;
    pair_shift[3 * slot + 2] = (int)(key & 1023) - kShiftBias;
  }
}

// reverse pair in a list in key order: lower bound of i among the js of j's segment of species(i), then
// the (few) images of i
// `n_dev` (one-pass builder, launched right behind build_pairs_kernel, before the host knows the count):
// the number of pairs is read from the device, the grid covers an estimate; a list that did not fit
// `capacity` was not written and is left alone.
__global__ __launch_bounds__(kBlock) void reverse_sorted_kernel(int64_t n_pairs, int nel,
                                                                const unsigned long long *n_dev, int64_t capacity,
                                                                const int32_t *species,
                                                                const int32_t *seg_start,
                                                                const int32_t *pair_i, const int32_t *pair_j,
                                                                const int32_t *pair_shift, int32_t *pair_rev,
                                                                int32_t *n_missing) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (n_dev) {
    const unsigned long long n = *n_dev;
    if (n > (unsigned long long)capacity) return;
    n_pairs = (int64_t)n < n_pairs ? (int64_t)n : n_pairs;
  }
  if (p >= n_pairs) return;
  const int i = pair_i[p], j = pair_j[p];
  const int sx = pair_shift[3 * p], sy = pair_shift[3 * p + 1], sz = pair_shift[3 * p + 2];
  const int32_t *seg = seg_start + (size_t)j * (nel + 1);
  const int si = species[i];
  int lo = seg[si];
  const int end = seg[si + 1];
  int hi = end;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (pair_j[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  int found = -1;
  for (int q = lo; q < end && pair_j[q] == i; ++q)
    if (pair_shift[3 * (size_t)q] == -sx && pair_shift[3 * (size_t)q + 1] == -sy &&
        pair_shift[3 * (size_t)q + 2] == -sz) {
      found = q;
      break;
    }
  pair_rev[p] = found;
  if (found < 0) atomicAdd(n_missing, 1);
}

// ---- MD loop: the exact list of a step from the resident skin list ------------------------------
// The resident list covers rmax + skin (ta_set_skin); while it is valid, the pairs inside rmax at the
// CURRENT positions are a subset of it. These kernels extract that subset with the layout of the
// builder's output (sorted by centre and neighbour species, reverse index, workgroup packing), so that
// the evaluation kernels run on an exact list and never see the skin.
//
// Round 3: TWO launches instead of four (count, scan, fill + pack, reverse: 37 us of a 200 us MD step).
// A global prefix sum over the centres is what forced count and fill apart, so there is none: every
// GROUP of 16 consecutive centres (one 1024-lane workgroup, a wavefront per centre) compacts its pairs
// in place, starting at the group's own offset in the skin list; inside a group the centres are
// contiguous, between groups a few slots stay unused, and the kernels take the end of a centre from
// `pair_stop` (DeviceBatch) instead of the next centre's start. (Leaving the 32-byte pair records
// {D, r^2} of the kept pairs as well, for the forward kernel's staging to read instead of gathering
// positions, shifts and cells again, was measured and bought nothing: 125.9 against 120.7 us per
// evaluation with / without, MD step 0.186 against 0.184 ms; removed.)
constexpr int kFilterGroup = 16;
constexpr int kFilterMaxEl = 8;
static_assert(kFilterGroup == kMaxCentersPerBlock, "a run of the angular kernels never crosses a group");

__global__ __launch_bounds__(64 * kFilterGroup) void filter_group_kernel(
    int n_atoms, int nel, double rmax, const double *pos, const double *cells,
    const int32_t *frame_of_atom, const int32_t *start_super, const int32_t *seg_super, const int32_t *pj_super,
    const int32_t *ps_super, int32_t *seg_exact, int32_t *pair_start, int32_t *pair_stop, int32_t *pi_out,
    int32_t *pj_out, int32_t *ps_out, int32_t *map, int32_t *slot_q, int cap, int32_t *blk_center) {
  __shared__ int cnt[kFilterGroup][kFilterMaxEl + 1];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = blockIdx.x * kFilterGroup;
  const int i = i0 + w;
  const bool act = i < n_atoms;
  double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ri[3] = {0, 0, 0};
  const int32_t *seg = seg_super + (size_t)(act ? i : 0) * (nel + 1);
  if (act) {
    const double *hc = cells + 9 * (size_t)frame_of_atom[i];
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = hc[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ri[k] = pos[3 * (size_t)i + k];
  }
  // one pair of the skin list: D, and whether it is inside rmax (the builder's test, pairs_kernel)
  auto probe = [&](int q, int &j, int (&S)[3], double (&D)[3]) {
    j = pj_super[q];
#pragma unroll
    for (int c = 0; c < 3; ++c) S[c] = ps_super[3 * (size_t)q + c];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      D[c] = pos[3 * (size_t)j + c] - ri[c] + (S[0] * h[c] + S[1] * h[3 + c] + S[2] * h[6 + c]);
    return sqrt(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]) < rmax;
  };
  for (int s = 0; s < nel && act; ++s) {
    int running = 0;
    for (int q0 = seg[s]; q0 < seg[s + 1]; q0 += 64) {
      const int q = q0 + lane;
      bool valid = false;
      if (q < seg[s + 1]) {
        int j, S[3];
        double D[3];
        valid = probe(q, j, S, D);
      }
      running += __popcll(__ballot(valid));
    }
    if (lane == 0) cnt[w][s] = running;
  }
  if (!act && lane < nel) cnt[w][lane] = 0;
  __syncthreads();
  // Workgroup packing of the angular kernels (runs of whole centres with <= cap pairs), per group: a
  // run never crosses a group (kFilterGroup = kMaxCentersPerBlock), so every group owns the 16 run slots
  // [16 g, 16 g + 16); slots it does not need are empty runs (first centre = last), whose workgroups
  // leave at once. No prefix sum over the batch, no separate packing launch.
  if (blk_center && threadIdx.x == 0) {
    int slot = i0, load = 0, nc = 0;
    const int iend = min(i0 + kFilterGroup, n_atoms);
    for (int k = 0; k < kFilterGroup && i0 + k < n_atoms; ++k) {
      int n = 0;
      for (int s = 0; s < nel; ++s) n += cnt[k][s];
      if (k == 0 || load + n > cap || nc >= kMaxCentersPerBlock) {
        blk_center[slot++] = i0 + k;
        load = 0;
        nc = 0;
      }
      load += n;
      ++nc;
    }
    for (; slot < i0 + kFilterGroup; ++slot) blk_center[slot] = iend;
    if (iend == n_atoms) blk_center[i0 + kFilterGroup] = n_atoms;  // closes the last group's last run
  }
  if (!act) return;
  int off = start_super[i0];  // the group keeps its place in the skin list
  for (int k = 0; k < w; ++k)
    for (int s = 0; s < nel; ++s) off += cnt[k][s];
  int32_t *sx = seg_exact + (size_t)i * (nel + 1);
  {
    int o = off;
    for (int s = 0; s < nel; ++s) {
      if (lane == 0) sx[s] = o;
      o += cnt[w][s];
    }
    if (lane == 0) {
      sx[nel] = o;
      pair_start[i] = off;
      pair_stop[i] = o;
    }
  }
  for (int s = 0; s < nel; ++s) {
    int running = off;
    off += cnt[w][s];
    for (int q0 = seg[s]; q0 < seg[s + 1]; q0 += 64) {
      const int q = q0 + lane;
      bool valid = false;
      int j = 0, S[3] = {0, 0, 0};
      double D[3] = {0, 0, 0};
      if (q < seg[s + 1]) valid = probe(q, j, S, D);
      const unsigned long long m = __ballot(valid);
      if (q < seg[s + 1]) {
        int slot = -1;
        if (valid) {
          slot = running + __popcll(m & ((1ull << lane) - 1ull));
          pi_out[slot] = i;
          pj_out[slot] = j;
#pragma unroll
          for (int c = 0; c < 3; ++c) ps_out[3 * (size_t)slot + c] = S[c];
          if (slot_q) slot_q[slot] = q;  // (no reverse-index launch: DeviceBatch::slot_q)
        }
        map[q] = slot;
      }
      running += __popcll(m);
    }
  }
}

// second launch: the reverse index of the exact list through the map (a pair and its reverse have the
// same length, so both are inside rmax or neither is)
__global__ __launch_bounds__(kBlock) void filter_rev_kernel(int64_t n_super, const int32_t *rev_super,
                                                            const int32_t *map, int32_t *rev_out) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= n_super) return;
  const int slot = map[q];
  if (slot >= 0) rev_out[slot] = map[rev_super[q]];
}

inline unsigned nblk(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

// Host: grid parameters of one frame. Returns false when the frame cannot use the device
// builder (a periodic axis thinner than rmax) or the cell is singular.
bool nl_make_grid(const ta_frame &fr, double rmax, int bin_offset, NlGrid &g) {
  for (int k = 0; k < 9; ++k) g.h[k] = fr.cell[k];
  auto norm = [](const double *a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
  auto cross = [](const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
  };
  for (int a = 0; a < 3; ++a)  // incomplete cells (zero rows): leave to the host builder
    if (g.h[3 * a] == 0.0 && g.h[3 * a + 1] == 0.0 && g.h[3 * a + 2] == 0.0) return false;
  double c0[3], c1[3], c2[3];
  cross(&g.h[3], &g.h[6], c0);
  cross(&g.h[6], &g.h[0], c1);
  cross(&g.h[0], &g.h[3], c2);
  const double det = g.h[0] * c0[0] + g.h[1] * c0[1] + g.h[2] * c0[2];
  if (det == 0.0 || !std::isfinite(det)) return false;
  for (int a = 0; a < 3; ++a) {
    g.hinv[3 * a + 0] = c0[a] / det;
    g.hinv[3 * a + 1] = c1[a] / det;
    g.hinv[3 * a + 2] = c2[a] / det;
  }
  const double vol = std::fabs(det);
  const double *cr[3] = {c0, c1, c2};
  for (int a = 0; a < 3; ++a) {
    g.pbc[a] = fr.pbc[a] != 0;
    const double height = vol / norm(cr[a]);
    const double wfrac = rmax / height;  // bin width in fractional units (perpendicular width = rmax)
    g.m[a] = 1;
    if (g.pbc[a]) {
      const int nb = (int)std::floor(1.0 / wfrac);
      if (nb < 1) {
        // cell thinner than the cutoff along this axis: ONE bin, whose images -m .. m are all looked at
        // (round 3; such cells took the host builder before). Two wrapped atoms are less than one cell
        // apart along the axis, so images beyond floor(rmax / height) + 1 cannot be inside rmax.
        const int m = (int)std::floor(rmax / height) + 1;
        if (m > 7) return false;  // 4-bit shift codes in the kernel; below ~1 A of cell height: host builder
        g.m[a] = m;
      }
      g.nb[a] = std::max(1, std::min(nb, 64));
      g.lo[a] = 0.0;
      g.inv_w[a] = (double)g.nb[a];
    } else {
      double mn = 1e300, mx = -1e300;
      for (int i = 0; i < fr.n_atoms; ++i) {
        const double *r = &fr.positions[3 * (size_t)i];
        const double f = r[0] * g.hinv[0 * 3 + a] + r[1] * g.hinv[1 * 3 + a] + r[2] * g.hinv[2 * 3 + a];
        if (!std::isfinite(f)) return false;
        mn = std::min(mn, f);
        mx = std::max(mx, f);
      }
      if (fr.n_atoms == 0) mn = mx = 0.0;
      const double ext = mx - mn;
      int nb = (ext > 0 && wfrac > 0) ? (int)std::floor(ext / wfrac) : 1;
      nb = std::max(1, std::min(nb, 64));
      g.nb[a] = nb;
      g.lo[a] = mn;
      g.inv_w[a] = ext > 0 ? nb / ext : 0.0;
    }
  }
  for (int a = 0; a < 3; ++a) {
    const double height = vol / norm(cr[a]);
    g.bw[a] = g.inv_w[a] > 0.0 ? height / g.inv_w[a] : 0.0;
  }
  auto dot = [&](int a, int b) { return g.h[3 * a] * g.h[3 * b] + g.h[3 * a + 1] * g.h[3 * b + 1] + g.h[3 * a + 2] * g.h[3 * b + 2]; };
  g.ortho = 1;
  for (int a = 0; a < 3; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (std::fabs(dot(a, b)) > 1e-12 * std::sqrt(dot(a, a) * dot(b, b))) g.ortho = 0;
  g.bin_offset = bin_offset;
  return true;
}

int nl_bins(const NlGrid &g) { return g.nb[0] * g.nb[1] * g.nb[2]; }

// Device pipeline. All pointers are device buffers sized by the caller:
// (see NlWork in ta_device.h); nl_count ends with the pair count in stats, the caller reads it,
// sizes the pair buffers and calls nl_fill.
void nl_count(int n_atoms, int n_bins, int nel, double rmax, const double *pos, const int32_t *species,
              const int32_t *frame_of_atom, const NlGrid *grids, NlWork &w, int32_t *pair_start,
              hipStream_t s) {
  (void)hipMemsetAsync(w.bin_count, 0, (size_t)(n_bins + 1) * sizeof(int32_t), s);
  (void)hipMemsetAsync(w.bin_cursor, 0, (size_t)(n_bins + 1) * sizeof(int32_t), s);
  (void)hipMemsetAsync(w.stats, 0, 8 * sizeof(unsigned long long), s);
  if (n_atoms == 0) return;
  hipLaunchKernelGGL(bin_atoms_kernel, dim3(nblk(n_atoms, kBlock)), dim3(kBlock), 0, s, n_atoms, pos,
                     frame_of_atom, grids, w.wrap, w.binid, w.bin_count, (int32_t *)nullptr, (unsigned long long *)nullptr, 0);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, n_bins, w.bin_count, w.bin_start,
                     (int32_t *)nullptr);
  hipLaunchKernelGGL(fill_bins_kernel, dim3(nblk(n_atoms, kBlock)), dim3(kBlock), 0, s, n_atoms,
                     w.binid, w.bin_start, w.bin_cursor, w.bin_atoms);
  hipLaunchKernelGGL(gather_bins_kernel, dim3(nblk(n_atoms, kBlock)), dim3(kBlock), 0, s, n_atoms,
                     pos, species, w.wrap, w.binid, w.bin_start, w.bin_atoms, w.recs);
  hipLaunchKernelGGL(pairs_kernel<0>, dim3(nblk((int64_t)n_atoms * 64, kBlock)), dim3(kBlock), 0, s,
                     n_atoms, nel, rmax, pos, frame_of_atom, grids, w.wrap, w.binid, w.bin_start,
                     w.recs, w.counts, (const int32_t *)nullptr, (int32_t *)nullptr,
                     (int32_t *)nullptr, (int32_t *)nullptr);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, n_atoms * (nel + 1), w.counts,
                     w.seg_start, reinterpret_cast<int32_t *>(w.stats) + 4);
  hipLaunchKernelGGL(finish_starts_kernel, dim3(nblk(n_atoms + 1, kBlock)), dim3(kBlock), 0, s,
                     n_atoms, nel, w.seg_start, w.counts, pair_start, w.stats,
                     reinterpret_cast<int32_t *>(w.stats) + 2, w.stats + 4);
}

void nl_fill(int n_atoms, int64_t n_pairs, int nel, double rmax, const double *pos,
             const int32_t *species, const int32_t *frame_of_atom, const NlGrid *grids, NlWork &w,
             int32_t *pair_i, int32_t *pair_j, int32_t *pair_shift, int32_t *pair_rev,
             hipStream_t s) {
  if (n_atoms == 0) return;
  hipLaunchKernelGGL(pairs_kernel<1>, dim3(nblk((int64_t)n_atoms * 64, kBlock)), dim3(kBlock), 0, s,
                     n_atoms, nel, rmax, pos, frame_of_atom, grids, w.wrap, w.binid, w.bin_start,
                     w.recs, (int32_t *)nullptr, w.seg_start, pair_i, pair_j, pair_shift);
  if (n_pairs > 0)
    hipLaunchKernelGGL(reverse_pairs_kernel, dim3(nblk(n_pairs * 8, kBlock)), dim3(kBlock), 0, s, n_pairs,
                       nel, species, w.seg_start, pair_i, pair_j, pair_shift, pair_rev,
                       reinterpret_cast<int32_t *>(w.stats) + 6);
}

// One-pass builder. `zero` is one block the caller sized with nl_build_zero_words(): statistics, the
// look-back words and the bin histogram, cleared by ONE memset. Pairs are written only while they fit
// `capacity` (entries of pair_i / pair_j / pair_rev; pair_shift holds three times as many): the caller
// reads stats[4] (pairs) and the `bad` flag, grows the arrays and calls again when they did not fit.
size_t nl_build_zero_words(int n_atoms, int n_bins) {
  return 8 + (size_t)nblk(n_atoms, kBuildGroup) + 1 + ((size_t)n_bins + 2 + 1) / 2;
}

void nl_build(int n_atoms, int n_bins, int nel, double rmax, const double *pos, const int32_t *species,
              const int32_t *frame_of_atom, const NlGrid *grids, NlWork &w, unsigned long long *zero,
              bool zero_is_clean, long long capacity, int32_t *pair_start, int32_t *host_pair_start,
              int32_t *pair_i, int32_t *pair_j, int32_t *pair_shift, int32_t *pair_rev, long long rev_cover,
              hipStream_t s) {
  const size_t n_groups = nblk(n_atoms, kBuildGroup);
  // `zero_is_clean`: the previous list had the same layout and left the histogram zero (build_pairs_kernel);
  // the statistics and look-back words are cleared by bin_atoms_kernel
  if (!zero_is_clean)
    (void)hipMemsetAsync(zero, 0, nl_build_zero_words(n_atoms, n_bins) * sizeof(unsigned long long), s);
  w.stats = zero;
  unsigned long long *gstate = zero + 8;
  w.bin_count = reinterpret_cast<int32_t *>(zero + 8 + n_groups + 1);
  if (n_atoms == 0) return;
  hipLaunchKernelGGL(bin_atoms_kernel, dim3(nblk(n_atoms, kBlock)), dim3(kBlock), 0, s, n_atoms, pos,
                     frame_of_atom, grids, w.wrap, w.binid, w.bin_count, w.bin_atoms, zero,
                     (int)(8 + n_groups + 1));
  if (n_bins <= kPlaceScanBins) {
    hipLaunchKernelGGL(place_recs_kernel<true>, dim3(nblk(n_atoms, kBlock)), dim3(kBlock), 0, s, n_atoms, n_bins,
                       pos, species, w.wrap, w.binid, w.bin_count, w.bin_start, w.bin_atoms, w.recs);
  } else {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, n_bins, w.bin_count, w.bin_start,
                       (int32_t *)nullptr);
    hipLaunchKernelGGL(place_recs_kernel<false>, dim3(nblk(n_atoms, kBlock)), dim3(kBlock), 0, s, n_atoms, n_bins,
                       pos, species, w.wrap, w.binid, w.bin_count, w.bin_start, w.bin_atoms, w.recs);
  }
  hipLaunchKernelGGL(build_pairs_kernel, dim3(n_groups), dim3(64 * kBuildGroup), 0, s, n_atoms, nel, rmax,
                     pos, frame_of_atom, grids, w.wrap, w.binid, w.bin_start, w.recs, w.bin_count, n_bins, gstate,
                     capacity, w.seg_start, pair_start, host_pair_start, pair_i, pair_j, pair_shift, zero);
  // the reverse index right behind it, over the first `rev_cover` pairs (the caller's estimate of the count:
  // it launches the kernel again should the list turn out longer)
  if (pair_rev && rev_cover > 0)
    hipLaunchKernelGGL(reverse_sorted_kernel, dim3(nblk(rev_cover, kBlock)), dim3(kBlock), 0, s, (int64_t)rev_cover, nel,
                       zero + 4, (int64_t)capacity, species, w.seg_start, pair_i, pair_j, pair_shift, pair_rev,
                       reinterpret_cast<int32_t *>(zero) + 6);
}

void nl_reverse_sorted(int64_t n_pairs, int nel, const int32_t *species, const int32_t *seg_start,
                       const int32_t *pair_i, const int32_t *pair_j, const int32_t *pair_shift,
                       int32_t *pair_rev, unsigned long long *stats, hipStream_t s) {
  if (n_pairs > 0)
    hipLaunchKernelGGL(reverse_sorted_kernel, dim3(nblk(n_pairs, kBlock)), dim3(kBlock), 0, s, n_pairs, nel,
                       (const unsigned long long *)nullptr, (int64_t)0, species, seg_start, pair_i, pair_j, pair_shift,
                       pair_rev,
                       reinterpret_cast<int32_t *>(stats) + 6);
}

// Exact list of the current positions out of the resident skin list (all device, no host round trip):
// exact (i, j, S, rev), pair_start / pair_stop, seg_start, and the
// workgroup packing for the angular kernels when `blk_center` is given ([nl_filter_blocks(n_atoms) + 1]
// entries; the grid of those kernels is nl_filter_blocks(n_atoms)). `map` is a work buffer.
int nl_filter_blocks(int n_atoms) { return (int)nblk(n_atoms, kFilterGroup) * kFilterGroup; }

void nl_filter(int n_atoms, int64_t n_super, int nel, double rmax, const double *pos, const double *cells,
               const int32_t *frame_of_atom, const int32_t *start_super, const int32_t *seg_super,
               const int32_t *pj_super, const int32_t *ps_super, const int32_t *rev_super, int32_t *map,
               int32_t *seg_exact, int32_t *pair_start, int32_t *pair_stop, int32_t *pi_out, int32_t *pj_out,
               int32_t *ps_out, int32_t *rev_out, int32_t *slot_q, int cap, int32_t *blk_center, hipStream_t s) {
  if (n_atoms == 0) return;
  if (nel > kFilterMaxEl) throw std::domain_error("the MD-step list filter handles at most 8 elements");
  hipLaunchKernelGGL(filter_group_kernel, dim3(nblk(n_atoms, kFilterGroup)), dim3(64 * kFilterGroup), 0, s, n_atoms,
                     nel, rmax, pos, cells, frame_of_atom, start_super, seg_super, pj_super, ps_super, seg_exact,
                     pair_start, pair_stop, pi_out, pj_out, ps_out, map, slot_q, cap, blk_center);
  // `slot_q` given: the one reader of the reverse index goes through the map itself (pair_rev_of)
  if (n_super > 0 && !slot_q)
    hipLaunchKernelGGL(filter_rev_kernel, dim3(nblk(n_super, kBlock)), dim3(kBlock), 0, s, n_super, rev_super, map,
                       rev_out);
}

}  // namespace ta
