// Per-atom descriptor assembly shared by the stand-alone reduce kernel and the
// MLP kernel's prologue.
#pragma once
#include <hip/hip_runtime.h>

#include "ta_device.h"
#include "ta_math.h"

namespace ta {

template <int W>
__device__ __forceinline__ double group_sum(double v) {
  static_assert(W == 16 || W == 32 || W == 64, "groups are one DPP row, half a wavefront or a wavefront");
  if constexpr (W == 16) {
    return row16_sum(v);
  } else if constexpr (W == 32) {
    v = row16_sum(v);
    return v + __shfl_xor(v, 16, 64);
  } else {
    return wave_sum(v);
  }
}

__device__ __forceinline__ int radial_term_of(int center, int other) {
  // [AA, AB (B != A, sorted)]  (reference utils.py:265-273)
  return other == center ? 0 : (other < center ? other + 1 : other);
}
__device__ __forceinline__ int angular_term_of(int s1, int s2, int nel) {
  // sorted pair (j <= k) in row-major upper-triangular order (utils.py:274-282)
  int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
  return a * nel - (a * (a - 1)) / 2 + (b - a);
}

// Descriptors of atom i by a group of W lanes (l = lane index inside the group):
// G2 straight from the pair records (sf.py:79-119), G4 from the per-pair partial
// sums of the angular kernels (sf.py:121-182), concatenated as sf.py:184-215.
// `emit(channel, value)` is called by lane 0 of the group.
template <int W, typename Emit>
__device__ __forceinline__ void atom_descriptors(const SFParams &sf, const DeviceBatch &b,
                                                 int64_t i, int l, bool active, Emit emit) {
  const int nel = sf.n_elements;
  const int sA = active ? b.species[i] : 0;
  const int32_t *seg = b.seg_start + (size_t)(active ? i : 0) * (nel + 1);
  for (int sb = 0; sb < nel; ++sb) {
    const int tr = radial_term_of(sA, sb);
    const int q0 = active ? seg[sb] : 0, q1 = active ? seg[sb + 1] : 0;
    // four channels per sweep over the pairs: one load of r^2 serves four Gaussians
    for (int c0 = 0; c0 < sf.n_rad; c0 += 4) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int q = q0 + l; q < q1; q += W) {
        const double r2 = pair_geom(b, (size_t)q)[1].y;
        const double u = r2 * sf.inv_rc2;
        const double r = sqrt(r2);
        const double f = (u < 1.0) ? cutoff_u_value(sf.cutoff, u) : 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = (c0 + k < sf.n_rad) ? c0 + k : c0;
          const double dr = r - sf.omega[c];
          acc[k] += ta_exp(-sf.eta[c] * dr * dr * sf.inv_rc2) * f;  // sf.py:101-108
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double v = group_sum<W>(acc[k]);
        if (l == 0 && active && c0 + k < sf.n_rad) emit(tr * sf.n_rad + c0 + k, v);
      }
    }
  }
  if (sf.angular) {
    for (int s1 = 0; s1 < nel; ++s1)
      for (int s2 = s1; s2 < nel; ++s2) {
        const int t = angular_term_of(s1, s2, nel);
        const int a0 = active ? seg[s1] : 0, a1 = active ? seg[s1 + 1] : 0;
        const int b0 = active ? seg[s2] : 0, b1 = active ? seg[s2 + 1] : 0;
        for (int c0 = 0; c0 < sf.n_ang; c0 += 4) {
          double acc[4] = {0.0, 0.0, 0.0, 0.0};
          const size_t P = (size_t)b.n_pairs;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = (c0 + k < sf.n_ang) ? c0 + k : c0;
            const double *col = b.part4 + (size_t)(s2 * sf.n_ang + c) * P;
            for (int q = a0 + l; q < a1; q += W) acc[k] += col[q];
            if (s1 != s2) {
              const double *col2 = b.part4 + (size_t)(s1 * sf.n_ang + c) * P;
              for (int q = b0 + l; q < b1; q += W) acc[k] += col2[q];
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double v = group_sum<W>(acc[k]);
            // first-generation kernels visit every {j, k} from both sides (ang_scale = 0.5)
            if (l == 0 && active && c0 + k < sf.n_ang)
              emit(sf.n_radial_dim + t * sf.n_ang + c0 + k, sf.ang_scale * v);
          }
        }
      }
  }
}

// Per-frame sums, first level. The kernels that finish the per-atom values (force_gather here, the
// EAM force kernel) own groups of 16 consecutive atoms; each group leaves ONE record {E, W[9]} in
// `bpart` (fixed order: atom after atom), so that frame_reduce reads N / 16 records of 80 bytes
// instead of N strided per-atom rows through a single CU (6.4 of its 8.8 us for the 4000-atom frame).
// A group that straddles a frame boundary cannot be attributed: its atoms keep their per-atom rows in
// `wat` / `eatom` and frame_reduce picks those up one by one (at most 30 atoms per frame).
// `row_leader`: the lane that holds atom i's virial `w` (valid when `active`); all 256 lanes call.
__device__ __forceinline__ void block_partials(const DeviceBatch &b, int group, int64_t i, bool active,
                                               bool row_leader, const double (&w)[9]) {
  __shared__ double part[16][10];
  // 16 atoms per workgroup: 16 lanes per atom in force_gather, 16 / 32 / 64 in the EAM force kernel
  const int slot = (int)(threadIdx.x / (blockDim.x >> 4));
  const int64_t first = (int64_t)group * 16, last = min(first + 16, b.n_atoms) - 1;
  const bool one_frame = b.frame_of_atom[first] == b.frame_of_atom[last];
  if (row_leader) {
    part[slot][0] = active ? b.eatom[i] : 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) part[slot][1 + k] = active ? w[k] : 0.0;
    if (active && !one_frame)
      for (int k = 0; k < 9; ++k) b.wat[9 * (size_t)i + k] = w[k];
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    double v = 0.0;
    for (int a = 0; a < 16; ++a) v += part[a][threadIdx.x];
    b.bpart[10 * (size_t)group + threadIdx.x] = one_frame ? v : 0.0;
  }
}

}  // namespace ta
