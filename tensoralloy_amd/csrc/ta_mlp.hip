// Per-atom element-wise MLP, forward + backward-to-inputs, on the fp64 matrix
// cores (v_mfma_f64_16x16x4_f64), batched over the atoms of one element.
//
// Replaces `convolution1x1` (reference nn/convolutional.py:154-300) and the
// part of `tf.gradients` that flows through it, plus the min-max scaling of
// nn/atomic/atomic.py:157-195 and the squeeze to atomic energies :250-264.
//
// One workgroup (4 wavefronts) owns 16 atoms (the M dimension of the 16x16x4
// tile); wavefront w computes the 16-column output tiles w, w+4, ... so the
// four SIMDs of a CU work on one layer together. Layer
// inputs live in LDS as [16][width] row-major (A operand: lane l reads
// X[l & 15][4 kk + (l >> 4)]); weights stream from L2 (B operand: lane l reads
// W[4 kk + (l >> 4)][16 nt + (l & 15)], 16 consecutive doubles per k row). The
// f64 accumulator tile holds Z[(l >> 4) + 4 r][l & 15] in register r.
// Activation derivatives are parked in a global scratch slab for the backward
// sweep, which runs the same tiles against the transposed weights.
#include <hip/hip_runtime.h>

#include "ta_device.h"
#include "ta_math.h"
#include "ta_reduce.h"

namespace ta {
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;

// Z[16][np] = X[16][kp] . W[kp][np] (+ bias), result handed to `emit(row, col, z)`.
// Column tiles are dealt round-robin to the workgroup's wavefronts; the B
// operands of 8 k-steps are fetched before their MFMAs so one L2 latency
// covers 8 matrix instructions.
template <typename Emit>
__device__ __forceinline__ void tile_gemm(const double *X, int xstride, const double *W, int wstride,
                                          int kp, int np, const double *bias, int lane, int wave,
                                          int nwaves, Emit emit) {
  const int m = lane & 15, kq = lane >> 4;
  for (int nt = wave; nt < np / 16; nt += nwaves) {
    const int col = 16 * nt + m;
    const double b0 = bias ? bias[col] : 0.0;
    double4_t acc = {b0, b0, b0, b0};
    const int nk = kp / 4;  // kp is a multiple of 16 -> nk is a multiple of 4
    for (int kk0 = 0; kk0 < nk; kk0 += 4) {
      double a[4], w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w[j] = W[(size_t)(4 * (kk0 + j) + kq) * wstride + col];
        a[j] = X[m * xstride + 4 * (kk0 + j) + kq];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], w[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) emit(kq + 4 * r, col, acc[r]);
  }
}

constexpr int kMlpThreads = 256;

// `fused` != 0: the workgroup first assembles the descriptors of its 16 atoms
// (16 lanes per atom) instead of reading them from G; G is still written.
__global__ __launch_bounds__(kMlpThreads) void mlp_kernel(SFParams sf, DeviceBatch db, MlpDev mlp,
                                                 int act, int ndim,
                                                 const int32_t *atoms, int n_atoms,
                                                 double *G, double *dEdG, double *eatom,
                                                 double *scratch, int stride, int fused) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kRows * stride;
  double *wl = lds + 2 * kRows * stride;  // LDS copy of every layer's W and W^T (when it fits)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = kMlpThreads / 64;
  const int a0 = blockIdx.x * kRows;
  const int L = mlp.n_layers;
  // activation derivatives: LDS when the plan has room for them, else a global scratch slab
  double *da = mlp.da_in_lds ? wl + mlp.lds_w_doubles
                             : scratch + (size_t)blockIdx.x * L * kRows * stride;

  if (mlp.w_in_lds) {
    // one streaming copy of the host-prepared LDS image (all layers' W and W^T, rows padded so
    // that the two k-rows a 32-lane LDS access touches fall on different banks)
    const double2 *src = reinterpret_cast<const double2 *>(mlp.lds_image);
    double2 *dst = reinterpret_cast<double2 *>(wl);
    for (int idx = tid; idx < mlp.lds_w_doubles / 2; idx += kMlpThreads) dst[idx] = src[idx];
  }
  if (fused) {
    const int row = tid >> 4, l = tid & 15;
    const bool active = a0 + row < n_atoms;
    const int64_t i = active ? atoms[a0 + row] : 0;
    double *Gi = G + (size_t)i * ndim;
    double *Gl = buf1 + row * stride;  // raw descriptors handed over through LDS
    atom_descriptors<16>(sf, db, i, l, active, [&](int c, double v) {
      Gi[c] = v;
      Gl[c] = v;
    });
    __syncthreads();
  }
  // layer-0 input: (optionally min-max scaled) descriptors, zero padded
  const int kp0 = mlp.layer[0].kp;
  for (int idx = tid; idx < kRows * kp0; idx += kMlpThreads) {
    const int row = idx / kp0, k = idx - row * kp0;
    double x = 0.0;
    if (a0 + row < n_atoms && k < ndim) {
      x = fused ? buf1[row * stride + k] : G[(size_t)atoms[a0 + row] * ndim + k];
      if (mlp.xlo) {
        const double den = mlp.xhi[k] - mlp.xlo[k];
        x = (den != 0.0) ? (mlp.xhi[k] - x) / den : 0.0;  // div_no_nan, atomic.py:195
      }
    }
    buf0[row * stride + k] = x;
  }
  __syncthreads();

  double *cur = buf0, *nxt = buf1;
  for (int l = 0; l < L; ++l) {
    const MlpLayerDev ly = mlp.layer[l];
    double *dal = da + (size_t)l * kRows * stride;
    const double *Wsrc = mlp.w_in_lds ? wl + ly.lds_w : ly.w;
    const int wstride = mlp.w_in_lds ? ly.ws : ly.np;
    tile_gemm(cur, stride, Wsrc, wstride, ly.kp, ly.np, ly.b, lane, wave, nwaves, [&](int row, int col, double z) {
      double h = z, dh = 1.0;
      if (ly.act) activation_fn(act, z, h, dh);
      if (ly.res) h += cur[row * stride + col];  // convolutional.py:272-273
      nxt[row * stride + col] = h;
      dal[row * stride + col] = dh;
    });
    __syncthreads();
    double *t = cur;
    cur = nxt;
    nxt = t;
  }
  // atomic energies: column 0 of the (padded) output layer
  if (tid < kRows && a0 + tid < n_atoms) eatom[atoms[a0 + tid]] = cur[tid * stride];
  __syncthreads();

  // backward: delta = dE_atom / d(layer output); start from the output column
  const int npL = mlp.layer[L - 1].np;
  for (int idx = tid; idx < kRows * npL; idx += kMlpThreads) {
    const int row = idx / npL, col = idx - row * npL;
    cur[row * stride + col] = (col == 0) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const MlpLayerDev ly = mlp.layer[l];
    const double *dal = da + (size_t)l * kRows * stride;
    // dz = delta * act'(z), in place; keep delta for the skip connection
    if (ly.res) {
      for (int idx = tid; idx < kRows * ly.np; idx += kMlpThreads) {
        const int row = idx / ly.np, col = idx - row * ly.np;
        nxt[row * stride + col] = cur[row * stride + col];
      }
    }
    for (int idx = tid; idx < kRows * ly.np; idx += kMlpThreads) {
      const int row = idx / ly.np, col = idx - row * ly.np;
      cur[row * stride + col] *= dal[row * stride + col];
    }
    __syncthreads();
    // delta_prev[16][kp] = dz[16][np] . W^T[np][kp]  (+ delta when skip)
    const bool res = ly.res != 0;
    double *dst = nxt;
    const double *Wtsrc = mlp.w_in_lds ? wl + ly.lds_wt : ly.wt;
    const int wtstride = mlp.w_in_lds ? ly.wts : ly.kp;
    tile_gemm(cur, stride, Wtsrc, wtstride, ly.np, ly.kp, nullptr, lane, wave, nwaves, [&](int row, int col, double z) {
      const double skip = res ? dst[row * stride + col] : 0.0;
      dst[row * stride + col] = z + skip;
    });
    __syncthreads();
    double *t = cur;
    cur = nxt;
    nxt = t;
  }
  for (int idx = tid; idx < kRows * ndim; idx += kMlpThreads) {
    const int row = idx / ndim, k = idx - row * ndim;
    if (a0 + row >= n_atoms) continue;
    double d = cur[row * stride + k];
    if (mlp.xlo) {
      const double den = mlp.xhi[k] - mlp.xlo[k];
      d = (den != 0.0) ? -d / den : 0.0;
    }
    dEdG[(size_t)atoms[a0 + row] * ndim + k] = d;
  }
}

}  // namespace

// scratch doubles needed per 16-atom tile
size_t mlp_scratch_doubles(const MlpDev &mlp) {
  const int w = mlp.max_np > mlp.max_kp ? mlp.max_np : mlp.max_kp;
  return (size_t)mlp.n_layers * kRows * (w + 2);
}

void launch_mlp_impl(const SFParams &sf, const MlpDev &mlp, int activation, int ndim,
                     const int32_t *atoms, int n_atoms, const DeviceBatch &b, double *scratch,
                     bool fused, hipStream_t s) {
  if (n_atoms == 0) return;
  const int w = mlp.max_np > mlp.max_kp ? mlp.max_np : mlp.max_kp;
  const int stride = w + 2;
  const size_t lds = (2 * (size_t)kRows * stride + (mlp.w_in_lds ? mlp.lds_w_doubles : 0) +
                      (mlp.da_in_lds ? (size_t)mlp.n_layers * kRows * stride : 0)) * sizeof(double);
  const unsigned blocks = (unsigned)((n_atoms + kRows - 1) / kRows);
  hipLaunchKernelGGL(mlp_kernel, dim3(blocks), dim3(kMlpThreads), lds, s, sf, b, mlp, activation,
                     ndim, atoms, n_atoms, b.G, b.dEdG, b.eatom, scratch, stride, fused ? 1 : 0);
}

}  // namespace ta
