// 16-row MLP tile on the fp64 matrix cores, shared by the stand-alone MLP
// kernel and the fused per-centre kernel.
//
// Replaces `convolution1x1` (reference nn/convolutional.py:154-300) and the
// part of `tf.gradients` that flows through it, plus the min-max scaling of
// nn/atomic/atomic.py:157-195.
//
// One workgroup owns 16 rows (the M dimension of v_mfma_f64_16x16x4_f64); its
// wavefronts split the 16-column output tiles. Layer inputs live in LDS as
// [16][stride] row-major (A operand: lane l reads X[l & 15][4 kk + (l >> 4)]);
// weights stream from L2 (B operand: lane l reads W[4 kk + (l >> 4)][16 nt +
// (l & 15)], 16 consecutive doubles per k row) in batches of 4 k-steps. The f64
// accumulator tile holds Z[(l >> 4) + 4 r][l & 15] in register r. Activation
// derivatives are parked in `da` (global scratch or LDS) for the backward
// sweep, which runs the same tiles against the transposed weights.
#pragma once
#include <hip/hip_runtime.h>

#include "ta_device.h"
#include "ta_math.h"

namespace ta {

typedef double mlp_f64x4 __attribute__((ext_vector_type(4)));
constexpr int kMlpRows = 16;

// Z[16][np] = X[16][kp] . W[kp][np] (+ bias), result handed to `emit(row, col, z)`.
// PF = k-steps whose B operands are fetched from L2 before the first of them is used (4 or 16:
// one dependent load round per PF k-steps).
template <int PF, typename Emit>
__device__ __forceinline__ void mlp_tile_gemm(const double *X, int xstride, const double *W,
                                              int wstride, int kp, int np, const double *bias,
                                              int lane, int wave, int nwaves, Emit emit) {
  const int m = lane & 15, kq = lane >> 4;
  for (int nt = wave; nt < np / 16; nt += nwaves) {
    const int col = 16 * nt + m;
    const double b0 = bias ? bias[col] : 0.0;
    mlp_f64x4 acc = {b0, b0, b0, b0};
    const int nk = kp / 4;  // kp is a multiple of 16 -> nk is a multiple of 4
    int kk0 = 0;
    if constexpr (PF > 4) {
      for (; kk0 + PF <= nk; kk0 += PF) {
        double w[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) w[j] = W[(size_t)(4 * (kk0 + j) + kq) * wstride + col];
#pragma unroll
        for (int j = 0; j < PF; ++j)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[m * xstride + 4 * (kk0 + j) + kq], w[j], acc, 0, 0, 0);
      }
    }
    for (; kk0 < nk; kk0 += 4) {
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

// Forward + backward-to-inputs for 16 rows. On entry buf0[row][k] holds the RAW
// descriptors of the rows (k < ndim); rows >= nrows are ignored. All threads of
// the workgroup must call it. `emit_energy(row, y)` and `emit_grad(row, k, dE/dG)`
// are called for row < nrows.
template <int PF = 4, typename EmitE, typename EmitG>
__device__ __forceinline__ void mlp_tile(const MlpDev &mlp, int act, int ndim, int nrows,
                                         double *buf0, double *buf1, int stride, double *da,
                                         EmitE emit_energy, EmitG emit_grad) {
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  const int L = mlp.n_layers;

  // layer-0 input: (optionally min-max scaled) descriptors, zero padded
  const int kp0 = mlp.layer[0].kp;
  for (int idx = tid; idx < kMlpRows * kp0; idx += nthreads) {
    const int row = idx / kp0, k = idx - row * kp0;
    double x = 0.0;
    if (row < nrows && k < ndim) {
      x = buf0[row * stride + k];
      if (mlp.xlo) {
        const double den = mlp.xhi[k] - mlp.xlo[k];
        x = (den != 0.0) ? (mlp.xhi[k] - x) / den : 0.0;  // div_no_nan, atomic.py:195
      }
    }
    buf1[row * stride + k] = x;
  }
  __syncthreads();

  double *cur = buf1, *nxt = buf0;
  for (int l = 0; l < L; ++l) {
    const MlpLayerDev ly = mlp.layer[l];
    double *dal = da + (size_t)l * kMlpRows * stride;
    mlp_tile_gemm<PF>(cur, stride, ly.w, ly.np, ly.kp, ly.np, ly.b, lane, wave, nwaves,
                  [&](int row, int col, double z) {
                    // rows beyond nrows are padding: no transcendental work for them
                    double h = 0.0, dh = 0.0;
                    if (row < nrows) {
                      h = z;
                      dh = 1.0;
                      if (ly.act) activation_fn(act, z, h, dh);
                      if (ly.res) h += cur[row * stride + col];  // convolutional.py:272-273
                    }
                    nxt[row * stride + col] = h;
                    dal[row * stride + col] = dh;
                  });
    __syncthreads();
    double *t = cur;
    cur = nxt;
    nxt = t;
  }
  // atomic energies: column 0 of the (padded) output layer
  if (tid < nrows) emit_energy(tid, cur[tid * stride]);
  __syncthreads();

  // backward: delta = dE_atom / d(layer output); start from the output column
  const int npL = mlp.layer[L - 1].np;
  for (int idx = tid; idx < kMlpRows * npL; idx += nthreads) {
    const int row = idx / npL, col = idx - row * npL;
    cur[row * stride + col] = (col == 0) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const MlpLayerDev ly = mlp.layer[l];
    const double *dal = da + (size_t)l * kMlpRows * stride;
    // dz = delta * act'(z), in place; keep delta for the skip connection
    if (ly.res) {
      for (int idx = tid; idx < kMlpRows * ly.np; idx += nthreads) {
        const int row = idx / ly.np, col = idx - row * ly.np;
        nxt[row * stride + col] = cur[row * stride + col];
      }
    }
    for (int idx = tid; idx < kMlpRows * ly.np; idx += nthreads) {
      const int row = idx / ly.np, col = idx - row * ly.np;
      cur[row * stride + col] *= dal[row * stride + col];
    }
    __syncthreads();
    // delta_prev[16][kp] = dz[16][np] . W^T[np][kp]  (+ delta when skip)
    const bool res = ly.res != 0;
    double *dst = nxt;
    mlp_tile_gemm<PF>(cur, stride, ly.wt, ly.kp, ly.np, ly.kp, nullptr, lane, wave, nwaves,
                  [&](int row, int col, double z) {
                    const double skip = res ? dst[row * stride + col] : 0.0;
                    dst[row * stride + col] = z + skip;
                  });
    __syncthreads();
    double *t = cur;
    cur = nxt;
    nxt = t;
  }
  for (int idx = tid; idx < kMlpRows * ndim; idx += nthreads) {
    const int row = idx / ndim, k = idx - row * ndim;
    if (row >= nrows) continue;
    double d = cur[row * stride + k];
    if (mlp.xlo) {
      const double den = mlp.xhi[k] - mlp.xlo[k];
      d = (den != 0.0) ? -d / den : 0.0;
    }
    emit_grad(row, k, d);
  }
  __syncthreads();
}

inline int mlp_stride(const MlpDev &mlp) {
  const int w = mlp.max_np > mlp.max_kp ? mlp.max_np : mlp.max_kp;
  return w + 2;
}

}  // namespace ta
