// Gradient of a batch loss with respect to the MLP weights (first part).
//
// Replaces, for the energy term of the loss, what `tf.gradients(total_loss, trainable_variables)`
// does in the reference's training graph (nn/basic.py:get_total_loss -> nn/losses.py:204-285,
// nn/opt.py:89-166): with E_f = sum of the atomic energies of frame f and a loss L(E_1 .. E_F),
//     dL/dtheta = sum_atoms c[frame(atom)] * d y_atom / d theta,     c_f = dL/dE_f (from the host).
// The descriptors do not depend on theta, so the resident batch's G is reused by every training
// step; this kernel only re-runs the per-element MLP on them.
//
// One workgroup = 16 atoms of one element (the M dimension of v_mfma_f64_16x16x4_f64), as in
// ta_mlp.hip. Forward keeps every layer's input x_l and activation derivative in a scratch slab;
// backward starts from delta = c_row and forms, per layer,
//     dW_l[k][n] = sum_rows x_l[row][k] dz[row][n]      (16 x 16 x 4 MFMA tiles, K = the 16 rows)
//     db_l[n]    = sum_rows dz[row][n]
// into this workgroup's slice of a partial buffer; a second kernel adds the slices in a fixed order
// (deterministic, no atomics).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "ta_device.h"
#include "ta_mlp_tile.h"

namespace ta {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;  // persistent workgroups of the gradient kernel

struct GradLayout {
  int w_off[kMaxLayers];  // offset of W_l (then b_l) inside one element's flat parameter block
  int b_off[kMaxLayers];
  int n_params;           // W and b of every layer (b of the output layer included)
};

__global__ __launch_bounds__(kThreads) void mlp_grad_kernel(MlpDev mlp, GradLayout lay, int act, int ndim,
                                                            const int32_t *atoms, int n_atoms,
                                                            const double *G, const int32_t *frame_of_atom,
                                                            const double *frame_coeff,
                                                            const double *row_coeff, double *scratch,
                                                            double *partial, int stride) {
  // rows: `atoms[.]` (null: the rows themselves); weight of a row: `row_coeff[row]` when given
  // (per-pair networks of nn-EAM), else c[frame of the atom]
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = kThreads >> 6;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int L = mlp.n_layers;
  // scratch: x_l [L][16][stride] then act' [L][16][stride]
  double *xs = scratch + (size_t)blockIdx.x * 2 * L * kMlpRows * stride;
  double *da = xs + (size_t)L * kMlpRows * stride;
  double *out = partial + (size_t)blockIdx.x * lay.n_params;
  for (int k = tid; k < lay.n_params; k += kThreads) out[k] = 0.0;
  // persistent workgroup: tiles of 16 atoms, strided; the partial sums stay in this
  // workgroup's slice (L2-resident) across tiles
  const int n_tiles = (n_atoms + kMlpRows - 1) / kMlpRows;
  for (int tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
  const int a0 = tile_id * kMlpRows;
  const int nrows = min(kMlpRows, n_atoms - a0);
  __syncthreads();

  // layer-0 input: (min-max scaled) descriptors, zero padded
  const int kp0 = mlp.layer[0].kp;
  for (int idx = tid; idx < kMlpRows * kp0; idx += kThreads) {
    const int row = idx / kp0, k = idx - row * kp0;
    double x = 0.0;
    if (row < nrows && k < ndim) {
      x = G[(size_t)(atoms ? atoms[a0 + row] : a0 + row) * ndim + k];
      if (mlp.xlo) {
        const double den = mlp.xhi[k] - mlp.xlo[k];
        x = (den != 0.0) ? (mlp.xhi[k] - x) / den : 0.0;
      }
    }
    buf1[row * stride + k] = x;
  }
  __syncthreads();

  double *cur = buf1, *nxt = buf0;
  for (int l = 0; l < L; ++l) {
    const MlpLayerDev ly = mlp.layer[l];
    double *xl = xs + (size_t)l * kMlpRows * stride, *dal = da + (size_t)l * kMlpRows * stride;
    for (int idx = tid; idx < kMlpRows * ly.kp; idx += kThreads) {
      const int row = idx / ly.kp, k = idx - row * ly.kp;
      xl[row * stride + k] = cur[row * stride + k];
    }
    mlp_tile_gemm<4>(cur, stride, ly.w, ly.np, ly.kp, ly.np, ly.b, lane, wave, nwaves,
                     [&](int row, int col, double z) {
                       double h = 0.0, dh = 0.0;
                       if (row < nrows) {
                         h = z;
                         dh = 1.0;
                         if (ly.act) activation_fn(act, z, h, dh);
                         if (ly.res) h += cur[row * stride + col];
                       }
                       nxt[row * stride + col] = h;
                       dal[row * stride + col] = dh;
                     });
    __syncthreads();
    double *t = cur;
    cur = nxt;
    nxt = t;
  }

  // backward from delta = c[frame(atom)] on the output column
  const int npL = mlp.layer[L - 1].np;
  for (int idx = tid; idx < kMlpRows * npL; idx += kThreads) {
    const int row = idx / npL, col = idx - row * npL;
    double c = 0.0;
    if (col == 0 && row < nrows) {
      const int id = atoms ? atoms[a0 + row] : a0 + row;
      c = row_coeff ? row_coeff[id] : frame_coeff[frame_of_atom[id]];
    }
    cur[row * stride + col] = c;
  }
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const MlpLayerDev ly = mlp.layer[l];
    const double *xl = xs + (size_t)l * kMlpRows * stride, *dal = da + (size_t)l * kMlpRows * stride;
    if (ly.res) {
      for (int idx = tid; idx < kMlpRows * ly.np; idx += kThreads) {
        const int row = idx / ly.np, col = idx - row * ly.np;
        nxt[row * stride + col] = cur[row * stride + col];
      }
    }
    for (int idx = tid; idx < kMlpRows * ly.np; idx += kThreads) {
      const int row = idx / ly.np, col = idx - row * ly.np;
      cur[row * stride + col] *= dal[row * stride + col];  // dz
    }
    __syncthreads();
    // dW[k][n] = sum_rows x[row][k] dz[row][n]: tiles (kt, nt) over the wavefronts
    const int nkt = ly.kp / 16, nnt = ly.np / 16;
    for (int tile = wave; tile < nkt * nnt; tile += nwaves) {
      const int kt = tile / nnt, nt = tile - kt * nnt;
      mlp_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = 4 * s + q4;
        const double a = xl[row * stride + 16 * kt + m16];   // A[m = k][kk = row]
        const double bq = cur[row * stride + 16 * nt + m16];  // B[kk = row][n]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + q4 + 4 * r, n = 16 * nt + m16;
        if (k < ly.k && n < ly.n) out[lay.w_off[l] + k * ly.n + n] += acc[r];
      }
    }
    for (int n = tid; n < ly.n; n += kThreads) {
      double s = 0.0;
      for (int row = 0; row < kMlpRows; ++row) s += cur[row * stride + n];
      out[lay.b_off[l] + n] += s;
    }
    // delta_prev = dz . W^T (+ delta when skip)
    const bool res = ly.res != 0;
    double *dst = nxt;
    mlp_tile_gemm<4>(cur, stride, ly.wt, ly.kp, ly.np, ly.kp, nullptr, lane, wave, nwaves,
                     [&](int row, int col, double z) {
                       const double skip = res ? dst[row * stride + col] : 0.0;
                       dst[row * stride + col] = z + skip;
                     });
    __syncthreads();
    double *t = cur;
    cur = nxt;
    nxt = t;
  }
  }  // tiles
}

// grad[p] = sum over workgroups of partial[blk][p], fixed order
__global__ __launch_bounds__(kThreads) void grad_reduce_kernel(const double *partial, int n_blocks,
                                                               int n_params, double *grad) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= n_params) return;
  double s = 0.0;
  for (int b = 0; b < n_blocks; ++b) s += partial[(size_t)b * n_params + p];
  grad[p] = s;
}

GradLayout make_layout(const MlpDev &mlp) {
  GradLayout lay;
  int off = 0;
  for (int l = 0; l < kMaxLayers; ++l) lay.w_off[l] = lay.b_off[l] = 0;
  for (int l = 0; l < mlp.n_layers; ++l) {
    lay.w_off[l] = off;
    off += mlp.layer[l].k * mlp.layer[l].n;
    lay.b_off[l] = off;
    off += mlp.layer[l].n;
  }
  lay.n_params = off;
  return lay;
}

}  // namespace

int mlp_param_count(const MlpDev &mlp) { return make_layout(mlp).n_params; }

// doubles of scratch / partial space one element needs for `n_atoms` atoms
size_t mlp_grad_scratch_doubles(const MlpDev &mlp, int n_atoms) {
  const size_t blocks = std::min<size_t>((size_t)(n_atoms + kMlpRows - 1) / kMlpRows, kMaxBlocks);
  return blocks * 2 * mlp.n_layers * kMlpRows * mlp_stride(mlp);
}
size_t mlp_grad_partial_doubles(const MlpDev &mlp, int n_atoms) {
  const size_t blocks = std::min<size_t>((size_t)(n_atoms + kMlpRows - 1) / kMlpRows, kMaxBlocks);
  return blocks * (size_t)make_layout(mlp).n_params;
}

// gradient of sum_f c_f E_f with respect to the flat parameters of one element -> grad (device)
void launch_mlp_grad(const MlpDev &mlp, int activation, int ndim, const int32_t *atoms, int n_atoms,
                     const DeviceBatch &b, const double *frame_coeff, double *scratch, double *partial,
                     double *grad, hipStream_t s) {
  const GradLayout lay = make_layout(mlp);
  if (n_atoms == 0) {
    (void)hipMemsetAsync(grad, 0, (size_t)lay.n_params * sizeof(double), s);
    return;
  }
  const int stride = mlp_stride(mlp);
  const size_t lds = 2 * (size_t)kMlpRows * stride * sizeof(double);
  const int blocks = std::min((n_atoms + kMlpRows - 1) / kMlpRows, kMaxBlocks);
  hipLaunchKernelGGL(mlp_grad_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, s, mlp, lay, activation,
                     ndim, atoms, n_atoms, b.G, b.frame_of_atom, frame_coeff, nullptr, scratch, partial, stride);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((lay.n_params + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, partial, blocks, lay.n_params, grad);
}

// The same for a scalar-input network evaluated on rows x[0 .. n_rows) with one weight per row
// (nn-EAM: rho / phi / u / w networks over the pairs); or, with `atoms`, on x[atoms[.]] weighted
// by c[frame of the atom] (the embedding networks over the atoms of one element).
void launch_mlp_grad_rows(const MlpDev &mlp, int activation, const int32_t *atoms, int n_rows,
                          const double *x, const double *row_coeff, const int32_t *frame_of_atom,
                          const double *frame_coeff, double *scratch, double *partial, double *grad,
                          hipStream_t s) {
  const GradLayout lay = make_layout(mlp);
  if (n_rows == 0) {
    (void)hipMemsetAsync(grad, 0, (size_t)lay.n_params * sizeof(double), s);
    return;
  }
  const int stride = mlp_stride(mlp);
  const size_t lds = 2 * (size_t)kMlpRows * stride * sizeof(double);
  const int blocks = std::min((n_rows + kMlpRows - 1) / kMlpRows, kMaxBlocks);
  hipLaunchKernelGGL(mlp_grad_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, s, mlp, lay, activation, 1,
                     atoms, n_rows, x, frame_of_atom, frame_coeff, row_coeff, scratch, partial, stride);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((lay.n_params + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, partial, blocks, lay.n_params, grad);
}

}  // namespace ta
