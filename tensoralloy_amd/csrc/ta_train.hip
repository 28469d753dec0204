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
#include "ta_reduce.h"

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

// ---- force / stress terms of the loss (nn/losses.py:285-437), analytically ---------------------------
// With u = dL/dF and the symmetric Y from dL/dstress, sum u.F + sum Y.W is the directional derivative
// D_delta E of the energy along delta R = R.Y - u, delta h = h.Y; for an MLP on descriptors that do
// not depend on the weights it is  J = sum_atoms w_i(theta) . dG_i,  w_i = dMLP/dG at G_i, dG_i the
// directional derivative of the descriptors (theta-free). dJ/dtheta is therefore a SECOND backward
// pass through the MLP only. Forward carries the value x and the tangent x' of every layer
//     z = W x + b,  z' = W x',  x_out = a(z) [+ x],  x'_out = a'(z) z' [+ x'],   s = z'_out
// and the reverse sweep two adjoints: kappa (of x) and nu (of x'):
//     mu = nu a'(z),  lambda = kappa a'(z) + nu a''(z) z',
//     dW += x^T lambda + x'^T mu,  db += sum lambda,  kappa_in = lambda W^T [+ kappa], nu_in = mu W^T [+ nu].
// Seeding kappa_out = c[frame] and nu_out = 1 yields d/dtheta (sum_f c_f E_f + J) in ONE pass: the
// whole loss gradient, energy term included. Same tiling as mlp_grad_kernel (16 atoms per tile).
__global__ __launch_bounds__(kThreads) void mlp_grad2_kernel(MlpDev mlp, GradLayout lay, int act, int ndim,
                                                             const int32_t *atoms, int n_atoms,
                                                             const double *G, const double *dG,
                                                             const int32_t *frame_of_atom,
                                                             const double *frame_coeff,
                                                             const double *row_coeff, double *scratch,
                                                             double *partial, int stride, double *kappa_out) {
  // `row_coeff` (scalar-input networks of nn-EAM over pairs): kappa seed of a row instead of c[frame]
  // `kappa_out` [rows][ndim] (may be null): the adjoint of the raw inputs, c dy/dG + (d^2 y / dG^2) dG: with
  // c = 0 the Hessian-vector product of the network (ta_hvp.hip)
  extern __shared__ double lds[];
  double *bufX0 = lds, *bufX1 = bufX0 + kMlpRows * stride, *bufT0 = bufX1 + kMlpRows * stride,
         *bufT1 = bufT0 + kMlpRows * stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = kThreads >> 6;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int L = mlp.n_layers;
  const size_t slab = (size_t)kMlpRows * stride;
  // scratch: x_l, x'_l, a'(z_l), a''(z_l) z'_l for every layer
  double *xs = scratch + (size_t)blockIdx.x * 4 * L * slab;
  double *ts = xs + (size_t)L * slab, *da = ts + (size_t)L * slab, *dd = da + (size_t)L * slab;
  double *out = partial + (size_t)blockIdx.x * lay.n_params;
  for (int k = tid; k < lay.n_params; k += kThreads) out[k] = 0.0;
  const int n_tiles = (n_atoms + kMlpRows - 1) / kMlpRows;
  for (int tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    const int a0 = tile_id * kMlpRows;
    const int nrows = min(kMlpRows, n_atoms - a0);
    __syncthreads();
    const int kp0 = mlp.layer[0].kp;
    for (int idx = tid; idx < kMlpRows * kp0; idx += kThreads) {
      const int row = idx / kp0, k = idx - row * kp0;
      double x = 0.0, xt = 0.0;
      if (row < nrows && k < ndim) {
        const size_t id = (size_t)(atoms ? atoms[a0 + row] : a0 + row);
        x = G[id * ndim + k];
        xt = dG[id * ndim + k];
        if (mlp.xlo) {  // x = (xhi - G) / (xhi - xlo): dx = -dG / (xhi - xlo)
          const double den = mlp.xhi[k] - mlp.xlo[k];
          x = (den != 0.0) ? (mlp.xhi[k] - x) / den : 0.0;
          xt = (den != 0.0) ? -xt / den : 0.0;
        }
      }
      bufX1[row * stride + k] = x;
      bufT1[row * stride + k] = xt;
    }
    __syncthreads();
    double *curX = bufX1, *nxtX = bufX0, *curT = bufT1, *nxtT = bufT0;
    for (int l = 0; l < L; ++l) {
      const MlpLayerDev ly = mlp.layer[l];
      double *xl = xs + l * slab, *tl = ts + l * slab, *dal = da + l * slab, *ddl = dd + l * slab;
      for (int idx = tid; idx < kMlpRows * ly.kp; idx += kThreads) {
        const int row = idx / ly.kp, k = idx - row * ly.kp;
        xl[row * stride + k] = curX[row * stride + k];
        tl[row * stride + k] = curT[row * stride + k];
      }
      // z' first (parked in nxtT), then z with the activation applied to both
      mlp_tile_gemm<4>(curT, stride, ly.w, ly.np, ly.kp, ly.np, nullptr, lane, wave, nwaves,
                       [&](int row, int col, double zt) { nxtT[row * stride + col] = zt; });
      mlp_tile_gemm<4>(curX, stride, ly.w, ly.np, ly.kp, ly.np, ly.b, lane, wave, nwaves,
                       [&](int row, int col, double z) {
                         double h = 0.0, dh = 0.0, d2 = 0.0, ht = 0.0;
                         if (row < nrows) {
                           const double zt = nxtT[row * stride + col];  // written by this same lane
                           h = z;
                           dh = 1.0;
                           if (ly.act) activation_fn2(act, z, h, dh, d2);
                           ht = dh * zt;
                           d2 *= zt;
                           if (ly.res) {
                             h += curX[row * stride + col];
                             ht += curT[row * stride + col];
                           }
                         }
                         nxtX[row * stride + col] = h;
                         nxtT[row * stride + col] = ht;
                         dal[row * stride + col] = dh;
                         ddl[row * stride + col] = d2;
                       });
      __syncthreads();
      double *t = curX; curX = nxtX; nxtX = t;
      t = curT; curT = nxtT; nxtT = t;
    }
    // seeds: kappa = c[frame] and nu = 1 on the output column
    const int npL = mlp.layer[L - 1].np;
    for (int idx = tid; idx < kMlpRows * npL; idx += kThreads) {
      const int row = idx / npL, col = idx - row * npL;
      double c = 0.0, one = 0.0;
      if (col == 0 && row < nrows) {
        const int id = atoms ? atoms[a0 + row] : a0 + row;
        c = row_coeff ? row_coeff[id] : (frame_coeff ? frame_coeff[frame_of_atom[id]] : 0.0);
        one = 1.0;
      }
      curX[row * stride + col] = c;    // kappa
      curT[row * stride + col] = one;  // nu
    }
    __syncthreads();
    for (int l = L - 1; l >= 0; --l) {
      const MlpLayerDev ly = mlp.layer[l];
      const double *xl = xs + l * slab, *tl = ts + l * slab, *dal = da + l * slab, *ddl = dd + l * slab;
      if (ly.res) {
        for (int idx = tid; idx < kMlpRows * ly.np; idx += kThreads) {
          const int row = idx / ly.np, col = idx - row * ly.np;
          nxtX[row * stride + col] = curX[row * stride + col];
          nxtT[row * stride + col] = curT[row * stride + col];
        }
      }
      for (int idx = tid; idx < kMlpRows * ly.np; idx += kThreads) {
        const int row = idx / ly.np, col = idx - row * ly.np;
        const double kap = curX[row * stride + col], nu = curT[row * stride + col];
        curX[row * stride + col] = kap * dal[row * stride + col] + nu * ddl[row * stride + col];  // lambda
        curT[row * stride + col] = nu * dal[row * stride + col];                                   // mu
      }
      __syncthreads();
      // dW[k][n] = sum_rows x[row][k] lambda[row][n] + x'[row][k] mu[row][n]
      const int nkt = ly.kp / 16, nnt = ly.np / 16;
      for (int tile = wave; tile < nkt * nnt; tile += nwaves) {
        const int kt = tile / nnt, nt = tile - kt * nnt;
        mlp_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int row = 4 * s + q4;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xl[row * stride + 16 * kt + m16],
                                                     curX[row * stride + 16 * nt + m16], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tl[row * stride + 16 * kt + m16],
                                                     curT[row * stride + 16 * nt + m16], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * kt + q4 + 4 * r, n = 16 * nt + m16;
          if (k < ly.k && n < ly.n) out[lay.w_off[l] + k * ly.n + n] += acc[r];
        }
      }
      for (int n = tid; n < ly.n; n += kThreads) {
        double s = 0.0;
        for (int row = 0; row < kMlpRows; ++row) s += curX[row * stride + n];
        out[lay.b_off[l] + n] += s;
      }
      const bool res = ly.res != 0;
      double *dstX = nxtX, *dstT = nxtT;
      mlp_tile_gemm<4>(curX, stride, ly.wt, ly.kp, ly.np, ly.kp, nullptr, lane, wave, nwaves,
                       [&](int row, int col, double z) {
                         dstX[row * stride + col] = z + (res ? dstX[row * stride + col] : 0.0);
                       });
      mlp_tile_gemm<4>(curT, stride, ly.wt, ly.kp, ly.np, ly.kp, nullptr, lane, wave, nwaves,
                       [&](int row, int col, double z) {
                         dstT[row * stride + col] = z + (res ? dstT[row * stride + col] : 0.0);
                       });
      __syncthreads();
      double *t = curX; curX = nxtX; nxtX = t;
      t = curT; curT = nxtT; nxtT = t;
    }
    if (kappa_out) {
      for (int idx = tid; idx < nrows * ndim; idx += kThreads) {
        const int row = idx / ndim, k = idx - row * ndim;
        const size_t id = (size_t)(atoms ? atoms[a0 + row] : a0 + row);
        double v = curX[row * stride + k];
        if (mlp.xlo) {
          const double den = mlp.xhi[k] - mlp.xlo[k];
          v = (den != 0.0) ? -v / den : 0.0;
        }
        kappa_out[id * ndim + k] = v;
      }
    }
  }
}

// pair part of the direction: dD_p = dR_j - dR_i + S.dh  (D = Rj - Ri + S.h, universal.py:463-468)
__global__ __launch_bounds__(kThreads) void pair_tangent_kernel(DeviceBatch b, const double *dR, const double *dh,
                                                                double *dD) {
  const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (p >= b.n_pairs) return;
  const int i = b.pair_i[p], j = b.pair_j[p];
  const double *h = dh + 9 * (size_t)b.frame_of_atom[i];
  const double sx = (double)b.pair_shift[3 * p], sy = (double)b.pair_shift[3 * p + 1], sz = (double)b.pair_shift[3 * p + 2];
  double *o = dD + 4 * (size_t)p;
  o[0] = dR[3 * (size_t)j] - dR[3 * (size_t)i] + (sx * h[0] + sy * h[3] + sz * h[6]);
  o[1] = dR[3 * (size_t)j + 1] - dR[3 * (size_t)i + 1] + (sx * h[1] + sy * h[4] + sz * h[7]);
  o[2] = dR[3 * (size_t)j + 2] - dR[3 * (size_t)i + 2] + (sx * h[2] + sy * h[5] + sz * h[8]);
  o[3] = 0.0;
}

// dG[i][c] = sum_{p in N(i)} J[c][p] . dD_p with J[c][p] = dG_{i,c} / dD_p (one backward launch per
// channel with a one-hot dE/dG, made once per resident batch); one wavefront per atom
__global__ __launch_bounds__(kThreads) void descriptor_jvp_kernel(DeviceBatch b, int ndim, const double *J,
                                                                  const double *dD, double *dG) {
  const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  const int p0 = b.pair_start[i], p1 = b.pair_start[i + 1];
  for (int c = 0; c < ndim; ++c) {
    const double *Jc = J + 4 * (size_t)c * (size_t)b.n_pairs;
    double s = 0.0;
    for (int p = p0 + lane; p < p1; p += 64) {
      const double2 a0 = reinterpret_cast<const double2 *>(Jc + 4 * (size_t)p)[0];
      const double a2 = Jc[4 * (size_t)p + 2];
      const double2 d0 = reinterpret_cast<const double2 *>(dD + 4 * (size_t)p)[0];
      const double d2 = dD[4 * (size_t)p + 2];
      s = fma(a0.x, d0.x, fma(a0.y, d0.y, fma(a2, d2, s)));
    }
    s = wave_sum(s);
    if (lane == 0) dG[(size_t)i * ndim + c] = s;
  }
}

__global__ __launch_bounds__(kThreads) void one_hot_kernel(double *dEdG, int64_t n, int ndim, int c) {
  const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (k < n * ndim) dEdG[k] = (k % ndim == c) ? 1.0 : 0.0;
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

size_t mlp_grad2_scratch_doubles(const MlpDev &mlp, int n_atoms) {
  const size_t blocks = std::min<size_t>((size_t)(n_atoms + kMlpRows - 1) / kMlpRows, kMaxBlocks);
  return blocks * 4 * mlp.n_layers * kMlpRows * mlp_stride(mlp);
}

// gradient of sum_f c_f E_f + D_delta E with respect to the flat parameters of one element; `dG` =
// directional derivative of the descriptors [N][ndim] (device)
void launch_mlp_grad2(const MlpDev &mlp, int activation, int ndim, const int32_t *atoms, int n_atoms,
                      const DeviceBatch &b, const double *dG, const double *frame_coeff, double *scratch,
                      double *partial, double *grad, hipStream_t s, double *kappa_out) {
  const GradLayout lay = make_layout(mlp);
  if (n_atoms == 0) {
    (void)hipMemsetAsync(grad, 0, (size_t)lay.n_params * sizeof(double), s);
    return;
  }
  const int stride = mlp_stride(mlp);
  const size_t lds = 4 * (size_t)kMlpRows * stride * sizeof(double);
  const int blocks = std::min((n_atoms + kMlpRows - 1) / kMlpRows, kMaxBlocks);
  hipLaunchKernelGGL(mlp_grad2_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, s, mlp, lay, activation,
                     ndim, atoms, n_atoms, b.G, dG, b.frame_of_atom, frame_coeff, nullptr, scratch, partial, stride, kappa_out);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((lay.n_params + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, partial, blocks, lay.n_params, grad);
}

void launch_pair_tangent(const DeviceBatch &b, const double *dR, const double *dh, double *dD, hipStream_t s) {
  if (b.n_pairs == 0) return;
  hipLaunchKernelGGL(pair_tangent_kernel, dim3((unsigned)((b.n_pairs + kThreads - 1) / kThreads)), dim3(kThreads),
                     0, s, b, dR, dh, dD);
}
void launch_descriptor_jvp(const DeviceBatch &b, int ndim, const double *J, const double *dD, double *dG,
                           hipStream_t s) {
  if (b.n_atoms == 0) return;
  hipLaunchKernelGGL(descriptor_jvp_kernel, dim3((unsigned)((b.n_atoms * 64 + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, b, ndim, J, dD, dG);
}
void launch_one_hot(double *dEdG, int64_t n_atoms, int ndim, int c, hipStream_t s) {
  if (n_atoms == 0) return;
  hipLaunchKernelGGL(one_hot_kernel, dim3((unsigned)((n_atoms * ndim + kThreads - 1) / kThreads)), dim3(kThreads),
                     0, s, dEdG, n_atoms, ndim, c);
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

// Second-order pass for a scalar-input network: gradient of  sum_rows [a_row f(x_row) + b_row f'(x_row)]
// with respect to the flat parameters (a = `row_coeff`, or c[frame of the atom] when null; b = `xdot`).
// The force / stress-loss gradient of the nn functions of an EAM model is made of such sums
// (ta_eam.hip::eam_loss_gradient).
void launch_mlp_grad2_rows(const MlpDev &mlp, int activation, const int32_t *atoms, int n_rows, const double *x,
                           const double *xdot, const double *row_coeff, const int32_t *frame_of_atom,
                           const double *frame_coeff, double *scratch, double *partial, double *grad,
                           hipStream_t s) {
  const GradLayout lay = make_layout(mlp);
  if (n_rows == 0) {
    (void)hipMemsetAsync(grad, 0, (size_t)lay.n_params * sizeof(double), s);
    return;
  }
  const int stride = mlp_stride(mlp);
  const size_t lds = 4 * (size_t)kMlpRows * stride * sizeof(double);
  const int blocks = std::min((n_rows + kMlpRows - 1) / kMlpRows, kMaxBlocks);
  hipLaunchKernelGGL(mlp_grad2_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, s, mlp, lay, activation, 1,
                     atoms, n_rows, x, xdot, frame_of_atom, frame_coeff, row_coeff, scratch, partial, stride, (double *)nullptr);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((lay.n_params + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, partial, blocks, lay.n_params, grad);
}

}  // namespace ta
