// Stand-alone per-atom element-wise MLP kernel: 16 atoms of one element per
// workgroup, forward + backward-to-inputs on the fp64 matrix cores (see
// ta_mlp_tile.h for the tile code and the reference ops it replaces). Used when
// the fused per-centre kernel does not apply (radial-only models, several
// parameter chunks, more than 3 elements, very large neighbour counts).
#include <hip/hip_runtime.h>

#include "ta_device.h"
#include "ta_mlp_tile.h"

namespace ta {
namespace {

constexpr int kMlpThreads = 256;

__global__ __launch_bounds__(kMlpThreads) void mlp_kernel(MlpDev mlp, int act, int ndim,
                                                          const int32_t *atoms, int n_atoms,
                                                          const double *G, double *dEdG,
                                                          double *eatom, double *scratch, int stride) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride;
  const int a0 = blockIdx.x * kMlpRows;
  const int nrows = min(kMlpRows, n_atoms - a0);
  double *da = scratch + (size_t)blockIdx.x * mlp.n_layers * kMlpRows * stride;
  for (int idx = threadIdx.x; idx < kMlpRows * ndim; idx += kMlpThreads) {
    const int row = idx / ndim, k = idx - row * ndim;
    buf0[row * stride + k] = row < nrows ? G[(size_t)atoms[a0 + row] * ndim + k] : 0.0;
  }
  __syncthreads();
  mlp_tile(
      mlp, act, ndim, nrows, buf0, buf1, stride, da,
      [&](int row, double y) { eatom[atoms[a0 + row]] = y; },
      [&](int row, int k, double d) { dEdG[(size_t)atoms[a0 + row] * ndim + k] = d; });
}

}  // namespace

// scratch doubles needed per 16-row tile
size_t mlp_scratch_doubles(const MlpDev &mlp) {
  return (size_t)mlp.n_layers * kMlpRows * mlp_stride(mlp);
}

void launch_mlp_impl(const MlpDev &mlp, int activation, int ndim, const int32_t *atoms, int n_atoms,
                     const DeviceBatch &b, double *scratch, hipStream_t s) {
  if (n_atoms == 0) return;
  const int stride = mlp_stride(mlp);
  const size_t lds = 2 * (size_t)kMlpRows * stride * sizeof(double);
  const unsigned blocks = (unsigned)((n_atoms + kMlpRows - 1) / kMlpRows);
  hipLaunchKernelGGL(mlp_kernel, dim3(blocks), dim3(kMlpThreads), lds, s, mlp, activation, ndim, atoms,
                     n_atoms, b.G, b.dEdG, b.eatom, scratch, stride);
}

}  // namespace ta
