// Stand-alone per-atom element-wise MLP kernel: 16 atoms of one element per
// workgroup, forward + backward-to-inputs on the fp64 matrix cores (see
// ta_mlp_tile.h for the tile code and the reference ops it replaces). Used when
// the fused per-centre kernel does not apply (radial-only models, several
// parameter chunks, more than 3 elements, very large neighbour counts).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "ta_device.h"
#include "ta_mlp_tile.h"

namespace ta {
namespace {

constexpr int kMlpThreads = 256;

template <int THREADS>
__global__ __launch_bounds__(THREADS) void mlp_kernel(MlpDev mlp, int act, int ndim,
                                                          const int32_t *atoms, int n_atoms,
                                                          const double *G, double *dEdG,
                                                          double *eatom, double *scratch, int stride) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride;
  const int a0 = blockIdx.x * kMlpRows;
  const int nrows = min(kMlpRows, n_atoms - a0);
  // activation derivatives of the forward sweep: behind the two activation buffers in LDS when the
  // launch reserved room for them (scratch == nullptr), else in the global scratch slab
  double *da = scratch ? scratch + (size_t)blockIdx.x * mlp.n_layers * kMlpRows * stride
                       : lds + 2 * kMlpRows * stride;
  for (int idx = threadIdx.x; idx < kMlpRows * ndim; idx += THREADS) {
    const int row = idx / ndim, k = idx - row * ndim;
    buf0[row * stride + k] = row < nrows ? G[(size_t)atoms[a0 + row] * ndim + k] : 0.0;
  }
  __syncthreads();
  mlp_tile<16>(
      mlp, act, ndim, nrows, buf0, buf1, stride, da,
      [&](int row, double y) { eatom[atoms[a0 + row]] = y; },
      [&](int row, int k, double d) { dEdG[(size_t)atoms[a0 + row] * ndim + k] = d; });
}

// All elements in one launch: blocks are laid out element after element, so the small
// per-element grids of an alloy run side by side instead of one launch after another.
struct MlpTiles {
  int32_t tile_start[kMaxElements + 1];  // first block of every element
  int32_t elem_start[kMaxElements + 1];  // first entry of every element in `atoms`
  int nel;
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void mlp_all_kernel(const MlpDev *__restrict__ mlps, MlpTiles tiles, int act,
                                                              int ndim, const int32_t *atoms,
                                                              const double *G, double *dEdG,
                                                              double *eatom, double *scratch,
                                                              int stride, int tile_doubles) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride;
  int e = 0;
  while (e + 1 < tiles.nel && (int)blockIdx.x >= tiles.tile_start[e + 1]) ++e;
  const MlpDev &mlp = mlps[e];
  const int32_t *el_atoms = atoms + tiles.elem_start[e];
  const int n_atoms = tiles.elem_start[e + 1] - tiles.elem_start[e];
  const int a0 = ((int)blockIdx.x - tiles.tile_start[e]) * kMlpRows;
  const int nrows = min(kMlpRows, n_atoms - a0);
  double *da = scratch ? scratch + (size_t)blockIdx.x * tile_doubles : lds + 2 * kMlpRows * stride;
  for (int idx = threadIdx.x; idx < kMlpRows * ndim; idx += THREADS) {
    const int row = idx / ndim, k = idx - row * ndim;
    buf0[row * stride + k] = row < nrows ? G[(size_t)el_atoms[a0 + row] * ndim + k] : 0.0;
  }
  __syncthreads();
  mlp_tile<16>(
      mlp, act, ndim, nrows, buf0, buf1, stride, da,
      [&](int row, double y) { eatom[el_atoms[a0 + row]] = y; },
      [&](int row, int k, double d) { dEdG[(size_t)el_atoms[a0 + row] * ndim + k] = d; });
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
  size_t lds = 2 * (size_t)kMlpRows * stride * sizeof(double);
  const unsigned blocks = (unsigned)((n_atoms + kMlpRows - 1) / kMlpRows);
  const size_t lds_da = (size_t)mlp.n_layers * kMlpRows * stride * sizeof(double);
  if (lds + lds_da <= 64 * 1024 && !getenv("TA_MLP_DA_GLOBAL")) {
    lds += lds_da;
    scratch = nullptr;
  }
  // one wavefront per 16-column tile of the widest layer, at most 8
  if (mlp.max_np >= 128)
    hipLaunchKernelGGL(mlp_kernel<512>, dim3(blocks), dim3(512), lds, s, mlp, activation, ndim, atoms,
                       n_atoms, b.G, b.dEdG, b.eatom, scratch, stride);
  else
    hipLaunchKernelGGL(mlp_kernel<kMlpThreads>, dim3(blocks), dim3(kMlpThreads), lds, s, mlp, activation,
                       ndim, atoms, n_atoms, b.G, b.dEdG, b.eatom, scratch, stride);
}

// one launch for every element; `mlps_dev` is the device copy of `mlps_host[0..nel)`
size_t mlp_all_scratch_doubles(const MlpDev *mlps_host, int nel, const int32_t *elem_start) {
  int stride = 0, layers = 0;
  size_t tiles = 0;
  for (int e = 0; e < nel; ++e) {
    stride = std::max(stride, mlp_stride(mlps_host[e]));
    layers = std::max(layers, mlps_host[e].n_layers);
    tiles += (size_t)(elem_start[e + 1] - elem_start[e] + kMlpRows - 1) / kMlpRows;
  }
  return tiles * layers * kMlpRows * stride;
}

void launch_mlp_all(const MlpDev *mlps_dev, const MlpDev *mlps_host, int nel, int activation, int ndim,
                    const DeviceBatch &b, double *scratch, hipStream_t s) {
  if (nel == 1) {  // the model description travels as a kernel argument: scalar loads
    launch_mlp_impl(mlps_host[0], activation, ndim, b.elem_atoms, b.elem_start[1] - b.elem_start[0], b,
                    scratch, s);
    return;
  }
  MlpTiles t;
  t.nel = nel;
  int stride = 0, layers = 0, blocks = 0, width = 0;
  for (int e = 0; e < nel; ++e) {
    width = std::max(width, mlps_host[e].max_np);
    stride = std::max(stride, mlp_stride(mlps_host[e]));
    layers = std::max(layers, mlps_host[e].n_layers);
    t.tile_start[e] = blocks;
    t.elem_start[e] = b.elem_start[e];
    blocks += (b.elem_start[e + 1] - b.elem_start[e] + kMlpRows - 1) / kMlpRows;
  }
  t.tile_start[nel] = blocks;
  t.elem_start[nel] = b.elem_start[nel];
  for (int e = nel + 1; e <= kMaxElements; ++e) t.tile_start[e] = t.elem_start[e] = 0;
  if (blocks == 0) return;
  size_t lds = 2 * (size_t)kMlpRows * stride * sizeof(double);
  const size_t lds_da = (size_t)layers * kMlpRows * stride * sizeof(double);
  if (lds + lds_da <= 64 * 1024 && !getenv("TA_MLP_DA_GLOBAL")) {  // act' slab in LDS, see mlp_kernel
    lds += lds_da;
    scratch = nullptr;
  }
  if (width >= 128)
    hipLaunchKernelGGL(mlp_all_kernel<512>, dim3((unsigned)blocks), dim3(512), lds, s, mlps_dev, t,
                       activation, ndim, b.elem_atoms, b.G, b.dEdG, b.eatom, scratch, stride,
                       layers * kMlpRows * stride);
  else
    hipLaunchKernelGGL(mlp_all_kernel<kMlpThreads>, dim3((unsigned)blocks), dim3(kMlpThreads), lds, s,
                       mlps_dev, t, activation, ndim, b.elem_atoms, b.G, b.dEdG, b.eatom, scratch,
                       stride, layers * kMlpRows * stride);
}

}  // namespace ta
