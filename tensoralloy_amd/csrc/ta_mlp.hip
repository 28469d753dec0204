// Stand-alone per-atom element-wise MLP kernel: 16 atoms of one element per
// workgroup, forward + backward-to-inputs on the fp64 matrix cores (see
// ta_mlp_tile.h for the tile code and the reference ops it replaces). Used when
// the fused per-centre kernel does not apply (radial-only models, several
// parameter chunks, more than 3 elements, very large neighbour counts).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>

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

// ---- one wavefront per 16 atoms, the whole network in registers ---------------------------------
// For the usual shapes (1 to 3 hidden layers of at most 64 units, scalar output, no skip
// connections) the GEMMs are run TRANSPOSED: Z^T[unit][atom] = W^T . X^T, with the weights as the A
// operand (lane (n', kq) reads W[4 kk + kq][16 nt + n'], 16 consecutive doubles) and the activations
// as the B operand (lane (m, kq) supplies X^T[4 kk + kq][atom m]). The f64 accumulator of tile nt
// then holds Z^T[16 nt + kq + 4 r][atom m] in register r -- which IS the B operand of k-step
// 4 nt + r of the next layer. So activations never leave the registers: no barrier and no
// transposition between the layers, forward or backward (delta^T = W . dz^T uses the transposed
// weight copy as A operand and the dz registers as B). The first layer reads the descriptors
// straight from G (k-steps beyond the true D are skipped, not padded to 16); the scalar output
// layer is a per-lane dot product and two cross-row adds instead of a 16-column tile.
// The weights (both orientations) are staged ONCE per workgroup in LDS, row stride = 16 mod 32
// doubles (the four kq rows of an A-operand read fall in disjoint banks); the 8 wavefronts of a
// workgroup then walk 16-atom tiles independently. A wavefront that fetched its A operands from
// L2 instead paid a round trip per 16 MFMAs: 49 us for one 4000-atom frame against 18 us for the
// 16-row tile kernel it replaces.
constexpr int kWaveNT = 4;       // hidden widths up to 64
constexpr int kWaveThreads = 512;
constexpr int kWaveMaxHidden = 3;

// LDS row strides. Everything up to 64 columns wide uses ONE compile-time stride, 80 doubles (= 16
// mod 32), so that the A-operand reads carry immediate offsets: with run-time strides the compiler
// hoists every read address out of the tile loop and spills (95 VGPRs for two hidden layers). Only
// the transposed first layer, whose rows are as long as the padded descriptor, has its own stride.
constexpr int kWaveStride = 80;
__host__ __device__ inline int wave_stride(int x) { return (x % 32 == 0) ? x + 16 : x; }

struct WaveLds {
  const double *w[kWaveMaxHidden], *wt[kWaveMaxHidden], *b[kWaveMaxHidden];
  int swt0;  // row stride of wt[0]
  const double *wout;
  double bout;
};

// doubles of LDS the staged network needs
__host__ __device__ inline size_t wave_lds_doubles(const MlpDev &mlp, int lh) {
  size_t n = 0;
  for (int l = 0; l < lh; ++l) {
    const MlpLayerDev &ly = mlp.layer[l];
    n += (size_t)ly.kp * kWaveStride + (size_t)ly.np * (l == 0 ? wave_stride(ly.kp) : kWaveStride) + ly.np;
  }
  return n + mlp.layer[lh].kp;
}

// dst[row * stride + col] = src[row * cols + col]: eight independent loads in flight per thread (a
// plain strided loop waits for every load before it issues the next one: 16 round trips to L2 for
// one 64 x 64 layer, 11 us)
__device__ __forceinline__ void wave_copy_rows(double *dst, int stride, const double *__restrict__ src, int rows,
                                               int cols) {
  const int n = rows * cols;
  for (int base = 0; base < n; base += 8 * kWaveThreads) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kWaveThreads + (int)threadIdx.x;
      v[u] = idx < n ? src[idx] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kWaveThreads + (int)threadIdx.x;
      if (idx < n) {
        const int row = idx / cols;
        dst[row * stride + (idx - row * cols)] = v[u];
      }
    }
  }
}

template <int LH>
__device__ __forceinline__ void wave_stage(const MlpDev &mlp, double *lds, WaveLds &L) {
  double *p = lds;
#pragma unroll
  for (int l = 0; l < LH; ++l) {
    const MlpLayerDev &ly = mlp.layer[l];
    const int sw = kWaveStride, swt = l == 0 ? wave_stride(ly.kp) : kWaveStride;
    double *w = p, *wt = w + (size_t)ly.kp * sw, *bb = wt + (size_t)ly.np * swt;
    wave_copy_rows(w, sw, ly.w, ly.kp, ly.np);
    wave_copy_rows(wt, swt, ly.wt, ly.np, ly.kp);
    wave_copy_rows(bb, ly.np, ly.b, 1, ly.np);
    L.w[l] = w;
    L.wt[l] = wt;
    L.b[l] = bb;
    if (l == 0) L.swt0 = swt;
    p = bb + ly.np;
  }
  const MlpLayerDev &lo = mlp.layer[LH];
  for (int idx = threadIdx.x; idx < lo.kp; idx += kWaveThreads) p[idx] = lo.w[(size_t)idx * lo.np];  // column 0
  L.wout = p;
  L.bout = lo.b[0];
}

// activation of one accumulator tile. NOT inlined: the one-wavefront kernel applies it to 8 tiles per
// hidden layer, and 32 inlined copies of the activation switch per layer made the kernel 112 KB of
// code -- more than the instruction cache, which a lone wavefront per SIMD then misses all the way
// (42 us for one 4000-atom frame; this call brings the code to a fifth of that).
struct WaveAct {
  mlp_f64x4 h, d;
};
__device__ __attribute__((noinline)) WaveAct wave_activate(int act, mlp_f64x4 z) {  // by value: registers
  WaveAct o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double hv, dv;
    activation_fn(act, z[r], hv, dv);
    o.h[r] = hv;
    o.d[r] = dv;
  }
  return o;
}

template <int LH>
__device__ __forceinline__ void mlp_wave_tile(const MlpDev &mlp, const WaveLds &L, int act, int ndim,
                                              const int32_t *atoms, int n_atoms, int a0,
                                              const double *__restrict__ G, double *__restrict__ dEdG,
                                              double *__restrict__ eatom) {
  const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
  const bool valid = a0 + m < n_atoms;
  const int atom = atoms[valid ? a0 + m : a0];
  mlp_f64x4 h[LH][kWaveNT], dh[LH][kWaveNT];
  // ---- forward ----
  {
    const MlpLayerDev &ly = mlp.layer[0];
    const int nt0 = ly.np / 16;
    constexpr int sw = kWaveStride;
#pragma unroll
    for (int nt = 0; nt < kWaveNT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[0][nt][r] = nt < nt0 ? L.b[0][16 * nt + kq + 4 * r] : 0.0;
    const double *g = G + (size_t)atom * ndim;
    for (int k0 = 0; k0 < ndim; k0 += 4) {
      const int k = k0 + kq;
      double x = 0.0;
      if (k < ndim) {
        x = g[k];
        if (mlp.xlo) {
          const double den = mlp.xhi[k] - mlp.xlo[k];
          x = (den != 0.0) ? (mlp.xhi[k] - x) / den : 0.0;  // div_no_nan, atomic.py:195
        }
      }
      const double *wrow = L.w[0] + k * sw + m;  // k < kp: rows beyond ndim are zero padding
#pragma unroll
      for (int nt = 0; nt < kWaveNT; ++nt)
        if (nt < nt0) h[0][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wrow[16 * nt], x, h[0][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < kWaveNT; ++nt)
      if (nt < nt0) {
        const WaveAct o = wave_activate(act, h[0][nt]);
        h[0][nt] = o.h;
        dh[0][nt] = o.d;
      }
  }
#pragma unroll
  for (int l = 1; l < LH; ++l) {
    const MlpLayerDev &ly = mlp.layer[l];
    const int ntl = ly.np / 16, ntp = ly.kp / 16;
    constexpr int sw = kWaveStride;
    // the four output tiles advance together: four independent accumulator chains per k-step
    mlp_f64x4 acc[kWaveNT];
#pragma unroll
    for (int nt = 0; nt < kWaveNT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nt][r] = nt < ntl ? L.b[l][16 * nt + kq + 4 * r] : 0.0;
    const double *wcol = L.w[l] + kq * sw + m;
#pragma unroll
    for (int t = 0; t < kWaveNT; ++t) {
      if (t >= ntp) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double a[kWaveNT];
#pragma unroll
        for (int nt = 0; nt < kWaveNT; ++nt) a[nt] = nt < ntl ? wcol[(16 * t + 4 * r) * sw + 16 * nt] : 0.0;
#pragma unroll
        for (int nt = 0; nt < kWaveNT; ++nt)
          if (nt < ntl) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[nt], h[l - 1][t][r], acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < kWaveNT; ++nt) {
      if (nt >= ntl) continue;
      const WaveAct o = wave_activate(act, acc[nt]);
      h[l][nt] = o.h;
      dh[l][nt] = o.d;
    }
  }
  // scalar output layer: y = b + sum_k h[k] w[k][0]; its backward seed dE/dh[k] = w[k][0]
  {
    const int ntp = mlp.layer[LH].kp / 16;
    double y = 0.0;
#pragma unroll
    for (int t = 0; t < kWaveNT; ++t)
      if (t < ntp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double wo = L.wout[16 * t + kq + 4 * r];
          y = fma(h[LH - 1][t][r], wo, y);
          dh[LH - 1][t][r] *= wo;  // dz of the last hidden layer
        }
      }
    y += __shfl_xor(y, 16);
    y += __shfl_xor(y, 32);
    if (kq == 0 && valid) eatom[atom] = y + L.bout;
  }
  // ---- backward: delta_{l-1}^T = W_l . dz_l^T, dz_{l-1} = delta_{l-1} * act'(z_{l-1}) ----
#pragma unroll
  for (int l = LH - 1; l >= 1; --l) {
    const MlpLayerDev &ly = mlp.layer[l];
    const int ntl = ly.np / 16, ntp = ly.kp / 16;
    constexpr int swt = kWaveStride;
    mlp_f64x4 acc[kWaveNT];
#pragma unroll
    for (int t = 0; t < kWaveNT; ++t) acc[t] = {0.0, 0.0, 0.0, 0.0};
    const double *wcol = L.wt[l] + kq * swt + m;
#pragma unroll
    for (int nt = 0; nt < kWaveNT; ++nt) {
      if (nt >= ntl) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double a[kWaveNT];
#pragma unroll
        for (int t = 0; t < kWaveNT; ++t) a[t] = t < ntp ? wcol[(16 * nt + 4 * r) * swt + 16 * t] : 0.0;
#pragma unroll
        for (int t = 0; t < kWaveNT; ++t)
          if (t < ntp) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], dh[l][nt][r], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < kWaveNT; ++t)
      if (t < ntp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[l - 1][t][r] *= acc[t][r];
      }
  }
  {
    const MlpLayerDev &ly = mlp.layer[0];
    const int nt0 = ly.np / 16, swt = L.swt0;
    for (int j0 = 0; j0 < ndim; j0 += 16) {
      mlp_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      // wt rows: hidden unit n, columns: input channel (kp columns, zero beyond ndim)
      const double *wcol = L.wt[0] + kq * swt + j0 + m;
#pragma unroll
      for (int nt = 0; nt < kWaveNT; ++nt) {
        if (nt >= nt0) continue;
        double a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = wcol[(16 * nt + 4 * r) * swt];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], dh[0][nt][r], acc, 0, 0, 0);
      }
      // acc[r] = dE/dG[atom m][j0 + kq + 4 r]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + kq + 4 * r;
        if (valid && j < ndim) {
          double d = acc[r];
          if (mlp.xlo) {
            const double den = mlp.xhi[j] - mlp.xlo[j];
            d = (den != 0.0) ? -d / den : 0.0;
          }
          dEdG[(size_t)atom * ndim + j] = d;
        }
      }
    }
  }
}

template <int LH>
__global__ __launch_bounds__(kWaveThreads) void mlp_wave_kernel(MlpDev mlp, int act, int ndim, const int32_t *atoms,
                                                                int n_atoms, const double *G, double *dEdG,
                                                                double *eatom) {
  extern __shared__ double lds[];
  WaveLds L;
  wave_stage<LH>(mlp, lds, L);
  __syncthreads();
  const int ntiles = (n_atoms + kMlpRows - 1) / kMlpRows, nw = kWaveThreads / 64;
  for (int tile = (int)blockIdx.x * nw + (int)(threadIdx.x >> 6); tile < ntiles; tile += (int)gridDim.x * nw)
    mlp_wave_tile<LH>(mlp, L, act, ndim, atoms, n_atoms, tile * kMlpRows, G, dEdG, eatom);
}

// all elements in one launch: blockIdx.y = element
template <int LH>
__global__ __launch_bounds__(kWaveThreads) void mlp_wave_all_kernel(const MlpDev *__restrict__ mlps, MlpTiles tiles,
                                                                    int act, int ndim, const int32_t *atoms,
                                                                    const double *G, double *dEdG, double *eatom) {
  extern __shared__ double lds[];
  const int e = blockIdx.y;
  const int n_atoms = tiles.elem_start[e + 1] - tiles.elem_start[e];
  const int ntiles = (n_atoms + kMlpRows - 1) / kMlpRows, nw = kWaveThreads / 64;
  if ((int)blockIdx.x * nw >= ntiles) return;  // nothing for this workgroup (uniform)
  const MlpDev &mlp = mlps[e];
  WaveLds L;
  wave_stage<LH>(mlp, lds, L);
  __syncthreads();
  for (int tile = (int)blockIdx.x * nw + (int)(threadIdx.x >> 6); tile < ntiles; tile += (int)gridDim.x * nw)
    mlp_wave_tile<LH>(mlp, L, act, ndim, atoms + tiles.elem_start[e], n_atoms, tile * kMlpRows, G, dEdG, eatom);
}

}  // namespace

// scratch doubles needed per 16-row tile
size_t mlp_scratch_doubles(const MlpDev &mlp) {
  return (size_t)mlp.n_layers * kMlpRows * mlp_stride(mlp);
}

constexpr size_t kWaveLdsLimit = 150 * 1024;

// hidden layers when the four-wavefront latency kernel applies (same shapes as the one-wavefront
// kernel, launches BELOW its tile threshold), else 0
int mlp_quad_shape(const MlpDev &mlp, int n_tiles);
int mlp_quad_tiles(const MlpDev &mlp);

// ---- four wavefronts per 16 atoms, transposed GEMMs, one barrier per layer ------------------------
// The LATENCY kernel for launches of few tiles (one frame: 250 tiles on 256 CUs). Same transposed
// formulation as the one-wavefront kernel, but the four 16-unit output tiles of a layer belong to the
// four wavefronts of the workgroup, so a tile's MFMAs (64 cycles each on gfx950) and activations run
// on the four SIMDs of a CU side by side, as in the generic tile kernel -- and because a wavefront's
// accumulator tile Z^T[16 nt + kq + 4 r][atom m] is stored to LDS exactly where the next layer's B
// operand is read ([unit][atom], 512 contiguous bytes per k-step), a layer costs ONE barrier instead
// of the generic kernel's elementwise passes (14 barriers for two hidden layers there, 4 here).
// Activation derivatives of a wavefront's own tile stay in its registers for the backward sweep.
// Weights are A operands read from global memory (L2 hits), first layer from G directly, scalar
// output layer on the VALU. Shapes as for the one-wavefront kernel; one model per launch.
// NT = 16-unit tiles per hidden layer = wavefronts per workgroup: 4 (hidden widths up to 64) or 8 (up to
// 128, the 2 x 128 networks of the Ni-Mo benchmark, whose generic tile kernel runs at a quarter of the
// matrix rate: 30 us for 245 tiles).
template <int NT>
struct QuadLds {
  double h[2][16 * NT * kMlpRows];   // activations / dz, ping-pong: [unit][atom]
  double ypart[NT][kMlpRows];
};

// A operands of one GEMM phase of a wavefront's tile (at most 4 NT k-steps), fetched BEFORE the barrier
// that precedes the phase: the L2 round trip overlaps the previous phase's tail and the barrier wait
template <int NT>
__device__ __forceinline__ void quad_fetch(double (&w)[4 * NT], const double *wcol, size_t stride, int n_tiles) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) w[4 * t + r] = t < n_tiles ? wcol[(size_t)(16 * t + 4 * r) * stride] : 0.0;
}
template <int NT>
__device__ __forceinline__ mlp_f64x4 quad_gemm(const double (&w)[4 * NT], const double *bin, int n_tiles,
                                               mlp_f64x4 acc) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < n_tiles) {
      double bq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bq[r] = bin[(16 * t + 4 * r) * kMlpRows];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[4 * t + r], bq[r], acc, 0, 0, 0);
    }
  return acc;
}

template <int LH, int NT>
__device__ __forceinline__ void mlp_quad_body(const MlpDev &mlp, QuadLds<NT> &L, int act, int ndim,
                                              const int32_t *atoms, int n_atoms, int a0,
                                              const double *__restrict__ G, double *__restrict__ dEdG,
                                              double *__restrict__ eatom) {
  const int lane = threadIdx.x & 63, nt = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
  const bool valid = a0 + m < n_atoms;
  const int atom = atoms[valid ? a0 + m : a0];
  mlp_f64x4 dh[LH];  // act'(z) of this wavefront's tile in every hidden layer
  int cur = 0;
  // ---- forward ----
  {
    const MlpLayerDev &ly = mlp.layer[0];
    if (nt < ly.np / 16) {
      mlp_f64x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = ly.b[16 * nt + kq + 4 * r];
      const double *g = G + (size_t)atom * ndim;
      for (int k0 = 0; k0 < ndim; k0 += 4) {
        const int k = k0 + kq;
        double x = 0.0;
        if (k < ndim) {
          x = g[k];
          if (mlp.xlo) {
            const double den = mlp.xhi[k] - mlp.xlo[k];
            x = (den != 0.0) ? (mlp.xhi[k] - x) / den : 0.0;  // div_no_nan, atomic.py:195
          }
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ly.w[(size_t)k * ly.np + 16 * nt + m], x, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double hv, dv;
        activation_fn(act, acc[r], hv, dv);
        dh[0][r] = dv;
        L.h[cur][(16 * nt + kq + 4 * r) * kMlpRows + m] = hv;
      }
    }
  }
  double wnext[4 * NT];  // A operands of the next GEMM phase, in flight across the barrier
  mlp_f64x4 bnext = {0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  if (LH > 1) {
    const MlpLayerDev &nx = mlp.layer[1];
    const int tile = nt < nx.np / 16 ? nt : 0;
    quad_fetch<NT>(wnext, nx.w + (size_t)kq * nx.np + 16 * tile + m, nx.np, nx.kp / 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) bnext[r] = nx.b[16 * tile + kq + 4 * r];
  } else {
    const MlpLayerDev &l0 = mlp.layer[0];
    quad_fetch<NT>(wnext, l0.wt + (size_t)kq * l0.kp + (16 * nt < ndim ? 16 * nt : 0) + m, l0.kp, l0.np / 16);
  }
#pragma unroll
  for (int l = 1; l < LH; ++l) {
    const MlpLayerDev &ly = mlp.layer[l];
    mlp_f64x4 acc = bnext;
    if (nt < ly.np / 16) acc = quad_gemm<NT>(wnext, L.h[cur] + kq * kMlpRows + m, ly.kp / 16, acc);
    if (nt < ly.np / 16) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double hv, dv;
        activation_fn(act, acc[r], hv, dv);
        dh[l][r] = dv;
        L.h[cur ^ 1][(16 * nt + kq + 4 * r) * kMlpRows + m] = hv;
      }
    }
    cur ^= 1;
    __syncthreads();
    // weights of the phase after this one: the next forward layer, or the first backward GEMM
    if (l + 1 < LH) {
      const MlpLayerDev &nx = mlp.layer[l + 1];
      const int tile = nt < nx.np / 16 ? nt : 0;
      quad_fetch<NT>(wnext, nx.w + (size_t)kq * nx.np + 16 * tile + m, nx.np, nx.kp / 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) bnext[r] = nx.b[16 * tile + kq + 4 * r];
    } else {
      const int tile = nt < ly.kp / 16 ? nt : 0;
      quad_fetch<NT>(wnext, ly.wt + (size_t)kq * ly.kp + 16 * tile + m, ly.kp, ly.np / 16);
    }
  }
  // scalar output layer: y = b + sum_k h[k] w[k][0]; dz of the last hidden layer = act' * w[k][0]
  {
    const MlpLayerDev &lo = mlp.layer[LH];
    double y = 0.0;
    if (nt < lo.kp / 16) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * nt + kq + 4 * r;
        const double wo = lo.w[(size_t)k * lo.np];
        y = fma(L.h[cur][k * kMlpRows + m], wo, y);
        L.h[cur ^ 1][k * kMlpRows + m] = dh[LH - 1][r] * wo;
      }
    }
    y += __shfl_xor(y, 16);
    y += __shfl_xor(y, 32);
    if (kq == 0) L.ypart[nt][m] = y;
    cur ^= 1;  // h[cur] now holds dz of the last hidden layer
    __syncthreads();
    if (nt == 0 && kq == 0 && valid) {
      double y4[NT / 4];  // fixed order: pairs of pairs, then the quartets in turn
#pragma unroll
      for (int q = 0; q < NT / 4; ++q)
        y4[q] = (L.ypart[4 * q][m] + L.ypart[4 * q + 1][m]) + (L.ypart[4 * q + 2][m] + L.ypart[4 * q + 3][m]);
      double ys = y4[0];
#pragma unroll
      for (int q = 1; q < NT / 4; ++q) ys += y4[q];
      eatom[atom] = ys + lo.b[0];
    }
  }
  // ---- backward: delta_{l-1}^T = W_l . dz_l^T, dz_{l-1} = delta_{l-1} * act'(z_{l-1}) ----
#pragma unroll
  for (int l = LH - 1; l >= 1; --l) {
    const MlpLayerDev &ly = mlp.layer[l];
    mlp_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    if (nt < ly.kp / 16) acc = quad_gemm<NT>(wnext, L.h[cur] + kq * kMlpRows + m, ly.np / 16, acc);
    if (nt < ly.kp / 16) {
#pragma unroll
      for (int r = 0; r < 4; ++r) L.h[cur ^ 1][(16 * nt + kq + 4 * r) * kMlpRows + m] = acc[r] * dh[l - 1][r];
    }
    cur ^= 1;
    __syncthreads();
    {  // weights of the next backward GEMM (layer l - 1, or the final dE/dG tiles of layer 0)
      const MlpLayerDev &nx = mlp.layer[l - 1];
      const int col = l > 1 ? (nt < nx.kp / 16 ? 16 * nt : 0) : (16 * nt < ndim ? 16 * nt : 0);
      quad_fetch<NT>(wnext, nx.wt + (size_t)kq * nx.kp + col + m, nx.kp, nx.np / 16);
    }
  }
  {
    const MlpLayerDev &ly = mlp.layer[0];
    // dE/dG: 16 input channels per tile, tiles dealt to the wavefronts
    for (int j0 = 16 * nt; j0 < ndim; j0 += 16 * NT) {
      mlp_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      // (the first tile's operands were fetched before the barrier; further tiles, D > 16 NT, stream)
      if (j0 != 16 * nt) quad_fetch<NT>(wnext, ly.wt + (size_t)kq * ly.kp + j0 + m, ly.kp, ly.np / 16);
      acc = quad_gemm<NT>(wnext, L.h[cur] + kq * kMlpRows + m, ly.np / 16, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + kq + 4 * r;
        if (valid && j < ndim) {
          double d = acc[r];
          if (mlp.xlo) {
            const double den = mlp.xhi[j] - mlp.xlo[j];
            d = (den != 0.0) ? -d / den : 0.0;
          }
          dEdG[(size_t)atom * ndim + j] = d;
        }
      }
    }
  }
}


template <int LH, int NT>
__global__ __launch_bounds__(64 * NT) void mlp_quad_kernel(MlpDev mlp, int act, int ndim, const int32_t *atoms,
                                                           int n_atoms, const double *__restrict__ G,
                                                           double *__restrict__ dEdG, double *__restrict__ eatom) {
  __shared__ QuadLds<NT> L;
  mlp_quad_body<LH, NT>(mlp, L, act, ndim, atoms, n_atoms, (int)blockIdx.x * kMlpRows, G, dEdG, eatom);
}

// every element of an alloy in ONE launch: blocks laid out element after element (as mlp_all_kernel)
template <int LH, int NT>
__global__ __launch_bounds__(64 * NT) void mlp_quad_all_kernel(const MlpDev *__restrict__ mlps, MlpTiles tiles,
                                                               int act, int ndim, const int32_t *atoms,
                                                               const double *__restrict__ G,
                                                               double *__restrict__ dEdG,
                                                               double *__restrict__ eatom) {
  __shared__ QuadLds<NT> L;
  int e = 0;
  while (e + 1 < tiles.nel && (int)blockIdx.x >= tiles.tile_start[e + 1]) ++e;
  mlp_quad_body<LH, NT>(mlps[e], L, act, ndim, atoms + tiles.elem_start[e],
                        tiles.elem_start[e + 1] - tiles.elem_start[e],
                        ((int)blockIdx.x - tiles.tile_start[e]) * kMlpRows, G, dEdG, eatom);
}

// The one-wavefront kernel is the THROUGHPUT kernel: 4000 tiles (16 frames of 4000 atoms) take 66 us
// against 98 us with the 16-row tile kernel. For one frame it loses (35 us against 18 us): 250 tiles
// are fewer than the 1024 SIMDs, and a tile's 152 MFMAs (64 cycles each on gfx950) and 32 activations
// per lane are serial in one wavefront where the tile kernel spreads them over the four SIMDs of a CU.
constexpr int kWaveMinTiles = 1024;

// number of hidden layers when the one-wavefront kernel applies to `n_tiles` tiles, else 0
int mlp_wave_shape(const MlpDev &mlp, int n_tiles) {
  const int lh = mlp.n_layers - 1;
  if (lh < 1 || lh > kWaveMaxHidden || getenv("TA_MLP_TILE_KERNEL")) return 0;
  if (n_tiles < kWaveMinTiles && !getenv("TA_MLP_WAVE_KERNEL")) return 0;
  for (int l = 0; l < lh; ++l) {
    const MlpLayerDev &ly = mlp.layer[l];
    if (ly.np > 16 * kWaveNT || ly.res || !ly.act) return 0;
    if (l > 0 && ly.kp != mlp.layer[l - 1].np) return 0;
  }
  const MlpLayerDev &lo = mlp.layer[lh];
  if (lo.n != 1 || lo.act || lo.res || lo.kp != mlp.layer[lh - 1].np) return 0;
  if (wave_lds_doubles(mlp, lh) * sizeof(double) > kWaveLdsLimit) return 0;  // wide descriptors
  return lh;
}

// Below one tile per CU the generic tile kernel is still the faster one in the angular pipeline
// (4000-atom frame, 250 tiles: 18.2-18.9 us against 19.7-20.3 us for this kernel, same session),
// although this kernel wins the isolated comparison on G2-only models (13.8 against 17.7 us); from
// three frames on it is ahead in both (7.1 against 7.7 us per frame). Hidden layers wider than 64
// (NT = 8) take this kernel from the first tile: there the generic kernel is the slow one.
constexpr int kQuadMinTiles = 257;
constexpr int kQuadMaxNT = 8;

// tiles per hidden layer this kernel would run with (4 or 8), 0 = shape not covered
int mlp_quad_tiles(const MlpDev &mlp) {
  const int lh = mlp.n_layers - 1;
  if (lh < 1 || lh > kWaveMaxHidden) return 0;
  int widest = 0;
  for (int l = 0; l < lh; ++l) {
    const MlpLayerDev &ly = mlp.layer[l];
    if (ly.np > 16 * kQuadMaxNT || ly.res || !ly.act) return 0;
    if (l > 0 && ly.kp != mlp.layer[l - 1].np) return 0;
    widest = std::max(widest, ly.np);
  }
  const MlpLayerDev &lo = mlp.layer[lh];
  if (lo.n != 1 || lo.act || lo.res || lo.kp != mlp.layer[lh - 1].np) return 0;
  return widest <= 16 * kWaveNT ? kWaveNT : kQuadMaxNT;
}

// number of hidden layers when the four / eight-wavefront kernel applies to `n_tiles` tiles, else 0
int mlp_quad_shape(const MlpDev &mlp, int n_tiles) {
  if (getenv("TA_MLP_TILE_KERNEL") || getenv("TA_MLP_WAVE_KERNEL")) return 0;
  const int nt = mlp_quad_tiles(mlp);
  if (!nt) return 0;
  if (getenv("TA_MLP_QUAD_KERNEL")) return mlp.n_layers - 1;
  if (nt == kWaveNT && (n_tiles >= kWaveMinTiles || n_tiles < kQuadMinTiles)) return 0;
  if (nt == kQuadMaxNT && n_tiles >= 4 * kWaveMinTiles) return 0;  // (no one-wavefront kernel at this width)
  return mlp.n_layers - 1;
}

// more than 64 KB of dynamic LDS needs the attribute; set once per kernel and device (the call is
// host-side work of tens of microseconds -- per launch it would stall the queue behind the host)
template <typename K>
void allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return;
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> allowed;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const auto key = std::make_pair(reinterpret_cast<const void *>(kernel), dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = allowed.find(key);
  if (it != allowed.end() && it->second >= bytes) return;
  if (hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWaveLdsLimit) != hipSuccess)
    throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  allowed[key] = kWaveLdsLimit;
}

void launch_mlp_impl(const MlpDev &mlp, int activation, int ndim, const int32_t *atoms, int n_atoms,
                     const DeviceBatch &b, double *scratch, hipStream_t s) {
  if (n_atoms == 0) return;
  if (const int lh = mlp_quad_shape(mlp, (n_atoms + kMlpRows - 1) / kMlpRows)) {
    const unsigned qblocks = (unsigned)((n_atoms + kMlpRows - 1) / kMlpRows);
    const int nt = mlp_quad_tiles(mlp);
#define TA_QUAD(LH, NT)                                                                                       \
  hipLaunchKernelGGL((mlp_quad_kernel<LH, NT>), dim3(qblocks), dim3(64 * NT), 0, s, mlp, activation, ndim, atoms, \
                     n_atoms, b.G, b.dEdG, b.eatom)
    if (nt == kWaveNT) {
      if (lh == 1) TA_QUAD(1, 4);
      else if (lh == 2) TA_QUAD(2, 4);
      else TA_QUAD(3, 4);
    } else {
      if (lh == 1) TA_QUAD(1, 8);
      else if (lh == 2) TA_QUAD(2, 8);
      else TA_QUAD(3, 8);
    }
#undef TA_QUAD
    return;
  }
  if (const int lh = mlp_wave_shape(mlp, (n_atoms + kMlpRows - 1) / kMlpRows)) {
    const int ntiles = (n_atoms + kMlpRows - 1) / kMlpRows, nw = kWaveThreads / 64;
    const unsigned wblocks = (unsigned)std::min((ntiles + nw - 1) / nw, 256);
    const size_t lds = wave_lds_doubles(mlp, lh) * sizeof(double);
    auto go = [&](auto kernel) {
      allow_lds(kernel, lds);
      hipLaunchKernelGGL(kernel, dim3(wblocks), dim3(kWaveThreads), lds, s, mlp, activation, ndim, atoms, n_atoms,
                         b.G, b.dEdG, b.eatom);
    };
    if (lh == 1) go(mlp_wave_kernel<1>);
    else if (lh == 2) go(mlp_wave_kernel<2>);
    else go(mlp_wave_kernel<3>);
    return;
  }
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
  // every element's network fits the four / eight-wavefront kernel (same depth and tile count)
  {
    int qlh = mlp_quad_shape(mlps_host[0], blocks), qnt = mlp_quad_tiles(mlps_host[0]);
    for (int e = 1; e < nel; ++e)
      if (mlp_quad_shape(mlps_host[e], blocks) != qlh || mlp_quad_tiles(mlps_host[e]) != qnt) qlh = 0;
    if (qlh) {
#define TA_QUAD_ALL(LH, NT)                                                                                \
  hipLaunchKernelGGL((mlp_quad_all_kernel<LH, NT>), dim3((unsigned)blocks), dim3(64 * NT), 0, s, mlps_dev, t, \
                     activation, ndim, b.elem_atoms, b.G, b.dEdG, b.eatom)
      if (qnt == kWaveNT) {
        if (qlh == 1) TA_QUAD_ALL(1, 4);
        else if (qlh == 2) TA_QUAD_ALL(2, 4);
        else TA_QUAD_ALL(3, 4);
      } else {
        if (qlh == 1) TA_QUAD_ALL(1, 8);
        else if (qlh == 2) TA_QUAD_ALL(2, 8);
        else TA_QUAD_ALL(3, 8);
      }
#undef TA_QUAD_ALL
      return;
    }
  }
  // every element's network has the same one-wavefront shape: one grid row per element
  int lh = mlp_wave_shape(mlps_host[0], blocks);
  size_t wlds = 0;
  int max_tiles = 0;
  for (int e = 0; e < nel; ++e) {
    if (mlp_wave_shape(mlps_host[e], blocks) != lh) lh = 0;
    if (lh) wlds = std::max(wlds, wave_lds_doubles(mlps_host[e], lh) * sizeof(double));
    max_tiles = std::max(max_tiles, (b.elem_start[e + 1] - b.elem_start[e] + kMlpRows - 1) / kMlpRows);
  }
  if (lh) {
    const int nw = kWaveThreads / 64;
    const unsigned wblocks = (unsigned)std::min((max_tiles + nw - 1) / nw, std::max(256 / nel, 1));
    auto go = [&](auto kernel) {
      allow_lds(kernel, wlds);
      hipLaunchKernelGGL(kernel, dim3(wblocks, (unsigned)nel), dim3(kWaveThreads), wlds, s, mlps_dev, t, activation,
                         ndim, b.elem_atoms, b.G, b.dEdG, b.eatom);
    };
    if (lh == 1) go(mlp_wave_all_kernel<1>);
    else if (lh == 2) go(mlp_wave_all_kernel<2>);
    else go(mlp_wave_all_kernel<3>);
    return;
  }
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
