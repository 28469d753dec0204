// C ABI of libtensoralloy_amd.so: context, device memory, launch sequencing.
// Declarations and the reference interfaces they replace: include/tensoralloy_amd.h
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ta_device.h"
#include "ta_internal.h"

namespace ta {
size_t mlp_scratch_doubles(const MlpDev &mlp);
void launch_mlp_impl(const MlpDev &mlp, int activation, int ndim, const int32_t *atoms, int n_atoms,
                     const DeviceBatch &b, double *scratch, hipStream_t s);
size_t mlp_all_scratch_doubles(const MlpDev *mlps_host, int nel, const int32_t *elem_start);
void launch_mlp_all(const MlpDev *mlps_dev, const MlpDev *mlps_host, int nel, int activation, int ndim,
                    const DeviceBatch &b, double *scratch, hipStream_t s);
// weight gradients (ta_train.hip)
int mlp_param_count(const MlpDev &mlp);
size_t mlp_grad_scratch_doubles(const MlpDev &mlp, int n_atoms);
size_t mlp_grad_partial_doubles(const MlpDev &mlp, int n_atoms);
void launch_mlp_grad(const MlpDev &mlp, int activation, int ndim, const int32_t *atoms, int n_atoms,
                     const DeviceBatch &b, const double *frame_coeff, double *scratch, double *partial,
                     double *grad, hipStream_t s);
size_t mlp_grad2_scratch_doubles(const MlpDev &mlp, int n_atoms);
void launch_mlp_grad2(const MlpDev &mlp, int activation, int ndim, const int32_t *atoms, int n_atoms,
                      const DeviceBatch &b, const double *dG, const double *frame_coeff, double *scratch,
                      double *partial, double *grad, hipStream_t s, double *kappa_out = nullptr);
// analytic Hessian-vector products of the descriptor models (ta_hvp.hip)
void launch_pair_vec(const DeviceBatch &b, double *Dv, hipStream_t s);
void launch_backward_hvp(const SFParams &sf, const AngChunk &ch, int nb, int ng, int nz, bool first, bool angular,
                         const DeviceBatch &b, const double *Dv, const double *Dd, const double *wdot, double *gv,
                         double *gd, hipStream_t s);
void launch_hvp_gather(const DeviceBatch &b, const double *Dv, const double *Dd, const double *gv, const double *gd,
                       double *fdot, double *wdot_at, hipStream_t s);
void launch_pair_tangent(const DeviceBatch &b, const double *dR, const double *dh, double *dD, hipStream_t s);
void launch_descriptor_jvp(const DeviceBatch &b, int ndim, const double *J, const double *dD, double *dG,
                           hipStream_t s);
void launch_one_hot(double *dEdG, int64_t n_atoms, int ndim, int c, hipStream_t s);
// GRAP (ta_grap.hip)
struct GrapModel;
bool grap_hvp_supported(const GrapModel *);
void launch_grap_hvp(GrapModel *, const DeviceBatch &b, double eps, const double *Dv, const double *Dd,
                     const double *wdot, double *gv, double *gd, hipStream_t s);
GrapModel *grap_create(const ta_model_desc *m, std::string &err);
int grap_ndim(const GrapModel *g);
bool grap_uses_filter_net(const GrapModel *g);
void grap_destroy(GrapModel *);
void grap_ensure(GrapModel *, const DeviceBatch &b);
void launch_grap_forward(GrapModel *, const DeviceBatch &b, double eps, hipStream_t s);
void launch_grap_backward(GrapModel *, const DeviceBatch &b, hipStream_t s);
// EAM / ADP (ta_eam.hip)
struct EamModel;
EamModel *eam_create(const ta_model_desc *m, std::string &err);
void eam_destroy(EamModel *);
void eam_ensure(EamModel *, const DeviceBatch &b);
void eam_tabulate(EamModel *m, int n_r, const double *r, int n_rho, const double *rho, double *rho_of_r,
                  double *phi_of_r, double *embed_of_rho, double *u_of_r, double *w_of_r,
                  hipStream_t s);
void eam_compute(EamModel *, const DeviceBatch &b, uint32_t want, hipStream_t s,
                 hipEvent_t *ev /* 2 events or null */);
void eam_set_list_cutoff(EamModel *, double rc /* 0: the list is exact, no test */);
bool eam_is_plain(const EamModel *);
void eam_set_nn_tables(EamModel *, bool on);
bool eam_hvp_supported(const EamModel *);
size_t eam_hvp_extra_doubles(const EamModel *, const DeviceBatch &b, int n_dir);
void eam_hvp(EamModel *, const DeviceBatch &b, int n_dir, bool unit, int first, const double *dR, const double *dh,
             double *dFdot, double *fdot, double *wdot, double *extra, hipStream_t s);
bool eam_nn_tables_on(const EamModel *);
void eam_mark_trained(EamModel *);
int64_t eam_param_count(const EamModel *);
void eam_update_weights(EamModel *, const double *flat, int64_t n);
int64_t eam_constant_count(const EamModel *);
void eam_get_constants(const EamModel *, double *flat);
void eam_update_constants(EamModel *, const double *flat, int64_t n);
void eam_constant_gradient(EamModel *, const DeviceBatch &b, const double *frame_coeff, const double *dR,
                           const double *dh, double *grad, hipStream_t s);
void eam_energy_gradient(EamModel *, const DeviceBatch &b, const double *frame_coeff, double *grad,
                         hipStream_t s);
bool eam_loss_gradient_supported(const EamModel *);
void eam_loss_gradient(EamModel *, const DeviceBatch &b, const double *frame_coeff, const double *dR,
                       const double *dh, double *grad, hipStream_t s);
}  // namespace ta

namespace {

thread_local std::string g_create_error;

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                     std::to_string(__LINE__) + ")");                                        \
  } while (0)

// grow-only device buffer
template <typename T>
struct DevBuf {
  T *ptr = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    // the old allocation stays valid until the new one exists: a failed hipMalloc must not
    // leave a dangling pointer behind (contents are not carried over: every user refills)
    size_t want = n + n / 8 + 64;
    T *fresh = nullptr;
    hipError_t e = hipMalloc((void **)&fresh, want * sizeof(T));
    if (e != hipSuccess && ptr) {  // retry once with the old block returned first
      (void)hipGetLastError();
      HIP_CHECK(hipFree(ptr));
      ptr = nullptr;
      cap = 0;
      e = hipMalloc((void **)&fresh, want * sizeof(T));
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      throw HipError(std::string("hipMalloc of ") + std::to_string(want * sizeof(T)) + " bytes: " +
                     hipGetErrorString(e));
    }
    if (ptr) HIP_CHECK(hipFree(ptr));
    ptr = fresh;
    cap = want;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

// Small transfers between the page-locked staging buffers and device memory as a kernel ON the
// compute stream (the device reads / writes the mapped host memory over PCIe) instead of an
// asynchronous copy: a copy command goes to a DMA engine behind cross-queue barriers, about 10 us
// of latency each way, which is what one MD step of a 4000-atom frame (96 KB in, 128 KB out) pays.
__global__ __launch_bounds__(256) void staged_copy_kernel(double *__restrict__ dst,
                                                          const double *__restrict__ src, size_t n) {
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) dst[k] = src[k];
}
constexpr size_t kStagedCopyMaxBytes = 4u << 20;  // beyond that the DMA engine's bandwidth wins

// `host_side`: the page-locked end of the transfer (for the fallback's copy direction)
void staged_copy(double *dst, const double *src, size_t n, bool to_host, hipStream_t s) {
  if (n == 0) return;
  static const bool use_dma = std::getenv("TA_STAGED_COPY_DMA") != nullptr;  // A/B switch
  if (n * sizeof(double) > kStagedCopyMaxBytes || use_dma) {
    HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(double), to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, s));
    return;
  }
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256);
  hipLaunchKernelGGL(staged_copy_kernel, dim3(blocks), dim3(256), 0, s, dst, src, n);
  HIP_CHECK(hipGetLastError());
}

// Wait for the stream on the latency-critical paths (one MD step has three waits around ~50 us kernels):
// poll the stream for a short while before blocking, since a blocked thread is woken by an interrupt
// many microseconds after the work is done. TA_SYNC_BLOCKING=1 goes straight to the blocking wait.
void wait_stream(hipStream_t s) {
  static const bool blocking = std::getenv("TA_SYNC_BLOCKING") && std::getenv("TA_SYNC_BLOCKING")[0] == '1';
  if (!blocking) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) {
        (void)hipGetLastError();
        break;  // let the blocking wait report it
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;
    }
  }
  HIP_CHECK(hipStreamSynchronize(s));
}

// grow-only page-locked host buffer (staging for the packed uploads / downloads)
struct PinnedBuf {
  char *ptr = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    if (ptr) HIP_CHECK(hipHostFree(ptr));
    ptr = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 256;
    HIP_CHECK(hipHostMalloc((void **)&ptr, want, hipHostMallocDefault));
    cap = want;
  }
  void release() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

struct ChunkPlan {
  ta::AngChunk ch;
  int nb, ng, nz;
};

}  // namespace

struct ta_context {
  int device = 0;
  hipStream_t stream = nullptr;      // stream all work is enqueued on
  hipStream_t own_stream = nullptr;  // created by ta_create
  int kind = 0;
  int n_elements = 0;
  int activation = 0;
  double rmax = 0.0;
  ta::SFParams sf;
  std::vector<ChunkPlan> chunks;     // first-generation kernels: up to 2 betas per launch
  std::vector<ChunkPlan> chunks_v2;  // second-generation kernels: one beta per launch
  bool use_v2 = false;
  ta::MlpDev *mlp_dev = nullptr;     // device copy of mlp[0..n_elements)
  ta::MlpDev mlp[ta::kMaxElements];
  std::vector<void *> model_allocs;
  ta::EamModel *eam = nullptr;
  ta::GrapModel *grap = nullptr;

  ta::HostPairs hp;
  ta::DeviceBatch db;
  bool have_batch = false;
  uint32_t last_want = 0;

  // Inputs travel in one packed upload: [pos | cells | grids | species | frame_of_atom |
  // atom_start | elem_atoms | blk_center]; results come back in one packed download:
  // [energy F | virial 9F | atomic N | forces 3N].
  PinnedBuf stage_in, stage_out;
  DevBuf<char> inbuf;
  DevBuf<double> results;
  size_t o_blk = 0;  // byte offset of blk_center in the packed input
  DevBuf<double> rec, part4, G, dEdG, g, wat, bpart, fown, benergy, mlp_scratch;
  DevBuf<unsigned long long> masks;
  DevBuf<uint32_t> job_word;
  DevBuf<int32_t> job_count;
  DevBuf<int32_t> pair_start, seg_start, pair_i, pair_j, pair_shift, pair_rev;
  ta::NlGrid *d_grids = nullptr;  // view into inbuf
  // device neighbour list (ta_nlist.hip)
  DevBuf<int32_t> nl_wrap, nl_binid, nl_bin_count, nl_bin_start, nl_bin_cursor, nl_bin_atoms, nl_counts;
  DevBuf<unsigned long long> nl_stats, nl_zero;
  unsigned long long *nl_stats_ptr = nullptr;  // statistics block of the builder that made the list
  bool nl_sorted = false;  // list in key order (one-pass builder): reverse pairs by binary search
  bool nl_rev_done = false;  // ... and the reverse index is already there (launched behind the pairs)
  bool nl_zero_clean = false;  // nl_zero is all zero where the next list needs it (see nl_build)
  int64_t nl_zero_atoms = -1;
  int nl_zero_bins = -1;
  DevBuf<double> hvp_buf;  // ta_hessian_vectors: tangents in, force / virial tangents out
#ifdef TA_PHASE_STAMPS
  DevBuf<unsigned long long> stamp_buf;
#endif
  DevBuf<ta::NlRec> nl_recs;
  bool pairs_on_device = false;  // hp holds only the counts; ta_get_pairs downloads on demand
  bool descriptors_valid = false;  // db.G holds the resident batch's descriptors
  DevBuf<double> train_scratch, train_partial, train_grad, train_coeff;
  // force / stress terms of the loss: J[c][p] = dG_c / dD_p of the resident batch (made once per
  // batch by one backward launch per descriptor channel), the direction and its images
  DevBuf<double> jvp_J, tan_dD, tan_dG, tan_dir;
  bool jvp_valid = false;

  // MD loop (ta_set_skin / ta_update_positions): the list covers rmax + skin and is kept while no
  // atom has moved more than skin / 2 from where it was when the list was built
  double skin = 0.0;
  double r_list = 0.0;                         // rmax + skin: cutoff of the resident list
  std::vector<int32_t> keep_species;           // [N]
  std::vector<int32_t> keep_natoms, keep_pbc;  // [F], [3 F]: what a rebuild needs besides stage_in
  std::vector<double> ref_pos, ref_cells;      // positions / cells the resident list was built for
  size_t o_pos = 0, o_cells = 0, o_species = 0;  // byte offsets in the packed input
  // exact list of the current step, extracted on the device from the resident skin list (nl_filter)
  DevBuf<int32_t> ex_pair_i, ex_pair_j, ex_pair_shift, ex_pair_rev, ex_pair_start, ex_pair_stop, ex_seg_start, ex_counts,
      ex_map, ex_blk, ex_slot_q;
  bool filtered = false;                       // db points at the ex_* arrays
  // a copy out of stage_in may still be in flight (set by ta_update_positions, cleared by every wait
  // for the stream on the step path): whoever rewrites stage_in waits for the stream first. (An event
  // recorded behind the copy cost a 5 us bubble between the copy and the next kernel of every MD step.)
  bool upload_pending = false;
  // ta_step: the next compute mirrors its results into stage_out (frame_reduce_kernel); `mirror_want` = the
  // want bits whose results the page-locked image holds right now (0: none), cleared by whoever reuses stage_out
  bool mirror_next = false;
  uint32_t mirror_want = 0;
  int64_t n_list_builds = 0, n_list_reuses = 0;

  hipEvent_t ev[2 * TA_N_KERNEL_SLOTS + 2] = {nullptr};
  std::string err;
};

namespace {

int fail(ta_context *h, int code, const std::string &msg) {
  if (h)
    h->err = msg;
  else
    g_create_error = msg;
  return code;
}

template <typename T>
T *upload(ta_context *h, const std::vector<T> &v) {
  T *d = nullptr;
  size_t n = v.size() ? v.size() : 1;
  HIP_CHECK(hipMalloc((void **)&d, n * sizeof(T)));
  h->model_allocs.push_back(d);
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Power series in u of Hd(u) = exp(-beta u) * 0.5 (1 + cos(pi sqrt(u))) on [0, 1]
// (cosine cutoff nn/cutoff.py:43-48 times the Gaussian of sf.py:166-168 for the
// third side of a triple), in long double, economised to 12, 16 or 24 coefficients when
// the error bound is below 1e-17 (1e-15 for the derivative), otherwise n_hd = 0 (exact path).
void hd_series(int cutoff, double beta, ta::AngChunk &ch) {
  ch.n_hd = 0;
  for (double &c : ch.hd) c = 0.0;
  if (cutoff != TA_CUTOFF_COSINE || !(beta >= 0.0)) return;
  constexpr int NT = 64;
  const long double pi2 = 9.869604401089358618834490999876151135L;
  long double fc[NT], ex[NT], prod[NT];
  long double t = 1.0L;  // (-pi^2)^k / (2k)!
  for (int k = 0; k < NT; ++k) {
    fc[k] = 0.5L * t + (k == 0 ? 0.5L : 0.0L);
    t *= -pi2 / (long double)((2 * k + 1) * (2 * k + 2));
  }
  t = 1.0L;  // (-beta)^k / k!
  for (int k = 0; k < NT; ++k) {
    ex[k] = t;
    t *= -(long double)beta / (long double)(k + 1);
  }
  for (int k = 0; k < NT; ++k) {
    long double acc = 0.0L;
    for (int j = 0; j <= k; ++j) acc += fc[j] * ex[k - j];
    prod[k] = acc;
  }
  // Chebyshev economisation: starting from the degree-(N0 - 1) Taylor polynomial, the leading term
  // a_n u^n is replaced by a_n (u^n - T*_n(u) / L_n), T*_n(u) = T_n(2u - 1) with leading coefficient
  // L_n = 2^(2n-1); that changes the polynomial by at most |a_n| / L_n on [0, 1] and its derivative
  // by at most 2 n^2 |a_n| / L_n, and lowers the degree by one. Repeated down to n coefficients, this
  // reaches the fp64 rounding floor with 12 coefficients for the default beta where the plain
  // series needs 16 (4 / 8 fewer FMAs per triple in the forward / backward kernels).
  constexpr int N0 = 40;
  static long double tstar[N0][N0];  // tstar[n][k]: coefficient of u^k in T*_n
  static bool have_t = false;
  if (!have_t) {
    for (auto &row : tstar)
      for (auto &c : row) c = 0.0L;
    tstar[0][0] = 1.0L;
    tstar[1][0] = -1.0L;
    tstar[1][1] = 2.0L;
    for (int n = 1; n + 1 < N0; ++n)
      for (int k = 0; k <= n + 1; ++k)
        tstar[n + 1][k] = (k > 0 ? 4.0L * tstar[n][k - 1] : 0.0L) - 2.0L * tstar[n][k] - tstar[n - 1][k];
    have_t = true;
  }
  long double tail0 = 0.0L, dtail0 = 0.0L;
  for (int k = N0; k < NT; ++k) {
    tail0 += fabsl(prod[k]);
    dtail0 += (long double)k * fabsl(prod[k]);
  }
  for (int n : {12, 16, 24}) {
    long double a[N0];
    for (int k = 0; k < N0; ++k) a[k] = prod[k];
    long double err = tail0, derr = dtail0;
    for (int d = N0 - 1; d >= n; --d) {
      const long double q = a[d] / tstar[d][d];
      for (int k = 0; k <= d; ++k) a[k] -= q * tstar[d][k];
      err += fabsl(q);
      derr += 2.0L * (long double)d * (long double)d * fabsl(q);
    }
    if (err < 1e-17L && derr < 1e-15L) {
      ch.n_hd = n;
      for (int k = 0; k < n; ++k) ch.hd[k] = (double)a[k];
      return;
    }
  }
}

void build_mlp(ta_context *h, const ta_model_desc *m, int ndim);

void build_sf_model(ta_context *h, const ta_model_desc *m) {
  using namespace ta;
  SFParams &sf = h->sf;
  std::memset(&sf, 0, sizeof(sf));
  const int nel = m->n_elements;
  if (m->n_eta < 1 || m->n_omega < 1 || !m->eta || !m->omega)
    throw std::invalid_argument("eta / omega must be non-empty");
  if (m->n_eta * m->n_omega > kMaxRadial)
    throw std::domain_error("more than 64 radial (eta x omega) combinations");
  if (!(m->rcut > 0.0)) throw std::invalid_argument("rcut must be positive");
  sf.rcut = m->rcut;
  sf.acut = m->angular ? m->acut : m->rcut;
  if (m->angular && !(m->acut > 0.0)) throw std::invalid_argument("acut must be positive");
  sf.inv_rc2 = 1.0 / (sf.rcut * sf.rcut);
  sf.inv_ac2 = 1.0 / (sf.acut * sf.acut);
  sf.two_inv_ac2 = 2.0 * sf.inv_ac2;
  sf.eps = m->eps > 0.0 ? m->eps : 1e-14;  // Precision.high / medium eps, precision.py:113-114
  sf.n_elements = nel;
  sf.n_rad = m->n_eta * m->n_omega;
  sf.angular = m->angular ? 1 : 0;
  sf.cutoff = m->cutoff_function;
  if (sf.cutoff != TA_CUTOFF_COSINE && sf.cutoff != TA_CUTOFF_POLYNOMIAL)
    throw std::invalid_argument("unknown cutoff function");
  // ParameterGrid order: eta outer, omega fastest (sf.py:47-48)
  for (int e = 0; e < m->n_eta; ++e)
    for (int o = 0; o < m->n_omega; ++o) {
      sf.eta[e * m->n_omega + o] = m->eta[e];
      sf.omega[e * m->n_omega + o] = m->omega[o];
    }
  for (int c = 1; c < sf.n_rad; ++c) {
    if ((c & 3) == 0 || sf.omega[c] != sf.omega[c - 1] || !(sf.eta[c - 1] > 0.0)) continue;
    const double k = sf.eta[c] / sf.eta[c - 1];
    if (k >= 2.0 && k <= 16.0 && k == std::nearbyint(k) && k * sf.eta[c - 1] == sf.eta[c] &&
        !std::getenv("TA_NO_ETA_CHAIN"))
      sf.eta_pow[c] = (signed char)k;
  }
  sf.n_radial_dim = nel * sf.n_rad;
  sf.n_ang = 0;
  sf.n_beta = 0;
  if (m->angular) {
    if (m->n_beta < 1 || m->n_gamma < 1 || m->n_zeta < 1 || !m->beta || !m->gamma || !m->zeta)
      throw std::invalid_argument("beta / gamma / zeta must be non-empty for an angular model");
    if (m->n_beta > kMaxBetaSlots)
      throw std::domain_error("more than 3 beta values are not supported by the pair record");
    sf.n_beta = m->n_beta;
    for (int k = 0; k < m->n_beta; ++k) sf.beta[k] = m->beta[k];
    sf.n_ang = m->n_beta * m->n_gamma * m->n_zeta;
    // channel (ib, ig, iz) -> (ib * n_gamma + ig) * n_zeta + iz  (zeta fastest, sf.py:49-51)
    for (int pass = 0; pass < 2; ++pass)
    for (int b0 = 0; b0 < m->n_beta; b0 += (pass == 0 ? 2 : 1))
      for (int g0 = 0; g0 < m->n_gamma; g0 += 2)
        for (int z0 = 0; z0 < m->n_zeta; z0 += 2) {
          ChunkPlan cp;
          std::memset(&cp, 0, sizeof(cp));
          cp.nb = std::min(pass == 0 ? 2 : 1, m->n_beta - b0);
          cp.ng = std::min(2, m->n_gamma - g0);
          cp.nz = std::min(2, m->n_zeta - z0);
          for (int ib = 0; ib < cp.nb; ++ib) {
            cp.ch.beta[ib] = m->beta[b0 + ib];
            cp.ch.hslot[ib] = b0 + ib;
          }
          cp.ch.safe_pow = m->safe_pow ? 1 : 0;
          for (int ig = 0; ig < cp.ng; ++ig) cp.ch.gamma[ig] = m->gamma[g0 + ig];
          for (int iz = 0; iz < cp.nz; ++iz) {
            const double z = m->zeta[z0 + iz];
            cp.ch.zeta[iz] = z;
            cp.ch.kz[iz] = std::pow(2.0, 1.0 - z);
            const double zr = std::nearbyint(z);
            cp.ch.zeta_int[iz] = (zr == z && z >= 1.0 && z <= 1024.0) ? (int)zr : -1;
          }
          for (int ib = 0; ib < cp.nb; ++ib)
            for (int ig = 0; ig < cp.ng; ++ig)
              for (int iz = 0; iz < cp.nz; ++iz)
                cp.ch.chan[(ib * cp.ng + ig) * cp.nz + iz] =
                    ((b0 + ib) * m->n_gamma + (g0 + ig)) * m->n_zeta + (z0 + iz);
          if (pass == 1) hd_series(sf.cutoff, cp.ch.beta[0], cp.ch);
          (pass == 0 ? h->chunks : h->chunks_v2).push_back(cp);
        }
  }
  const int n_aterms = m->angular ? nel * (nel + 1) / 2 : 0;
  (void)0;
  sf.ndim = sf.n_radial_dim + n_aterms * sf.n_ang;
  h->rmax = std::max(sf.rcut, sf.acut);
  build_mlp(h, m, sf.ndim);
}

// per-element MLP weights -> padded device copies, both orientations
void build_mlp(ta_context *h, const ta_model_desc *m, int ndim) {
  using namespace ta;
  const int nel = m->n_elements;
  struct { int ndim; } sf{ndim};
  if (!m->n_layers || !m->layer_sizes || !m->weights)
    throw std::invalid_argument("MLP description missing");
  h->activation = m->activation;
  if (m->activation < 0 || m->activation > TA_ACT_ELU)
    throw std::invalid_argument("unknown activation");
  const int32_t *sizes = m->layer_sizes;
  const double *wsrc = m->weights;
  for (int el = 0; el < nel; ++el) {
    MlpDev &md = h->mlp[el];
    const int L = m->n_layers[el];
    if (L < 1 || L > kMaxLayers) throw std::domain_error("MLP depth out of range (1..8 layers)");
    if (sizes[0] != sf.ndim)
      throw std::invalid_argument("MLP input size does not match the descriptor length");
    if (sizes[L] != 1) throw std::invalid_argument("MLP output size must be 1");
    md.n_layers = L;
    md.max_np = md.max_kp = 0;
    for (int l = 0; l < L; ++l) {
      MlpLayerDev &ly = md.layer[l];
      ly.k = sizes[l];
      ly.n = sizes[l + 1];
      if (ly.k < 1 || ly.n < 1 || ly.k > 512 || ly.n > 512)
        throw std::domain_error("MLP layer width out of range (1..512)");
      ly.kp = round_up(ly.k, 16);
      ly.np = round_up(ly.n, 16);
      ly.act = (l < L - 1) ? 1 : 0;
      // ResNet skip: hidden layer j > 0 with equal widths (convolutional.py:272)
      ly.res = (m->use_resnet_dt && l > 0 && l < L - 1 && ly.k == ly.n) ? 1 : 0;
      std::vector<double> w((size_t)ly.kp * ly.np, 0.0), wt((size_t)ly.np * ly.kp, 0.0),
          bb(ly.np, 0.0);
      for (int k = 0; k < ly.k; ++k)
        for (int n = 0; n < ly.n; ++n) {
          const double v = wsrc[(size_t)k * ly.n + n];
          w[(size_t)k * ly.np + n] = v;
          wt[(size_t)n * ly.kp + k] = v;
        }
      wsrc += (size_t)ly.k * ly.n;
      for (int n = 0; n < ly.n; ++n) bb[n] = wsrc[n];
      wsrc += ly.n;
      ly.w = upload(h, w);
      ly.wt = upload(h, wt);
      ly.b = upload(h, bb);
      md.max_np = std::max(md.max_np, ly.np);
      md.max_kp = std::max(md.max_kp, ly.kp);
    }
    if (m->minmax_scale) {
      if (!m->xlo || !m->xhi) throw std::invalid_argument("minmax_scale set but xlo/xhi missing");
      std::vector<double> lo(m->xlo + (size_t)el * sf.ndim, m->xlo + (size_t)(el + 1) * sf.ndim);
      std::vector<double> hi(m->xhi + (size_t)el * sf.ndim, m->xhi + (size_t)(el + 1) * sf.ndim);
      md.xlo = upload(h, lo);
      md.xhi = upload(h, hi);
    }
    sizes += L + 1;
  }
  {
    std::vector<MlpDev> all(h->mlp, h->mlp + nel);
    h->mlp_dev = upload(h, all);
  }
}

void upload_batch(ta_context *h) {
  using namespace ta;
  HostPairs &hp = h->hp;
  DeviceBatch &db = h->db;
  const size_t N = (size_t)hp.n_atoms, P = (size_t)hp.n_pairs;
  const int nel = h->n_elements;
  const size_t F = (size_t)db.n_frames;
  db.n_atoms = hp.n_atoms;
  db.n_pairs = hp.n_pairs;
  db.nnl_max = hp.nnl_max;

  auto put = [&](auto &buf, const auto &vec) {
    buf.ensure(vec.size());
    if (!vec.empty())
      HIP_CHECK(hipMemcpyAsync(buf.ptr, vec.data(), vec.size() * sizeof(vec[0]),
                               hipMemcpyHostToDevice, h->stream));
  };
  if (h->pairs_on_device) {
    h->pair_i.ensure(P);
    h->pair_j.ensure(P);
    h->pair_shift.ensure(3 * P);
    h->pair_rev.ensure(P);
  } else {
    put(h->pair_start, hp.pair_start);
    put(h->seg_start, hp.seg_start);
    put(h->pair_i, hp.pair_i);
    put(h->pair_j, hp.pair_j);
    put(h->pair_shift, hp.pair_shift);
    put(h->pair_rev, hp.pair_rev);
  }

  const bool has_mlp = h->kind == TA_MODEL_SF_MLP || h->kind == TA_MODEL_GRAP_MLP;
  const int D = has_mlp ? h->sf.ndim : 1;
  h->rec.ensure(P * kRecDoubles);
  if (h->kind == TA_MODEL_SF_MLP && h->sf.angular) {
    h->part4.ensure((size_t)nel * h->sf.n_ang * P);
    // one 64-bit candidate mask per pair and per block of 64 rotation steps (n/2 steps in all)
    h->masks.ensure((size_t)((hp.nnl_max / 2 + 63) / 64 + 1) * P);
  }
  h->G.ensure(N * D);
  h->dEdG.ensure(N * D);
  h->g.ensure(4 * P);
  h->wat.ensure(9 * N);
  h->bpart.ensure(10 * ((N + 15) / 16) + 10);
  h->fown.ensure(12 * N + 12);
  h->results.ensure(10 * F + 4 * N);
  h->benergy.ensure(1);

  db.pair_start = h->pair_start.ptr;
  db.seg_start = h->seg_start.ptr;
  db.pair_i = h->pair_i.ptr;
  db.pair_j = h->pair_j.ptr;
  db.pair_shift = h->pair_shift.ptr;
  db.pair_rev = h->pair_rev.ptr;
  db.rec = h->rec.ptr;
  db.part4 = h->part4.ptr;
  db.masks = h->masks.ptr;
  db.G = h->G.ptr;
  db.dEdG = h->dEdG.ptr;
  db.g = h->g.ptr;
  db.wat = h->wat.ptr;
  db.bpart = h->bpart.ptr;
  db.fown = h->fown.ptr;
  db.energy = h->results.ptr;
  db.virial = h->results.ptr + F;
  db.eatom = h->results.ptr + 10 * F;
  db.forces = h->results.ptr + 10 * F + N;
  db.batch_energy = h->benergy.ptr;
}

ta::NlWork nl_work(ta_context *h) {
  return ta::NlWork{h->nl_wrap.ptr,       h->nl_binid.ptr,     h->nl_bin_count.ptr, h->nl_bin_start.ptr,
                    h->nl_bin_cursor.ptr, h->nl_bin_atoms.ptr, h->nl_recs.ptr,      h->nl_counts.ptr,
                    h->seg_start.ptr,     h->nl_stats.ptr};
}

// Neighbour list on the device, part 1. Leaves the counts in h->hp (n_pairs, n_triples, nnl_max,
// pair_start) for the sizing done by the caller. The packed input (positions, species, grids ...) is
// already on its way to the device.
//
// One-pass builder (ta_nlist.hip::nl_build) first: it also WRITES the pairs (key order) while they fit
// the pair arrays as they stand; when they do not (first batch, or a list that grew) the arrays grow and
// it runs again. The two-pass builder (count, sizes to the host, fill) stays for what the one-pass one
// declines: more than 384 neighbours per atom, shifts beyond +-511 cells, TA_NL_TWO_PASS=1.
void build_pairs_on_device(ta_context *h, size_t N, int n_bins) {
  using namespace ta;
  HostPairs &hp = h->hp;
  hipStream_t s = h->stream;
  const int nel = h->n_elements;
  h->nl_wrap.ensure(3 * N);
  h->nl_binid.ensure(N);
  h->nl_bin_start.ensure((size_t)n_bins + 1);
  h->nl_bin_atoms.ensure(N);
  h->nl_recs.ensure(N);
  h->seg_start.ensure(N * (nel + 1) + 1);
  h->pair_start.ensure(N + 1);
  // counts and per-atom offsets come back through page-locked memory
  h->stage_out.ensure(64 + (N + 1) * sizeof(int32_t));
  unsigned long long *stats = reinterpret_cast<unsigned long long *>(h->stage_out.ptr);
  int32_t *starts = reinterpret_cast<int32_t *>(h->stage_out.ptr + 64);
  const int32_t *si = reinterpret_cast<const int32_t *>(stats);
  const bool two_pass_only = std::getenv("TA_NL_TWO_PASS") && std::getenv("TA_NL_TWO_PASS")[0] == '1';
  h->nl_sorted = false;
  const bool kernel_writes_host = !(std::getenv("TA_NL_COPY_STARTS") && std::getenv("TA_NL_COPY_STARTS")[0] == '1');
  if (!two_pass_only) {
    {  // the zero block stays clean from list to list while its layout (atoms, bins) does not change
      const unsigned long long *before = h->nl_zero.ptr;
      h->nl_zero.ensure(nl_build_zero_words((int)N, n_bins));
      if (h->nl_zero.ptr != before || h->nl_zero_atoms != (int64_t)N || h->nl_zero_bins != n_bins) h->nl_zero_clean = false;
      h->nl_zero_atoms = (int64_t)N;
      h->nl_zero_bins = n_bins;
    }
    h->nl_rev_done = false;
    for (int attempt = 0; attempt < 2 && !h->nl_sorted; ++attempt) {
      const size_t capacity = std::min(std::min(h->pair_i.cap, h->pair_j.cap), h->pair_shift.cap / 3);
      // the reverse index is launched right behind the pairs, before the count is known here: over as many
      // pairs as the previous list of these atoms had (+ a quarter), at most what the arrays hold
      h->pair_rev.ensure(capacity);
      const size_t guess = (hp.n_atoms == (int64_t)N && hp.n_pairs > 0) ? (size_t)hp.n_pairs + (size_t)hp.n_pairs / 4 + 4096
                                                                         : capacity;
      const size_t rev_cover = std::min(std::min(capacity, guess), (size_t)INT32_MAX);
      NlWork w = nl_work(h);
      // the kernel writes the per-atom offsets to the page-locked buffer itself; the statistics follow
      // in one small copy kernel once every group is through
      nl_build((int)N, n_bins, nel, h->r_list, h->db.pos, h->db.species, h->db.frame_of_atom, h->d_grids, w,
               h->nl_zero.ptr, h->nl_zero_clean, (long long)std::min<size_t>(capacity, (size_t)INT32_MAX), h->pair_start.ptr,
               kernel_writes_host ? starts : nullptr, h->pair_i.ptr, h->pair_j.ptr, h->pair_shift.ptr,
               h->pair_rev.ptr, (long long)rev_cover, s);
      HIP_CHECK(hipGetLastError());
      h->nl_zero_clean = false;  // until this list is through (an exception below leaves it marked dirty)
      if (!kernel_writes_host)
        staged_copy(reinterpret_cast<double *>(starts), reinterpret_cast<const double *>(h->pair_start.ptr),
                    (N + 2) / 2, true, s);
      staged_copy(reinterpret_cast<double *>(stats), reinterpret_cast<const double *>(h->nl_zero.ptr), 8, true, s);
      wait_stream(s);
      h->nl_zero_clean = N > 0;  // the kernels ran to the end: histogram cleared, the rest is cleared on entry
      if (si[3] != 0) break;  // beyond the one-pass builder's limits
      if (stats[4] > (unsigned long long)INT32_MAX) throw std::runtime_error("batch too large for 32-bit pair indices");
      if (stats[4] <= capacity) {
        h->nl_sorted = true;
        h->nl_rev_done = stats[4] <= rev_cover;  // (else fill_pairs_on_device launches it over the whole list)
        if (h->nl_rev_done && si[6] != 0)
          throw std::runtime_error("neighbour list is not symmetric (reverse pair missing)");
      } else {
        h->pair_i.ensure((size_t)stats[4]);
        h->pair_j.ensure((size_t)stats[4]);
        h->pair_shift.ensure(3 * (size_t)stats[4]);
      }
    }
    if (h->nl_sorted) h->nl_stats_ptr = h->nl_zero.ptr;
  }
  if (!h->nl_sorted) {
    h->nl_bin_count.ensure((size_t)n_bins + 1);
    h->nl_bin_cursor.ensure((size_t)n_bins + 1);
    h->nl_counts.ensure(N * (nel + 1) + 1);
    h->nl_stats.ensure(8);
    NlWork w = nl_work(h);
    nl_count((int)N, n_bins, nel, h->r_list, h->db.pos, h->db.species, h->db.frame_of_atom, h->d_grids, w,
             h->pair_start.ptr, s);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(stats, h->nl_stats.ptr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(starts, h->pair_start.ptr, (N + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (stats[4] > (unsigned long long)INT32_MAX || si[4] < 0 || (unsigned long long)si[4] != stats[4])
      throw std::runtime_error("batch too large for 32-bit pair indices");
    h->nl_stats_ptr = h->nl_stats.ptr;
  }
  hp.n_atoms = (int64_t)N;
  hp.n_pairs = (int64_t)stats[4];
  hp.n_triples = (int64_t)stats[0];
  hp.nnl_max = si[2];
  hp.pair_start.assign(starts, starts + N + 1);
  hp.seg_start.clear();
  hp.pair_i.clear();
  hp.pair_j.clear();
  hp.pair_shift.clear();
  hp.pair_rev.clear();
  h->pairs_on_device = true;
}

// part 2, after the pair buffers are sized: the pairs (two-pass builder only) and the reverse index
void fill_pairs_on_device(ta_context *h) {
  using namespace ta;
  if (h->nl_sorted) {
    nl_reverse_sorted(h->hp.n_pairs, h->n_elements, h->db.species, h->seg_start.ptr, h->pair_i.ptr,
                      h->pair_j.ptr, h->pair_shift.ptr, h->pair_rev.ptr, h->nl_stats_ptr, h->stream);
  } else {
    NlWork w = nl_work(h);
    nl_fill((int)h->hp.n_atoms, h->hp.n_pairs, h->n_elements, h->r_list, h->db.pos, h->db.species,
            h->db.frame_of_atom, h->d_grids, w, h->pair_i.ptr, h->pair_j.ptr, h->pair_shift.ptr,
            h->pair_rev.ptr, h->stream);
  }
  HIP_CHECK(hipGetLastError());
}

#ifdef TA_PHASE_STAMPS
// diagnostic builds: phase stamps of the angular kernels of the LAST evaluation, written to the file
// named by TA_PHASE_STAMPS_OUT at ta_destroy (scripts/phase_stamps.sh)
static void dump_stamps(ta_context *h) {
  const char *path = std::getenv("TA_PHASE_STAMPS_OUT");
  if (!path || !h->db.stamps || h->db.n_blk <= 0) return;
  const size_t n = (size_t)2 * h->db.n_blk * 8;
  std::vector<unsigned long long> v(n);
  (void)hipStreamSynchronize(h->stream);
  (void)hipMemcpy(v.data(), h->db.stamps, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  FILE *fp = std::fopen(path, "w");
  if (!fp) return;
  for (int k = 0; k < 2; ++k)
    for (int b = 0; b < h->db.n_blk; ++b) {
      std::fprintf(fp, "%d %d", k, b);
      for (int q = 0; q < 8; ++q) std::fprintf(fp, " %llu", v[((size_t)k * h->db.n_blk + b) * 8 + q]);
      std::fprintf(fp, "\n");
    }
  std::fclose(fp);
}
#endif

void compute_impl(ta_context *h, uint32_t want, bool timed, double *slot_ms) {
  using namespace ta;
#ifdef TA_PHASE_STAMPS
  if (h->db.n_blk > 0) {
    h->stamp_buf.ensure((size_t)2 * h->db.n_blk * 8 + 8);
    h->db.stamps = h->stamp_buf.ptr;
    (void)hipMemsetAsync(h->db.stamps, 0, (size_t)2 * h->db.n_blk * 8 * sizeof(unsigned long long), h->stream);
  }
#endif
  h->db.rec4 = nullptr;  // only the second-generation angular path below sets it
  h->db.own_sums = 0;    // likewise
  const DeviceBatch &db = h->db;
  hipStream_t s = h->stream;
  const bool need_forces = (want & (TA_WANT_FORCES | TA_WANT_VIRIAL)) != 0;
  auto begin = [&](int slot) {
    if (timed) HIP_CHECK(hipEventRecord(h->ev[2 * slot], s));
  };
  auto end = [&](int slot) {
    if (timed) HIP_CHECK(hipEventRecord(h->ev[2 * slot + 1], s));
  };
  bool used[TA_N_KERNEL_SLOTS] = {false};

  // energies again on the descriptors already resident (training steps: only the MLP changed)
  const bool has_mlp = h->kind == TA_MODEL_SF_MLP || h->kind == TA_MODEL_GRAP_MLP;
  if ((want & TA_WANT_REUSE_DESCRIPTORS) && has_mlp && h->descriptors_valid && !need_forces) {
    begin(TA_K_MLP);
    launch_mlp_all(h->mlp_dev, h->mlp, h->n_elements, h->activation, h->sf.ndim, db,
                   h->mlp_scratch.ptr, s);
    end(TA_K_MLP);
    used[TA_K_MLP] = true;
  } else if (h->kind == TA_MODEL_SF_MLP) {
    // second-generation angular path: 32-byte pair records {D, r^2} in the same buffer
    // (TA_FULL_RECORDS=1 keeps the 64-byte ones for A/B runs)
    static const bool full_records = getenv("TA_FULL_RECORDS") != nullptr;
    h->db.rec4 = (h->sf.angular && h->use_v2 && !full_records) ? h->db.rec : nullptr;
    if (!h->use_v2) {
      // second-generation forward kernels compute the pair geometry while staging
      begin(TA_K_PAIR_GEOMETRY);
      launch_pair_geometry(h->sf, db, s);
      end(TA_K_PAIR_GEOMETRY);
      used[TA_K_PAIR_GEOMETRY] = true;
    }
    h->sf.ang_scale = h->use_v2 ? 1.0 : 0.5;
    // the second-generation forward kernel assembles the descriptors itself in its last launch
    const bool reduce_in_forward = h->sf.angular && h->use_v2;
    if (h->sf.angular) {
      begin(TA_K_G4_FORWARD);
      if (h->use_v2) {
        bool geometry = true;
        size_t left = h->chunks_v2.size();
        for (const ChunkPlan &cp : h->chunks_v2) {
          --left;
          launch_g4_forward_v2(h->sf, cp.ch, cp.ng, cp.nz, geometry, left == 0, db, s);
          geometry = false;
        }
      } else
        for (const ChunkPlan &cp : h->chunks) launch_g4_forward(h->sf, cp.ch, cp.nb, cp.ng, cp.nz, db, s);
      end(TA_K_G4_FORWARD);
      used[TA_K_G4_FORWARD] = true;
    }
    if (!reduce_in_forward) {
      begin(TA_K_DESCRIPTOR_REDUCE);
      launch_descriptor_reduce(h->sf, db, s);
      end(TA_K_DESCRIPTOR_REDUCE);
      used[TA_K_DESCRIPTOR_REDUCE] = true;
    }
    begin(TA_K_MLP);
    launch_mlp_all(h->mlp_dev, h->mlp, h->n_elements, h->activation, h->sf.ndim, db,
                   h->mlp_scratch.ptr, s);
    end(TA_K_MLP);
    used[TA_K_MLP] = true;
    if (need_forces) {
      begin(TA_K_BACKWARD);
      if (h->sf.angular) {
        bool first = true;
        if (h->use_v2)
          for (const ChunkPlan &cp : h->chunks_v2) {
            launch_backward_v2(h->sf, cp.ch, cp.ng, cp.nz, first, db, s);
            first = false;
          }
        else
          for (const ChunkPlan &cp : h->chunks) {
            launch_backward(h->sf, cp.ch, cp.nb, cp.ng, cp.nz, first, false, db, s);
            first = false;
          }
      } else {
        AngChunk dummy;
        std::memset(&dummy, 0, sizeof(dummy));
        launch_backward(h->sf, dummy, 1, 1, 1, true, true, db, s);
      }
      end(TA_K_BACKWARD);
      used[TA_K_BACKWARD] = true;
      begin(TA_K_FORCE_GATHER);
      launch_force_gather(h->sf, db, s);
      end(TA_K_FORCE_GATHER);
      used[TA_K_FORCE_GATHER] = true;
    }
  } else if (h->kind == TA_MODEL_GRAP_MLP) {
    // 32-byte pair records, as on the second-generation angular path; the `nn` filter network
    // takes r from the full records its geometry pre-pass writes
    h->db.rec4 = (!ta::grap_uses_filter_net(h->grap) && !getenv("TA_FULL_RECORDS")) ? h->db.rec : nullptr;
    begin(TA_K_GRAP);
    launch_grap_forward(h->grap, db, h->sf.eps, s);  // computes the pair geometry while staging
    end(TA_K_GRAP);
    used[TA_K_GRAP] = true;
    begin(TA_K_MLP);
    launch_mlp_all(h->mlp_dev, h->mlp, h->n_elements, h->activation, h->sf.ndim, db,
                   h->mlp_scratch.ptr, s);
    end(TA_K_MLP);
    used[TA_K_MLP] = true;
    if (need_forces) {
      begin(TA_K_BACKWARD);
      h->db.own_sums = getenv("TA_NO_OWN_SUMS") ? 0 : 1;  // one wavefront per centre: the sums are nearly free there
      launch_grap_backward(h->grap, db, s);
      end(TA_K_BACKWARD);
      used[TA_K_BACKWARD] = true;
      begin(TA_K_FORCE_GATHER);
      launch_force_gather(h->sf, db, s);
      end(TA_K_FORCE_GATHER);
      used[TA_K_FORCE_GATHER] = true;
    }
  } else {
    h->db.rec4 = getenv("TA_FULL_RECORDS") ? nullptr : h->db.rec;  // 32-byte pair records {D, r^2}
    begin(TA_K_EAM);
    eam_compute(h->eam, db, want, s, nullptr);
    end(TA_K_EAM);
    used[TA_K_EAM] = true;
  }
  begin(TA_K_FRAME_REDUCE);
  {
    // MD step (ta_step / ta_step_view): the last kernel also writes the results' page-locked image
    double *mirror = nullptr;
    int64_t n_tail = 0;
    h->mirror_want = 0;
    if (h->mirror_next && !timed) {
      const size_t N = (size_t)db.n_atoms, F = (size_t)db.n_frames;
      h->stage_out.ensure((10 * F + 4 * N + 2) * sizeof(double));
      mirror = reinterpret_cast<double *>(h->stage_out.ptr);
      n_tail = (int64_t)(need_forces ? 4 * N : N);
      h->mirror_want = want | TA_WANT_ENERGY | TA_WANT_ATOMIC;
    }
    launch_frame_reduce(db, need_forces, mirror, n_tail, s);
  }
  end(TA_K_FRAME_REDUCE);
  used[TA_K_FRAME_REDUCE] = true;
  HIP_CHECK(hipGetLastError());
  h->last_want = want;
  h->descriptors_valid = true;

  if (timed) {
    HIP_CHECK(hipStreamSynchronize(s));
    for (int k = 0; k < TA_N_KERNEL_SLOTS; ++k) {
      if (!used[k]) continue;
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, h->ev[2 * k], h->ev[2 * k + 1]));
      slot_ms[k] += ms;
    }
  }
}

template <typename F>
int guarded(ta_context *h, F &&fn) {
  try {
    if (h) HIP_CHECK(hipSetDevice(h->device));
    fn();
    return TA_OK;
  } catch (const HipError &e) {
    return fail(h, TA_ERR_HIP, e.what());
  } catch (const std::domain_error &e) {
    return fail(h, TA_ERR_UNSUPPORTED, e.what());
  } catch (const std::invalid_argument &e) {
    return fail(h, TA_ERR_INVALID, e.what());
  } catch (const std::bad_alloc &) {
    return fail(h, TA_ERR_NOMEM, "host allocation failed");
  } catch (const std::exception &e) {
    return fail(h, TA_ERR_INVALID, e.what());
  }
}

}  // namespace

extern "C" {

int ta_abi_version(void) { return TA_ABI_VERSION; }
int ta_model_desc_size(void) { return (int)sizeof(ta_model_desc); }

int ta_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *ta_last_error(ta_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ta_create(const ta_model_desc *model, int device, ta_handle *out) {
  if (!model || !out) return fail(nullptr, TA_ERR_INVALID, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, TA_ERR_HIP, "no HIP device available: this library has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, TA_ERR_INVALID, "device index out of range");
  if (model->n_elements < 1 || model->n_elements > ta::kMaxElements)
    return fail(nullptr, TA_ERR_UNSUPPORTED, "n_elements must be in 1..8");
  ta_context *h = new ta_context();
  h->device = device;
  h->kind = model->kind;
  h->n_elements = model->n_elements;
  int rc = guarded(h, [&]() {
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      throw HipError(std::string("device is ") + prop.gcnArchName +
                     ", this library is built for gfx950 only");
    HIP_CHECK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    for (auto &e : h->ev) HIP_CHECK(hipEventCreate(&e));
    if (model->kind == TA_MODEL_SF_MLP) {
      build_sf_model(h, model);
    } else if (model->kind == TA_MODEL_GRAP_MLP) {
      if (model->cutoff_function != TA_CUTOFF_COSINE && model->cutoff_function != TA_CUTOFF_POLYNOMIAL)
        throw std::invalid_argument("unknown cutoff function");
      std::string err;
      h->grap = ta::grap_create(model, err);
      if (!h->grap) throw std::invalid_argument(err);
      h->rmax = model->rcut;
      std::memset(&h->sf, 0, sizeof(h->sf));
      h->sf.n_elements = model->n_elements;
      h->sf.eps = model->eps > 0.0 ? model->eps : 1e-14;
      h->sf.ndim = ta::grap_ndim(h->grap);
      build_mlp(h, model, h->sf.ndim);
    } else if (model->kind == TA_MODEL_EAM_ALLOY || model->kind == TA_MODEL_EAM_ADP) {
      std::string err;
      h->eam = ta::eam_create(model, err);
      if (!h->eam) throw std::invalid_argument(err);
      h->rmax = model->rcut;
      std::memset(&h->sf, 0, sizeof(h->sf));
      h->sf.n_elements = model->n_elements;
      h->sf.eps = model->eps > 0.0 ? model->eps : 1e-14;
    } else {
      throw std::invalid_argument("unknown model kind");
    }
  });
  if (rc != TA_OK) {
    g_create_error = h->err;
    ta_destroy(h);
    return rc;
  }
  *out = h;
  return TA_OK;
}

int ta_destroy(ta_handle h) {
  if (!h) return TA_OK;
  (void)hipSetDevice(h->device);
#ifdef TA_PHASE_STAMPS
  dump_stamps(h);
#endif
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void *p : h->model_allocs) (void)hipFree(p);
  if (h->eam) ta::eam_destroy(h->eam);
  if (h->grap) ta::grap_destroy(h->grap);
  h->stage_in.release(); h->stage_out.release(); h->inbuf.release(); h->results.release();
  h->rec.release(); h->part4.release(); h->G.release();
  h->dEdG.release(); h->g.release(); h->wat.release(); h->bpart.release(); h->fown.release();
  h->benergy.release(); h->mlp_scratch.release();
  h->train_scratch.release(); h->train_partial.release(); h->train_grad.release(); h->train_coeff.release();
  h->jvp_J.release(); h->tan_dD.release(); h->tan_dG.release(); h->tan_dir.release();
  h->pair_start.release(); h->seg_start.release(); h->pair_i.release(); h->pair_j.release();
  h->pair_shift.release(); h->pair_rev.release();
  for (auto *b : {&h->ex_pair_i, &h->ex_pair_j, &h->ex_pair_shift, &h->ex_pair_rev, &h->ex_pair_start, &h->ex_pair_stop,
                  &h->ex_seg_start, &h->ex_counts, &h->ex_map, &h->ex_blk, &h->ex_slot_q})
    b->release();
  h->masks.release(); h->job_word.release(); h->job_count.release();
  for (auto *b : {&h->nl_wrap, &h->nl_binid, &h->nl_bin_count, &h->nl_bin_start, &h->nl_bin_cursor,
                  &h->nl_bin_atoms, &h->nl_counts})
    b->release();
  h->nl_stats.release();
  h->nl_zero.release();
  h->hvp_buf.release();
  h->nl_recs.release();
  for (auto &e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return TA_OK;
}

}  // extern "C" (reopened below)

namespace {
// job lists of the angular kernels (ta_kernels_v2.hip::make_jobs): room for `n_blk` workgroups
void ensure_job_lists(ta_context *h, size_t n_blk) {
  static const bool off = std::getenv("TA_NO_JOBS") != nullptr;  // A/B switch: per-lane masks + re-dealing
  if (off || !h->use_v2 || n_blk == 0) {
    h->db.job_count = nullptr;
    return;
  }
  const int stride = ta::v2_job_stride(h->db.cap);
  h->job_word.ensure(n_blk * (size_t)stride + 8);
  h->job_count.ensure(n_blk + 8);
  h->db.job_word = h->job_word.ptr;
  h->db.job_count = h->job_count.ptr;
  h->db.job_stride = stride;
}

// MD loop with a Verlet skin: the evaluation kernels get the EXACT list of the current positions,
// extracted from the resident skin list on the device (ta_nlist.hip::nl_filter; no host round trip),
// so they never pay for the skin. Only for models whose kernels are driven by centres (second-
// generation symmetry-function kernels, plain EAM); the others run on the skin list itself, where
// pairs beyond the cutoff contribute nothing.
bool filter_applies(const ta_context *h) {
  if (!(h->skin > 0.0) || h->hp.n_atoms == 0 || std::getenv("TA_NO_LIST_FILTER")) return false;
  if (h->kind == TA_MODEL_SF_MLP) return h->use_v2;
  if (h->kind == TA_MODEL_EAM_ALLOY) return ta::eam_is_plain(h->eam);
  return false;
}

void apply_filter(ta_context *h) {
  using namespace ta;
  const size_t N = (size_t)h->hp.n_atoms, P = (size_t)h->hp.n_pairs;
  const int nel = h->n_elements;
  h->ex_pair_i.ensure(P + 1);
  h->ex_pair_j.ensure(P + 1);
  h->ex_pair_shift.ensure(3 * P + 3);
  h->ex_pair_rev.ensure(P + 1);
  h->ex_pair_start.ensure(N + 1);
  h->ex_pair_stop.ensure(N + 1);
  h->ex_seg_start.ensure(N * (nel + 1) + 1);
  h->ex_map.ensure(P + 1);
  const int n_run_slots = nl_filter_blocks((int)N);
  h->ex_blk.ensure((size_t)n_run_slots + 4);
  const bool blocks = h->kind == TA_MODEL_SF_MLP;
  // symmetry-function models: no reverse-index launch; force_gather looks the reverse pair up through the map
  // (TA_FILTER_REV_KERNEL=1: the launch, for A/B)
  const bool indirect = blocks && !std::getenv("TA_FILTER_REV_KERNEL");
  if (indirect) h->ex_slot_q.ensure(P + 1);
  nl_filter((int)N, (int64_t)P, nel, h->rmax, h->db.pos, h->db.cells, h->db.frame_of_atom,
            h->pair_start.ptr, h->seg_start.ptr, h->pair_j.ptr, h->pair_shift.ptr, h->pair_rev.ptr, h->ex_map.ptr,
            h->ex_seg_start.ptr, h->ex_pair_start.ptr, h->ex_pair_stop.ptr, h->ex_pair_i.ptr, h->ex_pair_j.ptr,
            h->ex_pair_shift.ptr, h->ex_pair_rev.ptr, indirect ? h->ex_slot_q.ptr : nullptr, h->db.cap,
            blocks ? h->ex_blk.ptr : nullptr, h->stream);
  h->db.slot_q = indirect ? h->ex_slot_q.ptr : nullptr;
  h->db.rev_super = indirect ? h->pair_rev.ptr : nullptr;
  h->db.rev_map = indirect ? h->ex_map.ptr : nullptr;
  HIP_CHECK(hipGetLastError());
  h->db.pair_start = h->ex_pair_start.ptr;
  h->db.pair_stop = h->ex_pair_stop.ptr;
  h->db.seg_start = h->ex_seg_start.ptr;
  h->db.pair_i = h->ex_pair_i.ptr;
  h->db.pair_j = h->ex_pair_j.ptr;
  h->db.pair_shift = h->ex_pair_shift.ptr;
  h->db.pair_rev = h->ex_pair_rev.ptr;
  if (blocks) {
    h->db.blk_center = h->ex_blk.ptr;
    h->db.n_blk = n_run_slots;  // 16 run slots per group of 16 centres; unused ones are empty runs
    h->db.blk_groups = n_run_slots / 16;
    h->db.n_blk_dev = nullptr;
    ensure_job_lists(h, (size_t)n_run_slots);
  }
  h->filtered = true;
}

// the body of ta_set_frames (also the rebuild path of ta_update_positions); throws
void set_frames_impl(ta_context *h, int32_t n_frames, const ta_frame *frames, ta_batch_info *info) {
  for (int f = 0; f < n_frames; ++f) {
    const ta_frame &fr = frames[f];
    if (fr.n_atoms < 0 || (fr.n_atoms > 0 && (!fr.species || !fr.positions)) || !fr.cell || !fr.pbc)
      throw std::invalid_argument("frame " + std::to_string(f) + ": null array");
  }
  const auto t_begin = std::chrono::steady_clock::now();
  if (h->upload_pending) {  // stage_in is about to be rewritten
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->upload_pending = false;
  }
  // no batch is resident until this call has succeeded: a failure below (allocation, too many
  // neighbours, asymmetric list ...) must not leave the previous batch's flags standing over
  // buffers that were already regrown or repointed
  h->have_batch = false;
  h->descriptors_valid = false;
  h->jvp_valid = false;
  h->mirror_want = 0;  // (stage_out carries the builder's counts from here on)
  h->filtered = false;
  h->db.slot_q = h->db.rev_super = h->db.rev_map = nullptr;
  h->db.pair_stop = nullptr;
  h->db.blk_groups = 0;
  h->db.n_blk_dev = nullptr;
  h->r_list = h->rmax + h->skin;
  size_t N = 0;
  for (int f = 0; f < n_frames; ++f) N += (size_t)frames[f].n_atoms;
  if (N >= (1u << 30)) throw std::runtime_error("batch too large for 32-bit atom indices");
  const size_t F = (size_t)n_frames;
  const int nel = h->n_elements;
  // packed input, every section 16-byte aligned
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_pos = 0;
  const size_t o_cells = align16(o_pos + 3 * N * sizeof(double));
  const size_t o_grids = align16(o_cells + 9 * F * sizeof(double));
  const size_t o_species = align16(o_grids + F * sizeof(ta::NlGrid));
  const size_t o_foa = align16(o_species + N * sizeof(int32_t));
  const size_t o_astart = align16(o_foa + N * sizeof(int32_t));
  const size_t o_elem = align16(o_astart + (F + 1) * sizeof(int32_t));
  const size_t o_blk = align16(o_elem + N * sizeof(int32_t));
  const size_t total = align16(o_blk + (N + 2) * sizeof(int32_t));
  h->stage_in.ensure(total);
  h->inbuf.ensure(total);
  char *hb = h->stage_in.ptr;
  double *pos = reinterpret_cast<double *>(hb + o_pos);
  double *cells = reinterpret_cast<double *>(hb + o_cells);
  ta::NlGrid *grids = reinterpret_cast<ta::NlGrid *>(hb + o_grids);
  int32_t *species = reinterpret_cast<int32_t *>(hb + o_species);
  int32_t *foa = reinterpret_cast<int32_t *>(hb + o_foa);
  int32_t *astart = reinterpret_cast<int32_t *>(hb + o_astart);
  int32_t *elem_atoms = reinterpret_cast<int32_t *>(hb + o_elem);
  size_t a = 0;
  astart[0] = 0;
  for (int f = 0; f < n_frames; ++f) {
    const ta_frame &fr = frames[f];
    const size_t n = (size_t)fr.n_atoms;
    if (n) {
      std::memcpy(&pos[3 * a], fr.positions, 3 * n * sizeof(double));
      std::memcpy(&species[a], fr.species, n * sizeof(int32_t));
    }
    std::memcpy(&cells[9 * (size_t)f], fr.cell, 9 * sizeof(double));
    for (size_t i = a; i < a + n; ++i) {
      foa[i] = f;
      if (species[i] < 0 || species[i] >= nel)
        throw std::runtime_error("frame " + std::to_string(f) + ": species index out of range");
      if (!std::isfinite(pos[3 * i]) || !std::isfinite(pos[3 * i + 1]) || !std::isfinite(pos[3 * i + 2]))
        throw std::runtime_error("frame " + std::to_string(f) + ": non-finite position");
    }
    a += n;
    astart[f + 1] = (int32_t)a;
  }
  {  // atoms grouped by element for the batched MLP
    std::vector<int32_t> count(nel + 1, 0);
    for (size_t i = 0; i < N; ++i) count[species[i] + 1]++;
    for (int e = 0; e < nel; ++e) count[e + 1] += count[e];
    for (int e = 0; e <= nel; ++e) h->db.elem_start[e] = count[e];
    std::vector<int32_t> fill(count.begin(), count.end() - 1);
    for (size_t i = 0; i < N; ++i) elem_atoms[fill[species[i]]++] = (int32_t)i;
  }
  // Neighbour list: on the device (ta_nlist.hip; cells thinner than the cutoff along a periodic axis
  // included since round 3: one bin there, several images of it), the host builder (ta_neighbor.cpp)
  // only for singular / incomplete cells, cells below ~1 A of height, or TA_HOST_NL=1.
  const auto t_nl = std::chrono::steady_clock::now();
  int n_bins = 0;
  bool device_nl = N > 0 && !(std::getenv("TA_HOST_NL") && std::getenv("TA_HOST_NL")[0] == '1');
  for (int f = 0; f < n_frames && device_nl; ++f) {
    device_nl = ta::nl_make_grid(frames[f], h->r_list, n_bins, grids[f]);
    if (device_nl) n_bins += ta::nl_bins(grids[f]);
    if (n_bins > (1 << 24)) device_nl = false;
  }
  if (!device_nl) std::memset(static_cast<void *>(grids), 0, F * sizeof(ta::NlGrid));
  // one upload for everything but blk_center (which needs the pair counts)
  if (o_blk)  // (small batches: by a kernel reading the page-locked buffer, see staged_copy; sections are 16-byte aligned)
    staged_copy(reinterpret_cast<double *>(h->inbuf.ptr), reinterpret_cast<const double *>(hb), o_blk / sizeof(double),
                false, h->stream);
  char *db_ = h->inbuf.ptr;
  h->db.n_frames = n_frames;
  h->db.pos = reinterpret_cast<double *>(db_ + o_pos);
  h->db.cells = reinterpret_cast<double *>(db_ + o_cells);
  h->d_grids = reinterpret_cast<ta::NlGrid *>(db_ + o_grids);
  h->db.species = reinterpret_cast<int32_t *>(db_ + o_species);
  h->db.frame_of_atom = reinterpret_cast<int32_t *>(db_ + o_foa);
  h->db.atom_start = reinterpret_cast<int32_t *>(db_ + o_astart);
  h->db.elem_atoms = reinterpret_cast<int32_t *>(db_ + o_elem);
  h->db.blk_center = reinterpret_cast<int32_t *>(db_ + o_blk);
  if (device_nl) {
    h->hp.atom_start.assign(astart, astart + F + 1);
    h->hp.frame_of_atom.assign(foa, foa + N);
    build_pairs_on_device(h, N, n_bins);
  } else {
    h->pairs_on_device = false;
    ta::build_pairs(n_frames, frames, nel, h->r_list, h->hp);
  }
  double nl_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_nl).count();
  upload_batch(h);
  const auto t_fill = std::chrono::steady_clock::now();
  const bool list_done = h->pairs_on_device && h->nl_sorted && h->nl_rev_done;  // nothing left but the run packing
  if (h->pairs_on_device && !list_done) {
    fill_pairs_on_device(h);
    // "reverse pair missing" counter, read after the synchronisation below
    staged_copy(reinterpret_cast<double *>(h->stage_out.ptr), reinterpret_cast<const double *>(h->nl_stats_ptr),
                8, true, h->stream);
  }
  // second-generation angular kernels: workgroups own whole centres (<= kCap pairs)
  // second-generation kernels: 1-3 elements for every channel grid; 4 and 5 elements for
  // launches of 2 gammas x 2 zetas (the default grid) only
  bool shapes_ok = h->n_elements <= 3;
  if (h->n_elements == 4 || h->n_elements == 5) {
    shapes_ok = !h->chunks_v2.empty();
    for (const ChunkPlan &cp : h->chunks_v2) shapes_ok = shapes_ok && cp.ng == 2 && cp.nz == 2;
  }
  h->use_v2 = h->kind == TA_MODEL_SF_MLP && h->sf.angular && shapes_ok &&
              h->hp.nnl_max <= ta::kCapMax && std::getenv("TA_FORCE_V1") == nullptr;
  const int cap = std::max(ta::kCapMin, (h->hp.nnl_max + 63) / 64 * 64);
  h->db.cap = cap;
  {
    int32_t *blk = reinterpret_cast<int32_t *>(hb + o_blk);
    int nb = 0;
    if (h->use_v2 && N) {
      blk[nb++] = 0;
      int32_t load = 0, ncent = 0;
      for (size_t i = 0; i < N; ++i) {
        const int32_t cnt = h->hp.pair_start[i + 1] - h->hp.pair_start[i];
        if (load + cnt > cap || ncent >= ta::kMaxCentersPerBlock) {
          blk[nb++] = (int32_t)i;
          load = 0;
          ncent = 0;
        }
        load += cnt;
        ++ncent;
      }
      blk[nb++] = (int32_t)N;
    }
    h->db.n_blk = nb ? nb - 1 : 0;
    ensure_job_lists(h, (size_t)h->db.n_blk);
    if (nb) {
      if (list_done)  // a kernel on the compute stream reads the page-locked buffer: no DMA hand-over
        staged_copy(reinterpret_cast<double *>(db_ + o_blk), reinterpret_cast<const double *>(blk), ((size_t)nb + 1) / 2,
                    false, h->stream);
      else
        HIP_CHECK(hipMemcpyAsync(db_ + o_blk, blk, (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    }
  }
  if (h->kind == TA_MODEL_SF_MLP) {
    if (!h->use_v2 && ta::g4_lds_bytes(h->hp.nnl_max) > 160 * 1024 && h->sf.angular)
      throw std::domain_error("more than 1150 neighbours per atom exceed the LDS staging buffer");
    size_t need = ta::mlp_all_scratch_doubles(h->mlp, h->n_elements, h->db.elem_start);
    h->mlp_scratch.ensure(need);
  } else if (h->kind == TA_MODEL_GRAP_MLP) {
    h->mlp_scratch.ensure(ta::mlp_all_scratch_doubles(h->mlp, h->n_elements, h->db.elem_start));
    ta::grap_ensure(h->grap, h->db);
  } else {
    ta::eam_ensure(h->eam, h->db);
  }
  // what ta_update_positions needs: the geometry this list was built for and the frames' shapes
  // (host-side copies, made while the device still works on the reverse index)
  h->o_pos = o_pos;
  h->o_cells = o_cells;
  h->o_species = o_species;
  h->ref_pos.assign(pos, pos + 3 * N);
  h->ref_cells.assign(cells, cells + 9 * F);
  h->keep_species.assign(species, species + N);
  h->keep_natoms.resize(F);
  h->keep_pbc.resize(3 * F);
  for (size_t f = 0; f < F; ++f) {
    h->keep_natoms[f] = frames[f].n_atoms;
    for (int a3 = 0; a3 < 3; ++a3) h->keep_pbc[3 * f + a3] = frames[f].pbc[a3] ? 1 : 0;
  }
  if (list_done) {
    // one-pass builder: the list was complete (and checked) at the builder's own wait; the run packing is
    // in flight out of stage_in, whose next writer waits for the stream (upload_pending)
    h->upload_pending = true;
  } else {
    wait_stream(h->stream);  // staging buffers are reused by the next call
    h->upload_pending = false;
  }
  if (h->pairs_on_device && !list_done) {
    nl_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fill).count();
    if (reinterpret_cast<const int32_t *>(h->stage_out.ptr)[6] != 0)
      throw std::runtime_error("neighbour list is not symmetric (reverse pair missing)");
  }
  if (filter_applies(h)) apply_filter(h);
  if (h->eam) ta::eam_set_list_cutoff(h->eam, (h->skin > 0.0 && !h->filtered) ? h->rmax : 0.0);
  ++h->n_list_builds;
  h->have_batch = true;
  if (info) {
    info->n_frames = n_frames;
    info->n_atoms = h->hp.n_atoms;
    info->n_pairs = h->hp.n_pairs;
    info->n_triples = h->hp.n_triples;
    info->nnl_max = h->hp.nnl_max;
    info->descriptor_dim = (h->kind == TA_MODEL_SF_MLP || h->kind == TA_MODEL_GRAP_MLP) ? h->sf.ndim : 0;
    info->nl_on_device = h->pairs_on_device ? 1 : 0;
    info->reserved_ = 0;
    info->nl_ms = nl_ms;
    info->set_frames_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
}
}  // namespace

extern "C" {

int ta_set_frames(ta_handle h, int32_t n_frames, const ta_frame *frames, ta_batch_info *info) {
  if (!h) return TA_ERR_INVALID;
  if (n_frames < 0 || (n_frames > 0 && !frames)) return fail(h, TA_ERR_INVALID, "bad frames argument");
  return guarded(h, [&]() { set_frames_impl(h, n_frames, frames, info); });
}

int ta_set_skin(ta_handle h, double skin) {
  if (!h) return TA_ERR_INVALID;
  if (!(skin >= 0.0) || !std::isfinite(skin)) return fail(h, TA_ERR_INVALID, "skin must be a finite length >= 0");
  if (skin != h->skin) {
    h->skin = skin;
    h->have_batch = false;  // the resident list was built for another cutoff
    h->descriptors_valid = false;
  }
  return TA_OK;
}

int ta_update_positions(ta_handle h, const double *positions, const double *cells, int32_t *rebuilt) {
  if (!h || !positions) return fail(h, TA_ERR_INVALID, "null argument");
  if (h->ref_pos.empty() && h->keep_natoms.empty())
    return fail(h, TA_ERR_INVALID, "ta_update_positions called before ta_set_frames");
  return guarded(h, [&]() {
    const size_t N = h->keep_species.size(), F = h->keep_natoms.size();
    // The list stays valid while every atom is within skin / 2 of where it was when the list was
    // built and the cells are the same: two atoms then approach each other by less than skin, so
    // every pair inside rmax is still in the list (pairs beyond rmax contribute nothing: cutoff
    // functions, or the explicit test of the EAM / ADP kernels). Checked on the host: the
    // positions pass through here anyway and no device round trip is needed.
    bool keep = h->have_batch && h->skin > 0.0;
    if (keep && cells) keep = std::memcmp(cells, h->ref_cells.data(), 9 * F * sizeof(double)) == 0;
    if (keep) {
      // Optimistic order: the coordinates go up and the exact list of the step is extracted (device work)
      // BEFORE the host has checked the displacements, which it then does while the device is busy. A list
      // that turns out stale is rebuilt below; what was launched for it is overwritten by the rebuild.
      if (h->upload_pending) {  // the staging buffer must be free again
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->upload_pending = false;
      }
      double *stage = reinterpret_cast<double *>(h->stage_in.ptr + h->o_pos);
      std::memcpy(stage, positions, 3 * N * sizeof(double));
      staged_copy(h->db.pos, stage, 3 * N, false, h->stream);
      h->upload_pending = true;
      if (h->filtered) apply_filter(h);  // the exact list of the new positions, on the device
      h->descriptors_valid = false;
      h->jvp_valid = false;
      const double lim2 = 0.25 * h->skin * h->skin;
      const double *ref = h->ref_pos.data();
      for (size_t i = 0; i < N; ++i) {
        const double dx = positions[3 * i] - ref[3 * i], dy = positions[3 * i + 1] - ref[3 * i + 1],
                     dz = positions[3 * i + 2] - ref[3 * i + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (!(d2 <= lim2)) {  // (a NaN fails the comparison too and is reported by the rebuild)
          keep = false;
          break;
        }
      }
    }
    if (rebuilt) *rebuilt = keep ? 0 : 1;
    if (keep) {
      ++h->n_list_reuses;
      return;
    }
    std::vector<ta_frame> frames(F);
    const std::vector<int32_t> species(h->keep_species), pbc(h->keep_pbc), natoms(h->keep_natoms);
    const std::vector<double> old_cells(h->ref_cells);
    size_t a = 0;
    for (size_t f = 0; f < F; ++f) {
      frames[f].n_atoms = natoms[f];
      frames[f].species = species.data() + a;
      frames[f].positions = positions + 3 * a;
      frames[f].cell = (cells ? cells : old_cells.data()) + 9 * f;
      frames[f].pbc = pbc.data() + 3 * f;
      a += (size_t)natoms[f];
    }
    set_frames_impl(h, (int32_t)F, frames.data(), nullptr);
  });
}

int ta_list_stats(ta_handle h, int64_t *n_builds, int64_t *n_reuses) {
  if (!h) return TA_ERR_INVALID;
  if (n_builds) *n_builds = h->n_list_builds;
  if (n_reuses) *n_reuses = h->n_list_reuses;
  return TA_OK;
}

int ta_list_sizes(ta_handle h, int64_t *n_pairs, int64_t *n_triples, int32_t *nnl_max) {
  if (!h) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  if (n_pairs) *n_pairs = h->hp.n_pairs;
  if (n_triples) *n_triples = h->hp.n_triples;
  if (nnl_max) *nnl_max = h->hp.nnl_max;
  return TA_OK;
}

int ta_compute(ta_handle h, uint32_t want) {
  if (!h) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "ta_compute called before ta_set_frames");
  return guarded(h, [&]() { compute_impl(h, want, false, nullptr); });
}

int ta_synchronize(ta_handle h) {
  if (!h) return TA_ERR_INVALID;
  return guarded(h, [&]() { HIP_CHECK(hipStreamSynchronize(h->stream)); });
}

namespace {
// one download of the span of [energy F | virial 9F | atomic N | forces 3N] that is asked for into the
// page-locked staging buffer; returns the pointer that indexes it like h->results (null: nothing asked)
const double *results_to_stage(ta_context *h, bool energy, bool forces, bool virial, bool atomic, double *descriptors) {
  const size_t N = (size_t)h->db.n_atoms, F = (size_t)h->db.n_frames;
  hipStream_t s = h->stream;
  const bool have_forces = (h->last_want & (TA_WANT_FORCES | TA_WANT_VIRIAL)) != 0;
  if ((forces || virial) && !have_forces)
    throw std::invalid_argument("forces / virial requested but the last ta_compute did not produce them");
  if (descriptors && h->kind != TA_MODEL_SF_MLP && h->kind != TA_MODEL_GRAP_MLP)
    throw std::invalid_argument("descriptors are only defined for symmetry-function and GRAP models");
  size_t lo = (size_t)-1, hi = 0;
  auto need = [&](bool on, size_t off, size_t n) {
    if (!on || n == 0) return;
    lo = std::min(lo, off);
    hi = std::max(hi, off + n);
  };
  need(energy, 0, F);
  need(virial, F, 9 * F);
  need(atomic, 10 * F, N);
  need(forces, 10 * F + N, 3 * N);
  const double *stage = nullptr;
  const bool fv = (h->mirror_want & (TA_WANT_FORCES | TA_WANT_VIRIAL)) != 0;
  if (hi > lo && h->mirror_want && (!(forces || virial) || fv)) {
    stage = reinterpret_cast<const double *>(h->stage_out.ptr);  // already there (frame_reduce_kernel's mirror)
  } else if (hi > lo) {
    h->mirror_want = 0;
    h->stage_out.ensure((hi - lo) * sizeof(double));
    stage = reinterpret_cast<const double *>(h->stage_out.ptr) - lo;
    staged_copy(reinterpret_cast<double *>(h->stage_out.ptr), h->results.ptr + lo, hi - lo, true, s);
  }
  if (descriptors && N)
    HIP_CHECK(hipMemcpyAsync(descriptors, h->db.G, N * h->sf.ndim * sizeof(double), hipMemcpyDeviceToHost, s));
  wait_stream(s);
  h->upload_pending = false;
  return stage;
}
}  // namespace

int ta_get_results(ta_handle h, double *energy, double *forces, double *virial, double *atomic,
                   double *descriptors) {
  if (!h) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  return guarded(h, [&]() {
    const size_t N = (size_t)h->db.n_atoms, F = (size_t)h->db.n_frames;
    const double *stage = results_to_stage(h, energy != nullptr, forces != nullptr, virial != nullptr,
                                           atomic != nullptr, descriptors);
    if (energy && F) std::memcpy(energy, stage, F * sizeof(double));
    if (virial && F) std::memcpy(virial, stage + F, 9 * F * sizeof(double));
    if (atomic && N) std::memcpy(atomic, stage + 10 * F, N * sizeof(double));
    if (forces && N) std::memcpy(forces, stage + 10 * F + N, 3 * N * sizeof(double));
  });
}

int ta_view_results(ta_handle h, uint32_t want, const double **energy, const double **forces,
                    const double **virial, const double **atomic) {
  if (!h) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  return guarded(h, [&]() {
    const size_t N = (size_t)h->db.n_atoms, F = (size_t)h->db.n_frames;
    const bool fv = (want & (TA_WANT_FORCES | TA_WANT_VIRIAL)) != 0;
    const double *stage = results_to_stage(h, energy != nullptr, forces && fv, virial && fv,
                                           atomic && (want & TA_WANT_ATOMIC), nullptr);
    if (energy) *energy = (stage && F) ? stage : nullptr;
    if (virial) *virial = (stage && fv && F) ? stage + F : nullptr;
    if (atomic) *atomic = (stage && (want & TA_WANT_ATOMIC) && N) ? stage + 10 * F : nullptr;
    if (forces) *forces = (stage && fv && N) ? stage + 10 * F + N : nullptr;
  });
}

int ta_step(ta_handle h, const double *positions, const double *cells, uint32_t want, double *energy,
            double *forces, double *virial, double *atomic, int32_t *rebuilt) {
  int rc = ta_update_positions(h, positions, cells, rebuilt);
  if (rc != TA_OK) return rc;
  h->mirror_next = true;
  rc = ta_compute(h, want);
  h->mirror_next = false;
  if (rc != TA_OK) return rc;
  return ta_get_results(h, energy, forces, virial, atomic, nullptr);
}

int ta_step_view(ta_handle h, const double *positions, const double *cells, uint32_t want, const double **energy,
                 const double **forces, const double **virial, const double **atomic, int32_t *rebuilt) {
  int rc = ta_update_positions(h, positions, cells, rebuilt);
  if (rc != TA_OK) return rc;
  h->mirror_next = true;
  rc = ta_compute(h, want);
  h->mirror_next = false;
  if (rc != TA_OK) return rc;
  return ta_view_results(h, want, energy, forces, virial, atomic);
}

int ta_eval(ta_handle h, int32_t n_frames, const ta_frame *frames, uint32_t want, double *energy,
            double *forces, double *virial, double *atomic) {
  int rc = ta_set_frames(h, n_frames, frames, nullptr);
  if (rc != TA_OK) return rc;
  rc = ta_compute(h, want);
  if (rc != TA_OK) return rc;
  return ta_get_results(h, energy, forces, virial, atomic, nullptr);
}

int ta_time_compute(ta_handle h, uint32_t want, int32_t warmup, int32_t steps, double *total_ms,
                    double *kernel_ms) {
  if (!h) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  if (steps < 1 || warmup < 0) return fail(h, TA_ERR_INVALID, "steps must be >= 1, warmup >= 0");
  return guarded(h, [&]() {
    hipStream_t s = h->stream;
    for (int k = 0; k < warmup; ++k) compute_impl(h, want, false, nullptr);
    HIP_CHECK(hipStreamSynchronize(s));
    hipEvent_t e0 = h->ev[2 * TA_N_KERNEL_SLOTS], e1 = h->ev[2 * TA_N_KERNEL_SLOTS + 1];
    HIP_CHECK(hipEventRecord(e0, s));
    for (int k = 0; k < steps; ++k) compute_impl(h, want, false, nullptr);
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (total_ms) *total_ms = ms;
    if (kernel_ms) {
      double acc[TA_N_KERNEL_SLOTS] = {0};
      for (int k = 0; k < steps; ++k) compute_impl(h, want, true, acc);
      for (int k = 0; k < TA_N_KERNEL_SLOTS; ++k) kernel_ms[k] = acc[k] / steps;
    }
  });
}

int ta_batch_energy_device_ptr(ta_handle h, void **dptr) {
  if (!h || !dptr) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  *dptr = h->db.batch_energy;
  return TA_OK;
}

namespace {
// 16 B per lane and four independent loads in flight per lane before the first store (a one-load
// grid-stride loop reached 4.8 TB/s of the 6.3 TB/s the microarch guide measures for a float4 copy)
__global__ __launch_bounds__(256) void hbm_copy_kernel(const double2 *__restrict__ src,
                                                       double2 *__restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; k + 3 * stride < n; k += 4 * stride) {
    const double2 a = src[k], b = src[k + stride], c = src[k + 2 * stride], d = src[k + 3 * stride];
    dst[k] = a;
    dst[k + stride] = b;
    dst[k + 2 * stride] = c;
    dst[k + 3 * stride] = d;
  }
  for (; k < n; k += stride) dst[k] = src[k];
}
// the same with non-temporal loads and stores (streamed once: nothing to keep in L2 / MALL)
__global__ __launch_bounds__(256) void hbm_copy_nt_kernel(const double2 *__restrict__ src,
                                                          double2 *__restrict__ dst, size_t n) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 *s = reinterpret_cast<const d2 *>(src);
  d2 *d = reinterpret_cast<d2 *>(dst);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; k + 3 * stride < n; k += 4 * stride) {
    const d2 a = __builtin_nontemporal_load(s + k), b = __builtin_nontemporal_load(s + k + stride),
             c = __builtin_nontemporal_load(s + k + 2 * stride), e = __builtin_nontemporal_load(s + k + 3 * stride);
    __builtin_nontemporal_store(a, d + k);
    __builtin_nontemporal_store(b, d + k + stride);
    __builtin_nontemporal_store(c, d + k + 2 * stride);
    __builtin_nontemporal_store(e, d + k + 3 * stride);
  }
  for (; k < n; k += stride) d[k] = s[k];
}
}  // namespace

namespace {
// unordered {j, k} of one centre with r_ij, r_ik and r_jk all below acut: the triples whose G4 term
// is not identically zero (one wavefront per centre, lane a walks b > a; reads the pair records)
__global__ __launch_bounds__(256) void count_triples_kernel(ta::DeviceBatch b, double ac2, double eps,
                                                            unsigned long long *out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  const int p0 = b.pair_start[i], p1 = ta::pair_stop_of(b, i);
  unsigned long long n = 0;
  for (int pa = p0 + lane; pa < p1; pa += 64) {
    const double2 *ra = ta::pair_geom(b, (size_t)pa);
    const double ax = ra[0].x, ay = ra[0].y, az = ra[1].x;
    if (!(ra[1].y < ac2)) continue;
    for (int pb = pa + 1; pb < p1; ++pb) {
      const double2 *rb = ta::pair_geom(b, (size_t)pb);
      const double ex = rb[0].x - ax, ey = rb[0].y - ay, ez = rb[1].x - az;
      n += (rb[1].y < ac2 && ex * ex + ey * ey + ez * ez + eps < ac2) ? 1ull : 0ull;
    }
  }
  for (int off = 32; off; off >>= 1) n += __shfl_xor(n, off);
  if (lane == 0 && n) atomicAdd(out, n);
}
}  // namespace

int ta_count_contributing_triples(ta_handle h, int64_t *n_contributing) {
  if (!h || !n_contributing) return fail(h, TA_ERR_INVALID, "null argument");
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  if (h->kind != TA_MODEL_SF_MLP || !h->sf.angular)
    return fail(h, TA_ERR_INVALID, "only symmetry-function models with angular terms have triples");
  return guarded(h, [&]() {
    compute_impl(h, TA_WANT_ENERGY, false, nullptr);  // fills the pair records
    h->nl_stats.ensure(8);
    hipStream_t s = h->stream;
    HIP_CHECK(hipMemsetAsync(h->nl_stats.ptr, 0, sizeof(unsigned long long), s));
    const int64_t N = h->db.n_atoms;
    if (N)
      hipLaunchKernelGGL(count_triples_kernel, dim3((unsigned)((N * 64 + 255) / 256)), dim3(256), 0, s, h->db,
                         h->sf.acut * h->sf.acut, h->sf.eps, h->nl_stats.ptr);
    HIP_CHECK(hipGetLastError());
    unsigned long long v = 0;
    HIP_CHECK(hipMemcpyAsync(&v, h->nl_stats.ptr, sizeof(v), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    *n_contributing = (int64_t)v;
  });
}

int ta_measure_hbm_copy(ta_handle h, int64_t bytes, int32_t reps, double *gbs) {
  if (!h || !gbs || bytes < 16 || reps < 1) return fail(h, TA_ERR_INVALID, "bad argument");
  return guarded(h, [&]() {
    const size_t n = (size_t)bytes / 16;
    double2 *src = nullptr, *dst = nullptr;
    HIP_CHECK(hipMalloc((void **)&src, n * 16));
    if (hipMalloc((void **)&dst, n * 16) != hipSuccess) {
      (void)hipFree(src);
      throw std::bad_alloc();
    }
    hipStream_t s = h->stream;
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipMemsetAsync(src, 0, n * 16, s));
    static const int per_cu = std::getenv("TA_COPY_WG_PER_CU") ? std::atoi(std::getenv("TA_COPY_WG_PER_CU")) : 8;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)256 * std::max(1, per_cu));
    // three ways to copy, the fastest counts (measured on this pool: 4.8-5.2, 5.0-5.5 and 5.0-5.5 TB/s
    // read + written; the microarch guide quotes 6.29 for a float4 copy): the grid-stride kernel, the
    // runtime's device-to-device copy, the kernel with non-temporal loads and stores
    static const int only = std::getenv("TA_COPY_MODE") ? std::atoi(std::getenv("TA_COPY_MODE")) : -1;
    double best = 0.0;
    for (int mode = 0; mode < 3; ++mode) {
      if (only >= 0 && mode != only) continue;
      auto one = [&]() {
        if (mode == 1) (void)hipMemcpyAsync(dst, src, n * 16, hipMemcpyDeviceToDevice, s);
        else if (mode == 2) hipLaunchKernelGGL(hbm_copy_nt_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n);
        else hipLaunchKernelGGL(hbm_copy_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n);
      };
      for (int k = 0; k < 2; ++k) one();
      HIP_CHECK(hipEventRecord(e0, s));
      for (int k = 0; k < reps; ++k) one();
      HIP_CHECK(hipEventRecord(e1, s));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms > 0.f) best = std::max(best, 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(dst);
    *gbs = best;
  });
}

int ta_set_stream(ta_handle h, void *stream) {
  if (!h) return TA_ERR_INVALID;
  return guarded(h, [&]() {
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->upload_pending = false;
    h->stream = stream ? (hipStream_t)stream : h->own_stream;
  });
}

int ta_copy_batch_energy(ta_handle h, void *dst_device) {
  if (!h || !dst_device) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  return guarded(h, [&]() {
    HIP_CHECK(hipMemcpyAsync(dst_device, h->db.batch_energy, sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
  });
}

int ta_param_count(ta_handle h, int64_t *n_params) {
  if (!h || !n_params) return TA_ERR_INVALID;
  if (h->eam) {  // the nn functions of an EAM / ADP model, slot after slot
    *n_params = ta::eam_param_count(h->eam);
    return TA_OK;
  }
  if (h->kind != TA_MODEL_SF_MLP && h->kind != TA_MODEL_GRAP_MLP)
    return fail(h, TA_ERR_INVALID, "the model has no MLP");
  int64_t n = 0;
  for (int e = 0; e < h->n_elements; ++e) n += ta::mlp_param_count(h->mlp[e]);
  *n_params = n;
  return TA_OK;
}

int ta_update_weights(ta_handle h, const double *weights, int64_t n_weights) {
  if (!h || !weights) return TA_ERR_INVALID;
  if (h->eam)
    return guarded(h, [&]() {
      HIP_CHECK(hipStreamSynchronize(h->stream));  // nothing may still read the old weights
      ta::eam_update_weights(h->eam, weights, n_weights);
    });
  if (h->kind != TA_MODEL_SF_MLP && h->kind != TA_MODEL_GRAP_MLP)
    return fail(h, TA_ERR_INVALID, "the model has no MLP");
  return guarded(h, [&]() {
    int64_t want = 0;
    for (int e = 0; e < h->n_elements; ++e) want += ta::mlp_param_count(h->mlp[e]);
    if (n_weights != want)
      throw std::invalid_argument("ta_update_weights: expected " + std::to_string(want) + " values");
    HIP_CHECK(hipStreamSynchronize(h->stream));  // nothing may still read the old weights
    const double *src = weights;
    for (int e = 0; e < h->n_elements; ++e) {
      ta::MlpDev &md = h->mlp[e];
      for (int l = 0; l < md.n_layers; ++l) {
        ta::MlpLayerDev &ly = md.layer[l];
        std::vector<double> w((size_t)ly.kp * ly.np, 0.0), wt((size_t)ly.np * ly.kp, 0.0), bb(ly.np, 0.0);
        for (int k = 0; k < ly.k; ++k)
          for (int n = 0; n < ly.n; ++n) {
            const double v = src[(size_t)k * ly.n + n];
            w[(size_t)k * ly.np + n] = v;
            wt[(size_t)n * ly.kp + k] = v;
          }
        src += (size_t)ly.k * ly.n;
        for (int n = 0; n < ly.n; ++n) bb[n] = src[n];
        src += ly.n;
        HIP_CHECK(hipMemcpy(ly.w, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(ly.wt, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(ly.b, bb.data(), bb.size() * sizeof(double), hipMemcpyHostToDevice));
      }
    }
  });
}

int ta_energy_gradient(ta_handle h, const double *frame_coeff, double *grad, int64_t n_grad) {
  if (!h || !frame_coeff || !grad) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  if (h->eam)
    return guarded(h, [&]() {
      const int64_t total = ta::eam_param_count(h->eam);
      if (n_grad != total)
        throw std::invalid_argument("ta_energy_gradient: expected room for " + std::to_string(total) + " values");
      if (total == 0) return;
      // weight gradients differentiate the networks themselves: from here on this handle evaluates its
      // nn pair functions exactly, not through their tables (eam_set_nn_tables)
      if (h->filtered)
        throw std::domain_error("ta_energy_gradient: not available on a skin-filtered batch; "
                                "ta_set_skin(h, 0) and ta_set_frames first");
      if (ta::eam_nn_tables_on(h->eam)) HIP_CHECK(hipStreamSynchronize(h->stream));
      ta::eam_mark_trained(h->eam);
      ta::eam_ensure(h->eam, h->db);  // the per-pair columns of the exact evaluation
      // every function depends on the weights: the forward pass runs again (rho, F', moments)
      compute_impl(h, TA_WANT_ENERGY, false, nullptr);
      hipStream_t s = h->stream;
      const size_t F = (size_t)h->db.n_frames;
      h->train_grad.ensure((size_t)total + 8);
      h->train_coeff.ensure(F + 8);
      if (F) HIP_CHECK(hipMemcpyAsync(h->train_coeff.ptr, frame_coeff, F * sizeof(double), hipMemcpyHostToDevice, s));
      ta::eam_energy_gradient(h->eam, h->db, h->train_coeff.ptr, h->train_grad.ptr, s);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(grad, h->train_grad.ptr, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    });
  if (h->kind != TA_MODEL_SF_MLP && h->kind != TA_MODEL_GRAP_MLP)
    return fail(h, TA_ERR_INVALID, "the model has no MLP");
  return guarded(h, [&]() {
    int64_t total = 0;
    for (int e = 0; e < h->n_elements; ++e) total += ta::mlp_param_count(h->mlp[e]);
    if (n_grad != total)
      throw std::invalid_argument("ta_energy_gradient: expected room for " + std::to_string(total) + " values");
    // descriptors do not depend on the weights: computed once per resident batch
    if (!h->descriptors_valid) compute_impl(h, TA_WANT_ENERGY, false, nullptr);
    hipStream_t s = h->stream;
    const size_t F = (size_t)h->db.n_frames;
    size_t scratch = 0, partial = 0;
    for (int e = 0; e < h->n_elements; ++e) {
      const int n_el = h->db.elem_start[e + 1] - h->db.elem_start[e];
      scratch = std::max(scratch, ta::mlp_grad_scratch_doubles(h->mlp[e], n_el));
      partial = std::max(partial, ta::mlp_grad_partial_doubles(h->mlp[e], n_el));
    }
    h->train_scratch.ensure(scratch + 8);
    h->train_partial.ensure(partial + 8);
    h->train_grad.ensure((size_t)total + 8);
    h->train_coeff.ensure(F + 8);
    if (F) HIP_CHECK(hipMemcpyAsync(h->train_coeff.ptr, frame_coeff, F * sizeof(double), hipMemcpyHostToDevice, s));
    size_t off = 0;
    for (int e = 0; e < h->n_elements; ++e) {
      const int n_el = h->db.elem_start[e + 1] - h->db.elem_start[e];
      ta::launch_mlp_grad(h->mlp[e], h->activation, h->sf.ndim, h->db.elem_atoms + h->db.elem_start[e], n_el,
                          h->db, h->train_coeff.ptr, h->train_scratch.ptr, h->train_partial.ptr,
                          h->train_grad.ptr + off, s);
      off += (size_t)ta::mlp_param_count(h->mlp[e]);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(grad, h->train_grad.ptr, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

namespace {
// the dE/dD launches of compute_impl alone, on the dE/dG that is in `db.dEdG` (the forward pass of
// this batch has run: pair records, candidate masks and, for GRAP, the moments are resident)
void backward_only(ta_context *h) {
  using namespace ta;
  const DeviceBatch &db = h->db;
  hipStream_t s = h->stream;
  if (h->kind == TA_MODEL_SF_MLP) {
    if (h->sf.angular) {
      bool first = true;
      if (h->use_v2)
        for (const ChunkPlan &cp : h->chunks_v2) {
          launch_backward_v2(h->sf, cp.ch, cp.ng, cp.nz, first, db, s);
          first = false;
        }
      else
        for (const ChunkPlan &cp : h->chunks) {
          launch_backward(h->sf, cp.ch, cp.nb, cp.ng, cp.nz, first, false, db, s);
          first = false;
        }
    } else {
      AngChunk dummy;
      std::memset(&dummy, 0, sizeof(dummy));
      launch_backward(h->sf, dummy, 1, 1, 1, true, true, db, s);
    }
  } else {
    launch_grap_backward(h->grap, db, s);
  }
}
// J[c][p] = dG_{i(p), c} / dD_p of the resident batch: the backward kernels with a one-hot dE/dG, once per
// channel and once per batch (ta_loss_gradient, ta_hessian_vectors of the descriptor models)
void ensure_pair_jacobians(ta_context *h) {
  hipStream_t s = h->stream;
  const size_t N = (size_t)h->db.n_atoms, P = (size_t)h->db.n_pairs;
  const int D = h->sf.ndim;
  if (!h->descriptors_valid || !h->jvp_valid)
    compute_impl(h, TA_WANT_ENERGY, false, nullptr);  // pair records, masks, moments, descriptors
  if (h->jvp_valid) return;
  h->jvp_J.ensure((size_t)D * 4 * P + 8);
  for (int c = 0; c < D; ++c) {
    ta::launch_one_hot(h->db.dEdG, (int64_t)N, D, c, s);
    backward_only(h);
    if (P)
      HIP_CHECK(hipMemcpyAsync(h->jvp_J.ptr + (size_t)c * 4 * P, h->db.g, 4 * P * sizeof(double),
                               hipMemcpyDeviceToDevice, s));
  }
  HIP_CHECK(hipGetLastError());
  h->jvp_valid = true;
}
}  // namespace

int ta_loss_gradient(ta_handle h, const double *frame_coeff, const double *dR, const double *dh, double *grad,
                     int64_t n_grad, double *dG_out) {
  if (!h || !grad) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  if (!dR && !dh) {
    if (!frame_coeff) return fail(h, TA_ERR_INVALID, "nothing to differentiate");
    return ta_energy_gradient(h, frame_coeff, grad, n_grad);
  }
  if (h->eam) {
    // nn functions of an EAM / ADP model (round 3): one second-order pass per network, ta_eam.hip::eam_loss_gradient
    if (!ta::eam_loss_gradient_supported(h->eam))
      return fail(h, TA_ERR_UNSUPPORTED, "ta_loss_gradient: the analytic force / stress term covers EAM / ADP models "
                                         "whose analytic parts are of the Zjw04 family or tabulated");
    if (h->filtered)
      return fail(h, TA_ERR_UNSUPPORTED, "ta_loss_gradient: not available on a skin-filtered batch; "
                                         "ta_set_skin(h, 0) and ta_set_frames first");
    return guarded(h, [&]() {
      const int64_t total = ta::eam_param_count(h->eam);
      if (n_grad != total)
        throw std::invalid_argument("ta_loss_gradient: expected room for " + std::to_string(total) + " values");
      if (dG_out) throw std::invalid_argument("ta_loss_gradient: dG_out is for the descriptor models");
      if (total == 0) return;
      if (ta::eam_nn_tables_on(h->eam)) HIP_CHECK(hipStreamSynchronize(h->stream));
      ta::eam_mark_trained(h->eam);  // exact networks from here on (see ta_energy_gradient)
      ta::eam_ensure(h->eam, h->db);
      compute_impl(h, TA_WANT_ENERGY, false, nullptr);  // rho, F', per-pair columns with the current weights
      hipStream_t s = h->stream;
      const size_t N = (size_t)h->db.n_atoms, F = (size_t)h->db.n_frames;
      h->train_grad.ensure((size_t)total + 8);
      h->train_coeff.ensure(F + 8);
      h->tan_dir.ensure(3 * N + 9 * F + 8);
      double *d_dR = h->tan_dir.ptr, *d_dh = h->tan_dir.ptr + 3 * N;
      if (dR) HIP_CHECK(hipMemcpyAsync(d_dR, dR, 3 * N * sizeof(double), hipMemcpyHostToDevice, s));
      if (dh) HIP_CHECK(hipMemcpyAsync(d_dh, dh, 9 * F * sizeof(double), hipMemcpyHostToDevice, s));
      if (frame_coeff && F)
        HIP_CHECK(hipMemcpyAsync(h->train_coeff.ptr, frame_coeff, F * sizeof(double), hipMemcpyHostToDevice, s));
      ta::eam_loss_gradient(h->eam, h->db, frame_coeff ? h->train_coeff.ptr : nullptr, dR ? d_dR : nullptr,
                            dh ? d_dh : nullptr, h->train_grad.ptr, s);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(grad, h->train_grad.ptr, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    });
  }
  if (h->kind != TA_MODEL_SF_MLP && h->kind != TA_MODEL_GRAP_MLP)
    return fail(h, TA_ERR_UNSUPPORTED, "ta_loss_gradient: the analytic force / stress term exists for the per-atom MLP models");
  // Per-pair buffers of this entry (tangents, Jacobian) are sized and walked by the resident
  // list's pair count. Under a Verlet skin the kernels run on the exact list extracted from it
  // (apply_filter: only the first pair_start[N] entries of the ex_* arrays are defined), so the two
  // do not match: refuse instead of reading the undefined tail. Training never needs a skin.
  if (h->filtered)
    return fail(h, TA_ERR_UNSUPPORTED, "ta_loss_gradient: not available on a skin-filtered batch; "
                                       "ta_set_skin(h, 0) and ta_set_frames first");
  return guarded(h, [&]() {
    int64_t total = 0;
    for (int e = 0; e < h->n_elements; ++e) total += ta::mlp_param_count(h->mlp[e]);
    if (n_grad != total)
      throw std::invalid_argument("ta_loss_gradient: expected room for " + std::to_string(total) + " values");
    hipStream_t s = h->stream;
    const size_t N = (size_t)h->db.n_atoms, P = (size_t)h->db.n_pairs, F = (size_t)h->db.n_frames;
    const int D = h->sf.ndim;
    ensure_pair_jacobians(h);
    // direction -> pairs -> descriptors
    h->tan_dir.ensure(3 * N + 9 * F + 8);
    h->tan_dD.ensure(4 * P + 8);
    h->tan_dG.ensure(N * (size_t)D + 8);
    double *d_dR = h->tan_dir.ptr, *d_dh = h->tan_dir.ptr + 3 * N;
    if (dR) HIP_CHECK(hipMemcpyAsync(d_dR, dR, 3 * N * sizeof(double), hipMemcpyHostToDevice, s));
    else HIP_CHECK(hipMemsetAsync(d_dR, 0, 3 * N * sizeof(double), s));
    if (dh) HIP_CHECK(hipMemcpyAsync(d_dh, dh, 9 * F * sizeof(double), hipMemcpyHostToDevice, s));
    else HIP_CHECK(hipMemsetAsync(d_dh, 0, 9 * F * sizeof(double), s));
    ta::launch_pair_tangent(h->db, d_dR, d_dh, h->tan_dD.ptr, s);
    ta::launch_descriptor_jvp(h->db, D, h->jvp_J.ptr, h->tan_dD.ptr, h->tan_dG.ptr, s);
    size_t scratch = 0, partial = 0;
    for (int e = 0; e < h->n_elements; ++e) {
      const int n_el = h->db.elem_start[e + 1] - h->db.elem_start[e];
      scratch = std::max(scratch, ta::mlp_grad2_scratch_doubles(h->mlp[e], n_el));
      partial = std::max(partial, ta::mlp_grad_partial_doubles(h->mlp[e], n_el));
    }
    h->train_scratch.ensure(scratch + 8);
    h->train_partial.ensure(partial + 8);
    h->train_grad.ensure((size_t)total + 8);
    h->train_coeff.ensure(F + 8);
    if (frame_coeff && F)
      HIP_CHECK(hipMemcpyAsync(h->train_coeff.ptr, frame_coeff, F * sizeof(double), hipMemcpyHostToDevice, s));
    size_t off = 0;
    for (int e = 0; e < h->n_elements; ++e) {
      const int n_el = h->db.elem_start[e + 1] - h->db.elem_start[e];
      ta::launch_mlp_grad2(h->mlp[e], h->activation, D, h->db.elem_atoms + h->db.elem_start[e], n_el, h->db,
                           h->tan_dG.ptr, frame_coeff ? h->train_coeff.ptr : nullptr, h->train_scratch.ptr,
                           h->train_partial.ptr, h->train_grad.ptr + off, s);
      off += (size_t)ta::mlp_param_count(h->mlp[e]);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(grad, h->train_grad.ptr, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, s));
    if (dG_out && N)
      HIP_CHECK(hipMemcpyAsync(dG_out, h->tan_dG.ptr, N * (size_t)D * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

int ta_constant_count(ta_handle h, int64_t *n_constants) {
  if (!h || !n_constants) return TA_ERR_INVALID;
  if (!h->eam) return fail(h, TA_ERR_INVALID, "the model has no empirical potential");
  *n_constants = ta::eam_constant_count(h->eam);
  return TA_OK;
}

int ta_get_constants(ta_handle h, double *constants, int64_t n_constants) {
  if (!h || !constants) return TA_ERR_INVALID;
  if (!h->eam) return fail(h, TA_ERR_INVALID, "the model has no empirical potential");
  if (n_constants != ta::eam_constant_count(h->eam)) return fail(h, TA_ERR_INVALID, "ta_get_constants: wrong length");
  ta::eam_get_constants(h->eam, constants);
  return TA_OK;
}

int ta_update_constants(ta_handle h, const double *constants, int64_t n_constants) {
  if (!h || !constants) return TA_ERR_INVALID;
  if (!h->eam) return fail(h, TA_ERR_INVALID, "the model has no empirical potential");
  return guarded(h, [&]() {
    HIP_CHECK(hipStreamSynchronize(h->stream));
    ta::eam_update_constants(h->eam, constants, n_constants);
  });
}

int ta_constant_gradient(ta_handle h, const double *frame_coeff, const double *dR, const double *dh, double *grad,
                         int64_t n_grad) {
  if (!h || !grad) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  if (!h->eam) return fail(h, TA_ERR_INVALID, "the model has no empirical potential");
  if (!frame_coeff && !dR && !dh) return fail(h, TA_ERR_INVALID, "nothing to differentiate");
  if (h->filtered)  // see ta_loss_gradient
    return fail(h, TA_ERR_UNSUPPORTED, "ta_constant_gradient: not available on a skin-filtered batch; "
                                       "ta_set_skin(h, 0) and ta_set_frames first");
  return guarded(h, [&]() {
    const int64_t total = ta::eam_constant_count(h->eam);
    if (n_grad != total)
      throw std::invalid_argument("ta_constant_gradient: expected room for " + std::to_string(total) + " values");
    // positions, cells and the list's cutoff test as the inference kernels see them; models with nn functions
    // beside the analytic ones: the networks evaluated exactly (per-pair columns, F'), as for their own gradient
    if (ta::eam_param_count(h->eam) > 0) {
      if (ta::eam_nn_tables_on(h->eam)) HIP_CHECK(hipStreamSynchronize(h->stream));
      ta::eam_mark_trained(h->eam);
      ta::eam_ensure(h->eam, h->db);
    }
    compute_impl(h, TA_WANT_ENERGY, false, nullptr);
    hipStream_t s = h->stream;
    const size_t N = (size_t)h->db.n_atoms, F = (size_t)h->db.n_frames;
    h->train_grad.ensure((size_t)total + 8);
    h->train_coeff.ensure(F + 8);
    if (frame_coeff && F)
      HIP_CHECK(hipMemcpyAsync(h->train_coeff.ptr, frame_coeff, F * sizeof(double), hipMemcpyHostToDevice, s));
    double *d_dR = nullptr, *d_dh = nullptr;
    if (dR || dh) {
      h->tan_dir.ensure(3 * N + 9 * F + 8);
      d_dR = h->tan_dir.ptr;
      d_dh = h->tan_dir.ptr + 3 * N;
      if (dR) HIP_CHECK(hipMemcpyAsync(d_dR, dR, 3 * N * sizeof(double), hipMemcpyHostToDevice, s));
      else HIP_CHECK(hipMemsetAsync(d_dR, 0, 3 * N * sizeof(double), s));
      if (dh) HIP_CHECK(hipMemcpyAsync(d_dh, dh, 9 * F * sizeof(double), hipMemcpyHostToDevice, s));
      else HIP_CHECK(hipMemsetAsync(d_dh, 0, 9 * F * sizeof(double), s));
    }
    ta::eam_constant_gradient(h->eam, h->db, frame_coeff ? h->train_coeff.ptr : nullptr, d_dR, d_dh,
                              h->train_grad.ptr, s);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(grad, h->train_grad.ptr, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  });
}

int ta_set_batch_energy_target(ta_handle h, void *dst_device) {
  if (!h) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  // no synchronisation: only launches made after this call see the new target
  h->db.batch_energy = dst_device ? static_cast<double *>(dst_device) : h->benergy.ptr;
  return TA_OK;
}

int ta_neighbor_list(const ta_frame *frame, int32_t n_elements, double rc, int64_t *n_pairs,
                     int32_t **i, int32_t **j, int32_t **shift, int32_t **rev) {
  if (!frame || !n_pairs || n_elements < 1 || !(rc > 0.0))
    return fail(nullptr, TA_ERR_INVALID, "bad argument");
  try {
    ta::HostPairs hp;
    ta::build_pairs(1, frame, n_elements, rc, hp);
    const size_t P = (size_t)hp.n_pairs;
    *n_pairs = hp.n_pairs;
    auto dup = [&](const std::vector<int32_t> &v) {
      int32_t *p = (int32_t *)std::malloc((v.size() ? v.size() : 1) * sizeof(int32_t));
      if (!p) throw std::bad_alloc();
      if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(int32_t));
      return p;
    };
    (void)P;
    if (i) *i = dup(hp.pair_i);
    if (j) *j = dup(hp.pair_j);
    if (shift) *shift = dup(hp.pair_shift);
    if (rev) *rev = dup(hp.pair_rev);
    return TA_OK;
  } catch (const std::bad_alloc &) {
    return fail(nullptr, TA_ERR_NOMEM, "host allocation failed");
  } catch (const std::exception &e) {
    return fail(nullptr, TA_ERR_INVALID, e.what());
  }
}

void ta_free(void *p) { std::free(p); }

int ta_hessian_vectors(ta_handle h, int32_t n_dir, int32_t first, const double *dR, const double *dh, double *dF,
                       double *dW) {
  if (!h || !dF || n_dir < 0) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  const bool grap_model = h->kind == TA_MODEL_GRAP_MLP;
  const bool sf_model = h->kind == TA_MODEL_SF_MLP || grap_model;  // the descriptor + MLP models
  if (sf_model) {
    if (h->filtered)
      return fail(h, TA_ERR_UNSUPPORTED, "ta_hessian_vectors: not available on a skin-filtered batch; "
                                         "ta_set_skin(h, 0) and ta_set_frames first");
    if (grap_model && !ta::grap_hvp_supported(h->grap))
      return fail(h, TA_ERR_UNSUPPORTED, "ta_hessian_vectors: analytic second derivatives of GRAP models exist for the "
                                         "sf / morse / density / pexp filters, not for the filter network");
    for (const ChunkPlan &cp : h->chunks)
      for (int iz = 0; iz < cp.nz && !grap_model; ++iz)
        if (cp.ch.zeta_int[iz] <= 0)
          return fail(h, TA_ERR_UNSUPPORTED, "ta_hessian_vectors: integer zetas only");
  } else if (!h->eam || !ta::eam_hvp_supported(h->eam)) {
    return fail(h, TA_ERR_UNSUPPORTED, "ta_hessian_vectors: analytic second derivatives exist for the symmetry-function and GRAP "
                                       "models and for EAM / ADP models whose functions are of the Zjw04 family, networks or tabulated");
  }
  const size_t N = (size_t)h->db.n_atoms, F = (size_t)h->db.n_frames;
  const bool unit = !dR && !dh;
  if (unit && (first < 0 || (size_t)first + (size_t)n_dir > 3 * N))
    return fail(h, TA_ERR_INVALID, "ta_hessian_vectors: without dR / dh the directions are unit displacements "
                                   "first .. first + n_dir - 1 of the 3 N");
  if ((size_t)n_dir > 65535) return fail(h, TA_ERR_INVALID, "ta_hessian_vectors: at most 65535 directions per call");
  if (sf_model)
    return guarded(h, [&]() {
      // Per direction: D-dot per pair, G-dot through the pair Jacobians, w-dot = H_mlp G-dot from the
      // second-order MLP pass, then g and g-dot from the dual-arithmetic backward expression (ta_hvp.hip).
      if (n_dir == 0 || N == 0) return;
      hipStream_t s = h->stream;
      const size_t P = (size_t)h->db.n_pairs, nd = (size_t)n_dir;
      const int D = h->sf.ndim;
      ensure_pair_jacobians(h);
      compute_impl(h, TA_WANT_ENERGY, false, nullptr);  // dE/dG of the resident positions (the one-hots overwrote it)
      size_t scratch = 0, partial = 0, total = 0;
      for (int e = 0; e < h->n_elements; ++e) {
        const int n_el = h->db.elem_start[e + 1] - h->db.elem_start[e];
        scratch = std::max(scratch, ta::mlp_grad2_scratch_doubles(h->mlp[e], n_el));
        partial = std::max(partial, ta::mlp_grad_partial_doubles(h->mlp[e], n_el));
        total += (size_t)ta::mlp_param_count(h->mlp[e]);
      }
      h->train_scratch.ensure(scratch + 8);
      h->train_partial.ensure(partial + 8);
      h->train_grad.ensure(total + 8);
      h->tan_dir.ensure(3 * N + 9 * F + 8);
      h->tan_dD.ensure(4 * P + 8);
      h->tan_dG.ensure(N * (size_t)D + 8);
      // Dv, gv, gd [P][4]; w-dot [N][D]; F-dot [N][3], W-dot rows [N][9] of one direction
      h->hvp_buf.ensure(12 * P + N * (size_t)D + 12 * N + 64);
      double *Dv = h->hvp_buf.ptr, *gv = Dv + 4 * P, *gd = gv + 4 * P, *wdot = gd + 4 * P, *fdot = wdot + N * (size_t)D,
             *wrow = fdot + 3 * N;
      ta::launch_pair_vec(h->db, Dv, s);
      std::vector<double> dir_R(3 * N), dir_h(9 * F), rows(9 * N);
      double *d_dR = h->tan_dir.ptr, *d_dh = h->tan_dir.ptr + 3 * N;
      for (size_t d = 0; d < nd; ++d) {
        if (unit) {
          std::fill(dir_R.begin(), dir_R.end(), 0.0);
          dir_R[(size_t)first + d] = 1.0;
        } else if (dR) {
          std::copy(dR + d * 3 * N, dR + (d + 1) * 3 * N, dir_R.begin());
        } else {
          std::fill(dir_R.begin(), dir_R.end(), 0.0);
        }
        if (dh) std::copy(dh + d * 9 * F, dh + (d + 1) * 9 * F, dir_h.begin());
        else std::fill(dir_h.begin(), dir_h.end(), 0.0);
        HIP_CHECK(hipMemcpyAsync(d_dR, dir_R.data(), 3 * N * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_dh, dir_h.data(), 9 * F * sizeof(double), hipMemcpyHostToDevice, s));
        ta::launch_pair_tangent(h->db, d_dR, d_dh, h->tan_dD.ptr, s);
        ta::launch_descriptor_jvp(h->db, D, h->jvp_J.ptr, h->tan_dD.ptr, h->tan_dG.ptr, s);
        size_t off = 0;
        for (int e = 0; e < h->n_elements; ++e) {
          const int n_el = h->db.elem_start[e + 1] - h->db.elem_start[e];
          ta::launch_mlp_grad2(h->mlp[e], h->activation, D, h->db.elem_atoms + h->db.elem_start[e], n_el, h->db,
                               h->tan_dG.ptr, nullptr, h->train_scratch.ptr, h->train_partial.ptr,
                               h->train_grad.ptr + off, s, wdot);
          off += (size_t)ta::mlp_param_count(h->mlp[e]);
        }
        if (grap_model) {
          ta::launch_grap_hvp(h->grap, h->db, h->sf.eps, Dv, h->tan_dD.ptr, wdot, gv, gd, s);
        } else if (h->sf.angular) {
          bool first_launch = true;
          for (const ChunkPlan &cp : h->chunks) {
            ta::launch_backward_hvp(h->sf, cp.ch, cp.nb, cp.ng, cp.nz, first_launch, true, h->db, Dv, h->tan_dD.ptr, wdot,
                                    gv, gd, s);
            first_launch = false;
          }
        } else {
          ta::AngChunk dummy;
          std::memset(&dummy, 0, sizeof(dummy));
          ta::launch_backward_hvp(h->sf, dummy, 1, 1, 1, true, false, h->db, Dv, h->tan_dD.ptr, wdot, gv, gd, s);
        }
        ta::launch_hvp_gather(h->db, Dv, h->tan_dD.ptr, gv, gd, fdot, dW ? wrow : nullptr, s);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dF + d * 3 * N, fdot, 3 * N * sizeof(double), hipMemcpyDeviceToHost, s));
        if (dW) HIP_CHECK(hipMemcpyAsync(rows.data(), wrow, 9 * N * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));  // (the direction buffers are reused)
        if (dW) {
          std::fill(dW + d * F * 9, dW + (d + 1) * F * 9, 0.0);
          const std::vector<int32_t> &foa = h->hp.frame_of_atom;
          for (size_t i = 0; i < N; ++i)
            for (int k = 0; k < 9; ++k) dW[(d * F + (size_t)foa[i]) * 9 + k] += rows[9 * i + k];
        }
      }
    });
  return guarded(h, [&]() {
    if (n_dir == 0 || N == 0) return;
    hipStream_t s = h->stream;
    compute_impl(h, TA_WANT_ENERGY, false, nullptr);  // F'(rho_i) of the resident positions
    const size_t nd = (size_t)n_dir;
    const size_t extra_n = ta::eam_hvp_extra_doubles(h->eam, h->db, n_dir);
    h->hvp_buf.ensure(nd * N * (1 + 3 + (dW ? 9 : 0)) + (dR ? nd * N * 3 : 0) + (dh ? nd * F * 9 : 0) + extra_n + 8);
    double *p = h->hvp_buf.ptr;
    double *extra = extra_n ? p : nullptr;
    p += extra_n;
    double *dFdot = p; p += nd * N;
    double *fdot = p; p += nd * N * 3;
    double *wdot = nullptr;
    if (dW) { wdot = p; p += nd * N * 9; }
    double *d_dR = nullptr, *d_dh = nullptr;
    if (dR) {
      d_dR = p; p += nd * N * 3;
      HIP_CHECK(hipMemcpyAsync(d_dR, dR, nd * N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    if (dh) {
      d_dh = p; p += nd * F * 9;
      HIP_CHECK(hipMemcpyAsync(d_dh, dh, nd * F * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    ta::eam_hvp(h->eam, h->db, n_dir, unit, first, d_dR, d_dh, dFdot, fdot, wdot, extra, s);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(dF, fdot, nd * N * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    std::vector<double> wat;
    if (dW) {
      wat.resize(nd * N * 9);
      HIP_CHECK(hipMemcpyAsync(wat.data(), wdot, nd * N * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    if (dW) {  // per-frame sums of the per-atom rows, in atom order
      std::fill(dW, dW + nd * F * 9, 0.0);
      const std::vector<int32_t> &foa = h->hp.frame_of_atom;
      for (size_t d = 0; d < nd; ++d)
        for (size_t i = 0; i < N; ++i) {
          const size_t f = (size_t)foa[i];
          for (int k = 0; k < 9; ++k) dW[(d * F + f) * 9 + k] += wat[(d * N + i) * 9 + k];
        }
    }
  });
}

int ta_set_nn_tables(ta_handle h, int on) {
  if (!h) return TA_ERR_INVALID;
  if (!h->eam) return TA_OK;
  return guarded(h, [&]() {
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->filtered && !on)
      throw std::domain_error("ta_set_nn_tables: a skin-filtered batch is resident; ta_set_frames again afterwards "
                              "(set the mode before ta_set_frames)");
    ta::eam_set_nn_tables(h->eam, on != 0);
    if (h->have_batch) ta::eam_ensure(h->eam, h->db);
  });
}

int ta_eam_tabulate(ta_handle h, int32_t n_r, const double *r, int32_t n_rho, const double *rho,
                    double *rho_of_r, double *phi_of_r, double *embed_of_rho, double *u_of_r,
                    double *w_of_r) {
  if (!h) return TA_ERR_INVALID;
  if (h->kind != TA_MODEL_EAM_ALLOY && h->kind != TA_MODEL_EAM_ADP)
    return fail(h, TA_ERR_INVALID, "ta_eam_tabulate needs an EAM / ADP model");
  if (n_r < 0 || n_rho < 0 || (n_r > 0 && (!r || !rho_of_r || !phi_of_r)) ||
      (n_rho > 0 && (!rho || !embed_of_rho)))
    return fail(h, TA_ERR_INVALID, "bad table arguments");
  return guarded(h, [&]() {
    const size_t nel = (size_t)h->n_elements, npair = nel * (nel + 1) / 2;
    const bool adp = h->kind == TA_MODEL_EAM_ADP && u_of_r && w_of_r;
    DevBuf<double> buf;
    const size_t n_in = (size_t)n_r + (size_t)n_rho;
    const size_t n_out = (nel + npair * (adp ? 3 : 1)) * (size_t)n_r + nel * (size_t)n_rho;
    buf.ensure(n_in + n_out + 8);
    double *d_r = buf.ptr, *d_rho = d_r + n_r;
    double *d_rho_r = d_rho + n_rho, *d_phi = d_rho_r + nel * n_r, *d_embed = d_phi + npair * n_r;
    double *d_u = adp ? d_embed + nel * n_rho : nullptr, *d_w = adp ? d_u + npair * n_r : nullptr;
    hipStream_t s = h->stream;
    if (n_r) HIP_CHECK(hipMemcpyAsync(d_r, r, (size_t)n_r * sizeof(double), hipMemcpyHostToDevice, s));
    if (n_rho) HIP_CHECK(hipMemcpyAsync(d_rho, rho, (size_t)n_rho * sizeof(double), hipMemcpyHostToDevice, s));
    ta::eam_tabulate(h->eam, n_r, d_r, n_rho, d_rho, d_rho_r, d_phi, d_embed, d_u, d_w, s);
    HIP_CHECK(hipGetLastError());
    auto back = [&](double *dst, const double *src, size_t n) {
      if (dst && n) HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, s));
    };
    back(rho_of_r, d_rho_r, nel * n_r);
    back(phi_of_r, d_phi, npair * n_r);
    back(embed_of_rho, d_embed, nel * n_rho);
    if (adp) {
      back(u_of_r, d_u, npair * n_r);
      back(w_of_r, d_w, npair * n_r);
    }
    HIP_CHECK(hipStreamSynchronize(s));
    buf.release();
  });
}

int ta_get_pairs(ta_handle h, int32_t *i, int32_t *j, int32_t *shift) {
  if (!h) return TA_ERR_INVALID;
  if (!h->have_batch) return fail(h, TA_ERR_INVALID, "no resident batch");
  const size_t P = (size_t)h->hp.n_pairs;
  if (h->pairs_on_device) {
    return guarded(h, [&]() {
      HIP_CHECK(hipStreamSynchronize(h->stream));
      // the RESIDENT list (n_pairs of ta_batch_info): under a Verlet skin that is the skin list, not
      // the exact list the kernels extract from it per step (whose arrays are defined only up to
      // their own, smaller count)
      const int32_t *pi = h->filtered ? h->pair_i.ptr : h->db.pair_i;
      const int32_t *pj = h->filtered ? h->pair_j.ptr : h->db.pair_j;
      const int32_t *ps = h->filtered ? h->pair_shift.ptr : h->db.pair_shift;
      if (i && P) HIP_CHECK(hipMemcpy(i, pi, P * sizeof(int32_t), hipMemcpyDeviceToHost));
      if (j && P) HIP_CHECK(hipMemcpy(j, pj, P * sizeof(int32_t), hipMemcpyDeviceToHost));
      if (shift && P) HIP_CHECK(hipMemcpy(shift, ps, 3 * P * sizeof(int32_t), hipMemcpyDeviceToHost));
    });
  }
  if (i) std::memcpy(i, h->hp.pair_i.data(), P * sizeof(int32_t));
  if (j) std::memcpy(j, h->hp.pair_j.data(), P * sizeof(int32_t));
  if (shift) std::memcpy(shift, h->hp.pair_shift.data(), 3 * P * sizeof(int32_t));
  return TA_OK;
}

}  // extern "C"
