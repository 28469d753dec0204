// Device-side data structures and launcher declarations (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tensoralloy_amd.h"

namespace ta {

constexpr int kMaxRadial = 64;    // n_eta * n_omega
constexpr int kMaxElements = 8;
constexpr int kMaxBetaSlots = 3;  // H_beta slots in a pair record
constexpr int kRecDoubles = 8;    // pair record = 64 bytes
constexpr int kMaxLayers = 8;
constexpr int kCapMin = 192;      // pair records one v2 workgroup (one wavefront) stages in LDS
constexpr int kCapMax = 1024;
constexpr int kMaxCentersPerBlock = 16;  // centres one angular workgroup may own
constexpr int kMaxHd = 24;        // coefficients of the Hd(u) expansion

// Symmetry-function constants, passed by value to every kernel.
struct SFParams {
  double rcut, acut;
  double inv_rc2, inv_ac2;  // 1 / rcut^2, 1 / acut^2
  double two_inv_ac2;       // 2 / acut^2 (du/dr = 2 r / acut^2: one scalar operand in the triple bodies)
  double eps;               // added under the square root (universal.py:470-472)
  double ang_scale;         // 0.5 when every {j,k} is visited twice (v1 kernels), else 1
  int n_elements;
  int n_rad;                // radial parameter combinations (eta x omega)
  int n_ang;                // angular parameter combinations (beta x gamma x zeta)
  int n_radial_dim;         // n_elements * n_rad
  int ndim;                 // descriptor length per atom
  int angular;
  int cutoff;               // TA_CUTOFF_*
  int n_beta;
  double eta[kMaxRadial], omega[kMaxRadial];
  double beta[kMaxBetaSlots];
  // Power chain of the radial exponentials: eta_pow[c] = k >= 2 when eta[c] = k eta[c - 1] exactly,
  // omega[c] = omega[c - 1] and c - 1 lies in the same group of four channels, so that
  // exp(-eta[c] x) = exp(-eta[c - 1] x)^k costs a few multiplications instead of an exponential
  // (the reference's default grid eta = {0.05, 4, 20, 80}: two exponentials instead of four);
  // 0 = evaluate the exponential.
  signed char eta_pow[kMaxRadial];
};

// One launch of the angular kernels handles NB x NG x NZ channels.
struct AngChunk {
  double beta[2];
  int hslot[2];      // which H slot of the pair record holds exp(-beta r^2/ac^2) fc(r)
  double gamma[2];
  double zeta[2];
  double kz[2];      // 2^(1 - zeta)
  int zeta_int[2];   // zeta as integer >= 1, or -1 when not an integer
  int chan[8];       // [(ib*NG + ig)*NZ + iz] -> channel index in [0, n_ang)
  int safe_pow;      // 1: the masked `safe_pow` of extension/grad_ops.py:16-66 for non-integer zeta
  int n_hd;          // 0 (exact), 12, 16 or 24 coefficients of Hd(u) = exp(-beta u) fc(u)
  double hd[kMaxHd]; // power-series coefficients in u (v2 kernels, beta[0] only)
};

struct DeviceBatch {
  int64_t n_atoms = 0, n_pairs = 0;
  int n_frames = 0, nnl_max = 0;
  // inputs
  double *pos = nullptr;         // [N][3]
  double *cells = nullptr;       // [F][9]
  int32_t *species = nullptr;    // [N]
  int32_t *frame_of_atom = nullptr;
  int32_t *atom_start = nullptr; // [F+1]
  int32_t *pair_start = nullptr; // [N+1]
  // [N] or null. Null: the pairs of centre i end where those of i + 1 begin. Set on the MD path
  // (ta_nlist.hip::nl_filter), whose exact list keeps every group of 16 centres at the group's offset
  // in the skin list (no prefix sum over the whole batch), so groups are separated by unused slots.
  int32_t *pair_stop = nullptr;
  int32_t *seg_start = nullptr;  // [N][nel+1]
  int32_t *pair_i = nullptr, *pair_j = nullptr, *pair_shift = nullptr, *pair_rev = nullptr;
  // MD path of the symmetry-function models: the exact list has no reverse index of its own; the reverse of
  // slot p is rev_map[rev_super[slot_q[p]]] (slot -> skin-list pair -> its reverse there -> that pair's slot),
  // looked up by the one kernel that needs it (force_gather) instead of a launch that writes it out
  const int32_t *slot_q = nullptr, *rev_super = nullptr, *rev_map = nullptr;
  int32_t *blk_center = nullptr; // [n_blk+1] first centre of every v2 workgroup
  int n_blk = 0;
  // MD step (ta_nlist.hip::filter_group_kernel): > 0 = number of groups of 16 centres, each owning 16
  // run slots of which the later ones are usually empty. Workgroup b then takes slot
  // 16 (b mod groups) + b / groups, so that the slots in use come first in launch order (interleaved
  // with the empty ones they ran on half of the shader engines: 48 -> 77 us for the forward kernel)
  int blk_groups = 0;
  const int32_t *n_blk_dev = nullptr;  // MD loop: the packing was made on the device, `n_blk` is only an upper
                                       // bound of the grid; workgroups beyond *n_blk_dev leave at once
  int cap = kCapMin;             // records per v2 workgroup (multiple of 64)
  int32_t *elem_atoms = nullptr; // atoms grouped by element
  int32_t elem_start[kMaxElements + 1] = {0};

  // work buffers
  double *rec = nullptr;    // [P][8]  {Dx,Dy,Dz,r2,1/r,H0,H1,H2}
  double *rec4 = nullptr;   // [P][4]  {Dx,Dy,Dz,r2}: compact records of the second-generation angular,
                            // GRAP and EAM / ADP paths (set instead of `rec`: half the record traffic
                            // of the forward, backward and gather kernels; 1/r is recomputed)
  double *part4 = nullptr;  // [nel*n_ang][P] per-pair partial angular sums
  unsigned long long *masks = nullptr;  // [ceil(nnl_max/128)][P] candidate masks, forward -> backward
  // Lane balance of the angular kernels: the forward kernel cuts every lane's 64 candidate positions
  // into four windows of 16, makes every non-empty window a job, sorts the jobs by size and leaves
  // the list here for the backward kernel: per workgroup `job_count[blk]` words at
  // [blk * job_stride ..): bits 0-7 the pair (item of the workgroup), bits 8-9 the window, bits
  // 16-31 the window's candidate bits. null: the per-lane masks above are used.
  uint32_t *job_word = nullptr;
  int32_t *job_count = nullptr;
  int job_stride = 0;
  double *G = nullptr;      // [N][D]
  double *dEdG = nullptr;   // [N][D]
  double *eatom = nullptr;  // [N]
  double *g = nullptr;      // [P][4]  dE/dD per directed pair (x, y, z, pad)
  double *forces = nullptr; // [N][3]
  double *wat = nullptr;    // [N][9] per-atom virial (only atoms of groups that straddle two frames)
  // Per-atom own-side sums {sum_p g[p] (3), sum_p g[p] (x) D[p] (9)} over the atom's pairs, left by the
  // GRAP backward kernel (one wavefront per centre: nearly free there); force_gather then only gathers
  // g[rev p]. `own_sums` != 0: valid for this evaluation. (The angular backward kernel does not: its
  // per-centre reduction cost what the gather saved, profiles/r02_tuning_notes.md.)
  double *fown = nullptr;
  int own_sums = 0;
#ifdef TA_PHASE_STAMPS
  // diagnostic builds only (scripts/phase_stamps.sh): s_memrealtime (100 MHz) at the phase boundaries
  // of the angular kernels, [kernel 0 / 1][workgroup][8]
  unsigned long long *stamps = nullptr;
#endif
  double *bpart = nullptr;  // [ceil(N / 16)][10] {E, W[9]} of every group of 16 consecutive atoms
  double *energy = nullptr; // [F]
  double *virial = nullptr; // [F][9]
  double *batch_energy = nullptr;  // [1]
};

// end of centre i's pairs (see DeviceBatch::pair_stop)
#ifdef __HIPCC__
__device__ __forceinline__ int pair_stop_of(const DeviceBatch &b, int64_t i) {
  return b.pair_stop ? b.pair_stop[i] : b.pair_start[i + 1];
}
#endif

// {Dx, Dy}, {Dz, r^2} of pair q, from the compact or the full record
#ifdef __HIPCC__
__device__ __forceinline__ int pair_rev_of(const DeviceBatch &b, int q) {
  return b.slot_q ? b.rev_map[b.rev_super[b.slot_q[q]]] : b.pair_rev[q];
}
__device__ __forceinline__ const double2 *pair_geom(const DeviceBatch &b, size_t q) {
  return reinterpret_cast<const double2 *>(b.rec4 ? b.rec4 + 4 * q : b.rec + kRecDoubles * q);
}
#endif

// Per-element MLP on the device: padded weights, both orientations.
struct MlpLayerDev {
  int k, n;        // logical in / out
  int kp, np;      // padded: kp % 4 == 0, np % 16 == 0
  double *w;       // [kp][np]
  double *wt;      // [np][kp]
  double *b;       // [np]
  int act;         // apply activation
  int res;         // resnet skip
};
struct MlpDev {
  int n_layers = 0;  // incl. output layer
  MlpLayerDev layer[kMaxLayers];
  double *xlo = nullptr, *xhi = nullptr;  // [D] or null
  int max_np = 0, max_kp = 0;
};

void launch_pair_geometry(const SFParams &sf, const DeviceBatch &b, hipStream_t s);
void launch_g4_forward(const SFParams &sf, const AngChunk &ch, int nb, int ng, int nz,
                       const DeviceBatch &b, hipStream_t s);
void launch_descriptor_reduce(const SFParams &sf, const DeviceBatch &b, hipStream_t s);
void launch_mlp(const SFParams &sf, const MlpDev &mlp, int activation, int element,
                const DeviceBatch &b, hipStream_t s);
void launch_backward(const SFParams &sf, const AngChunk &ch, int nb, int ng, int nz,
                     bool first, bool radial_only, const DeviceBatch &b, hipStream_t s);
void launch_force_gather(const SFParams &sf, const DeviceBatch &b, hipStream_t s);
void launch_frame_reduce(const DeviceBatch &b, bool want_virial, double *mirror, int64_t n_tail, hipStream_t s);

size_t g4_lds_bytes(int nnl_max);

// second-generation angular kernels (ta_kernels_v2.hip); `ch` holds one beta
size_t v2_lds_bytes(bool backward, int cap, int n_local = 0, int nspec = 0);
int v2_job_stride(int cap);
// `reduce`: last forward launch of an evaluation, also assembles the descriptor vectors
void launch_g4_forward_v2(const SFParams &sf, const AngChunk &ch, int ng, int nz, bool geometry,
                          bool reduce, const DeviceBatch &b, hipStream_t s);
void launch_backward_v2(const SFParams &sf, const AngChunk &ch, int ng, int nz, bool first,
                        const DeviceBatch &b, hipStream_t s);

// device-side neighbour list (ta_nlist.hip)
struct NlGrid {  // linked-cell grid of one frame, in fractional coordinates
  double h[9];
  double hinv[9];
  double lo[3];     // lower edge of the grid (0 on periodic axes)
  double inv_w[3];  // bins per unit of fractional coordinate
  int32_t nb[3];
  int32_t pbc[3];
  int32_t bin_offset;  // first bin of this frame in the batch-wide bin arrays
  int32_t pad_;
  // neighbouring bins looked at along each axis: offsets -m .. m (1 unless the cell is thinner than the
  // cutoff along a periodic axis: then nb = 1 there and m = floor(rmax / height) + 1 images of the one bin)
  int32_t m[3];
  int32_t ortho;    // cell vectors mutually orthogonal: the per-axis gaps to a bin add in quadrature
  double bw[3];     // perpendicular (Cartesian) width of one bin along each axis; 0 = unknown, no pruning
};
struct NlRec {  // one atom, stored in bin order
  double x, y, z;
  int32_t wx, wy, wz;  // integer wrap into the cell
  int32_t j;           // atom index
  int32_t sp;          // species
  int32_t pad_;
};
struct NlWork {  // device work buffers, sized by the caller
  int32_t *wrap;        // [3 N] integer wrap of each atom into its cell
  int32_t *binid;       // [N]
  int32_t *bin_count;   // [n_bins + 1]
  int32_t *bin_start;   // [n_bins + 1]
  int32_t *bin_cursor;  // [n_bins + 1]
  int32_t *bin_atoms;   // [N]
  NlRec *recs;          // [N]
  int32_t *counts;      // [N (nel + 1) + 1]
  int32_t *seg_start;   // [N (nel + 1) + 1]
  // 8 words. stats[0] = number of triples; as int32: [2] nnl_max, [4] number of pairs (32-bit scan
  // total), [6] pairs without a reverse partner (must stay 0); stats[4] = number of pairs, 64-bit
  unsigned long long *stats;
};
bool nl_make_grid(const ta_frame &fr, double rmax, int bin_offset, NlGrid &g);
int nl_bins(const NlGrid &g);
void nl_count(int n_atoms, int n_bins, int nel, double rmax, const double *pos, const int32_t *species,
              const int32_t *frame_of_atom, const NlGrid *grids, NlWork &w, int32_t *pair_start,
              hipStream_t s);
size_t nl_build_zero_words(int n_atoms, int n_bins);
void nl_build(int n_atoms, int n_bins, int nel, double rmax, const double *pos, const int32_t *species,
              const int32_t *frame_of_atom, const NlGrid *grids, NlWork &w, unsigned long long *zero,
              bool zero_is_clean, long long capacity, int32_t *pair_start, int32_t *host_pair_start,
              int32_t *pair_i, int32_t *pair_j, int32_t *pair_shift, int32_t *pair_rev, long long rev_cover,
              hipStream_t s);
void nl_reverse_sorted(int64_t n_pairs, int nel, const int32_t *species, const int32_t *seg_start,
                       const int32_t *pair_i, const int32_t *pair_j, const int32_t *pair_shift,
                       int32_t *pair_rev, unsigned long long *stats, hipStream_t s);
int nl_filter_blocks(int n_atoms);
void nl_filter(int n_atoms, int64_t n_super, int nel, double rmax, const double *pos, const double *cells,
               const int32_t *frame_of_atom, const int32_t *start_super, const int32_t *seg_super,
               const int32_t *pj_super, const int32_t *ps_super, const int32_t *rev_super, int32_t *map,
               int32_t *seg_exact, int32_t *pair_start, int32_t *pair_stop, int32_t *pi_out, int32_t *pj_out,
               int32_t *ps_out, int32_t *rev_out, int32_t *slot_q, int cap, int32_t *blk_center, hipStream_t s);
void nl_fill(int n_atoms, int64_t n_pairs, int nel, double rmax, const double *pos,
             const int32_t *species, const int32_t *frame_of_atom, const NlGrid *grids, NlWork &w,
             int32_t *pair_i, int32_t *pair_j, int32_t *pair_shift, int32_t *pair_rev, hipStream_t s);

}  // namespace ta
