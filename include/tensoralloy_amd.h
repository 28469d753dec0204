/*
 * tensoralloy_amd.h — C ABI of libtensoralloy_amd.so (MI355X / gfx950 only).
 *
 * The reference (Bismarrck/tensoralloy) has NO native plugin ABI: its hot path
 * is a frozen TensorFlow-1 graph executed by `Session.run`
 * (tensoralloy/calculator.py:335-370). This library is what sits UNDER the
 * reference's two Python surfaces instead of that graph:
 *
 *   tensoralloy/calculator.py:31-383      TensorAlloyCalculator.calculate
 *   tensoralloy/transformer/universal.py  UniversalTransformer (feed dict)
 *
 * Each entry point below names the reference interface it replaces. All
 * buffers are caller-owned, C-contiguous; the library copies host->device and
 * never keeps a host pointer after the call returns. A handle owns one device
 * and one HIP stream; a handle is not thread-safe, different handles may be
 * used from different threads. Every function returns TA_OK (0) or a negative
 * error code; `ta_last_error` gives the message.
 *
 * There is no CPU fallback: without a gfx950 device `ta_create` fails with
 * TA_ERR_HIP.
 */
#ifndef TENSORALLOY_AMD_H
#define TENSORALLOY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ta_context *ta_handle;

enum {
  TA_OK = 0,
  TA_ERR_INVALID = -1,     /* bad argument / inconsistent model  -> ValueError  */
  TA_ERR_UNSUPPORTED = -2, /* model feature not implemented      -> ValueError  */
  TA_ERR_HIP = -3,         /* HIP runtime failure / no device    -> RuntimeError */
  TA_ERR_NOMEM = -4
};

/* `want` bits of ta_compute / ta_eval: which outputs of
 * tensoralloy/nn/basic.py:679-787 (`BasicNN.build`) are produced. */
enum {
  TA_WANT_ENERGY = 1,      /* Output/Energy/energy     atomic.py:289-302        */
  TA_WANT_FORCES = 2,      /* Output/Forces/forces     basic.py:277-290         */
  TA_WANT_VIRIAL = 4,      /* Output/Stress (virial)   basic.py:293-331         */
  TA_WANT_ATOMIC = 8,      /* Output/Energy/atomic     atomic.py:289-299        */
  TA_WANT_DESCRIPTORS = 16, /* Atomic/<El> descriptors  sf.py:184-215 (debug)   */
  TA_WANT_REUSE_DESCRIPTORS = 32 /* energy-only call on the descriptors of the last ta_compute of
                                    this batch (only the MLP weights changed: training steps) */
};

enum { TA_MODEL_SF_MLP = 1, TA_MODEL_EAM_ALLOY = 2, TA_MODEL_EAM_ADP = 3, TA_MODEL_GRAP_MLP = 4 };
enum { TA_CUTOFF_COSINE = 0, TA_CUTOFF_POLYNOMIAL = 1 }; /* nn/cutoff.py:20-85 */

/* activation ids follow `actfn_map` of atomic.py:323 for 0..3 */
enum {
  TA_ACT_RELU = 0, TA_ACT_SOFTPLUS = 1, TA_ACT_TANH = 2, TA_ACT_SQUAREPLUS = 3,
  TA_ACT_LEAKY_RELU = 4, TA_ACT_SIGMOID = 5, TA_ACT_SOFTSIGN = 6, TA_ACT_ELU = 7
};

/*
 * Model description = what `BasicNN.export` bakes into the frozen graph
 * (basic.py:1075-1092): `UniversalTransformer.as_dict()` (universal.py:323-331),
 * `SymmetryFunction.as_dict()` (sf.py:58-68), `AtomicNN.as_dict()`
 * (atomic.py:116-132) and the variables `Atomic/<El>/Conv1d{j}/{kernel,bias}`,
 * `Atomic/<El>/Output/{kernel,bias}`, `Atomic/<El>/MinMax/{xlo,xhi}`.
 * Elements are the SORTED unique symbols (utils.py:262); species indices in
 * frames index this list.
 */
typedef struct {
  int32_t kind;            /* TA_MODEL_*                                        */
  int32_t n_elements;
  double rcut;             /* UniversalTransformer.rcut                          */
  double acut;             /* UniversalTransformer.acut (== rcut if not angular) */
  int32_t angular;         /* G4 terms present                                   */
  int32_t cutoff_function; /* TA_CUTOFF_*                                        */

  /* SymmetryFunction parameter axes; grids are eta x omega (omega fastest) and
   * beta x gamma x zeta (zeta fastest), sf.py:47-51. */
  int32_t n_eta, n_omega, n_beta, n_gamma, n_zeta;
  const double *eta, *omega, *beta, *gamma, *zeta;

  /* per-element MLP (convolution1x1, convolutional.py:154-300) */
  int32_t activation;      /* TA_ACT_*                                           */
  int32_t use_resnet_dt;
  int32_t minmax_scale;    /* atomic.py:157-195                                  */
  const int32_t *n_layers; /* [n_elements] dense layers incl. the output layer   */
  const int32_t *layer_sizes; /* per element [in, h1, ..., 1], concatenated      */
  const double *weights;   /* per element, per layer: W[in][out] row-major, then
                              b[out] (zeros when the layer has no bias)          */
  const double *xlo, *xhi; /* [n_elements * in] when minmax_scale, else NULL     */

  /* EAM / ADP analytic potentials (nn/eam/potentials/zjw04.py, mishin.py), one flat block:
   *   per element (sorted): 20 constants -- Zjw04 [r_eq f_eq rho_e rho_s alpha beta A B kappa lamda
   *     Fn0 Fn1 Fn2 Fn3 F0 F1 F2 F3 eta Fe], AgSutton90 [a b] (sutton90.py:37-44) or AgrawalBe "Be/1"
   *     [A B D alpha re F0 F1 beta gamma m rc] (agrawal.py:49-55) or RWGrimes "grimes" [G n A rho C D
   *     gamma r0] (grimmes.py:33-37), zero padded -- + embed kind (0 Zjw04 piecewise, 1 Zjw04xc
   *     blended) + potential kind (0 Zjw04 family, 1 sutton90, 2 Be/1, 3 grimes);
   *   per element pair a <= b (upper triangle, row-major): phi kind (0 Zjw04, 1 Zjw04xcp own
   *     constants) + [r_eq A B alpha beta kappa lamda];
   *   ADP only, per pair: [d1 d2 d3 q1 q2 q3 h rc] (all zero = no angular term).           */
  int32_t n_eam_params;
  const double *eam_params;

  /* added under sqrt(D.D + eps) (universal.py:470-472): 1e-14 for 'high' precision models,
   * 1e-8 for 'medium' ones (precision.py:113-114). 0 selects 1e-14. Arithmetic is fp64
   * either way.                                                                      */
  double eps;

  /* GRAP descriptor (TA_MODEL_GRAP_MLP; nn/atomic/grap.py:272-704), with the MLP fields above and
   * `rcut`, `cutoff_function`: [algorithm (0 sf, 1 morse, 2 density, 3 pexp), K filters,
   * max moment (0..3), legacy_mode, symmetric, moment mask (bit m: moment m listed), then K x 3
   * filter constants: sf (eta, omega, -), morse (D, gamma, r0), density (A, beta, re),
   * pexp (rl, pl, -)].                                                                    */
  int32_t n_grap_params;
  const double *grap_params;

  /* EAM / ADP "nn" functions -- the reference's default potentials (nn/eam/alloy.py:110-112,
   * adp.py:120-124): rho(r), F(rho), phi(r), u(r), w(r) given by `convolution1x1` on the scalar
   * argument (nn/eam/eam.py:174-190; no output bias). 0 = every function is analytic. Otherwise
   * the number of function slots, 2 n_elements + n_pairs (ADP: + 2 n_pairs), in the order
   * rho[element], embed[element], phi[pair a <= b], dipole[pair], quadrupole[pair]; `n_layers`
   * [slot] (0 = analytic function, read from `eam_params`), `layer_sizes` ([1, h1, ..., 1] per
   * nn slot) and `weights` (as above, per nn slot) describe them, `activation` applies to all. */
  int32_t n_eam_nets;

  /* EAM / ADP tabulated functions: a LAMMPS setfl / adp file's rho(r), F(rho), phi(r), u(r), w(r)
   * (io/lammps.py:62-235), evaluated as natural cubic splines the way the reference's
   * `CubicInterpolator(x, y, natural_boundary=True)` does (potentials/tests/test_mishin.py:60-70;
   * `spline@...` potentials, train/training.py:258-262). Same slots as the nn functions
   * (`n_eam_nets` must be set): `eam_table_n[slot]` = number of knots (0 = not tabulated; knots at
   * k * eam_table_dx[slot]), `eam_table_coef` = per tabulated slot (n - 1) x 4 doubles, the cubic
   * c0 + c1 t + c2 t^2 + c3 t^3 of every interval with t = x - x_k. Arguments beyond the last knot
   * use the last interval's cubic. All three NULL = no tables. */
  const int32_t *eam_table_n;
  const double *eam_table_dx;
  const double *eam_table_coef;

  /* 1: powers with a non-integer exponent ((1 + gamma cos)^zeta, sf.py:162-164) use the reference's
   * custom-gradient `safe_pow` (extension/grad_ops.py:16-66, selected there by the environment
   * variable TENSORALLOY_USE_CUSTOM_POW): an infinite value and an infinite or NaN derivative factor
   * become 0. 0: plain pow, as `tf.pow`. Integer exponents are products and never singular. */
  int32_t safe_pow;
} ta_model_desc;

/* One structure = what `UniversalTransformer.get_np_feed_dict(atoms)`
 * (universal.py:851-893) receives: an `ase.Atoms`. Atom order is the caller's
 * (ASE) order; the GSL/VAP permutation (vap.py:26-137) is applied by the
 * Python boundary, not here. */
typedef struct {
  int32_t n_atoms;
  const int32_t *species;  /* [n_atoms] index into the sorted element list       */
  const double *positions; /* [n_atoms][3] Angstrom                              */
  const double *cell;      /* [3][3] row-major lattice vectors                   */
  const int32_t *pbc;      /* [3]                                                */
} ta_frame;

typedef struct {
  int32_t n_frames;
  int64_t n_atoms;         /* total over frames                                  */
  int64_t n_pairs;         /* directed pairs within max(rcut, acut)  (= nij)     */
  int64_t n_triples;       /* sum_i n_i (n_i - 1) / 2               (= nijk)     */
  int32_t nnl_max;         /* max neighbours of one centre                       */
  int32_t descriptor_dim;  /* D per atom                                         */
  int32_t nl_on_device;    /* 1: neighbour list built by the GPU kernels, 0: host */
  int32_t reserved_;
  double nl_ms;            /* wall time of the neighbour-list part of ta_set_frames */
  double set_frames_ms;    /* wall time of the whole ta_set_frames call          */
} ta_batch_info;

/* number of kernel-timing slots filled by ta_time_compute */
#define TA_N_KERNEL_SLOTS 10
/* slot ids */
enum {
  TA_K_PAIR_GEOMETRY = 0, TA_K_G4_FORWARD = 1, TA_K_DESCRIPTOR_REDUCE = 2,
  TA_K_MLP = 3, TA_K_BACKWARD = 4, TA_K_FORCE_GATHER = 5, TA_K_FRAME_REDUCE = 6,
  TA_K_EAM = 7,
  TA_K_NEIGHBOR_UPDATE = 8, /* ta_update_positions: displacement check (+ list rebuild) */
  TA_K_GRAP = 9   /* GRAP moments + features (forward) */
};

int ta_device_count(void);

/* Version of this header's ABI (entry points AND struct layouts; bumped whenever either changes) and
 * sizeof(ta_model_desc) as the library was compiled: a binding checks both before the first real
 * call, so that a stale or foreign build of the library is refused instead of misreading a struct.
 * (The reference has no counterpart: its "ABI" is the frozen graph's `Metadata/api`, basic.py:43.) */
#define TA_ABI_VERSION 4
int ta_abi_version(void);
int ta_model_desc_size(void);

/* replaces TensorAlloyCalculator.__init__ graph import + Session creation
 * (calculator.py:40-87): validates the model, uploads weights to `device`. */
int ta_create(const ta_model_desc *model, int device, ta_handle *out);
int ta_destroy(ta_handle h);
const char *ta_last_error(ta_handle h); /* h may be NULL: last create error */

/* replaces UniversalTransformer.get_np_feed_dict (universal.py:851-893):
 * neighbour list (ASE `neighbor_list('ijS')` semantics, universal.py:58),
 * pair buffers sorted by centre, reverse-pair index; uploads everything so the
 * batch is resident in HBM. Frames are independent units. */
int ta_set_frames(ta_handle h, int32_t n_frames, const ta_frame *frames,
                  ta_batch_info *info);

/* MD loop. The reference rebuilds its whole feed dict for every call of `calculate`
 * (calculator.py:355-366 -> get_np_feed_dict, universal.py:851-893). Here the resident batch keeps
 * its atoms, species and periodicity; only coordinates travel:
 *   ta_set_skin          Verlet skin (Angstrom, >= 0; default 0). Lists built from now on cover
 *                        max(rcut, acut) + skin. Pairs beyond the cutoff contribute nothing (cutoff
 *                        functions; explicit r < rcut test in the EAM / ADP kernels), so results are
 *                        those of the exact list up to summation order. Changing the skin drops the
 *                        resident list (the next ta_update_positions / ta_set_frames rebuilds it).
 *   ta_update_positions  positions [n_atoms_total][3] (and cells [n_frames][9], or NULL = unchanged)
 *                        for the frames of the last ta_set_frames, in the same order. The list is
 *                        reused while no atom is further than skin / 2 from where it was when the
 *                        list was built and no cell has changed; otherwise it is rebuilt exactly as
 *                        ta_set_frames does. *rebuilt (may be NULL) = 1 when it was rebuilt.
 *                        With skin = 0 every call rebuilds. Asynchronous on the reuse path.
 *   ta_list_stats        lists built / reused by this handle so far.
 *   ta_list_sizes        directed pairs, triples and the largest per-(centre, species) neighbour
 *                        count of the RESIDENT list (what ta_batch_info reported at ta_set_frames,
 *                        refreshed when ta_update_positions rebuilt the list; nij / nijk / nnl_max
 *                        of the reference's metadata, transformer/universal.py:878-887).
 *                        Any pointer may be NULL. */
int ta_set_skin(ta_handle h, double skin);
int ta_update_positions(ta_handle h, const double *positions, const double *cells, int32_t *rebuilt);
int ta_list_stats(ta_handle h, int64_t *n_builds, int64_t *n_reuses);
int ta_list_sizes(ta_handle h, int64_t *n_pairs, int64_t *n_triples, int32_t *nnl_max);

/* replaces Session.run(ops, feed_dict) (calculator.py:368): enqueues the
 * kernels on the handle's stream; asynchronous. */
int ta_compute(ta_handle h, uint32_t want);

/* copies results device->host and synchronises. Any pointer may be NULL.
 *   energy      [n_frames]            eV
 *   forces      [n_atoms_total][3]    eV/A        (caller's atom order)
 *   virial      [n_frames][9]         eV, W_ab = sum_pairs dE/dD_a * D_b
 *   atomic      [n_atoms_total]       eV
 *   descriptors [n_atoms_total][D]    raw G (before min-max)             */
int ta_get_results(ta_handle h, double *energy, double *forces, double *virial,
                   double *atomic, double *descriptors);

/* ta_set_frames + ta_compute + ta_get_results */
int ta_eval(ta_handle h, int32_t n_frames, const ta_frame *frames, uint32_t want,
            double *energy, double *forces, double *virial, double *atomic);

/* One MD / relaxation step in ONE call: ta_update_positions + ta_compute + ta_get_results (what
 * `TensorAlloyCalculator.calculate` does per call once the structure is resident, calculator.py:335-370;
 * one library entry instead of three saves the binding's round trips between them). Arguments as
 * those three; forces / virial / atomic / rebuilt may be NULL. */
int ta_step(ta_handle h, const double *positions, const double *cells, uint32_t want, double *energy,
            double *forces, double *virial, double *atomic, int32_t *rebuilt);

/* The same results WITHOUT the copy into caller arrays: pointers into the handle's page-locked staging
 * buffer, where the device wrote them (layout as ta_get_results; NULL for what `want` did not ask for).
 * They stay valid until the next call on this handle. For bindings that wrap the memory as arrays
 * (numpy.ctypeslib): one MD step hands back 128 KB for a 4000-atom frame, and copying them out of
 * freshly DMA-written memory costs the host more than the transfer itself.
 *   ta_view_results  after ta_compute;  ta_step_view = ta_update_positions + ta_compute + ta_view_results */
int ta_view_results(ta_handle h, uint32_t want, const double **energy, const double **forces,
                    const double **virial, const double **atomic);
int ta_step_view(ta_handle h, const double *positions, const double *cells, uint32_t want, const double **energy,
                 const double **forces, const double **virial, const double **atomic, int32_t *rebuilt);

/* Enqueue all further work of this handle on `stream` (a hipStream_t of the
 * handle's device owned by the caller, e.g. the stream a RCCL collective is
 * ordered against); NULL restores the handle's own stream (to name the legacy
 * default stream, whose handle is also 0, pass hipStreamLegacy). Drains the old
 * stream first. */
int ta_set_stream(ta_handle h, void *stream);

/* blocks until the handle's stream is idle */
int ta_synchronize(ta_handle h);

/* sum of the resident batch's frame energies, left on the device for a
 * collective: returns a device pointer to one double (valid until destroy). */
int ta_batch_energy_device_ptr(ta_handle h, void **dptr);

/* enqueues, on the handle's stream, a device-to-device copy of that double into
 * `dst_device` (e.g. the buffer a RCCL all-reduce will sum across ranks). */
int ta_copy_batch_energy(ta_handle h, void *dst_device);
/* Alternative without the copy: later ta_compute calls write the batch energy (one double) straight
 * to `dst_device` (caller-owned device memory, e.g. the buffer a collective reduces in place);
 * NULL restores the library's own buffer. Takes effect for launches made after the call;
 * ta_set_frames restores the library's own buffer. */
int ta_set_batch_energy_target(ta_handle h, void *dst_device);

/* Host-only (no GPU needed): the neighbour list the library builds for one
 * frame, i.e. `ase.neighborlist.neighbor_list('ijS', atoms, rc)` as called at
 * transformer/universal.py:58, sorted by centre then neighbour species then
 * distance. Output arrays are malloc'ed by the library; release with ta_free.
 * rev[p] = index of the reverse pair (j -> i, -S). */
int ta_neighbor_list(const ta_frame *frame, int32_t n_elements, double rc, int64_t *n_pairs,
                     int32_t **i, int32_t **j, int32_t **shift /*[n][3]*/, int32_t **rev);
void ta_free(void *p);

/* --- training support: gradients with respect to the network weights -----------
 * Parameter vector layout = `ta_model_desc.weights`: per element (sorted) -- for EAM / ADP models
 * per nn-function slot (rho[element], embed[element], phi / dipole / quadrupole[pair]; analytic and
 * tabulated slots hold nothing) --, per layer W[in][out] row-major then b[out]. */

/* length of that vector */
int ta_param_count(ta_handle h, int64_t *n_params);

/* replace the MLP weights of a live handle (an optimiser step), same layout */
int ta_update_weights(ta_handle h, const double *weights, int64_t n_weights);

/* grad = sum_f frame_coeff[f] * dE_f/dtheta for the resident batch: with frame_coeff[f] = dL/dE_f
 * this is the gradient of a loss L(E_1 .. E_F), what `tf.gradients(loss, variables)` gives for the
 * energy term of nn/losses.py:204-285. The batch's descriptors are computed once and reused. */
int ta_energy_gradient(ta_handle h, const double *frame_coeff, double *grad, int64_t n_grad);

/* The whole loss gradient of an energy + forces + stress loss (nn/losses.py:204-437 through
 * `tf.gradients`, nn/opt.py:89-166) for the per-atom MLP models, analytically. With u = dL/dF per
 * atom and the symmetric Y = (dL/dstress) / V per frame, sum u.F + sum Y.W is the directional
 * derivative D_delta E of the energy along
 *     dR = R.Y - u   [n_atoms_total][3],      dh = h.Y   [n_frames][9]
 * (W = -F^T R + (dE/dh)^T h, basic.py:306-316), and
 *     grad = d/dtheta ( sum_f frame_coeff[f] E_f  +  D_delta E ).
 * The descriptors do not depend on theta: their Jacobian with respect to the pair vectors is made
 * once per resident batch (one backward launch per descriptor channel), every call then costs a
 * pair sweep and ONE second-order pass through the MLP. frame_coeff, dR, dh may each be NULL (= 0).
 * dG_out (may be NULL): the directional derivative of the raw descriptors [n_atoms_total][D] that
 * entered the pass (parity tests).
 * EAM / ADP models with nn functions (the reference's default potentials, alloy.py:110-112, adp.py:120-124)
 * since round 3: every network enters D_delta E through its value and its input derivative at known
 * points, so the gradient is one second-order pass per network over its rows (pairs / atoms); dG_out
 * must be NULL. The sutton90 / Be/1 / grimes families: TA_ERR_UNSUPPORTED (callers difference
 * ta_energy_gradient on displaced frames instead). */
int ta_loss_gradient(ta_handle h, const double *frame_coeff, const double *dR, const double *dh,
                     double *grad, int64_t n_grad, double *dG_out);

/* Constants of the analytic functions of an EAM model as trainable parameters. The reference makes
 * every constant of its empirical potentials a tf.Variable (potentials/potentials.py:129-163;
 * zjw04.py: shared variables per element) trained under the same loss. Layout of the vector:
 * element e, constant k -> [20 e + k] in the order of the model description's `eam_el` rows
 * (Zjw04: r_eq f_eq rho_e rho_s alpha beta A B kappa lamda Fn0..Fn3 F0..F3 eta Fe), then the
 * Zjw04xcp cross terms [20 nel + 7 pt + q] (r_eq A B alpha beta kappa lamda), pt = sorted pair type,
 * then for ADP models the dipole / quadrupole constants [20 nel + 7 npt + 8 pt + k]
 * (d1 d2 d3 q1 q2 q3 h rc; mishin.py:62-66).
 *   ta_constant_count     length of the vector
 *   ta_get_constants      current values
 *   ta_update_constants   new values (finite); synchronises the stream first
 *   ta_constant_gradient  d/dconstants of  sum_f frame_coeff[f] E_f + D_(dR, dh) E  on the resident
 *                         batch, arguments as ta_loss_gradient. Forward-mode (dual numbers), one
 *                         pass over the pairs per constant in a single launch. Models that mix
 *                         networks or tables with analytic functions (round 3): the functions
 *                         without constants enter as plain values (exact forward pass), an
 *                         embedding network through F', F''; callers take the weights' half of
 *                         the gradient from ta_loss_gradient. */
int ta_constant_count(ta_handle h, int64_t *n_constants);
int ta_get_constants(ta_handle h, double *constants, int64_t n_constants);
int ta_update_constants(ta_handle h, const double *constants, int64_t n_constants);
int ta_constant_gradient(ta_handle h, const double *frame_coeff, const double *dR, const double *dh,
                         double *grad, int64_t n_grad);

/* Analytic second derivatives of the resident batch's energy: for each of `n_dir` directions
 * (dR [n_dir][n_atoms_total][3] displacements of the atoms, dh [n_dir][n_frames][9] of the cells; either
 * may be NULL; BOTH NULL = unit displacements first .. first + n_dir - 1 of the 3 N, direction d moving
 * atom d / 3 along axis d % 3; `first` is ignored otherwise)
 * the directional derivative of the forces, dF [n_dir][n_atoms_total][3] = d F / d eps = -(H v), and of
 * the virials, dW [n_dir][n_frames][9] (may be NULL). Replaces `tf.hessians(energy, positions)`
 * (nn/basic.py:411-421: Hessian column d = -dF[d]) and the cell derivative of the virial behind the
 * elastic constants (nn/constraint/elastic.py:24-44: dh = unit matrices, dR = NULL). Forward-mode
 * (dual-number) tangents through the analytic force kernels: exact, no step size. Available for EAM and
 * ADP models whose functions are of the Zjw04 / MishinH families, tabulated or networks (nn pair functions through
 * their tables, embedding networks by a second-derivative sweep), and (round 3) for the
 * symmetry-function + MLP models with integer zetas and the GRAP + MLP models (per direction: the descriptors' tangent through the
 * pair Jacobians, the MLP's Hessian-vector product, then the backward expression in dual arithmetic);
 * TA_ERR_UNSUPPORTED otherwise (the caller then differences the analytic forces). At most 65535
 * directions per call. */
int ta_hessian_vectors(ta_handle h, int32_t n_dir, int32_t first, const double *dR, const double *dh, double *dF,
                       double *dW);

/* "nn" pair functions of an EAM / ADP model (rho(r), phi(r), u(r), w(r) as `convolution1x1` networks of
 * the pair distance, nn/eam/eam.py:174-190 — the reference's DEFAULT potentials, alloy.py:110-112):
 * `on` != 0 evaluates them through cubic Hermite tables of 32769 knots over [0, rcut] that the
 * library builds from the networks (value and derivative exact at every knot; rebuilt by
 * ta_update_weights), `on` = 0 evaluates the networks for every pair. Tables are the default for
 * inference (environment TA_EAM_NN_TABLES=0 turns them off at ta_create); they differ from the exact
 * evaluation by ~1e-12 eV per structure, and they are how the reference deploys these potentials
 * itself (`export_to_setfl`, alloy.py:198-381, with a ~30x coarser table). Weight gradients
 * (ta_energy_gradient) need the networks: the first such call switches the handle to exact
 * evaluation for good, after which `on` != 0 is ignored. The embedding networks F(rho) are always
 * exact. No effect on models without nn pair functions. Drops nothing resident: the next
 * ta_compute uses the new mode (call ta_set_frames again if a Verlet skin is in use). */
int ta_set_nn_tables(ta_handle h, int on);

/* Tables of an EAM / ADP model's functions (analytic, nn or tabulated) on caller-supplied abscissae: what
 * `EamAlloyNN.export_to_setfl` (nn/eam/alloy.py:198-381) evaluates through a TF session before it
 * writes a LAMMPS setfl file. Rows: elements (sorted) for rho(r) [n_elements][n_r] and F(rho)
 * [n_elements][n_rho]; element pairs a <= b (upper triangle, row-major) for phi(r), and for an
 * ADP model u(r), w(r) (may be NULL), each [n_pairs][n_r]. Evaluated by the same device functions
 * the energy kernels use. */
int ta_eam_tabulate(ta_handle h, int32_t n_r, const double *r, int32_t n_rho, const double *rho,
                    double *rho_of_r, double *phi_of_r, double *embed_of_rho, double *u_of_r,
                    double *w_of_r);

/* --- measurement and diagnostics ------------------------------------------------------------------
 * Used by bench.py, scripts/ and tests/ only; NOT part of the drop-in path (nothing in
 * tensoralloy_amd/calculator.py or transformer/ calls them, and the reference has no counterpart). */

/* Measurement (bench.py): runs `warmup` untimed then `steps` timed passes of
 * ta_compute over the resident batch, timed with HIP events on the handle's
 * stream. total_ms = wall of the `steps` passes; kernel_ms[k] (may be NULL) =
 * average duration per pass of kernel slot k, measured in a second run with
 * events around every launch. */
int ta_time_compute(ta_handle h, uint32_t want, int32_t warmup, int32_t steps,
                    double *total_ms, double *kernel_ms /*[TA_N_KERNEL_SLOTS]*/);

/* Measurement (bench.py, SURVEY 8(d) "achievable-copy figure"): device-to-device copy of `bytes`
 * bytes on the handle's stream, three ways (grid-stride kernel with 16 B per lane and four loads in
 * flight, the same with non-temporal accesses, the runtime's hipMemcpyAsync), each with `reps` timed
 * repetitions after 2 untimed ones; *gbs = the best (bytes read + bytes written) / average time, GB/s. */
int ta_measure_hbm_copy(ta_handle h, int64_t bytes, int32_t reps, double *gbs);

/* Measurement (bench.py, SURVEY 8(d) FP64 figure): number of unordered neighbour pairs {j, k} of the
 * resident batch's centres with r_ij, r_ik and r_jk all below acut, i.e. the triples whose G4 term
 * (sf.py:126-173) is not identically zero; ta_batch_info.n_triples counts all of them. Runs one
 * energy evaluation first (the count reads the pair records). */
int ta_count_contributing_triples(ta_handle h, int64_t *n_contributing);

/* debugging / parity: host copy of the pair list of the resident batch
 * (centre, neighbour, shift[3]) in the library's order. Arrays sized n_pairs. */
int ta_get_pairs(ta_handle h, int32_t *i, int32_t *j, int32_t *shift /*[n][3]*/);

#ifdef __cplusplus
}
#endif
#endif /* TENSORALLOY_AMD_H */
