// EAM / ADP kernels with the analytic Zhou-Johnson-Wadley and Mishin functions.
//
// Replaces, for `custom_potentials = 'zjw04'` (+ 'mishinh' dipole/quadrupole):
//   EamAlloyNN._build_rho_nn        reference nn/eam/alloy.py:128-196
//   EamNN._build_embed_nn           reference nn/eam/eam.py:401-449
//   EamNN._build_phi_nn             reference nn/eam/eam.py:300-362
//   AdpNN._build_dipole_nn          reference nn/eam/adp.py:315-392
//   AdpNN._build_quadrupole_nn      reference nn/eam/adp.py:394-498
//   Zjw04.rho / phi / embed         reference nn/eam/potentials/zjw04.py:187-389
//   Zjw04xc.embed (sigmoid-blended) reference nn/eam/potentials/zjw04.py:440-550 (also Zjw04uxc)
//   Zjw04xcp.phi (own AB constants) reference nn/eam/potentials/zjw04.py:642-696
//   mishin_polar / mishin_cutoff    reference nn/eam/potentials/generic.py:52-84
// and the tf.gradients that give forces and virial (nn/basic.py:277-331).
//
// Two passes over the packed pair list:
//   eam_atom_kernel  16 / 32 / 64 lanes per atom: pair geometry (written to the pair records),
//                    rho_i, sum phi, and (ADP) the
//                    dipole / quadrupole moments per neighbour species; lane 0
//                    applies the embedding function and stores F'(rho_i).
//   eam_force_kernel (plain EAM) forces and per-atom virial of a centre in one pass.
//   adp_force_kernel (ADP) the same with the dipole / quadrupole terms (moments of j gathered).
//   eam_pair_kernel  (nn pair functions) one lane per directed pair: dE/dD from the centre's F',
//                    moments and the pair functions' derivatives; forces / virial then come from the
//                    same force_gather kernel as the symmetry-function path.
// frame_reduce sums the frames as for the symmetry-function path.
//
// "nn" functions (the reference's default potentials, alloy.py:110-112, adp.py:120-124): rho(r),
// phi(r), u(r), w(r) and F(rho) given by `convolution1x1` on the scalar argument (eam.py:174-190,
// convolutional.py:154-300). They run as 16-row tiles of the fp64 MFMA MLP (ta_mlp_tile.h), whose
// backward sweep to the single input gives f'(x) with the value:
//   eam_nn_pair_kernel   16 consecutive pairs per workgroup: geometry, then one tile pass per
//                        (function class, slot) present among the 16 pairs (pairs are sorted by
//                        centre and neighbour species, so that is one slot per class except at
//                        segment boundaries); f, f' -> per-pair columns `pf`
//   eam_atom_kernel      reads the columns instead of evaluating the analytic function
//   eam_nn_embed_kernel  16 atoms of one element per workgroup: F(rho_i), F'(rho_i)
//   eam_pair_kernel      reads f' from the columns
// Any mix of nn and analytic functions is handled per function by the bit masks of `EamParams`.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ta_device.h"
#include "ta_dual.h"
#include "ta_math.h"
#include "ta_mlp_tile.h"
#include "ta_reduce.h"

namespace ta {

struct EamModel;
void eam_destroy(EamModel *);
// weight gradients (ta_train.hip)
int mlp_param_count(const MlpDev &mlp);
size_t mlp_grad_scratch_doubles(const MlpDev &mlp, int n_atoms);
size_t mlp_grad_partial_doubles(const MlpDev &mlp, int n_atoms);
void launch_mlp_grad_rows(const MlpDev &mlp, int activation, const int32_t *atoms, int n_rows,
                          const double *x, const double *row_coeff, const int32_t *frame_of_atom,
                          const double *frame_coeff, double *scratch, double *partial, double *grad,
                          hipStream_t s);
size_t mlp_grad2_scratch_doubles(const MlpDev &mlp, int n_atoms);
void launch_mlp_grad2_rows(const MlpDev &mlp, int activation, const int32_t *atoms, int n_rows, const double *x,
                           const double *xdot, const double *row_coeff, const int32_t *frame_of_atom,
                           const double *frame_coeff, double *scratch, double *partial, double *grad,
                           hipStream_t s);

namespace {

constexpr int kBlock = 256;
constexpr int kMaxEamElements = 5;  // keeps EamParams + DeviceBatch inside the 4 KB kernarg limit
constexpr int kMaxPairTypes = kMaxEamElements * (kMaxEamElements + 1) / 2;

struct EamParams {
  int nel;
  int adp;
  int embed_kind[kMaxEamElements];  // 0: piecewise (Zjw04), 1: sigmoid-blended (Zjw04xc)
  int el_kind[kMaxEamElements];     // potential of the element's analytic rho / embed / phi_AA:
                                    // 0 Zjw04 family, 1 AgSutton90 (el = {a, b}), 2 AgrawalBe "Be/1"
                                    // (el = {A, B, D, alpha, re, F0, F1, beta, gamma, m, rc}),
                                    // 3 RWGrimes (el = {G, n, A, rho, C, D, gamma, r0})
  int phi_kind[kMaxPairTypes];      // 0: Zjw04 (AA, or AB by density mixing), 1: own constants (Zjw04xcp)
  double el[kMaxEamElements][20];   // ZJW04_KEYS order (tensoralloy_amd/eam.py)
  double phi[kMaxPairTypes][7];     // r_eq A B alpha beta kappa lamda of a Zjw04xcp cross term
  double pair[kMaxPairTypes][8];    // d1 d2 d3 q1 q2 q3 h rc
  // bit k set: the function of element / pair type k is an nn function
  uint32_t nn_rho, nn_embed, nn_phi, nn_u, nn_w;
  // bit k set: tabulated function (natural cubic spline of a setfl / adp table)
  uint32_t tab_rho, tab_embed, tab_phi, tab_u, tab_w;
  // > 0: the resident list is a superset (built with a Verlet skin, ta_set_skin): pairs with
  // r^2 + eps >= list_rc2 are skipped. 0: the list is exact (strict r < rc), no test.
  double list_rc2;
};

// one tabulated function: knots at k dx, (n - 1) cubics {c0, c1, c2, c3} in t = x - x_k
struct TabDev {
  int n;
  double dx, inv_dx;
  const double *c;
};

// natural cubic spline the way the reference's CubicInterpolator evaluates a setfl table
// (potentials/tests/test_mishin.py:60-70); beyond the last knot the last cubic continues
__device__ __forceinline__ void spline_eval(const TabDev &t, double x, double &f, double &df) {
  int k = (int)(x * t.inv_dx);
  k = k < 0 ? 0 : (k > t.n - 2 ? t.n - 2 : k);
  const double tt = x - (double)k * t.dx;
  const double2 *c = reinterpret_cast<const double2 *>(t.c + 4 * (size_t)k);
  const double2 a = c[0], b = c[1];
  f = fma(fma(fma(b.y, tt, b.x), tt, a.y), tt, a.x);
  df = fma(fma(3.0 * b.y, tt, 2.0 * b.x), tt, a.y);
}

// columns of the per-pair function buffer `pf` (each `ps` doubles long)
enum { PF_RHO = 0, PF_DRHO, PF_PHI, PF_DPHI, PF_U, PF_DU, PF_W, PF_DW };

enum { R_EQ, F_EQ, RHO_E, RHO_S, ALPHA, BETA, PA, PB, KAPPA, LAMDA, FN0, FN1, FN2, FN3, F0, F1, F2, F3, ETA, FE };

__device__ __forceinline__ int pair_type(int s1, int s2, int nel) {
  int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
  return a * nel - (a * (a - 1)) / 2 + (b - a);
}

// slot order of the function networks / tables: rho[element], embed[element], phi[pair type],
// dipole[pair type], quadrupole[pair type]
__host__ __device__ __forceinline__ int slot_rho(int e) { return e; }
__host__ __device__ __forceinline__ int slot_embed(int nel, int e) { return nel + e; }
__host__ __device__ __forceinline__ int slot_pair(int nel, int cls /* 1 phi, 2 u, 3 w */, int pt) {
  return 2 * nel + (cls - 1) * (nel * (nel + 1) / 2) + pt;
}

// The analytic functions are templates over the scalar type of the CONSTANTS: double for inference,
// Dual (ta_dual.h) for the gradient with respect to one seeded constant (eam_const_grad_kernel).
// `el`: the per-element constants [nel][20], `phx`: the Zjw04xcp cross terms [pair types][7].

// f(r) = a exp(-b (r/re - 1)) / (1 + (r/re - c)^20)   (generic.py:102-117)
// 1 / re is loop-invariant in every caller (one division per wavefront, not two per pair)
// R = type of the distance: double, or Dual for the second derivatives (forward-mode tangent along a
// displacement direction: df then carries f''(r) r-dot), where the constants are lifted to Dual too
template <typename T, typename R = double>
__device__ __forceinline__ void zhou_exp(R r, T a, T b, T c, T re, T &f, T &df) {
  const T inv_re = 1.0 / re;
  const T x = r * inv_re;
  const T t = x - c;
  const T t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
  const T t20 = t16 * t4, t19 = t16 * t2 * t;
  const T den = 1.0 / (1.0 + t20);
  f = a * t_exp(-b * (x - 1.0)) * den;
  df = f * (-b - 20.0 * t19 * den) * inv_re;
}
// Like pairs of a Zjw04 element: rho(r) and the second term of phi(r) are the SAME function up to the
// prefactor (f_e against B; both use beta, lambda: zjw04.py:229-243, generic.py:102-117), so the pair
// costs two exponentials and two quotients instead of three.
template <typename T>
__device__ __forceinline__ void zjw_rho_phi_aa(const T *p, double r, T &rho, T &drho, T &phi, T &dphi) {
  const T inv_re = 1.0 / p[R_EQ];
  const T x = r * inv_re;
  T fa, dfa;
  {
    const T t = x - p[KAPPA];
    const T t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
    const T den = 1.0 / (1.0 + t16 * t4);
    fa = p[PA] * t_exp(-p[ALPHA] * (x - 1.0)) * den;
    dfa = fa * (-p[ALPHA] - 20.0 * (t16 * t2 * t) * den) * inv_re;
  }
  const T t = x - p[LAMDA];
  const T t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
  const T den = 1.0 / (1.0 + t16 * t4);
  const T core = t_exp(-p[BETA] * (x - 1.0)) * den;
  const T dlog = (-p[BETA] - 20.0 * (t16 * t2 * t) * den) * inv_re;
  const T fb = p[PB] * core;
  rho = p[F_EQ] * core;
  drho = rho * dlog;
  phi = fa - fb;
  dphi = dfa - fb * dlog;
}

template <typename T, typename R = double>
__device__ __forceinline__ void zjw_rho(const T *p, R r, T &f, T &df) {
  zhou_exp<T, R>(r, p[F_EQ], p[BETA], p[LAMDA], p[R_EQ], f, df);
}
template <typename T, typename R = double>
__device__ __forceinline__ void zjw_phi_aa(const T *p, R r, T &f, T &df) {
  T fa, dfa, fb, dfb;
  zhou_exp<T, R>(r, p[PA], p[ALPHA], p[KAPPA], p[R_EQ], fa, dfa);
  zhou_exp<T, R>(r, p[PB], p[BETA], p[LAMDA], p[R_EQ], fb, dfb);
  f = fa - fb;
  df = dfa - dfb;
}
// phi_AB = 0.5 (rho_A/rho_B phi_BB + rho_B/rho_A phi_AA)   (zjw04.py:229-243)
template <typename T, typename R = double>
__device__ __forceinline__ void zjw_phi(const EamParams &P, const T (*el)[20], const T (*phx)[7], int sa, int sb,
                                        R r, T &f, T &df) {
  if (sa == sb) {
    zjw_phi_aa<T, R>(el[sa], r, f, df);
    return;
  }
  const int pt = pair_type(sa, sb, P.nel);
  if (P.phi_kind[pt] == 1) {  // zjw04.py:689-693
    const T *q = phx[pt];
    T fa, dfa, fb, dfb;
    zhou_exp<T, R>(r, q[1], q[3], q[5], q[0], fa, dfa);
    zhou_exp<T, R>(r, q[2], q[4], q[6], q[0], fb, dfb);
    f = fa - fb;
    df = dfa - dfb;
    return;
  }
  T pha, dpha, phb, dphb, ra, dra, rb, drb;
  zjw_phi_aa<T, R>(el[sa], r, pha, dpha);
  zjw_phi_aa<T, R>(el[sb], r, phb, dphb);
  zjw_rho<T, R>(el[sa], r, ra, dra);
  zjw_rho<T, R>(el[sb], r, rb, drb);
  const T q1 = ra / rb, q2 = rb / ra;
  const T dq1 = (dra * rb - ra * drb) / (rb * rb);
  const T dq2 = (drb * ra - rb * dra) / (ra * ra);
  f = 0.5 * (q1 * phb + q2 * pha);
  df = 0.5 * (dq1 * phb + q1 * dphb + dq2 * pha + q2 * dpha);
}

// piecewise embedding energy, thresholds 0.85 rho_e and 1.15 rho_e (zjw04.py:319-386)
template <typename T>
__device__ __forceinline__ void zjw_embed(const T *p, int kind, T rho, T &F, T &dF) {
  const T rho_n = 0.85 * p[RHO_E], rho_0 = 1.15 * p[RHO_E];
  if (kind == 1) {
    // Zjw04xc: the three branches blended by sigmoids of width 1/2 (zjw04.py:482-543)
    const T x1 = rho / rho_n - 1.0;
    const T y1 = p[FN0] + x1 * (p[FN1] + x1 * (p[FN2] + x1 * p[FN3]));
    const T d1 = (p[FN1] + x1 * (2.0 * p[FN2] + 3.0 * p[FN3] * x1)) / rho_n;
    const T x2 = rho / p[RHO_E] - 1.0;
    const T y2 = p[F0] + x2 * (p[F1] + x2 * (p[F2] + x2 * p[F3]));
    const T d2 = (p[F1] + x2 * (2.0 * p[F2] + 3.0 * p[F3] * x2)) / p[RHO_E];
    const T x3 = rho / p[RHO_S] + 1e-8;
    const T lnx = t_log(x3);
    const T xe = t_pow(x3, p[ETA]);
    const T y3 = p[FE] * (1.0 - p[ETA] * lnx) * xe;
    const T d3 = -p[FE] * p[ETA] * p[ETA] * lnx * xe / x3 / p[RHO_S];
    const T c1 = 1.0 / (1.0 + t_exp(-2.0 * (rho_n - rho)));
    const T c3 = 1.0 / (1.0 + t_exp(-2.0 * (rho - rho_0)));
    const T c2 = 1.0 - (c1 + c3);
    const T dc1 = -2.0 * c1 * (1.0 - c1), dc3 = 2.0 * c3 * (1.0 - c3);
    F = c1 * y1 + c2 * y2 + c3 * y3;
    dF = c1 * d1 + c2 * d2 + c3 * d3 + dc1 * y1 - (dc1 + dc3) * y2 + dc3 * y3;
    return;
  }
  if (t_val(rho) < t_val(rho_n)) {
    const T x = rho / rho_n - 1.0;
    F = p[FN0] + x * (p[FN1] + x * (p[FN2] + x * p[FN3]));
    dF = (p[FN1] + x * (2.0 * p[FN2] + 3.0 * p[FN3] * x)) / rho_n;
  } else if (t_val(rho) < t_val(rho_0)) {
    const T x = rho / p[RHO_E] - 1.0;
    F = p[F0] + x * (p[F1] + x * (p[F2] + x * p[F3]));
    dF = (p[F1] + x * (2.0 * p[F2] + 3.0 * p[F3] * x)) / p[RHO_E];
  } else {
    const T x = rho / p[RHO_S];
    const T lnx = t_log(x);
    const T xe = t_pow(x, p[ETA]);
    F = p[FE] * (1.0 - p[ETA] * lnx) * xe;
    dF = -p[FE] * p[ETA] * p[ETA] * lnx * xe / x / p[RHO_S];
  }
}

// ---- the other empirical potentials of the reference's `available_potentials` ---------------------
// AgSutton90 (potentials/sutton90.py:47-100): rho = (a / r)^6, phi = (b / r)^12, F = -sqrt(rho).
// AgrawalBe "Be/1" (potentials/agrawal.py:57-152): with s(r) = (rc / m) (1 - (r / rc)^m),
//   rho = A e^{-B (r - re)} - A e^{-B (rc - re)} - s(r) A B e^{-B (rc - re)},
//   phi = M(r) - M(rc) + s(r) M'(rc),  M = Morse(D, alpha, re) (generic.py:15-30, agrawal.py:20-32),
//   F = F0 (1 - beta ln max(rho, 1e-12)) rho^beta + F1 rho^gamma.
// RWGrimes "grimes" (potentials/grimmes.py:33-100): rho = n / r^8 (1/2 + 1/2 erf(20 (r - 3/2))),
//   phi = Morse(D, gamma, r0) + Buckingham(A, rho, C) = ... + A e^{-r / rho} - C / r^6 (generic.py:15-49),
//   F = -G sqrt(rho).
enum { AG_A = 0, AG_B, AG_D, AG_ALPHA, AG_RE, AG_F0, AG_F1, AG_BETA, AG_GAMMA, AG_M, AG_RC };
enum { GR_G = 0, GR_N, GR_A, GR_RHO, GR_C, GR_D, GR_GAMMA, GR_R0 };

template <typename T, typename R>
__device__ __forceinline__ void morse_fn(R r, T d, T g, T r0, T &f, T &df) {
  const T e1 = t_exp_libm(-g * (r - r0)), e2 = e1 * e1;
  f = d * (e2 - 2.0 * e1);
  df = 2.0 * d * g * (e1 - e2);
}

// OTHER = false: every analytic function of the model is of the Zjw04 family (the kind tests and the
// pow calls of the other potentials stay out of the kernels: -9 % on the plain Zjw04 path otherwise)
template <bool OTHER, typename T>
__device__ __forceinline__ void el_rho(const EamParams &P, const T (*el)[20], int e, double r, T &f, T &df) {
  const T *p = el[e];
  if (!OTHER) {
    zjw_rho<T>(p, r, f, df);
  } else if (P.el_kind[e] == 1) {
    const T t = p[0] / r, t2 = t * t;
    f = t2 * t2 * t2;
    df = -6.0 * f / r;
  } else if (P.el_kind[e] == 2) {
    const T ev = p[AG_A] * t_exp_libm(-p[AG_B] * (r - p[AG_RE]));
    const T ec = p[AG_A] * t_exp_libm(-p[AG_B] * (p[AG_RC] - p[AG_RE]));
    const T x = r / p[AG_RC], xm1 = t_pow(x, p[AG_M] - 1.0);
    f = ev - ec - p[AG_RC] / p[AG_M] * (1.0 - xm1 * x) * p[AG_B] * ec;
    df = -p[AG_B] * ev + xm1 * p[AG_B] * ec;
  } else if (P.el_kind[e] == 3) {
    const double i2 = 1.0 / (r * r), i8 = i2 * i2 * i2 * i2;
    const double t = 20.0 * (r - 1.5);
    const double sw = 0.5 + 0.5 * erf(t);
    const double dsw = 20.0 * 0.56418958354775628 * exp(-t * t);  // 10 * 2 / sqrt(pi) * e^{-t^2}
    f = p[GR_N] * (i8 * sw);
    df = p[GR_N] * (i8 * (dsw - 8.0 * sw / r));
  } else {
    zjw_rho<T>(p, r, f, df);
  }
}

template <bool OTHER, typename T>
__device__ __forceinline__ void pair_phi(const EamParams &P, const T (*el)[20], const T (*phx)[7], int sa, int sb,
                                         double r, T &f, T &df) {
  const int kind = (OTHER && sa == sb) ? P.el_kind[sa] : 0;
  if (!OTHER) {
    zjw_phi<T>(P, el, phx, sa, sb, r, f, df);
  } else if (kind == 1) {
    const T t = el[sa][1] / r, t2 = t * t, t4 = t2 * t2;
    f = t4 * t4 * t4;
    df = -12.0 * f / r;
  } else if (kind == 2) {
    const T *p = el[sa];
    T m0, dm0, mc, dmc;
    morse_fn<T, double>(r, p[AG_D], p[AG_ALPHA], p[AG_RE], m0, dm0);
    morse_fn<T, T>(p[AG_RC], p[AG_D], p[AG_ALPHA], p[AG_RE], mc, dmc);
    const T x = r / p[AG_RC], xm1 = t_pow(x, p[AG_M] - 1.0);
    f = m0 - mc + p[AG_RC] / p[AG_M] * (1.0 - xm1 * x) * dmc;
    df = dm0 - xm1 * dmc;
  } else if (kind == 3) {
    const T *p = el[sa];
    T m0, dm0;
    morse_fn<T, double>(r, p[GR_D], p[GR_GAMMA], p[GR_R0], m0, dm0);
    const T eb = p[GR_A] * t_exp_libm(-r / p[GR_RHO]);
    const double i2 = 1.0 / (r * r), i6 = i2 * i2 * i2;
    f = m0 + eb - p[GR_C] * i6;
    df = dm0 - eb / p[GR_RHO] + 6.0 * p[GR_C] * (i6 / r);
  } else {
    zjw_phi<T>(P, el, phx, sa, sb, r, f, df);
  }
}

template <bool OTHER, typename T>
__device__ __forceinline__ void el_embed(const EamParams &P, const T (*el)[20], int e, T rho, T &F, T &dF) {
  const T *p = el[e];
  if (!OTHER) {
    zjw_embed<T>(p, P.embed_kind[e], rho, F, dF);
  } else if (P.el_kind[e] == 1) {
    const T s = t_sqrt(rho);
    F = -s;
    dF = -0.5 / s;
  } else if (P.el_kind[e] == 2) {
    const T L = t_log(t_floor_at(rho, 1e-12));
    const T xb = t_pow(rho, p[AG_BETA] - 1.0), yg = t_pow(rho, p[AG_GAMMA] - 1.0);
    F = p[AG_F0] * (1.0 - p[AG_BETA] * L) * xb * rho + p[AG_F1] * yg * rho;
    dF = -p[AG_F0] * p[AG_BETA] * p[AG_BETA] * L * xb + p[AG_F1] * p[AG_GAMMA] * yg;
  } else if (P.el_kind[e] == 3) {
    const T s = t_sqrt(rho);
    F = -p[GR_G] * s;
    dF = -0.5 * p[GR_G] / s;
  } else {
    zjw_embed<T>(p, P.embed_kind[e], rho, F, dF);
  }
}


// (p1 exp(-p2 r) + p3) psi((r - rc)/h), psi(x) = x^4/(1+x^4) for x < 0  (generic.py:52-84)
template <typename T>
__device__ __forceinline__ void mishin_polar(double r, T p1, T p2, T p3, T rc, T h, T &f, T &df) {
  const T z = (r - rc) / h;
  if (t_val(z) >= 0.0) {
    f = T{};
    df = T{};
    return;
  }
  const T zz = -z, z2 = zz * zz, z4 = z2 * z2;
  const T den = 1.0 / (1.0 + z4);
  const T psi = z4 * den;
  const T dpsi = -4.0 * z2 * zz * den * den / h;
  const T e = t_exp(-p2 * r);
  const T left = p1 * e + p3;
  f = left * psi;
  df = -p1 * p2 * e * psi + left * dpsi;
}

// the same with the distance of type T as well (Hessian-vector products: r is dual, the constants plain)
template <typename T>
__device__ __forceinline__ void mishin_polar_r(T r, double p1, double p2, double p3, double rc, double h, T &f,
                                               T &df) {
  const T z = (r - rc) / h;
  if (t_val(z) >= 0.0) {
    f = T{};
    df = T{};
    return;
  }
  const T zz = -z, z2 = zz * zz, z4 = z2 * z2;
  const T den = 1.0 / (1.0 + z4);
  const T psi = z4 * den;
  const T dpsi = -4.0 * z2 * zz * den * den / h;
  const T e = t_exp(-(p2 * r));
  const T left = p1 * e + p3;
  f = left * psi;
  df = -(p1 * p2) * e * psi + left * dpsi;
}

// moments per (atom, neighbour species): mu[3], Lambda[6] = lambda - (tr lambda / 3) I
// in the order xx yy zz yz xz xy
// W lanes per atom (16: one DPP row, four atoms per wavefront; 64: one wavefront per atom): an atom
// with n neighbours occupies ceil(n / W) W lane slots, so narrow groups waste fewer lanes (n = 90: 96
// slots against 128) while wide ones put more wavefronts in flight for a single small frame.
template <bool OTHER, int W>
__global__ __launch_bounds__(kBlock) void eam_atom_kernel(EamParams P, DeviceBatch b, double *dF,
                                                          double *mom, double eps,
                                                          const double *__restrict__ pf, size_t ps,
                                                          double *rho_buf, int geom_done,
                                                          const TabDev *__restrict__ tabs) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / W;
  const int lane = threadIdx.x & (W - 1);
  if (i >= b.n_atoms) return;
  const int nel = P.nel;
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  double rho = 0.0, phis = 0.0, eadp = 0.0;
  for (int sb = 0; sb < nel; ++sb) {
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int pt = pair_type(sA, sb, nel);
    const double *pp = P.pair[pt];
    const bool rho_nn = (P.nn_rho >> sb) & 1u, phi_nn = (P.nn_phi >> pt) & 1u;
    const bool u_nn = (P.nn_u >> pt) & 1u, w_nn = (P.nn_w >> pt) & 1u;
    const bool rho_tab = (P.tab_rho >> sb) & 1u, phi_tab = (P.tab_phi >> pt) & 1u;
    const bool u_tab = (P.tab_u >> pt) & 1u, w_tab = (P.tab_w >> pt) & 1u;
    // like pairs of a Zjw04 element: rho and phi share an exponential and a quotient
    const bool fused_aa = !OTHER && sb == sA && !rho_nn && !rho_tab && !phi_nn && !phi_tab;
    // The pair loop is a chain of dependent loads (pair_j -> position of j) in front of ~300
    // instructions, and a group makes several passes: the neighbour index and shift of the pass after
    // next and the neighbour position of the next pass are fetched before this pass is evaluated.
    const int q1 = seg[sb + 1];
    int q = seg[sb] + lane;
    int jn = 0, sn[3] = {0, 0, 0}, j2 = 0, s2[3] = {0, 0, 0};
    double pn[3] = {0.0, 0.0, 0.0};
    if (geom_done != 1) {
      if (q < q1) {
        jn = b.pair_j[q];
        for (int c = 0; c < 3; ++c) sn[c] = b.pair_shift[3 * (size_t)q + c];
      }
      if (q + W < q1) {
        j2 = b.pair_j[q + W];
        for (int c = 0; c < 3; ++c) s2[c] = b.pair_shift[3 * (size_t)(q + W) + c];
      }
      if (q < q1)
        for (int c = 0; c < 3; ++c) pn[c] = b.pos[3 * (size_t)jn + c];
    }
    for (; q < q1; q += W) {
      // pair geometry D = Rj - Ri + S.h, r^2 = D.D + eps (universal.py:448-474), computed here and
      // left in the pair record for the pair kernel and the force gather
      double rec[5];
      if (geom_done == 1) {  // eam_geom_kernel has been here
        const double2 *src = pair_geom(b, (size_t)q);
        const double2 a = src[0], c = src[1];
        rec[0] = a.x;
        rec[1] = a.y;
        rec[2] = c.x;
        rec[3] = c.y;
      } else {
        const double rj[3] = {pn[0], pn[1], pn[2]};
        const double sx = (double)sn[0], sy = (double)sn[1], sz = (double)sn[2];
        if (q + W < q1) {
          for (int c = 0; c < 3; ++c) pn[c] = b.pos[3 * (size_t)j2 + c];
          for (int c = 0; c < 3; ++c) sn[c] = s2[c];
        }
        if (q + 2 * W < q1) {
          j2 = b.pair_j[q + 2 * W];
          for (int c = 0; c < 3; ++c) s2[c] = b.pair_shift[3 * (size_t)(q + 2 * W) + c];
        }
        const double *h = b.cells + 9 * (size_t)b.frame_of_atom[i];
        const double *ri = b.pos + 3 * (size_t)i;
        rec[0] = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
        rec[1] = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
        rec[2] = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
        rec[3] = rec[0] * rec[0] + rec[1] * rec[1] + rec[2] * rec[2] + eps;
        rec[4] = 1.0 / sqrt(rec[3]);
        if (geom_done == 2) {
          // no record: the one-pass force kernels recompute D from the neighbour's position and the
          // shift (positions stay in L2; 16 bytes of indices per pair instead of a 32-byte store here
          // and a 32-byte load there)
        } else if (b.rec4) {  // compact 32-byte record {D, r^2}: the readers recompute 1 / r
          double2 *dst = reinterpret_cast<double2 *>(b.rec4 + 4 * (size_t)q);
          dst[0] = make_double2(rec[0], rec[1]);
          dst[1] = make_double2(rec[2], rec[3]);
        } else {
          double2 *dst = reinterpret_cast<double2 *>(b.rec + kRecDoubles * (size_t)q);
          dst[0] = make_double2(rec[0], rec[1]);
          dst[1] = make_double2(rec[2], rec[3]);
          dst[2] = make_double2(rec[4], 0.0);
        }
      }
      if (P.list_rc2 > 0.0 && !(rec[3] < P.list_rc2)) continue;  // beyond rc: not a neighbour
      const double r = sqrt(rec[3]);
      double f, df = 0.0, fp, dfp = 0.0;
      if (fused_aa) {
        zjw_rho_phi_aa<double>(P.el[sb], r, f, df, fp, dfp);
      } else {
        // density function of the NEIGHBOUR's element (alloy.py:176)
        if (rho_nn) f = pf[PF_RHO * ps + q];
        else if (rho_tab) spline_eval(tabs[slot_rho(sb)], r, f, df);
        else el_rho<OTHER, double>(P, P.el, sb, r, f, df);
        if (phi_nn) fp = pf[PF_PHI * ps + q];
        else if (phi_tab) spline_eval(tabs[slot_pair(nel, 1, pt)], r, fp, dfp);
        else pair_phi<OTHER, double>(P, P.el, P.phi, sA, sb, r, fp, dfp);
      }
      rho += f;
      phis += fp;
      if (P.adp) {
        const double dx = rec[0], dy = rec[1], dz = rec[2];
        double u, du, w, dw;
        if (u_nn) u = pf[PF_U * ps + q];
        else if (u_tab) spline_eval(tabs[slot_pair(nel, 2, pt)], r, u, du);
        else mishin_polar<double>(r, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
        if (w_nn) w = pf[PF_W * ps + q];
        else if (w_tab) spline_eval(tabs[slot_pair(nel, 3, pt)], r, w, dw);
        else mishin_polar<double>(r, pp[3], pp[4], pp[5], pp[7], pp[6], w, dw);
        m[0] = fma(u, dx, m[0]);
        m[1] = fma(u, dy, m[1]);
        m[2] = fma(u, dz, m[2]);
        m[3] = fma(w * dx, dx, m[3]);
        m[4] = fma(w * dy, dy, m[4]);
        m[5] = fma(w * dz, dz, m[5]);
        m[6] = fma(w * dy, dz, m[6]);
        m[7] = fma(w * dx, dz, m[7]);
        m[8] = fma(w * dx, dy, m[8]);
      }
    }
    if (P.adp) {
#pragma unroll
      for (int k = 0; k < 9; ++k) m[k] = group_sum<W>(m[k]);
      if (lane == 0) {
        const double nu = m[3] + m[4] + m[5];
        // 1/2 |mu|^2 + 1/2 sum_ab lambda_ab^2 - 1/6 (tr lambda)^2, per k-body term (adp.py:371-392, :458-492)
        eadp += 0.5 * (m[0] * m[0] + m[1] * m[1] + m[2] * m[2]) +
                0.5 * (m[3] * m[3] + m[4] * m[4] + m[5] * m[5] +
                       2.0 * (m[6] * m[6] + m[7] * m[7] + m[8] * m[8])) -
                nu * nu / 6.0;
        double *dst = mom + ((size_t)i * nel + sb) * 9;
        dst[0] = m[0];
        dst[1] = m[1];
        dst[2] = m[2];
        dst[3] = m[3] - nu / 3.0;
        dst[4] = m[4] - nu / 3.0;
        dst[5] = m[5] - nu / 3.0;
        dst[6] = m[6];
        dst[7] = m[7];
        dst[8] = m[8];
      }
    }
  }
  rho = group_sum<W>(rho);
  phis = group_sum<W>(phis);
  if (lane == 0) {
    if ((P.nn_embed >> sA) & 1u) {  // F(rho) is added by eam_nn_embed_kernel
      rho_buf[i] = rho;
      b.eatom[i] = 0.5 * phis + eadp;
    } else {
      double F, d;
      if ((P.tab_embed >> sA) & 1u) spline_eval(tabs[slot_embed(nel, sA)], rho, F, d);
      else el_embed<OTHER, double>(P, P.el, sA, rho, F, d);
      b.eatom[i] = F + 0.5 * phis + eadp;  // eam.py:353-355, :568
      dF[i] = d;
    }
  }
}

// ---- nn functions ------------------------------------------------------------------------
// pair geometry for the nn path: D = Rj - Ri + S.h, r^2 = D.D + eps (universal.py:448-474) into the
// pair records, r into `rbuf`; one lane per pair
__global__ __launch_bounds__(kBlock) void eam_geom_kernel(DeviceBatch b, double eps, double *rbuf) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= b.n_pairs) return;
  const int i = b.pair_i[q], j = b.pair_j[q];
  const double *h = b.cells + 9 * (size_t)b.frame_of_atom[i];
  const double sx = (double)b.pair_shift[3 * (size_t)q], sy = (double)b.pair_shift[3 * (size_t)q + 1],
               sz = (double)b.pair_shift[3 * (size_t)q + 2];
  const double *ri = b.pos + 3 * (size_t)i, *rj = b.pos + 3 * (size_t)j;
  const double dx = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
  const double dy = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
  const double dz = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
  const double r2 = dx * dx + dy * dy + dz * dz + eps;
  const double r = sqrt(r2);
  if (b.rec4) {
    double2 *dst = reinterpret_cast<double2 *>(b.rec4 + 4 * (size_t)q);
    dst[0] = make_double2(dx, dy);
    dst[1] = make_double2(dz, r2);
  } else {
    double2 *dst = reinterpret_cast<double2 *>(b.rec + kRecDoubles * (size_t)q);
    dst[0] = make_double2(dx, dy);
    dst[1] = make_double2(dz, r2);
    dst[2] = make_double2(1.0 / r, 0.0);
  }
  rbuf[q] = r;
}

// nn functions the launch evaluates: (class, element or pair type), one per blockIdx.y
struct NnFnList {
  int n;
  int8_t cls[3 * kMaxPairTypes + kMaxEamElements];
  int8_t k[3 * kMaxPairTypes + kMaxEamElements];
};

// Fast path for the usual shape 1 -> H1 -> H2 -> 1 (Defaults.hidden_sizes = [64, 32]): one
// wavefront evaluates f and f' for 16 pairs without leaving its registers.
//   layer 1 has K = 1: lane (m = l & 15, kq = l >> 4) forms h1 = act(w1[k] r_m + b1[k]) and
//     h1' = act'(.) w1[k] for k = 4 kk + kq itself -- exactly the A operand of the next MFMA;
//   layer 2: [h1; h1'] . W2 as v_mfma_f64_16x16x4_f64 (value and derivative share the B operand,
//     read from the workgroup's LDS copy of W2; row stride = 16 mod 32 doubles: conflict-free);
//   layer 3 has N = 1: h2 . w3 is a 16-lane DPP row sum of the accumulator columns.
// The four wavefronts of a workgroup share one function's weights in LDS and stride over tiles.
template <int ACT, int NT>
__global__ __launch_bounds__(kBlock) void eam_nn_pair_fast_kernel(EamParams P, const MlpDev *__restrict__ nets,
                                                                  int act_rt, NnFnList fl, DeviceBatch b,
                                                                  const double *__restrict__ rbuf,
                                                                  double *pf, size_t ps) {
  extern __shared__ double lds[];
  const int act = ACT >= 0 ? ACT : act_rt;
  const int cls = fl.cls[blockIdx.y], k = fl.k[blockIdx.y];
  const int nel = P.nel;
  const MlpDev &net = nets[cls == 0 ? slot_rho(k) : slot_pair(nel, cls, k)];
  const int H1 = net.layer[0].np, H2 = 16 * NT;
  const int s2 = (H2 % 32 == 0) ? H2 + 16 : H2;
  double *w1 = lds, *b1 = w1 + H1, *W2 = b1 + H1, *b2 = W2 + (size_t)H1 * s2, *w3 = b2 + H2;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < H1; idx += kBlock) {
    w1[idx] = net.layer[0].w[idx];  // row 0 of [16][H1]
    b1[idx] = net.layer[0].b[idx];
  }
  for (int idx = tid; idx < H1 * H2; idx += kBlock) {
    const int row = idx / H2, col = idx - row * H2;
    W2[row * s2 + col] = net.layer[1].w[idx];
  }
  for (int idx = tid; idx < H2; idx += kBlock) {
    b2[idx] = net.layer[1].b[idx];
    w3[idx] = net.layer[2].w[(size_t)idx * net.layer[2].np];  // column 0 of [H2][16]
  }
  const double b3 = net.layer[2].b[0];
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, m = lane & 15, kq = lane >> 4;
  const int64_t ntiles = (b.n_pairs + kMlpRows - 1) / kMlpRows;
  double *val = pf + (size_t)(2 * cls) * ps, *der = pf + (size_t)(2 * cls + 1) * ps;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t p = t * kMlpRows + m;
    const bool valid = p < b.n_pairs;
    int match = 0;
    double r = 0.0;
    if (valid) {
      const int sb = b.species[b.pair_j[p]];
      const int key = cls == 0 ? sb : pair_type(b.species[b.pair_i[p]], sb, nel);
      match = key == k;
      r = rbuf[p];
    }
    if (!__any(match)) continue;
    mlp_f64x4 accv[NT], accd[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const double bb = b2[16 * nt + m];
      accv[nt] = {bb, bb, bb, bb};
      accd[nt] = {0.0, 0.0, 0.0, 0.0};
    }
    // four k-steps per trip (H1 is a multiple of 16): four independent activation chains for the
    // scheduler to interleave, then the MFMAs
    for (int kk = 0; kk < H1 / 4; kk += 4) {
      double h[4], dh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ki = 4 * (kk + j) + kq;
        const double wk = w1[ki];
        activation_fn(act, fma(wk, r, b1[ki]), h[j], dh[j]);
        dh[j] *= wk;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ki = 4 * (kk + j) + kq;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const double B = W2[ki * s2 + 16 * nt + m];
          accv[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(h[j], B, accv[nt], 0, 0, 0);
          accd[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(dh[j], B, accd[nt], 0, 0, 0);
        }
      }
    }
    // accv[nt][q] = z2[pair kq + 4 q][unit 16 nt + m]
    double fv[4] = {0.0, 0.0, 0.0, 0.0}, fd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const double wo = w3[16 * nt + m];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double h, dh;
        activation_fn(act, accv[nt][q], h, dh);
        fv[q] = fma(h, wo, fv[q]);
        fd[q] = fma(dh * accd[nt][q], wo, fd[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fv[q] = row16_sum(fv[q]);
      fd[q] = row16_sum(fd[q]);
    }
    // lane m = q (< 4) of row group kq stores pair kq + 4 q
    const int q = m & 3, pr = kq + 4 * q;
    const int ok = __shfl(match, pr, 64);
    const double ov = q == 0 ? fv[0] : q == 1 ? fv[1] : q == 2 ? fv[2] : fv[3];
    const double od = q == 0 ? fd[0] : q == 1 ? fd[1] : q == 2 ? fd[2] : fd[3];
    if (m < 4 && ok) {
      val[t * kMlpRows + pr] = ov + b3;
      der[t * kMlpRows + pr] = od;
    }
  }
}

// One hidden layer, 1 -> H1 -> 1: no GEMM at all. One lane per pair, the weights are LDS broadcasts,
// four independent activation chains per trip.
template <int ACT>
__global__ __launch_bounds__(kBlock) void eam_nn_pair_1h_kernel(EamParams P, const MlpDev *__restrict__ nets,
                                                                int act_rt, NnFnList fl, DeviceBatch b,
                                                                const double *__restrict__ rbuf, double *pf,
                                                                size_t ps) {
  extern __shared__ double lds[];
  const int act = ACT >= 0 ? ACT : act_rt;
  const int cls = fl.cls[blockIdx.y], k = fl.k[blockIdx.y];
  const int nel = P.nel;
  const MlpDev &net = nets[cls == 0 ? slot_rho(k) : slot_pair(nel, cls, k)];
  const int H1 = net.layer[0].np;  // multiple of 16; padded units have zero output weight
  double *w1 = lds, *b1 = w1 + H1, *w2 = b1 + H1;
  for (int idx = threadIdx.x; idx < H1; idx += kBlock) {
    w1[idx] = net.layer[0].w[idx];
    b1[idx] = net.layer[0].b[idx];
    w2[idx] = net.layer[1].w[(size_t)idx * net.layer[1].np];  // column 0 of [H1][16]
  }
  const double b2 = net.layer[1].b[0];
  __syncthreads();
  double *val = pf + (size_t)(2 * cls) * ps, *der = pf + (size_t)(2 * cls + 1) * ps;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < b.n_pairs; p += (int64_t)gridDim.x * kBlock) {
    const int sb = b.species[b.pair_j[p]];
    const int key = cls == 0 ? sb : pair_type(b.species[b.pair_i[p]], sb, nel);
    if (key != k) continue;
    const double r = rbuf[p];
    double fv = b2, fd = 0.0;
    for (int k0 = 0; k0 < H1; k0 += 4) {
      double h[4], dh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) activation_fn(act, fma(w1[k0 + j], r, b1[k0 + j]), h[j], dh[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fv = fma(h[j], w2[k0 + j], fv);
        fd = fma(dh[j] * w1[k0 + j], w2[k0 + j], fd);
      }
    }
    val[p] = fv;
    der[p] = fd;
  }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void eam_nn_pair_kernel(EamParams P, const MlpDev *__restrict__ nets,
                                                              int act, DeviceBatch b,
                                                              const double *__restrict__ rbuf, double *pf,
                                                              size_t ps, int stride) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride, *da = lds + 2 * kMlpRows * stride;
  __shared__ double xr[kMlpRows];
  __shared__ int key_sb[kMlpRows], key_pt[kMlpRows];
  const int nel = P.nel, npt = nel * (nel + 1) / 2;
  const int64_t p0 = (int64_t)blockIdx.x * kMlpRows;
  const int nrows = (int)min((int64_t)kMlpRows, b.n_pairs - p0);
  const int tid = threadIdx.x;
  if (tid < kMlpRows) {
    double r = 0.0;
    int sb = -1, pt = -1;
    if (tid < nrows) {
      const int64_t q = p0 + tid;
      r = rbuf[q];
      sb = b.species[b.pair_j[q]];
      pt = pair_type(b.species[b.pair_i[q]], sb, nel);
    }
    xr[tid] = r;
    key_sb[tid] = sb;
    key_pt[tid] = pt;
  }
  __syncthreads();
  const int ncls = P.adp ? 4 : 2;
  for (int cls = 0; cls < ncls; ++cls) {
    const uint32_t nn = cls == 0 ? P.nn_rho : cls == 1 ? P.nn_phi : cls == 2 ? P.nn_u : P.nn_w;
    if (!nn) continue;
    const int *key = cls == 0 ? key_sb : key_pt;
    const int nk = cls == 0 ? nel : npt;
    for (int k = 0; k < nk; ++k) {
      if (!((nn >> k) & 1u)) continue;
      bool need = false;  // the same for every thread: read from LDS
      for (int row = 0; row < nrows; ++row) need |= key[row] == k;
      if (!need) continue;
      if (tid < kMlpRows) buf0[tid * stride] = xr[tid];
      __syncthreads();
      const MlpDev &net = nets[cls == 0 ? slot_rho(k) : slot_pair(nel, cls, k)];
      double *val = pf + (size_t)(2 * cls) * ps + p0, *der = pf + (size_t)(2 * cls + 1) * ps + p0;
      mlp_tile<16>(
          net, act, 1, nrows, buf0, buf1, stride, da,
          [&](int row, double y) { if (key[row] == k) val[row] = y; },
          [&](int row, int, double d) { if (key[row] == k) der[row] = d; });
    }
  }
}

struct EmbedTiles {
  int32_t tile_start[kMaxEamElements + 1];  // first workgroup of every element (nn embeddings only)
  int32_t elem_start[kMaxEamElements + 1];  // first entry of every element in `elem_atoms`
  int nel;
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void eam_nn_embed_kernel(const MlpDev *__restrict__ nets, EmbedTiles t,
                                                               int act, DeviceBatch b,
                                                               const double *__restrict__ rho_buf,
                                                               double *dF, int stride) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride, *da = lds + 2 * kMlpRows * stride;
  int e = 0;
  while (e + 1 < t.nel && (int)blockIdx.x >= t.tile_start[e + 1]) ++e;
  const int32_t *atoms = b.elem_atoms + t.elem_start[e];
  const int n_atoms = t.elem_start[e + 1] - t.elem_start[e];
  const int a0 = ((int)blockIdx.x - t.tile_start[e]) * kMlpRows;
  const int nrows = min(kMlpRows, n_atoms - a0);
  if (threadIdx.x < kMlpRows)
    buf0[threadIdx.x * stride] = (int)threadIdx.x < nrows ? rho_buf[atoms[a0 + threadIdx.x]] : 0.0;
  __syncthreads();
  mlp_tile<16>(
      nets[slot_embed(t.nel, e)], act, 1, nrows, buf0, buf1, stride, da,
      [&](int row, double y) { b.eatom[atoms[a0 + row]] += y; },  // eam.py:568: y = phi + embed
      [&](int row, int, double d) { dF[atoms[a0 + row]] = d; });
}

template <bool OTHER>
__global__ __launch_bounds__(kBlock) void eam_pair_kernel(EamParams P, DeviceBatch b,
                                                          const double *dF, const double *mom,
                                                          const double *__restrict__ pf, size_t ps,
                                                          const TabDev *__restrict__ tabs) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n_pairs) return;
  const int nel = P.nel;
  const int i = b.pair_i[p];
  const int sA = b.species[i], sa = b.species[b.pair_j[p]];
  const double2 *rec = pair_geom(b, (size_t)p);
  const double2 v0 = rec[0], v1 = rec[1];
  const double dx = v0.x, dy = v0.y, dz = v1.x;
  if (P.list_rc2 > 0.0 && !(v1.y < P.list_rc2)) {  // beyond rc: not a neighbour
    b.g[4 * (size_t)p] = 0.0;
    b.g[4 * (size_t)p + 1] = 0.0;
    b.g[4 * (size_t)p + 2] = 0.0;
    return;
  }
  const double r = sqrt(v1.y);
  const double inv_r = 1.0 / r;  // the writer's 1 / sqrt(r^2)
  double f, drho, dphi;
  const int pt = pair_type(sA, sa, nel);
  if ((P.nn_rho >> sa) & 1u) drho = pf[PF_DRHO * ps + p];
  else if ((P.tab_rho >> sa) & 1u) spline_eval(tabs[slot_rho(sa)], r, f, drho);
  else el_rho<OTHER, double>(P, P.el, sa, r, f, drho);
  if ((P.nn_phi >> pt) & 1u) dphi = pf[PF_DPHI * ps + p];
  else if ((P.tab_phi >> pt) & 1u) spline_eval(tabs[slot_pair(nel, 1, pt)], r, f, dphi);
  else pair_phi<OTHER, double>(P, P.el, P.phi, sA, sa, r, f, dphi);
  // dE/dD of the directed pair: the centre's terms only; the reverse pair carries the other half
  double c = (dF[i] * drho + 0.5 * dphi) * inv_r;
  double gx = c * dx, gy = c * dy, gz = c * dz;
  if (P.adp) {
    const double *pp = P.pair[pt];
    const double *m = mom + ((size_t)i * nel + sa) * 9;
    double u, du, w, dw;
    if ((P.nn_u >> pt) & 1u) {
      u = pf[PF_U * ps + p];
      du = pf[PF_DU * ps + p];
    } else if ((P.tab_u >> pt) & 1u) {
      spline_eval(tabs[slot_pair(nel, 2, pt)], r, u, du);
    } else {
      mishin_polar<double>(r, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
    }
    if ((P.nn_w >> pt) & 1u) {
      w = pf[PF_W * ps + p];
      dw = pf[PF_DW * ps + p];
    } else if ((P.tab_w >> pt) & 1u) {
      spline_eval(tabs[slot_pair(nel, 3, pt)], r, w, dw);
    } else {
      mishin_polar<double>(r, pp[3], pp[4], pp[5], pp[7], pp[6], w, dw);
    }
    const double muD = m[0] * dx + m[1] * dy + m[2] * dz;
    const double lx = m[3] * dx + m[8] * dy + m[7] * dz;
    const double ly = m[8] * dx + m[4] * dy + m[6] * dz;
    const double lz = m[7] * dx + m[6] * dy + m[5] * dz;
    const double DLD = dx * lx + dy * ly + dz * lz;
    const double k = (muD * du + DLD * dw) * inv_r;
    gx += k * dx + u * m[0] + 2.0 * w * lx;
    gy += k * dy + u * m[1] + 2.0 * w * ly;
    gz += k * dz + u * m[2] + 2.0 * w * lz;
  }
  b.g[4 * (size_t)p] = gx;
  b.g[4 * (size_t)p + 1] = gy;
  b.g[4 * (size_t)p + 2] = gz;
}

// Plain EAM (no dipole / quadrupole terms, no nn pair functions): forces and per-atom virial in ONE
// pass per centre instead of eam_pair_kernel + force_gather. With g[p] = dE/dD of the directed pair p,
//   F_i = sum_{p in N(i)} (g[p] - g[rev p]),   g[p] = (F'(rho_i) rho'_b(r) + phi'_ab(r) / 2) D / r,
// and the reverse pair has the same r and -D, so g[rev p] = -(F'(rho_j) rho'_a(r) + phi'_ab(r) / 2) D / r:
// it is recomputed from an 8-byte gather of F'(rho_j) instead of a 32-byte random gather of a stored
// g[rev p]; neither g nor the reverse-pair index is touched (eam.py:495-570 differentiated;
// basic.py:277-331). One wavefront per atom, 16 atoms per workgroup (= one record of `bpart`).
// W lanes per atom (see eam_atom_kernel), 16 atoms per workgroup (= one record of `bpart`).
// The geometry of a pair in the one-pass force kernels: from the pair record (`from_pos` = 0) or
// recomputed with the expressions of eam_atom_kernel from the neighbour's position and the shift,
// which are prefetched one pass ahead as raw values (the arithmetic waits for them only when the pass
// that needs them starts).
struct PairFetch {
  double2 n0, n1;  // record, or {x_j, y_j}, {z_j, -}
  int s[3];
};
__device__ __forceinline__ void fetch_pair(const DeviceBatch &b, int from_pos, int q, int j, PairFetch &f) {
  if (from_pos) {
    const double *rj = b.pos + 3 * (size_t)j;
    f.n0 = make_double2(rj[0], rj[1]);
    f.n1 = make_double2(rj[2], 0.0);
#pragma unroll
    for (int c = 0; c < 3; ++c) f.s[c] = b.pair_shift[3 * (size_t)q + c];
  } else {
    const double2 *rec = pair_geom(b, (size_t)q);
    f.n0 = rec[0];
    f.n1 = rec[1];
  }
}
__device__ __forceinline__ void pair_vector(const PairFetch &f, int from_pos, const double *ri, const double *h,
                                            double eps, double &dx, double &dy, double &dz, double &r2) {
  if (from_pos) {
    const double sx = (double)f.s[0], sy = (double)f.s[1], sz = (double)f.s[2];
    dx = (f.n0.x - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
    dy = (f.n0.y - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
    dz = (f.n1.x - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
    r2 = dx * dx + dy * dy + dz * dz + eps;
  } else {
    dx = f.n0.x;
    dy = f.n0.y;
    dz = f.n1.x;
    r2 = f.n1.y;
  }
}

template <bool OTHER, int W>
__global__ __launch_bounds__(16 * W) void eam_force_kernel(EamParams P, DeviceBatch b, const double *dF,
                                                           const TabDev *__restrict__ tabs, int from_pos,
                                                           double eps) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x / W);
  const int lane = threadIdx.x & (W - 1);
  const bool active = i < b.n_atoms;
  const int nel = P.nel;
  double f[3] = {0, 0, 0}, w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (active) {
    const int sA = b.species[i];
    const double dFi = dF[i];
    const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
    const bool rhoA_tab = (P.tab_rho >> sA) & 1u;
    const double *hcell = b.cells + 9 * (size_t)b.frame_of_atom[i];
    const double ri[3] = {b.pos[3 * (size_t)i], b.pos[3 * (size_t)i + 1], b.pos[3 * (size_t)i + 2]};
    for (int sb = 0; sb < nel; ++sb) {
      const int pt = pair_type(sA, sb, nel);
      const bool rhoB_tab = (P.tab_rho >> sb) & 1u, phi_tab = (P.tab_phi >> pt) & 1u;
      // as in eam_atom_kernel: the geometry and F'(rho_j) of the next pass and the neighbour index of the
      // pass after next are fetched before this pass is evaluated
      const int q1 = seg[sb + 1];
      int q = seg[sb] + lane;
      PairFetch nx;
      nx.n0 = nx.n1 = make_double2(0.0, 0.0);
      nx.s[0] = nx.s[1] = nx.s[2] = 0;
      double dFn = 0.0;
      int j2 = 0;
      if (q < q1) {
        const int j = b.pair_j[q];
        fetch_pair(b, from_pos, q, j, nx);
        dFn = dF[j];
      }
      if (q + W < q1) j2 = b.pair_j[q + W];
      for (; q < q1; q += W) {
        const PairFetch cur = nx;
        const double dFj = dFn;
        if (q + W < q1) {
          fetch_pair(b, from_pos, q + W, j2, nx);
          dFn = dF[j2];
        }
        if (q + 2 * W < q1) j2 = b.pair_j[q + 2 * W];
        double2 v0, v1;
        pair_vector(cur, from_pos, ri, hcell, eps, v0.x, v0.y, v1.x, v1.y);
        if (P.list_rc2 > 0.0 && !(v1.y < P.list_rc2)) continue;  // beyond rc: not a neighbour
        const double r = sqrt(v1.y);
        double fn, drhoB, drhoA, dphi;
        if (!OTHER && sb == sA && !rhoB_tab && !phi_tab) {
          zjw_rho_phi_aa<double>(P.el[sb], r, fn, drhoB, fn, dphi);
          drhoA = drhoB;
        } else {
          if (rhoB_tab) spline_eval(tabs[slot_rho(sb)], r, fn, drhoB);
          else el_rho<OTHER, double>(P, P.el, sb, r, fn, drhoB);
          if (sb == sA) drhoA = drhoB;
          else if (rhoA_tab) spline_eval(tabs[slot_rho(sA)], r, fn, drhoA);
          else el_rho<OTHER, double>(P, P.el, sA, r, fn, drhoA);
          if (phi_tab) spline_eval(tabs[slot_pair(nel, 1, pt)], r, fn, dphi);
          else pair_phi<OTHER, double>(P, P.el, P.phi, sA, sb, r, fn, dphi);
        }
        const double inv_r = 1.0 / r;
        const double own = (dFi * drhoB + 0.5 * dphi) * inv_r;   // g[p] = own D
        const double both = own + (dFj * drhoA + 0.5 * dphi) * inv_r;  // g[p] - g[rev p] = both D
        const double d[3] = {v0.x, v0.y, v1.x};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          f[c] = fma(both, d[c], f[c]);
#pragma unroll
          for (int e = 0; e < 3; ++e) w[3 * c + e] = fma(own * d[c], d[e], w[3 * c + e]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) f[k] = group_sum<W>(f[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = group_sum<W>(w[k]);
  if (lane == 0 && active)
    for (int k = 0; k < 3; ++k) b.forces[3 * (size_t)i + k] = f[k];
  block_partials(b, blockIdx.x, i, active, lane == 0, w);
}

// ADP with analytic / tabulated functions: the same one pass per centre. The reverse pair (centre j,
// neighbour i, -D) has the same r and pair type, hence the same u, u', w, w', rho', phi'; what differs
// are the centre's quantities, F'(rho_j) and the moments mu_j, Lambda_j of j for the neighbour species
// of i, a 72-byte gather instead of the 32-byte store and the 64 bytes read back per pair by
// eam_pair_kernel + force_gather (adp.py:315-498 differentiated; basic.py:277-331):
//   g[p]     =  (c_i + k_i) D + u mu_i + 2 w Lambda_i D,   c = (F' rho' + phi' / 2) / r,
//   g[rev p] = -(c_j + k_j) D + u mu_j - 2 w Lambda_j D,   k_i = ( mu_i.D u' + D.Lambda_i.D w') / r,
//                                                          k_j = (-mu_j.D u' + D.Lambda_j.D w') / r.
template <bool OTHER, int W>
__global__ __launch_bounds__(16 * W) void adp_force_kernel(EamParams P, DeviceBatch b, const double *dF,
                                                           const double *__restrict__ mom,
                                                           const TabDev *__restrict__ tabs, int from_pos,
                                                           double eps) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x / W);
  const int lane = threadIdx.x & (W - 1);
  const bool active = i < b.n_atoms;
  const int nel = P.nel;
  double f[3] = {0, 0, 0}, w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (active) {
    const int sA = b.species[i];
    const double dFi = dF[i];
    const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
    const bool rhoA_tab = (P.tab_rho >> sA) & 1u;
    const double *hcell = b.cells + 9 * (size_t)b.frame_of_atom[i];
    const double ri[3] = {b.pos[3 * (size_t)i], b.pos[3 * (size_t)i + 1], b.pos[3 * (size_t)i + 2]};
    for (int sb = 0; sb < nel; ++sb) {
      const int pt = pair_type(sA, sb, nel);
      const double *pp = P.pair[pt];
      const bool rhoB_tab = (P.tab_rho >> sb) & 1u, phi_tab = (P.tab_phi >> pt) & 1u;
      const bool u_tab = (P.tab_u >> pt) & 1u, w_tab = (P.tab_w >> pt) & 1u;
      double mi[9];
      {
        const double *src = mom + ((size_t)i * nel + sb) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) mi[k] = src[k];
      }
      // pipelined as eam_force_kernel: record, F'(rho_j) and the moments of j for the next pass
      const int q1 = seg[sb + 1];
      int q = seg[sb] + lane;
      PairFetch nx;
      nx.n0 = nx.n1 = make_double2(0.0, 0.0);
      nx.s[0] = nx.s[1] = nx.s[2] = 0;
      double dFn = 0.0, mn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      int j2 = 0;
      if (q < q1) {
        const int j = b.pair_j[q];
        fetch_pair(b, from_pos, q, j, nx);
        dFn = dF[j];
        const double *src = mom + ((size_t)j * nel + sA) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) mn[k] = src[k];
      }
      if (q + W < q1) j2 = b.pair_j[q + W];
      for (; q < q1; q += W) {
        const PairFetch cur = nx;
        const double dFj = dFn;
        double mj[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) mj[k] = mn[k];
        if (q + W < q1) {
          fetch_pair(b, from_pos, q + W, j2, nx);
          dFn = dF[j2];
          const double *src = mom + ((size_t)j2 * nel + sA) * 9;
#pragma unroll
          for (int k = 0; k < 9; ++k) mn[k] = src[k];
        }
        if (q + 2 * W < q1) j2 = b.pair_j[q + 2 * W];
        double2 v0, v1;
        pair_vector(cur, from_pos, ri, hcell, eps, v0.x, v0.y, v1.x, v1.y);
        if (P.list_rc2 > 0.0 && !(v1.y < P.list_rc2)) continue;  // beyond rc: not a neighbour
        const double r = sqrt(v1.y);
        double fn, drhoB, drhoA, dphi;
        if (!OTHER && sb == sA && !rhoB_tab && !phi_tab) {
          zjw_rho_phi_aa<double>(P.el[sb], r, fn, drhoB, fn, dphi);
          drhoA = drhoB;
        } else {
          if (rhoB_tab) spline_eval(tabs[slot_rho(sb)], r, fn, drhoB);
          else el_rho<OTHER, double>(P, P.el, sb, r, fn, drhoB);
          if (sb == sA) drhoA = drhoB;
          else if (rhoA_tab) spline_eval(tabs[slot_rho(sA)], r, fn, drhoA);
          else el_rho<OTHER, double>(P, P.el, sA, r, fn, drhoA);
          if (phi_tab) spline_eval(tabs[slot_pair(nel, 1, pt)], r, fn, dphi);
          else pair_phi<OTHER, double>(P, P.el, P.phi, sA, sb, r, fn, dphi);
        }
        double u, du, wq, dw;
        if (u_tab) spline_eval(tabs[slot_pair(nel, 2, pt)], r, u, du);
        else mishin_polar<double>(r, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
        if (w_tab) spline_eval(tabs[slot_pair(nel, 3, pt)], r, wq, dw);
        else mishin_polar<double>(r, pp[3], pp[4], pp[5], pp[7], pp[6], wq, dw);
        const double inv_r = 1.0 / r;
        const double dx = v0.x, dy = v0.y, dz = v1.x;
        // own side
        const double muD = mi[0] * dx + mi[1] * dy + mi[2] * dz;
        const double lx = mi[3] * dx + mi[8] * dy + mi[7] * dz;
        const double ly = mi[8] * dx + mi[4] * dy + mi[6] * dz;
        const double lz = mi[7] * dx + mi[6] * dy + mi[5] * dz;
        const double ci = (dFi * drhoB + 0.5 * dphi) * inv_r +
                          (muD * du + (dx * lx + dy * ly + dz * lz) * dw) * inv_r;
        const double g[3] = {ci * dx + u * mi[0] + 2.0 * wq * lx, ci * dy + u * mi[1] + 2.0 * wq * ly,
                             ci * dz + u * mi[2] + 2.0 * wq * lz};
        // reverse pair: centre j, -D
        const double muDj = mj[0] * dx + mj[1] * dy + mj[2] * dz;
        const double jx = mj[3] * dx + mj[8] * dy + mj[7] * dz;
        const double jy = mj[8] * dx + mj[4] * dy + mj[6] * dz;
        const double jz = mj[7] * dx + mj[6] * dy + mj[5] * dz;
        const double cj = (dFj * drhoA + 0.5 * dphi) * inv_r +
                          (-muDj * du + (dx * jx + dy * jy + dz * jz) * dw) * inv_r;
        const double gr[3] = {-cj * dx + u * mj[0] - 2.0 * wq * jx, -cj * dy + u * mj[1] - 2.0 * wq * jy,
                              -cj * dz + u * mj[2] - 2.0 * wq * jz};
        const double d[3] = {dx, dy, dz};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          f[c] += g[c] - gr[c];
#pragma unroll
          for (int e = 0; e < 3; ++e) w[3 * c + e] = fma(g[c], d[e], w[3 * c + e]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) f[k] = group_sum<W>(f[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = group_sum<W>(w[k]);
  if (lane == 0 && active)
    for (int k = 0; k < 3; ++k) b.forces[3 * (size_t)i + k] = f[k];
  block_partials(b, blockIdx.x, i, active, lane == 0, w);
}

// Tables of the analytic functions on caller-supplied abscissae (setfl / ADP export,
// reference nn/eam/alloy.py:198-381): rows = elements (rho(r), F(rho)) or element pairs a <= b
// (phi, u, w), evaluated by the same device functions the energy kernels use.
__global__ __launch_bounds__(kBlock) void eam_tabulate_kernel(EamParams P, int n_r, const double *r,
                                                              int n_rho, const double *rho,
                                                              double *rho_of_r, double *phi_of_r,
                                                              double *embed_of_rho, double *u_of_r,
                                                              double *w_of_r,
                                                              const TabDev *__restrict__ tabs) {
  const int nel = P.nel, npair = nel * (nel + 1) / 2;
  const int64_t n_rows_r = nel + npair;  // rho rows, then pair rows
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t total_r = n_rows_r * n_r;
  if (idx < total_r) {
    const int row = (int)(idx / n_r), k = (int)(idx % n_r);
    const double x = r[k];
    double f, df;
    if (row < nel) {
      if ((P.tab_rho >> row) & 1u) spline_eval(tabs[slot_rho(row)], x, f, df);
      else el_rho<true, double>(P, P.el, row, x, f, df);
      rho_of_r[(size_t)row * n_r + k] = f;
    } else {
      const int pt = row - nel;
      int a = 0, rem = pt;
      while (rem >= nel - a) {
        rem -= nel - a;
        ++a;
      }
      const int b2 = a + rem;
      if ((P.tab_phi >> pt) & 1u) spline_eval(tabs[slot_pair(nel, 1, pt)], x, f, df);
      else pair_phi<true, double>(P, P.el, P.phi, a, b2, x, f, df);
      phi_of_r[(size_t)pt * n_r + k] = f;
      if (P.adp && u_of_r && w_of_r) {
        const double *pp = P.pair[pt];
        double u, du, w, dw;
        if ((P.tab_u >> pt) & 1u) spline_eval(tabs[slot_pair(nel, 2, pt)], x, u, du);
        else mishin_polar<double>(x, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
        if ((P.tab_w >> pt) & 1u) spline_eval(tabs[slot_pair(nel, 3, pt)], x, w, dw);
        else mishin_polar<double>(x, pp[3], pp[4], pp[5], pp[7], pp[6], w, dw);
        u_of_r[(size_t)pt * n_r + k] = u;
        w_of_r[(size_t)pt * n_r + k] = w;
      }
    }
    return;
  }
  const int64_t j = idx - total_r;
  if (j < (int64_t)nel * n_rho) {
    const int row = (int)(j / n_rho), k = (int)(j % n_rho);
    double F, dF;
    if ((P.tab_embed >> row) & 1u) spline_eval(tabs[slot_embed(nel, row)], rho[k], F, dF);
    else el_embed<true, double>(P, P.el, row, rho[k], F, dF);
    embed_of_rho[(size_t)row * n_rho + k] = F;
  }
}

// ---- weight gradients of the nn functions (training) ------------------------------
// dL/dtheta = sum_f c_f dE_f/dtheta. For a per-pair network f of class rho / phi / u / w the energy
// depends on theta only through f(r_p) of the pairs p it serves, so
//   dL/dtheta = sum_p w_p df(r_p)/dtheta,  w_p = c[frame] dE/df_p:
//   rho: F'(rho_i)   phi: 1/2   u: mu_i . D   w: D . Lambda_i . D     (i = centre of p; eam.py:353-355,
//   adp.py:371-392, :458-492 differentiated). This kernel writes w_p for the pairs of one function
//   (class, k) and 0 for all others; the MLP gradient kernel then runs over all pairs with these
//   weights. Embedding networks: rows = atoms of the element, weight c[frame].
__global__ __launch_bounds__(kBlock) void eam_grad_coeff_kernel(EamParams P, DeviceBatch b, int cls, int k,
                                                                const double *__restrict__ frame_coeff,
                                                                const double *__restrict__ dF,
                                                                const double *__restrict__ mom,
                                                                double *coeff) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n_pairs) return;
  const int nel = P.nel;
  const int i = b.pair_i[p];
  const int sA = b.species[i], sb = b.species[b.pair_j[p]];
  const int key = cls == 0 ? sb : pair_type(sA, sb, nel);
  double w = 0.0;
  const bool listed = !(P.list_rc2 > 0.0) || pair_geom(b, (size_t)p)[1].y < P.list_rc2;
  if (key == k && listed) {
    const double c = frame_coeff[b.frame_of_atom[i]];
    if (cls == 0) {
      w = c * dF[i];
    } else if (cls == 1) {
      w = 0.5 * c;
    } else {
      const double2 *rec = pair_geom(b, (size_t)p);
      const double dx = rec[0].x, dy = rec[0].y, dz = rec[1].x;
      const double *m = mom + ((size_t)i * nel + sb) * 9;
      if (cls == 2) {
        w = c * (m[0] * dx + m[1] * dy + m[2] * dz);
      } else {
        const double lx = m[3] * dx + m[8] * dy + m[7] * dz;
        const double ly = m[8] * dx + m[4] * dy + m[6] * dz;
        const double lz = m[7] * dx + m[6] * dy + m[5] * dz;
        w = c * (dx * lx + dy * ly + dz * lz);
      }
    }
  }
  coeff[p] = w;
}

// table of one nn function on caller-supplied abscissae (setfl export)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void eam_nn_table_kernel(const MlpDev *__restrict__ nets, int slot,
                                                               int act, const double *__restrict__ x,
                                                               int n, double *out, int stride) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride, *da = lds + 2 * kMlpRows * stride;
  const int a0 = blockIdx.x * kMlpRows;
  const int nrows = min(kMlpRows, n - a0);
  if (threadIdx.x < kMlpRows) buf0[threadIdx.x * stride] = (int)threadIdx.x < nrows ? x[a0 + threadIdx.x] : 0.0;
  __syncthreads();
  mlp_tile<16>(
      nets[slot], act, 1, nrows, buf0, buf1, stride, da, [&](int row, double y) { out[a0 + row] = y; },
      [&](int, int, double) {});
}

// Value and derivative of one nn function at the knots k dx of a table (forward + backward sweep of
// the same 16-row tile the exact kernels use): fv[2 k] = f, fv[2 k + 1] = f'
template <int THREADS>
__global__ __launch_bounds__(THREADS) void eam_nn_knots_kernel(const MlpDev *__restrict__ nets, int slot,
                                                               int act, double dx, int n, double *fv,
                                                               int stride) {
  extern __shared__ double lds[];
  double *buf0 = lds, *buf1 = lds + kMlpRows * stride, *da = lds + 2 * kMlpRows * stride;
  const int a0 = blockIdx.x * kMlpRows;
  const int nrows = min(kMlpRows, n - a0);
  if (threadIdx.x < kMlpRows) buf0[threadIdx.x * stride] = (double)(a0 + (int)threadIdx.x) * dx;
  __syncthreads();
  mlp_tile<16>(
      nets[slot], act, 1, nrows, buf0, buf1, stride, da, [&](int row, double y) { fv[2 * (size_t)(a0 + row)] = y; },
      [&](int row, int, double d) { fv[2 * (size_t)(a0 + row) + 1] = d; });
}

// cubic Hermite pieces {c0, c1, c2, c3} in t = x - x_k from the knot values and derivatives
// (the layout `spline_eval` reads): the interpolant matches f and f' at every knot
__global__ __launch_bounds__(kBlock) void hermite_coef_kernel(int n, double dx, const double *__restrict__ fv,
                                                              double *__restrict__ c) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n - 1) return;
  const double f0 = fv[2 * (size_t)k], d0 = fv[2 * (size_t)k + 1], f1 = fv[2 * (size_t)k + 2],
               d1 = fv[2 * (size_t)k + 3];
  const double inv = 1.0 / dx, slope = (f1 - f0) * inv;
  c[4 * (size_t)k] = f0;
  c[4 * (size_t)k + 1] = d0;
  c[4 * (size_t)k + 2] = (3.0 * slope - 2.0 * d0 - d1) * inv;
  c[4 * (size_t)k + 3] = (d0 + d1 - 2.0 * slope) * inv * inv;
}

// ---- analytic second derivatives: Hessian-vector products ------------------------------------------
// d/d eps of the forces (and per-atom virial rows) along a direction (dR, dh) of positions and cells:
// forward-mode tangents (Dual, ta_dual.h) through the reverse-mode force expression of eam_force_kernel,
//   F_i = sum_p [ (F'(rho_i) rho_b'(r) + F'(rho_j) rho_a'(r) + phi'(r)) / r ] D_p,
// with D_p, r, rho', phi' and F' all dual: rho'(r) of a dual r carries rho''(r) r-dot, F'(rho_i) of the
// dual density carries F''(rho_i) rho_i-dot. Replaces the reference's `tf.hessians(E, R)`
// (nn/basic.py:411-421; H v = -dF/d eps) and the cell derivative of the virial behind the elastic
// constants (nn/constraint/elastic.py:24-44), which round 2 took by central differences with a 1e-4 A
// step. Plain EAM models whose functions are of the Zjw04 family or tabulated (setfl tables, nn pair
// functions through their tables); everything else reports "unsupported" and keeps the differences.
__device__ __forceinline__ void spline_eval_dual(const TabDev &t, Dual x, Dual &f, Dual &df) {
  int k = (int)(x.v * t.inv_dx);
  k = k < 0 ? 0 : (k > t.n - 2 ? t.n - 2 : k);
  const double tt = x.v - (double)k * t.dx;
  const double2 *c = reinterpret_cast<const double2 *>(t.c + 4 * (size_t)k);
  const double2 a = c[0], b = c[1];
  const double d1 = fma(fma(3.0 * b.y, tt, 2.0 * b.x), tt, a.y);
  const double d2 = fma(6.0 * b.y, tt, 2.0 * b.x);
  f = make_dual(fma(fma(fma(b.y, tt, b.x), tt, a.y), tt, a.x), d1 * x.d);
  df = make_dual(d1, d2 * x.d);
}

struct HvpArgs {
  int n_dir, unit;     // unit: direction d displaces atom (first + d) / 3 along axis (first + d) % 3 (dR null)
  int first;
  const double *dR;    // [n_dir][N][3] or null
  const double *dh;    // [n_dir][F][9] or null
  double eps;
};

// tangent of the pair vector D = Rj - Ri + S.h along direction `dir`
__device__ __forceinline__ void hvp_pair_tangent(const HvpArgs &a, const DeviceBatch &b, int dir, int64_t i, int j,
                                                 int fr, const int *S, double (&T)[3]) {
  T[0] = T[1] = T[2] = 0.0;
  if (a.unit) {
    const int k = (a.first + dir) / 3, c = (a.first + dir) % 3;
    T[c] = (j == k ? 1.0 : 0.0) - (i == k ? 1.0 : 0.0);
    return;
  }
  if (a.dR) {
    const double *ui = a.dR + ((size_t)dir * b.n_atoms + i) * 3, *uj = a.dR + ((size_t)dir * b.n_atoms + j) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) T[c] = uj[c] - ui[c];
  }
  if (a.dh) {
    const double *g = a.dh + ((size_t)dir * b.n_frames + fr) * 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) T[c] += S[0] * g[c] + S[1] * g[3 + c] + S[2] * g[6 + c];
  }
}

// pass 1: F''(rho_i) rho_i-dot per (direction, atom); one wavefront per atom, grid.y = direction
__global__ __launch_bounds__(kBlock) void eam_hvp_atom_kernel(EamParams P, DeviceBatch b, HvpArgs a,
                                                              const TabDev *__restrict__ tabs, double *dFdot) {
  __shared__ Dual el[kMaxEamElements][20];
  const int nel = P.nel;
  for (int t = threadIdx.x; t < nel * 20; t += kBlock) el[t / 20][t % 20] = make_dual(P.el[t / 20][t % 20]);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, dir = blockIdx.y;
  if (i >= b.n_atoms) return;
  const int fr = b.frame_of_atom[i];
  double *out = dFdot + (size_t)dir * b.n_atoms + i;
  if (a.unit && b.frame_of_atom[(a.first + dir) / 3] != fr) {  // another structure of the batch: no coupling
    if (lane == 0) *out = 0.0;
    return;
  }
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  const double *h = b.cells + 9 * (size_t)fr;
  const double *ri = b.pos + 3 * (size_t)i;
  double rho = 0.0, rhodot = 0.0;
  for (int sb = 0; sb < nel; ++sb) {
    const bool rho_tab = (P.tab_rho >> sb) & 1u;
    for (int q = seg[sb] + lane; q < seg[sb + 1]; q += 64) {
      const int j = b.pair_j[q];
      const int S[3] = {b.pair_shift[3 * (size_t)q], b.pair_shift[3 * (size_t)q + 1], b.pair_shift[3 * (size_t)q + 2]};
      const double *rj = b.pos + 3 * (size_t)j;
      double D[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) D[c] = (rj[c] - ri[c]) + (S[0] * h[c] + S[1] * h[3 + c] + S[2] * h[6 + c]);
      const double r2 = D[0] * D[0] + D[1] * D[1] + D[2] * D[2] + a.eps;
      if (P.list_rc2 > 0.0 && !(r2 < P.list_rc2)) continue;
      const double r = sqrt(r2);
      double T[3];
      hvp_pair_tangent(a, b, dir, i, j, fr, S, T);
      const double rdot = (D[0] * T[0] + D[1] * T[1] + D[2] * T[2]) / r;
      double f, df;
      if (rho_tab) spline_eval(tabs[slot_rho(sb)], r, f, df);
      else zjw_rho<double>(P.el[sb], r, f, df);
      rho += f;
      rhodot = fma(df, rdot, rhodot);
    }
  }
  rho = wave_sum(rho);
  rhodot = wave_sum(rhodot);
  if (lane == 0) {
    if ((P.nn_embed >> sA) & 1u) {
      *out = rhodot;  // an embedding NETWORK: F'' is applied by scalar_net_d2_kernel (eam_hvp)
    } else {
      Dual F, dF;
      if ((P.tab_embed >> sA) & 1u) spline_eval_dual(tabs[slot_embed(nel, sA)], make_dual(rho, 1.0), F, dF);
      else zjw_embed<Dual>(el[sA], P.embed_kind[sA], make_dual(rho, 1.0), F, dF);
      *out = dF.d * rhodot;  // F''(rho_i) rho_i-dot
    }
  }
}

// pass 2: d forces / d eps per (direction, atom) and, when asked for, the per-atom virial rows' tangents
__global__ __launch_bounds__(kBlock) void eam_hvp_force_kernel(EamParams P, DeviceBatch b, HvpArgs a,
                                                               const TabDev *__restrict__ tabs,
                                                               const double *__restrict__ dF,
                                                               const double *__restrict__ dFdot, double *fdot,
                                                               double *wdot) {
  __shared__ Dual el[kMaxEamElements][20];
  __shared__ Dual phx[kMaxPairTypes][7];
  const int nel = P.nel, npt = nel * (nel + 1) / 2;
  for (int t = threadIdx.x; t < nel * 20; t += kBlock) el[t / 20][t % 20] = make_dual(P.el[t / 20][t % 20]);
  for (int t = threadIdx.x; t < npt * 7; t += kBlock) phx[t / 7][t % 7] = make_dual(P.phi[t / 7][t % 7]);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, dir = blockIdx.y;
  if (i >= b.n_atoms) return;
  const int fr = b.frame_of_atom[i];
  double *fo = fdot + ((size_t)dir * b.n_atoms + i) * 3;
  double *wo = wdot ? wdot + ((size_t)dir * b.n_atoms + i) * 9 : nullptr;
  double fd[3] = {0, 0, 0}, wd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!(a.unit && b.frame_of_atom[(a.first + dir) / 3] != fr)) {
    const int sA = b.species[i];
    const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
    const double *h = b.cells + 9 * (size_t)fr;
    const double *ri = b.pos + 3 * (size_t)i;
    const Dual dFi = make_dual(dF[i], dFdot[(size_t)dir * b.n_atoms + i]);
    const bool rhoA_tab = (P.tab_rho >> sA) & 1u;
    for (int sb = 0; sb < nel; ++sb) {
      const int pt = pair_type(sA, sb, nel);
      const bool rhoB_tab = (P.tab_rho >> sb) & 1u, phi_tab = (P.tab_phi >> pt) & 1u;
      for (int q = seg[sb] + lane; q < seg[sb + 1]; q += 64) {
        const int j = b.pair_j[q];
        const int S[3] = {b.pair_shift[3 * (size_t)q], b.pair_shift[3 * (size_t)q + 1], b.pair_shift[3 * (size_t)q + 2]};
        const double *rj = b.pos + 3 * (size_t)j;
        double Dv[3], T[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) Dv[c] = (rj[c] - ri[c]) + (S[0] * h[c] + S[1] * h[3 + c] + S[2] * h[6 + c]);
        const double r2v = Dv[0] * Dv[0] + Dv[1] * Dv[1] + Dv[2] * Dv[2] + a.eps;
        if (P.list_rc2 > 0.0 && !(r2v < P.list_rc2)) continue;
        hvp_pair_tangent(a, b, dir, i, j, fr, S, T);
        const Dual D[3] = {make_dual(Dv[0], T[0]), make_dual(Dv[1], T[1]), make_dual(Dv[2], T[2])};
        const Dual r = t_sqrt(make_dual(r2v, 2.0 * (Dv[0] * T[0] + Dv[1] * T[1] + Dv[2] * T[2])));
        const Dual dFj = make_dual(dF[j], dFdot[(size_t)dir * b.n_atoms + j]);
        Dual fn, drhoB, drhoA, dphi;
        if (rhoB_tab) spline_eval_dual(tabs[slot_rho(sb)], r, fn, drhoB);
        else zjw_rho<Dual, Dual>(el[sb], r, fn, drhoB);
        if (sb == sA) drhoA = drhoB;
        else if (rhoA_tab) spline_eval_dual(tabs[slot_rho(sA)], r, fn, drhoA);
        else zjw_rho<Dual, Dual>(el[sA], r, fn, drhoA);
        if (phi_tab) spline_eval_dual(tabs[slot_pair(nel, 1, pt)], r, fn, dphi);
        else zjw_phi<Dual, Dual>(P, el, phx, sA, sb, r, fn, dphi);
        const Dual inv_r = 1.0 / r;
        const Dual own = (dFi * drhoB + 0.5 * dphi) * inv_r;              // g[p] = own D
        const Dual both = own + (dFj * drhoA + 0.5 * dphi) * inv_r;       // g[p] - g[rev p] = both D
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          fd[c] += (both * D[c]).d;
          if (wo) {
            const Dual od = own * D[c];
#pragma unroll
            for (int e = 0; e < 3; ++e) wd[3 * c + e] += (od * D[e]).d;
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) fd[k] = wave_sum(fd[k]);
  if (wo) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wd[k] = wave_sum(wd[k]);
  }
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) fo[k] = fd[k];
    if (wo)
      for (int k = 0; k < 9; ++k) wo[k] = wd[k];
  }
}

// ---- the same for ADP models ----------------------------------------------------------------------
// adp_force_kernel's expression in dual arithmetic: besides D, r, rho', phi', F' the dipole / quadrupole
// functions u, w (u', w' of a dual r carry u'' rdot, w'' rdot: mishin_polar_r, or the spline's second
// derivative) and the moments of BOTH atoms of a pair are dual, mu = (mu, mu-dot), Lambda = (Lambda,
// Lambda-dot) with mu-dot = sum (u' rdot D + u T), lambda-dot = sum (w' rdot D (x) D + w (T (x) D + D (x) T))
// from the first pass (trace removed as in `mom`).
__global__ __launch_bounds__(kBlock) void adp_hvp_atom_kernel(EamParams P, DeviceBatch b, HvpArgs a,
                                                              const TabDev *__restrict__ tabs, double *dFdot,
                                                              double *momdot) {
  __shared__ Dual el[kMaxEamElements][20];
  const int nel = P.nel;
  for (int t = threadIdx.x; t < nel * 20; t += kBlock) el[t / 20][t % 20] = make_dual(P.el[t / 20][t % 20]);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, dir = blockIdx.y;
  if (i >= b.n_atoms) return;
  const int fr = b.frame_of_atom[i];
  double *out = dFdot + (size_t)dir * b.n_atoms + i;
  double *mout = momdot + (((size_t)dir * b.n_atoms + i) * nel) * 9;
  if (a.unit && b.frame_of_atom[(a.first + dir) / 3] != fr) {  // another structure of the batch: no coupling
    if (lane == 0) *out = 0.0;
    for (int k = lane; k < nel * 9; k += 64) mout[k] = 0.0;
    return;
  }
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  const double *h = b.cells + 9 * (size_t)fr;
  const double *ri = b.pos + 3 * (size_t)i;
  double rho = 0.0, rhodot = 0.0;
  for (int sb = 0; sb < nel; ++sb) {
    const bool rho_tab = (P.tab_rho >> sb) & 1u;
    const int pt = pair_type(sA, sb, nel);
    const double *pp = P.pair[pt];
    const bool u_tab = (P.tab_u >> pt) & 1u, w_tab = (P.tab_w >> pt) & 1u;
    double md[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = seg[sb] + lane; q < seg[sb + 1]; q += 64) {
      const int j = b.pair_j[q];
      const int S[3] = {b.pair_shift[3 * (size_t)q], b.pair_shift[3 * (size_t)q + 1], b.pair_shift[3 * (size_t)q + 2]};
      const double *rj = b.pos + 3 * (size_t)j;
      double D[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) D[c] = (rj[c] - ri[c]) + (S[0] * h[c] + S[1] * h[3 + c] + S[2] * h[6 + c]);
      const double r2 = D[0] * D[0] + D[1] * D[1] + D[2] * D[2] + a.eps;
      if (P.list_rc2 > 0.0 && !(r2 < P.list_rc2)) continue;
      const double r = sqrt(r2);
      double T[3];
      hvp_pair_tangent(a, b, dir, i, j, fr, S, T);
      const double rd = (D[0] * T[0] + D[1] * T[1] + D[2] * T[2]) / r;
      double f, df;
      if (rho_tab) spline_eval(tabs[slot_rho(sb)], r, f, df);
      else zjw_rho<double>(P.el[sb], r, f, df);
      rho += f;
      rhodot = fma(df, rd, rhodot);
      double u, du, w, dw;
      if (u_tab) spline_eval(tabs[slot_pair(nel, 2, pt)], r, u, du);
      else mishin_polar<double>(r, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
      if (w_tab) spline_eval(tabs[slot_pair(nel, 3, pt)], r, w, dw);
      else mishin_polar<double>(r, pp[3], pp[4], pp[5], pp[7], pp[6], w, dw);
      const double ur = du * rd, wr = dw * rd;
#pragma unroll
      for (int c = 0; c < 3; ++c) md[c] += ur * D[c] + u * T[c];
      md[3] += wr * D[0] * D[0] + 2.0 * w * T[0] * D[0];
      md[4] += wr * D[1] * D[1] + 2.0 * w * T[1] * D[1];
      md[5] += wr * D[2] * D[2] + 2.0 * w * T[2] * D[2];
      md[6] += wr * D[1] * D[2] + w * (T[1] * D[2] + D[1] * T[2]);
      md[7] += wr * D[0] * D[2] + w * (T[0] * D[2] + D[0] * T[2]);
      md[8] += wr * D[0] * D[1] + w * (T[0] * D[1] + D[0] * T[1]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) md[k] = wave_sum(md[k]);
    if (lane == 0) {
      const double nu = md[3] + md[4] + md[5];
      for (int k = 0; k < 9; ++k) mout[sb * 9 + k] = (k >= 3 && k < 6) ? md[k] - nu / 3.0 : md[k];
    }
  }
  rho = wave_sum(rho);
  rhodot = wave_sum(rhodot);
  if (lane == 0) {
    if ((P.nn_embed >> sA) & 1u) {
      *out = rhodot;
    } else {
      Dual F, dF;
      if ((P.tab_embed >> sA) & 1u) spline_eval_dual(tabs[slot_embed(nel, sA)], make_dual(rho, 1.0), F, dF);
      else zjw_embed<Dual>(el[sA], P.embed_kind[sA], make_dual(rho, 1.0), F, dF);
      *out = dF.d * rhodot;
    }
  }
}

__global__ __launch_bounds__(kBlock) void adp_hvp_force_kernel(EamParams P, DeviceBatch b, HvpArgs a,
                                                               const TabDev *__restrict__ tabs,
                                                               const double *__restrict__ dF,
                                                               const double *__restrict__ dFdot,
                                                               const double *__restrict__ mom,
                                                               const double *__restrict__ momdot, double *fdot,
                                                               double *wdot) {
  __shared__ Dual el[kMaxEamElements][20];
  __shared__ Dual phx[kMaxPairTypes][7];
  const int nel = P.nel, npt = nel * (nel + 1) / 2;
  for (int t = threadIdx.x; t < nel * 20; t += kBlock) el[t / 20][t % 20] = make_dual(P.el[t / 20][t % 20]);
  for (int t = threadIdx.x; t < npt * 7; t += kBlock) phx[t / 7][t % 7] = make_dual(P.phi[t / 7][t % 7]);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, dir = blockIdx.y;
  if (i >= b.n_atoms) return;
  const int fr = b.frame_of_atom[i];
  double *fo = fdot + ((size_t)dir * b.n_atoms + i) * 3;
  double *wo = wdot ? wdot + ((size_t)dir * b.n_atoms + i) * 9 : nullptr;
  double fd[3] = {0, 0, 0}, wd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!(a.unit && b.frame_of_atom[(a.first + dir) / 3] != fr)) {
    const int sA = b.species[i];
    const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
    const double *h = b.cells + 9 * (size_t)fr;
    const double *ri = b.pos + 3 * (size_t)i;
    const Dual dFi = make_dual(dF[i], dFdot[(size_t)dir * b.n_atoms + i]);
    const bool rhoA_tab = (P.tab_rho >> sA) & 1u;
    auto moments = [&](int64_t atom, int sp, Dual (&m)[9]) {
      const double *v = mom + ((size_t)atom * nel + sp) * 9;
      const double *d = momdot + (((size_t)dir * b.n_atoms + atom) * nel + sp) * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) m[k] = make_dual(v[k], d[k]);
    };
    for (int sb = 0; sb < nel; ++sb) {
      const int pt = pair_type(sA, sb, nel);
      const double *pp = P.pair[pt];
      const bool rhoB_tab = (P.tab_rho >> sb) & 1u, phi_tab = (P.tab_phi >> pt) & 1u;
      const bool u_tab = (P.tab_u >> pt) & 1u, w_tab = (P.tab_w >> pt) & 1u;
      Dual mi[9];
      moments(i, sb, mi);
      for (int q = seg[sb] + lane; q < seg[sb + 1]; q += 64) {
        const int j = b.pair_j[q];
        const int S[3] = {b.pair_shift[3 * (size_t)q], b.pair_shift[3 * (size_t)q + 1], b.pair_shift[3 * (size_t)q + 2]};
        const double *rj = b.pos + 3 * (size_t)j;
        double Dv[3], T[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) Dv[c] = (rj[c] - ri[c]) + (S[0] * h[c] + S[1] * h[3 + c] + S[2] * h[6 + c]);
        const double r2v = Dv[0] * Dv[0] + Dv[1] * Dv[1] + Dv[2] * Dv[2] + a.eps;
        if (P.list_rc2 > 0.0 && !(r2v < P.list_rc2)) continue;
        hvp_pair_tangent(a, b, dir, i, j, fr, S, T);
        const Dual D[3] = {make_dual(Dv[0], T[0]), make_dual(Dv[1], T[1]), make_dual(Dv[2], T[2])};
        const Dual r = t_sqrt(make_dual(r2v, 2.0 * (Dv[0] * T[0] + Dv[1] * T[1] + Dv[2] * T[2])));
        const Dual dFj = make_dual(dF[j], dFdot[(size_t)dir * b.n_atoms + j]);
        Dual mj[9];
        moments(j, sA, mj);
        Dual fn, drhoB, drhoA, dphi, u, du, w, dw;
        if (rhoB_tab) spline_eval_dual(tabs[slot_rho(sb)], r, fn, drhoB);
        else zjw_rho<Dual, Dual>(el[sb], r, fn, drhoB);
        if (sb == sA) drhoA = drhoB;
        else if (rhoA_tab) spline_eval_dual(tabs[slot_rho(sA)], r, fn, drhoA);
        else zjw_rho<Dual, Dual>(el[sA], r, fn, drhoA);
        if (phi_tab) spline_eval_dual(tabs[slot_pair(nel, 1, pt)], r, fn, dphi);
        else zjw_phi<Dual, Dual>(P, el, phx, sA, sb, r, fn, dphi);
        if (u_tab) spline_eval_dual(tabs[slot_pair(nel, 2, pt)], r, u, du);
        else mishin_polar_r<Dual>(r, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
        if (w_tab) spline_eval_dual(tabs[slot_pair(nel, 3, pt)], r, w, dw);
        else mishin_polar_r<Dual>(r, pp[3], pp[4], pp[5], pp[7], pp[6], w, dw);
        const Dual inv_r = 1.0 / r;
        auto lam = [&](const Dual (&m)[9], Dual (&l)[3]) {
          l[0] = m[3] * D[0] + m[8] * D[1] + m[7] * D[2];
          l[1] = m[8] * D[0] + m[4] * D[1] + m[6] * D[2];
          l[2] = m[7] * D[0] + m[6] * D[1] + m[5] * D[2];
        };
        Dual li[3], lj[3];
        lam(mi, li);
        lam(mj, lj);
        const Dual muD = mi[0] * D[0] + mi[1] * D[1] + mi[2] * D[2];
        const Dual DLD = D[0] * li[0] + D[1] * li[1] + D[2] * li[2];
        const Dual ci = (dFi * drhoB + 0.5 * dphi) * inv_r + (muD * du + DLD * dw) * inv_r;
        const Dual muDj = mj[0] * D[0] + mj[1] * D[1] + mj[2] * D[2];
        const Dual DLDj = D[0] * lj[0] + D[1] * lj[1] + D[2] * lj[2];
        const Dual cj = (dFj * drhoA + 0.5 * dphi) * inv_r + (DLDj * dw - muDj * du) * inv_r;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const Dual g = ci * D[c] + u * mi[c] + 2.0 * (w * li[c]);      // own side, dE/dD of (i -> j)
          const Dual gr = u * mj[c] - cj * D[c] - 2.0 * (w * lj[c]);     // the reverse pair's, centre j, -D
          fd[c] += (g - gr).d;
          if (wo) {
#pragma unroll
            for (int e = 0; e < 3; ++e) wd[3 * c + e] += (g * D[e]).d;
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) fd[k] = wave_sum(fd[k]);
  if (wo) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wd[k] = wave_sum(wd[k]);
  }
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) fo[k] = fd[k];
    if (wo)
      for (int k = 0; k < 9; ++k) wo[k] = wd[k];
  }
}

constexpr int kNetThreads = 256;
// Knots of an nn function's table over [0, rcut]: dx = rcut / 32768 (2e-4 A at rcut = 6.5). Cubic
// Hermite error: value dx^4 / 384 |f''''|, derivative about dx^3 / 72 |f''''| (1e-17 / 1e-13 per unit of
// f''''): far below the 1e-6 eV / 1e-5 eV/A the path is held to, for any network this side of a step
// function. 1 MB per function: the tables of a model stay in L2.
constexpr int kNnTableKnots = 32769;

}  // namespace

struct EamModel {
  EamParams p;
  double *dF = nullptr, *mom = nullptr;
  size_t cap_atoms = 0;
  double eps = 1e-14;
  // nn functions
  int n_slots = 0;
  int activation = TA_ACT_SOFTPLUS;
  std::vector<MlpDev> nets;       // host copies (device pointers inside), n_layers == 0: analytic
  MlpDev *nets_dev = nullptr;
  std::vector<double *> owned;    // device allocations of the weights
  int stride = 0, max_layers = 0; // LDS row stride / deepest network
  bool pair_nets = false, embed_nets = false;
  double *pf = nullptr;           // [8 or 4][cap_pairs] value / derivative columns, then r [cap_pairs]
  size_t cap_pairs = 0;
  TabDev *tabs_dev = nullptr;     // [n_slots] tabulated functions (n == 0: none)
  // training: per-pair weights, scratch and partial sums of the gradient kernels
  double *gcoeff = nullptr, *gscratch = nullptr, *gpartial = nullptr;
  size_t cap_gcoeff = 0, cap_gscratch = 0, cap_gpartial = 0;
  bool fast_1h = false;           // every pair function is 1 -> H1 -> 1 (no-GEMM kernel)
  int fast_nt = 0;                // > 0: every pair function is 1 -> H1 -> 16 fast_nt -> 1 (fast kernel)
  size_t fast_lds = 0;
  NnFnList fns;                   // the pair functions that are nn functions
  double *rho_buf = nullptr;      // [cap_atoms]
  // nn pair functions through tables (inference): see eam_set_nn_tables
  double rcut = 0.0;
  bool tables_on = false, trained = false;
  bool pair_nets_exact = false;
  EamParams p_exact;              // function masks of the exact evaluation
  std::vector<TabDev> tabs_host;  // [n_slots] file tables and, while tables_on, the nn pair functions
  std::vector<double *> nn_coef;  // [n_slots] device coefficients of an nn function's table (or null)
  double *knot_fv = nullptr;      // [2 kNnTableKnots] scratch
};

namespace {

template <typename T>
T *eam_upload(EamModel *e, const std::vector<T> &v) {
  T *d = nullptr;
  if (hipMalloc((void **)&d, std::max<size_t>(1, v.size()) * sizeof(T)) != hipSuccess) throw std::bad_alloc();
  e->owned.push_back(reinterpret_cast<double *>(d));
  if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    throw std::runtime_error("hipMemcpy of the EAM function networks failed");
  return d;
}

int round16(int x) { return (x + 15) / 16 * 16; }

// function networks of the model description -> padded device copies, both orientations
// (same layout as the per-element MLPs, ta_api.hip build_mlp)
void build_nets(EamModel *e, const ta_model_desc *m, int n_slots) {
  if (!m->n_layers || !m->layer_sizes || !m->weights)
    throw std::invalid_argument("n_eam_nets set but the network description is missing");
  if (m->activation < 0 || m->activation > TA_ACT_ELU) throw std::invalid_argument("unknown activation");
  e->activation = m->activation;
  e->n_slots = n_slots;
  e->nets.assign(n_slots, MlpDev());
  const int32_t *sizes = m->layer_sizes;
  const double *wsrc = m->weights;
  for (int sl = 0; sl < n_slots; ++sl) {
    const int L = m->n_layers[sl];
    if (L == 0) continue;
    if (L < 2 || L > kMaxLayers) throw std::domain_error("nn function depth out of range (2..8 layers)");
    if (sizes[0] != 1 || sizes[L] != 1)
      throw std::invalid_argument("an nn function maps one input to one output");
    MlpDev &md = e->nets[sl];
    md.n_layers = L;
    for (int l = 0; l < L; ++l) {
      MlpLayerDev &ly = md.layer[l];
      ly.k = sizes[l];
      ly.n = sizes[l + 1];
      if (ly.k < 1 || ly.n < 1 || ly.k > 512 || ly.n > 512)
        throw std::domain_error("nn function layer width out of range (1..512)");
      ly.kp = round16(ly.k);
      ly.np = round16(ly.n);
      ly.act = (l < L - 1) ? 1 : 0;
      ly.res = (m->use_resnet_dt && l > 0 && l < L - 1 && ly.k == ly.n) ? 1 : 0;
      std::vector<double> w((size_t)ly.kp * ly.np, 0.0), wt((size_t)ly.np * ly.kp, 0.0), bb(ly.np, 0.0);
      for (int k = 0; k < ly.k; ++k)
        for (int n = 0; n < ly.n; ++n) {
          const double v = wsrc[(size_t)k * ly.n + n];
          w[(size_t)k * ly.np + n] = v;
          wt[(size_t)n * ly.kp + k] = v;
        }
      wsrc += (size_t)ly.k * ly.n;
      for (int n = 0; n < ly.n; ++n) bb[n] = wsrc[n];
      wsrc += ly.n;
      ly.w = eam_upload(e, w);
      ly.wt = eam_upload(e, wt);
      ly.b = eam_upload(e, bb);
      md.max_np = std::max(md.max_np, ly.np);
      md.max_kp = std::max(md.max_kp, ly.kp);
    }
    sizes += L + 1;
    e->stride = std::max(e->stride, mlp_stride(md));
    e->max_layers = std::max(e->max_layers, L);
  }
  if ((size_t)(2 + e->max_layers) * kMlpRows * e->stride * sizeof(double) > 150 * 1024)
    throw std::domain_error("nn functions too wide / deep for the LDS tile");
  e->nets_dev = eam_upload(e, e->nets);
}

size_t net_lds_bytes(const EamModel *e) {
  return (size_t)(2 + e->max_layers) * kMlpRows * e->stride * sizeof(double);
}

}  // namespace

void eam_set_nn_tables(EamModel *m, bool on);

EamModel *eam_create(const ta_model_desc *m, std::string &err) {
  const int nel = m->n_elements;
  const bool adp = m->kind == TA_MODEL_EAM_ADP;
  const int npair = nel * (nel + 1) / 2;
  // per element 20 constants + embed kind + potential kind; per pair phi kind + 7 constants; ADP: + 8 per pair
  const int need = nel * 22 + npair * 8 + (adp ? npair * 8 : 0);
  if (!m->eam_params || m->n_eam_params != need) {
    err = "eam_params must hold " + std::to_string(need) + " doubles for this model";
    return nullptr;
  }
  if (!(m->rcut > 0.0)) {
    err = "rcut must be positive";
    return nullptr;
  }
  if (nel > kMaxEamElements) {
    err = "EAM/ADP models support at most 5 elements";
    return nullptr;
  }
  const int n_slots = 2 * nel + npair * (adp ? 3 : 1);
  if (m->n_eam_nets != 0 && m->n_eam_nets != n_slots) {
    err = "n_eam_nets must be 0 or " + std::to_string(n_slots) + " for this model";
    return nullptr;
  }
  const bool have_tables = m->eam_table_n || m->eam_table_dx || m->eam_table_coef;
  if (have_tables && (!m->n_eam_nets || !m->eam_table_n || !m->eam_table_dx || !m->eam_table_coef)) {
    err = "tabulated functions need n_eam_nets, eam_table_n, eam_table_dx and eam_table_coef";
    return nullptr;
  }
  EamModel *e = new EamModel();
  std::memset(&e->p, 0, sizeof(e->p));
  e->rcut = m->rcut;
  e->p.nel = nel;
  e->p.adp = adp ? 1 : 0;
  e->eps = m->eps > 0.0 ? m->eps : 1e-14;
  if (have_tables) {
    try {
      std::vector<TabDev> tabs(n_slots);
      const double *src = m->eam_table_coef;
      for (int sl = 0; sl < n_slots; ++sl) {
        TabDev &t = tabs[sl];
        t.n = m->eam_table_n[sl];
        t.dx = t.inv_dx = 0.0;
        t.c = nullptr;
        if (t.n == 0) continue;
        if (t.n < 3 || !(m->eam_table_dx[sl] > 0.0))
          throw std::invalid_argument("a tabulated function needs at least 3 knots and a positive spacing");
        if (m->n_layers && m->n_layers[sl] != 0)
          throw std::invalid_argument("a function cannot be both tabulated and an nn function");
        t.dx = m->eam_table_dx[sl];
        t.inv_dx = 1.0 / t.dx;
        std::vector<double> c(src, src + 4 * (size_t)(t.n - 1));
        src += 4 * (size_t)(t.n - 1);
        t.c = eam_upload(e, c);
      }
      e->tabs_dev = eam_upload(e, tabs);
      e->tabs_host = tabs;
      for (int k = 0; k < nel; ++k) {
        if (tabs[slot_rho(k)].n) e->p.tab_rho |= 1u << k;
        if (tabs[slot_embed(nel, k)].n) e->p.tab_embed |= 1u << k;
      }
      for (int k = 0; k < npair; ++k) {
        if (tabs[slot_pair(nel, 1, k)].n) e->p.tab_phi |= 1u << k;
        if (adp && tabs[slot_pair(nel, 2, k)].n) e->p.tab_u |= 1u << k;
        if (adp && tabs[slot_pair(nel, 3, k)].n) e->p.tab_w |= 1u << k;
      }
    } catch (const std::exception &ex) {
      err = ex.what();
      eam_destroy(e);
      return nullptr;
    }
  }
  if (m->n_eam_nets && m->n_layers) {
    try {
      build_nets(e, m, n_slots);
    } catch (const std::exception &ex) {
      err = ex.what();
      eam_destroy(e);
      return nullptr;
    }
    for (int k = 0; k < nel; ++k) {
      if (e->nets[slot_rho(k)].n_layers) e->p.nn_rho |= 1u << k;
      if (e->nets[slot_embed(nel, k)].n_layers) e->p.nn_embed |= 1u << k;
    }
    for (int k = 0; k < npair; ++k) {
      if (e->nets[slot_pair(nel, 1, k)].n_layers) e->p.nn_phi |= 1u << k;
      if (adp && e->nets[slot_pair(nel, 2, k)].n_layers) e->p.nn_u |= 1u << k;
      if (adp && e->nets[slot_pair(nel, 3, k)].n_layers) e->p.nn_w |= 1u << k;
    }
    e->pair_nets = e->p.nn_rho || e->p.nn_phi || e->p.nn_u || e->p.nn_w;
    e->embed_nets = e->p.nn_embed != 0;
    // list of the pair functions; the fast kernel applies when all of them are 1 -> H1 -> H2 -> 1
    // with the same padded H2 <= 64 and the LDS image of one function stays below 64 KB
    std::memset(&e->fns, 0, sizeof(e->fns));
    bool fast = true, all_1h = true;
    int h2 = 0;
    size_t lds = 0, lds_1h = 0;
    auto add = [&](int cls, int k) {
      const MlpDev &n = e->nets[cls == 0 ? slot_rho(k) : slot_pair(nel, cls, k)];
      if (!n.n_layers) return;
      e->fns.cls[e->fns.n] = (int8_t)cls;
      e->fns.k[e->fns.n] = (int8_t)k;
      ++e->fns.n;
      all_1h = all_1h && n.n_layers == 2;
      lds_1h = std::max(lds_1h, (size_t)3 * n.layer[0].np * sizeof(double));
      if (n.n_layers != 3 || n.layer[1].res || n.layer[1].np > 64 || (h2 && n.layer[1].np != h2)) {
        fast = false;
        return;
      }
      h2 = n.layer[1].np;
      const int H1 = n.layer[0].np, s2 = (h2 % 32 == 0) ? h2 + 16 : h2;
      lds = std::max(lds, (size_t)(2 * H1 + (size_t)H1 * s2 + 2 * h2 + 2) * sizeof(double));
    };
    for (int k = 0; k < nel; ++k) add(0, k);
    for (int cls = 1; cls < (adp ? 4 : 2); ++cls)
      for (int k = 0; k < npair; ++k) add(cls, k);
    if (fast && e->fns.n && lds <= 64 * 1024 && !getenv("TA_EAM_NN_GENERIC")) {
      e->fast_nt = h2 / 16;
      e->fast_lds = lds;
    } else if (all_1h && e->fns.n && !getenv("TA_EAM_NN_GENERIC")) {
      e->fast_1h = true;
      e->fast_lds = lds_1h;
    }
  }
  for (int k = 0; k < nel; ++k) {
    for (int c = 0; c < 20; ++c) e->p.el[k][c] = m->eam_params[k * 22 + c];
    e->p.embed_kind[k] = m->eam_params[k * 22 + 20] != 0.0 ? 1 : 0;
    const int kind = (int)m->eam_params[k * 22 + 21];
    if (kind < 0 || kind > 3) {
      eam_destroy(e);
      err = "unknown empirical potential kind";
      return nullptr;
    }
    e->p.el_kind[k] = kind;
  }
  const double *pp = m->eam_params + nel * 22;
  for (int k = 0; k < npair; ++k) {
    e->p.phi_kind[k] = pp[k * 8] != 0.0 ? 1 : 0;
    for (int c = 0; c < 7; ++c) e->p.phi[k][c] = pp[k * 8 + 1 + c];
    if (e->p.phi_kind[k] == 1 && !(e->p.phi[k][0] > 0.0)) {
      eam_destroy(e);
      err = "r_eq of a zjw04xcp pair term must be positive";
      return nullptr;
    }
    if (e->p.phi_kind[k] == 0) e->p.phi[k][0] = 1.0;
  }
  pp += npair * 8;
  for (int k = 0; k < npair; ++k) {
    for (int c = 0; c < 8; ++c) e->p.pair[k][c] = adp ? pp[k * 8 + c] : 0.0;
    if (!adp || e->p.pair[k][6] == 0.0) e->p.pair[k][6] = 1.0;  // h: avoid 0/0 for absent terms
  }
  for (int k = 0; k < nel; ++k)
    if (e->p.el_kind[k] == 0 &&
        (!(e->p.el[k][R_EQ] > 0.0) || !(e->p.el[k][RHO_E] > 0.0) || !(e->p.el[k][RHO_S] > 0.0))) {
      eam_destroy(e);
      err = "r_eq, rho_e and rho_s must be positive";
      return nullptr;
    }
  e->p_exact = e->p;
  e->pair_nets_exact = e->pair_nets;
  if (e->pair_nets) {
    const char *env = getenv("TA_EAM_NN_TABLES");
    if (!(env && env[0] == '0')) {
      try {
        eam_set_nn_tables(e, true);
      } catch (const std::exception &ex) {
        err = ex.what();
        eam_destroy(e);
        return nullptr;
      }
    }
  }
  return e;
}

// (Re)tabulate every nn PAIR function (rho, phi, u, w of r; the embedding networks stay exact: one
// evaluation per atom) on kNnTableKnots knots over [0, rcut]; device work on the null stream, synchronous.
static void eam_build_nn_tables(EamModel *m) {
  const int nel = m->p.nel, npair = nel * (nel + 1) / 2;
  const int n = kNnTableKnots;
  const double dx = m->rcut / (double)(n - 1);
  if (!m->knot_fv && hipMalloc((void **)&m->knot_fv, 2 * (size_t)n * sizeof(double)) != hipSuccess) throw std::bad_alloc();
  if ((int)m->tabs_host.size() != m->n_slots) m->tabs_host.assign(m->n_slots, TabDev{0, 0.0, 0.0, nullptr});
  if ((int)m->nn_coef.size() != m->n_slots) m->nn_coef.assign(m->n_slots, nullptr);
  auto build = [&](int slot) {
    if (m->nets[slot].n_layers == 0) return;
    if (!m->nn_coef[slot]) {
      double *c = nullptr;
      if (hipMalloc((void **)&c, 4 * (size_t)(n - 1) * sizeof(double)) != hipSuccess) throw std::bad_alloc();
      m->owned.push_back(c);
      m->nn_coef[slot] = c;
    }
    hipLaunchKernelGGL(eam_nn_knots_kernel<kNetThreads>, dim3((unsigned)((n + kMlpRows - 1) / kMlpRows)),
                       dim3(kNetThreads), net_lds_bytes(m), 0, m->nets_dev, slot, m->activation, dx, n, m->knot_fv,
                       m->stride);
    hipLaunchKernelGGL(hermite_coef_kernel, dim3((unsigned)((n - 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, n,
                       dx, m->knot_fv, m->nn_coef[slot]);
    m->tabs_host[slot] = TabDev{n, dx, 1.0 / dx, m->nn_coef[slot]};
  };
  for (int e = 0; e < nel; ++e) build(slot_rho(e));
  for (int cls = 1; cls < (m->p.adp ? 4 : 2); ++cls)
    for (int pt = 0; pt < npair; ++pt) build(slot_pair(nel, cls, pt));
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess)
    throw std::runtime_error("tabulating the nn functions failed");
  if (!m->tabs_dev) {
    TabDev *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)m->n_slots * sizeof(TabDev)) != hipSuccess) throw std::bad_alloc();
    m->owned.push_back(reinterpret_cast<double *>(d));
    m->tabs_dev = d;
  }
  if (hipMemcpy(m->tabs_dev, m->tabs_host.data(), (size_t)m->n_slots * sizeof(TabDev), hipMemcpyHostToDevice) !=
      hipSuccess)
    throw std::runtime_error("hipMemcpy of the function tables failed");
}

// nn PAIR functions through cubic Hermite tables (`on`) or evaluated exactly per pair (off).
// rho(r), phi(r), u(r), w(r) are functions of one scalar, the pair distance, which a step evaluates
// for ~90 pairs per atom: 96 softplus + sigmoid evaluations and 64 fp64 MFMAs per (pair, function)
// exactly, one 32-byte gather and 7 fused multiply-adds from the table. This is how the reference
// deploys these potentials itself (`export_to_setfl` tabulates them for LAMMPS, alloy.py:198-381); the
// table here is ~30x finer than a setfl file's and carries exact derivatives at the knots. With
// tables the model runs the one-pass kernels of the analytic / tabulated potentials (no per-pair
// columns, no pair records, no force gather). Weight gradients need the networks themselves: a
// handle that trains (ta_energy_gradient) switches to exact evaluation for good.
void eam_set_nn_tables(EamModel *m, bool on) {
  if (!m->pair_nets_exact) return;  // nothing to tabulate
  if (on && m->trained) return;
  if (on) {
    eam_build_nn_tables(m);
    const double keep = m->p.list_rc2;
    m->p = m->p_exact;
    m->p.list_rc2 = keep;
    m->p.tab_rho |= m->p.nn_rho;
    m->p.tab_phi |= m->p.nn_phi;
    m->p.tab_u |= m->p.nn_u;
    m->p.tab_w |= m->p.nn_w;
    m->p.nn_rho = m->p.nn_phi = m->p.nn_u = m->p.nn_w = 0;
    m->pair_nets = false;
  } else {
    const double keep = m->p.list_rc2;
    m->p = m->p_exact;
    m->p.list_rc2 = keep;
    m->pair_nets = m->pair_nets_exact;
  }
  m->tables_on = on;
}
bool eam_nn_tables_on(const EamModel *m) { return m->tables_on; }
void eam_mark_trained(EamModel *m) {
  m->trained = true;
  if (m->tables_on) eam_set_nn_tables(m, false);
}

void eam_destroy(EamModel *m) {
  if (!m) return;
  if (m->dF) (void)hipFree(m->dF);
  if (m->mom) (void)hipFree(m->mom);
  if (m->pf) (void)hipFree(m->pf);
  if (m->rho_buf) (void)hipFree(m->rho_buf);
  if (m->gcoeff) (void)hipFree(m->gcoeff);
  if (m->gscratch) (void)hipFree(m->gscratch);
  if (m->gpartial) (void)hipFree(m->gpartial);
  if (m->knot_fv) (void)hipFree(m->knot_fv);
  for (double *d : m->owned) (void)hipFree(d);
  delete m;
}

void eam_ensure(EamModel *m, const DeviceBatch &b) {
  const size_t n = (size_t)b.n_atoms;
  if (n > m->cap_atoms) {
    if (m->dF) (void)hipFree(m->dF);
    if (m->mom) (void)hipFree(m->mom);
    if (m->rho_buf) (void)hipFree(m->rho_buf);
    m->dF = m->mom = m->rho_buf = nullptr;
    m->cap_atoms = 0;
    const size_t cap = n + n / 8 + 64;
    if (hipMalloc((void **)&m->dF, cap * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&m->mom, cap * (size_t)m->p.nel * 9 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&m->rho_buf, cap * sizeof(double)) != hipSuccess)
      throw std::bad_alloc();
    m->cap_atoms = cap;
  }
  const size_t np = (size_t)b.n_pairs;
  if (m->pair_nets && np > m->cap_pairs) {
    if (m->pf) (void)hipFree(m->pf);
    m->pf = nullptr;
    m->cap_pairs = 0;
    const size_t cap = np + np / 8 + 64;
    if (hipMalloc((void **)&m->pf, cap * ((m->p.adp ? 8 : 4) + 1) * sizeof(double)) != hipSuccess)
      throw std::bad_alloc();
    m->cap_pairs = cap;
  }
}

void eam_tabulate(EamModel *m, int n_r, const double *r, int n_rho, const double *rho, double *rho_of_r,
                  double *phi_of_r, double *embed_of_rho, double *u_of_r, double *w_of_r,
                  hipStream_t s) {
  const int nel = m->p.nel, npair = nel * (nel + 1) / 2;
  const int64_t total = (int64_t)(nel + npair) * n_r + (int64_t)nel * n_rho;
  if (total == 0) return;
  hipLaunchKernelGGL(eam_tabulate_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, s, m->p, n_r, r, n_rho, rho, rho_of_r, phi_of_r, embed_of_rho, u_of_r, w_of_r,
                     m->tabs_dev);
  if (!m->n_slots) return;
  // rows of nn functions: one launch per function over the same abscissae
  const size_t lds = net_lds_bytes(m);
  auto table = [&](int slot, const double *x, int n, double *out) {
    if (!out || n == 0 || m->nets[slot].n_layers == 0) return;
    hipLaunchKernelGGL(eam_nn_table_kernel<kNetThreads>, dim3((unsigned)((n + kMlpRows - 1) / kMlpRows)),
                       dim3(kNetThreads), lds, s, m->nets_dev, slot, m->activation, x, n, out, m->stride);
  };
  for (int e = 0; e < nel; ++e) {
    table(slot_rho(e), r, n_r, rho_of_r + (size_t)e * n_r);
    table(slot_embed(nel, e), rho, n_rho, embed_of_rho + (size_t)e * n_rho);
  }
  for (int pt = 0; pt < npair; ++pt) {
    table(slot_pair(nel, 1, pt), r, n_r, phi_of_r + (size_t)pt * n_r);
    if (m->p.adp && u_of_r && w_of_r) {
      table(slot_pair(nel, 2, pt), r, n_r, u_of_r + (size_t)pt * n_r);
      table(slot_pair(nel, 3, pt), r, n_r, w_of_r + (size_t)pt * n_r);
    }
  }
}

// centre-driven kernels only (atom + force kernel): no per-pair launch that walks [0, n_pairs)
bool eam_is_plain(const EamModel *m) { return !m->p.adp && !m->pair_nets; }

void eam_set_list_cutoff(EamModel *m, double rc) { m->p.list_rc2 = rc > 0.0 ? rc * rc + m->eps : 0.0; }

// ---- training support: parameter vector = the nn slots in order, per layer W [in][out] then b [out] --
int64_t eam_param_count(const EamModel *m) {
  int64_t n = 0;
  for (const MlpDev &net : m->nets)
    if (net.n_layers) n += mlp_param_count(net);
  return n;
}

void eam_update_weights(EamModel *m, const double *flat, int64_t n) {
  if (n != eam_param_count(m)) throw std::invalid_argument("ta_update_weights: wrong number of values");
  const double *src = flat;
  for (MlpDev &md : m->nets) {
    for (int l = 0; l < md.n_layers; ++l) {
      MlpLayerDev &ly = md.layer[l];
      std::vector<double> w((size_t)ly.kp * ly.np, 0.0), wt((size_t)ly.np * ly.kp, 0.0), bb(ly.np, 0.0);
      for (int k = 0; k < ly.k; ++k)
        for (int c = 0; c < ly.n; ++c) {
          const double v = src[(size_t)k * ly.n + c];
          w[(size_t)k * ly.np + c] = v;
          wt[(size_t)c * ly.kp + k] = v;
        }
      src += (size_t)ly.k * ly.n;
      for (int c = 0; c < ly.n; ++c) bb[c] = src[c];
      src += ly.n;
      if (hipMemcpy(ly.w, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(ly.wt, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(ly.b, bb.data(), bb.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("hipMemcpy of the EAM function networks failed");
    }
  }
  if (m->tables_on) eam_build_nn_tables(m);  // the tables follow the weights
}

namespace {
void grow(double *&ptr, size_t &cap, size_t need) {
  if (need <= cap) return;
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
  if (hipMalloc((void **)&ptr, need * sizeof(double)) != hipSuccess) throw std::bad_alloc();
  cap = need;
}
}  // namespace

// grad (device, eam_param_count values) = sum_f frame_coeff[f] dE_f/dtheta; needs the forward pass of
// eam_compute on this batch (dF, moments, rho, pair records) to have run with the current weights
void eam_energy_gradient(EamModel *m, const DeviceBatch &b, const double *frame_coeff, double *grad,
                         hipStream_t s) {
  const int nel = m->p.nel, npair = nel * (nel + 1) / 2;
  const int n_pairs = (int)b.n_pairs;
  size_t scratch = 1, partial = 1;
  for (int sl = 0; sl < m->n_slots; ++sl) {
    const MlpDev &net = m->nets[sl];
    if (!net.n_layers) continue;
    const bool embed = sl >= nel && sl < 2 * nel;
    const int rows = embed ? (int)b.n_atoms : n_pairs;
    scratch = std::max(scratch, mlp_grad_scratch_doubles(net, rows));
    partial = std::max(partial, mlp_grad_partial_doubles(net, rows));
  }
  grow(m->gscratch, m->cap_gscratch, scratch + 8);
  grow(m->gpartial, m->cap_gpartial, partial + 8);
  grow(m->gcoeff, m->cap_gcoeff, (size_t)n_pairs + 8);
  const double *rbuf = m->pf ? m->pf + (size_t)(m->p.adp ? 8 : 4) * m->cap_pairs : nullptr;
  size_t off = 0;
  for (int sl = 0; sl < m->n_slots; ++sl) {
    const MlpDev &net = m->nets[sl];
    if (!net.n_layers) continue;
    double *g = grad + off;
    off += (size_t)mlp_param_count(net);
    if (sl >= nel && sl < 2 * nel) {  // embedding network of element e: rows = its atoms, x = rho_i
      const int e = sl - nel;
      launch_mlp_grad_rows(net, m->activation, b.elem_atoms + b.elem_start[e],
                           b.elem_start[e + 1] - b.elem_start[e], m->rho_buf, nullptr, b.frame_of_atom,
                           frame_coeff, m->gscratch, m->gpartial, g, s);
      continue;
    }
    int cls, k;
    if (sl < nel) {
      cls = 0;
      k = sl;
    } else {
      cls = 1 + (sl - 2 * nel) / npair;
      k = (sl - 2 * nel) % npair;
    }
    if (n_pairs > 0)
      hipLaunchKernelGGL(eam_grad_coeff_kernel, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock),
                         0, s, m->p, b, cls, k, frame_coeff, m->dF, m->mom, m->gcoeff);
    launch_mlp_grad_rows(net, m->activation, nullptr, n_pairs, rbuf, m->gcoeff, nullptr, nullptr, m->gscratch,
                         m->gpartial, g, s);
  }
}

namespace {
// ---- force / stress terms of the loss for the nn functions of a plain EAM model ------------------
// (reference nn/losses.py:285-437 through tf.gradients; round 2 took a central difference of dE/dtheta
// on two displaced copies of every frame.) With the loss written as  L = sum_f c_f E_f + D E,  D E the
// directional derivative of the energy along (dR, dh) (train.py), and
//   E = sum_i F(rho_i) + 1/2 sum_p phi(r_p),   rho_i = sum_p rho(r_p),
//   D E = sum_i F'(rho_i) rhodot_i + 1/2 sum_p phi'(r_p) rdot_p,   rhodot_i = sum_p rho'(r_p) rdot_p,
// every network enters through its value f and its input derivative f' at known points with known
// weights, so dL/dtheta is  sum_rows [a f(x) + b f'(x)]  differentiated with respect to theta:
//   embedding net of element e: rows = its atoms, x = rho_i,  a = c_f,                          b = rhodot_i
//   density net of species s:   rows = pairs to s, x = r_p,    a = c_f F'(rho_i) + F''(rho_i) rhodot_i,  b = F'(rho_i) rdot_p
//   pair net of type t:         rows = pairs of t, x = r_p,    a = c_f / 2,                      b = rdot_p / 2
// ADP adds 1/2 |mu|^2 + 1/2 sum lambda_ab^2 - (tr lambda)^2 / 6 per (atom, neighbour species), mu = sum u(r) D,
// lambda = sum w(r) D (x) D; with T = the tangent of D, mudot = sum (u' rdot D + u T), lambdadot likewise, and
// Lambda = lambda - tr lambda / 3 (what `mom` stores), D E_adp = mu . mudot + Lambda : lambdadot, so
//   dipole net of type t:       a = c_f mu.D + mudot.D + mu.T,                         b = rdot_p mu.D
//   quadrupole net of type t:   a = c_f D.Lambda.D + D.Lambdadot.D + 2 T.Lambda.D,     b = rdot_p D.Lambda.D
// (launch_mlp_grad2_rows: one second-order pass per network). F'' of an embedding NETWORK comes from a
// value / first / second derivative sweep (scalar_net_d2_kernel), of an analytic or tabulated
// embedding function from dual arithmetic as in the Hessian-vector kernels.
constexpr int kNetMaxWidth = 128;

__global__ __launch_bounds__(kBlock) void scalar_net_d2_kernel(MlpDev net, int act, const int32_t *atoms,
                                                               int n_rows, const double *__restrict__ x,
                                                               const double *scale, double *out, size_t dir_stride) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_rows) return;
  const int id = atoms ? atoms[t] : t;
  if (scale) scale += blockIdx.y * dir_stride;  // grid.y = direction (Hessian-vector products); scale may be out
  out += blockIdx.y * dir_stride;
  double v[2][kNetMaxWidth], d1[2][kNetMaxWidth], d2[2][kNetMaxWidth];
  int cur = 0;
  v[0][0] = x[id];
  d1[0][0] = 1.0;
  d2[0][0] = 0.0;
  for (int l = 0; l < net.n_layers; ++l) {
    const MlpLayerDev ly = net.layer[l];
    const int nxt = cur ^ 1;
    for (int n = 0; n < ly.n; ++n) {
      double z = ly.b ? ly.b[n] : 0.0, z1 = 0.0, z2 = 0.0;
      for (int k = 0; k < ly.k; ++k) {
        const double w = ly.w[(size_t)k * ly.np + n];
        z = fma(w, v[cur][k], z);
        z1 = fma(w, d1[cur][k], z1);
        z2 = fma(w, d2[cur][k], z2);
      }
      double h = z, dh = 1.0, d2h = 0.0;
      if (ly.act) activation_fn2(act, z, h, dh, d2h);
      double o = h, o1 = dh * z1, o2 = d2h * z1 * z1 + dh * z2;
      if (ly.res) {
        o += v[cur][n];
        o1 += d1[cur][n];
        o2 += d2[cur][n];
      }
      v[nxt][n] = o;
      d1[nxt][n] = o1;
      d2[nxt][n] = o2;
    }
    cur = nxt;
  }
  out[id] = d2[cur][0] * (scale ? scale[id] : 1.0);
}

// one wavefront per atom: rdot of its pairs, rhodot_i, and F''(rho_i) rhodot_i when the embedding
// function is analytic or tabulated (networks: scalar_net_d2_kernel afterwards)
__global__ __launch_bounds__(kBlock) void eam_lg_atom_kernel(EamParams P, DeviceBatch b, HvpArgs a,
                                                             const TabDev *__restrict__ tabs,
                                                             const double *__restrict__ pf, size_t ps,
                                                             double *rdot, double *rhodot, double *d2F,
                                                             double *momdot) {
  __shared__ Dual el[kMaxEamElements][20];
  const int nel = P.nel;
  for (int t = threadIdx.x; t < nel * 20; t += kBlock) el[t / 20][t % 20] = make_dual(P.el[t / 20][t % 20]);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= b.n_atoms) return;
  const int fr = b.frame_of_atom[i];
  const int sA = b.species[i];
  const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
  const double *h = b.cells + 9 * (size_t)fr;
  const double *ri = b.pos + 3 * (size_t)i;
  double acc = 0.0, rho_sum = 0.0;
  for (int sb = 0; sb < nel; ++sb) {
    const bool rho_nn = (P.nn_rho >> sb) & 1u, rho_tab = (P.tab_rho >> sb) & 1u;
    const int pt = pair_type(sA, sb, nel);
    const double *pp = P.pair[pt];
    const bool u_nn = (P.nn_u >> pt) & 1u, u_tab = (P.tab_u >> pt) & 1u;
    const bool w_nn = (P.nn_w >> pt) & 1u, w_tab = (P.tab_w >> pt) & 1u;
    double md[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // ADP: mu-dot, lambda-dot (xx yy zz yz xz xy) of (i, sb)
    for (int q = seg[sb] + lane; q < seg[sb + 1]; q += 64) {
      const int j = b.pair_j[q];
      const int S[3] = {b.pair_shift[3 * (size_t)q], b.pair_shift[3 * (size_t)q + 1], b.pair_shift[3 * (size_t)q + 2]};
      const double *rj = b.pos + 3 * (size_t)j;
      double D[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) D[c] = (rj[c] - ri[c]) + (S[0] * h[c] + S[1] * h[3 + c] + S[2] * h[6 + c]);
      const double r2 = D[0] * D[0] + D[1] * D[1] + D[2] * D[2] + a.eps;
      if (P.list_rc2 > 0.0 && !(r2 < P.list_rc2)) {
        rdot[q] = 0.0;
        continue;
      }
      const double r = sqrt(r2);
      double T[3];
      hvp_pair_tangent(a, b, 0, i, j, fr, S, T);
      const double rd = (D[0] * T[0] + D[1] * T[1] + D[2] * T[2]) / r;
      rdot[q] = rd;
      double f, df;
      if (rho_nn) {
        f = pf[PF_RHO * ps + q];
        df = pf[PF_DRHO * ps + q];
      } else if (rho_tab) {
        spline_eval(tabs[slot_rho(sb)], r, f, df);
      } else {
        zjw_rho<double>(P.el[sb], r, f, df);
      }
      rho_sum += f;
      acc = fma(df, rd, acc);
      if (P.adp) {  // mu-dot = sum (u' rdot D + u T), lambda-dot = sum (w' rdot D (x) D + w (T (x) D + D (x) T))
        double u, du, w, dw;
        if (u_nn) {
          u = pf[PF_U * ps + q];
          du = pf[PF_DU * ps + q];
        } else if (u_tab) {
          spline_eval(tabs[slot_pair(nel, 2, pt)], r, u, du);
        } else {
          mishin_polar<double>(r, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
        }
        if (w_nn) {
          w = pf[PF_W * ps + q];
          dw = pf[PF_DW * ps + q];
        } else if (w_tab) {
          spline_eval(tabs[slot_pair(nel, 3, pt)], r, w, dw);
        } else {
          mishin_polar<double>(r, pp[3], pp[4], pp[5], pp[7], pp[6], w, dw);
        }
        const double ur = du * rd, wr = dw * rd;
#pragma unroll
        for (int c = 0; c < 3; ++c) md[c] += ur * D[c] + u * T[c];
        md[3] += wr * D[0] * D[0] + 2.0 * w * T[0] * D[0];
        md[4] += wr * D[1] * D[1] + 2.0 * w * T[1] * D[1];
        md[5] += wr * D[2] * D[2] + 2.0 * w * T[2] * D[2];
        md[6] += wr * D[1] * D[2] + w * (T[1] * D[2] + D[1] * T[2]);
        md[7] += wr * D[0] * D[2] + w * (T[0] * D[2] + D[0] * T[2]);
        md[8] += wr * D[0] * D[1] + w * (T[0] * D[1] + D[0] * T[1]);
      }
    }
    if (P.adp) {
#pragma unroll
      for (int k = 0; k < 9; ++k) md[k] = wave_sum(md[k]);
      if (lane == 0) {
        const double nu = md[3] + md[4] + md[5];  // stored like the moments themselves: trace removed
        double *dst = momdot + ((size_t)i * nel + sb) * 9;
        for (int k = 0; k < 9; ++k) dst[k] = (k >= 3 && k < 6) ? md[k] - nu / 3.0 : md[k];
      }
    }
  }
  acc = wave_sum(acc);
  rho_sum = wave_sum(rho_sum);
  if (lane == 0) {
    rhodot[i] = acc;
    if (!((P.nn_embed >> sA) & 1u)) {
      Dual F, dF;
      const double rho = rho_sum;  // (rho_buf holds the densities of the atoms with an embedding NETWORK only)
      if ((P.tab_embed >> sA) & 1u) spline_eval_dual(tabs[slot_embed(nel, sA)], make_dual(rho, 1.0), F, dF);
      else zjw_embed<Dual>(el[sA], P.embed_kind[sA], make_dual(rho, 1.0), F, dF);
      d2F[i] = dF.d * acc;
    }
  }
}

// rows of one per-pair network (cls 0: density of species k, cls 1: pair function of type k): the
// weights a (value) and b (input derivative) of every pair, zero for the pairs it does not serve
__global__ __launch_bounds__(kBlock) void eam_lg_coeff_kernel(EamParams P, DeviceBatch b, int cls, int k,
                                                              const double *__restrict__ frame_coeff,
                                                              const double *__restrict__ dF,
                                                              const double *__restrict__ d2F,
                                                              const double *__restrict__ rdot,
                                                              const double *__restrict__ mom,
                                                              const double *__restrict__ momdot, HvpArgs a,
                                                              double *ca, double *cb) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n_pairs) return;
  const int nel = P.nel;
  const int i = b.pair_i[p], j = b.pair_j[p];
  const int sA = b.species[i], sb = b.species[j];
  const int key = cls == 0 ? sb : pair_type(sA, sb, nel);
  double va = 0.0, vb = 0.0;
  // the pair vector again from the coordinates (the records need not exist in every mode of the forward pass)
  const int fr = b.frame_of_atom[i];
  const double *h = b.cells + 9 * (size_t)fr;
  const int S[3] = {b.pair_shift[3 * (size_t)p], b.pair_shift[3 * (size_t)p + 1], b.pair_shift[3 * (size_t)p + 2]};
  double D[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    D[c] = (b.pos[3 * (size_t)j + c] - b.pos[3 * (size_t)i + c]) + (S[0] * h[c] + S[1] * h[3 + c] + S[2] * h[6 + c]);
  const double r2 = D[0] * D[0] + D[1] * D[1] + D[2] * D[2] + a.eps;
  const bool listed = !(P.list_rc2 > 0.0) || r2 < P.list_rc2;
  if (key == k && listed) {
    const double c = frame_coeff ? frame_coeff[fr] : 0.0;
    if (cls == 0) {
      va = c * dF[i] + d2F[i];
      vb = dF[i] * rdot[p];
    } else if (cls == 1) {
      va = 0.5 * c;
      vb = 0.5 * rdot[p];
    } else {
      double T[3];
      hvp_pair_tangent(a, b, 0, i, j, fr, S, T);
      const double *m = mom + ((size_t)i * nel + sb) * 9, *md = momdot + ((size_t)i * nel + sb) * 9;
      if (cls == 2) {  // dipole function u: E through mu = sum u D
        const double muD = m[0] * D[0] + m[1] * D[1] + m[2] * D[2];
        va = c * muD + (md[0] * D[0] + md[1] * D[1] + md[2] * D[2]) + (m[0] * T[0] + m[1] * T[1] + m[2] * T[2]);
        vb = rdot[p] * muD;
      } else {         // quadrupole function w: E through lambda = sum w D (x) D; m[3..8] = Lambda (trace removed)
        auto quad = [](const double *q, const double *x, const double *y) {
          const double lx = q[3] * y[0] + q[8] * y[1] + q[7] * y[2];
          const double ly = q[8] * y[0] + q[4] * y[1] + q[6] * y[2];
          const double lz = q[7] * y[0] + q[6] * y[1] + q[5] * y[2];
          return x[0] * lx + x[1] * ly + x[2] * lz;
        };
        const double DLD = quad(m, D, D);
        va = c * DLD + quad(md, D, D) + 2.0 * quad(m, T, D);
        vb = rdot[p] * DLD;
      }
    }
  }
  ca[p] = va;
  cb[p] = vb;
}
}  // namespace

bool eam_loss_gradient_supported(const EamModel *m) {
  for (int e = 0; e < m->p.nel; ++e)
    if (m->p.el_kind[e] != 0) return false;
  for (int sl = 0; sl < m->n_slots; ++sl)
    if (m->nets[sl].n_layers && (m->nets[sl].max_np > kNetMaxWidth || m->nets[sl].max_kp > kNetMaxWidth || m->nets[sl].xlo))
      return false;
  return true;
}

// grad (device) = d/dtheta [ sum_f frame_coeff[f] E_f + D_(dR, dh) E ]; needs the forward pass of eam_compute on
// this batch with the current weights (dF, rho, per-pair columns). frame_coeff may be null (no energy term).
void eam_loss_gradient(EamModel *m, const DeviceBatch &b, const double *frame_coeff, const double *dR,
                       const double *dh, double *grad, hipStream_t s) {
  const int nel = m->p.nel, npair = nel * (nel + 1) / 2;
  const int n_pairs = (int)b.n_pairs, n_atoms = (int)b.n_atoms;
  size_t scratch = 1, partial = 1;
  for (int sl = 0; sl < m->n_slots; ++sl) {
    const MlpDev &net = m->nets[sl];
    if (!net.n_layers) continue;
    const bool embed = sl >= nel && sl < 2 * nel;
    const int rows = embed ? n_atoms : n_pairs;
    scratch = std::max(scratch, mlp_grad2_scratch_doubles(net, rows));
    partial = std::max(partial, mlp_grad_partial_doubles(net, rows));
  }
  grow(m->gscratch, m->cap_gscratch, scratch + 8);
  grow(m->gpartial, m->cap_gpartial, partial + 8);
  grow(m->gcoeff, m->cap_gcoeff, 3 * (size_t)n_pairs + (2 + (m->p.adp ? 9 * (size_t)nel : 0)) * (size_t)n_atoms + 8);
  double *ca = m->gcoeff, *cb = ca + n_pairs, *rdot = cb + n_pairs, *rhodot = rdot + n_pairs, *d2F = rhodot + n_atoms;
  double *momdot = d2F + n_atoms;  // ADP: tangents of the moments [n_atoms][nel][9]
  const double *rbuf = m->pf ? m->pf + (size_t)(m->p.adp ? 8 : 4) * m->cap_pairs : nullptr;
  const HvpArgs a{1, 0, 0, dR, dh, m->eps};
  if (n_atoms > 0)
    hipLaunchKernelGGL(eam_lg_atom_kernel, dim3((unsigned)((n_atoms + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock),
                       0, s, m->p, b, a, m->tabs_dev, m->pf, m->cap_pairs, rdot, rhodot, d2F, momdot);
  for (int e = 0; e < nel; ++e) {
    const MlpDev &net = m->nets[slot_embed(nel, e)];
    const int n_el = b.elem_start[e + 1] - b.elem_start[e];
    if (!net.n_layers || n_el == 0) continue;
    hipLaunchKernelGGL(scalar_net_d2_kernel, dim3((unsigned)((n_el + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, net,
                       m->activation, b.elem_atoms + b.elem_start[e], n_el, m->rho_buf, rhodot, d2F, (size_t)0);
  }
  size_t off = 0;
  for (int sl = 0; sl < m->n_slots; ++sl) {
    const MlpDev &net = m->nets[sl];
    if (!net.n_layers) continue;
    double *g = grad + off;
    off += (size_t)mlp_param_count(net);
    if (sl >= nel && sl < 2 * nel) {
      const int e = sl - nel;
      launch_mlp_grad2_rows(net, m->activation, b.elem_atoms + b.elem_start[e], b.elem_start[e + 1] - b.elem_start[e],
                            m->rho_buf, rhodot, nullptr, b.frame_of_atom, frame_coeff, m->gscratch, m->gpartial, g, s);
      continue;
    }
    const int cls = sl < nel ? 0 : 1 + (sl - 2 * nel) / npair;
    const int k = sl < nel ? sl : (sl - 2 * nel) % npair;
    if (n_pairs > 0)
      hipLaunchKernelGGL(eam_lg_coeff_kernel, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                         m->p, b, cls, k, frame_coeff, m->dF, d2F, rdot, m->mom, momdot, a, ca, cb);
    launch_mlp_grad2_rows(net, m->activation, nullptr, n_pairs, rbuf, cb, ca, nullptr, nullptr, m->gscratch,
                          m->gpartial, g, s);
  }
}

namespace {
// ---- gradients with respect to the constants of the analytic functions ---------------------------
// The reference trains the constants of its empirical potentials as `tf.Variable`s
// (potentials/potentials.py:129-163, zjw04.py: every constant a shared variable) under the same
// energy + forces + stress loss (nn/losses.py:204-437). With the loss written as
//   L = sum_f c_f E_f + D_(dR, dh) E
// (train.py: the force and stress terms are the directional derivative of the energy along
// dR = R.Y - u, dh = h.Y), the derivative with respect to ONE constant is the dual part of
//   sum_i  c_f(i) [F(rho_i) + 1/2 sum_j phi(r_ij)] + F'(rho_i) rhodot_i + 1/2 sum_j phi'(r_ij) rdot_ij,
//   rho_i = sum_j rho_{s_j}(r_ij),  rhodot_i = sum_j rho'_{s_j}(r_ij) rdot_ij,  rdot = D . dD / r,
// evaluated in dual arithmetic with that constant seeded: F' of a dual rho_i brings F'' along.
// ADP models add, per neighbour species, mu = sum_j u(r) D, lambda = sum_j w(r) D (x) D and the energy
// 1/2 |mu|^2 + 1/2 sum lambda_ab^2 - (tr lambda)^2 / 6 (adp.py:371-392, :458-492); their directional
// derivatives mudot = sum_j (u' rdot D + u dD), lambdadot = sum_j (w' rdot D (x) D + w (dD (x) D + D (x) dD))
// enter as mu . mudot + sum lambda_ab lambdadot_ab - tr lambda tr lambdadot / 3, all in duals, so the
// constants of the MishinH dipole / quadrupole functions (mishin.py:62-66, :269-315) are covered too.
// One wavefront per atom; blockIdx.y = the seeded constant: e * 20 + k for element e (ZJW04_KEYS
// order), then 20 nel + pt * 7 + q for the Zjw04xcp cross terms, then (ADP) 20 nel + 7 npt + pt * 8 + k
// for d1 d2 d3 q1 q2 q3 h rc of pair type pt.
template <bool OTHER>
__global__ __launch_bounds__(kBlock) void eam_const_grad_kernel(EamParams P, DeviceBatch b,
                                                                const double *__restrict__ frame_coeff,
                                                                const double *__restrict__ dR,
                                                                const double *__restrict__ dh, double eps,
                                                                const TabDev *__restrict__ tabs,
                                                                const double *__restrict__ pf, size_t ps,
                                                                const double *__restrict__ dFv,
                                                                const double *__restrict__ d2Fv, double *partial) {
  // Mixed models (round 3): a function that is a network or a table has no constants; it enters as plain
  // numbers, value and derivative from the exact forward pass (`pf` columns) or the spline, and an embedding
  // network through F'(rho_i), F''(rho_i) per atom (dFv, d2Fv): F' of the dual density is F' + eps F'' rho.d.
  __shared__ Dual el[kMaxEamElements][20];
  __shared__ Dual phx[kMaxPairTypes][7];
  __shared__ Dual prs[kMaxPairTypes][8];
  __shared__ double wpart[kBlock / 64];
  const int seeded = blockIdx.y;
  const int nel = P.nel, npt = nel * (nel + 1) / 2;
  for (int t = threadIdx.x; t < nel * 20; t += kBlock)
    el[t / 20][t % 20] = make_dual(P.el[t / 20][t % 20], t == seeded ? 1.0 : 0.0);
  for (int t = threadIdx.x; t < npt * 7; t += kBlock)
    phx[t / 7][t % 7] = make_dual(P.phi[t / 7][t % 7], 20 * nel + t == seeded ? 1.0 : 0.0);
  for (int t = threadIdx.x; t < npt * 8; t += kBlock)
    prs[t / 8][t % 8] = make_dual(P.pair[t / 8][t % 8], 20 * nel + 7 * npt + t == seeded ? 1.0 : 0.0);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double contrib = 0.0;
  if (i < b.n_atoms) {
    const int sA = b.species[i];
    const int fr = b.frame_of_atom[i];
    const int32_t *seg = b.seg_start + (size_t)i * (nel + 1);
    const double *h = b.cells + 9 * (size_t)fr;
    const double *ri = b.pos + 3 * (size_t)i;
    Dual rho = make_dual(0.0), rhodot = make_dual(0.0), phis = make_dual(0.0), phidot = make_dual(0.0);
    Dual eadp = make_dual(0.0);  // c E_adp + D_delta E_adp of this atom (lane 0)
    const double cf = frame_coeff ? frame_coeff[fr] : 0.0;
    for (int sb = 0; sb < nel; ++sb) {
      // ADP moments of this neighbour species and their directional derivatives:
      // m[0..2] = mu, m[3..8] = lambda (xx yy zz yz xz xy); md = the same for the tangent
      Dual m[9], md[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) m[k] = md[k] = make_dual(0.0);
      const Dual *pp = prs[pair_type(sA, sb, nel)];
      for (int q = seg[sb] + lane; q < seg[sb + 1]; q += 64) {
        const int j = b.pair_j[q];
        const double sx = (double)b.pair_shift[3 * (size_t)q], sy = (double)b.pair_shift[3 * (size_t)q + 1],
                     sz = (double)b.pair_shift[3 * (size_t)q + 2];
        const double *rj = b.pos + 3 * (size_t)j;
        const double dx = (rj[0] - ri[0]) + (sx * h[0] + sy * h[3] + sz * h[6]);
        const double dy = (rj[1] - ri[1]) + (sx * h[1] + sy * h[4] + sz * h[7]);
        const double dz = (rj[2] - ri[2]) + (sx * h[2] + sy * h[5] + sz * h[8]);
        const double r2 = dx * dx + dy * dy + dz * dz + eps;
        if (P.list_rc2 > 0.0 && !(r2 < P.list_rc2)) continue;
        const double r = sqrt(r2);
        double rdot = 0.0, tx = 0.0, ty = 0.0, tz = 0.0;
        if (dR) {
          const double *g = dh + 9 * (size_t)fr;
          const double *ui = dR + 3 * (size_t)i, *uj = dR + 3 * (size_t)j;
          tx = (uj[0] - ui[0]) + (sx * g[0] + sy * g[3] + sz * g[6]);
          ty = (uj[1] - ui[1]) + (sx * g[1] + sy * g[4] + sz * g[7]);
          tz = (uj[2] - ui[2]) + (sx * g[2] + sy * g[5] + sz * g[8]);
          rdot = (dx * tx + dy * ty + dz * tz) / r;
        }
        Dual f, df;
        const int pt = pair_type(sA, sb, nel);
        auto plain = [&](bool nn, bool tab, int col, int slot, Dual &fv, Dual &dfv) {  // a function without constants
          double a0, a1;
          if (nn) {
            a0 = pf[(size_t)col * ps + q];
            a1 = pf[(size_t)(col + 1) * ps + q];
          } else {
            spline_eval(tabs[slot], r, a0, a1);
          }
          (void)tab;
          fv = make_dual(a0);
          dfv = make_dual(a1);
        };
        // density function of the NEIGHBOUR's element
        if (((P.nn_rho | P.tab_rho) >> sb) & 1u) plain((P.nn_rho >> sb) & 1u, true, PF_RHO, slot_rho(sb), f, df);
        else el_rho<OTHER, Dual>(P, el, sb, r, f, df);
        rho += f;
        rhodot += df * rdot;
        if (((P.nn_phi | P.tab_phi) >> pt) & 1u) plain((P.nn_phi >> pt) & 1u, true, PF_PHI, slot_pair(nel, 1, pt), f, df);
        else pair_phi<OTHER, Dual>(P, el, phx, sA, sb, r, f, df);
        phis += f;
        phidot += df * rdot;
        if (P.adp) {
          Dual u, du, w, dw;
          if (((P.nn_u | P.tab_u) >> pt) & 1u) plain((P.nn_u >> pt) & 1u, true, PF_U, slot_pair(nel, 2, pt), u, du);
          else mishin_polar<Dual>(r, pp[0], pp[1], pp[2], pp[7], pp[6], u, du);
          if (((P.nn_w | P.tab_w) >> pt) & 1u) plain((P.nn_w >> pt) & 1u, true, PF_W, slot_pair(nel, 3, pt), w, dw);
          else mishin_polar<Dual>(r, pp[3], pp[4], pp[5], pp[7], pp[6], w, dw);
          const double D[3] = {dx, dy, dz}, Td[3] = {tx, ty, tz};
          const Dual ud = du * rdot, wd = dw * rdot;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            m[c] += u * D[c];
            md[c] += ud * D[c] + u * Td[c];
          }
          // xx yy zz yz xz xy
          const int ia[6] = {0, 1, 2, 1, 0, 0}, ib[6] = {0, 1, 2, 2, 2, 1};
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const double dd = D[ia[c]] * D[ib[c]];
            m[3 + c] += w * dd;
            md[3 + c] += wd * dd + w * (Td[ia[c]] * D[ib[c]] + D[ia[c]] * Td[ib[c]]);
          }
        }
      }
      if (P.adp) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          m[k] = make_dual(wave_sum(m[k].v), wave_sum(m[k].d));
          md[k] = make_dual(wave_sum(md[k].v), wave_sum(md[k].d));
        }
        const Dual nu = m[3] + m[4] + m[5], nud = md[3] + md[4] + md[5];
        const Dual e = 0.5 * (m[0] * m[0] + m[1] * m[1] + m[2] * m[2]) +
                       0.5 * (m[3] * m[3] + m[4] * m[4] + m[5] * m[5] + 2.0 * (m[6] * m[6] + m[7] * m[7] + m[8] * m[8])) -
                       nu * nu / 6.0;
        const Dual ed = m[0] * md[0] + m[1] * md[1] + m[2] * md[2] + m[3] * md[3] + m[4] * md[4] + m[5] * md[5] +
                        2.0 * (m[6] * md[6] + m[7] * md[7] + m[8] * md[8]) - nu * nud / 3.0;
        eadp += cf * e + ed;
      }
    }
    rho = make_dual(wave_sum(rho.v), wave_sum(rho.d));
    rhodot = make_dual(wave_sum(rhodot.v), wave_sum(rhodot.d));
    phis = make_dual(wave_sum(phis.v), wave_sum(phis.d));
    phidot = make_dual(wave_sum(phidot.v), wave_sum(phidot.d));
    if (lane == 0) {
      Dual F, dFd;
      if ((P.nn_embed >> sA) & 1u) {         // (only the dual parts matter below)
        F = make_dual(0.0, dFv[i] * rho.d);
        dFd = make_dual(dFv[i], d2Fv[i] * rho.d);
      } else if ((P.tab_embed >> sA) & 1u) {
        spline_eval_dual(tabs[slot_embed(nel, sA)], rho, F, dFd);
      } else {
        el_embed<OTHER, Dual>(P, el, sA, rho, F, dFd);
      }
      const Dual L = cf * (F + 0.5 * phis) + dFd * rhodot + 0.5 * phidot + eadp;
      contrib = L.d;
    }
  }
  if (lane == 0) wpart[threadIdx.x >> 6] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) v += wpart[w];
    partial[(size_t)seeded * gridDim.x + blockIdx.x] = v;
  }
}

// grad[q] = sum over the blocks, in a fixed order (one wavefront per constant)
__global__ __launch_bounds__(64) void eam_const_reduce_kernel(const double *partial, int n_blocks, double *grad) {
  const int q = blockIdx.x;
  double v = 0.0;
  for (int k = threadIdx.x; k < n_blocks; k += 64) v += partial[(size_t)q * n_blocks + k];
  v = wave_sum(v);
  if (threadIdx.x == 0) grad[q] = v;
}

}  // namespace

int64_t eam_constant_count(const EamModel *m) {
  const int nel = m->p.nel, npt = nel * (nel + 1) / 2;
  return 20 * (int64_t)nel + 7 * (int64_t)npt + (m->p.adp ? 8 * (int64_t)npt : 0);
}

void eam_get_constants(const EamModel *m, double *flat) {
  const int nel = m->p.nel, npt = nel * (nel + 1) / 2;
  for (int t = 0; t < nel * 20; ++t) flat[t] = m->p.el[t / 20][t % 20];
  // pair types under the Zjw04 mixing rule have no constants of their own: reported as zeros
  for (int t = 0; t < npt * 7; ++t) flat[20 * nel + t] = m->p.phi_kind[t / 7] == 1 ? m->p.phi[t / 7][t % 7] : 0.0;
  if (m->p.adp)
    for (int t = 0; t < npt * 8; ++t) flat[20 * nel + 7 * npt + t] = m->p.pair[t / 8][t % 8];
}

// the constants travel to the kernels by value (EamParams is a kernel argument): no device copy
void eam_update_constants(EamModel *m, const double *flat, int64_t n) {
  if (n != eam_constant_count(m)) throw std::invalid_argument("ta_update_constants: wrong number of values");
  const int nel = m->p.nel, npt = nel * (nel + 1) / 2;
  for (int64_t t = 0; t < n; ++t)
    if (!std::isfinite(flat[t])) throw std::invalid_argument("ta_update_constants: non-finite value");
  for (int t = 0; t < nel * 20; ++t) m->p.el[t / 20][t % 20] = flat[t];
  for (int t = 0; t < npt * 7; ++t)
    if (m->p.phi_kind[t / 7] == 1) m->p.phi[t / 7][t % 7] = flat[20 * nel + t];
  if (m->p.adp) {
    for (int pt = 0; pt < npt; ++pt)
      if (!(flat[20 * nel + 7 * npt + 8 * pt + 6] != 0.0))
        throw std::invalid_argument("ta_update_constants: the width h of a dipole / quadrupole cutoff must not be 0");
    for (int t = 0; t < npt * 8; ++t) m->p.pair[t / 8][t % 8] = flat[20 * nel + 7 * npt + t];
  }
}

// grad (device, eam_constant_count values); frame_coeff / dR / dh are device pointers (dR and dh both
// or neither). EAM / ADP with analytic functions only.
void eam_constant_gradient(EamModel *m, const DeviceBatch &b, const double *frame_coeff, const double *dR,
                           const double *dh, double *grad, hipStream_t s) {
  const EamParams &P = m->p;
  // Models that mix networks / tables with analytic functions (round 3): the caller has run the exact
  // forward pass (per-pair columns, F'); F'' of the embedding networks is made here.
  const bool has_nets = m->pair_nets || m->embed_nets;
  for (int sl = 0; sl < m->n_slots && m->embed_nets; ++sl)
    if (m->nets[sl].n_layers && sl >= P.nel && sl < 2 * P.nel &&
        (m->nets[sl].max_np > kNetMaxWidth || m->nets[sl].max_kp > kNetMaxWidth || m->nets[sl].xlo))
      throw std::invalid_argument("ta_constant_gradient: embedding networks wider than 128 units are not covered");
  const int64_t nq = eam_constant_count(m);
  if (b.n_atoms == 0) {
    (void)hipMemsetAsync(grad, 0, (size_t)nq * sizeof(double), s);
    return;
  }
  const unsigned blocks = (unsigned)((b.n_atoms + kBlock / 64 - 1) / (kBlock / 64));
  grow(m->gpartial, m->cap_gpartial, (size_t)nq * blocks + 8);
  double *d2F = nullptr;
  if (m->embed_nets) {
    grow(m->gcoeff, m->cap_gcoeff, (size_t)b.n_atoms + 8);
    d2F = m->gcoeff;
    for (int e = 0; e < P.nel; ++e) {
      const MlpDev &net = m->nets[slot_embed(P.nel, e)];
      const int n_el = b.elem_start[e + 1] - b.elem_start[e];
      if (!net.n_layers || n_el == 0) continue;
      hipLaunchKernelGGL(scalar_net_d2_kernel, dim3((unsigned)((n_el + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, net,
                         m->activation, b.elem_atoms + b.elem_start[e], n_el, m->rho_buf, (const double *)nullptr, d2F,
                         (size_t)0);
    }
  }
  (void)has_nets;
  bool other = false;
  for (int e = 0; e < P.nel; ++e) other = other || P.el_kind[e] != 0;
  if (other)
    hipLaunchKernelGGL(eam_const_grad_kernel<true>, dim3(blocks, (unsigned)nq), dim3(kBlock), 0, s, P, b,
                       frame_coeff, dR, dh, m->eps, m->tabs_dev, m->pf, m->cap_pairs, m->dF, d2F, m->gpartial);
  else
    hipLaunchKernelGGL(eam_const_grad_kernel<false>, dim3(blocks, (unsigned)nq), dim3(kBlock), 0, s, P, b,
                       frame_coeff, dR, dh, m->eps, m->tabs_dev, m->pf, m->cap_pairs, m->dF, d2F, m->gpartial);
  hipLaunchKernelGGL(eam_const_reduce_kernel, dim3((unsigned)nq), dim3(64), 0, s, m->gpartial, (int)blocks, grad);
}

// Hessian-vector products on the resident batch (the forward pass of eam_compute must have run: F'(rho)
// in m->dF). Device pointers; `dFdot` [n_dir][N] scratch. False = this model keeps the central differences.
bool eam_hvp_supported(const EamModel *m) {
  if (m->pair_nets) return false;  // (nn pair functions: through their tables only)
  for (int e = 0; e < m->p.nel && m->embed_nets; ++e) {  // embedding networks: F'' by the second-derivative sweep
    if (slot_embed(m->p.nel, e) >= m->n_slots) return false;
    const MlpDev &net = m->nets[slot_embed(m->p.nel, e)];
    if (net.n_layers && (net.max_np > kNetMaxWidth || net.max_kp > kNetMaxWidth || net.xlo)) return false;
  }
  for (int e = 0; e < m->p.nel; ++e)
    if (m->p.el_kind[e] != 0) return false;  // sutton90 / Be/1 / grimes: first derivatives only
  return true;
}
// doubles of work space eam_hvp needs besides dFdot / fdot / wdot (ADP: the moments' tangents)
size_t eam_hvp_extra_doubles(const EamModel *m, const DeviceBatch &b, int n_dir) {
  return m->p.adp ? (size_t)n_dir * (size_t)b.n_atoms * m->p.nel * 9 : 0;
}
void eam_hvp(EamModel *m, const DeviceBatch &b, int n_dir, bool unit, int first, const double *dR, const double *dh,
             double *dFdot, double *fdot, double *wdot, double *extra, hipStream_t s) {
  if (b.n_atoms == 0 || n_dir == 0) return;
  const HvpArgs a{n_dir, unit ? 1 : 0, first, dR, dh, m->eps};
  const dim3 grid((unsigned)((b.n_atoms + kBlock / 64 - 1) / (kBlock / 64)), (unsigned)n_dir);
  if (m->p.adp) hipLaunchKernelGGL(adp_hvp_atom_kernel, grid, dim3(kBlock), 0, s, m->p, b, a, m->tabs_dev, dFdot, extra);
  else hipLaunchKernelGGL(eam_hvp_atom_kernel, grid, dim3(kBlock), 0, s, m->p, b, a, m->tabs_dev, dFdot);
  for (int e = 0; e < m->p.nel && m->embed_nets; ++e) {  // atoms with an embedding network: rho-dot -> F''(rho) rho-dot
    const MlpDev &net = m->nets[slot_embed(m->p.nel, e)];
    const int n_el = b.elem_start[e + 1] - b.elem_start[e];
    if (!net.n_layers || n_el == 0) continue;
    hipLaunchKernelGGL(scalar_net_d2_kernel, dim3((unsigned)((n_el + kBlock - 1) / kBlock), (unsigned)n_dir), dim3(kBlock),
                       0, s, net, m->activation, b.elem_atoms + b.elem_start[e], n_el, m->rho_buf, dFdot, dFdot,
                       (size_t)b.n_atoms);
  }
  if (m->p.adp)
    hipLaunchKernelGGL(adp_hvp_force_kernel, grid, dim3(kBlock), 0, s, m->p, b, a, m->tabs_dev, m->dF, dFdot, m->mom,
                       extra, fdot, wdot);
  else
    hipLaunchKernelGGL(eam_hvp_force_kernel, grid, dim3(kBlock), 0, s, m->p, b, a, m->tabs_dev, m->dF, dFdot, fdot,
                       wdot);
}

void eam_compute(EamModel *m, const DeviceBatch &b, uint32_t want, hipStream_t s, hipEvent_t *) {
  if (b.n_atoms == 0) return;
  SFParams sf;
  std::memset(&sf, 0, sizeof(sf));
  sf.n_elements = m->p.nel;
  sf.eps = m->eps;
  const size_t ps = m->cap_pairs;
  const bool pair_nets = m->pair_nets && b.n_pairs > 0;
  if (pair_nets) {
    double *rbuf = m->pf + (size_t)(m->p.adp ? 8 : 4) * ps;
    hipLaunchKernelGGL(eam_geom_kernel, dim3((unsigned)((b.n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       s, b, m->eps, rbuf);
    const unsigned tiles = (unsigned)((b.n_pairs + kMlpRows - 1) / kMlpRows);
    if (m->fast_nt) {
      // workgroups of 4 wavefronts stride over the tiles; enough of them to fill the chip a few
      // times over, few enough that the LDS image of the weights is amortised
      const dim3 grid(std::min((tiles + 3) / 4, 2048u), (unsigned)m->fns.n);
#define TA_NN_FAST(ACT, NT)                                                                              \
  hipLaunchKernelGGL((eam_nn_pair_fast_kernel<ACT, NT>), grid, dim3(kBlock), m->fast_lds, s, m->p,        \
                     m->nets_dev, m->activation, m->fns, b, rbuf, m->pf, ps)
#define TA_NN_FAST_NT(ACT)                                                                               \
  switch (m->fast_nt) {                                                                                  \
    case 1: TA_NN_FAST(ACT, 1); break;                                                                   \
    case 2: TA_NN_FAST(ACT, 2); break;                                                                   \
    case 3: TA_NN_FAST(ACT, 3); break;                                                                   \
    default: TA_NN_FAST(ACT, 4); break;                                                                  \
  }
      if (m->activation == TA_ACT_SOFTPLUS) {
        TA_NN_FAST_NT(TA_ACT_SOFTPLUS)
      } else {
        TA_NN_FAST_NT(-1)
      }
#undef TA_NN_FAST_NT
#undef TA_NN_FAST
    } else if (m->fast_1h) {
      const dim3 grid(std::min((unsigned)((b.n_pairs + kBlock - 1) / kBlock), 4096u), (unsigned)m->fns.n);
      if (m->activation == TA_ACT_SOFTPLUS)
        hipLaunchKernelGGL(eam_nn_pair_1h_kernel<TA_ACT_SOFTPLUS>, grid, dim3(kBlock), m->fast_lds, s, m->p,
                           m->nets_dev, m->activation, m->fns, b, rbuf, m->pf, ps);
      else
        hipLaunchKernelGGL(eam_nn_pair_1h_kernel<-1>, grid, dim3(kBlock), m->fast_lds, s, m->p, m->nets_dev,
                           m->activation, m->fns, b, rbuf, m->pf, ps);
    } else {
      hipLaunchKernelGGL(eam_nn_pair_kernel<kNetThreads>, dim3(tiles), dim3(kNetThreads), net_lds_bytes(m), s,
                         m->p, m->nets_dev, m->activation, b, rbuf, m->pf, ps, m->stride);
    }
  }
  bool other = false;
  for (int e = 0; e < m->p.nel; ++e) other = other || m->p.el_kind[e] != 0;
  // analytic / tabulated pair functions: forces in one pass per centre (eam_force_kernel,
  // adp_force_kernel); nn pair functions: dE/dD per pair, then the shared force gather
  static const bool no_fold = getenv("TA_EAM_NO_FOLD") != nullptr;    // A/B switches
  static const int w_env = getenv("TA_EAM_W") ? atoi(getenv("TA_EAM_W")) : 0;
  const bool want_f = (want & (TA_WANT_FORCES | TA_WANT_VIRIAL)) && b.n_pairs > 0;
  static const bool no_fold_adp = getenv("TA_ADP_NO_FOLD") != nullptr;
  const bool fold = want_f && !pair_nets && !no_fold && !(m->p.adp && no_fold_adp);
  // One-pass force kernels (and energy-only evaluations) need no pair records: the force kernels
  // recompute D from pos[j] + S.h. TA_EAM_RECORDS=1 keeps the record round trip (A/B switch).
  static const bool keep_rec = getenv("TA_EAM_RECORDS") != nullptr;
  const bool no_rec = !pair_nets && !keep_rec && (fold || !(want & (TA_WANT_FORCES | TA_WANT_VIRIAL)));
  // lanes per atom (measured, 4000-atom Ni frames, rc 6.5, us per frame for W = 16 / 32 / 64): EAM one
  // frame 25.5 / 26.2 / 28.0, 64 frames 11.3 / 14.1 / 16.7; ADP (one-pass force kernel, W = 16 / 32)
  // one frame 36.6 / 36.0, 64 frames 18.0 / 21.0
  const int W = w_env == 16 || w_env == 32 || w_env == 64 ? w_env
                : (!m->p.adp || b.n_atoms >= 32768) ? 16 : 32;
  const dim3 agrid((unsigned)((b.n_atoms * W + kBlock - 1) / kBlock));
#define TA_EAM_ATOM(O, WW)                                                                                   \
  hipLaunchKernelGGL((eam_atom_kernel<O, WW>), agrid, dim3(kBlock), 0, s, m->p, b, m->dF, m->mom, m->eps,    \
                     m->pf, ps, m->rho_buf, pair_nets ? 1 : (no_rec ? 2 : 0), m->tabs_dev)
#define TA_EAM_BY_W(MACRO)                            \
  do {                                                \
    if (other) {                                      \
      if (W == 16) MACRO(true, 16);                   \
      else if (W == 32) MACRO(true, 32);              \
      else MACRO(true, 64);                           \
    } else {                                          \
      if (W == 16) MACRO(false, 16);                  \
      else if (W == 32) MACRO(false, 32);             \
      else MACRO(false, 64);                          \
    }                                                 \
  } while (0)
  TA_EAM_BY_W(TA_EAM_ATOM);
#undef TA_EAM_ATOM
  if (m->embed_nets) {
    EmbedTiles t;
    std::memset(&t, 0, sizeof(t));
    t.nel = m->p.nel;
    int blocks = 0;
    for (int e = 0; e < t.nel; ++e) {
      t.tile_start[e] = blocks;
      t.elem_start[e] = b.elem_start[e];
      if ((m->p.nn_embed >> e) & 1u) blocks += (b.elem_start[e + 1] - b.elem_start[e] + kMlpRows - 1) / kMlpRows;
    }
    t.tile_start[t.nel] = blocks;
    t.elem_start[t.nel] = b.elem_start[t.nel];
    if (blocks)
      hipLaunchKernelGGL(eam_nn_embed_kernel<kNetThreads>, dim3((unsigned)blocks), dim3(kNetThreads),
                         net_lds_bytes(m), s, m->nets_dev, t, m->activation, b, m->rho_buf, m->dF,
                         m->stride);
  }
  if (want_f) {
    if (fold) {
      const dim3 fgrid((unsigned)((b.n_atoms + 15) / 16));
#define TA_EAM_FORCE(O, WW) \
  hipLaunchKernelGGL((eam_force_kernel<O, WW>), fgrid, dim3(16 * WW), 0, s, m->p, b, m->dF, m->tabs_dev, \
                     no_rec ? 1 : 0, m->eps)
#define TA_ADP_FORCE(O, WW)                                                                              \
  hipLaunchKernelGGL((adp_force_kernel<O, WW>), fgrid, dim3(16 * WW), 0, s, m->p, b, m->dF, m->mom, \
                     m->tabs_dev, no_rec ? 1 : 0, m->eps)
      if (m->p.adp) TA_EAM_BY_W(TA_ADP_FORCE);
      else TA_EAM_BY_W(TA_EAM_FORCE);
#undef TA_EAM_FORCE
#undef TA_ADP_FORCE
    } else {
      const dim3 pgrid((unsigned)((b.n_pairs + kBlock - 1) / kBlock));
      if (other)
        hipLaunchKernelGGL(eam_pair_kernel<true>, pgrid, dim3(kBlock), 0, s, m->p, b, m->dF, m->mom, m->pf, ps,
                           m->tabs_dev);
      else
        hipLaunchKernelGGL(eam_pair_kernel<false>, pgrid, dim3(kBlock), 0, s, m->p, b, m->dF, m->mom, m->pf, ps,
                           m->tabs_dev);
      launch_force_gather(sf, b, s);
    }
  } else if (want & (TA_WANT_FORCES | TA_WANT_VIRIAL)) {
    launch_force_gather(sf, b, s);
  }
#undef TA_EAM_BY_W
}

}  // namespace ta
