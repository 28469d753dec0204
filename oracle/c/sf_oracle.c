/*
 * ORACLE (test infrastructure only) — plain C restatement of the reference's
 * symmetry-function + per-atom MLP path with analytic forces and virial.
 * Used (a) to check the HIP path at sizes NumPy cannot hold (4000 atoms,
 * 14.6 M triples) and (b) as the `cpu_baseline` ("port") leg of bench.py.
 * Never linked into or called by the product library.
 *
 * It follows the reference formulas literally (libm cos / exp / pow, explicit
 * r_jk square root, half enumeration j < k) — deliberately NOT the algebra of
 * the HIP kernels (ordered enumeration, polynomial cutoff in r^2), so the two
 * are independent statements of the same maths:
 *
 *   pair geometry ............. transformer/universal.py:448-474
 *   cutoffs ................... nn/cutoff.py:20-85
 *   G2 ........................ nn/atomic/sf.py:79-119
 *   G4 ........................ nn/atomic/sf.py:121-182
 *   min-max ................... nn/atomic/atomic.py:157-195
 *   MLP ....................... nn/convolutional.py:257-290, nn/utils.py:39-74
 *   energy / forces / virial .. nn/atomic/atomic.py:270-302, nn/basic.py:277-331
 *
 * Build: make -C oracle/c   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_LAYERS 8
#define PI 3.14159265358979323846

typedef struct {
  int n_elements;
  double rcut, acut;
  int angular;
  int cutoff; /* 0 cosine, 1 polynomial */
  int n_rad, n_ang;
  const double *eta, *omega;         /* [n_rad] flattened grid, omega fastest */
  const double *beta, *gamma, *zeta; /* [n_ang] flattened grid, zeta fastest */
  int activation; /* ids of include/tensoralloy_amd.h */
  int use_resnet_dt;
  int minmax;
  const int *n_layers;     /* [n_elements] */
  const int *layer_sizes;  /* per element [in, h.., 1] concatenated */
  const double *weights;   /* per element per layer: W[in][out], b[out] */
  const double *xlo, *xhi; /* [n_elements * D] */
} sf_model;

static double fcut(const sf_model *m, double r, double rc, double *df) {
  double x = r / rc;
  if (x >= 1.0) {
    *df = 0.0;
    return (m->cutoff == 0) ? 0.5 * (cos(PI) + 1.0) : 0.0;
  }
  if (m->cutoff == 0) {
    *df = -0.5 * PI / rc * sin(PI * x);
    return 0.5 * (cos(PI * x) + 1.0);
  } else {
    const double g = 5.0;
    *df = g * (g + 1.0) * (pow(x, g) - pow(x, g - 1.0)) / rc;
    return 1.0 + g * pow(x, g + 1.0) - (g + 1.0) * pow(x, g);
  }
}

static void act(int id, double x, double *h, double *dh) {
  switch (id) {
    case 1: { /* softplus */
      double e = exp(-fabs(x));
      *h = (x > 0 ? x : 0.0) + log1p(e);
      *dh = x >= 0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
      break;
    }
    case 0: *h = x > 0 ? x : 0; *dh = x > 0 ? 1 : 0; break;
    case 4: *h = x > 0 ? x : 0.2 * x; *dh = x > 0 ? 1 : 0.2; break;
    case 2: { double t = tanh(x); *h = t; *dh = 1 - t * t; break; }
    case 5: { double s = 1.0 / (1.0 + exp(-x)); *h = s; *dh = s * (1 - s); break; }
    case 6: { double d = 1 + fabs(x); *h = x / d; *dh = 1 / (d * d); break; }
    case 7: *h = x > 0 ? x : expm1(x); *dh = x > 0 ? 1 : exp(x); break;
    case 3: { double s = sqrt(x * x + 4.0); *h = 0.5 * (x + s); *dh = 0.5 * (1 + x / s); break; }
    default: *h = x; *dh = 1;
  }
}

static int radial_term(int center, int other) {
  return other == center ? 0 : (other < center ? other + 1 : other);
}
static int angular_term(int s1, int s2, int nel) {
  int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
  return a * nel - (a * (a - 1)) / 2 + (b - a);
}

/* MLP forward + backward to inputs for one atom. x: [D] raw descriptors. */
static double mlp_atom(const sf_model *m, int el, const double *x, double *dEdx, int D,
                       double *work /* >= 4 * maxw * (MAX_LAYERS + 1) */, int maxw) {
  const int *sizes = m->layer_sizes;
  const double *w = m->weights;
  for (int e = 0; e < el; ++e) {
    int L = m->n_layers[e];
    for (int l = 0; l < L; ++l) w += (size_t)sizes[l] * sizes[l + 1] + sizes[l + 1];
    sizes += L + 1;
  }
  const int L = m->n_layers[el];
  double *h = work;                          /* [(L+1)][maxw] layer inputs */
  double *da = work + (size_t)(L + 1) * maxw; /* [L][maxw] act' */
  double *delta = da + (size_t)L * maxw, *delta2 = delta + maxw;
  const double *wl[MAX_LAYERS];
  for (int k = 0; k < D; ++k) {
    double v = x[k];
    if (m->minmax) {
      double den = m->xhi[el * D + k] - m->xlo[el * D + k];
      v = den != 0.0 ? (m->xhi[el * D + k] - v) / den : 0.0;
    }
    h[k] = v;
  }
  const double *wp = w;
  for (int l = 0; l < L; ++l) {
    const int K = sizes[l], N = sizes[l + 1];
    wl[l] = wp;
    const double *W = wp, *b = wp + (size_t)K * N;
    double *hin = h + (size_t)l * maxw, *hout = h + (size_t)(l + 1) * maxw;
    const int res = (m->use_resnet_dt && l > 0 && l < L - 1 && K == N);
    for (int n = 0; n < N; ++n) {
      double z = b[n];
      for (int k = 0; k < K; ++k) z += hin[k] * W[(size_t)k * N + n];
      if (l < L - 1) {
        double hv, dv;
        act(m->activation, z, &hv, &dv);
        hout[n] = res ? hv + hin[n] : hv;
        da[(size_t)l * maxw + n] = dv;
      } else {
        hout[n] = z;
        da[(size_t)l * maxw + n] = 1.0;
      }
    }
    wp += (size_t)K * N + N;
  }
  const double y = h[(size_t)L * maxw];
  delta[0] = 1.0;
  for (int l = L - 1; l >= 0; --l) {
    const int K = sizes[l], N = sizes[l + 1];
    const double *W = wl[l];
    const int res = (m->use_resnet_dt && l > 0 && l < L - 1 && K == N);
    for (int k = 0; k < K; ++k) {
      double s = res ? delta[k] : 0.0;
      for (int n = 0; n < N; ++n) s += delta[n] * da[(size_t)l * maxw + n] * W[(size_t)k * N + n];
      delta2[k] = s;
    }
    memcpy(delta, delta2, (size_t)K * sizeof(double));
  }
  for (int k = 0; k < D; ++k) {
    double d = delta[k];
    if (m->minmax) {
      double den = m->xhi[el * D + k] - m->xlo[el * D + k];
      d = den != 0.0 ? -d / den : 0.0;
    }
    dEdx[k] = d;
  }
  return y;
}

/*
 * Pairs must be sorted by centre i (any order inside a centre).
 * Returns 0 on success. Outputs may be NULL except energy.
 */
int sf_oracle_eval(const sf_model *m, int n_atoms, const int *species, const double *pos,
                   const double *cell, long n_pairs, const int *pi, const int *pj, const int *pS,
                   int want_forces, int nthreads, double *energy, double *atomic, double *forces,
                   double *virial, double *descriptors) {
  const int nel = m->n_elements, nr = m->n_rad, na = m->n_ang;
  const int n_radial = nel * nr;
  const int D = n_radial + (m->angular ? nel * (nel + 1) / 2 * na : 0);
  const double eps = 1e-14;
  const double rc2 = m->rcut * m->rcut, ac2 = m->acut * m->acut;
  double *Dv = (double *)malloc(sizeof(double) * 4 * (size_t)(n_pairs ? n_pairs : 1)); /* dx dy dz r */
  double *G = (double *)calloc((size_t)n_atoms * D + 1, sizeof(double));
  double *dEdG = (double *)calloc((size_t)n_atoms * D + 1, sizeof(double));
  long *start = (long *)calloc((size_t)n_atoms + 1, sizeof(long));
  if (!Dv || !G || !dEdG || !start) return -1;
  for (long p = 0; p < n_pairs; ++p) start[pi[p] + 1]++;
  for (int i = 0; i < n_atoms; ++i) start[i + 1] += start[i];
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  int maxw = D;
  {
    const int *s = m->layer_sizes;
    for (int e = 0; e < nel; ++e) {
      for (int l = 0; l <= m->n_layers[e]; ++l)
        if (s[l] > maxw) maxw = s[l];
      s += m->n_layers[e] + 1;
    }
  }
#pragma omp parallel for schedule(static)
  for (long p = 0; p < n_pairs; ++p) {
    const int i = pi[p], j = pj[p];
    double d[3];
    for (int b = 0; b < 3; ++b)
      d[b] = pos[3 * j + b] - pos[3 * i + b] +
             (pS[3 * p] * cell[b] + pS[3 * p + 1] * cell[3 + b] + pS[3 * p + 2] * cell[6 + b]);
    Dv[4 * p] = d[0];
    Dv[4 * p + 1] = d[1];
    Dv[4 * p + 2] = d[2];
    Dv[4 * p + 3] = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + eps);
  }
  /* ---- descriptors ---- */
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n_atoms; ++i) {
    double *Gi = G + (size_t)i * D;
    const int sA = species[i];
    for (long p = start[i]; p < start[i + 1]; ++p) {
      const double r = Dv[4 * p + 3];
      double df;
      const double f = fcut(m, r, m->rcut, &df);
      const int t = radial_term(sA, species[pj[p]]);
      for (int c = 0; c < nr; ++c) {
        const double dr = r - m->omega[c];
        Gi[t * nr + c] += exp(-m->eta[c] * dr * dr / rc2) * f;
      }
    }
    if (!m->angular) continue;
    for (long p = start[i]; p < start[i + 1]; ++p) {
      const double rij = Dv[4 * p + 3];
      double dfa;
      const double fa = fcut(m, rij, m->acut, &dfa);
      for (long q = p + 1; q < start[i + 1]; ++q) {
        const double rik = Dv[4 * q + 3];
        double dfb, dfd;
        const double fb = fcut(m, rik, m->acut, &dfb);
        const double ex = Dv[4 * q] - Dv[4 * p], ey = Dv[4 * q + 1] - Dv[4 * p + 1],
                     ez = Dv[4 * q + 2] - Dv[4 * p + 2];
        const double rjk = sqrt(ex * ex + ey * ey + ez * ez + eps);
        const double fd = fcut(m, rjk, m->acut, &dfd);
        if (fa * fb * fd == 0.0) continue; /* zero summand */
        const double lower = 2.0 * rij * rik;
        const double cth = lower != 0.0 ? (rij * rij + rik * rik - rjk * rjk) / lower : 0.0;
        const double z = (rij * rij + rik * rik + rjk * rjk) / ac2;
        const int t = angular_term(species[pj[p]], species[pj[q]], nel);
        const double fprod = fa * fb * fd;
        for (int c = 0; c < na; ++c) {
          const double v = pow(2.0, 1.0 - m->zeta[c]) * pow(1.0 + m->gamma[c] * cth, m->zeta[c]) *
                           exp(-m->beta[c] * z) * fprod;
          Gi[n_radial + t * na + c] += v;
        }
      }
    }
  }
  if (descriptors) memcpy(descriptors, G, sizeof(double) * (size_t)n_atoms * D);
  if (!m->weights) {
    free(Dv); free(G); free(dEdG); free(start);
    return 0;
  }
  /* ---- MLP ---- */
  double etot = 0.0;
  double *eat = (double *)calloc((size_t)n_atoms ? n_atoms : 1, sizeof(double));
#pragma omp parallel
  {
    double *work = (double *)malloc(sizeof(double) * (size_t)maxw * (2 * MAX_LAYERS + 4));
#pragma omp for schedule(static)
    for (int i = 0; i < n_atoms; ++i)
      eat[i] = mlp_atom(m, species[i], G + (size_t)i * D, dEdG + (size_t)i * D, D, work, maxw);
    free(work);
  }
  for (int i = 0; i < n_atoms; ++i) etot += eat[i];
  *energy = etot;
  if (atomic) memcpy(atomic, eat, sizeof(double) * (size_t)n_atoms);
  free(eat);
  if (!want_forces) {
    free(Dv); free(G); free(dEdG); free(start);
    return 0;
  }
  /* ---- forces and virial ---- */
  int nt = 1;
#ifdef _OPENMP
  nt = omp_get_max_threads();
#endif
  double *Fbuf = (double *)calloc((size_t)nt * 3 * (size_t)n_atoms + 1, sizeof(double));
  double *Wbuf = (double *)calloc((size_t)nt * 9, sizeof(double));
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double *F = Fbuf + (size_t)tid * 3 * n_atoms, *W = Wbuf + (size_t)tid * 9;
#pragma omp for schedule(dynamic, 8)
    for (int i = 0; i < n_atoms; ++i) {
      const double *wi = dEdG + (size_t)i * D;
      const int sA = species[i];
      for (long p = start[i]; p < start[i + 1]; ++p) {
        const double r = Dv[4 * p + 3];
        double df;
        const double f = fcut(m, r, m->rcut, &df);
        const int t = radial_term(sA, species[pj[p]]);
        double s = 0.0;
        for (int c = 0; c < nr; ++c) {
          const double dr = r - m->omega[c];
          const double e = exp(-m->eta[c] * dr * dr / rc2);
          s += wi[t * nr + c] * e * (df - 2.0 * m->eta[c] * dr * f / rc2);
        }
        const int j = pj[p];
        for (int a = 0; a < 3; ++a) {
          const double g = s / r * Dv[4 * p + a];
          F[3 * i + a] += g;
          F[3 * j + a] -= g;
          for (int b = 0; b < 3; ++b) W[3 * a + b] += g * Dv[4 * p + b];
        }
      }
      if (!m->angular) continue;
      for (long p = start[i]; p < start[i + 1]; ++p) {
        const double a_ = Dv[4 * p + 3];
        double dfa;
        const double fa = fcut(m, a_, m->acut, &dfa);
        const int j = pj[p];
        for (long q = p + 1; q < start[i + 1]; ++q) {
          const double b_ = Dv[4 * q + 3];
          double dfb, dfd;
          const double fb = fcut(m, b_, m->acut, &dfb);
          double e3[3] = {Dv[4 * q] - Dv[4 * p], Dv[4 * q + 1] - Dv[4 * p + 1],
                          Dv[4 * q + 2] - Dv[4 * p + 2]};
          const double d_ = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2] + eps);
          const double fd = fcut(m, d_, m->acut, &dfd);
          if ((fa == 0.0 && dfa == 0.0) || (fb == 0.0 && dfb == 0.0) || (fd == 0.0 && dfd == 0.0))
            continue; /* every term carries one of these factors */
          const double cth = (a_ * a_ + b_ * b_ - d_ * d_) / (2.0 * a_ * b_);
          const double z = (a_ * a_ + b_ * b_ + d_ * d_) / ac2;
          const double dca = 1.0 / b_ - cth / a_, dcb = 1.0 / a_ - cth / b_, dcd = -d_ / (a_ * b_);
          const int t = angular_term(species[j], species[pj[q]], nel);
          double dva = 0, dvb = 0, dvd = 0;
          for (int c = 0; c < na; ++c) {
            const double w = wi[n_radial + t * na + c];
            const double zt = m->zeta[c], gm = m->gamma[c], bt = m->beta[c];
            const double base = 1.0 + gm * cth;
            const double P = pow(base, zt);
            double dP;
            if (base != 0.0)
              dP = zt * gm * pow(base, zt - 1.0);
            else
              dP = (zt == 1.0) ? zt * gm : 0.0;
            const double e = pow(2.0, 1.0 - zt) * exp(-bt * z);
            dva += w * e * (dP * dca * fa - 2.0 * bt * a_ / ac2 * P * fa + P * dfa) * fb * fd;
            dvb += w * e * (dP * dcb * fb - 2.0 * bt * b_ / ac2 * P * fb + P * dfb) * fa * fd;
            dvd += w * e * (dP * dcd * fd - 2.0 * bt * d_ / ac2 * P * fd + P * dfd) * fa * fb;
          }
          const int k = pj[q];
          for (int x = 0; x < 3; ++x) {
            const double ta = dva / a_ * Dv[4 * p + x], tb = dvb / b_ * Dv[4 * q + x],
                         td = dvd / d_ * e3[x];
            F[3 * i + x] += ta + tb;
            F[3 * j + x] += -ta + td;
            F[3 * k + x] += -tb - td;
            for (int y = 0; y < 3; ++y)
              W[3 * x + y] += ta * Dv[4 * p + y] + tb * Dv[4 * q + y] + td * e3[y];
          }
        }
      }
    }
  }
  if (forces) {
    memset(forces, 0, sizeof(double) * 3 * (size_t)n_atoms);
    for (int t = 0; t < nt; ++t)
      for (size_t k = 0; k < 3 * (size_t)n_atoms; ++k) forces[k] += Fbuf[(size_t)t * 3 * n_atoms + k];
  }
  if (virial) {
    memset(virial, 0, sizeof(double) * 9);
    for (int t = 0; t < nt; ++t)
      for (int k = 0; k < 9; ++k) virial[k] += Wbuf[t * 9 + k];
  }
  free(Fbuf); free(Wbuf); free(Dv); free(G); free(dEdG); free(start);
  return 0;
}
