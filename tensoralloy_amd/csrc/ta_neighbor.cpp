// Host-side neighbour list for libtensoralloy_amd.so.
//
// Replaces the host part of reference transformer/universal.py:46-112
// (`get_radial_metadata`): `ase.neighborlist.neighbor_list('ijS', atoms, rc)`
// (universal.py:58) followed by per-pair Python loops. Semantics kept:
//   * full list, both directions, strict |R_j - R_i + S.h| < rc;
//   * periodic images (incl. self-images) are distinct neighbours;
//   * S is relative to the positions as given (atoms may lie outside the cell);
//   * non-periodic axes never shift.
// The dense slot indices (`v2g_map`) of the reference are NOT built here: the
// kernels work on the packed list; the Python transformer mirror builds them
// only when a caller asks for the reference's feed dict.
//
// Linked-cell search over the wrapped atoms plus their ghost images.

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <thread>

#include "ta_internal.h"

namespace ta {
namespace {

struct Cand {  // one neighbour of a centre
  int32_t species, j, sx, sy, sz;
  double r2;
};

inline void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
inline double dot3(const double *a, const double *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
inline double norm3(const double *a) { return std::sqrt(dot3(a, a)); }

// Zero lattice vectors (non-periodic axes without a box) are replaced by unit
// vectors orthogonal to the others, as `Atoms.get_cell(complete=True)` does
// (reference universal.py:864).
void complete_cell(const double *in, double h[9]) {
  std::memcpy(h, in, 9 * sizeof(double));
  int missing[3], nm = 0, present[3], np = 0;
  for (int a = 0; a < 3; ++a) {
    if (h[3 * a] == 0.0 && h[3 * a + 1] == 0.0 && h[3 * a + 2] == 0.0)
      missing[nm++] = a;
    else
      present[np++] = a;
  }
  if (nm == 3) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) h[3 * a + b] = (a == b) ? 1.0 : 0.0;
  } else if (nm == 2) {
    double v[3], n = norm3(&h[3 * present[0]]);
    for (int b = 0; b < 3; ++b) v[b] = h[3 * present[0] + b] / n;
    int k = 0;
    for (int b = 1; b < 3; ++b)
      if (std::fabs(v[b]) < std::fabs(v[k])) k = b;
    double t[3] = {0, 0, 0}, u[3], w[3];
    t[k] = 1.0;
    cross3(v, t, u);
    n = norm3(u);
    for (int b = 0; b < 3; ++b) u[b] /= n;
    cross3(v, u, w);
    for (int b = 0; b < 3; ++b) {
      h[3 * missing[0] + b] = u[b];
      h[3 * missing[1] + b] = w[b];
    }
  } else if (nm == 1) {
    double n[3];
    cross3(&h[3 * present[0]], &h[3 * present[1]], n);
    double len = norm3(n);
    for (int b = 0; b < 3; ++b) h[3 * missing[0] + b] = n[b] / len;
  }
}

bool invert3(const double *h, double *inv, double *det_out) {
  double c0[3], c1[3], c2[3];
  cross3(&h[3], &h[6], c0);
  cross3(&h[6], &h[0], c1);
  cross3(&h[0], &h[3], c2);
  double det = dot3(&h[0], c0);
  *det_out = det;
  if (det == 0.0 || !std::isfinite(det)) return false;
  for (int a = 0; a < 3; ++a) {
    inv[3 * a + 0] = c0[a] / det;
    inv[3 * a + 1] = c1[a] / det;
    inv[3 * a + 2] = c2[a] / det;
  }
  return true;
}

struct FramePairs {
  std::vector<int32_t> count;      // per atom
  std::vector<Cand> cands;         // concatenated per atom, sorted
  std::string error;
};

void build_frame(const ta_frame &fr, int32_t n_elements, double rmax, FramePairs &out) {
  const int n = fr.n_atoms;
  out.count.assign(n, 0);
  out.cands.clear();
  if (n == 0) return;
  double h[9], hinv[9], det;
  complete_cell(fr.cell, h);
  if (!invert3(h, hinv, &det)) {
    out.error = "singular cell matrix";
    return;
  }
  const double vol = std::fabs(det);
  bool pbc[3];
  for (int a = 0; a < 3; ++a) pbc[a] = fr.pbc[a] != 0;

  // fractional coordinates, wrap into the cell along periodic axes
  std::vector<double> fw(3 * (size_t)n), rw(3 * (size_t)n);
  std::vector<int32_t> wrap(3 * (size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    const double *r = &fr.positions[3 * (size_t)i];
    if (fr.species[i] < 0 || fr.species[i] >= n_elements) {
      out.error = "species index out of range";
      return;
    }
    for (int a = 0; a < 3; ++a) {
      double f = r[0] * hinv[0 * 3 + a] + r[1] * hinv[1 * 3 + a] + r[2] * hinv[2 * 3 + a];
      if (!std::isfinite(f)) {
        out.error = "non-finite position";
        return;
      }
      int32_t w = 0;
      if (pbc[a]) {
        w = (int32_t)std::floor(f);
        f -= w;
      }
      fw[3 * (size_t)i + a] = f;
      wrap[3 * (size_t)i + a] = w;
    }
    for (int b = 0; b < 3; ++b)
      rw[3 * (size_t)i + b] = r[b] - (wrap[3 * (size_t)i] * h[b] + wrap[3 * (size_t)i + 1] * h[3 + b] +
                                     wrap[3 * (size_t)i + 2] * h[6 + b]);
  }

  // image ranges and fractional padding
  int nmax[3];
  double pad[3], lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    int b = (a + 1) % 3, c = (a + 2) % 3;
    double cr[3];
    cross3(&h[3 * b], &h[3 * c], cr);
    double height = vol / norm3(cr);
    pad[a] = rmax / height;
    nmax[a] = pbc[a] ? (int)std::floor(rmax / height) + 1 : 0;
    if (pbc[a]) {
      lo[a] = -pad[a];
      hi[a] = 1.0 + pad[a];
    } else {
      double mn = fw[a], mx = fw[a];
      for (int i = 1; i < n; ++i) {
        mn = std::min(mn, fw[3 * (size_t)i + a]);
        mx = std::max(mx, fw[3 * (size_t)i + a]);
      }
      lo[a] = mn;
      hi[a] = mx;
    }
  }

  // ghost images (s = 0 included) inside the padded region
  struct Ghost {
    int32_t j, sx, sy, sz;
    double f[3], x[3];
  };
  std::vector<Ghost> ghosts;
  ghosts.reserve((size_t)n * 4);
  const double tol = 1e-9;
  for (int sx = -nmax[0]; sx <= nmax[0]; ++sx)
    for (int sy = -nmax[1]; sy <= nmax[1]; ++sy)
      for (int sz = -nmax[2]; sz <= nmax[2]; ++sz) {
        const int s[3] = {sx, sy, sz};
        for (int j = 0; j < n; ++j) {
          Ghost g;
          bool inside = true;
          for (int a = 0; a < 3; ++a) {
            g.f[a] = fw[3 * (size_t)j + a] + s[a];
            if (g.f[a] < lo[a] - tol || g.f[a] > hi[a] + tol) {
              inside = false;
              break;
            }
          }
          if (!inside) continue;
          g.j = j;
          g.sx = sx;
          g.sy = sy;
          g.sz = sz;
          for (int b = 0; b < 3; ++b)
            g.x[b] = rw[3 * (size_t)j + b] + sx * h[b] + sy * h[3 + b] + sz * h[6 + b];
          ghosts.push_back(g);
        }
      }

  // bins of fractional width >= pad (perpendicular width >= rmax)
  int nb[3];
  for (int a = 0; a < 3; ++a) {
    double ext = hi[a] - lo[a];
    int k = (pad[a] > 0 && ext > 0) ? (int)std::floor(ext / pad[a]) : 1;
    nb[a] = std::max(1, std::min(k, 256));
  }
  auto bin_of = [&](const double *f, int *b) {
    for (int a = 0; a < 3; ++a) {
      double ext = hi[a] - lo[a];
      int k = ext > 0 ? (int)std::floor((f[a] - lo[a]) / ext * nb[a]) : 0;
      b[a] = std::max(0, std::min(nb[a] - 1, k));
    }
  };
  const size_t nbins = (size_t)nb[0] * nb[1] * nb[2];
  std::vector<int32_t> bin_start(nbins + 1, 0), bin_items(ghosts.size());
  std::vector<int32_t> gbin(ghosts.size());
  for (size_t g = 0; g < ghosts.size(); ++g) {
    int b[3];
    bin_of(ghosts[g].f, b);
    gbin[g] = (b[0] * nb[1] + b[1]) * nb[2] + b[2];
    bin_start[gbin[g] + 1]++;
  }
  for (size_t k = 0; k < nbins; ++k) bin_start[k + 1] += bin_start[k];
  {
    std::vector<int32_t> fill(bin_start.begin(), bin_start.end() - 1);
    for (size_t g = 0; g < ghosts.size(); ++g) bin_items[fill[gbin[g]]++] = (int32_t)g;
  }

  const double r2pre = rmax * rmax * (1.0 + 1e-9);
  std::vector<std::vector<Cand>> per_atom(n);
  for (int i = 0; i < n; ++i) {
    int bi[3];
    bin_of(&fw[3 * (size_t)i], bi);
    const double *ri0 = &fr.positions[3 * (size_t)i];
    auto &list = per_atom[i];
    for (int dx = -1; dx <= 1; ++dx) {
      int bx = bi[0] + dx;
      if (bx < 0 || bx >= nb[0]) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        int by = bi[1] + dy;
        if (by < 0 || by >= nb[1]) continue;
        for (int dz = -1; dz <= 1; ++dz) {
          int bz = bi[2] + dz;
          if (bz < 0 || bz >= nb[2]) continue;
          size_t bin = ((size_t)bx * nb[1] + by) * nb[2] + bz;
          for (int32_t t = bin_start[bin]; t < bin_start[bin + 1]; ++t) {
            const Ghost &g = ghosts[bin_items[t]];
            if (g.j == i && g.sx == 0 && g.sy == 0 && g.sz == 0) continue;
            double d0 = g.x[0] - rw[3 * (size_t)i], d1 = g.x[1] - rw[3 * (size_t)i + 1],
                   d2 = g.x[2] - rw[3 * (size_t)i + 2];
            if (d0 * d0 + d1 * d1 + d2 * d2 >= r2pre) continue;
            // exact test on D = R_j - R_i + S.h with S relative to the given positions
            Cand c;
            c.j = g.j;
            c.sx = g.sx - wrap[3 * (size_t)g.j] + wrap[3 * (size_t)i];
            c.sy = g.sy - wrap[3 * (size_t)g.j + 1] + wrap[3 * (size_t)i + 1];
            c.sz = g.sz - wrap[3 * (size_t)g.j + 2] + wrap[3 * (size_t)i + 2];
            const double *rj0 = &fr.positions[3 * (size_t)g.j];
            double D[3];
            for (int b = 0; b < 3; ++b)
              D[b] = rj0[b] - ri0[b] + (c.sx * h[b] + c.sy * h[3 + b] + c.sz * h[6 + b]);
            c.r2 = D[0] * D[0] + D[1] * D[1] + D[2] * D[2];
            if (!(std::sqrt(c.r2) < rmax)) continue;
            c.species = fr.species[g.j];
            list.push_back(c);
          }
        }
      }
    }
    std::sort(list.begin(), list.end(), [](const Cand &a, const Cand &b) {
      if (a.species != b.species) return a.species < b.species;
      if (a.r2 != b.r2) return a.r2 < b.r2;
      if (a.j != b.j) return a.j < b.j;
      if (a.sx != b.sx) return a.sx < b.sx;
      if (a.sy != b.sy) return a.sy < b.sy;
      return a.sz < b.sz;
    });
    out.count[i] = (int32_t)list.size();
  }
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += per_atom[i].size();
  out.cands.reserve(total);
  for (int i = 0; i < n; ++i) out.cands.insert(out.cands.end(), per_atom[i].begin(), per_atom[i].end());
}

}  // namespace

void build_pairs(int32_t n_frames, const ta_frame *frames, int32_t n_elements, double rmax,
                 HostPairs &out) {
  if (n_frames < 0) throw std::runtime_error("negative frame count");
  std::vector<FramePairs> fp(n_frames);
  {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned nthreads = std::min<unsigned>(hw, (unsigned)std::max(1, n_frames));
    if (nthreads <= 1) {
      for (int f = 0; f < n_frames; ++f) build_frame(frames[f], n_elements, rmax, fp[f]);
    } else {
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t]() {
          for (int f = (int)t; f < n_frames; f += (int)nthreads)
            build_frame(frames[f], n_elements, rmax, fp[f]);
        });
      for (auto &th : pool) th.join();
    }
  }
  for (int f = 0; f < n_frames; ++f)
    if (!fp[f].error.empty())
      throw std::runtime_error("frame " + std::to_string(f) + ": " + fp[f].error);

  int64_t n_atoms = 0, n_pairs = 0;
  out.atom_start.assign(n_frames + 1, 0);
  for (int f = 0; f < n_frames; ++f) {
    n_atoms += frames[f].n_atoms;
    n_pairs += (int64_t)fp[f].cands.size();
    out.atom_start[f + 1] = (int32_t)n_atoms;
  }
  if (n_atoms > INT32_MAX / 4 || n_pairs > INT32_MAX / 2)
    throw std::runtime_error("batch too large for 32-bit pair indices");
  out.n_atoms = n_atoms;
  out.n_pairs = n_pairs;
  out.pair_start.assign(n_atoms + 1, 0);
  out.seg_start.assign((size_t)n_atoms * (n_elements + 1), 0);
  out.pair_i.resize(n_pairs);
  out.pair_j.resize(n_pairs);
  out.pair_shift.resize(3 * (size_t)n_pairs);
  out.pair_rev.assign(n_pairs, -1);
  out.frame_of_atom.resize(n_atoms);
  out.n_triples = 0;
  out.nnl_max = 0;

  int64_t p = 0;
  for (int f = 0; f < n_frames; ++f) {
    const int32_t a0 = out.atom_start[f];
    size_t c = 0;
    for (int i = 0; i < frames[f].n_atoms; ++i) {
      const int32_t gi = a0 + i;
      out.frame_of_atom[gi] = f;
      out.pair_start[gi] = (int32_t)p;
      const int32_t cnt = fp[f].count[i];
      out.nnl_max = std::max(out.nnl_max, cnt);
      out.n_triples += (int64_t)cnt * (cnt - 1) / 2;
      int32_t *seg = &out.seg_start[(size_t)gi * (n_elements + 1)];
      int s_cur = 0;
      seg[0] = (int32_t)p;
      for (int k = 0; k < cnt; ++k, ++c, ++p) {
        const Cand &cd = fp[f].cands[c];
        while (s_cur < cd.species) seg[++s_cur] = (int32_t)p;
        out.pair_i[p] = gi;
        out.pair_j[p] = a0 + cd.j;
        out.pair_shift[3 * p + 0] = cd.sx;
        out.pair_shift[3 * p + 1] = cd.sy;
        out.pair_shift[3 * p + 2] = cd.sz;
      }
      while (s_cur < n_elements) seg[++s_cur] = (int32_t)p;
    }
  }
  out.pair_start[n_atoms] = (int32_t)p;

  // reverse pairs: (i -> j, S)  <->  (j -> i, -S)
  auto find_rev = [&](int64_t lo_p, int64_t hi_p) {
    for (int64_t q = lo_p; q < hi_p; ++q) {
      const int32_t i = out.pair_i[q], j = out.pair_j[q];
      const int32_t *s = &out.pair_shift[3 * q];
      int32_t found = -1;
      for (int32_t t = out.pair_start[j]; t < out.pair_start[j + 1]; ++t) {
        if (out.pair_j[t] != i) continue;
        const int32_t *u = &out.pair_shift[3 * (size_t)t];
        if (u[0] == -s[0] && u[1] == -s[1] && u[2] == -s[2]) {
          found = t;
          break;
        }
      }
      out.pair_rev[q] = found;
    }
  };
  {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned nthreads = (n_pairs > 200000) ? hw : 1;
    if (nthreads <= 1) {
      find_rev(0, n_pairs);
    } else {
      std::vector<std::thread> pool;
      int64_t chunk = (n_pairs + nthreads - 1) / nthreads;
      for (unsigned t = 0; t < nthreads; ++t) {
        int64_t a = t * chunk, b = std::min<int64_t>(n_pairs, a + chunk);
        if (a < b) pool.emplace_back(find_rev, a, b);
      }
      for (auto &th : pool) th.join();
    }
  }
  for (int64_t q = 0; q < n_pairs; ++q)
    if (out.pair_rev[q] < 0)
      throw std::runtime_error("neighbour list is not symmetric (reverse pair missing)");
}

}  // namespace ta
