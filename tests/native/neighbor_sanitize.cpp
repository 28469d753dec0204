// CPU sanitizer driver for the threaded host neighbour list (csrc/ta_neighbor.cpp).
// Built by tests/test_host_logic.py with -fsanitize=address,undefined; reads frames from stdin,
// prints the pair count and order-independent checksums of the list, and checks the list's invariants.
//
// stdin:  n_frames n_elements rmax, then per frame: n_atoms pbc[3] cell[9] (species x y z) * n_atoms
// stdout: one line "pairs <P> triples <T> nnl <M> sum_i <..> sum_j <..> sum_s <..>"; exit 0
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "ta_internal.h"

namespace {
[[noreturn]] void fail(const char *what) {
  std::fprintf(stderr, "invariant violated: %s\n", what);
  std::exit(3);
}
}  // namespace

int main() {
  int n_frames = 0, n_elements = 0;
  double rmax = 0.0;
  if (std::scanf("%d %d %lf", &n_frames, &n_elements, &rmax) != 3) return 2;
  std::vector<std::vector<int32_t>> species(n_frames), pbc(n_frames);
  std::vector<std::vector<double>> pos(n_frames), cell(n_frames);
  std::vector<ta_frame> frames(n_frames);
  for (int f = 0; f < n_frames; ++f) {
    int n = 0;
    if (std::scanf("%d", &n) != 1) return 2;
    pbc[f].resize(3);
    cell[f].resize(9);
    for (int k = 0; k < 3; ++k)
      if (std::scanf("%d", &pbc[f][k]) != 1) return 2;
    for (int k = 0; k < 9; ++k)
      if (std::scanf("%lf", &cell[f][k]) != 1) return 2;
    species[f].resize(n);
    pos[f].resize(3 * (size_t)n);
    for (int a = 0; a < n; ++a)
      if (std::scanf("%d %lf %lf %lf", &species[f][a], &pos[f][3 * a], &pos[f][3 * a + 1], &pos[f][3 * a + 2]) != 4)
        return 2;
    frames[f] = ta_frame{n, species[f].data(), pos[f].data(), cell[f].data(), pbc[f].data()};
  }
  ta::HostPairs hp;
  try {
    ta::build_pairs(n_frames, frames.data(), n_elements, rmax, hp);
  } catch (const std::runtime_error &e) {
    std::printf("error %s\n", e.what());
    return 0;
  }
  const int64_t P = hp.n_pairs, N = hp.n_atoms;
  if ((int64_t)hp.pair_start.size() != N + 1 || hp.pair_start[N] != P) fail("pair_start");
  if ((int64_t)hp.pair_i.size() != P || (int64_t)hp.pair_j.size() != P || (int64_t)hp.pair_rev.size() != P ||
      (int64_t)hp.pair_shift.size() != 3 * P)
    fail("array sizes");
  long long sum_i = 0, sum_j = 0, sum_s = 0;
  for (int64_t i = 0; i < N; ++i) {
    const int32_t *seg = hp.seg_start.data() + (size_t)i * (n_elements + 1);
    if (seg[0] != hp.pair_start[i] || seg[n_elements] != hp.pair_start[i + 1]) fail("seg_start ends");
    for (int s = 0; s < n_elements; ++s)
      if (seg[s] > seg[s + 1]) fail("seg_start order");
    for (int32_t p = hp.pair_start[i]; p < hp.pair_start[i + 1]; ++p)
      if (hp.pair_i[p] != i) fail("pair_i");
  }
  for (int64_t p = 0; p < P; ++p) {
    const int32_t q = hp.pair_rev[p];
    if (q < 0 || q >= P) fail("pair_rev range");
    if (hp.pair_rev[q] != p) fail("pair_rev involution");
    if (hp.pair_i[q] != hp.pair_j[p] || hp.pair_j[q] != hp.pair_i[p]) fail("pair_rev ends");
    for (int k = 0; k < 3; ++k)
      if (hp.pair_shift[3 * q + k] != -hp.pair_shift[3 * p + k]) fail("pair_rev shift");
    if (hp.frame_of_atom[hp.pair_i[p]] != hp.frame_of_atom[hp.pair_j[p]]) fail("pair crosses frames");
    sum_i += hp.pair_i[p];
    sum_j += (long long)hp.pair_j[p] * (hp.pair_i[p] % 7 + 1);
    sum_s += (long long)(hp.pair_shift[3 * p] + 3 * hp.pair_shift[3 * p + 1] + 9 * hp.pair_shift[3 * p + 2]) * (hp.pair_j[p] % 5 + 1);
  }
  std::printf("pairs %lld triples %lld nnl %d sum_i %lld sum_j %lld sum_s %lld\n", (long long)P,
              (long long)hp.n_triples, (int)hp.nnl_max, sum_i, sum_j, sum_s);
  return 0;
}
