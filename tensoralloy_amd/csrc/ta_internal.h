// Internal declarations shared by the translation units of libtensoralloy_amd.so.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "tensoralloy_amd.h"

namespace ta {

// Host-side pair list of a batch of frames, sorted by centre atom, and inside a
// centre by (neighbour species, r^2, j, shift). Semantics of the list = ASE
// `neighbor_list('ijS')` as used by reference transformer/universal.py:58.
struct HostPairs {
  int64_t n_atoms = 0;
  int64_t n_pairs = 0;
  int64_t n_triples = 0;
  int32_t nnl_max = 0;
  std::vector<int32_t> pair_start;   // [n_atoms + 1]
  std::vector<int32_t> seg_start;    // [n_atoms * (n_elements + 1)] absolute pair offsets
  std::vector<int32_t> pair_i;       // [P] centre (global atom index)
  std::vector<int32_t> pair_j;       // [P] neighbour (global atom index)
  std::vector<int32_t> pair_shift;   // [P * 3] integer cell shifts S
  std::vector<int32_t> pair_rev;     // [P] index of the reverse pair (j -> i, -S)
  std::vector<int32_t> frame_of_atom;  // [n_atoms]
  std::vector<int32_t> atom_start;     // [n_frames + 1]
};

// Builds the list for all frames (cutoff `rmax`, strict `<`). Throws
// std::runtime_error on invalid input.
void build_pairs(int32_t n_frames, const ta_frame *frames, int32_t n_elements,
                 double rmax, HostPairs &out);

}  // namespace ta
