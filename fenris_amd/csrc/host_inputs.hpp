// Host-side helpers shared between host_inputs.cpp and the engine (engine.hip).
#pragma once
#include <cstdint>
#include <vector>

namespace fenris_hip {

bool gauss_rule(unsigned n, std::vector<double>& w, std::vector<double>& x);
bool tensor_rule(unsigned dim, unsigned n, double* w_out, double* p_out);

// reference-identical greedy colouring of ragged element node lists (fenris-paradis coloring.rs:6-70)
void greedy_coloring(uint64_t num_elements, const uint64_t* elem_offsets, const uint64_t* elem_nodes,
                     std::vector<uint64_t>& color_offsets, std::vector<uint64_t>& labels);

}  // namespace fenris_hip
