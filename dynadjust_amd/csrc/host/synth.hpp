// Synthetic network generator (see synth.cpp); writes <dir>/<name>.{bst,bms,asl,seg,truth}.
#pragma once
#include <cstdint>
#include <string>

namespace dynadjust {
namespace synth {

struct Spec {
    uint32_t rows = 10, cols = 10;
    uint64_t n_baselines = 0;   // 0 = keep every E / N / NE neighbour baseline
    uint32_t n_blocks = 1;
    uint64_t seed = 20260928;
    double initial_sigma = 0.05;             // metres, perturbation of the initial coordinates per axis
    double sigma_e = 0.003, sigma_n = 0.003, sigma_up = 0.006;   // baseline noise in the local frame
    uint32_t x_clusters = 0;   // the baselines leaving each of the first x_clusters stations form one 'X' cluster (correlated VCV)
    bool y_cluster = false;    // datum from two 'Y' point clusters over the corner stations instead of CCC constraints
    bool y_llh = false;        // ... given as latitude / longitude / height ("LLh" and "LLH") with geographic variance matrices
    bool scalars = false;      // v- / p- / l- / h-scale columns set on part of the measurements
};

struct Summary {
    uint64_t stations = 0, baselines = 0, measurement_rows = 0, blocks = 0, max_block_unknowns = 0;
};

void write_network(const std::string& dir, const std::string& name, const Spec& spec, Summary* summary = nullptr);

}  // namespace synth
}  // namespace dynadjust
