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
    // Uneven segmentations (dnasegment cuts a real network into blocks of whatever its --max-block-stns / --min-inner-stns thresholds and the
    // network's shape give, dnasegment.cpp:235-348, 528-700; default threshold 150 stations, dnaoptions.hpp:381-382):
    double ragged = 0.0;       // strip heights proportional to 1 + ragged * U(-1, 1) instead of equal (0 <= ragged < 1), n_blocks strips
    uint32_t rows_lo = 0, rows_hi = 0;   // rows_hi > 0: strip heights drawn uniformly from [rows_lo, rows_hi] rows until the grid is used up
                                         // (n_blocks is then a result, not an input): blocks of rows_lo * cols ... rows_hi * cols stations
};

struct Summary {
    uint64_t stations = 0, baselines = 0, measurement_rows = 0, blocks = 0, max_block_unknowns = 0;
};

void write_network(const std::string& dir, const std::string& name, const Spec& spec, Summary* summary = nullptr);

}  // namespace synth
}  // namespace dynadjust
