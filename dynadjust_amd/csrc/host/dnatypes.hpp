// On-disk record layouts and the settings subset of DynAdjust that the adjustment path
// consumes.  The layouts are the reference's binary file formats (raw struct dumps made by
// GCC x86-64), so field order, types and padding are pinned by static_asserts:
//   station_t      = 352 bytes  (include/config/dnatypes-structs.hpp:270-323)
//   measurement_t  = 208 bytes  (include/measurement_types/dnameasurement.hpp:133-194)
// (paths under /root/reference/dynadjust/).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace dynadjust {

typedef uint32_t UINT32;
typedef uint16_t UINT16;

// widths, include/config/dnatypes-basic.hpp:66-76
constexpr int STN_NAME_WIDTH = 31;
constexpr int STN_NAME_ORIG_WIDTH = 40;
constexpr int STN_DESC_WIDTH = 129;
constexpr int STN_CONST_WIDTH = 4;
constexpr int STN_TYPE_WIDTH = 4;
constexpr int STN_EPSG_WIDTH = 7;
constexpr int STN_EPOCH_WIDTH = 12;
constexpr int STN_PLATE_WIDTH = 3;
constexpr int MOD_NAME_WIDTH = 20;
constexpr int FILE_NAME_WIDTH = 256;

// include/config/dnatypes-basic.hpp:129-132, 161-162
enum { XYZ_type_i = 0, LLh_type_i = 1, LLH_type_i = 2, UTM_type_i = 3 };
enum { ORTHOMETRIC_type_i = 0, ELLIPSOIDAL_type_i = 1 };

struct station_t {
    char stationName[STN_NAME_WIDTH];
    char stationNameOrig[STN_NAME_ORIG_WIDTH];
    char stationConst[STN_CONST_WIDTH];   // "CCC", "FFF", ... lat, long, height
    char stationType[STN_TYPE_WIDTH];     // "LLH", "UTM", "XYZ"
    UINT16 suppliedStationType;
    double initialLatitude;
    double currentLatitude;               // radians
    double initialLongitude;
    double currentLongitude;              // radians
    double initialHeight;
    double currentHeight;                 // ellipsoidal, metres
    UINT16 suppliedHeightRefFrame;
    float geoidSep;
    float geoidSepUnc;
    double meridianDef;
    double verticalDef;
    short zone;
    char description[STN_DESC_WIDTH];
    UINT32 fileOrder;
    UINT32 nameOrder;
    UINT32 clusterID;
    UINT16 unusedStation;
    char epsgCode[STN_EPSG_WIDTH];
    char epoch[STN_EPOCH_WIDTH];
    char observation_epoch[STN_EPOCH_WIDTH];
    char plate[STN_PLATE_WIDTH];
};
static_assert(sizeof(station_t) == 352, "station_t must match the .bst record");
static_assert(offsetof(station_t, stationConst) == 71, "");
static_assert(offsetof(station_t, suppliedStationType) == 80, "");
static_assert(offsetof(station_t, currentLatitude) == 96, "");
static_assert(offsetof(station_t, currentLongitude) == 112, "");
static_assert(offsetof(station_t, currentHeight) == 128, "");
static_assert(offsetof(station_t, geoidSep) == 140, "");
static_assert(offsetof(station_t, fileOrder) == 300, "");

struct measurement_t {
    char measType;              // 'G' GPS baseline, 'X' baseline cluster, 'Y' point cluster, ...
    char measStart;             // 0 = X element, 1 = Y, 2 = Z, 3..5 covariance rows
    char measurementStations;
    char epsgCode[7];
    char epoch[STN_EPOCH_WIDTH];
    char observation_epoch[STN_EPOCH_WIDTH];
    char coordType[4];
    bool ignore;
    UINT32 station1;
    UINT32 station2;
    UINT32 station3;
    UINT32 vectorCount1;
    UINT32 vectorCount2;
    UINT32 clusterID;
    UINT32 fileOrder;
    UINT32 sourceFileIndex;
    double term1;               // measurement (dX / dY / dZ)
    double term2;               // XX | XY | XZ variance
    double term3;               // YY | YZ
    double term4;               // ZZ
    double scale1, scale2, scale3, scale4;   // phi, lambda, height, matrix (v) scalars
    double measAdj, measCorr, measAdjPrec, residualPrec, NStat, TStat, PelzerRel, preAdjCorr, preAdjMeas;
};
static_assert(sizeof(measurement_t) == 208, "measurement_t must match the .bms record");
static_assert(offsetof(measurement_t, ignore) == 38, "");
static_assert(offsetof(measurement_t, station1) == 40, "");
static_assert(offsetof(measurement_t, vectorCount1) == 52, "");
static_assert(offsetof(measurement_t, clusterID) == 60, "");
static_assert(offsetof(measurement_t, term1) == 72, "");
static_assert(offsetof(measurement_t, scale1) == 104, "");
static_assert(offsetof(measurement_t, measAdj) == 136, "");
static_assert(offsetof(measurement_t, preAdjMeas) == 200, "");

// include/config/dnatypes-structs.hpp:332-420 (as written by DynadjustFile::WriteFileMetadata,
// include/io/dynadjust_file.cpp:83-117)
struct input_file_meta_t {
    char filename[FILE_NAME_WIDTH + 1];
    char epsgCode[STN_EPSG_WIDTH];
    char epoch[STN_EPOCH_WIDTH];
    char observation_epoch[STN_EPOCH_WIDTH];
    UINT16 filetype;
    UINT16 datatype;
};
struct source_file_meta_t {
    char filename[FILE_NAME_WIDTH + 1];
};
struct binary_file_meta_t {
    uint64_t binCount = 0;
    bool reduced = false;
    char modifiedBy[MOD_NAME_WIDTH + 1] = {0};
    char epsgCode[STN_EPSG_WIDTH] = {0};
    char epoch[STN_EPOCH_WIDTH] = {0};
    char observation_epoch[STN_EPOCH_WIDTH] = {0};
    bool reftran = false;
    bool geoid = false;
    std::vector<input_file_meta_t> inputFileMeta;
    std::vector<source_file_meta_t> sourceFileMeta;
};

// .asl record: include/functions/dnatemplatestnmsrfuncs.hpp:884-918
struct asl_entry_t {
    UINT32 assocMsrCount = 0;
    UINT32 amlStnIndex = 0;
    UINT16 validity = 1;   // 1 = valid station
};

// block metadata, include/config/dnatypes-structs.hpp:258-268
struct blockMeta_t {
    bool _blockIsolated = false, _blockFirst = false, _blockLast = false, _blockIntermediate = false;
};

// station appearance, include/config/dnatypes-structs.hpp:36-61
struct stn_appear {
    UINT32 station_id = 0;
    bool first_appearance_fwd = false;
    bool first_appearance_rev = false;
};

// include/config/dnaoptions.hpp:50-52
enum { SimultaneousMode = 0, PhasedMode = 1, Phased_Block_1Mode = 2 };

// include/config/dnatypes-basic.hpp:79-84
typedef enum _SIGMA_ZERO_STAT_PASS_ { test_stat_pass = 0, test_stat_warning = 1, test_stat_fail = 2 } SIGMA_ZERO_STAT_PASS;

// include/config/dnaconsts.hpp:119-120
constexpr double UNRELIABLE = 999.99;
constexpr double STABLE_LIMIT = 700.0;
constexpr double PRECISION_1E10 = 1.0e-10;

// include/exception/dnaexception.hpp:51-59
typedef enum _ADJUST_STATUS_ {
    ADJUST_SUCCESS = 0,
    ADJUST_MAX_ITERATIONS_EXCEEDED = 1,
    ADJUST_THRESHOLD_EXCEEDED = 2,
    ADJUST_TEST_FAILED = 3,
    ADJUST_BLOCK_ERROR = 4,
    ADJUST_EXCEPTION_RAISED = 5,
    ADJUST_CANCELLED = 6
} ADJUST_STATUS;

// The members of project_settings (include/config/dnaoptions.hpp) that dna_adjust reads on this path.
struct general_settings {
    std::string network_name;
    std::string output_folder = ".";
    std::string input_folder = ".";
    UINT16 verbose = 0;
    UINT16 quiet = 0;
};
struct segment_settings {
    std::string asl_file;
    std::string aml_file;
    std::string seg_file;
};
struct adjust_settings {
    UINT16 adjust_mode = SimultaneousMode;
    UINT16 max_iterations = 10;
    float confidence_interval = 95.0f;
    UINT16 report_mode = 0;
    UINT16 multi_thread = 0;
    UINT16 stage = 0;
    UINT16 scale_normals_to_unity = 0;
    // Not in the reference (device path only): keep every block's forward / reverse / combined inverse resident and
    // reuse it from the second iteration on.  For a GNSS-only network neither the design nor the weights change between
    // iterations, so those inverses are bit-identical every time; the reference exploits this in simultaneous mode only
    // (dnaadjust.cpp:2457).  Costs two more n x n matrices per block in HBM.
    UINT16 reuse_inverses = 0;
    // Not in the reference (device path only): a forward / reverse step that only feeds the next block (every one but the
    // last forward and the first reverse step of a network) needs the junction stations' weight matrix and estimates, not
    // the block inverse: the inner unknowns are eliminated (dnagpu_schur_carry) instead of Solve()'s full inverse.
    UINT16 schur_carry = 1;
    // with schur_carry: the condensing step of a block keeps its factor resident and the block's rigorous solve completes it
    // (dnagpu_block_reduce(keep) + dnagpu_partial_complete) instead of forming and inverting the block's normals again
    UINT16 keep_factors = 1;
    float iteration_threshold = 0.0005f;
    double free_std_dev = 10.0;
    double fixed_std_dev = 1.0e-6;   // PRECISION_1E6
    std::string bst_file, bms_file, seg_file;
    std::string stage_path;          // where <network>-rva.mtx / -pam.mtx go (default: g.output_folder)
    int max_threads = 0;
    // device selection (not in the reference): which GPU this process drives
    int device = 0;
    // Multi-GPU (not in the reference).  Either one process per GPU -- this one is rank dist_rank of dist_world
    // (the launcher's RANK / WORLD_SIZE), its GPU is `device` -- or one process for all of them: `devices` lists the GPUs and
    // AdjustNetwork() spreads the blocks over them from one host thread per GPU.  The exchange step runs over RCCL
    // (dist_transport "rccl", the default wherever every rank has a GPU of its own) or, for ranks of one process, through
    // device-to-device copies ("local").
    int dist_rank = 0;
    int dist_world = 1;
    std::vector<int> devices;
    std::string dist_transport;      // "" = choose, "rccl", "local"
    // condensed chains across ranks: 1 (default) = two-level where possible (each rank reduces its own run of blocks, the ranks'
    // boundary systems are scanned, every rank finishes its own blocks); 0 = every rank runs both chains on all condensed blocks
    // (one broadcast per block)
    UINT16 dist_two_level = 1;
    // condensed schedule with kept factors: an iteration takes its corrections from the completed FACTOR of every block (two triangular
    // matrix-vector products) and the inverses -- the rigorous variance matrices, n^3 / 3 per block -- are formed once, after the last
    // iteration, instead of in every iteration.  Same results (the variances are those of the last iteration's normals either way).
    // 2 (default): the light form on top -- the condensing step stops at the factor (~0.34 n_i^3 instead of 2/3 n_i^3), the iterations
    // substitute block by block, and the inverse of the factor is paid once, with the variance matrices, after the last iteration.
    // 1: the condensing step inverts the eliminated part's factor in every iteration (needed by nothing but the final inverse).
    UINT16 defer_variances = 2;
    // blocks of one shape (equal padded orders of the eliminated and the kept part) go through the large steps of the condensed
    // schedule as ONE batch of up to this many members: merged launches, in lock step (include/dnagpu.h, dnagpu_*_batched).  The
    // reference's blocks have no such coupling -- its Solve() calls follow each other (ADJ:2812, ADJ:3512, ADJ:3556) -- and every
    // member's results are the bits of the unbatched calls.  0 / 1: off.  DNAGPU_BATCH overrides.
    UINT16 batch_blocks = 32;
    // Not in the reference's phased mode (it has the same licence in simultaneous mode: SolveTry(CurrentIteration() < 2 ||
    // ContainsNonGPS()), dnaadjust.cpp:2452-2457): in a GNSS-only network neither the design nor the weights move with the estimates,
    // so the normals of every block -- and with them every factor of the condensed schedule: the blocks' light factors, the kept
    // blocks' factors, the factors of the chain steps on the condensed blocks -- are the same in every iteration.  1 (default):
    // iterations >= 2 keep the factors of iteration 1 and renew right-hand sides only (substitutions at HBM speed); the variance
    // matrices are still formed once, after the last iteration.  0: every iteration factors again.  Needs the condensed schedule
    // with light kept factors (schur_carry, keep_factors, defer_variances = 2); blocks that may not keep a factor (HBM budget)
    // go on as before.
    UINT16 reuse_factors = 1;
    // condensed schedule on one GPU, many small blocks: the two junction chains cut into this many runs whose steps advance together in
    // merged launches (dna_adjust::LockstepChains).  -1 = choose (32 runs from 512 blocks, 16 from 64, else one), 0 / 1 = one run.
    int chain_runs = -1;
};
struct output_settings {
    UINT16 _adj_msr_tstat = 0;   // --output-tstat-adj-msr: Student's t statistic of every adjusted measurement
};
struct project_settings {
    general_settings g;
    segment_settings s;
    adjust_settings a;
    output_settings o;
};

}  // namespace dynadjust
