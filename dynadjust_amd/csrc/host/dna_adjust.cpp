#include "dna_adjust.hpp"
#include "statfuncs.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>

#include "geodesy.hpp"
#include "gnss_vcv.hpp"
#include "../terrestrial.h"

namespace dynadjust {
namespace networkadjust {

namespace {
double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
constexpr double PRECISION_1E5 = 1.0e-5;
}  // namespace

dna_adjust::dna_adjust() {}

dna_adjust::~dna_adjust() {
    peers_.clear();
    FreeDevice();
}

void dna_adjust::FreeDevice() {
    if (!ctx_) return;
    if (small_batch_) dnagpu_small_batch_destroy(ctx_, small_batch_);       // (before what it refers to: factors, junction matrices, blocks)
    small_batch_ = nullptr;
    small_batch_blocks_.clear();
    small_batch_denied_ = small_batch_armed_ = false;
    FreeLockstepChains();      // (before the matrices and blocks its plan refers to)
    FreeTwoLevel();
    if (xbuf_dev_) dnagpu_device_free(ctx_, xbuf_dev_);
    xbuf_dev_ = nullptr;
    if (block_table_) dnagpu_block_table_destroy(ctx_, block_table_);
    block_table_ = nullptr;
    block_table_denied_ = false;
    if (initial_dev_) dnagpu_device_free(ctx_, initial_dev_);
    initial_dev_ = nullptr;
    if (agree_dev_) dnagpu_device_free(ctx_, agree_dev_);
    agree_dev_ = nullptr;
    xbuf_cap_ = 0;
    for (block_t& b : blocks_) {
        if (b.jfwd) dnagpu_matrix_destroy(ctx_, b.jfwd);
        if (b.jrev) dnagpu_matrix_destroy(ctx_, b.jrev);
        if (b.rigvar) dnagpu_matrix_destroy(ctx_, b.rigvar);
        if (b.finv) dnagpu_matrix_destroy(ctx_, b.finv);
        if (b.rinv) dnagpu_matrix_destroy(ctx_, b.rinv);
        if (b.red) dnagpu_matrix_destroy(ctx_, b.red);
        if (b.rig_host) (b.rig_on_device ? dnagpu_device_free : dnagpu_host_free)(ctx_, b.rig_host);
        b.rig_host = nullptr;
        if (b.part && !b.part_transient) dnagpu_partial_destroy(ctx_, b.part);
        b.part = nullptr;
        for (dnagpu_partial*& tp : b.tpart) {
            if (tp) dnagpu_partial_destroy(ctx_, tp);
            tp = nullptr;
        }
        for (int d = 0; d < 2; ++d) {
            if (b.cfac[d]) dnagpu_partial_destroy(ctx_, b.cfac[d]);
            b.cfac[d] = nullptr;
            b.cfac_live[d] = b.cfac_denied[d] = false;
        }
        b.factor_live = b.factor_reused = false;
        b.jfwd = b.jrev = b.rigvar = b.finv = b.rinv = b.red = nullptr;
    }
    for (int c = 0; c < DNAGPU_NUM_CHAINS; ++c) {
        if (work_[c]) dnagpu_matrix_destroy(ctx_, work_[c]);
        if (kwork_[c]) dnagpu_matrix_destroy(ctx_, kwork_[c]);
        if (tmpfac_[c]) dnagpu_matrix_destroy(ctx_, tmpfac_[c]);
        work_[c] = kwork_[c] = tmpfac_[c] = nullptr;
        for (dnagpu_matrix* m : kbatch_[c]) dnagpu_matrix_destroy(ctx_, m);
        kbatch_[c].clear();
    }
    dnagpu_destroy(ctx_);
    ctx_ = nullptr;
    osc_ready_ = false;
}

// ADJ:10049-10069
void dna_adjust::SignalExceptionAdjustment(const std::string& msg, UINT32 block_no) {
    adjustStatus_ = ADJUST_EXCEPTION_RAISED;
    exceptionRaised_ = true;
    isPreparing_ = false;
    isCombining_ = false;
    isAdjusting_ = false;
    std::string error_msg(msg);
    switch (projectSettings_.a.adjust_mode) {
        case Phased_Block_1Mode:
        case PhasedMode: {
            std::stringstream ss;
            ss << msg << std::endl << "  Phased adjustment terminated whilst processing block " << block_no + 1 << std::endl;
            error_msg = ss.str();
        }
    }
    throw NetAdjustException(error_msg, block_no);
}

void dna_adjust::Check(int rc, UINT32 block, const char* where) {
    if (rc == DNAGPU_OK) return;
    std::string msg = ctx_ ? dnagpu_last_error(ctx_) : "device context unavailable";
    if (rc == DNAGPU_ENOTPOSDEF)
        // matrix_2d::cholesky_inverse -> MatrixInversionFailure (dnamatrix_contiguous.cpp:983) -> SolveTry (ADJ:6575-6582)
        SignalExceptionAdjustment("Matrix inversion failed, the matrix is singular.", block);
    std::stringstream ss;
    ss << where << ": device error " << rc << " (" << msg << ")";
    SignalExceptionAdjustment(ss.str(), block);
}

UINT32 dna_adjust::LocalIndex(UINT32 block, UINT32 stn) const {
    const std::vector<UINT32>& l = v_parameterStationList_[block];
    auto it = std::lower_bound(l.begin(), l.end(), stn);
    if (it == l.end() || *it != stn) {
        std::stringstream ss;
        ss << "Station " << stn << " is not a parameter station of block " << block + 1 << ".";
        throw std::runtime_error(ss.str());
    }
    return (UINT32)(it - l.begin());
}

// ADJ:10107 LoadNetworkFiles (+ NetworkDataLoader): bst, asl, bms
void dna_adjust::LoadNetworkFiles() {
    iostreams::read_bst(projectSettings_.a.bst_file, bstBinaryRecords_, bst_meta_);
    iostreams::read_bms(projectSettings_.a.bms_file, bmsBinaryRecords_, bms_meta_);
    if (!projectSettings_.s.asl_file.empty())
        iostreams::read_asl(projectSettings_.s.asl_file, vAssocStnList_);
    else
        vAssocStnList_.assign(bstBinaryRecords_.size(), asl_entry_t());
    if (vAssocStnList_.size() != bstBinaryRecords_.size())
        throw std::runtime_error("LoadNetworkFiles(): the associated station list does not match the binary station file.");
}

// simultaneous mode: network_data_loader.cpp:103-134, measurement_processor.cpp:47-164
void dna_adjust::BuildSimultaneousLists() {
    blockCount_ = 1;
    v_ISL_.assign(1, {});
    v_JSL_.assign(1, {});
    v_CML_.assign(1, {});
    for (UINT32 s = 0; s < bstBinaryRecords_.size(); ++s)
        if (vAssocStnList_[s].validity) v_ISL_[0].push_back(s);
    UINT32 rows = 0, clusterID = 0;
    bool in_cluster = false;
    for (UINT32 m = 0; m < bmsBinaryRecords_.size(); ++m) {
        const measurement_t& r = bmsBinaryRecords_[m];
        if (r.ignore) continue;
        if (r.measStart > 2) continue;    // covariance rows (measurement_processor.cpp:79-83)
        if (r.measType == 'D') {
            // direction sets: the record of the reference direction enters the CML, one design row per angle
            // (measurement_processor.cpp:92-110); the records of the other directions carry no counts of their own
            if (r.measStart == 0 && r.vectorCount1 >= 1) {
                v_CML_[0].push_back(m);
                rows += r.vectorCount2 > 0 ? r.vectorCount2 - 1 : 1;
            }
            continue;
        }
        rows++;                           // every X / Y / Z element (and every terrestrial measurement) is one design row
        if (r.measStart != 0) continue;   // only the first element starts a measurement
        if (r.measType == 'X' || r.measType == 'Y') {
            // only the first vector of a cluster enters the CML (measurement_processor.cpp:86-122)
            if (in_cluster && clusterID == r.clusterID) continue;
            in_cluster = true;
            clusterID = r.clusterID;
        }
        v_CML_[0].push_back(m);
    }
    v_ContiguousNetList_.assign(1, 0);
    v_measurementCount_.assign(1, rows);
    v_unknownsCount_.assign(1, (UINT32)v_ISL_[0].size() * 3);
    v_parameterStationCount_.assign(1, (UINT32)v_ISL_[0].size());
}

// ADJ:10426-10626
void dna_adjust::LoadSegmentationMetrics() {
    v_blockMeta_.assign(blockCount_, blockMeta_t());
    v_parameterStationList_.assign(blockCount_, {});
    UINT32 netID = 999999;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        v_blockMeta_[b]._blockFirst = (netID != v_ContiguousNetList_[b]);
        netID = v_ContiguousNetList_[b];
        if (b < blockCount_ - 1)
            v_blockMeta_[b]._blockLast = (v_ContiguousNetList_[b] != v_ContiguousNetList_[b + 1]);
        else
            v_blockMeta_[b]._blockLast = true;
        if (v_blockMeta_[b]._blockFirst && v_blockMeta_[b]._blockLast)
            v_blockMeta_[b]._blockIsolated = true;
        else if (!v_blockMeta_[b]._blockFirst && !v_blockMeta_[b]._blockLast)
            v_blockMeta_[b]._blockIntermediate = true;
        std::vector<UINT32>& p = v_parameterStationList_[b];
        p = v_ISL_[b];
        p.insert(p.end(), v_JSL_[b].begin(), v_JSL_[b].end());
        std::sort(p.begin(), p.end());
        if (std::adjacent_find(p.begin(), p.end()) != p.end())
            throw std::runtime_error("LoadSegmentationMetrics(): a station is listed twice in one block.");
        for (UINT32 s : p)
            if (s >= bstBinaryRecords_.size()) throw std::runtime_error("LoadSegmentationMetrics(): station index out of range.");
    }
    // total unknown parameters over unique stations (ADJ:10559-10578)
    std::vector<UINT32> all;
    for (auto& p : v_parameterStationList_) all.insert(all.end(), p.begin(), p.end());
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    unknownsCount_ = (UINT32)all.size() * 3;
    unknownParams_ = unknownsCount_;
    for (UINT32 s : all)
        for (int c = 0; c < 3; ++c)
            if (bstBinaryRecords_[s].stationConst[c] == 'C') unknownParams_--;
    allStationsFixed_ = (unknownParams_ == 0 && unknownsCount_ > 0);
    measurementParams_ = 0;
    for (UINT32 b = 0; b < blockCount_; ++b) measurementParams_ += v_measurementCount_[b];
}

// SegFile::CreateStnAppearanceList, seg_file.cpp:432-487 (simultaneous: BuildSimultaneousStnAppearance ADJ:1834)
void dna_adjust::CreateStnAppearanceList() {
    v_paramStnAppearance_.assign(blockCount_, {});
    std::vector<char> free_fwd(bstBinaryRecords_.size(), 1), free_rev;
    for (size_t s = 0; s < free_fwd.size(); ++s)
        if (!vAssocStnList_[s].validity) free_fwd[s] = 0;
    free_rev = free_fwd;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        v_paramStnAppearance_[b].resize(v_parameterStationList_[b].size());
        for (size_t p = 0; p < v_parameterStationList_[b].size(); ++p) {
            stn_appear& a = v_paramStnAppearance_[b][p];
            a.station_id = v_parameterStationList_[b][p];
            if (free_fwd[a.station_id]) {
                a.first_appearance_fwd = true;
                free_fwd[a.station_id] = 0;
            }
        }
    }
    for (UINT32 bb = blockCount_; bb-- > 0;)
        for (stn_appear& a : v_paramStnAppearance_[bb])
            if (free_rev[a.station_id]) {
                a.first_appearance_rev = true;
                free_rev[a.station_id] = 0;
            }
}

// ADJ:2041-2137
void dna_adjust::FormConstraintStationVarianceMatrix(UINT32 stn, double w9[9]) const {
    const station_t& st = bstBinaryRecords_[stn];
    for (int i = 0; i < 9; ++i) w9[i] = 0.0;
    const char* c = st.stationConst;
    if (c[0] == 'C' && c[1] == 'C' && c[2] == 'C') {
        w9[0] = w9[4] = w9[8] = 1. / var_C_;
        return;
    }
    if (c[0] == 'F' && c[1] == 'F' && c[2] == 'F') {
        w9[0] = w9[4] = w9[8] = 1. / var_F_;
        return;
    }
    // mixed constraints: variances in the local frame, propagated to cartesian, then inverted
    double vl[3] = {0, 0, 0};
    const bool geographic = (st.suppliedStationType == LLH_type_i || st.suppliedStationType == LLh_type_i);
    const double v0 = (c[0] == 'F') ? var_F_ : var_C_;
    const double v1 = (c[1] == 'F') ? var_F_ : var_C_;
    if (geographic) {
        vl[1] = v0;   // latitude constraint acts in the north direction
        vl[0] = v1;   // longitude constraint acts in the east direction
    } else {
        vl[0] = v0;
        vl[1] = v1;
    }
    vl[2] = (c[2] == 'F') ? var_F_ : var_C_;
    double V[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (st.suppliedStationType == XYZ_type_i) {
        for (int i = 0; i < 3; ++i) V[i][i] = vl[i];
    } else {
        double R[3][3];
        geodesy::LocalToCartRotation(st.currentLatitude, st.currentLongitude, R);   // Vc = R Vl R^T
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += R[i][k] * vl[k] * R[j][k];
                V[i][j] = s;
            }
    }
    // FormInverseVarianceMatrix: 3x3 Cholesky inverse
    double l11 = std::sqrt(V[0][0]);
    double l21 = V[1][0] / l11, l31 = V[2][0] / l11;
    double l22 = std::sqrt(V[1][1] - l21 * l21);
    double l32 = (V[2][1] - l31 * l21) / l22;
    double l33 = std::sqrt(V[2][2] - l31 * l31 - l32 * l32);
    if (!(l11 > 0) || !(l22 > 0) || !(l33 > 0)) throw MatrixInversionFailure("Matrix inversion failed, the matrix is singular.");
    double t11 = 1 / l11, t22 = 1 / l22, t33 = 1 / l33;
    double t21 = -l21 * t11 * t22;
    double t32 = -l32 * t22 * t33;
    double t31 = -(l21 * t32 + l31 * t33) * t11;
    double W[3][3];
    W[0][0] = t11 * t11 + t21 * t21 + t31 * t31;
    W[1][0] = W[0][1] = t21 * t22 + t31 * t32;
    W[2][0] = W[0][2] = t31 * t33;
    W[1][1] = t22 * t22 + t32 * t32;
    W[2][1] = W[1][2] = t32 * t33;
    W[2][2] = t33 * t33;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) w9[j * 3 + i] = W[i][j];
}

// One CML entry -> vectors + the cluster's full variance matrix.
//   'G' UpdateDesignNormalMeasMatrices_G ADJ:5353 / LoadVarianceMatrix_G ADJ:4214
//   'X' UpdateDesignNormalMeasMatrices_X ADJ:6056 / LoadVarianceMatrix_X ADJ:4312
//   'Y' UpdateDesignNormalMeasMatrices_Y ADJ:6249 / LoadVarianceMatrix_Y ADJ:4494 (cartesian clusters)
// Record layout of a cluster in the .bms: per vector 3 records (X, Y, Z: term1 = value, term2..4 = upper triangle of its
// 3x3 variance block), followed by vectorCount2 covariance blocks of 3 records each (rows of the 3x3 block against the
// following vectors, term1..3).
void dna_adjust::ParseGnssMeasurement(UINT32 block, UINT32 m, block_t& B) {
    const measurement_t& first = bmsBinaryRecords_[m];
    const char type = first.measType;
    if ((type != 'G' && type != 'X' && type != 'Y') || first.measStart != 0) {
        std::stringstream ss;
        ss << "UpdateNormals(): measurement type '" << type << "' is not handled by the device path yet (GNSS types G, X, Y only).";
        SignalExceptionAdjustment(ss.str(), block);
    }
    // a Y cluster may be given as latitude / longitude / height (orthometric "LLH" or ellipsoidal "LLh")
    const bool llH = type == 'Y' && strncmp(first.coordType, "LLH", 3) == 0;
    const bool llh = type == 'Y' && strncmp(first.coordType, "LLh", 3) == 0;
    const bool geographic = llH || llh;
    if (type == 'Y' && !geographic && strncmp(first.coordType, "XYZ", 3) != 0)
        SignalExceptionAdjustment("UpdateDesignNormalMeasMatrices_Y(): unknown coordinate type of a GPS point cluster.", block);
    // LoadVarianceScaling (ADJ:4453-4491).  A "reduced" .bms was written by an earlier adjustment: its variances are
    // already scaled and propagated (ADJ:4190-4211)
    const bool reduced = bms_meta_.reduced;
    const double tiny = std::min(PRECISION_1E5, projectSettings_.a.fixed_std_dev);
    auto unit = [&](double v) { return v < tiny ? 1.0 : v; };
    double vScale = unit(first.scale4), pScale = unit(first.scale1), lScale = unit(first.scale2), hScale = unit(first.scale3);
    const bool scaleMatrix = !reduced && std::fabs(vScale - 1.0) > PRECISION_1E5;
    const bool scalePartial = !reduced && (std::fabs(pScale - 1.0) > PRECISION_1E5 || std::fabs(lScale - 1.0) > PRECISION_1E5 ||
                                           std::fabs(hScale - 1.0) > PRECISION_1E5);
    if (scalePartial && scaleMatrix) {
        pScale *= vScale;
        lScale *= vScale;
        hScale *= vScale;
    }
    const UINT32 k = (type == 'G') ? 1 : first.vectorCount1;
    if (k == 0) SignalExceptionAdjustment("PrepareAdjustment(): a GNSS cluster without vectors.", block);
    const UINT32 nc = 3 * k;
    std::vector<double> V((size_t)nc * nc, 0.0);   // column-major, both triangles
    std::vector<double> positions(3 * (size_t)k);  // where each vector's geographic frame is formed (ADJ:4352, ADJ:4572)
    // G / X: the matrix scalar is applied on the fly (ADJ:4236, ADJ:4360); Y: afterwards (ADJ:4650)
    const double fly = (type != 'Y' && scaleMatrix) ? vScale : 1.0;
    auto put = [&](UINT32 r, UINT32 c, double v) {
        if (fly != 1.0) v *= fly;
        V[(size_t)c * nc + r] = v;
        V[(size_t)r * nc + c] = v;
    };
    size_t idx = m;
    for (UINT32 j = 0; j < k; ++j) {
        if (idx + 2 >= bmsBinaryRecords_.size()) SignalExceptionAdjustment("PrepareAdjustment(): truncated GNSS cluster.", block);
        measurement_t& mx = bmsBinaryRecords_[idx];
        measurement_t& my = bmsBinaryRecords_[idx + 1];
        measurement_t& mz = bmsBinaryRecords_[idx + 2];
        const station_t& at = bstBinaryRecords_.at(mx.station1);
        positions[3 * j] = at.currentLatitude;
        positions[3 * j + 1] = at.currentLongitude;
        positions[3 * j + 2] = at.currentHeight;
        if (geographic && !reduced) {
            // UpdateDesignNormalMeasMatrices_Y (ADJ:6281-6318): the point becomes cartesian once and for all; the
            // original values stay in preAdjMeas, the original frame in station3
            double ellipsoidHeight = mz.term1;
            mx.preAdjMeas = mx.term1;
            my.preAdjMeas = my.term1;
            mz.preAdjMeas = mz.term1;
            if (llH && std::fabs(at.geoidSep) > 1.0e-4) {
                mz.preAdjCorr = at.geoidSep;
                ellipsoidHeight += mz.preAdjCorr;
            }
            double x, y, z;
            geodesy::GeoToCart(mx.term1, my.term1, ellipsoidHeight, &x, &y, &z);
            mx.term1 = x;
            my.term1 = y;
            mz.term1 = z;
            mx.station3 = llH ? 2u : 1u;   // "retain original reference frame" (ADJ:6317); _COORD_TYPE_: LLh_type_i = 1, LLH_type_i = 2
        }
        B.obs.push_back(mx.term1);
        B.obs.push_back(my.term1);
        B.obs.push_back(mz.term1);
        if (type == 'Y') {
            B.stn1.push_back(DNAGPU_NO_STATION);
            B.stn2.push_back(LocalIndex(block, mx.station1));
        } else {
            B.stn1.push_back(LocalIndex(block, mx.station1));
            B.stn2.push_back(LocalIndex(block, mx.station2));
        }
        const UINT32 r0 = 3 * j;
        put(r0, r0, mx.term2);
        put(r0, r0 + 1, my.term2);
        put(r0 + 1, r0 + 1, my.term3);
        put(r0, r0 + 2, mz.term2);
        put(r0 + 1, r0 + 2, mz.term3);
        put(r0 + 2, r0 + 2, mz.term4);
        const UINT32 ncov = (type == 'G') ? 0 : mx.vectorCount2;
        idx += 3;
        for (UINT32 c = 0; c < ncov; ++c) {
            if (idx + 2 >= bmsBinaryRecords_.size() || j + 1 + c >= k)
                SignalExceptionAdjustment("PrepareAdjustment(): malformed GNSS cluster covariances.", block);
            const UINT32 c0 = 3 * (j + 1 + c);
            for (UINT32 r = 0; r < 3; ++r) {
                const measurement_t& cv = bmsBinaryRecords_[idx + r];
                put(r0 + r, c0, cv.term1);
                put(r0 + r, c0 + 1, cv.term2);
                put(r0 + r, c0 + 2, cv.term3);
            }
            idx += 3;
        }
    }
    bool changed = fly != 1.0;
    if (type != 'Y') {
        if (scalePartial) {
            gnssvcv::ScaleGPSVCV(V, k, positions, pScale, lScale, hScale, false);       // ADJ:4263 / ADJ:4418
            changed = true;
        }
    } else if (!reduced) {
        if (scalePartial) {
            gnssvcv::ScaleGPSVCV(V, k, positions, pScale, lScale, hScale, geographic);  // ADJ:4633
            changed = true;
        } else if (geographic) {
            gnssvcv::PropagateGeoCart(V, k, positions, true);                            // ADJ:4640
            changed = true;
        }
        if (scaleMatrix && !scalePartial) {
            for (double& v : V) v *= vScale;                                             // ADJ:4650
            changed = true;
        }
    }
    if (geographic && !reduced) {
        for (size_t r = m; r < idx; ++r) snprintf(bmsBinaryRecords_[r].coordType, sizeof(bmsBinaryRecords_[r].coordType), "%s", "XYZ");
    }
    if (changed) {
        // SetGPSVarianceMatrix (ADJ:4282): the scaled / propagated variances replace the ones held in memory, so that the
        // statistics (sigma zero, Pelzer reliability, ...) see what the adjustment used
        size_t r = m;
        for (UINT32 j = 0; j < k; ++j) {
            const UINT32 r0 = 3 * j;
            measurement_t& mx = bmsBinaryRecords_[r];
            measurement_t& my = bmsBinaryRecords_[r + 1];
            measurement_t& mz = bmsBinaryRecords_[r + 2];
            const UINT32 ncov = (type == 'G') ? 0 : mx.vectorCount2;
            mx.term2 = V[(size_t)r0 * nc + r0];
            my.term2 = V[(size_t)(r0 + 1) * nc + r0];
            my.term3 = V[(size_t)(r0 + 1) * nc + r0 + 1];
            mz.term2 = V[(size_t)(r0 + 2) * nc + r0];
            mz.term3 = V[(size_t)(r0 + 2) * nc + r0 + 1];
            mz.term4 = V[(size_t)(r0 + 2) * nc + r0 + 2];
            r += 3;
            for (UINT32 c = 0; c < ncov; ++c) {
                const UINT32 c0 = 3 * (j + 1 + c);
                for (UINT32 e = 0; e < 3; ++e) {
                    measurement_t& cv = bmsBinaryRecords_[r + e];
                    cv.term1 = V[(size_t)c0 * nc + r0 + e];
                    cv.term2 = V[(size_t)(c0 + 1) * nc + r0 + e];
                    cv.term3 = V[(size_t)(c0 + 2) * nc + r0 + e];
                }
                r += 3;
            }
        }
    }
    B.vcv.insert(B.vcv.end(), V.begin(), V.end());
    B.cluster_off.push_back((UINT32)B.stn1.size());
}

// One terrestrial measurement (types A B C E H K L M R S V Z): station indices, variance, instrument / target height,
// and the one-time reductions the reference applies when the matrices are first built (InitialiseMeasurement ADJ:3913;
// deflection of the vertical and geoid separation in the type's UpdateDesignNormalMeasMatrices_*, see terrestrial.h)
void dna_adjust::ParseTerrestrialMeasurement(UINT32 block, UINT32 m, block_t& B, const std::vector<double>& xyz) {
    measurement_t& rec = bmsBinaryRecords_[m];
    const char type = rec.measType;
    if (rec.measStart != 0) SignalExceptionAdjustment("PrepareAdjustment(): malformed terrestrial measurement record.", block);
    if (!(rec.term2 > 0.0)) SignalExceptionAdjustment("PrepareAdjustment(): a measurement with a non-positive variance.", block);
    const int nst = dnagpu::tm::station_count(type);
    const UINT32 gl[3] = {rec.station1, rec.station2, rec.station3};
    UINT32 loc[3] = {0, 0, 0};
    dnagpu::tm::StationGeo g[3];
    const double* X[3];
    static const double zero3[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < 3; ++q) {
        X[q] = zero3;
        g[q] = dnagpu::tm::StationGeo{0, 0, 0, 0, 0, 0};
        if (q >= nst) continue;
        loc[q] = LocalIndex(block, gl[q]);
        const station_t& st = bstBinaryRecords_.at(gl[q]);
        g[q] = dnagpu::tm::StationGeo{st.currentLatitude, st.currentLongitude, st.currentHeight, (double)st.geoidSep, st.verticalDef, st.meridianDef};
        X[q] = &xyz[3 * (size_t)loc[q]];
    }
    if (bms_meta_.reduced)
        rec.term1 = rec.preAdjMeas;   // a file written by an earlier adjustment: start again from the supplied value (ADJ:3922-3924)
    else
        rec.preAdjMeas = rec.term1;
    double value = rec.term1;
    rec.preAdjCorr = dnagpu::tm::reduce(type, &value, X[0], X[1], X[2], g[0], g[1], g[2], rec.term3, rec.term4);
    rec.term1 = value;
    B.t_type.push_back(type);
    for (int q = 0; q < 3; ++q) B.t_stn.push_back(loc[q]);
    B.t_rec.push_back(m);
    B.t_val.push_back(rec.term1);
    B.t_pre.push_back(rec.preAdjMeas);
    B.t_var.push_back(rec.term2);
    B.t_ih.push_back(rec.term3);
    B.t_th.push_back(rec.term4);
}

// One direction set (type D): the record of the reference direction followed by vectorCount1 - 1 direction records
// (CDnaDirectionSet::WriteBinaryMsr, dnadirectionset.cpp:430).  UpdateDesignNormalMeasMatrices_D (ADJ:5082): the angles between
// consecutive non-ignored directions are 'A' measurements (instrument, earlier target, later target; the heights of the earlier
// direction's record); LoadVarianceMatrix_D (ADJ:4059): the differences of independent directions have the tridiagonal variance
// matrix  V_aa = s_a^2 + s_a+1^2,  V_a,a+1 = -s_a+1^2, whose inverse weights the whole set.  The derived angle, its variance and
// covariance live in the later direction's record (scale1, scale2, scale3; preAdjMeas = the angle before the deflection correction).
UINT32 dna_adjust::ParseDirectionSet(UINT32 block, UINT32 m, block_t& B, const std::vector<double>& xyz) {
    measurement_t& ro = bmsBinaryRecords_[m];
    const UINT32 total = ro.vectorCount1;
    if (ro.measStart != 0 || total < 2 || (size_t)m + total > bmsBinaryRecords_.size())
        SignalExceptionAdjustment("PrepareAdjustment(): malformed direction set.", block);
    std::vector<UINT32> recs(1, m);
    for (UINT32 j = 1; j < total; ++j) {
        const measurement_t& d = bmsBinaryRecords_[m + j];
        if (d.measType != 'D') SignalExceptionAdjustment("PrepareAdjustment(): malformed direction set.", block);
        if (!d.ignore) recs.push_back(m + j);
    }
    const UINT32 k = (UINT32)recs.size() - 1;
    if (k < 1 || ro.vectorCount2 != k + 1) SignalExceptionAdjustment("PrepareAdjustment(): a direction set without a second direction.", block);
    B.dset_first.push_back((UINT32)B.t_type.size());
    B.dset_size.push_back(k);
    std::vector<double> V((size_t)k * k, 0.0);
    auto geo_of = [&](UINT32 g) {
        const station_t& st = bstBinaryRecords_.at(g);
        return dnagpu::tm::StationGeo{st.currentLatitude, st.currentLongitude, st.currentHeight, (double)st.geoidSep, st.verticalDef, st.meridianDef};
    };
    const UINT32 inst = LocalIndex(block, ro.station1);
    for (UINT32 a = 0; a < k; ++a) {
        const measurement_t& ra = bmsBinaryRecords_[recs[a]];
        measurement_t& rb = bmsBinaryRecords_[recs[a + 1]];
        if (!(ra.term2 > 0.0) || !(rb.term2 > 0.0)) SignalExceptionAdjustment("PrepareAdjustment(): a measurement with a non-positive variance.", block);
        const UINT32 l2 = LocalIndex(block, ra.station2), l3 = LocalIndex(block, rb.station2);
        double angle;
        if (bms_meta_.reduced) {
            angle = rb.preAdjMeas;                                   // ADJ:5133-5136, variances as stored (GetDirectionsVarianceMatrix)
            V[a + (size_t)a * k] = rb.scale2;
            if (a + 1 < k) V[a + (size_t)(a + 1) * k] = V[(a + 1) + (size_t)a * k] = rb.scale3;
        } else {
            angle = rb.term1 - ra.term1;                             // ADJ:5140-5144
            if (angle < 0) angle += dnagpu::tm::TWO_PI;
            if (angle > dnagpu::tm::TWO_PI) angle -= dnagpu::tm::TWO_PI;
            V[a + (size_t)a * k] = ra.term2 + rb.term2;
            if (a + 1 < k) V[a + (size_t)(a + 1) * k] = V[(a + 1) + (size_t)a * k] = -rb.term2;
            rb.scale2 = V[a + (size_t)a * k];                        // SetDirectionsVarianceMatrix
            rb.scale3 = a + 1 < k ? -rb.term2 : 0.0;
        }
        rb.preAdjMeas = angle;
        double value = angle;
        rb.preAdjCorr = dnagpu::tm::reduce('A', &value, &xyz[3 * (size_t)inst], &xyz[3 * (size_t)l2], &xyz[3 * (size_t)l3], geo_of(ro.station1),
                                           geo_of(ra.station2), geo_of(rb.station2), ra.term3, ra.term4);
        rb.scale1 = value;
        B.t_type.push_back('D');
        B.t_stn.push_back(inst);
        B.t_stn.push_back(l2);
        B.t_stn.push_back(l3);
        B.t_rec.push_back(recs[a + 1]);
        B.t_val.push_back(value);
        B.t_pre.push_back(angle);
        B.t_var.push_back(rb.scale2);
        B.t_ih.push_back(ra.term3);
        B.t_th.push_back(ra.term4);
    }
    // FormInverseVarianceMatrix: Cholesky inverse of the k x k matrix
    std::vector<double> L(V);
    for (UINT32 j = 0; j < k; ++j) {
        double d = L[j + (size_t)j * k];
        for (UINT32 q = 0; q < j; ++q) d -= L[j + (size_t)q * k] * L[j + (size_t)q * k];
        if (!(d > 0.0)) SignalExceptionAdjustment("Matrix inversion failed, the matrix is singular.", block);
        d = std::sqrt(d);
        L[j + (size_t)j * k] = d;
        for (UINT32 i = j + 1; i < k; ++i) {
            double s = L[i + (size_t)j * k];
            for (UINT32 q = 0; q < j; ++q) s -= L[i + (size_t)q * k] * L[j + (size_t)q * k];
            L[i + (size_t)j * k] = s / d;
        }
    }
    std::vector<double> Li((size_t)k * k, 0.0);       // L^-1, lower
    for (UINT32 j = 0; j < k; ++j) {
        Li[j + (size_t)j * k] = 1.0 / L[j + (size_t)j * k];
        for (UINT32 i = j + 1; i < k; ++i) {
            double s = 0.0;
            for (UINT32 q = j; q < i; ++q) s -= L[i + (size_t)q * k] * Li[q + (size_t)j * k];
            Li[i + (size_t)j * k] = s / L[i + (size_t)i * k];
        }
    }
    for (UINT32 i = 0; i < k; ++i)
        for (UINT32 j = 0; j < k; ++j) {
            double s = 0.0;
            for (UINT32 q = std::max(i, j); q < k; ++q) s += Li[q + (size_t)i * k] * Li[q + (size_t)j * k];
            B.dset_w.push_back(s);          // (symmetric: row / column order does not matter)
        }
    return k;
}

// PrepareAdjustmentBlock (ADJ:2873) for every block: host lists + device upload
void dna_adjust::PrepareBlocks() {
    const bool phased = projectSettings_.a.adjust_mode != SimultaneousMode;
    blocks_.assign(blockCount_, block_t());
    containsNonGPS_ = false;
    initial_xyz_.assign(blockCount_, {});
    max_unknowns_ = 0;
    max_junction_ = 0;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        max_unknowns_ = std::max<UINT32>(max_unknowns_, (UINT32)v_parameterStationList_[b].size() * 3);
        max_junction_ = std::max<UINT32>(max_junction_, (UINT32)v_JSL_[b].size() * 3);
    }
    int rc = plan_only_ ? DNAGPU_OK : dnagpu_create(projectSettings_.a.device, &ctx_);
    if (rc != DNAGPU_OK) {
        ctx_ = nullptr;
        SignalExceptionAdjustment("PrepareAdjustment(): no MI355X device available (the adjustment has no CPU path).", 0);
    }
    // several GPUs, one block (simultaneous adjustment): every GPU holds the block, the inverse itself is distributed -- large
    // launches split by tile columns, the parts exchanged over the communicator (dnagpu_set_inverse_exchange)
    if (!plan_only_ && comm_ && comm_->world() > 1 && projectSettings_.a.adjust_mode == SimultaneousMode)
        Check(dnagpu_set_inverse_exchange(ctx_, comm_->rank(), comm_->world(), &dna_adjust::ExchangeTrampoline, this), 0, "PrepareAdjustment()");
    {
        // a chain costs three matrices of the largest block's order (work matrix, X, W): no more chains than blocks, and
        // no more than half of the free HBM for all of them together
        size_t free_b = 0, total_b = 0;
        MemInfo(&free_b, &total_b);
        const double per_chain = 3.0 * ((double)max_unknowns_ + 256.0) * ((double)max_unknowns_ + 256.0) * 8.0;
        while (mt_chains_ > 1 && ((UINT32)mt_chains_ > blockCount_ || per_chain * mt_chains_ > 0.5 * (double)free_b)) --mt_chains_;
        if (mt_chains_ < 2 && blockCount_ > 1 && 2.0 * per_chain <= 0.5 * (double)free_b) mt_chains_ = 2;   // the two junction chains
    }
    const int chains = NumChains();
    for (int c = 0; c < chains; ++c) NewMatrix(max_unknowns_, &work_[c], 0, "PrepareAdjustment(): work matrix");

    for (UINT32 b = 0; b < blockCount_; ++b) {
        currentBlock_ = b;
        block_t& B = blocks_[b];
        const std::vector<UINT32>& plist = v_parameterStationList_[b];
        const UINT32 ns = (UINT32)plist.size();
        // PopulateEstimatedStationMatrix (ADJ:632)
        std::vector<double> xyz(3 * (size_t)ns);
        for (UINT32 p = 0; p < ns; ++p) {
            const station_t& st = bstBinaryRecords_[plist[p]];
            geodesy::GeoToCart(st.currentLatitude, st.currentLongitude, st.currentHeight, &xyz[3 * p], &xyz[3 * p + 1], &xyz[3 * p + 2]);
        }
        initial_xyz_[b] = xyz;
        // measurements of the block (CML order): 'G' baselines, 'X' baseline clusters, 'Y' point clusters
        const std::vector<UINT32>& cml = v_CML_[b];
        B.cluster_off.assign(1, 0);
        for (UINT32 m : cml) {
            if ((size_t)m >= bmsBinaryRecords_.size()) SignalExceptionAdjustment("PrepareAdjustment(): measurement index out of range.", b);
            if (bmsBinaryRecords_[m].ignore) continue;   // InitialiseandValidateMsrPointer
            const UINT32 pos = (UINT32)(B.c_pos.size() + B.t_pos.size());
            if (bmsBinaryRecords_[m].measType == 'D') {
                const UINT32 k = ParseDirectionSet(b, m, B, xyz);
                B.t_pos.insert(B.t_pos.end(), k, pos);     // one measurement, k design rows
                containsNonGPS_ = true;
            } else if (dnagpu::tm::is_terrestrial(bmsBinaryRecords_[m].measType)) {
                ParseTerrestrialMeasurement(b, m, B, xyz);
                B.t_pos.push_back(pos);
                containsNonGPS_ = true;
            } else {
                ParseGnssMeasurement(b, m, B);
                B.c_pos.push_back(pos);
            }
        }
        // constraint lists (ADJ:1884-2037)
        for (UINT32 p = 0; p < ns; ++p) {
            double w9[9];
            FormConstraintStationVarianceMatrix(plist[p], w9);
            const stn_appear& a = v_paramStnAppearance_[b][p];
            auto push = [&](constraint_list& l) {
                l.stn.push_back(p);
                l.w9.insert(l.w9.end(), w9, w9 + 9);
            };
            if (!phased) {
                push(B.con_sim);
                continue;
            }
            if (a.first_appearance_fwd) push(B.con_fwd);
            if (a.first_appearance_rev) push(B.con_rev);
            if (!a.first_appearance_fwd) push(B.con_cmb);
        }
        // junction index lists
        if (phased) {
            for (UINT32 s : v_JSL_[b]) B.jsl_here.push_back(LocalIndex(b, s));
            if (!v_blockMeta_[b]._blockLast && !v_blockMeta_[b]._blockIsolated)
                for (UINT32 s : v_JSL_[b]) B.jsl_in_next.push_back(LocalIndex(b + 1, s));   // JSL(b) must live in block b+1 (ADJ:1072)
            if (!v_blockMeta_[b]._blockFirst && !v_blockMeta_[b]._blockIsolated)
                for (UINT32 s : v_JSL_[b - 1]) B.jslprev_here.push_back(LocalIndex(b, s));
        }
        // device
        if (plan_only_) {
            // (stations: original, rigorous, estimates / rhs / corrections per chain; measurements: stations, observation, weights, b and W b
            //  per chain, pair lists -- ~ what dnagpu_block_create + dnagpu_block_set_clusters allocate)
            plan_bytes_ += (double)ns * 24.0 * (2.0 + 3.0 * chains) + (double)B.stn1.size() * (8.0 + 24.0 + 72.0 + 48.0 * chains + 60.0);
            continue;
        }
        Check(dnagpu_block_create(ctx_, b, ns, (UINT32)B.stn1.size()), b, "PrepareAdjustment(): block allocation");
        Check(dnagpu_block_set_stations(ctx_, b, xyz.data()), b, "PrepareAdjustment(): stations");
        {
            // station records the terrestrial measurement models read (current geodetic position, N, deflections)
            std::vector<double> llh(3 * (size_t)ns), geoid(ns), defl(2 * (size_t)ns);
            for (UINT32 p = 0; p < ns; ++p) {
                const station_t& st = bstBinaryRecords_[plist[p]];
                llh[3 * p] = st.currentLatitude;
                llh[3 * p + 1] = st.currentLongitude;
                llh[3 * p + 2] = st.currentHeight;
                geoid[p] = st.geoidSep;
                defl[2 * p] = st.verticalDef;
                defl[2 * p + 1] = st.meridianDef;
            }
            Check(dnagpu_block_set_station_geo(ctx_, b, llh.data(), geoid.data(), defl.data()), b, "PrepareAdjustment(): station records");
        }
        Check(dnagpu_block_set_terrestrial(ctx_, b, (UINT32)B.t_type.size(), B.t_type.data(), B.t_stn.data(), B.t_val.data(), B.t_pre.data(),
                                           B.t_var.data(), B.t_ih.data(), B.t_th.data(), B.t_pos.data(), B.c_pos.data(), (UINT32)B.c_pos.size()),
              b, "PrepareAdjustment(): terrestrial measurements");
        if (!B.dset_first.empty())
            Check(dnagpu_block_set_direction_sets(ctx_, b, (UINT32)B.dset_first.size(), B.dset_first.data(), B.dset_size.data(), B.dset_w.data()), b,
                  "PrepareAdjustment(): direction sets");
        Check(dnagpu_block_set_clusters(ctx_, b, B.stn1.data(), B.stn2.data(), B.obs.data(), (UINT32)B.cluster_off.size() - 1,
                                        B.cluster_off.data(), B.vcv.data()),
              b, "PrepareAdjustment(): measurements");
        for (int c = 0; c < chains; ++c) Check(dnagpu_block_compute_b(ctx_, c, b), b, "PrepareAdjustment(): meas-minus-computed");
        // junction matrices: AllocateChainData (once it is known which blocks this rank works on);
        // v_rigorousVariances_ is allocated when the block is finalised (only on the rank / chain that owns it)
    }
    ComputeBlockOwners(CondensedWanted() && !ReuseInverses());
    DecideStaging();
    PrepareCondensedBlocks();     // (+ ownership under the reference's schedule, the two-level plan, junction matrices / condensed blocks)
    ReserveBuffers();
    if (!plan_only_) Check(dnagpu_sync(ctx_), 0, "PrepareAdjustment()");
}

// the reference's files and the lists derived from them (LoadNetworkFiles ... CreateStnAppearanceList): host only
void dna_adjust::LoadAndListNetwork() {
    LoadNetworkFiles();
    switch (projectSettings_.a.adjust_mode) {
        case SimultaneousMode: BuildSimultaneousLists(); break;
        case PhasedMode:
        case Phased_Block_1Mode: {
            iostreams::seg_data_t seg;
            iostreams::read_seg(projectSettings_.a.seg_file, seg, &bmsBinaryRecords_);
            blockCount_ = seg.blockCount;
            v_ISL_ = seg.ISL;
            v_JSL_ = seg.JSL;
            v_CML_ = seg.CML;
            v_ContiguousNetList_ = seg.ContiguousNetList;
            v_measurementCount_ = seg.measurementCount;
            v_unknownsCount_ = seg.unknownsCount;
            v_parameterStationCount_ = seg.parameterStationCount;
            if (v_ISL_.size() != v_JSL_.size() || v_JSL_.size() != v_CML_.size())
                throw std::runtime_error(
                    "LoadPhasedBlocks(): An unrecoverable error was encountered when loading the phased adjustment blocks.");
            break;
        }
        default: throw std::runtime_error("AdjustNetwork(): Unknown adjustment type");
    }
    if (blockCount_ == 0) throw std::runtime_error("PrepareAdjustment(): the network has no blocks.");
    LoadSegmentationMetrics();
    CreateStnAppearanceList();
}

// ADJ:258-442
void dna_adjust::PrepareAdjustment(const project_settings& projectSettings) {
    if (projectSettings.a.devices.size() > 1 && !is_peer_) {
        PrepareMultiDevice(projectSettings);      // one instance + host thread per GPU; each comes back here as a rank
        return;
    }
    if (!in_collective_) peers_.clear();
    isPreparing_ = true;
    isAdjusting_ = true;
    isCombining_ = false;
    exceptionRaised_ = false;
    adjustStatus_ = ADJUST_SUCCESS;
    FreeDevice();
    projectSettings_ = projectSettings;
    mt_chains_ = DNAGPU_DEFAULT_CHAINS;
    profileTimings_ = getenv("DYNADJUST_PROFILE") != nullptr;
    profileUpdateNormalsNs_ = profileStageLoadNs_ = profileStageStoreNs_ = 0;
    stageCopiedBytes_ = stageWaitNs_ = 0;
    if (const char* e = getenv("DNAGPU_CHAINS")) mt_chains_ = std::max(2, std::min(DNAGPU_NUM_CHAINS, atoi(e)));
    if (const char* e = getenv("DNAGPU_FORCE_DISTRIBUTED")) force_distributed_ = atoi(e) != 0;
    if (comm_ && (comm_->world() != std::max(1, projectSettings_.a.dist_world) || comm_->rank() != projectSettings_.a.dist_rank)) {
        if (projectSettings_.a.dist_world <= 1) {        // an attached communicator names the rank
            projectSettings_.a.dist_world = comm_->world();
            projectSettings_.a.dist_rank = comm_->rank();
        } else {
            SignalExceptionAdjustment("PrepareAdjustment(): the attached communicator does not match a.dist_rank / a.dist_world.", 0);
        }
    }
    if (!comm_ && (projectSettings_.a.dist_world > 1 || force_distributed_)) {
        // one process per GPU and nobody attached a communicator: RCCL, the unique id from rank 0 over TCP (MASTER_ADDR / MASTER_PORT)
        try {
            const int world = std::max(1, projectSettings_.a.dist_world);
            std::string transport = projectSettings_.a.dist_transport;
            if (transport.empty() && getenv("DNAGPU_DIST_TRANSPORT")) transport = getenv("DNAGPU_DIST_TRANSPORT");
            if (transport == "shared") {
                // processes that share a GPU (or have no fabric between theirs): host-staged over TCP (dist_comm_shared.cpp)
                comm_ = shared_comm_create(projectSettings_.a.dist_rank, world, projectSettings_.a.device);
            } else {
                unsigned char id[DIST_UNIQUE_ID_BYTES] = {0};
                if (projectSettings_.a.dist_rank == 0) rccl_unique_id(id);
                tcp_share_unique_id(projectSettings_.a.dist_rank, world, id);
                comm_ = rccl_comm_create(projectSettings_.a.dist_rank, world, id, projectSettings_.a.device);
            }
        } catch (const std::exception& e) {
            SignalExceptionAdjustment(std::string("PrepareAdjustment(): cannot join the other GPUs' processes. Details: ") + e.what(), 0);
        }
    }
    staged_ = projectSettings_.a.stage != 0;   // staged: rigorous variances in page-locked host memory (PrepareCondensedBlocks may switch it on)
    // InitialiseAdjustment (ADJ:232-245)
    var_C_ = projectSettings_.a.fixed_std_dev * projectSettings_.a.fixed_std_dev;
    var_F_ = projectSettings_.a.free_std_dev * projectSettings_.a.free_std_dev;
    currentBlock_ = 0;
    currentIteration_ = 0;
    try {
        LoadAndListNetwork();
        PrepareBlocks();
    } catch (const NetAdjustException&) {
        throw;
    } catch (const std::runtime_error& e) {
        std::stringstream ss;
        ss << "PrepareAdjustment(): Process terminated while preparing the " << std::endl
           << "  adjustment matrices. Details: " << std::endl
           << "  " << e.what() << std::endl;
        SignalExceptionAdjustment(ss.str(), currentBlock_);
    }
    degreesofFreedom_ = (int)measurementParams_ - (int)unknownParams_;
    isPreparing_ = false;
}

void dna_adjust::AddConstraints(int chain, dnagpu_matrix* m, const constraint_list& c, int sign, UINT32 block) {
    if (c.stn.empty()) return;
    Check(dnagpu_add_diag3x3(ctx_, chain, m, c.stn.data(), c.w9.data(), c.stn.size(), sign), block,
          "AddConstraintStationstoNormals()");
}

// SolveTry (ADJ:6569) / Solve (ADJ:6586): inverse + corrections; rhs already on the device
void dna_adjust::SolveTry(int chain, UINT32 block, dnagpu_matrix* m) {
    Check(dnagpu_invert(ctx_, chain, m, projectSettings_.a.scale_normals_to_unity ? 1 : 0), block, "Solve()");
    double n = 3.0 * (double)v_parameterStationList_[block].size();
    {
        std::lock_guard<std::mutex> lk(corr_mutex_);
        solve_flops_ += n * n * n;
        CountFlops(n * n * n, 0);
        solve_count_++;
    }
    Check(dnagpu_solve_corrections(ctx_, chain, block, m), block, "Solve()");
}

// ADJ:2140
_ADJUST_STATUS_ dna_adjust::AdjustNetwork() {
    if (!peers_.empty() && !in_collective_) {
        OnEveryDevice([](dna_adjust& a) { a.AdjustNetwork(); });
        return adjustStatus_;
    }
    if (!ctx_) SignalExceptionAdjustment("AdjustNetwork(): PrepareAdjustment() has not been called.", 0);
    isAdjusting_ = true;
    adjustStatus_ = ADJUST_SUCCESS;
    iterationCorrections_.clear();
    solve_flops_ = 0.0;
    solve_count_ = 0;
    elimination_count_ = 0;
    condense_count_ = 0;
    completion_count_ = 0;
    batched_members_ = 0;
    batched_flops_ = 0.0;
    stageCopiedBytes_ = stageWaitNs_ = 0;
    transient_count_ = 0;
    algorithmic_flops_ = min_work_flops_ = 0.0;
    for (block_t& b : blocks_) {
        b.inverse_kept = b.inverse_pending = b.part_valid = b.rig_direct = b.var_deferred = false;
        b.factor_live = b.factor_reused = b.cfac_live[0] = b.cfac_live[1] = false;      // (a.reuse_factors: within one adjustment only)
        b.red_iter = 0;
    }
    lock_factored_ = false;
    factor_reuses_ = chain_reuses_ = 0;
    small_batch_steps_ = 0;
    osc_ready_ = false;              // corrPrev_ / stnOscCount_ / oscHistory_ start empty (ADJ:2419-2421, 2584-2586)
    oscHistory_.clear();
    const double t0 = now_ms();
    switch (projectSettings_.a.adjust_mode) {
        case SimultaneousMode: AdjustSimultaneous(); break;
        case PhasedMode: AdjustPhased(); break;
        case Phased_Block_1Mode: AdjustPhasedBlock1(); break;
        default: SignalExceptionAdjustment("AdjustNetwork(): Unknown adjustment type", 0);
    }
    Check(dnagpu_sync(ctx_), 0, "AdjustNetwork()");
    adjust_ms_ = now_ms() - t0;
    if (getenv("DNAGPU_PHASE_TIMES")) fprintf(stderr, "[phase] AdjustNetwork             %6.1f ms\n", adjust_ms_);
    PrintPerformanceProfile();
    return adjustStatus_;
}

// ADJ:2413-2511 (GNSS only: the inverse is formed once, ADJ:2457)
void dna_adjust::AdjustSimultaneous() {
    const int c = 0;
    block_t& B = blocks_[0];
    dnagpu_matrix* W = work_[0];
    currentIteration_ = 0;
    for (UINT32 i = 0; i < projectSettings_.a.max_iterations; ++i) {
        if (IsCancelled()) break;
        const double it_t0 = now_ms();
        ++currentIteration_;
        currentBlock_ = 0;
        // the inverse is only formed again if the network has non-GPS measurements, whose design follows the estimates
        // (SolveTry(CurrentIteration() < 2 || ContainsNonGPS()), ADJ:2457; UpdateNormals in UpdateAdjustment, ADJ:582-590)
        const bool invert = currentIteration_ < 2 || containsNonGPS_;
        if (invert) {
            Check(dnagpu_form_normals(ctx_, c, 0, W), 0, "UpdateNormals()");
            AddConstraints(c, W, B.con_sim, +1, 0);
        }
        Check(dnagpu_form_rhs(ctx_, c, 0), 0, "Solve()");
        if (invert)
            SolveTry(c, 0, W);
        else
            Check(dnagpu_solve_corrections(ctx_, c, 0, W), 0, "Solve()");
        double mv = 0.0;
        UINT32 row = 0;
        Check(dnagpu_update_estimates(ctx_, c, 0, &mv, &row), 0, "AdjustSimultaneous()");
        maxCorr_ = mv;
        B.corr_chain = c;
        UpdateIterationDiagnostics();                   // (ADJ:2468)
        iterationCorrections_.push_back(maxCorr_);
        NoteIterationDone(it_t0);                       // the progress thread's message of this iteration (ADJ:2471-2472)
        bool iterate = !IsCancelled() && std::fabs(maxCorr_) > projectSettings_.a.iteration_threshold;
        if (!iterate) break;
        UpdateAdjustment(true);
    }
    // v_rigorousVariances_[0] = v_normals_[0] (ADJ:2536); rigorous stations = estimates
    Check(dnagpu_block_copy_stations(ctx_, c, 0, 2, 1), 0, "ValidateandFinaliseAdjustment()");
    ValidateandFinaliseAdjustment();
}

// ADJ:2513-2549
void dna_adjust::ValidateandFinaliseAdjustment() {
    isAdjusting_ = false;
    if (adjustStatus_ > ADJUST_TEST_FAILED) return;
    if (IsCancelled()) {
        adjustStatus_ = ADJUST_CANCELLED;
        return;
    }
    if (currentIteration_ == projectSettings_.a.max_iterations && std::fabs(maxCorr_) > projectSettings_.a.iteration_threshold)
        adjustStatus_ = ADJUST_MAX_ITERATIONS_EXCEEDED;
}

// ADJ:473-627: new meas-minus-computed from the latest estimates; the GNSS normals do not change
void dna_adjust::UpdateAdjustment(bool iterate) {
    isPreparing_ = true;
    const bool phased = projectSettings_.a.adjust_mode != SimultaneousMode;
    const int chains = NumChains();
    // across GPUs with the condensed schedule a rank forms and solves its own blocks only (the chains run on condensed blocks,
    // whose right-hand sides arrive with them)
    const bool own_only = Distributed() && CondensedSchedule() && DistWorld() > 1;
    // every chain has its own estimates and meas-minus-computed of every block (any chain may take any block step): each chain's host thread
    // renews its own (a dnasegment-default cut has hundreds of blocks: eight launches per block from ONE thread were 19 ms per iteration)
    auto per_chain = [&](int c) {
        for (UINT32 b = 0; b < blockCount_; ++b) {
            if (IsCancelled()) break;
            if (own_only && !OwnsBlock(b)) continue;
            // a last block: estimated = original = rigorous (ADJ:516-517)
            if (phased && v_blockMeta_[b]._blockLast && c == 0) Check(dnagpu_block_copy_stations(ctx_, 0, b, 0, 2), b, "UpdateAdjustment()");
            // every chain restarts from the rigorous estimates of the iteration just finished (multi-thread mode:
            // v_estimatedStationsR_ = v_rigorousStations_ ADJ:569; v_estimatedStations_ = v_estimatedStationsR_ ADJ:3799)
            if (phased) Check(dnagpu_block_copy_stations(ctx_, c, b, 1, 2), b, "UpdateAdjustment()");
            // non-GPS networks: the station records follow the estimates, the design of the next iteration is formed
            // in their local frames (UpdateGeographicCoords[Phased], ADJ:496-531, ADJ:541-545)
            if (containsNonGPS_) Check(dnagpu_block_update_geodetic(ctx_, c, b), b, "UpdateGeographicCoords()");
            Check(dnagpu_block_compute_b(ctx_, c, b), b, "UpdateAdjustment()");
        }
    };
    if (phased && !own_only && !IsCancelled() && EnsureBlockTable()) {
        // GNSS-only: the same for every block and chain in one launch (666 blocks x 4 chains x 2 enqueues were 7 ms per iteration)
        Check(dnagpu_sync(ctx_), 0, "UpdateAdjustment()");         // (the rigorous estimates of every chain's blocks are final)
        Check(dnagpu_block_table_apply(ctx_, 0, block_table_, 1, chains), 0, "UpdateAdjustment()");
    } else if (chains > 1 && blockCount_ >= 32) {
        OnEveryChain(per_chain);
    } else {
        for (int c = 0; c < chains; ++c) per_chain(c);
    }
    // Everything above was enqueued chain by chain; what follows reads across chains -- the reverse thread's chain takes the last block's
    // originals that chain 0 has just set, any chain the estimates another one restored.  The chains meet here, once per iteration
    // (found in round 3: under host load the reverse pass of the reference's multi-thread schedule started from the last block's
    // previous originals now and then -- an 8 cm correction in iteration 2, healed by two further iterations, 1e-8 m left in the result).
    Check(dnagpu_sync(ctx_), 0, "UpdateAdjustment()");
    // no further iterations: the station records take the adjusted coordinates (ADJ:496-531, ADJ:541-545)
    if (!iterate && !IsCancelled()) UpdateGeographicCoords();
    isPreparing_ = false;
}

// AdjustPhasedBlock1 (ADJ:2675): one reverse pass (AdjustPhasedReverse, ADJ:3594) -- every block solved in isolation with the
// junctions carried from the blocks after it, so that the first block of the network comes out rigorous; the other blocks
// but the last keep the estimates and variances of their reverse solve (UpdateEstimatesFinal, ADJ:3667)
void dna_adjust::AdjustPhasedBlock1() {
    currentIteration_ = 1;
    maxCorr_ = 0.0;
    forward_ = false;
    double first_corr = 0.0;
    for (UINT32 kk = blockCount_; kk-- > 0;) {
        if (IsCancelled()) break;
        const UINT32 k = kk;
        currentBlock_ = k;
        if (v_blockMeta_[k]._blockIsolated) continue;          // PrepareAdjustmentReverse: nothing to do for a single block
        const double mv = PhasedReverseBlock(0, k);
        if (k == 0) first_corr = mv;
        // UpdateEstimatesFinal returns at once for the last block of a network in this mode (ADJ:3750-3753): its rigorous
        // coordinates and variances stay as PrepareAdjustment left them
        if (!v_blockMeta_[k]._blockLast) PhasedFinaliseBlock(0, k);
    }
    maxCorr_ = first_corr;                                      // "largest correction for block 1 only" (ADJ:2705)
    iterationCorrections_.push_back(maxCorr_);
    if (std::fabs(maxCorr_) > projectSettings_.a.iteration_threshold) adjustStatus_ = ADJUST_THRESHOLD_EXCEEDED;
    if (!IsCancelled()) UpdateGeographicCoords();
    ValidateandFinaliseAdjustment();
}

// ADJ:2579-2670
void dna_adjust::AdjustPhased() {
    if (Distributed()) {
        AdjustPhasedDistributed();
        return;
    }
    currentIteration_ = 0;
    const bool times = getenv("DNAGPU_PHASE_TIMES") != nullptr;
    for (UINT32 i = 0; i < projectSettings_.a.max_iterations; ++i) {
        if (IsCancelled()) break;
        maxCorr_ = 0.0;
        ++currentIteration_;
        const double it_t0 = now_ms();
        // staged mode: the previous iteration's copies to host memory overlap UpdateAdjustment and, with the condensed schedule,
        // the condensing and chain phases as well -- they only have to be home before the same buffers are written again
        if (!CondensedSchedule()) FinishStagedCopies();
        if (CondensedSchedule()) {
            AdjustPhasedCondensedIteration();
        } else if (projectSettings_.a.multi_thread && NumChains() >= 2) {
            AdjustPhasedMultiThreadIteration();
        } else {
            AdjustPhasedForward();
            if (IsCancelled()) break;
            AdjustPhasedReverseCombine();
        }
        if (IsCancelled()) break;
        const double td = now_ms();
        UpdateIterationDiagnostics();                   // (ADJ:2631)
        iterationCorrections_.push_back(maxCorr_);
        NoteIterationDone(it_t0);
        bool iterate = !IsCancelled() && std::fabs(maxCorr_) > projectSettings_.a.iteration_threshold;
        if (times) fprintf(stderr, "[phase] iteration %u diagnostics %6.1f ms\n", (unsigned)currentIteration_, now_ms() - td);
        if (!iterate) break;
        const double tu = now_ms();
        UpdateAdjustment(iterate);
        if (times) {
            Check(dnagpu_sync(ctx_), 0, "AdjustPhased()");
            fprintf(stderr, "[phase] iteration %u update      %6.1f ms\n", (unsigned)currentIteration_, now_ms() - tu);
        }
    }
    if (times) Check(dnagpu_sync(ctx_), 0, "AdjustPhased()");
    const double tv = now_ms();
    if (!IsCancelled()) FinishDeferredVariances();
    if (times) {
        Check(dnagpu_sync(ctx_), 0, "AdjustPhased()");
        fprintf(stderr, "[phase] variance matrices    %8.1f ms\n", now_ms() - tv);
    }
    FinishStagedCopies();
    const double tf = now_ms();
    ValidateandFinaliseAdjustment();
    if (times) fprintf(stderr, "[phase] validate and finalise %6.1f ms\n", now_ms() - tf);
}

// staged mode: the rigorous variance matrices of the iteration are on their way to host memory on the chains' copy streams
void dna_adjust::FinishStagedCopies() {
    if (!ctx_ || !Staged()) return;
    const auto t0 = std::chrono::steady_clock::now();
    Check(dnagpu_copies_sync(ctx_), 0, "SerialiseBlockToMappedFile()");
    stageWaitNs_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

void dna_adjust::GetBlockRigorousStations(UINT32 block, std::vector<double>& xyz) {
    if (!ctx_ || block >= blockCount_) throw std::runtime_error("GetBlockRigorousStations(): no such block");
    xyz.resize(3 * v_parameterStationList_[block].size());
    Check(dnagpu_block_get_stations(ctx_, 0, block, 2, xyz.data()), block, "GetBlockRigorousStations()");
}

void dna_adjust::GetBlockRigorousVariancesPacked(UINT32 block, std::vector<double>& packed) {
    if (!ctx_ || block >= blockCount_) throw std::runtime_error("GetBlockRigorousVariancesPacked(): no such block");
    if (!peers_.empty() && !OwnsBlock(block)) {     // another GPU of this process holds it
        DeviceInstance(BlockOwner(block))->GetBlockRigorousVariancesPacked(block, packed);
        return;
    }
    size_t n = 3 * v_parameterStationList_[block].size();
    packed.resize(n * (n + 1) / 2);
    if (projectSettings_.a.adjust_mode != SimultaneousMode && Staged() && blocks_[block].rig_host && blocks_[block].has_rigvar) {
        if (blocks_[block].rig_on_device)
            Check(dnagpu_copy(ctx_, packed.data(), blocks_[block].rig_host, packed.size() * sizeof(double)), block, "GetBlockRigorousVariancesPacked()");
        else
            memcpy(packed.data(), blocks_[block].rig_host, packed.size() * sizeof(double));
        return;
    }
    dnagpu_matrix* m = (projectSettings_.a.adjust_mode == SimultaneousMode) ? work_[0] : blocks_[block].rigvar;
    // (between the condensing step of an iteration and the end of the adjustment the matrix's storage holds the block's kept factor)
    if (!m || (projectSettings_.a.adjust_mode != SimultaneousMode && !blocks_[block].has_rigvar))
        throw std::runtime_error("GetBlockRigorousVariancesPacked(): this process holds no rigorous variances for the block");
    Check(dnagpu_matrix_download_packed(ctx_, 0, m, packed.data()), block, "GetBlockRigorousVariancesPacked()");
}

void dna_adjust::GetAdjustedCoordinates(std::vector<double>& xyz) {
    xyz.assign(3 * bstBinaryRecords_.size(), 0.0);
    for (size_t s = 0; s < bstBinaryRecords_.size(); ++s) {
        const station_t& st = bstBinaryRecords_[s];
        geodesy::GeoToCart(st.currentLatitude, st.currentLongitude, st.currentHeight, &xyz[3 * s], &xyz[3 * s + 1], &xyz[3 * s + 2]);
    }
    // every block holds rigorous values for all of its stations; a station shared by two blocks has the
    // same rigorous estimate in both, the block of first appearance is used (BuildUniqueBlockStationMap ADJ:1849)
    std::vector<double> bx;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        GetBlockRigorousStations(b, bx);
        for (size_t p = 0; p < v_parameterStationList_[b].size(); ++p)
            if (v_paramStnAppearance_[b][p].first_appearance_fwd)
                for (int c = 0; c < 3; ++c) xyz[3 * (size_t)v_parameterStationList_[b][p] + c] = bx[3 * p + c];
    }
}

// the a-priori coordinates of every block, once more in HBM: a reset is then one launch instead of six copies from the host and a launch
// per chain for every block (113 ms -> 3 ms for the 666 blocks of a default dnasegment cut)
void dna_adjust::EnsureInitialOnDevice() {
    if (initial_dev_) return;
    initial_off_.assign(blockCount_ + 1, 0);
    for (UINT32 b = 0; b < blockCount_; ++b) initial_off_[b + 1] = initial_off_[b] + initial_xyz_[b].size();
    std::vector<double> all(initial_off_[blockCount_]);
    for (UINT32 b = 0; b < blockCount_; ++b) std::copy(initial_xyz_[b].begin(), initial_xyz_[b].end(), all.begin() + initial_off_[b]);
    void* p = nullptr;
    Check(dnagpu_device_alloc(ctx_, std::max<size_t>(all.size(), 1) * sizeof(double), &p), 0, "ResetAdjustment()");
    initial_dev_ = (double*)p;
    Check(dnagpu_copy(ctx_, initial_dev_, all.data(), all.size() * sizeof(double)), 0, "ResetAdjustment()");
}

// GNSS-only phased networks: the blocks' coordinate bookkeeping between iterations (dnagpu_block_table_*) for all blocks in one launch
bool dna_adjust::EnsureBlockTable() {
    if (block_table_) return true;
    if (block_table_denied_ || containsNonGPS_ || projectSettings_.a.adjust_mode == SimultaneousMode || blockCount_ < 8) return false;
    EnsureInitialOnDevice();
    std::vector<UINT32> ids(blockCount_);
    std::vector<int> last(blockCount_);
    for (UINT32 b = 0; b < blockCount_; ++b) {
        ids[b] = b;
        last[b] = v_blockMeta_[b]._blockLast ? 1 : 0;
    }
    if (dnagpu_block_table_create(ctx_, blockCount_, ids.data(), last.data(), initial_dev_, initial_off_.data(), &block_table_) != DNAGPU_OK) {
        block_table_ = nullptr;
        block_table_denied_ = true;
        return false;
    }
    return true;
}

void dna_adjust::ResetAdjustment() {
    if (!peers_.empty() && !in_collective_) {
        OnEveryDevice([](dna_adjust& a) { a.ResetAdjustment(); });
        return;
    }
    if (!ctx_) SignalExceptionAdjustment("ResetAdjustment(): PrepareAdjustment() has not been called.", 0);
    const double t_reset = now_ms();
    exchange_ms_ = chain_ms_ = 0.0;
    const int chains = NumChains();
    EnsureInitialOnDevice();
    if (EnsureBlockTable()) {
        Check(dnagpu_block_table_apply(ctx_, 0, block_table_, 0, chains), 0, "ResetAdjustment()");
    } else {
        for (UINT32 b = 0; b < blockCount_; ++b)
            Check(dnagpu_block_reset_stations(ctx_, 0, b, initial_dev_ + initial_off_[b], blocks_[b].t_pos.empty() ? 1 : 0), b, "ResetAdjustment()");
    }
    if (containsNonGPS_) Check(dnagpu_chain_sync(ctx_, 0), 0, "ResetAdjustment()");
    for (UINT32 b = 0; b < blockCount_; ++b) {
        if (!blocks_[b].t_pos.empty())
            for (int c = 0; c < chains; ++c) Check(dnagpu_block_compute_b(ctx_, c, b), b, "ResetAdjustment()");
        blocks_[b].has_rigvar = false;
        blocks_[b].has_finv = blocks_[b].has_rinv = blocks_[b].has_cinv = false;
        blocks_[b].inverse_kept = blocks_[b].inverse_pending = blocks_[b].part_valid = blocks_[b].rig_direct = blocks_[b].var_deferred = false;
        blocks_[b].part_transient = false;
        blocks_[b].fac_packed = false;
        blocks_[b].factor_live = blocks_[b].factor_reused = blocks_[b].cfac_live[0] = blocks_[b].cfac_live[1] = false;
        blocks_[b].red_iter = 0;
    }
    lock_factored_ = false;
    Check(dnagpu_sync(ctx_), 0, "ResetAdjustment()");
    if (getenv("DNAGPU_PHASE_TIMES")) fprintf(stderr, "[phase] ResetAdjustment           %6.1f ms\n", now_ms() - t_reset);
    osc_ready_ = false;              // (a second adjustment on the handle must not compare with the first one's corrections)
    oscHistory_.clear();
    currentIteration_ = 0;
    maxCorr_ = 0.0;
    iterationCorrections_.clear();
    solve_flops_ = 0.0;
    solve_count_ = 0;
    elimination_count_ = 0;
    condense_count_ = 0;
    completion_count_ = 0;
    batched_members_ = 0;
    batched_flops_ = 0.0;
    algorithmic_flops_ = min_work_flops_ = 0.0;
    cancel_.store(false);
    cancel_agreed_ = false;
    adjustStatus_ = ADJUST_SUCCESS;
}

// ADJ:6802-6841: UpdateAdjustment(false) + ComputeStatistics()
void dna_adjust::GenerateStatistics() {
    if (!peers_.empty() && !in_collective_) {
        OnEveryDevice([](dna_adjust& a) { a.GenerateStatistics(); });
        return;
    }
    if (!ctx_) SignalExceptionAdjustment("GenerateStatistics(): PrepareAdjustment() has not been called.", 0);
    if (Distributed() && projectSettings_.a.adjust_mode == PhasedMode) {
        GenerateStatisticsDistributed();
        isAdjustmentQuestionable_ = adjustStatus_ != ADJUST_SUCCESS || sigmaZero_ > 10.0 * chiSquaredUpperLimit_ ||
                                    std::fabs(maxCorr_) > projectSettings_.a.iteration_threshold;
        return;
    }
    // meas-minus-computed from the final estimates; the inverses are kept (ADJ:549-557)
    UpdateAdjustment(false);
    ComputeStatistics();
    isAdjustmentQuestionable_ = adjustStatus_ != ADJUST_SUCCESS || sigmaZero_ > 10.0 * chiSquaredUpperLimit_ ||
                                std::fabs(maxCorr_) > projectSettings_.a.iteration_threshold;
}

// Exchange of the per-record results between processes: 9 doubles per record -- touched (1 / 0), measAdj, measCorr, measAdjPrec,
// residualPrec, NStat, PelzerRel, preAdjCorr, term1 -- zero for records this process did not compute, so that a sum over the
// processes holds every record once
void dna_adjust::GetRecordStatistics(double* out9) const {
    for (size_t i = 0; i < bmsBinaryRecords_.size(); ++i) {
        double* o = out9 + 9 * i;
        const measurement_t& r = bmsBinaryRecords_[i];
        const bool t = i < record_touched_.size() && record_touched_[i];
        const double v[9] = {1.0, r.measAdj, r.measCorr, r.measAdjPrec, r.residualPrec, r.NStat, r.PelzerRel, r.preAdjCorr, r.term1};
        for (int q = 0; q < 9; ++q) o[q] = t ? v[q] : 0.0;
    }
}
void dna_adjust::SetRecordStatistics(const double* in9) {
    record_touched_.assign(bmsBinaryRecords_.size(), 0);
    for (size_t i = 0; i < bmsBinaryRecords_.size(); ++i) {
        const double* v = in9 + 9 * i;
        if (v[0] < 0.5) continue;
        measurement_t& r = bmsBinaryRecords_[i];
        r.measAdj = v[1]; r.measCorr = v[2]; r.measAdjPrec = v[3]; r.residualPrec = v[4];
        r.NStat = v[5]; r.PelzerRel = v[6]; r.preAdjCorr = v[7]; r.term1 = v[8];
        record_touched_[i] = 1;
    }
}

// ADJ:7116-7147 for GNSS measurements.  The per-vector gathers from the rigorous variances (still resident in HBM)
// and the W.b products run on the device (dnagpu_block_msr_statistics); the O(measurements) scalar bookkeeping of
// UpdateMsrRecord (ADJ:8187) happens here on the records held in memory.
void dna_adjust::ComputeStatistics() {
    StatisticsBegin();
    for (UINT32 b = 0; b < blockCount_; ++b) StatisticsBlock(b);
    StatisticsFinish();
}

// The three parts of ComputeStatistics, separately callable so that one process per GPU can each do the blocks whose rigorous
// variances it holds (dynadjust_amd/parallel.py, distributed_statistics): per-run initialisation ...
void dna_adjust::StatisticsBegin() {
    // critical value of the normal distribution for the outlier flag (InitialiseAdjustment, ADJ:203-206)
    double conf = projectSettings_.a.confidence_interval * 0.01;
    conf += (1.0 - conf) / 2.0;
    criticalValue_ = stat::normal_quantile(conf);
    potentialOutlierCount_ = 0;
    chiSquared_ = 0.0;
    record_touched_.assign(bmsBinaryRecords_.size(), 0);
}

// ... one block: precisions of the adjusted measurements from its rigorous variances, the per-record statistics, its chi-square terms ...
void dna_adjust::StatisticsBlock(UINT32 b) {
    const bool phased = projectSettings_.a.adjust_mode != SimultaneousMode;
    double chiSquared = 0.0;
    std::vector<double> prec6, chi, bvec;
    {
        block_t& B = blocks_[b];
        const size_t nv = B.stn1.size();
        dnagpu_matrix* var = phased ? (B.has_rigvar ? B.rigvar : nullptr) : work_[0];
        if (phased && Staged() && B.has_rigvar && B.rig_host) {
            // staged: the block's rigorous variances come back from host memory into the work matrix (lower triangle: all the
            // statistics kernels read)
            var = work_[0];
            const auto t0 = std::chrono::steady_clock::now();
            if (B.rig_on_device)
                Check(dnagpu_matrix_unpack_device(ctx_, 0, var, B.rig_host, (UINT32)v_parameterStationList_[b].size() * 3), b, "ComputePrecisionAdjMsrs()");
            else
                Check(dnagpu_matrix_upload_packed(ctx_, 0, var, B.rig_host, (UINT32)v_parameterStationList_[b].size() * 3), b, "ComputePrecisionAdjMsrs()");
            profileStageLoadNs_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        }
        if (!var) SignalExceptionAdjustment("ComputePrecisionAdjMsrs(): this process holds no rigorous variances for the block.", b);
        prec6.assign(6 * nv + 1, 0.0);
        chi.assign(nv + 1, 0.0);
        bvec.assign(3 * nv + 1, 0.0);
        Check(dnagpu_block_msr_statistics(ctx_, 0, b, var, prec6.data(), chi.data()), b, "ComputePrecisionAdjMsrs()");
        Check(dnagpu_block_get_b(ctx_, 0, b, bvec.data()), b, "UpdateMsrRecords()");
        B.prec_adj_msrs.assign(prec6.begin(), prec6.begin() + 6 * nv);            // v_precAdjMsrsFull_ (ADJ:7792)
        // UpdateMsrRecords (ADJ:8083) -> UpdateMsrRecords_GXY (ADJ:8152)
        size_t v = 0;
        for (UINT32 m : v_CML_[b]) {
            if (bmsBinaryRecords_[m].ignore || bmsBinaryRecords_[m].measStart != 0) continue;
            const char type = bmsBinaryRecords_[m].measType;
            if (dnagpu::tm::is_terrestrial(type)) continue;   // below
            const UINT32 k = (type == 'G') ? 1 : bmsBinaryRecords_[m].vectorCount1;
            size_t r = m;
            for (UINT32 j = 0; j < k; ++j, ++v) {
                const UINT32 ncov = (type == 'G') ? 0 : bmsBinaryRecords_[r].vectorCount2;
                static const int diag6[3] = {0, 3, 5};
                for (int e = 0; e < 3; ++e) {
                    measurement_t& rec = bmsBinaryRecords_[r + e];
                    const double measPrec = e == 0 ? rec.term2 : (e == 1 ? rec.term3 : rec.term4);
                    UpdateMsrRecord(rec, -bvec[3 * v + e], prec6[6 * v + diag6[e]], measPrec);
                }
                r += 3 + 3 * (size_t)ncov;
            }
        }
        // ComputeChiSquare (ADJ:7257): the per-vector terms b.(W b) summed in CML order
        double cs = 0.0;
        for (size_t i = 0; i < nv; ++i) cs += chi[i];
        // terrestrial measurements: a S a^T from the device, the rest of UpdateMsrRecord (ADJ:8187) here
        const size_t nt = B.t_type.size();
        if (nt) {
            std::vector<double> tprec(nt), tb(nt), llh(3 * v_parameterStationList_[b].size()), xr;
            Check(dnagpu_block_terrestrial_precisions(ctx_, 0, b, var, tprec.data()), b, "ComputePrecisionAdjMsrs()");
            Check(dnagpu_block_get_terrestrial(ctx_, 0, b, tb.data(), nullptr), b, "UpdateMsrRecords()");
            Check(dnagpu_block_get_station_llh(ctx_, 0, b, llh.data()), b, "UpdateMsrRecords()");
            GetBlockStations(b, 1, xr);
            for (size_t t = 0; t < nt; ++t) {
                measurement_t& rec = bmsBinaryRecords_[B.t_rec[t]];
                const char type = B.t_type[t];
                const UINT32* l = &B.t_stn[3 * t];
                auto geo = [&](UINT32 p) {
                    const station_t& st = bstBinaryRecords_[v_parameterStationList_[b][p]];
                    return dnagpu::tm::StationGeo{llh[3 * p], llh[3 * p + 1], llh[3 * p + 2], (double)st.geoidSep, st.verticalDef, st.meridianDef};
                };
                const dnagpu::tm::StationGeo g1 = geo(l[0]), g2 = geo(dnagpu::tm::station_count(type) > 1 ? l[1] : l[0]);
                const double* X1 = &xr[3 * (size_t)l[0]];
                const double* X2 = &xr[3 * (size_t)(dnagpu::tm::station_count(type) > 1 ? l[1] : l[0])];
                if (type == 'D') {
                    // an angle of a direction set, kept in the later direction's record: derived angle scale1, variance scale2
                    // (UpdateMsrRecords_D ADJ:8121, UpdateMsrRecord ADJ:8194-8199, :8255-8261; ComputeChiSquare_D ADJ:8440 uses the
                    // angle's own variance, not the set's weight matrix)
                    const double direction = rec.term1;
                    rec.term1 = rec.scale1;
                    UpdateMsrRecord(rec, -tb[t], tprec[t], rec.scale2);
                    rec.term1 = direction;
                    if (rec.measAdj > dnagpu::tm::TWO_PI) rec.measAdj -= dnagpu::tm::TWO_PI;
                    rec.measAdj += rec.preAdjCorr;
                    cs += tb[t] * tb[t] / rec.scale2;
                    continue;
                }
                // E and M work with the ellipsoid chord derived from the supplied arc (ADJ:5254, ADJ:5412)
                rec.term1 = dnagpu::tm::working_value(type, rec.term1, rec.preAdjMeas, X1, X2, g1, g2);
                if (type == 'E' || type == 'M') rec.preAdjCorr = rec.term1 - rec.preAdjMeas;
                UpdateMsrRecord(rec, -tb[t], tprec[t], rec.term2);
                switch (type) {   // "Recompute measurements using the original types" (ADJ:8211-8268)
                    case 'E': rec.measAdj = dnagpu::tm::ellipsoid_chord_to_arc(rec.measAdj, X1, X2, g1, g2); break;
                    case 'M': rec.measAdj = dnagpu::tm::ellipsoid_chord_to_msl_arc(rec.measAdj, g1, g2); break;
                    case 'H': case 'L': case 'V': rec.measAdj -= rec.preAdjCorr; break;
                    case 'A': case 'I': case 'J': case 'K': case 'Z': rec.measAdj += rec.preAdjCorr; break;
                    default: break;
                }
                cs += tb[t] * tb[t] / rec.term2;   // ComputeChiSquare_ABCEHIJKLMPQRSVZ (ADJ:8430)
            }
            // v_precAdjMsrsFull_ runs in CML order: 6 values per GNSS vector, 1 per terrestrial measurement (ADJ:7792-7875)
            std::vector<double> merged;
            merged.reserve(6 * nv + nt);
            size_t ci = 0, ti = 0;
            while (ci < B.c_pos.size() || ti < nt) {
                if (ti >= nt || (ci < B.c_pos.size() && B.c_pos[ci] < B.t_pos[ti])) {
                    for (UINT32 vv = B.cluster_off[ci]; vv < B.cluster_off[ci + 1]; ++vv)
                        merged.insert(merged.end(), prec6.begin() + 6 * (size_t)vv, prec6.begin() + 6 * (size_t)vv + 6);
                    ++ci;
                } else {
                    merged.push_back(tprec[ti++]);
                }
            }
            B.prec_adj_msrs.swap(merged);
        }
        chiSquared += cs;
    }
    chiSquared_ += chiSquared;                                                       // ComputeChiSquareNetwork (ADJ:7315)
}

// ... and what needs every record and the network's chi-square: sigma zero, T statistics, global Pelzer reliability, the global test
void dna_adjust::StatisticsFinish() {
    const bool phased = projectSettings_.a.adjust_mode != SimultaneousMode;
    // ComputeGlobalNetStat (ADJ:6854)
    degreesofFreedom_ = (int)measurementParams_ - (int)unknownParams_;
    sigmaZero_ = sigmaZeroSqRt_ = 0.0;
    if (degreesofFreedom_ != 0) {
        sigmaZero_ = chiSquared_ / degreesofFreedom_;
        sigmaZeroSqRt_ = std::sqrt(sigmaZero_);
    }
    // ComputeTstatistics (ADJ:7094) -> UpdateMsrTstatistic_GXY (ADJ:7019)
    if (projectSettings_.o._adj_msr_tstat)
        ForEachMeasurementComponent([&](measurement_t& rec) {
            rec.TStat = std::fabs(sigmaZeroSqRt_ - 0.0) < PRECISION_1E10 ? 0.0 : rec.NStat / sigmaZeroSqRt_;
        });
    // ComputeGlobalPelzer (ADJ:8302) -> _GXY (ADJ:8396)
    double sum = 0.0;
    UINT32 numMsr = 0;
    ForEachMeasurementComponent([&](measurement_t& rec) {
        // (ADJ:8338 for the single-row types, ADJ:8408 for G / X / Y)
        const double limit = dnagpu::tm::is_terrestrial(rec.measType) ? STABLE_LIMIT : UNRELIABLE;
        if (rec.PelzerRel > 0.0 && rec.PelzerRel < limit) {
            sum += (rec.PelzerRel * rec.PelzerRel - 1.0);
            numMsr++;
        } else
            rec.PelzerRel = UNRELIABLE;
    });
    globalPelzerReliability_ = numMsr > 0 ? std::sqrt(sum / numMsr) : UNRELIABLE;
    // ComputeGlobalTestStat (ADJ:6914) -> ComputeTestStat (ADJ:6866)
    const double half = (100.0 - projectSettings_.a.confidence_interval) * 0.01 * 0.5;
    if (degreesofFreedom_ > 0) {
        chiSquaredUpperLimit_ = stat::chi_squared_quantile(degreesofFreedom_, 1.0 - half) / degreesofFreedom_;
        chiSquaredLowerLimit_ = stat::chi_squared_quantile(degreesofFreedom_, half) / degreesofFreedom_;
        if (phased) sigmaZero_ = chiSquared_ / degreesofFreedom_;
        if (sigmaZero_ < chiSquaredLowerLimit_)
            passFail_ = test_stat_warning;
        else if (sigmaZero_ > chiSquaredUpperLimit_)
            passFail_ = test_stat_fail;
        else
            passFail_ = test_stat_pass;
    } else {
        // boost::math::chi_squared throws for zero degrees of freedom; the reference reports a failed test (ADJ:6894-6908)
        passFail_ = test_stat_fail;
    }
}

// UpdateMsrRecord (ADJ:8187) + UpdateMsrRecordStats (ADJ:8291) for one X / Y / Z element of a GNSS measurement
void dna_adjust::UpdateMsrRecord(measurement_t& rec, double measCorr, double measAdjPrec, double measPrec) {
    const size_t index = (size_t)(&rec - bmsBinaryRecords_.data());
    if (index < record_touched_.size()) record_touched_[index] = 1;
    rec.measCorr = measCorr;
    rec.measAdj = rec.term1 + rec.measCorr;
    rec.measAdjPrec = measAdjPrec;
    rec.residualPrec = measPrec - rec.measAdjPrec;
    if (rec.residualPrec < 0.0) rec.residualPrec = std::fabs(rec.residualPrec);
    rec.PelzerRel = std::sqrt(measPrec) / std::sqrt(rec.residualPrec);
    if (rec.PelzerRel < 0.0 || rec.PelzerRel > STABLE_LIMIT) rec.PelzerRel = UNRELIABLE;
    rec.NStat = rec.measCorr / std::sqrt(rec.residualPrec);
    if (std::fabs(rec.NStat) > criticalValue_) potentialOutlierCount_++;
}

// visits the X, Y, Z records of every GNSS vector, block by block in CML order
void dna_adjust::ForEachMeasurementComponent(const std::function<void(measurement_t&)>& fn) {
    for (UINT32 b = 0; b < blockCount_; ++b)
        for (UINT32 m : v_CML_[b]) {
            if (bmsBinaryRecords_[m].ignore || bmsBinaryRecords_[m].measStart != 0) continue;
            const char type = bmsBinaryRecords_[m].measType;
            if (type == 'D') {      // the angles live in the records of the non-ignored directions after the first
                for (UINT32 j = 1; j < bmsBinaryRecords_[m].vectorCount1; ++j)
                    if (!bmsBinaryRecords_[m + j].ignore) fn(bmsBinaryRecords_[m + j]);
                continue;
            }
            if (dnagpu::tm::is_terrestrial(type)) {
                fn(bmsBinaryRecords_[m]);
                continue;
            }
            const UINT32 k = (type == 'G') ? 1 : bmsBinaryRecords_[m].vectorCount1;
            size_t r = m;
            for (UINT32 j = 0; j < k; ++j) {
                const UINT32 ncov = (type == 'G') ? 0 : bmsBinaryRecords_[r].vectorCount2;
                for (int e = 0; e < 3; ++e) fn(bmsBinaryRecords_[r + e]);
                r += 3 + 3 * (size_t)ncov;
            }
        }
}
// UpdateGeographicCoords (ADJ:8734) / UpdateGeographicCoordsPhased (ADJ:8711): the station records take the
// adjusted coordinates (each station once, from the block of its first appearance)
void dna_adjust::UpdateGeographicCoords() {
    std::vector<double> bx;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        GetBlockStations(b, 1, bx);
        for (size_t p = 0; p < v_parameterStationList_[b].size(); ++p) {
            if (!v_paramStnAppearance_[b][p].first_appearance_fwd) continue;
            station_t& st = bstBinaryRecords_[v_parameterStationList_[b][p]];
            geodesy::CartToGeo(bx[3 * p], bx[3 * p + 1], bx[3 * p + 2], &st.currentLatitude, &st.currentLongitude, &st.currentHeight);
        }
    }
}

namespace {
// matrix_2d binary stream layout (include/math/dnamatrix_contiguous.cpp:39-91): type, rows, cols, mem_rows, mem_cols,
// pad, data, maxvalRow, maxvalCol
void write_mtx_header(std::ofstream& f, UINT32 type, UINT32 rows, UINT32 cols) {
    const UINT32 hdr[6] = {type, rows, cols, rows, cols, 0};
    f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
}
void write_mtx_trailer(std::ofstream& f) {
    const UINT32 maxval[2] = {0, 0};
    f.write(reinterpret_cast<const char*>(maxval), sizeof(maxval));
}
}  // namespace

// ADJ:6770-6799: <network>-rva.mtx (rigorous variances, one lower-triangular matrix per block, columns left to right)
// and <network>-pam.mtx (precisions of adjusted measurements, one column vector per block), the files that
// `dnaadjust --report-results` and the printers of the reference read back (ADJ:6720-6767)
void dna_adjust::SerialiseAdjustedVarianceMatrices() {
    if (!ctx_) SignalExceptionAdjustment("SerialiseAdjustedVarianceMatrices(): PrepareAdjustment() has not been called.", 0);
    const std::string folder = projectSettings_.a.stage_path.empty() ? projectSettings_.g.output_folder : projectSettings_.a.stage_path;
    const std::string base = folder + "/" + projectSettings_.g.network_name + "-";
    // (simultaneous adjustment on several GPUs: every rank holds the same results, rank 0 writes)
    if (Distributed() && projectSettings_.a.adjust_mode == SimultaneousMode && DistRank() != 0) return;
    // one process per GPU: collective -- every block's results travel to rank 0, which writes the files
    const bool across_processes = Distributed() && peers_.empty() && !is_peer_ && DistWorld() > 1 &&
                                  projectSettings_.a.adjust_mode != SimultaneousMode;
    const bool writer = !across_processes || DistRank() == 0;
    std::ofstream rva, pam;
    auto open_files = [&] {
        if (!writer) return;
        rva.open(base + "rva.mtx", std::ios::out | std::ios::binary | std::ios::trunc);
        pam.open(base + "pam.mtx", std::ios::out | std::ios::binary | std::ios::trunc);
        if (!rva || !pam) SignalExceptionAdjustment("SerialiseAdjustedVarianceMatrices(): cannot create " + base + "rva.mtx / pam.mtx", 0);
    };
    // (across processes the per-block collection below is collective: a writer that cannot create its files must not leave the other
    //  ranks waiting in the first block's agreement -- every rank learns of it here)
    if (across_processes)
        AgreeOnPhase("creating the result files", open_files);
    else
        open_files();
    std::vector<double> packed, fetched_prec;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        if (across_processes)
            CollectBlockResults(b, packed, fetched_prec);
        else
            GetBlockRigorousVariancesPacked(b, packed);
        if (!writer) continue;
        const UINT32 n = (UINT32)v_parameterStationList_[b].size() * 3;
        write_mtx_header(rva, 1 /* mtx_lower */, n, n);
        rva.write(reinterpret_cast<const char*>(packed.data()), (std::streamsize)(packed.size() * sizeof(double)));
        write_mtx_trailer(rva);
        const std::vector<double>& prec = across_processes ? fetched_prec : GetBlockPrecAdjMsrs(b);
        const UINT32 rows = (UINT32)(6 * blocks_[b].stn1.size() + blocks_[b].t_type.size());   // v_measurementVarianceCount_ (ADJ:10513-10560)
        write_mtx_header(pam, 0 /* mtx_full */, rows, 1);
        if (prec.size() == rows)
            pam.write(reinterpret_cast<const char*>(prec.data()), (std::streamsize)(prec.size() * sizeof(double)));
        else {
            std::vector<double> zeros(rows, 0.0);   // GenerateStatistics() has not run: redim'd and zeroed (ADJ:938)
            pam.write(reinterpret_cast<const char*>(zeros.data()), (std::streamsize)(zeros.size() * sizeof(double)));
        }
        write_mtx_trailer(pam);
    }
    auto check_written = [&] {
        if (writer && (!rva || !pam)) SignalExceptionAdjustment("SerialiseAdjustedVarianceMatrices(): write failed", 0);
    };
    if (across_processes)
        AgreeOnPhase("writing the result files", check_written);
    else
        check_written();
}

// ADJ:6720-6767: the inverse of SerialiseAdjustedVarianceMatrices -- `dnaadjust --report-results` prints an earlier adjustment
// from these files without adjusting again.  PrepareAdjustment must have run (block sizes, device blocks).
void dna_adjust::DeSerialiseAdjustedVarianceMatrices() {
    if (!ctx_) SignalExceptionAdjustment("DeSerialiseAdjustedVarianceMatrices(): PrepareAdjustment() has not been called.", 0);
    const std::string folder = projectSettings_.a.stage_path.empty() ? projectSettings_.g.output_folder : projectSettings_.a.stage_path;
    const std::string base = folder + "/" + projectSettings_.g.network_name + "-";
    std::ifstream rva(base + "rva.mtx", std::ios::in | std::ios::binary);
    std::ifstream pam(base + "pam.mtx", std::ios::in | std::ios::binary);
    if (!rva || !pam) SignalExceptionAdjustment("DeSerialiseAdjustedVarianceMatrices(): cannot open " + base + "rva.mtx / pam.mtx", 0);
    const bool phased = projectSettings_.a.adjust_mode != SimultaneousMode;
    std::vector<double> packed;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        UINT32 hdr[6], tail[2];
        const UINT32 n = (UINT32)v_parameterStationList_[b].size() * 3;
        rva.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
        if (!rva || hdr[0] != 1 || hdr[1] != n || hdr[2] != n)
            SignalExceptionAdjustment("DeSerialiseAdjustedVarianceMatrices(): " + base + "rva.mtx does not match the dimensions of the network.", b);
        packed.resize((size_t)n * (n + 1) / 2);
        rva.read(reinterpret_cast<char*>(packed.data()), (std::streamsize)(packed.size() * sizeof(double)));
        rva.read(reinterpret_cast<char*>(tail), sizeof(tail));
        if (phased && Staged()) {
            // (through the staged store's own allocation: a device slot is sized for the block's packed factor as well, which a later
            //  adjustment on this handle may park there -- ADVICE r4)
            if (!blocks_[b].rig_host) AllocateStagedSlot(b);
            Check(dnagpu_copy(ctx_, blocks_[b].rig_host, packed.data(), packed.size() * sizeof(double)), b, "DeSerialiseAdjustedVarianceMatrices()");
        } else {
            dnagpu_matrix** slot = phased ? &blocks_[b].rigvar : &work_[0];
            if (!*slot) Check(dnagpu_matrix_create(ctx_, phased ? RigvarCapacity(b) : n, slot), b, "rigorous variance matrix");
            Check(dnagpu_matrix_upload_packed(ctx_, 0, *slot, packed.data(), n), b, "DeSerialiseAdjustedVarianceMatrices()");
        }
        blocks_[b].has_rigvar = true;
        pam.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
        const UINT32 rows = (UINT32)(6 * blocks_[b].stn1.size() + blocks_[b].t_type.size());
        if (!pam || hdr[0] != 0 || hdr[1] != rows || hdr[2] != 1)
            SignalExceptionAdjustment("DeSerialiseAdjustedVarianceMatrices(): " + base + "pam.mtx does not match the dimensions of the network.", b);
        blocks_[b].prec_adj_msrs.resize(rows);
        pam.read(reinterpret_cast<char*>(blocks_[b].prec_adj_msrs.data()), (std::streamsize)(rows * sizeof(double)));
        pam.read(reinterpret_cast<char*>(tail), sizeof(tail));
        if (!rva || !pam) SignalExceptionAdjustment("DeSerialiseAdjustedVarianceMatrices(): read failed", b);
    }
}

// ADJ:2562: printed to stderr when DYNADJUST_PROFILE is set
void dna_adjust::PrintPerformanceProfile() const {
    if (!profileTimings_) return;
    auto ms = [](uint64_t ns) { return (double)ns / 1.0e6; };
    std::stringstream ss;
    ss << "DynAdjust profile timings:" << std::fixed << std::setprecision(3) << " update_normals=" << ms(profileUpdateNormalsNs_.load()) << "ms"
       << " stage_load=" << ms(profileStageLoadNs_.load()) << "ms"
       << " stage_store=" << ms(profileStageStoreNs_.load()) << "ms";
    fprintf(stderr, "%s\n", ss.str().c_str());
}

// ADJ:10628: the number of blocks of a segmentation file, before anything is prepared
void dna_adjust::LoadSegmentationFileParameters(const std::string& seg_filename) {
    try {
        iostreams::seg_data_t seg;
        iostreams::read_seg(seg_filename, seg, nullptr);
        blockCount_ = seg.blockCount;
    } catch (const std::runtime_error& e) {
        SignalExceptionAdjustment(e.what(), 0);
    }
}

void dna_adjust::NoteIterationDone(double t0_ms) {
    std::lock_guard<std::mutex> lk(msg_mutex_);
    if (iterationMs_.size() < currentIteration_) iterationMs_.resize(currentIteration_, 0.0);
    iterationMs_[currentIteration_ - 1] = now_ms() - t0_ms;
    iterationQueue_.push_back(currentIteration_);
}

bool dna_adjust::NewMessagesAvailable() {
    std::lock_guard<std::mutex> lk(msg_mutex_);
    return !iterationQueue_.empty();
}
bool dna_adjust::GetMessageIteration(UINT32& iteration) {
    std::lock_guard<std::mutex> lk(msg_mutex_);
    if (iterationQueue_.empty()) return false;
    iteration = iterationQueue_.front();
    iterationQueue_.pop_front();
    return true;
}
std::string dna_adjust::GetMaxCorrection(const UINT32& iteration) const {
    if (iteration == 0 || iteration > iterationCorrections_.size()) return std::string();
    std::stringstream ss;
    ss << std::fixed << std::setprecision(4) << iterationCorrections_[iteration - 1];
    return ss.str();
}

DynAdjustPrinter* dna_adjust::GetPrinter() {
    if (!printer_) printer_.reset(new DynAdjustPrinter(*this));
    return printer_.get();
}

void dna_adjust::CloseOutputFiles() {
    if (printer_) printer_->Close();
}

// dna_adjust::UpdateIterationDiagnostics (ADJ:7450-7554): after every iteration, block by block in order, every station's correction
// against the one it was last seen with; a station whose corrections have turned round twice in a row with similar size is recorded
// (first / last iteration, cycle count, first / last magnitude and the last correction in the local frame).  The comparison runs on the
// device, where the corrections are (dnagpu_osc_block: osc_update_kernel, state per station of the network); the host keeps
// oscHistory_ and only fetches a block's corrections when one of its visits was flagged -- none in an adjustment that converges.
// Which corrections: those of the block's LAST solution, as in the reference's v_corrections_ -- under the reference's schedule
// (a.schur_carry = 0, one chain) the same vectors, the last block's reverse solve "in isolation" included; under the condensed schedule
// every block has one solution per iteration, the rigorous one.  Across GPUs every rank watches its own blocks.
void dna_adjust::UpdateIterationDiagnostics() {
    if (!ctx_ || blockCount_ == 0) return;
    if (!osc_ready_) {
        oscHistory_.clear();
        Check(dnagpu_osc_reset(ctx_, bstBinaryRecords_.size()), 0, "UpdateIterationDiagnostics()");
        osc_ready_ = true;
    }
    {
        // (all blocks of this rank in one launch, in block order)
        std::vector<UINT32> ids;
        std::vector<int> chains;
        std::vector<const UINT32*> lists;
        for (UINT32 b = 0; b < blockCount_; ++b) {
            if (!OwnsBlock(b) || v_parameterStationList_[b].empty()) continue;
            ids.push_back(b);
            chains.push_back(blocks_[b].corr_chain);
            lists.push_back(v_parameterStationList_[b].data());
        }
        Check(dnagpu_osc_blocks(ctx_, (uint32_t)ids.size(), ids.data(), chains.data(), lists.data()), 0, "UpdateIterationDiagnostics()");
    }
    UINT32 flagged = 0;
    Check(dnagpu_osc_flagged(ctx_, &flagged), 0, "UpdateIterationDiagnostics()");
    if (!flagged) return;
    std::vector<UINT32> visit;
    std::vector<double> corr;
    for (UINT32 b = 0; b < blockCount_; ++b) {
        if (!OwnsBlock(b) || v_parameterStationList_[b].empty()) continue;
        const std::vector<UINT32>& plist = v_parameterStationList_[b];
        visit.resize(plist.size());
        Check(dnagpu_osc_block_visits(ctx_, b, visit.data()), b, "UpdateIterationDiagnostics()");
        bool any = false;
        for (UINT32 v : visit) any = any || v != 0;
        if (!any) continue;
        corr.resize(3 * plist.size());
        Check(dnagpu_block_get_corrections(ctx_, blocks_[b].corr_chain, b, corr.data()), b, "UpdateIterationDiagnostics()");
        for (size_t s = 0; s < plist.size(); ++s) {
            if (!visit[s]) continue;
            const UINT32 stnIdx = plist[s];
            // Rotate_CartLocal at the station's current position (ADJ:7511-7518)
            double R[3][3];
            geodesy::LocalToCartRotation(bstBinaryRecords_[stnIdx].currentLatitude, bstBinaryRecords_[stnIdx].currentLongitude, R);
            double l[3];
            for (int i = 0; i < 3; ++i) l[i] = R[0][i] * corr[3 * s] + R[1][i] * corr[3 * s + 1] + R[2][i] * corr[3 * s + 2];
            const double localMag = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
            auto hit = oscHistory_.find(stnIdx);
            if (hit == oscHistory_.end()) {
                OscillationRecord rec;
                rec.stnBstIdx = stnIdx;
                rec.firstIteration = rec.lastIteration = currentIteration_;
                rec.maxCycles = visit[s];
                rec.firstMag = rec.lastMag = localMag;
                rec.lastE = l[0];
                rec.lastN = l[1];
                rec.lastUp = l[2];
                oscHistory_[stnIdx] = rec;
            } else {
                hit->second.lastIteration = currentIteration_;
                hit->second.maxCycles = visit[s];
                hit->second.lastMag = localMag;
                hit->second.lastE = l[0];
                hit->second.lastN = l[1];
                hit->second.lastUp = l[2];
            }
        }
    }
}

// ADJ:7556-7608: the twenty stations with the largest oscillation (0.1 m and more), in the reference's words
void dna_adjust::PrintOscillationSummary(std::ostream& os) {
    if (oscHistory_.empty()) return;
    std::vector<const OscillationRecord*> sorted;
    for (auto& kv : oscHistory_)
        if (std::max(kv.second.firstMag, kv.second.lastMag) >= 0.1) sorted.push_back(&kv.second);
    if (sorted.empty()) return;
    std::stable_sort(sorted.begin(), sorted.end(), [](const OscillationRecord* a, const OscillationRecord* b) {
        return std::max(a->firstMag, a->lastMag) > std::max(b->firstMag, b->lastMag);
    });
    const size_t limit = std::min(sorted.size(), (size_t)20);
    os << std::endl;
    os << "+ Oscillating stations detected (" << sorted.size() << " total, showing top " << limit << "):" << std::endl;
    for (size_t i = 0; i < limit; ++i) {
        const OscillationRecord* rec = sorted[i];
        const double horizMag = std::sqrt(rec->lastE * rec->lastE + rec->lastN * rec->lastN), vertMag = std::fabs(rec->lastUp);
        const char* direction = vertMag < 0.01 * horizMag ? "horizontal" : horizMag < 0.01 * vertMag ? "vertical" : "3D";
        const station_t& st = bstBinaryRecords_.at(rec->stnBstIdx);
        os << "  - " << std::string(st.stationName, strnlen(st.stationName, sizeof(st.stationName))) << std::fixed << std::setprecision(1) << " \xe2\x80\x94 "
           << rec->firstMag << "m to " << rec->lastMag << "m" << ", " << direction << ", " << rec->maxCycles << " cycles" << " (iterations "
           << rec->firstIteration << "-" << rec->lastIteration << ")" << std::endl;
    }
}

// GetMsrStations (include/functions/dnatemplatestnmsrfuncs.hpp:70-131): the stations of the measurement a .bms record belongs to --
// from its first-component records only (a G baseline's Y / Z rows name none), a cluster's from its beginning
void dna_adjust::GetMsrStations(UINT32 msrIndex, std::vector<UINT32>& out) const {
    out.clear();
    const UINT32 id = bmsBinaryRecords_[msrIndex].clusterID;
    size_t i = msrIndex;
    bool cluster = false;
    switch (bmsBinaryRecords_[msrIndex].measType) {
        case 'D': case 'X': case 'Y':
            while (i > 0 && bmsBinaryRecords_[i - 1].clusterID == id) --i;
            cluster = true;
            break;
        default: break;
    }
    while (i < bmsBinaryRecords_.size() && bmsBinaryRecords_[i].clusterID == id) {
        const measurement_t& m = bmsBinaryRecords_[i];
        if (m.measType != 'D' && m.measStart != 0) {
            ++i;
            continue;
        }
        out.push_back(m.station1);
        const char t = m.measType;
        const bool one = t == 'H' || t == 'I' || t == 'J' || t == 'P' || t == 'Q' || t == 'R' || t == 'Y';
        if (!one) out.push_back(m.station2);
        if (t == 'A') out.push_back(m.station3);
        if (!cluster) break;
        ++i;
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
}

bool dna_adjust::MeasurementTouchesOscillatingStation(UINT32 msrIndex) const {
    if (oscHistory_.empty() || msrIndex >= bmsBinaryRecords_.size()) return false;
    std::vector<UINT32> stns;
    GetMsrStations(msrIndex, stns);
    for (UINT32 s : stns)
        if (oscHistory_.count(s)) return true;
    return false;
}

std::string dna_adjust::MeasurementStationNames(UINT32 msrIndex) const {
    if (msrIndex >= bmsBinaryRecords_.size()) return "(measurement index unavailable)";
    std::vector<UINT32> stns;
    GetMsrStations(msrIndex, stns);
    std::string names;
    for (UINT32 s : stns) {
        if (s >= bstBinaryRecords_.size()) continue;
        if (!names.empty()) names += " -> ";
        names += std::string(bstBinaryRecords_[s].stationName, strnlen(bstBinaryRecords_[s].stationName, sizeof(bstBinaryRecords_[s].stationName)));
    }
    return names.empty() ? "(stations unavailable)" : names;
}

// ADJ:7652-7780: the measurements tied to an oscillating station, then those whose N-statistic exceeds the critical value of the chosen
// confidence interval, each list by |N-stat| (ties by record), `limit` lines each, in the reference's words
void dna_adjust::PrintSuspectMeasurementSummary(std::ostream& os, size_t limit) const {
    if (bmsBinaryRecords_.empty() || limit == 0) return;
    struct rec_t {
        UINT32 idx;
        double absN;
        bool critical, osc;
    };
    std::vector<rec_t> osc, out;
    for (UINT32 i = 0; i < bmsBinaryRecords_.size(); ++i) {
        const measurement_t& m = bmsBinaryRecords_[i];
        if (m.ignore || !std::isfinite(m.NStat) || !std::isfinite(m.residualPrec) || m.residualPrec <= 0.0) continue;
        const double absN = std::fabs(m.NStat);
        const bool critical = absN > criticalValue_, touches = MeasurementTouchesOscillatingStation(i);
        if (!critical && !touches) continue;
        if (touches) osc.push_back({i, absN, critical, true});
        if (critical && !touches) out.push_back({i, absN, true, false});
    }
    if (osc.empty() && out.empty()) return;
    auto by_nstat = [](const rec_t& a, const rec_t& b) { return a.absN == b.absN ? a.idx < b.idx : a.absN > b.absN; };
    std::sort(osc.begin(), osc.end(), by_nstat);
    std::sort(out.begin(), out.end(), by_nstat);
    const std::ios::fmtflags flags = os.flags();
    const std::streamsize prec = os.precision();
    auto print_list = [&](const std::string& title, const std::vector<rec_t>& list) {
        if (list.empty()) return;
        const size_t shown = std::min(list.size(), limit);
        os << std::endl << "+ " << title << " (" << list.size() << " total, showing top " << shown << "):" << std::endl;
        for (size_t k = 0; k < shown; ++k) {
            const measurement_t& m = bmsBinaryRecords_[list[k].idx];
            os << "  - " << m.measType << " msr " << list[k].idx << " cluster " << m.clusterID << " file-order " << m.fileOrder << " "
               << MeasurementStationNames(list[k].idx) << ": N=" << std::fixed << std::setprecision(2) << m.NStat;
            if (std::isfinite(m.TStat) && std::fabs(m.TStat) > 0.0) os << ", T=" << std::fixed << std::setprecision(2) << m.TStat;
            os << ", corr=" << std::scientific << std::setprecision(3) << m.measCorr << ", residual precision=" << std::scientific << std::setprecision(3)
               << m.residualPrec << ", Pelzer=" << std::fixed << std::setprecision(2) << m.PelzerRel;
            if (list[k].critical) os << ", exceeds critical";
            if (list[k].osc) os << ", touches oscillating station";
            os << std::endl;
        }
    };
    print_list("Suspect measurements connected to oscillating stations", osc);
    print_list(osc.empty() ? "Largest measurement N-statistics" : "Largest remaining measurement N-statistics", out);
    os.flags(flags);
    os.precision(prec);
}

std::string dna_adjust::GetIterationTime(const UINT32& iteration) const {
    if (iteration == 0 || iteration > iterationMs_.size()) return std::string();
    std::stringstream ss;
    ss << std::fixed << std::setprecision(3) << iterationMs_[iteration - 1] / 1000.0 << "s";
    return ss.str();
}

// ADJ:445-470: the station and measurement records (adjusted coordinates, adjusted measurements and their
// statistics, scaled variances) go back to the .bst / .bms files, flagged as reduced
void dna_adjust::UpdateBinaryFiles() {
    // one process per GPU: every rank holds the same records after the collective GenerateStatistics(); rank 0 writes them
    if (Distributed() && peers_.empty() && !is_peer_ && DistRank() != 0) return;
    try {
        snprintf(bst_meta_.modifiedBy, sizeof(bst_meta_.modifiedBy), "%s", "dnaadjust");
        bst_meta_.reduced = true;
        iostreams::write_bst(projectSettings_.a.bst_file, bstBinaryRecords_, bst_meta_, "dnaadjust");
        snprintf(bms_meta_.modifiedBy, sizeof(bms_meta_.modifiedBy), "%s", "dnaadjust");
        bms_meta_.reduced = true;
        iostreams::write_bms(projectSettings_.a.bms_file, bmsBinaryRecords_, bms_meta_, "dnaadjust");
    } catch (const std::runtime_error& e) {
        SignalExceptionAdjustment(e.what(), 0);
    }
}

}  // namespace networkadjust
}  // namespace dynadjust
