// Synthetic GNSS networks written as real .bst/.bms/.asl/.seg files:
// an R x C grid of stations, E / N / NE baselines thinned to an exact count, full 3x3
// baseline VCVs, four constrained corner stations and a strip segmentation into B blocks
// whose junction stations satisfy JSL(k) subset ISL(k+1) (the invariant the reference's
// CarryStnEstimatesandVariancesForward relies on, dnaadjust.cpp:1072).
#include "synth.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <stdexcept>

#include "dnaio.hpp"
#include "geodesy.hpp"
#include "gnss_vcv.hpp"

namespace dynadjust {
namespace synth {

namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
    }
};

void put_str(char* dst, size_t cap, const char* src) {
    memset(dst, 0, cap);
    snprintf(dst, cap, "%s", src);
}

}  // namespace

void write_network(const std::string& dir, const std::string& name, const Spec& sp, Summary* summary) {
    using namespace geodesy;
    const uint32_t R = sp.rows, C = sp.cols;
    // strip k = grid rows cut[k] .. cut[k + 1] - 1.  Equal strips: row r belongs to strip floor(r * B / R) (every strip floor(R/B) or
    // ceil(R/B) rows); sp.ragged / sp.rows_hi: uneven heights from a generator of their own (the stream of the stations and measurements --
    // and with it every committed record of an equal-strip network -- is untouched)
    std::vector<uint32_t> cut;
    {
        SplitMix64 hr(sp.seed ^ 0x5EC7105A11ull);
        if (sp.rows_hi > 0) {
            const uint32_t lo = std::max<uint32_t>(1, std::min(sp.rows_lo, sp.rows_hi)), hi = std::max(lo, sp.rows_hi);
            cut.push_back(0);
            while (cut.back() < R) {
                uint32_t h = lo + (uint32_t)(hr.uniform() * (double)(hi - lo + 1));
                if (h > hi) h = hi;
                if (R - cut.back() < h + lo) h = R - cut.back();          // (the last strip takes what is left rather than leave a sliver)
                cut.push_back(cut.back() + h);
            }
        } else {
            const uint32_t Bq = std::max<uint32_t>(1, sp.n_blocks);
            if (Bq > R) throw std::runtime_error("synth: more blocks than grid rows");
            if (sp.ragged > 0.0 && Bq > 1) {
                if (sp.ragged >= 1.0) throw std::runtime_error("synth: ragged must be below 1");
                std::vector<double> w(Bq);
                double sum = 0.0;
                for (double& x : w) sum += (x = 1.0 + sp.ragged * (2.0 * hr.uniform() - 1.0));
                cut.assign(Bq + 1, 0);
                double acc = 0.0;
                for (uint32_t k = 0; k < Bq; ++k) {
                    acc += w[k];
                    uint32_t c = (uint32_t)std::llround(acc / sum * (double)R);
                    c = std::max(c, cut[k] + 1);                          // at least one row each ...
                    c = std::min(c, R - (Bq - 1 - k));                    // ... and one left for every strip behind
                    cut[k + 1] = c;
                }
                cut[Bq] = R;
            } else {
                for (uint32_t k = 0; k <= Bq; ++k) cut.push_back((uint32_t)(((uint64_t)k * R + Bq - 1) / Bq));   // = first r with floor(r * B / R) >= k
            }
        }
    }
    const uint32_t B = (uint32_t)cut.size() - 1;
    if (R < 2 || C < 2) throw std::runtime_error("synth: the grid needs at least 2 x 2 stations");
    const uint64_t n_stn = (uint64_t)R * C;
    const uint64_t n_cand = (uint64_t)R * (C - 1) + (uint64_t)(R - 1) * C + (uint64_t)(R - 1) * (C - 1);
    uint64_t target = sp.n_baselines ? sp.n_baselines : n_cand;
    if (target > n_cand) throw std::runtime_error("synth: more baselines requested than the grid has E/N/NE neighbours");
    const Ellipsoid ell;
    const double deg = 3.14159265358979323846 / 180.0;
    const double lat0 = -36.5 * deg, lon0 = 146.0 * deg, step = 0.01 * deg;
    SplitMix64 rng(sp.seed);

    // stations: truth, then perturbed initial coordinates
    std::vector<double> truth(3 * n_stn);
    std::vector<double> tlat(n_stn), tlon(n_stn);
    std::vector<station_t> bst(n_stn);
    for (uint32_t r = 0; r < R; ++r)
        for (uint32_t c = 0; c < C; ++c) {
            uint64_t s = (uint64_t)r * C + c;
            double lat = lat0 + ((double)r - 0.5 * (R - 1)) * step;
            double lon = lon0 + ((double)c - 0.5 * (C - 1)) * step;
            double h = 200.0 + 100.0 * rng.uniform();
            tlat[s] = lat;
            tlon[s] = lon;
            GeoToCart(lat, lon, h, &truth[3 * s], &truth[3 * s + 1], &truth[3 * s + 2], ell);
            double x = truth[3 * s] + sp.initial_sigma * rng.normal();
            double y = truth[3 * s + 1] + sp.initial_sigma * rng.normal();
            double z = truth[3 * s + 2] + sp.initial_sigma * rng.normal();
            double ilat, ilon, ih;
            CartToGeo(x, y, z, &ilat, &ilon, &ih, ell);
            station_t& st = bst[s];
            memset(&st, 0, sizeof(st));
            char nm[32];
            snprintf(nm, sizeof(nm), "S%08llu", (unsigned long long)s);
            put_str(st.stationName, sizeof(st.stationName), nm);
            put_str(st.stationNameOrig, sizeof(st.stationNameOrig), nm);
            bool corner = (r == 0 || r == R - 1) && (c == 0 || c == C - 1);
            put_str(st.stationConst, sizeof(st.stationConst), corner ? "CCC" : "FFF");
            put_str(st.stationType, sizeof(st.stationType), "LLh");
            st.suppliedStationType = LLh_type_i;
            st.initialLatitude = st.currentLatitude = ilat;
            st.initialLongitude = st.currentLongitude = ilon;
            st.initialHeight = st.currentHeight = ih;
            st.suppliedHeightRefFrame = ELLIPSOIDAL_type_i;
            st.fileOrder = (UINT32)s;
            st.nameOrder = (UINT32)s;
            put_str(st.epsgCode, sizeof(st.epsgCode), "7843");
            put_str(st.epoch, sizeof(st.epoch), "01.01.2020");
            put_str(st.observation_epoch, sizeof(st.observation_epoch), "01.01.2020");
            put_str(st.description, sizeof(st.description), "synthetic grid station");
        }

    // baselines: selection sampling of exactly `target` of the candidate edges (Knuth 3.4.2 S)
    std::vector<uint32_t> bl_s1, bl_s2;
    bl_s1.reserve(target);
    bl_s2.reserve(target);
    {
        uint64_t remaining = n_cand, need = target;
        auto consider = [&](uint64_t s1, uint64_t s2) {
            bool take = (double)need > rng.uniform() * (double)remaining;
            --remaining;
            if (!take || need == 0) return;
            --need;
            bl_s1.push_back((uint32_t)s1);
            bl_s2.push_back((uint32_t)s2);
        };
        for (uint32_t r = 0; r < R; ++r)
            for (uint32_t c = 0; c < C; ++c) {
                uint64_t s = (uint64_t)r * C + c;
                if (c + 1 < C) consider(s, s + 1);
                if (r + 1 < R) consider(s, s + C);
                if (r + 1 < R && c + 1 < C) consider(s, s + C + 1);
            }
    }
    const uint64_t n_bl = bl_s1.size();

    // measurement records.  A run of baselines leaving one station becomes one 'X' cluster for the first
    // sp.x_clusters stations that have at least two; every other baseline is a 'G'.
    std::vector<measurement_t> bms;
    bms.reserve(3 * n_bl + 64);
    std::vector<uint32_t> msr_first;      // first record of every measurement (CML entry)
    std::vector<uint32_t> msr_station;    // the station whose strip owns the measurement
    const double sd[3] = {sp.sigma_e, sp.sigma_n, sp.sigma_up};
    uint32_t cluster_id = 1;
    auto base_record = [&](char type, int start, uint32_t s1, uint32_t s2, uint32_t id) {
        measurement_t m;
        memset(&m, 0, sizeof(m));
        m.measType = type;
        m.measStart = (char)start;
        m.measurementStations = (type == 'Y') ? 1 : 2;
        put_str(m.epsgCode, sizeof(m.epsgCode), "7843");
        put_str(m.epoch, sizeof(m.epoch), "01.01.2020");
        put_str(m.observation_epoch, sizeof(m.observation_epoch), "01.01.2020");
        put_str(m.coordType, sizeof(m.coordType), "XYZ");
        m.ignore = false;
        m.station1 = s1;
        m.station2 = s2;
        m.clusterID = id;
        m.fileOrder = id;
        m.scale1 = m.scale2 = m.scale3 = m.scale4 = 1.0;
        return m;
    };
    // emits a cluster of k vectors with the full symmetric variance matrix V (3k x 3k, row-major here)
    auto emit_cluster = [&](char type, const std::vector<uint32_t>& s1, const std::vector<uint32_t>& s2,
                            const std::vector<double>& obs, const std::vector<double>& V) {
        const uint32_t k = (uint32_t)s2.size(), nc = 3 * k;
        msr_first.push_back((uint32_t)bms.size());
        msr_station.push_back(type == 'Y' ? s2[0] : s1[0]);
        const uint32_t id = cluster_id++;
        for (uint32_t j = 0; j < k; ++j) {
            for (int e = 0; e < 3; ++e) {
                measurement_t m = base_record(type, e, type == 'Y' ? s2[j] : s1[j], type == 'Y' ? 0 : s2[j], id);
                m.vectorCount1 = (type == 'G') ? (e == 0 ? 1 : 0) : k;
                m.vectorCount2 = (type == 'G') ? 0 : k - 1 - j;
                m.term1 = obs[3 * j + e];
                const uint32_t r0 = 3 * j;
                if (e == 0) {
                    m.term2 = V[(size_t)r0 * nc + r0];
                } else if (e == 1) {
                    m.term2 = V[(size_t)r0 * nc + r0 + 1];
                    m.term3 = V[(size_t)(r0 + 1) * nc + r0 + 1];
                } else {
                    m.term2 = V[(size_t)r0 * nc + r0 + 2];
                    m.term3 = V[(size_t)(r0 + 1) * nc + r0 + 2];
                    m.term4 = V[(size_t)(r0 + 2) * nc + r0 + 2];
                }
                m.preAdjMeas = m.term1;
                bms.push_back(m);
            }
            for (uint32_t c = j + 1; c < k; ++c)
                for (int r = 0; r < 3; ++r) {
                    measurement_t m = base_record(type, 3 + r, type == 'Y' ? s2[j] : s1[j], type == 'Y' ? 0 : s2[j], id);
                    m.vectorCount1 = k;
                    m.vectorCount2 = k - 1 - j;
                    m.term1 = V[(size_t)(3 * j + r) * nc + 3 * c];
                    m.term2 = V[(size_t)(3 * j + r) * nc + 3 * c + 1];
                    m.term3 = V[(size_t)(3 * j + r) * nc + 3 * c + 2];
                    bms.push_back(m);
                }
        }
    };
    // variance matrix of k vectors: block-diagonal local-frame variances + a common low-rank part (correlations)
    auto make_vcv = [&](const std::vector<uint32_t>& at, bool correlated, std::vector<double>& V, std::vector<double>& noise) {
        const uint32_t k = (uint32_t)at.size(), nc = 3 * k;
        V.assign((size_t)nc * nc, 0.0);
        noise.assign(nc, 0.0);
        for (uint32_t j = 0; j < k; ++j) {
            double Rm[3][3];
            LocalToCartRotation(tlat[at[j]], tlon[at[j]], Rm);
            double zr[3] = {rng.normal(), rng.normal(), rng.normal()};
            for (int i = 0; i < 3; ++i) {
                for (int q = 0; q < 3; ++q) {
                    double v = 0.0;
                    for (int t = 0; t < 3; ++t) v += Rm[i][t] * sd[t] * sd[t] * Rm[q][t];
                    V[(size_t)(3 * j + i) * nc + 3 * j + q] = v;
                }
                for (int t = 0; t < 3; ++t) noise[3 * j + i] += Rm[i][t] * sd[t] * zr[t];
            }
        }
        if (correlated && k > 1) {
            std::vector<double> g((size_t)nc * 2);
            for (double& x : g) x = 0.002 * rng.normal();
            for (uint32_t i = 0; i < nc; ++i)
                for (uint32_t q = 0; q < nc; ++q) V[(size_t)i * nc + q] += g[2 * i] * g[2 * q] + g[2 * i + 1] * g[2 * q + 1];
        }
    };
    uint32_t x_left = sp.x_clusters;
    for (uint64_t i = 0; i < n_bl;) {
        uint64_t e = i;
        while (e < n_bl && bl_s1[e] == bl_s1[i]) ++e;
        const bool cluster = x_left > 0 && e - i >= 2;
        const uint64_t step = cluster ? e - i : 1;
        std::vector<uint32_t> s1(bl_s1.begin() + i, bl_s1.begin() + i + step), s2(bl_s2.begin() + i, bl_s2.begin() + i + step);
        std::vector<double> V, noise, obs(3 * step);
        make_vcv(s1, cluster, V, noise);
        for (uint64_t j = 0; j < step; ++j)
            for (int c = 0; c < 3; ++c) obs[3 * j + c] = (truth[3 * (size_t)s2[j] + c] - truth[3 * (size_t)s1[j] + c]) + noise[3 * j + c];
        emit_cluster(cluster ? 'X' : 'G', s1, s2, obs, V);
        if (cluster) --x_left;
        i += step;
    }
    if (sp.y_cluster) {
        // the datum as a GNSS point cluster over the four corner stations (which are then free, FFF)
        std::vector<uint32_t> none, corners = {0u, (uint32_t)(C - 1), (uint32_t)((uint64_t)(R - 1) * C), (uint32_t)((uint64_t)R * C - 1)};
        std::vector<double> V, noise, obs(12);
        make_vcv(corners, true, V, noise);
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 3; ++c) obs[3 * j + c] = truth[3 * (size_t)corners[j] + c] + noise[3 * j + c];
        // one cluster lives in one block: the corners of the first and the last strip are tied by one Y cluster each
        std::vector<uint32_t> lo(corners.begin(), corners.begin() + 2), hi(corners.begin() + 2, corners.end());
        auto sub = [&](int off, std::vector<double>& Vs, std::vector<double>& os) {
            Vs.assign(36, 0.0);
            os.assign(obs.begin() + 6 * off, obs.begin() + 6 * off + 6);
            for (int i = 0; i < 6; ++i)
                for (int q = 0; q < 6; ++q) Vs[i * 6 + q] = V[(size_t)(6 * off + i) * 12 + 6 * off + q];
        };
        // sp.y_llh: the point clusters are written the way a geodetic campaign supplies them, latitude / longitude /
        // height with a variance matrix in that frame (radians^2, m^2): the first with ellipsoidal heights ("LLh"), the
        // second with orthometric heights ("LLH", reduced with the stations' geoid separation)
        auto to_geographic = [&](const std::vector<uint32_t>& pts, std::vector<double>& Vs, std::vector<double>& os, bool ortho) {
            const uint32_t k = (uint32_t)pts.size(), nc = 3 * k;
            std::vector<double> Vc((size_t)nc * nc), llh(3 * (size_t)k);
            for (uint32_t i = 0; i < nc; ++i)
                for (uint32_t q = 0; q < nc; ++q) Vc[(size_t)q * nc + i] = Vs[(size_t)i * nc + q];   // row-major -> column-major
            for (uint32_t j = 0; j < k; ++j) {
                station_t& st = bst[pts[j]];
                if (ortho) st.geoidSep = 7.25f + 0.5f * (float)j;
                llh[3 * j] = st.currentLatitude;
                llh[3 * j + 1] = st.currentLongitude;
                llh[3 * j + 2] = st.currentHeight;
                double lat, lon, h;
                CartToGeo(os[3 * j], os[3 * j + 1], os[3 * j + 2], &lat, &lon, &h);
                os[3 * j] = lat;
                os[3 * j + 1] = lon;
                os[3 * j + 2] = ortho ? h - (double)st.geoidSep : h;
            }
            gnssvcv::PropagateGeoCart(Vc, k, llh, false);
            for (uint32_t i = 0; i < nc; ++i)
                for (uint32_t q = 0; q < nc; ++q) Vs[(size_t)i * nc + q] = 0.5 * (Vc[(size_t)q * nc + i] + Vc[(size_t)i * nc + q]);
        };
        auto set_coord_type = [&](size_t rec0, const char* ct) {
            for (size_t r = rec0; r < bms.size(); ++r) put_str(bms[r].coordType, sizeof(bms[r].coordType), ct);
        };
        std::vector<double> Vs, os;
        sub(0, Vs, os);
        size_t rec0 = bms.size();
        if (sp.y_llh) to_geographic(lo, Vs, os, false);
        emit_cluster('Y', none, lo, os, Vs);
        if (sp.y_llh) set_coord_type(rec0, "LLh");
        sub(1, Vs, os);
        rec0 = bms.size();
        if (sp.y_llh) to_geographic(hi, Vs, os, true);
        emit_cluster('Y', none, hi, os, Vs);
        if (sp.y_llh) set_coord_type(rec0, "LLH");
        for (uint32_t s : corners) put_str(bst[s].stationConst, sizeof(bst[s].stationConst), "FFF");
    }
    if (sp.scalars) {
        // variance scalars (the DNA format's v-, p-, l-, h-scale columns): every third measurement carries
        // phi / lambda / height scalars, every fourth a matrix scalar, some both
        uint32_t q = 0, last_id = 0xffffffffu;
        double ps = 1.0, ls = 1.0, hs = 1.0, vs = 1.0;
        for (measurement_t& rec : bms) {
            if (rec.clusterID != last_id) {
                last_id = rec.clusterID;
                ++q;
                const bool partial = (q % 3 == 1);
                ps = partial ? 2.0 : 1.0;
                ls = partial ? 3.0 : 1.0;
                hs = partial ? 0.5 : 1.0;
                vs = (q % 4 == 2) ? 4.0 : ((q % 12 == 1) ? 1.5 : 1.0);
            }
            rec.scale1 = ps;
            rec.scale2 = ls;
            rec.scale3 = hs;
            rec.scale4 = vs;
        }
    }
    const uint64_t n_msr = msr_first.size();

    // associated station list
    std::vector<asl_entry_t> asl(n_stn);
    for (const measurement_t& m : bms) {
        if (m.measStart != 0) continue;
        asl[m.station1].assocMsrCount++;
        if (m.measType != 'Y') asl[m.station2].assocMsrCount++;
    }
    uint32_t off = 0;
    for (uint64_t s = 0; s < n_stn; ++s) {
        asl[s].amlStnIndex = off;
        off += asl[s].assocMsrCount;
        asl[s].validity = 1;
    }

    // strip segmentation
    iostreams::seg_data_t seg;
    seg.blockCount = B;
    seg.blockThreshold = 0;
    seg.minInnerStns = 0;
    seg.ISL.assign(B, {});
    seg.JSL.assign(B, {});
    seg.CML.assign(B, {});
    seg.ContiguousNetList.assign(B, 0);
    std::vector<uint32_t> strip_of_row(R);
    for (uint32_t k = 0; k < B; ++k)
        for (uint32_t r = cut[k]; r < cut[k + 1]; ++r) strip_of_row[r] = k;
    auto strip_of = [&](uint64_t s) { return strip_of_row[(size_t)(s / C)]; };
    for (uint64_t s = 0; s < n_stn; ++s) seg.ISL[strip_of(s)].push_back((UINT32)s);
    for (uint64_t q = 0; q < n_msr; ++q) {
        // a measurement belongs to the strip of the station it leaves; end stations in the next strip are junctions
        const uint32_t k = strip_of(msr_station[q]);
        seg.CML[k].push_back((UINT32)msr_first[q]);
        const uint32_t end = (q + 1 < n_msr) ? msr_first[q + 1] : (uint32_t)bms.size();
        for (uint32_t r = msr_first[q]; r < end; ++r) {
            const measurement_t& m = bms[r];
            if (m.measStart != 0) continue;
            if (strip_of(m.station1) != k) seg.JSL[k].push_back(m.station1);
            if (m.measType != 'Y' && strip_of(m.station2) != k) seg.JSL[k].push_back(m.station2);
        }
    }
    for (uint32_t k = 0; k < B; ++k) {
        std::sort(seg.JSL[k].begin(), seg.JSL[k].end());
        seg.JSL[k].erase(std::unique(seg.JSL[k].begin(), seg.JSL[k].end()), seg.JSL[k].end());
        if (seg.ISL[k].empty()) throw std::runtime_error("synth: empty strip (reduce the block count)");
    }

    binary_file_meta_t meta;
    put_str(meta.modifiedBy, sizeof(meta.modifiedBy), "dnagpu-synth");
    put_str(meta.epsgCode, sizeof(meta.epsgCode), "7843");
    put_str(meta.epoch, sizeof(meta.epoch), "01.01.2020");
    put_str(meta.observation_epoch, sizeof(meta.observation_epoch), "01.01.2020");
    meta.reduced = false;
    const std::string base = dir + "/" + name;
    iostreams::write_bst(base + ".bst", bst, meta);
    iostreams::write_bms(base + ".bms", bms, meta);
    iostreams::write_asl(base + ".asl", asl);
    iostreams::write_seg(base + ".seg", seg, base + ".bst", base + ".bms", bms);
    {
        std::ofstream t(base + ".truth", std::ios::binary | std::ios::trunc);
        t.write(reinterpret_cast<const char*>(truth.data()), (std::streamsize)(truth.size() * sizeof(double)));
        if (!t) throw std::runtime_error("synth: cannot write " + base + ".truth");
    }
    if (summary) {
        summary->stations = n_stn;
        summary->baselines = n_bl;
        summary->measurement_rows = 3 * n_bl + (sp.y_cluster ? 12 : 0);
        summary->blocks = B;
        summary->max_block_unknowns = 0;
        for (uint32_t k = 0; k < B; ++k)
            summary->max_block_unknowns = std::max<uint64_t>(summary->max_block_unknowns, 3 * (seg.ISL[k].size() + seg.JSL[k].size()));
    }
}

}  // namespace synth
}  // namespace dynadjust
