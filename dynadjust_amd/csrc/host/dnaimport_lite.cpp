// DNA text files -> binary station / measurement files, GNSS measurements aligned to the stations' frame (dnaimport_lite.hpp).
#include "dnaimport_lite.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "dnaio.hpp"
#include "geodesy.hpp"

namespace dynadjust {
namespace import {

namespace {

constexpr double PI = 3.14159265358979323846;

std::string trim(const std::string& s) {
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    size_t e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
}

std::string field(const std::string& line, size_t pos, size_t len) { return pos < line.size() ? trim(line.substr(pos, len)) : std::string(); }

// numbers of a fixed-column line: the columns may touch ("12647.1455-1.0467927495000e-05"), so a sign that does not follow an
// exponent letter starts a new number
std::vector<double> numbers(const std::string& s) {
    std::vector<double> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && !(isdigit((unsigned char)s[i]) || s[i] == '-' || s[i] == '+' || s[i] == '.')) ++i;
        if (i >= s.size()) break;
        size_t j = i + 1;
        while (j < s.size()) {
            const char c = s[j];
            if (isdigit((unsigned char)c) || c == '.') {
                ++j;
            } else if ((c == 'e' || c == 'E') && j + 1 < s.size() && (isdigit((unsigned char)s[j + 1]) || s[j + 1] == '-' || s[j + 1] == '+')) {
                j += 2;
            } else {
                break;
            }
        }
        const std::string tok = s.substr(i, j - i);
        if (tok.find_first_of("0123456789") != std::string::npos) out.push_back(atof(tok.c_str()));
        i = j;
    }
    return out;
}

// ddd.mmssssss (the DNA notation of angles) -> radians
double dms_to_radians(double v) {
    const double sgn = v < 0.0 ? -1.0 : 1.0;
    v = std::fabs(v);
    const double d = std::floor(v + 1e-12);
    const double m = std::floor((v - d) * 100.0 + 1e-9);
    const double sec = ((v - d) * 100.0 - m) * 100.0;
    return sgn * (d + m / 60.0 + sec / 3600.0) * PI / 180.0;
}

void put(char* dst, size_t cap, const std::string& s) {
    memset(dst, 0, cap);
    memcpy(dst, s.data(), std::min(cap - 1, s.size()));
}

const std::map<std::string, std::string>& epsg_codes() {
    // include/parameters/dnaepsg.hpp: geocentric codes of the frames the sample data uses
    static const std::map<std::string, std::string> m = {{"GDA2020", "7843"}, {"GDA94", "4939"},    {"ITRF2020", "9988"}, {"ITRF2014", "7789"},
                                                         {"ITRF2008", "5332"}, {"ITRF2005", "4896"}, {"ITRF2000", "4919"}, {"ITRF1997", "4918"}};
    return m;
}

bool comment_or_blank(const std::string& l) { return l.empty() || l[0] == '*' || l.compare(0, 3, "!#=") == 0 || trim(l).empty(); }

}  // namespace

void utm_to_geographic(double easting, double northing, int zone, double* lat, double* lon) {
    const double a = geodesy::GRS80_A, f = 1.0 / geodesy::GRS80_INV_F, k0 = 0.9996, fe = 500000.0, fn = 10000000.0;
    const double n = f / (2.0 - f), n2 = n * n, n3 = n2 * n, n4 = n3 * n;
    const double A = a / (1.0 + n) * (1.0 + n2 / 4.0 + n4 / 64.0 + n4 * n2 / 256.0);
    const double beta[4] = {n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360, n2 / 48 + n3 / 15 - 437 * n4 / 1440, 17 * n3 / 480 - 37 * n4 / 840,
                            4397 * n4 / 161280};
    const double delta[4] = {2 * n - 2 * n2 / 3 - 2 * n3 + 116 * n4 / 45, 7 * n2 / 3 - 8 * n3 / 5 - 227 * n4 / 45, 56 * n3 / 15 - 136 * n4 / 35,
                             4279 * n4 / 630};
    const double xi = (northing - fn) / (k0 * A), eta = (easting - fe) / (k0 * A);
    double xi1 = xi, eta1 = eta;
    for (int j = 0; j < 4; ++j) {
        xi1 -= beta[j] * std::sin(2 * (j + 1) * xi) * std::cosh(2 * (j + 1) * eta);
        eta1 -= beta[j] * std::cos(2 * (j + 1) * xi) * std::sinh(2 * (j + 1) * eta);
    }
    const double chi = std::asin(std::sin(xi1) / std::cosh(eta1));
    double phi = chi;
    for (int j = 0; j < 4; ++j) phi += delta[j] * std::sin(2 * (j + 1) * chi);
    *lat = phi;
    *lon = (zone * 6 - 183) * PI / 180.0 + std::atan2(std::sinh(eta1), std::cos(xi1));
}

bool helmert_to_gda2020(const std::string& frame, double p[14], double* reference_epoch) {
    struct set_t {
        const char* frame;
        double v[14];
    };
    // tx ty tz (mm)  scale (ppb)  rx ry rz (mas)  then their rates per year; all with reference epoch 2020.0
    static const set_t sets[] = {
        {"ITRF2014", {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.50379, 1.18346, 1.20716}},
        {"ITRF2020", {-1.4, -1.4, 2.4, -0.42, 0, 0, 0, 0.0, -0.1, 0.2, 0.0, 1.50379, 1.18346, 1.20716}},
        {"ITRF2008", {13.790, 4.550, 15.220, 2.5500, 0.2808, 0.2677, -0.4638, 1.420, 1.340, 0.900, 0.1090, 1.5461, 1.1820, 1.1551}},
        {"ITRF2005", {40.320, -33.850, -16.720, 4.2860, -1.2893, -0.8492, -0.3342, 2.250, -0.620, -0.560, 0.2940, 1.4707, 1.1443, 1.1701}},
        {"ITRF2000", {-105.520, 51.580, 231.680, 3.5500, 4.2175, 6.3941, 0.8617, -4.660, 3.550, 11.240, 0.2490, 1.7454, 1.4868, 1.2240}},
        {"ITRF1997", {-176.680, -29.130, 226.990, -3.1170, 1.3427, 6.1880, 3.9809, -8.600, 0.360, 11.250, 0.0070, 1.6394, 1.5198, 1.3801}},
    };
    for (const set_t& s : sets)
        if (frame == s.frame) {
            memcpy(p, s.v, sizeof(s.v));
            if (reference_epoch) *reference_epoch = 2020.0;
            return true;
        }
    return false;
}

double decimal_year(const std::string& ddmmyyyy) {
    int d = 0, m = 0, y = 0;
    if (sscanf(ddmmyyyy.c_str(), "%d.%d.%d", &d, &m, &y) != 3 || m < 1 || m > 12 || d < 1 || d > 31)
        throw std::runtime_error("cannot parse the epoch \"" + ddmmyyyy + "\" (dd.mm.yyyy)");
    const bool leap = (y % 400 == 0) || (y % 100 != 0 && y % 4 == 0);
    static const int cum[2][12] = {{0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334}, {0, 31, 60, 91, 121, 152, 182, 213, 244, 274, 305, 335}};
    const double doy = cum[leap ? 1 : 0][m - 1] + d;
    return (double)y + (doy - 0.5) / (leap ? 366.0 : 365.0);
}

void transform_point_to_gda2020(const double p[14], double reference_epoch, double epoch, const double in[3], double out[3]) {
    const double dt = epoch - reference_epoch;
    const double mas = PI / 180.0 / 3600.0 / 1000.0;
    const double t[3] = {(p[0] + p[7] * dt) / 1000.0, (p[1] + p[8] * dt) / 1000.0, (p[2] + p[9] * dt) / 1000.0};
    const double s = (p[3] + p[10] * dt) / 1.0e9;
    const double rx = (p[4] + p[11] * dt) * mas, ry = (p[5] + p[12] * dt) * mas, rz = (p[6] + p[13] * dt) * mas;
    const double R[3][3] = {{1.0, rz, -ry}, {-rz, 1.0, rx}, {ry, -rx, 1.0}};
    for (int i = 0; i < 3; ++i) {
        double v = 0.0;
        for (int j = 0; j < 3; ++j) v += R[i][j] * in[j];
        out[i] = v * (1.0 + s) + t[i];
    }
}

void import_dna_text(const std::string& stn_file, const std::string& msr_file, const std::string& out_base, import_summary* summary,
                     const std::string& geo_file) {
    import_summary sum;
    // ---- stations ---------------------------------------------------------------------------------------------------------
    std::ifstream sf(stn_file);
    if (!sf) throw std::runtime_error("cannot open " + stn_file);
    std::vector<station_t> stations;
    std::map<std::string, UINT32> index;
    std::string line;
    sum.station_frame = "GDA2020";
    sum.station_epoch = "01.01.2020";
    while (std::getline(sf, line)) {
        if (line.compare(0, 3, "!#=") == 0) {
            // !#=DNA 3.01 STN    <date>   <frame>   <epoch>   <count>
            std::istringstream hs(line.substr(3));
            std::string dna, ver, kind, date, frame, epoch;
            hs >> dna >> ver >> kind >> date >> frame >> epoch;
            if (!frame.empty()) sum.station_frame = frame;
            if (!epoch.empty()) sum.station_epoch = epoch;
            continue;
        }
        if (comment_or_blank(line)) continue;
        station_t st;
        memset(&st, 0, sizeof(st));
        const std::string name = field(line, 0, 20), con = field(line, 20, 3), type = field(line, 24, 3);
        if (name.empty() || con.size() != 3) throw std::runtime_error(stn_file + ": malformed station line: " + line);
        // (a UTM record carries its zone behind the three coordinate columns)
        const std::vector<double> v = numbers(line.size() > 27 ? (type == "UTM" ? line.substr(27) : line.substr(27, 60)) : std::string());
        if (v.size() < 3) throw std::runtime_error(stn_file + ": station " + name + " has no coordinates");
        double lat, lon, h;
        if (type == "LLH" || type == "LLh") {
            lat = dms_to_radians(v[0]);
            lon = dms_to_radians(v[1]);
            h = v[2];
            st.suppliedStationType = type == "LLH" ? LLH_type_i : LLh_type_i;
            st.suppliedHeightRefFrame = type == "LLH" ? ORTHOMETRIC_type_i : ELLIPSOIDAL_type_i;
        } else if (type == "XYZ") {
            geodesy::CartToGeo(v[0], v[1], v[2], &lat, &lon, &h);
            st.suppliedStationType = XYZ_type_i;
            st.suppliedHeightRefFrame = ELLIPSOIDAL_type_i;
        } else if (type == "UTM") {
            if (v.size() < 4) throw std::runtime_error(stn_file + ": UTM station " + name + " has no zone");
            utm_to_geographic(v[0], v[1], (int)v[3], &lat, &lon);
            h = v[2];
            st.suppliedStationType = UTM_type_i;
            st.suppliedHeightRefFrame = ORTHOMETRIC_type_i;
            st.zone = (short)v[3];
        } else {
            throw std::runtime_error(stn_file + ": station coordinate type '" + type + "' is not supported by this importer (LLH, LLh, XYZ, UTM)");
        }
        put(st.stationName, sizeof(st.stationName), name);
        put(st.stationNameOrig, sizeof(st.stationNameOrig), name);
        put(st.stationConst, sizeof(st.stationConst), con);
        put(st.stationType, sizeof(st.stationType), type == "UTM" ? "UTM" : "LLH");
        st.initialLatitude = st.currentLatitude = lat;
        st.initialLongitude = st.currentLongitude = lon;
        st.initialHeight = st.currentHeight = h;
        {
            std::string desc = line.size() > 87 ? trim(line.substr(87)) : std::string();
            if (type == "UTM") {                     // "<zone> <description>"
                size_t sp = desc.find(' ');
                desc = sp == std::string::npos ? std::string() : trim(desc.substr(sp));
            }
            put(st.description, sizeof(st.description), desc);
        }
        st.fileOrder = st.nameOrder = (UINT32)stations.size();
        auto code = epsg_codes().find(sum.station_frame);
        put(st.epsgCode, sizeof(st.epsgCode), code == epsg_codes().end() ? std::string("7843") : code->second);
        put(st.epoch, sizeof(st.epoch), sum.station_epoch);
        if (!index.emplace(name, (UINT32)stations.size()).second) throw std::runtime_error(stn_file + ": station " + name + " appears twice");
        stations.push_back(st);
    }
    if (stations.empty()) throw std::runtime_error(stn_file + ": no stations");
    if (!geo_file.empty()) {
        // dnageoid (dnageoid.cpp:815 writes this file): N and the deflections per station; orthometric heights become ellipsoidal
        std::ifstream gf(geo_file);
        if (!gf) throw std::runtime_error("cannot open " + geo_file);
        const double sec = PI / 648000.0;
        while (std::getline(gf, line)) {
            if (line.empty() || line[0] == '#' || line[0] == '!' || line[0] == '*' || trim(line).empty()) continue;
            const std::string name = field(line, 0, 41);
            const std::vector<double> v = numbers(line.size() > 41 ? line.substr(41) : std::string());
            auto it = index.find(name);
            if (it == index.end() || v.size() < 3) continue;
            station_t& st = stations[it->second];
            st.geoidSep = (float)v[0];
            st.meridianDef = v[1] * sec;
            st.verticalDef = v[2] * sec;
            if (st.suppliedHeightRefFrame == ORTHOMETRIC_type_i) {
                st.initialHeight += v[0];
                st.currentHeight += v[0];
            }
        }
    }
    // ---- measurements -----------------------------------------------------------------------------------------------------
    std::ifstream mf(msr_file);
    if (!mf) throw std::runtime_error("cannot open " + msr_file);
    std::vector<std::string> lines;
    std::string msr_frame = sum.station_frame, msr_epoch = sum.station_epoch;
    while (std::getline(mf, line)) {
        if (line.compare(0, 3, "!#=") == 0) {
            std::istringstream hs(line.substr(3));
            std::string dna, ver, kind, date, frame, epoch;
            hs >> dna >> ver >> kind >> date >> frame >> epoch;
            if (!frame.empty()) msr_frame = frame;
            if (!epoch.empty()) msr_epoch = epoch;
            continue;
        }
        if (comment_or_blank(line)) continue;
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        lines.push_back(line);
    }
    auto station_of = [&](const std::string& nm) {
        auto it = index.find(nm);
        if (it == index.end()) throw std::runtime_error(msr_file + ": station " + nm + " is not in the station file");
        return it->second;
    };
    std::vector<measurement_t> recs;
    std::vector<UINT32> counts(stations.size(), 0);
    UINT32 cluster_id = 0;
    size_t i = 0;
    while (i < lines.size()) {
        const std::string& head = lines[i];
        const char t = head[0];
        const bool ignore = head.size() > 1 && head[1] == '*';
        auto epsg_of_stations = [&]() {
            auto code = epsg_codes().find(sum.station_frame);
            return code == epsg_codes().end() ? std::string("7843") : code->second;
        };
        // "ddd mm ss.sss  sd(seconds)" after the station columns -> radians, radians
        auto angle_and_sd = [&](const std::string& rec, double* value, double* sd) -> size_t {
            std::istringstream ts(rec.size() > 62 ? rec.substr(62) : std::string());
            std::vector<std::string> tok;
            std::string w;
            while (ts >> w) tok.push_back(w);
            if (tok.size() < 4) throw std::runtime_error(msr_file + ": malformed angular measurement: " + rec);
            const double sgn = tok[0][0] == '-' ? -1.0 : 1.0;
            *value = sgn * (std::fabs(atof(tok[0].c_str())) + atof(tok[1].c_str()) / 60.0 + atof(tok[2].c_str()) / 3600.0) * PI / 180.0;
            *sd = atof(tok[3].c_str()) * PI / 648000.0;
            return tok.size();
        };
        if (t == 'D') {
            // a set of directions (dnainterop.cpp:3060-3160 ParseDNAMSRDirections): the first line carries the instrument, the reference
            // direction's target and the number of further directions, one line per further direction follows with its target in the
            // third station column.  Binary (read back by dnaadjust.cpp:5100-5160): one record per direction, the first with
            // vectorCount1 = all directions of the set, vectorCount2 = those not ignored (each counting the first one)
            const UINT32 inst = station_of(field(head, 2, 20)), ro = station_of(field(head, 22, 20));
            const std::vector<double> c = numbers(field(head, 42, 20));
            if (c.empty() || c[0] < 1) throw std::runtime_error(msr_file + ": direction set without a direction count: " + head);
            const UINT32 k = (UINT32)c[0];
            if (i + k >= lines.size()) throw std::runtime_error(msr_file + ": direction set cut short: " + head);
            ++cluster_id;
            const size_t first = recs.size();
            UINT32 live = 1;
            for (UINT32 d = 0; d <= k; ++d) {
                const std::string& rec = lines[i + d];
                if (d && rec[0] != 'D' && rec[0] != ' ') throw std::runtime_error(msr_file + ": expected a direction of the set: " + rec);
                const bool sub_ignore = d ? (rec.size() > 1 && rec[1] == '*') : false;
                double value = 0.0, sd = 0.0;
                angle_and_sd(rec, &value, &sd);
                measurement_t m;
                memset(&m, 0, sizeof(m));
                m.measType = 'D';
                m.measStart = d ? 1 : 0;
                m.measurementStations = 2;
                put(m.epsgCode, sizeof(m.epsgCode), epsg_of_stations());
                put(m.epoch, sizeof(m.epoch), sum.station_epoch);
                put(m.coordType, sizeof(m.coordType), "XYZ");
                m.ignore = d ? sub_ignore : ignore;
                m.station1 = inst;
                m.station2 = d ? station_of(field(rec, 42, 20)) : ro;
                m.clusterID = m.fileOrder = cluster_id;
                m.scale1 = m.scale2 = m.scale3 = m.scale4 = 1.0;
                m.term1 = m.preAdjMeas = value;
                m.term2 = sd * sd;
                recs.push_back(m);
                if (d && !sub_ignore) ++live;
                if (!ignore) counts[m.station2]++;
            }
            if (live < 2 && !ignore) throw std::runtime_error(msr_file + ": a direction set without a second direction that is not ignored: " + head);
            recs[first].vectorCount1 = k + 1;
            recs[first].vectorCount2 = live;
            if (!ignore) counts[inst]++;
            sum.clusters++;
            i += k + 1;
            continue;
        }
        if (t != 'G' && t != 'X' && t != 'Y') {
            // one-line terrestrial measurement (dnaimport: term1 = value, term2 = variance, term3 / term4 = instrument / target height);
            // I / J (astronomic latitude / longitude) and P / Q (geodetic) are angles at one station
            static const std::string supported = "ABCEHIJKLMPQRSVZ", angular = "ABIJKPQVZ", with_heights = "SVZ";
            if (supported.find(t) == std::string::npos)
                throw std::runtime_error(msr_file + ": measurement type '" + std::string(1, t) + "' is not supported by this importer");
            std::vector<std::string> names = {field(head, 2, 20), field(head, 22, 20), field(head, 42, 20)};
            std::vector<UINT32> ids;
            for (const std::string& nm : names)
                if (!nm.empty()) ids.push_back(station_of(nm));
            std::istringstream ts(head.size() > 62 ? head.substr(62) : std::string());
            std::vector<std::string> tok;
            std::string w;
            while (ts >> w) tok.push_back(w);
            double value = 0.0, sd = 0.0;
            size_t used = 0;
            const double sec = PI / 648000.0;
            if (angular.find(t) != std::string::npos) {
                if (tok.size() < 4) throw std::runtime_error(msr_file + ": malformed angular measurement: " + head);
                const double sgn = tok[0][0] == '-' ? -1.0 : 1.0;
                value = sgn * (std::fabs(atof(tok[0].c_str())) + atof(tok[1].c_str()) / 60.0 + atof(tok[2].c_str()) / 3600.0) * PI / 180.0;
                sd = atof(tok[3].c_str()) * sec;
                used = 4;
            } else {
                if (tok.size() < 2) throw std::runtime_error(msr_file + ": malformed measurement: " + head);
                value = atof(tok[0].c_str());
                sd = atof(tok[1].c_str());
                used = 2;
            }
            const size_t need = (t == 'A') ? 3 : (std::string("HRIJPQ").find(t) != std::string::npos ? 1 : 2);
            if (ids.size() != need) throw std::runtime_error(msr_file + ": wrong number of stations: " + head);
            measurement_t m;
            memset(&m, 0, sizeof(m));
            m.measType = t;
            m.measurementStations = (char)ids.size();
            auto code = epsg_codes().find(sum.station_frame);
            put(m.epsgCode, sizeof(m.epsgCode), code == epsg_codes().end() ? std::string("7843") : code->second);
            put(m.epoch, sizeof(m.epoch), sum.station_epoch);
            put(m.coordType, sizeof(m.coordType), "XYZ");
            m.ignore = ignore;
            m.station1 = ids[0];
            m.station2 = ids.size() > 1 ? ids[1] : 0;
            m.station3 = ids.size() > 2 ? ids[2] : 0;
            m.clusterID = m.fileOrder = ++cluster_id;
            m.scale1 = m.scale2 = m.scale3 = m.scale4 = 1.0;
            m.term1 = m.preAdjMeas = value;
            m.term2 = sd * sd;
            if (with_heights.find(t) != std::string::npos && tok.size() >= used + 2) {
                m.term3 = atof(tok[used].c_str());
                m.term4 = atof(tok[used + 1].c_str());
            }
            recs.push_back(m);
            if (!ignore)
                for (UINT32 id : ids) counts[id]++;
            sum.clusters++;
            ++i;
            continue;
        }
        const bool point = t == 'Y';
        const std::string coord_type = point ? field(head, 22, 20) : std::string("XYZ");
        UINT32 k = 1;
        if (t != 'G') {
            const std::vector<double> c = numbers(field(head, 42, 20));
            if (c.empty() || c[0] < 1) throw std::runtime_error(msr_file + ": cluster without a vector count: " + head);
            k = (UINT32)c[0];
        }
        // scalars (matrix, phi, lambda, height), frame and epoch follow the count columns
        double vs = 1.0, ps = 1.0, ls = 1.0, hs_ = 1.0;
        std::string frame = msr_frame, epoch = msr_epoch;
        if (head.size() > 62) {
            std::istringstream ts(head.substr(62));
            std::vector<std::string> tok;
            std::string w;
            while (ts >> w) tok.push_back(w);
            size_t nnum = 0;
            while (nnum < tok.size() && nnum < 4 && (isdigit((unsigned char)tok[nnum][0]) || tok[nnum][0] == '.' || tok[nnum][0] == '-')) ++nnum;
            if (nnum > 0) vs = atof(tok[0].c_str());
            if (nnum > 1) ps = atof(tok[1].c_str());
            if (nnum > 2) ls = atof(tok[2].c_str());
            if (nnum > 3) hs_ = atof(tok[3].c_str());
            if (tok.size() > nnum) frame = tok[nnum];
            if (tok.size() > nnum + 1) epoch = tok[nnum + 1];
        }
        // alignment with the stations' frame (TransformMeasurement_GX / _Y)
        double hp[14], t0 = 0.0;
        bool transform = false;
        if (frame != sum.station_frame) {
            if (sum.station_frame != "GDA2020" || !helmert_to_gda2020(frame, hp, &t0))
                throw std::runtime_error(msr_file + ": no transformation from " + frame + " to " + sum.station_frame + " in this importer");
            transform = true;
        }
        const double t_obs = transform ? decimal_year(epoch) : 0.0;
        ++cluster_id;
        auto code = epsg_codes().find(sum.station_frame);
        const std::string epsg = code == epsg_codes().end() ? std::string("7843") : code->second;
        for (UINT32 j = 0; j < k; ++j) {
            if (i >= lines.size()) throw std::runtime_error(msr_file + ": truncated cluster");
            const std::string& h = lines[i];
            if (h[0] != t) throw std::runtime_error(msr_file + ": cluster of type " + std::string(1, t) + " interrupted: " + h);
            const UINT32 s1 = station_of(field(h, 2, 20));
            const UINT32 s2 = point ? 0u : station_of(field(h, 22, 20));
            double obs[3], V[3][3] = {{0}};
            for (int r = 0; r < 3; ++r) {
                if (i + 1 + r >= lines.size()) throw std::runtime_error(msr_file + ": truncated measurement");
                const std::string& rl = lines[i + 1 + r];
                const std::vector<double> v = numbers(rl.size() > 62 ? rl.substr(62) : std::string());
                if (v.size() != (size_t)(2 + r)) throw std::runtime_error(msr_file + ": malformed measurement row: " + rl);
                obs[r] = v[0];
                for (int c = 0; c <= r; ++c) V[c][r] = V[r][c] = v[1 + c];
            }
            i += 4;
            if (point && (coord_type == "LLH" || coord_type == "LLh")) {     // latitude / longitude as ddd.mmssss
                obs[0] = dms_to_radians(obs[0]);
                obs[1] = dms_to_radians(obs[1]);
            }
            if (transform && !ignore) {
                if (point) {
                    if (coord_type != "XYZ") throw std::runtime_error(msr_file + ": only cartesian point clusters can be transformed by this importer");
                    double o[3];
                    transform_point_to_gda2020(hp, t0, t_obs, obs, o);
                    memcpy(obs, o, sizeof(o));
                } else {
                    // both ends as points -- end 1 from the station file, end 2 = end 1 + vector -- then the difference
                    const station_t& a = stations[s1];
                    double x1[3], x2[3], y1[3], y2[3];
                    geodesy::GeoToCart(a.currentLatitude, a.currentLongitude, a.currentHeight, &x1[0], &x1[1], &x1[2]);
                    for (int r = 0; r < 3; ++r) x2[r] = x1[r] + obs[r];
                    transform_point_to_gda2020(hp, t0, t_obs, x1, y1);
                    transform_point_to_gda2020(hp, t0, t_obs, x2, y2);
                    for (int r = 0; r < 3; ++r) obs[r] = y2[r] - y1[r];
                }
                sum.vectors_transformed++;
            }
            auto base_record = [&](char start) {
                measurement_t m;
                memset(&m, 0, sizeof(m));
                m.measType = t;
                m.measStart = start;
                m.measurementStations = point ? 1 : 2;
                put(m.epsgCode, sizeof(m.epsgCode), epsg);
                put(m.epoch, sizeof(m.epoch), sum.station_epoch);
                put(m.observation_epoch, sizeof(m.observation_epoch), epoch);
                put(m.coordType, sizeof(m.coordType), coord_type.substr(0, 3));
                m.ignore = ignore;
                m.station1 = s1;
                m.station2 = s2;
                m.vectorCount1 = t == 'G' ? 1 : k;
                m.vectorCount2 = t == 'G' ? 0 : k - 1 - j;
                m.clusterID = cluster_id;
                m.fileOrder = cluster_id;
                m.scale1 = ps;
                m.scale2 = ls;
                m.scale3 = hs_;
                m.scale4 = vs;
                return m;
            };
            for (int e = 0; e < 3; ++e) {
                measurement_t m = base_record((char)e);
                m.term1 = m.preAdjMeas = obs[e];
                m.term2 = V[0][e];
                if (e >= 1) m.term3 = V[1][e];
                if (e == 2) m.term4 = V[2][2];
                recs.push_back(m);
            }
            // covariances with the later vectors of the cluster: three rows of three per vector
            for (UINT32 c = j + 1; c < k; ++c) {
                for (int e = 0; e < 3; ++e) {
                    if (i >= lines.size()) throw std::runtime_error(msr_file + ": truncated covariance block");
                    const std::vector<double> v = numbers(lines[i]);
                    if (v.size() != 3) throw std::runtime_error(msr_file + ": malformed covariance row: " + lines[i]);
                    measurement_t m = base_record((char)(3 + e));
                    m.term1 = v[0];
                    m.term2 = v[1];
                    m.term3 = v[2];
                    recs.push_back(m);
                    ++i;
                }
            }
            if (!ignore) {
                counts[s1]++;
                if (!point) counts[s2]++;
            }
            sum.vectors++;
        }
        sum.clusters++;
    }
    // ---- out ----------------------------------------------------------------------------------------------------------------
    binary_file_meta_t meta;
    auto code = epsg_codes().find(sum.station_frame);
    put(meta.epsgCode, sizeof(meta.epsgCode), code == epsg_codes().end() ? std::string("7843") : code->second);
    put(meta.epoch, sizeof(meta.epoch), sum.station_epoch);
    meta.reftran = sum.vectors_transformed > 0;
    iostreams::write_bst(out_base + ".bst", stations, meta, "dnaimport");
    iostreams::write_bms(out_base + ".bms", recs, meta, "dnaimport");
    std::vector<asl_entry_t> asl(stations.size());
    UINT32 off = 0;
    for (size_t s = 0; s < stations.size(); ++s) {
        asl[s].assocMsrCount = counts[s];
        asl[s].amlStnIndex = off;
        asl[s].validity = 1;
        off += counts[s];
    }
    iostreams::write_asl(out_base + ".asl", asl, "dnaimport");
    sum.stations = stations.size();
    sum.records = recs.size();
    if (summary) *summary = sum;
}

}  // namespace import
}  // namespace dynadjust
