// Reduced reports over the adjustment's results (dna_printer.hpp).  Reference: DynAdjustPrinter (dnaadjustprinter.cpp) through
// dna_adjust::GetPrinter(), dnaadjustwrapper.cpp:296-458.
#include "dna_printer.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>

#include "dna_adjust.hpp"
#include "geodesy.hpp"

namespace dynadjust {
namespace networkadjust {

struct DynAdjustPrinter::files {
    std::ofstream adj;
};

DynAdjustPrinter::~DynAdjustPrinter() { delete f_; }

void DynAdjustPrinter::Close() {
    delete f_;
    f_ = nullptr;
}

std::string DynAdjustPrinter::ReportFile(const char* extension) const {
    const project_settings& p = a_.projectSettings_;
    std::string dir = p.g.output_folder.empty() ? std::string(".") : p.g.output_folder;
    std::string net = p.g.network_name.empty() ? std::string("network") : p.g.network_name;
    return dir + "/" + net + (p.a.adjust_mode == SimultaneousMode ? ".simult." : ".phased.") + extension;
}

std::ostream& DynAdjustPrinter::Adj() {
    if (!f_) {
        f_ = new files();
        f_->adj.open(ReportFile("adj"));
        if (!f_->adj) a_.SignalExceptionAdjustment("PrintAdjustedNetwork...(): cannot open " + ReportFile("adj") + " for writing.", 0);
        f_->adj << "DYNADJUST ADJUSTMENT OUTPUT (reduced tables of libdnagpu: same quantities as dnaadjust's .adj, plain layout)\n"
                << "Network              " << a_.projectSettings_.g.network_name << "\n"
                << "Mode                 " << (a_.projectSettings_.a.adjust_mode == SimultaneousMode ? "simultaneous" : "phased") << "\n"
                << "Iterations           " << a_.CurrentIteration() << "\n"
                << "Measurements         " << a_.GetMeasurementCount() << "\n"
                << "Unknowns             " << a_.GetUnknownsCount() << "\n"
                << "Degrees of freedom   " << a_.GetDegreesOfFreedom() << "\n"
                << std::fixed << std::setprecision(4) << "Chi squared          " << a_.GetChiSquared() << "\n"
                << "Sigma zero           " << a_.GetSigmaZero() << "   (" << a_.GetChiSquaredLowerLimit() << " < . < " << a_.GetChiSquaredUpperLimit() << ")\n"
                << "Potential outliers   " << a_.GetPotentialOutlierCount() << "\n\n";
    }
    return f_->adj;
}

std::vector<DynAdjustPrinter::station_result> DynAdjustPrinter::StationResults() {
    const std::vector<station_t>& bst = a_.bstBinaryRecords_;
    std::vector<station_result> out(bst.size());
    for (size_t s = 0; s < bst.size(); ++s) {
        station_result& r = out[s];
        r.index = (unsigned)s;
        r.name = std::string(bst[s].stationName, strnlen(bst[s].stationName, sizeof(bst[s].stationName)));
        r.constraint = std::string(bst[s].stationConst, strnlen(bst[s].stationConst, sizeof(bst[s].stationConst)));
        r.adjusted = false;
        std::fill(r.var, r.var + 6, 0.0);
        std::fill(r.sd_enu, r.sd_enu + 3, 0.0);
        r.llh[0] = bst[s].currentLatitude; r.llh[1] = bst[s].currentLongitude; r.llh[2] = bst[s].currentHeight;
        geodesy::GeoToCart(r.llh[0], r.llh[1], r.llh[2], &r.xyz[0], &r.xyz[1], &r.xyz[2]);
    }
    // a station's rigorous estimate and variances are those of the block in which it is an inner station (ISL): every station is
    // inner to exactly one block (dnasegment.cpp:529-531)
    std::vector<double> xyz, packed;
    for (UINT32 b = 0; b < a_.blockCount(); ++b) {
        const std::vector<UINT32>& stations = a_.GetBlockStationList(b);
        a_.GetBlockRigorousStations(b, xyz);
        a_.GetBlockRigorousVariancesPacked(b, packed);
        const size_t n = 3 * stations.size();
        auto at = [&](size_t i, size_t j) {       // packed lower, column-major (matrix_2d::packed_index)
            if (i < j) std::swap(i, j);
            return packed[j * n - j * (j - 1) / 2 - j + i];
        };
        for (UINT32 s : a_.v_ISL_[b]) {
            const size_t l = (size_t)(std::lower_bound(stations.begin(), stations.end(), s) - stations.begin());
            if (l >= stations.size() || stations[l] != s) continue;
            station_result& r = out[s];
            r.adjusted = true;
            for (int c = 0; c < 3; ++c) r.xyz[c] = xyz[3 * l + c];
            geodesy::CartToGeo(r.xyz[0], r.xyz[1], r.xyz[2], &r.llh[0], &r.llh[1], &r.llh[2]);
            double V[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) V[i][j] = at(3 * l + i, 3 * l + j);
            r.var[0] = V[0][0]; r.var[1] = V[0][1]; r.var[2] = V[0][2]; r.var[3] = V[1][1]; r.var[4] = V[1][2]; r.var[5] = V[2][2];
            double R[3][3];                       // columns: e, n, up in cartesian axes
            geodesy::LocalToCartRotation(r.llh[0], r.llh[1], R);
            for (int k = 0; k < 3; ++k) {
                double q = 0.0;
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) q += R[i][k] * V[i][j] * R[j][k];
                r.sd_enu[k] = q > 0.0 ? std::sqrt(q) : 0.0;
            }
        }
    }
    return out;
}

void DynAdjustPrinter::PrintAdjustedNetworkMeasurements() {
    std::ostream& os = Adj();
    const std::vector<measurement_t>& bms = a_.GetMeasurementRecords();
    const std::vector<station_t>& bst = a_.bstBinaryRecords_;
    auto name = [&](UINT32 s) { return s < bst.size() ? std::string(bst[s].stationName, strnlen(bst[s].stationName, sizeof(bst[s].stationName))) : std::string("-"); };
    os << "Adjusted Measurements\n"
       << std::left << std::setw(3) << "M" << std::setw(21) << "Station 1" << std::setw(21) << "Station 2" << std::setw(3) << "*" << std::right << std::setw(18)
       << "Measured" << std::setw(18) << "Adjusted" << std::setw(13) << "Correction" << std::setw(12) << "Meas. SD" << std::setw(12) << "Adj. SD" << std::setw(12)
       << "Corr. SD" << std::setw(10) << "N-stat" << std::setw(10) << "Pelzer" << "\n";
    static const char axis[] = "XYZ";
    for (size_t i = 0; i < bms.size(); ++i) {
        const measurement_t& m = bms[i];
        if (m.ignore || m.measStart > 2) continue;             // (covariance rows of a cluster carry no measurement)
        const bool gnss = m.measType == 'G' || m.measType == 'X' || m.measType == 'Y';
        // the measurement's own variance: a GNSS vector keeps XX in the X row's term2, YY in the Y row's term3, ZZ in the Z row's term4
        // (dnaadjust.cpp:4236-4249); every other type its variance in term2
        const double variance = gnss ? (m.measStart == 0 ? m.term2 : (m.measStart == 1 ? m.term3 : m.term4)) : m.term2;
        os << std::left << std::setw(3) << m.measType << std::setw(21) << name(m.station1) << std::setw(21) << (m.measType == 'Y' || m.measType == 'H' || m.measType == 'R' || m.measType == 'I' || m.measType == 'J' || m.measType == 'P' || m.measType == 'Q' ? std::string("") : name(m.station2))
           << std::setw(3) << (gnss ? std::string(1, axis[(int)m.measStart]) : std::string(" ")) << std::right << std::fixed << std::setprecision(4) << std::setw(18) << m.term1 << std::setw(18) << m.measAdj
           << std::setw(13) << m.measCorr << std::setw(12) << std::sqrt(std::max(0.0, variance))
           << std::setw(12) << (m.measAdjPrec > 0.0 ? std::sqrt(m.measAdjPrec) : 0.0) << std::setw(12) << (m.residualPrec > 0.0 ? std::sqrt(m.residualPrec) : 0.0) << std::setprecision(2)
           << std::setw(10) << m.NStat << std::setw(10) << m.PelzerRel << "\n";
    }
    os << "\n";
}

void DynAdjustPrinter::PrintMeasurementsToStation() {
    std::ostream& os = Adj();
    const std::vector<measurement_t>& bms = a_.GetMeasurementRecords();
    const std::vector<station_t>& bst = a_.bstBinaryRecords_;
    std::vector<unsigned> count(bst.size(), 0);
    for (const measurement_t& m : bms) {
        if (m.ignore || m.measStart != 0) continue;
        if (m.station1 < bst.size()) count[m.station1]++;
        const bool one = m.measType == 'Y' || m.measType == 'H' || m.measType == 'R' || m.measType == 'I' || m.measType == 'J' || m.measType == 'P' || m.measType == 'Q';
        if (!one && m.station2 < bst.size()) count[m.station2]++;
        if (m.measType == 'A' && m.station3 < bst.size()) count[m.station3]++;
    }
    os << "Measurements to Station\n" << std::left << std::setw(21) << "Station" << std::right << std::setw(10) << "Count" << "\n";
    for (size_t s = 0; s < bst.size(); ++s)
        os << std::left << std::setw(21) << std::string(bst[s].stationName, strnlen(bst[s].stationName, sizeof(bst[s].stationName))) << std::right << std::setw(10) << count[s] << "\n";
    os << "\n";
}

void DynAdjustPrinter::PrintAdjustedNetworkStations() {
    const std::vector<station_result> st = StationResults();
    std::ofstream xyz(ReportFile("xyz"));
    auto table = [&](std::ostream& os) {
        os << "Adjusted Coordinates\n"
           << std::left << std::setw(21) << "Station" << std::setw(6) << "Const" << std::right << std::setw(17) << "X" << std::setw(17) << "Y" << std::setw(17) << "Z" << std::setw(11)
           << "SD(e)" << std::setw(11) << "SD(n)" << std::setw(11) << "SD(up)" << "\n";
        for (const station_result& r : st)
            os << std::left << std::setw(21) << r.name << std::setw(6) << r.constraint << std::right << std::fixed << std::setprecision(4) << std::setw(17) << r.xyz[0] << std::setw(17)
               << r.xyz[1] << std::setw(17) << r.xyz[2] << std::setw(11) << r.sd_enu[0] << std::setw(11) << r.sd_enu[1] << std::setw(11) << r.sd_enu[2] << "\n";
        os << "\n";
    };
    table(Adj());
    if (xyz) table(xyz);
}

void DynAdjustPrinter::PrintPositionalUncertainty() {
    std::ofstream os(ReportFile("apu"));
    if (!os) a_.SignalExceptionAdjustment("PrintPositionalUncertainty(): cannot open " + ReportFile("apu") + " for writing.", 0);
    os << "Positional Uncertainty (rigorous variance matrix of every station, cartesian, m^2)\n"
       << std::left << std::setw(21) << "Station" << std::right << std::setw(16) << "XX" << std::setw(16) << "XY" << std::setw(16) << "XZ" << std::setw(16) << "YY" << std::setw(16) << "YZ"
       << std::setw(16) << "ZZ" << "\n";
    for (const station_result& r : StationResults()) {
        os << std::left << std::setw(21) << r.name << std::right << std::scientific << std::setprecision(6);
        for (int k = 0; k < 6; ++k) os << std::setw(16) << r.var[k];
        os << "\n";
    }
}

void DynAdjustPrinter::PrintNetworkStationCorrections() {
    std::ofstream os(ReportFile("cor"));
    if (!os) a_.SignalExceptionAdjustment("PrintNetworkStationCorrections(): cannot open " + ReportFile("cor") + " for writing.", 0);
    const std::vector<station_t>& bst = a_.bstBinaryRecords_;
    os << "Corrections to Stations (adjusted minus initial, local frame, m)\n"
       << std::left << std::setw(21) << "Station" << std::right << std::setw(12) << "east" << std::setw(12) << "north" << std::setw(12) << "up" << "\n";
    for (const station_result& r : StationResults()) {
        double x0, y0, z0, R[3][3];
        geodesy::GeoToCart(bst[r.index].initialLatitude, bst[r.index].initialLongitude, bst[r.index].initialHeight, &x0, &y0, &z0);
        geodesy::LocalToCartRotation(bst[r.index].initialLatitude, bst[r.index].initialLongitude, R);
        const double d[3] = {r.xyz[0] - x0, r.xyz[1] - y0, r.xyz[2] - z0};
        os << std::left << std::setw(21) << r.name << std::right << std::fixed << std::setprecision(4);
        for (int k = 0; k < 3; ++k) os << std::setw(12) << (R[0][k] * d[0] + R[1][k] * d[1] + R[2][k] * d[2]);
        os << "\n";
    }
}

void DynAdjustPrinter::PrintEstimatedStationCoordinatestoDNAXML(const std::string& file, int, bool) {
    a_.SignalExceptionAdjustment("PrintEstimatedStationCoordinatestoDNAXML(" + file + "): the DynaML / DNA exporters are dnaimport's formats and not part of libdnagpu; "
                                 "export from the updated .bst / .bms files (UpdateBinaryFiles) with the reference's tools.", 0);
}
void DynAdjustPrinter::PrintEstimatedStationCoordinatestoDNAXML_Y(const std::string& file, int type) { PrintEstimatedStationCoordinatestoDNAXML(file, type, false); }
bool DynAdjustPrinter::PrintEstimatedStationCoordinatestoSNX(std::string&) { return false; }

}  // namespace networkadjust
}  // namespace dynadjust
