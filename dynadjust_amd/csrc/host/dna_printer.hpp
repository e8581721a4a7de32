// The seam towards the reference's report printers.
//
// In the reference dna_adjust owns a DynAdjustPrinter (dnaadjust.hpp:277 GetPrinter(); friends at dnaadjust.hpp:214-215) and
// dnaadjustwrapper calls netAdjust->GetPrinter()->Print...() for every report (dnaadjustwrapper.cpp:296-458).  The printers are out
// of this build's scope (formatting code, thousands of lines); what is IN scope is that a caller written against
// the reference's interface compiles, links and gets its results.  This class has the members the wrapper calls, reads the
// adjustment through the same data the reference's printer reads through friend access (station and measurement records, rigorous
// coordinates, rigorous variance matrices, precisions of the adjusted measurements), and writes REDUCED reports: the same
// quantities in plain fixed-width tables, not the reference's page layout.  A maintainer who wants the reference's own reports binds
// the reference's DynAdjustPrinter to the getters used here, or runs `dnaadjust --report-results` on the -rva.mtx / -pam.mtx and
// .bst / .bms files this library writes.
#pragma once
#include <iosfwd>
#include <string>
#include <vector>

namespace dynadjust {
namespace networkadjust {

class dna_adjust;

class DynAdjustPrinter {
public:
    explicit DynAdjustPrinter(dna_adjust& a) : a_(a) {}
    // dnaadjustwrapper.cpp:296 / 312 / 326 / 342 / 358
    void PrintAdjustedNetworkMeasurements();     // -> <net>.<mode>.adj : adjusted measurements, corrections, precisions, N-statistics
    void PrintMeasurementsToStation();           // -> <net>.<mode>.adj : measurements per station
    void PrintAdjustedNetworkStations();         // -> <net>.<mode>.adj and .xyz : adjusted coordinates and their standard deviations (e, n, up)
    void PrintPositionalUncertainty();           // -> <net>.<mode>.apu : 3 x 3 variance matrix per station
    void PrintNetworkStationCorrections();       // -> <net>.<mode>.cor : adjusted minus initial coordinates
    // dnaadjustwrapper.cpp:389-458: the exporters belong to dnaimport's formats (DynaML, DNA, SINEX): not part of this library
    void PrintEstimatedStationCoordinatestoDNAXML(const std::string& file, int type, bool flagUnused = false);
    void PrintEstimatedStationCoordinatestoDNAXML_Y(const std::string& file, int type);
    bool PrintEstimatedStationCoordinatestoSNX(std::string& sinex_file);
    void Close();                                // dna_adjust::CloseOutputFiles

    // what the tables above are made of (also for callers that format their own reports)
    struct station_result {
        unsigned index;              // .bst record
        std::string name, constraint;
        double xyz[3];               // adjusted cartesian coordinates
        double llh[3];               // latitude, longitude (radians), ellipsoidal height
        double var[6];               // xx xy xz yy yz zz of the rigorous variance matrix
        double sd_enu[3];            // standard deviations east, north, up
        bool adjusted;               // false: not in any block (no estimate)
    };
    std::vector<station_result> StationResults();
    std::string ReportFile(const char* extension) const;      // <output_folder>/<network_name>.<simult|phased>.<extension>

private:
    dna_adjust& a_;
    std::ostream& Adj();
    struct files;
    files* f_ = nullptr;
public:
    ~DynAdjustPrinter();
};

}  // namespace networkadjust
}  // namespace dynadjust
