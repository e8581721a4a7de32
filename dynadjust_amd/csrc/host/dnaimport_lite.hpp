// DNA text station / measurement files -> .bst / .bms / .asl, with the reference-frame alignment of GNSS measurements:
// the two stages that stand between the reference's sample data (sampleData/*.stn, *.msr) and the adjustment path.
//
// Not a rebuild of dnaimport / dnareftran (XML, SINEX, geoid, discontinuities, renaming, ... stay out of scope):
// the DNA v3 fixed-column text formats for stations (LLH / LLh / XYZ) and GNSS measurements (G baselines, X baseline clusters,
// Y point clusters), laid out in the binary records the way dnaimport does (include/measurement_types/dnagpsbaseline.cpp
// WriteBinaryMsr: one record per X / Y / Z element, covariance records after each vector; read back by dnaadjust.cpp:4214-4560),
// and the Helmert transformation dnareftran applies to GNSS measurements whose frame / epoch differs from the stations'
// (dnareftran.cpp:1740-1835 TransformMeasurement_GX: both ends of a baseline are transformed as points and differenced;
// dnareftran.cpp:1033-1143: direct 14-parameter set, rates applied over t - t0, t = the measurement's epoch as a decimal year).
#pragma once
#include <string>
#include <vector>

#include "dnatypes.hpp"

namespace dynadjust {
namespace import {

struct import_summary {
    size_t stations = 0, records = 0, vectors = 0, clusters = 0, vectors_transformed = 0;
    std::string station_frame, station_epoch;
};

// Published 14-parameter sets (mm, ppb, mas and their rates per year; reference epoch) towards GDA2020, the frame the sample
// data is adjusted in: ICSM GDA2020 technical manual / IERS ITRF2014 transformation tables, as tabulated in the reference's
// include/parameters/dnatransformationparameters.hpp.  Returns false for a frame without an entry.
bool helmert_to_gda2020(const std::string& frame, double params14[14], double* reference_epoch);

// decimal year of "dd.mm.yyyy": year + (day of year - 0.5) / days in year (dnatemplatedatetimefuncs.hpp:292-328)
double decimal_year(const std::string& ddmmyyyy);

// xyz (metres, `frame` at `epoch`) -> GDA2020; position vector form  x' = (1 + s) R x + T  with the rates applied over
// (epoch - reference epoch) (ReduceParameters / Transform_7parameter, dnatemplatematrixfuncs.hpp:729-806)
void transform_point_to_gda2020(const double params14[14], double reference_epoch, double epoch, const double in[3], double out[3]);

// MGA / UTM grid coordinates -> latitude, longitude (radians): Krueger series on GRS80 (GDA technical manual), southern hemisphere
void utm_to_geographic(double easting, double northing, int zone, double* lat, double* lon);

// Throws std::runtime_error on malformed input, unknown stations, unsupported measurement types or frames.
//   stations      LLH / LLh (ddd.mmssss), XYZ, UTM (easting northing height zone)
//   measurements  GNSS: G, X, Y (cartesian, or LLH / LLh point clusters); terrestrial: A (horizontal angle), B / K (geodetic / astronomic
//                 azimuth), C / E / M (chord, ellipsoid arc, MSL arc), D (direction sets), S (slope distance), V / Z (zenith distance,
//                 vertical angle), L (height difference), H / R (orthometric / ellipsoidal height), I / J (astronomic latitude / longitude),
//                 P / Q (geodetic latitude / longitude) -- angles d m s with standard deviations in seconds, lengths in metres; S, V, Z
//                 carry instrument and target heights (dnaimport: term1 value, term2 variance, term3 / term4 heights)
//   geo_file      optional DNA geoid file (dnageoid's export: station, N [m], deflections in the meridian / prime vertical [seconds]):
//                 geoid separation and deflections into the station records, orthometric station heights (LLH, UTM) to ellipsoidal
void import_dna_text(const std::string& stn_file, const std::string& msr_file, const std::string& out_base, import_summary* summary = nullptr,
                     const std::string& geo_file = std::string());

}  // namespace import
}  // namespace dynadjust
