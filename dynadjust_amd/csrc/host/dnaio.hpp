// Readers / writers of the binary station (.bst), binary measurement (.bms), associated
// station list (.asl) and segmentation (.seg) files -- the on-disk boundary of the
// adjustment path.  Formats follow
//   include/io/dynadjust_file.cpp:75-283   60-byte file info + metadata
//   include/io/bst_file.cpp, bms_file.cpp  raw station_t / measurement_t dumps
//   include/io/asl_file.cpp:75-100         u64 count + {u32, u32, u16} records
//   include/io/seg_file.cpp:57-408, 489-721  fixed-column text
// of /root/reference/dynadjust/.  All functions throw std::runtime_error on failure.
#pragma once
#include <string>
#include <vector>
#include "dnatypes.hpp"

namespace dynadjust {
namespace iostreams {

constexpr const char* FILE_VERSION = "1.2";

struct file_info_t {
    std::string version, date, app;
};

void read_bst(const std::string& path, std::vector<station_t>& stations, binary_file_meta_t& meta, file_info_t* info = nullptr);
void write_bst(const std::string& path, const std::vector<station_t>& stations, binary_file_meta_t meta,
               const std::string& app_name = "DNAGPU");
void read_bms(const std::string& path, std::vector<measurement_t>& msrs, binary_file_meta_t& meta, file_info_t* info = nullptr);
void write_bms(const std::string& path, const std::vector<measurement_t>& msrs, binary_file_meta_t meta,
               const std::string& app_name = "DNAGPU");
void read_asl(const std::string& path, std::vector<asl_entry_t>& asl);
void write_asl(const std::string& path, const std::vector<asl_entry_t>& asl, const std::string& app_name = "DNAGPU");

struct seg_data_t {
    UINT32 blockCount = 0, blockThreshold = 0, minInnerStns = 0;
    std::vector<std::vector<UINT32>> ISL, JSL, CML;
    std::vector<UINT32> ContiguousNetList;       // network id per block
    // metrics filled when the measurement records are supplied (SegFile::LoadSegFile, loadMetrics)
    std::vector<UINT32> measurementCount, unknownsCount, parameterStationCount;
};
// bms may be null (no metrics)
void read_seg(const std::string& path, seg_data_t& seg, const std::vector<measurement_t>* bms);
void write_seg(const std::string& path, const seg_data_t& seg, const std::string& bst_file, const std::string& bms_file,
               const std::vector<measurement_t>& bms);

}  // namespace iostreams
}  // namespace dynadjust
