#include "dnaio.hpp"

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <stdexcept>

namespace dynadjust {
namespace iostreams {

namespace {

constexpr int FIELD = 10;  // identifier_field_width, include/io/dynadjust_file.hpp:58

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

void write_field(std::ofstream& f, const char* header, const std::string& value) {
    char buf[FIELD + 1];
    f.write(header, FIELD);
    snprintf(buf, sizeof(buf), "%*s", FIELD, value.substr(0, FIELD).c_str());
    f.write(buf, FIELD);
}

std::string read_field(std::ifstream& f) {
    char buf[FIELD + 1];
    buf[FIELD] = '\0';
    f.read(buf, FIELD);  // field name
    f.read(buf, FIELD);  // value
    return trim(buf);
}

// DynadjustFile::WriteFileInfo, include/io/dynadjust_file.cpp:190-283
void write_file_info(std::ofstream& f, const std::string& app) {
    write_field(f, "VERSION   ", FILE_VERSION);
    char date[32];
    std::time_t t = std::time(nullptr);
    std::tm tmv;
    localtime_r(&t, &tmv);
    std::strftime(date, sizeof(date), "%Y-%m-%d", &tmv);
    write_field(f, "CREATED ON", date);
    write_field(f, "CREATED BY", app);
}

file_info_t read_file_info(std::ifstream& f) {
    file_info_t i;
    i.version = read_field(f);
    i.date = read_field(f);
    i.app = read_field(f);
    return i;
}

bool version_at_least(const std::string& v, int major, int minor) {
    int a = 0, b = 0;
    sscanf(v.c_str(), "%d.%d", &a, &b);
    return a > major || (a == major && b >= minor);
}

// DynadjustFile::WriteFileMetadata, include/io/dynadjust_file.cpp:83-117
void write_meta(std::ofstream& f, const binary_file_meta_t& m) {
    uint64_t cnt = m.binCount;
    f.write(reinterpret_cast<const char*>(&cnt), sizeof(uint64_t));
    f.write(reinterpret_cast<const char*>(&m.reduced), sizeof(bool));
    f.write(m.modifiedBy, MOD_NAME_WIDTH);
    f.write(m.epsgCode, STN_EPSG_WIDTH);
    f.write(m.epoch, STN_EPOCH_WIDTH);
    f.write(m.observation_epoch, STN_EPOCH_WIDTH);
    f.write(reinterpret_cast<const char*>(&m.reftran), sizeof(bool));
    f.write(reinterpret_cast<const char*>(&m.geoid), sizeof(bool));
    uint64_t nin = m.inputFileMeta.size();
    f.write(reinterpret_cast<const char*>(&nin), sizeof(uint64_t));
    for (const input_file_meta_t& i : m.inputFileMeta) {
        f.write(i.filename, FILE_NAME_WIDTH);
        f.write(i.epsgCode, STN_EPSG_WIDTH);
        f.write(i.epoch, STN_EPOCH_WIDTH);
        f.write(i.observation_epoch, STN_EPOCH_WIDTH);
        f.write(reinterpret_cast<const char*>(&i.filetype), sizeof(UINT16));
        f.write(reinterpret_cast<const char*>(&i.datatype), sizeof(UINT16));
    }
    uint64_t nsrc = m.sourceFileMeta.size();
    f.write(reinterpret_cast<const char*>(&nsrc), sizeof(uint64_t));
    for (const source_file_meta_t& s : m.sourceFileMeta) f.write(s.filename, FILE_NAME_WIDTH);
}

// DynadjustFile::ReadFileMetadata, include/io/dynadjust_file.cpp:119-181
void read_meta(std::ifstream& f, binary_file_meta_t& m, const std::string& version) {
    const bool has_obs_epoch = version_at_least(version, 1, 2);
    f.read(reinterpret_cast<char*>(&m.binCount), sizeof(uint64_t));
    f.read(reinterpret_cast<char*>(&m.reduced), sizeof(bool));
    f.read(m.modifiedBy, MOD_NAME_WIDTH);
    f.read(m.epsgCode, STN_EPSG_WIDTH);
    f.read(m.epoch, STN_EPOCH_WIDTH);
    if (has_obs_epoch)
        f.read(m.observation_epoch, STN_EPOCH_WIDTH);
    else
        memcpy(m.observation_epoch, m.epoch, STN_EPOCH_WIDTH);
    f.read(reinterpret_cast<char*>(&m.reftran), sizeof(bool));
    f.read(reinterpret_cast<char*>(&m.geoid), sizeof(bool));
    uint64_t nin = 0;
    f.read(reinterpret_cast<char*>(&nin), sizeof(uint64_t));
    if (!f || nin > (1u << 20)) throw std::runtime_error("corrupt file metadata (input file count)");
    m.inputFileMeta.assign(nin, input_file_meta_t());
    for (input_file_meta_t& i : m.inputFileMeta) {
        memset(&i, 0, sizeof(i));
        f.read(i.filename, FILE_NAME_WIDTH);
        f.read(i.epsgCode, STN_EPSG_WIDTH);
        f.read(i.epoch, STN_EPOCH_WIDTH);
        if (has_obs_epoch)
            f.read(i.observation_epoch, STN_EPOCH_WIDTH);
        else
            memcpy(i.observation_epoch, i.epoch, STN_EPOCH_WIDTH);
        f.read(reinterpret_cast<char*>(&i.filetype), sizeof(UINT16));
        f.read(reinterpret_cast<char*>(&i.datatype), sizeof(UINT16));
    }
    m.sourceFileMeta.clear();
    if (version_at_least(version, 1, 1)) {
        uint64_t nsrc = 0;
        f.read(reinterpret_cast<char*>(&nsrc), sizeof(uint64_t));
        if (!f || nsrc > (1u << 20)) throw std::runtime_error("corrupt file metadata (source file count)");
        m.sourceFileMeta.assign(nsrc, source_file_meta_t());
        for (source_file_meta_t& s : m.sourceFileMeta) {
            memset(&s, 0, sizeof(s));
            f.read(s.filename, FILE_NAME_WIDTH);
        }
    }
}

template <class Rec>
void read_records(const std::string& path, std::vector<Rec>& out, binary_file_meta_t& meta, file_info_t* info, bool need_v12,
                  const char* what) {
    std::ifstream f(path, std::ios::in | std::ios::binary);
    if (!f) throw std::runtime_error(std::string("LoadFile(): An error was encountered when opening ") + path + ".");
    file_info_t fi = read_file_info(f);
    read_meta(f, meta, fi.version);
    if (need_v12 && !version_at_least(fi.version, 1, 2))
        // bms_file.cpp:150-156
        throw std::runtime_error(std::string(what) + " file version " + fi.version +
                                 " predates observation_epoch support (v1.2); please re-run dnaimport.");
    if (!f) throw std::runtime_error(std::string("LoadFile(): An error was encountered when reading from ") + path + ".");
    out.resize(meta.binCount);
    if (meta.binCount) f.read(reinterpret_cast<char*>(out.data()), (std::streamsize)(meta.binCount * sizeof(Rec)));
    if (!f) throw std::runtime_error(std::string("LoadFile(): An error was encountered when reading from ") + path + ".");
    if (info) *info = fi;
}

template <class Rec>
void write_records(const std::string& path, const std::vector<Rec>& recs, binary_file_meta_t meta, const std::string& app) {
    std::ofstream f(path, std::ios::out | std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error(std::string("WriteFile(): An error was encountered when opening ") + path + ".");
    write_file_info(f, app);
    meta.binCount = recs.size();
    write_meta(f, meta);
    if (!recs.empty()) f.write(reinterpret_cast<const char*>(recs.data()), (std::streamsize)(recs.size() * sizeof(Rec)));
    if (!f) throw std::runtime_error(std::string("WriteFile(): An error was encountered when writing to ") + path + ".");
}

// widths, include/config/dnaconsts-iostream.hpp:40-73
constexpr int PRINT_VAR_PAD = 35;
constexpr int BLOCK = 14, NETID = 14, INNER = 16, JUNCT = 16, MEASR = 16, TOTAL = 16, PAD = 5;
const char* OUTPUTLINE = "--------------------------------------------------------------------------------";

bool get_line(std::ifstream& f, std::string& s) { return static_cast<bool>(std::getline(f, s)); }

UINT32 field_u32(const std::string& line, size_t col, size_t width, const char* what) {
    if (col >= line.size()) throw std::runtime_error(std::string("  Segmentation file is corrupt: Could not extract ") + what);
    std::string t = trim(line.substr(col, width));
    if (t.empty()) throw std::runtime_error(std::string("  Segmentation file is corrupt: Could not extract ") + what);
    char* end = nullptr;
    unsigned long v = strtoul(t.c_str(), &end, 10);
    if (end == t.c_str()) throw std::runtime_error(std::string("  Segmentation file is corrupt: Could not extract ") + what);
    return (UINT32)v;
}

}  // namespace

void read_bst(const std::string& path, std::vector<station_t>& stations, binary_file_meta_t& meta, file_info_t* info) {
    read_records(path, stations, meta, info, false, "BST");
}
void write_bst(const std::string& path, const std::vector<station_t>& stations, binary_file_meta_t meta, const std::string& app) {
    write_records(path, stations, meta, app);
}
void read_bms(const std::string& path, std::vector<measurement_t>& msrs, binary_file_meta_t& meta, file_info_t* info) {
    read_records(path, msrs, meta, info, true, "BMS");
}
void write_bms(const std::string& path, const std::vector<measurement_t>& msrs, binary_file_meta_t meta, const std::string& app) {
    write_records(path, msrs, meta, app);
}

// asl_file.cpp:75-100 / dnatemplatestnmsrfuncs.hpp:884-918
void read_asl(const std::string& path, std::vector<asl_entry_t>& asl) {
    std::ifstream f(path, std::ios::in | std::ios::binary);
    if (!f) throw std::runtime_error(std::string("LoadFile(): An error was encountered when opening ") + path + ".");
    read_file_info(f);
    uint64_t count = 0;
    f.read(reinterpret_cast<char*>(&count), sizeof(uint64_t));
    if (!f || count > (1ull << 32)) throw std::runtime_error(std::string("LoadFile(): An error was encountered when reading from ") + path + ".");
    asl.assign(count, asl_entry_t());
    for (asl_entry_t& e : asl) {
        f.read(reinterpret_cast<char*>(&e.assocMsrCount), sizeof(UINT32));
        f.read(reinterpret_cast<char*>(&e.amlStnIndex), sizeof(UINT32));
        f.read(reinterpret_cast<char*>(&e.validity), sizeof(UINT16));
    }
    if (!f) throw std::runtime_error(std::string("LoadFile(): An error was encountered when reading from ") + path + ".");
}

void write_asl(const std::string& path, const std::vector<asl_entry_t>& asl, const std::string& app) {
    std::ofstream f(path, std::ios::out | std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error(std::string("WriteFile(): An error was encountered when opening ") + path + ".");
    write_file_info(f, app);
    uint64_t count = asl.size();
    f.write(reinterpret_cast<const char*>(&count), sizeof(uint64_t));
    for (const asl_entry_t& e : asl) {
        f.write(reinterpret_cast<const char*>(&e.assocMsrCount), sizeof(UINT32));
        f.write(reinterpret_cast<const char*>(&e.amlStnIndex), sizeof(UINT32));
        f.write(reinterpret_cast<const char*>(&e.validity), sizeof(UINT16));
    }
    if (!f) throw std::runtime_error(std::string("WriteFile(): An error was encountered when writing to ") + path + ".");
}

// SegFile::LoadSegFileHeader + LoadSegFile, seg_file.cpp:57-408
void read_seg(const std::string& path, seg_data_t& seg, const std::vector<measurement_t>* bms) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error(std::string("load_seg_file(): An error was encountered when opening ") + path + ".");
    const std::string err = std::string("load_seg_file(): An error was encountered when reading from ") + path + ".\n";
    std::string line;
    auto skip = [&](int n) {
        for (int i = 0; i < n; ++i)
            if (!get_line(f, line)) throw std::runtime_error(err + "  unexpected end of file");
    };
    skip(13);                       // rule, title, blank, version, build, created, file name, blank, args, blank, stn file, msr file, blank
    skip(1);                        // Minimum inner stations
    seg.minInnerStns = line.size() > (size_t)PRINT_VAR_PAD ? (UINT32)strtoul(line.c_str() + PRINT_VAR_PAD, nullptr, 0) : 0;
    skip(1);                        // Block size threshold
    seg.blockThreshold = line.size() > (size_t)PRINT_VAR_PAD ? (UINT32)strtoul(line.c_str() + PRINT_VAR_PAD, nullptr, 0) : 0;
    skip(1);                        // Starting stations
    skip(1);
    while (line.empty() || line[0] != '-') skip(1);
    skip(3);                        // blank, SEGMENTATION SUMMARY, blank
    skip(1);                        // No. blocks produced
    seg.blockCount = line.size() > (size_t)PRINT_VAR_PAD ? (UINT32)strtoul(line.c_str() + PRINT_VAR_PAD, nullptr, 0) : 0;
    const UINT32 B = seg.blockCount;
    seg.ISL.assign(B, {});
    seg.JSL.assign(B, {});
    seg.CML.assign(B, {});
    seg.ContiguousNetList.assign(B, 0);
    seg.measurementCount.assign(B, 0);
    seg.unknownsCount.assign(B, 0);
    seg.parameterStationCount.assign(B, 0);
    skip(2);                        // rule, column header
    UINT32 blk = 0;
    for (UINT32 t = 0; t < B; ++t) {
        skip(1);
        if (line.compare(0, 20, "--------------------") == 0) throw std::runtime_error(err + "  Segmentation file is corrupt.");
        size_t col = 0;
        blk = field_u32(line, col, BLOCK, "Block number");
        col += BLOCK;
        UINT32 netID = field_u32(line, col, NETID, "Network ID");
        col += NETID;
        UINT32 jsl = field_u32(line, col, JUNCT, "Junction station count");
        col += JUNCT;
        UINT32 isl = field_u32(line, col, INNER, "Inner station count");
        col += INNER;
        UINT32 msr = field_u32(line, col, MEASR, "Measurement count");
        col += MEASR;
        UINT32 tot = field_u32(line, col, std::string::npos, "Total station count");
        if (tot != isl + jsl) throw std::runtime_error(err + "  Segmentation file is corrupt.");
        seg.JSL[t].assign(jsl, 0);
        seg.ISL[t].assign(isl, 0);
        seg.CML[t].assign(msr, 0);
        seg.ContiguousNetList[t] = netID;
        seg.parameterStationCount[t] = tot;
    }
    if (B && blk != B) throw std::runtime_error("load_seg_file: Segmentation file is corrupt.");
    skip(4);                        // rule, blank, INDIVIDUAL BLOCK DATA, rule
    for (UINT32 b = 0; b < B; ++b) {
        skip(1);                    // blank
        skip(1);                    // Block #
        unsigned bn = 0;
        if (line.size() < 6 || sscanf(line.c_str() + 5, "%u", &bn) != 1 || bn != b + 1)
            throw std::runtime_error("load_seg_file: segmentation file is corrupt.");
        skip(8);                    // rule, junction, inner, measurements, total, blank, header, rule
        skip(1);
        const UINT32 nj = (UINT32)seg.JSL[b].size(), ni = (UINT32)seg.ISL[b].size(), nm = (UINT32)seg.CML[b].size();
        UINT32 c = 0;
        while (line.compare(0, 20, "--------------------") != 0) {
            unsigned v;
            // the reference scans "%16u" at columns 0, 12 and 24 (seg_file.cpp:333-347)
            if (c < ni) {
                if (sscanf(line.c_str(), "%16u", &v) != 1) throw std::runtime_error(err + "  bad inner station");
                seg.ISL[b][c] = v;
                seg.unknownsCount[b] += 3;
            }
            if (c < nj) {
                if (line.size() <= 12 || sscanf(line.c_str() + 12, "%16u", &v) != 1) throw std::runtime_error(err + "  bad junction station");
                seg.JSL[b][c] = v;
                seg.unknownsCount[b] += 3;
            }
            if (c < nm) {
                if (line.size() <= 24 || sscanf(line.c_str() + 24, "%16u", &v) != 1) throw std::runtime_error(err + "  bad measurement");
                seg.CML[b][c] = v;
                if (bms) {
                    if (v >= bms->size()) throw std::runtime_error(err + "  measurement index out of range");
                    const measurement_t& m = (*bms)[v];
                    switch (m.measType) {   // seg_file.cpp:355-386
                        case 'G': seg.measurementCount[b] += 3; break;
                        case 'X':
                        case 'Y': seg.measurementCount[b] += m.vectorCount1 * 3; break;
                        case 'D': seg.measurementCount[b] += m.vectorCount2 - 1; break;
                        default: seg.measurementCount[b] += 1; break;
                    }
                }
            }
            ++c;
            skip(1);
        }
        if (c < ni || c < nj || c < nm) throw std::runtime_error(err + "  block lists are shorter than the summary table");
    }
}

// SegFile::WriteSegFile / WriteSegBlock, seg_file.cpp:489-721
void write_seg(const std::string& path, const seg_data_t& seg, const std::string& bst_file, const std::string& bms_file,
               const std::vector<measurement_t>& bms) {
    std::ofstream f(path, std::ios::out | std::ios::trunc);
    if (!f) throw std::runtime_error(std::string("write_seg_file(): An error was encountered when opening ") + path + ".");
    auto L = [&](const char* name) -> std::ostream& { return f << std::setw(PRINT_VAR_PAD) << std::left << name; };
    char created[64];
    std::time_t t = std::time(nullptr);
    std::tm tmv;
    localtime_r(&t, &tmv);
    std::strftime(created, sizeof(created), "%A, %d %B %Y, %X", &tmv);
    f << OUTPUTLINE << "\n" << "DYNADJUST SEGMENTATION OUTPUT FILE" << "\n\n";
    L("Version: ") << "1.4.0 (dnagpu strip segmenter)" << "\n";
    L("Build: ") << __DATE__ << ", " << __TIME__ << "\n";
    L("File created:") << created << "\n";
    L("File name:") << path << "\n\n";
    L("Command line arguments: ") << "synthetic" << "\n\n";
    L("Stations file:") << bst_file << "\n";
    L("Measurements file:") << bms_file << "\n";
    f << "\n";
    L("Minimum inner stations") << seg.minInnerStns << "\n";
    L("Block size threshold") << seg.blockThreshold << "\n";
    L("Starting station(s)") << " " << "\n";
    f << OUTPUTLINE << "\n\n";
    L("SEGMENTATION SUMMARY") << "\n\n";
    L("No. blocks produced") << seg.ISL.size() << "\n";
    const int rule = BLOCK + NETID + INNER + JUNCT + TOTAL + MEASR - 2;
    f << std::string(rule, '-') << "\n";
    f << std::setw(BLOCK) << std::left << "  Block" << std::setw(NETID) << std::left << "Network ID" << std::setw(JUNCT) << std::left
      << "Junction stns" << std::setw(INNER) << std::left << "Inner stns" << std::setw(MEASR) << std::left << "Measurements"
      << std::setw(TOTAL) << std::left << "Total stns" << "\n";
    for (size_t b = 0; b < seg.ISL.size(); ++b) {
        f << "  " << std::setw(BLOCK - 2) << std::left << (b + 1);
        f << std::setw(NETID) << std::left << seg.ContiguousNetList[b];
        f << std::setw(JUNCT) << std::left << seg.JSL[b].size();
        f << std::setw(INNER) << std::left << seg.ISL[b].size();
        f << std::setw(MEASR) << std::left << seg.CML[b].size();
        f << std::setw(TOTAL) << std::left << (seg.ISL[b].size() + seg.JSL[b].size()) << "\n";
    }
    f << std::string(rule, '-') << "\n\n" << "INDIVIDUAL BLOCK DATA" << "\n" << std::string(rule, '-') << "\n";
    const int brule = INNER + JUNCT + MEASR + PAD;
    for (size_t b = 0; b < seg.ISL.size(); ++b) {
        const auto &I = seg.ISL[b], &J = seg.JSL[b], &M = seg.CML[b];
        f << "\n" << "Block " << (b + 1) << "\n" << std::string(brule, '-') << "\n";
        f << std::setw(JUNCT) << std::left << "Junction stns:" << std::setw(BLOCK) << J.size() << "\n";
        f << std::setw(JUNCT) << std::left << "Inner stns:" << std::setw(BLOCK) << I.size() << "\n";
        f << std::setw(JUNCT) << std::left << "Measurements:" << std::setw(BLOCK) << M.size() << "\n";
        f << std::setw(JUNCT) << std::left << "Total stns:" << std::setw(BLOCK) << (J.size() + I.size()) << "\n\n";
        f << std::setw(INNER) << std::left << "Inner stns" << std::setw(JUNCT) << std::left << "Junction stns" << std::setw(MEASR)
          << "Measurements" << std::setw(PAD) << "Type" << "\n";
        f << std::string(brule, '-') << "\n";
        size_t rows = std::max(I.size(), std::max(J.size(), M.size()));
        for (size_t r = 0; r < rows || r == 0; ++r) {
            if (r < I.size()) f << std::setw(INNER) << std::left << I[r]; else f << std::setw(INNER) << std::left << " ";
            if (r < J.size()) f << std::setw(JUNCT) << std::left << J[r]; else f << std::setw(JUNCT) << std::left << " ";
            if (r < M.size())
                f << std::setw(MEASR) << std::left << M[r] << std::setw(PAD) << std::left
                  << (M[r] < bms.size() ? bms[M[r]].measType : '?') << "\n";
            else
                f << "\n";
            if (rows == 0) break;
        }
        f << std::string(brule, '-') << "\n";
    }
    f << "\n";
    if (!f) throw std::runtime_error(std::string("write_seg_file(): An error was encountered when writing to ") + path + ".");
}

}  // namespace iostreams
}  // namespace dynadjust
