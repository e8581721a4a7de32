// A caller written against the REFERENCE's dna_adjust interface, compiled against this repository's drop-in class
// (dynadjust_amd/csrc/host/dna_adjust.hpp) and linked to libdnagpu.so: the proof that dnaadjustwrapper's use of the class builds and
// runs unchanged.  It follows the wrapper statement by statement --
//   main():                      dnaadjustwrapper.cpp:1142-1452 (LoadSegmentationFileParameters, SIGINT -> CancelAdjustment, SetMaxBlasThreads,
//                                progress thread + adjustment functor, adjustTime().count(), GetStatus, PrintOscillationSummary, the report calls,
//                                CloseOutputFiles, UpdateBinaryFiles, PrintSuspectMeasurementSummary)
//   dna_adjust_thread:           dnaadjustprogress.cpp:49-117 (PrepareAdjustment / AdjustNetwork with the wrapper's catch ladder, SetExceptionRaised)
//   dna_adjust_progress_thread:  dnaadjustprogress.cpp:203-330 (IsPreparing / IsAdjusting / ExceptionRaised / NewMessagesAvailable /
//                                GetMessageIteration / GetMaxCorrection(it) / GetIterationTime(it) / CurrentBlock / processingForward / ...)
//   PrintSummaryMessage, GenerateStatistics, Serialise..., Print...: dnaadjustwrapper.cpp:104-372
// -- and prints one JSON line with what it read through the getters, which tests/test_gpu_boundary.py compares with the results
// the same adjustment gives through the C view.  Built by __graft_entry__.build(); usage:
//   wrapper_sequence <folder> <network> <simult|phased> [multi-thread: 0|1] [report-mode: 0|1]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>

#include "dna_adjust.hpp"

using namespace dynadjust;
using namespace dynadjust::networkadjust;

static std::mutex cout_mutex;
static std::atomic<bool> running{false};
static dna_adjust* g_netAdjust = nullptr;
static std::ostringstream progress_log;

static void sigint_handler(int) {
    if (g_netAdjust) g_netAdjust->CancelAdjustment();
}

// dnaadjustprogress.cpp:49-117
struct dna_adjust_thread {
    dna_adjust* _dnaAdj;
    project_settings* _p;
    _ADJUST_STATUS_* _adjustStatus;
    void operator()() {
        running = true;
        if (prepareAdjustment()) processAdjustment();
        running = false;
    }
    void handle(const char* stage, const std::string& msg) {
        std::lock_guard<std::mutex> g(cout_mutex);
        std::cerr << "- Error (" << stage << "): " << msg << std::endl;
    }
    bool prepareAdjustment() {
        try {
            *_adjustStatus = ADJUST_EXCEPTION_RAISED;
            _dnaAdj->PrepareAdjustment(*_p);
            *_adjustStatus = ADJUST_SUCCESS;
            return true;
        } catch (const NetAdjustException& e) {
            handle("prepare", e.what());
            _dnaAdj->SetExceptionRaised();
        } catch (const NetMemoryException& e) {
            handle("prepare", e.what());
            _dnaAdj->SetExceptionRaised();
        } catch (const std::runtime_error& e) {
            handle("prepare", e.what());
            _dnaAdj->SetExceptionRaised();
        } catch (const std::exception& e) {
            handle("prepare", std::string("Standard exception: ") + e.what());
            _dnaAdj->SetExceptionRaised();
        } catch (...) {
            handle("prepare", "Undefined error.");
            _dnaAdj->SetExceptionRaised();
        }
        *_adjustStatus = ADJUST_EXCEPTION_RAISED;
        return false;
    }
    bool processAdjustment() {
        try {
            *_adjustStatus = _dnaAdj->AdjustNetwork();
            return true;
        } catch (const NetAdjustException& e) {
            handle("adjust", e.what());
            _dnaAdj->SetExceptionRaised();
        } catch (const NetMemoryException& e) {
            handle("adjust", e.what());
            _dnaAdj->SetExceptionRaised();
        } catch (const std::runtime_error& e) {
            handle("adjust", e.what());
            _dnaAdj->SetExceptionRaised();
        } catch (...) {
            handle("adjust", "Undefined error.");
            _dnaAdj->SetExceptionRaised();
        }
        *_adjustStatus = ADJUST_EXCEPTION_RAISED;
        return false;
    }
};

// dnaadjustprogress.cpp:203-330 (the progress line goes to a log instead of the terminal)
struct dna_adjust_progress_thread {
    dna_adjust* _dnaAdj;
    project_settings* _p;
    void processMessages() {
        UINT32 currentIteration = 0;
        while (_dnaAdj->NewMessagesAvailable()) {
            if (!_dnaAdj->GetMessageIteration(currentIteration)) break;
            progress_log << "  Iteration " << std::right << std::setw(2) << currentIteration << ", max station corr: " << std::right << std::setw(12)
                         << _dnaAdj->GetMaxCorrection(currentIteration) << ", time: " << _dnaAdj->GetIterationTime(currentIteration) << "\n";
        }
    }
    void operator()() {
        UINT32 block = 0, polls = 0;
        while (running) {
            if (_dnaAdj->ExceptionRaised()) return;
            if (_dnaAdj->IsPreparing()) {
                ++polls;
            } else if (_dnaAdj->IsAdjusting()) {
                processMessages();
                if (_p->a.adjust_mode != SimultaneousMode && block != _dnaAdj->CurrentBlock()) {
                    block = _dnaAdj->CurrentBlock();
                    progress_log << "  block " << block + 1 << " of " << _dnaAdj->blockCount() << (_dnaAdj->processingCombine() ? " (combine)" : (_dnaAdj->processingForward() ? " (forward)" : " (reverse)"))
                                 << ", " << _dnaAdj->CurrentBlockStationCount() << " stations, last block " << _dnaAdj->LastBlockElapsedMs() << " ms, iteration " << _dnaAdj->CurrentIteration() << "\n";
                }
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        processMessages();
    }
};

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cerr << "usage: wrapper_sequence <folder> <network> <simult|phased> [multi-thread] [report-mode]" << std::endl;
        return 2;
    }
    project_settings p;
    const std::string base = std::string(argv[1]) + "/" + argv[2];
    p.g.network_name = argv[2];
    p.g.output_folder = argv[1];
    p.g.input_folder = argv[1];
    p.a.adjust_mode = std::string(argv[3]) == "phased" ? PhasedMode : SimultaneousMode;
    p.a.multi_thread = argc > 4 ? (UINT16)atoi(argv[4]) : 0;
    p.a.report_mode = argc > 5 ? (UINT16)atoi(argv[5]) : 0;
    p.a.bst_file = base + ".bst";
    p.a.bms_file = base + ".bms";
    p.a.seg_file = base + ".seg";
    p.s.asl_file = base + ".asl";
    p.s.seg_file = p.a.seg_file;

    dna_adjust netAdjust;
    _ADJUST_STATUS_ adjustStatus = ADJUST_SUCCESS;
    try {
        if (p.a.adjust_mode == PhasedMode) netAdjust.LoadSegmentationFileParameters(p.a.seg_file);     // dnaadjustwrapper.cpp:1223
        const UINT32 blocks_from_seg = netAdjust.blockCount();

        running = true;
        g_netAdjust = &netAdjust;
        std::signal(SIGINT, sigint_handler);
        dna_adjust::SetMaxBlasThreads(4);
        dna_adjust_progress_thread prog{&netAdjust, &p};
        std::thread progress(prog);
        dna_adjust_thread{&netAdjust, &p, &adjustStatus}();
        progress.join();
        if (adjustStatus == ADJUST_EXCEPTION_RAISED) return EXIT_FAILURE;

        if (p.a.report_mode) netAdjust.DeSerialiseAdjustedVarianceMatrices();                          // :205
        const long long elapsed_ms = netAdjust.adjustTime().count();                                  // :1386
        if (netAdjust.GetStatus() > ADJUST_THRESHOLD_EXCEEDED) {                                      // :1391
            netAdjust.PrintOscillationSummary();
            std::cout << "{\"status\": " << (int)netAdjust.GetStatus() << "}" << std::endl;
            return EXIT_SUCCESS;
        }
        netAdjust.GenerateStatistics();                                                               // :217
        if (p.a.max_iterations > 0) netAdjust.SerialiseAdjustedVarianceMatrices();                     // :186
        netAdjust.GetPrinter()->PrintAdjustedNetworkMeasurements();                                   // :296
        netAdjust.GetPrinter()->PrintMeasurementsToStation();                                         // :312
        netAdjust.GetPrinter()->PrintAdjustedNetworkStations();                                       // :326
        netAdjust.CloseOutputFiles();                                                                 // :1414
        netAdjust.GetPrinter()->PrintPositionalUncertainty();                                         // :342
        netAdjust.GetPrinter()->PrintNetworkStationCorrections();                                     // :358
        netAdjust.UpdateBinaryFiles();                                                                // :372
        std::string sinex = base + ".snx";
        const bool snx = netAdjust.GetPrinter()->PrintEstimatedStationCoordinatestoSNX(sinex);        // :458 (not part of this library: false)

        std::ostringstream suspects;
        netAdjust.PrintOscillationSummary();                                                          // :1442
        netAdjust.PrintSuspectMeasurementSummary(suspects);                                           // :1443
        const std::string progress_text = progress_log.str(), suspect_text = suspects.str();
        std::cerr << progress_text << suspect_text;

        std::cout << std::setprecision(17) << "{\"status\": " << (int)netAdjust.GetStatus() << ", \"thread_status\": " << (int)adjustStatus << ", \"blocks\": " << netAdjust.blockCount()
                  << ", \"blocks_from_seg\": " << blocks_from_seg << ", \"iterations\": " << netAdjust.CurrentIteration() << ", \"elapsed_ms\": " << elapsed_ms
                  << ", \"unknowns\": " << netAdjust.GetUnknownsCount() << ", \"measurements\": " << netAdjust.GetMeasurementCount() << ", \"all_fixed\": " << (netAdjust.GetAllFixed() ? 1 : 0)
                  << ", \"dof\": " << netAdjust.GetDegreesOfFreedom() << ", \"chi_squared\": " << netAdjust.GetChiSquared() << ", \"sigma_zero\": " << netAdjust.GetSigmaZero()
                  << ", \"pelzer\": " << netAdjust.GetGlobalPelzerRel() << ", \"outliers\": " << netAdjust.GetPotentialOutlierCount() << ", \"lower\": " << netAdjust.GetChiSquaredLowerLimit()
                  << ", \"upper\": " << netAdjust.GetChiSquaredUpperLimit() << ", \"test\": " << netAdjust.GetTestResult() << ", \"max_correction\": " << netAdjust.GetMaxCorrection()
                  << ", \"progress_lines\": " << std::count(progress_text.begin(), progress_text.end(), '\n') << ", \"suspect_lines\": "
                  << std::count(suspect_text.begin(), suspect_text.end(), '\n') << ", \"sinex\": " << (snx ? 1 : 0) << ", \"blas_threads\": " << dna_adjust::GetMaxBlasThreads() << "}" << std::endl;
    } catch (const NetAdjustException& e) {                                                           // :1434
        std::lock_guard<std::mutex> g(cout_mutex);
        std::cout << std::endl << "- Error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
