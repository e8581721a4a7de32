// The exchange step of the phased adjustment across GPUs, as the C++ host sees it.
//
// The reference's parallel driver lives inside the class (dna_adjust::AdjustPhasedMultiThread, dnaadjust-multi.cpp:92-244:
// forward / reverse / combine threads exchanging junction matrices through shared memory).  Here the blocks are spread over
// GPUs -- one rank per GPU, either one process each (launched by mpirun / torchrun / a job scheduler) or one host thread each
// inside a single process -- and what the reference's threads hand to each other through v_junctionVariances_ travels over
// RCCL on xGMI: broadcast of a condensed block from its owner, all-reduce of the coordinate vector, point-to-point sends of
// junction matrices in the reference's schedule.
//
// Two transports behind one interface:
//   rccl   ncclBroadcast / ncclAllReduce / ncclSend / ncclRecv from librccl.so (loaded with dlopen, so that a process that
//          never leaves one GPU needs no RCCL); buffers are device memory and are used in place
//   local  ranks are threads of this process: device-to-device copies between the ranks' buffers, rendezvous through a
//          barrier in host memory.  For ranks that share a device (RCCL refuses those) and for tests on a one-GPU box.
//   shared ranks are PROCESSES without RCCL between them (several on one GPU, or no fabric): host-staged payloads over TCP
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <memory>
#include <string>
#include <vector>

namespace dynadjust {
namespace networkadjust {

constexpr size_t DIST_UNIQUE_ID_BYTES = 128;   // NCCL_UNIQUE_ID_BYTES

class DistComm {
public:
    virtual ~DistComm() {}
    virtual int rank() const = 0;
    virtual int world() const = 0;
    virtual const char* transport() const = 0;
    virtual int communicator_ranks() const { return 0; }   // ncclCommCount of the RCCL communicator (0: not RCCL)
    // Collectives on device buffers of this rank's device.  Calls between group_begin() and group_end() are issued together
    // (one ncclGroup); all of them are complete, for the host and for every stream, when wait() returns.
    virtual void group_begin() = 0;
    virtual void group_end() = 0;
    virtual void broadcast(double* buf, size_t count, int root) = 0;
    virtual void all_reduce_sum(double* buf, size_t count) = 0;
    virtual void send(const double* buf, size_t count, int peer) = 0;
    virtual void recv(double* buf, size_t count, int peer) = 0;
    virtual void wait() = 0;
    // Part q (bufs[q], counts[q] doubles) is produced by rank q; afterwards every rank holds all parts.  Ordered with the caller's
    // stream: enqueued on it (RCCL: one ncclBroadcast per part, one group) or completed before returning (local: the stream is
    // synchronised first).  The exchange of the intra-block distributed inverse (dnagpu_set_inverse_exchange).
    virtual void broadcast_parts_on(hipStream_t stream, int nparts, double* const* bufs, const size_t* counts) = 0;
    // bytes this rank moved through the transport since creation (sent + received payload, for the exchange model)
    virtual uint64_t bytes_moved() const = 0;
};

// Every wait of a transport has a deadline (default 600 s; DNAGPU_COLLECTIVE_TIMEOUT_S): past it the communicator is aborted and the
// waiting rank throws -- a rank that died cannot hang the others (the reference's sentinel, dnaadjust-multi.cpp:36-58).
void dist_set_collective_timeout(double seconds);
double dist_collective_timeout();

// ---- RCCL -----------------------------------------------------------------------------------------------------------------
// true when librccl could be loaded (DNAGPU_RCCL_LIB overrides the name; an already loaded librccl.so.1 -- e.g. the one a host
// application such as PyTorch brought -- is reused)
bool rccl_available(std::string* why = nullptr);
// rank 0 (or any one rank) creates the id, every rank needs the same 128 bytes before rccl_comm_create
void rccl_unique_id(unsigned char id[DIST_UNIQUE_ID_BYTES]);
// collective over all ranks (ncclCommInitRank); `device` is this rank's HIP device
std::shared_ptr<DistComm> rccl_comm_create(int rank, int world, const unsigned char id[DIST_UNIQUE_ID_BYTES], int device);

// Out-of-band distribution of the id for processes that have nothing else in common (no MPI, no Python): rank 0 listens on
// addr:port, the others connect (retrying for `timeout_s`) and read the 128 bytes.  addr / port default to MASTER_ADDR /
// MASTER_PORT + 17 of the environment (the variables torchrun, Slurm wrappers and MPI launch scripts export).
void tcp_share_unique_id(int rank, int world, unsigned char id[DIST_UNIQUE_ID_BYTES], const char* addr = nullptr, int port = 0,
                         double timeout_s = 120.0);

// ---- shared (processes without RCCL between them) ------------------------------------------------------------------------------
// Ranks that are processes sharing ONE GPU (RCCL refuses two ranks on a device) or GPUs without a fabric: payloads staged through host
// memory, one TCP connection per pair of ranks (the higher rank connects to addr : port + lower rank; addr / port default to MASTER_ADDR /
// MASTER_PORT + 18 ...), all transfers of a group driven together.  Same deadline and "a rank has gone" behaviour as the other transports.
// Collective over all ranks (dist_comm_shared.cpp).
std::shared_ptr<DistComm> shared_comm_create(int rank, int world, int device, const char* addr = nullptr, int port = 0, double timeout_s = 120.0);

// ---- plan (no transport) ---------------------------------------------------------------------------------------------------
// a communicator that only knows its rank and the world size: what dna_adjust::PlanDistributed prepares against; every exchange throws
std::shared_ptr<DistComm> plan_comm_create(int rank, int world);

// ---- local (threads of one process) ---------------------------------------------------------------------------------------
// creates the communicators of all `world` ranks at once; rank r's calls must come from one thread
std::vector<std::shared_ptr<DistComm>> local_comm_create(int world, const std::vector<int>& devices);

}  // namespace networkadjust
}  // namespace dynadjust
