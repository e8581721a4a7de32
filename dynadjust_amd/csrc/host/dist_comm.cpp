// Transports of the inter-GPU exchange (dist_comm.hpp): RCCL through dlopen, and the in-process one.
#include "dist_comm.hpp"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

#include <rccl/rccl.h>   // types and prototypes only: the library itself is loaded at run time

namespace dynadjust {
namespace networkadjust {

// How long a rank waits in a collective for the others before it gives up (dist_set_collective_timeout; DNAGPU_COLLECTIVE_TIMEOUT_S).
// The reference's threads cannot lose each other -- they share an address space and unblock their queues with a sentinel
// (dnaadjust-multi.cpp:36-58, 457-463); ranks on different GPUs can: one that dies inside a kernel never posts its side of the
// next ncclBroadcast, and hipStreamSynchronize on the others would wait for good.
static std::atomic<double> g_collective_timeout_s{[] {
    const char* e = getenv("DNAGPU_COLLECTIVE_TIMEOUT_S");
    return e && atof(e) > 0.0 ? atof(e) : 600.0;
}()};
void dist_set_collective_timeout(double seconds) { g_collective_timeout_s.store(seconds > 0.0 ? seconds : 600.0); }
double dist_collective_timeout() { return g_collective_timeout_s.load(); }

namespace {

std::string timeout_text(const char* where) {
    char buf[256];
    snprintf(buf, sizeof(buf), "inter-GPU exchange: no answer from the other GPUs within %.0f s (%s): a rank has failed or left the schedule.",
             g_collective_timeout_s.load(), where);
    return buf;
}

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("inter-GPU exchange: ") + what + ": " + hipGetErrorString(e));
}

// ---- librccl, loaded on first use ------------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {getenv("DNAGPU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // a copy the host application already loaded (matching soname) is taken first: one RCCL per process
        a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        for (const char* nm : names) {
            if (a.handle) break;
            if (nm && *nm) a.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.handle) {
            const char* e = dlerror();
            a.error = std::string("librccl could not be loaded") + (e ? std::string(": ") + e : std::string());
            return a;
        }
        bool ok = true;
        auto sym = [&](const char* nm) {
            void* p = dlsym(a.handle, nm);
            if (!p) {
                ok = false;
                a.error = std::string("librccl lacks ") + nm;
            }
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.CommAbort = (decltype(a.CommAbort))dlsym(a.handle, "ncclCommAbort");      // (optional: without it a timed-out communicator is only dropped)
        a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
        a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.Send = (decltype(a.Send))sym("ncclSend");
        a.Recv = (decltype(a.Recv))sym("ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        if (!ok) a.handle = nullptr;
        return a;
    }();
    return api;
}

void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess && r != ncclInProgress)
        throw std::runtime_error(std::string("RCCL ") + what + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "error"));
}

class RcclComm : public DistComm {
public:
    RcclComm(int rank, int world, const unsigned char* id, int device) : rank_(rank), world_(world), device_(device) {
        RcclApi& api = rccl();
        if (!api.handle) throw std::runtime_error(api.error);
        hip_check(hipSetDevice(device_), "hipSetDevice");
        hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
        ncclUniqueId uid;
        static_assert(sizeof(uid.internal) == DIST_UNIQUE_ID_BYTES, "unique id size");
        memcpy(uid.internal, id, DIST_UNIQUE_ID_BYTES);
        nccl_check(api.CommInitRank(&comm_, world_, uid, rank_), "ncclCommInitRank");
    }
    ~RcclComm() override {
        hipSetDevice(device_);
        if (stream_ && comm_) hipStreamSynchronize(stream_);      // (an aborted communicator's stream may never drain)
        if (comm_) rccl().CommDestroy(comm_);
        if (stream_) hipStreamDestroy(stream_);
    }
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    const char* transport() const override { return "rccl"; }
    int communicator_ranks() const override {
        int n = 0;
        return (comm_ && rccl().CommCount && rccl().CommCount(comm_, &n) == ncclSuccess) ? n : 0;
    }
    void group_begin() override { nccl_check(rccl().GroupStart(), "ncclGroupStart"); }
    void group_end() override { nccl_check(rccl().GroupEnd(), "ncclGroupEnd"); }
    void broadcast(double* buf, size_t count, int root) override {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        nccl_check(rccl().Broadcast(buf, buf, count, ncclDouble, root, comm_, stream_), "ncclBroadcast");
        bytes_ += count * sizeof(double);        // (payload offered to the transport, whatever the number of ranks)
    }
    void all_reduce_sum(double* buf, size_t count) override {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        nccl_check(rccl().AllReduce(buf, buf, count, ncclDouble, ncclSum, comm_, stream_), "ncclAllReduce");
        bytes_ += 2 * count * sizeof(double);
    }
    void send(const double* buf, size_t count, int peer) override {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        nccl_check(rccl().Send(buf, count, ncclDouble, peer, comm_, stream_), "ncclSend");
        bytes_ += count * sizeof(double);
    }
    void recv(double* buf, size_t count, int peer) override {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        nccl_check(rccl().Recv(buf, count, ncclDouble, peer, comm_, stream_), "ncclRecv");
        bytes_ += count * sizeof(double);
    }
    // everything enqueued on the RCCL stream is complete -- or the deadline has passed: the communicator is aborted (ncclCommAbort
    // releases the kernels that spin on a peer that will never answer) and the caller gets an exception instead of a hang; the
    // adjustment ends with ADJUST_EXCEPTION_RAISED on every rank that is still alive
    void wait() override {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        if (!comm_) throw std::runtime_error("inter-GPU exchange: the communicator was aborted after a time-out");
        const auto t0 = std::chrono::steady_clock::now();
        for (long spin = 0;; ++spin) {
            const hipError_t e = hipStreamQuery(stream_);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) hip_check(e, "hipStreamQuery (RCCL stream)");
            (void)hipGetLastError();
            if (spin > 4000) std::this_thread::sleep_for(std::chrono::microseconds(spin > 40000 ? 500 : 20));     // (first a pure poll: the usual wait is short)
            if ((spin & 255) == 255 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > g_collective_timeout_s.load()) {
                if (rccl().CommAbort) rccl().CommAbort(comm_);
                comm_ = nullptr;
                throw std::runtime_error(timeout_text("RCCL"));
            }
        }
    }
    void broadcast_parts_on(hipStream_t stream, int nparts, double* const* bufs, const size_t* counts) override {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        nccl_check(rccl().GroupStart(), "ncclGroupStart");
        for (int q = 0; q < nparts; ++q) {
            if (!counts[q]) continue;
            nccl_check(rccl().Broadcast(bufs[q], bufs[q], counts[q], ncclDouble, q, comm_, stream), "ncclBroadcast");
            if (q != rank_) bytes_ += counts[q] * sizeof(double);
        }
        nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    }
    uint64_t bytes_moved() const override { return bytes_; }

private:
    int rank_, world_, device_;
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    uint64_t bytes_ = 0;
};

// ---- ranks as threads of one process ---------------------------------------------------------------------------------------
struct LocalOp {
    enum Kind { BCAST, ALLREDUCE, SEND, RECV } kind;
    double* buf;
    size_t count;
    int peer;   // root / destination / source
};

struct LocalGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    std::vector<std::vector<LocalOp>> ops;            // per rank: the group being executed
    std::vector<std::vector<std::vector<double>>> contrib;   // per rank, per all-reduce of the group: its input on the host

    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (broken) throw std::runtime_error("inter-GPU exchange: another rank of the process failed");
        const uint64_t g = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            const bool in_time = cv.wait_for(lk, std::chrono::duration<double>(g_collective_timeout_s.load()), [&] { return generation != g || broken; });
            if (!in_time) {                 // a rank never arrived: nobody waits any longer, here or in a later barrier
                broken = true;
                cv.notify_all();
                throw std::runtime_error(timeout_text("ranks of one process"));
            }
            if (broken && generation == g) throw std::runtime_error("inter-GPU exchange: another rank of the process failed");
        }
    }
    void abandon() {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
};

class LocalComm : public DistComm {
public:
    LocalComm(std::shared_ptr<LocalGroup> g, int rank, int device) : g_(g), rank_(rank), device_(device) {}
    ~LocalComm() override { g_->abandon(); }    // a rank that goes away must not leave the others waiting
    int rank() const override { return rank_; }
    int world() const override { return g_->world; }
    const char* transport() const override { return "local"; }
    void group_begin() override { grouping_ = true; }
    void group_end() override {
        grouping_ = false;
        execute();
    }
    void broadcast(double* buf, size_t count, int root) override { post({LocalOp::BCAST, buf, count, root}); }
    void all_reduce_sum(double* buf, size_t count) override { post({LocalOp::ALLREDUCE, buf, count, 0}); }
    void send(const double* buf, size_t count, int peer) override { post({LocalOp::SEND, const_cast<double*>(buf), count, peer}); }
    void recv(double* buf, size_t count, int peer) override { post({LocalOp::RECV, buf, count, peer}); }
    void wait() override {}                    // group_end() returns with everything done
    void broadcast_parts_on(hipStream_t stream, int nparts, double* const* bufs, const size_t* counts) override {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");     // this rank's part is complete
        group_begin();
        for (int q = 0; q < nparts; ++q)
            if (counts[q]) broadcast(bufs[q], counts[q], q);
        group_end();
    }
    uint64_t bytes_moved() const override { return bytes_; }

private:
    void post(const LocalOp& op) {
        pending_.push_back(op);
        if (!grouping_) execute();
    }
    // every rank executes its group at the same point of its program (the contract of a collective): publish, download the
    // all-reduce inputs, then copy / sum, with a barrier between the phases and one before the buffers may change again
    void execute() {
        LocalGroup& g = *g_;
        hip_check(hipSetDevice(device_), "hipSetDevice");
        try {
            g.ops[rank_] = pending_;
            g.contrib[rank_].clear();
            for (const LocalOp& op : pending_)
                if (op.kind == LocalOp::ALLREDUCE) {
                    g.contrib[rank_].emplace_back(op.count);
                    hip_check(hipMemcpy(g.contrib[rank_].back().data(), op.buf, op.count * sizeof(double), hipMemcpyDeviceToHost), "all-reduce download");
                }
            g.barrier();
            size_t coll = 0, red = 0;
            std::vector<size_t> recv_seen(g.world, 0);
            std::vector<double> sum;
            for (const LocalOp& op : pending_) {
                switch (op.kind) {
                    case LocalOp::BCAST: {
                        if (op.peer != rank_) {
                            const LocalOp& src = nth_collective(g.ops[op.peer], coll);
                            if (src.kind != LocalOp::BCAST || src.count != op.count || src.peer != op.peer)
                                throw std::runtime_error("inter-GPU exchange: mismatched broadcast");
                            hip_check(hipMemcpy(op.buf, src.buf, op.count * sizeof(double), hipMemcpyDefault), "broadcast copy");
                            bytes_ += op.count * sizeof(double);
                        }
                        ++coll;
                        break;
                    }
                    case LocalOp::ALLREDUCE: {
                        sum.assign(op.count, 0.0);
                        for (int r = 0; r < g.world; ++r) {           // rank order: the same bits on every rank
                            const std::vector<double>& c = g.contrib[r].at(red);
                            if (c.size() != op.count) throw std::runtime_error("inter-GPU exchange: mismatched all-reduce");
                            for (size_t i = 0; i < op.count; ++i) sum[i] += c[i];
                        }
                        hip_check(hipMemcpy(op.buf, sum.data(), op.count * sizeof(double), hipMemcpyHostToDevice), "all-reduce upload");
                        bytes_ += 2 * op.count * sizeof(double);
                        ++coll;
                        ++red;
                        break;
                    }
                    case LocalOp::RECV: {
                        // the k-th receive from a peer matches that peer's k-th send to this rank
                        size_t want = recv_seen[op.peer]++, seen = 0;
                        const LocalOp* src = nullptr;
                        for (const LocalOp& o : g.ops[op.peer])
                            if (o.kind == LocalOp::SEND && o.peer == rank_ && seen++ == want) {
                                src = &o;
                                break;
                            }
                        if (!src || src->count != op.count) throw std::runtime_error("inter-GPU exchange: unmatched receive");
                        hip_check(hipMemcpy(op.buf, src->buf, op.count * sizeof(double), hipMemcpyDefault), "receive copy");
                        bytes_ += op.count * sizeof(double);
                        break;
                    }
                    case LocalOp::SEND: bytes_ += op.count * sizeof(double); break;
                }
            }
            // (a device-to-device hipMemcpy is ordered on the null stream but may return before it has run: the chains' streams do
            // not wait for the null stream, so the copies are completed here -- large parts lost this race, small ones never did)
            hip_check(hipStreamSynchronize(nullptr), "hipStreamSynchronize (copies)");
            g.barrier();          // nobody's source buffer changes before every copy out of it is done
        } catch (...) {
            pending_.clear();
            g.abandon();
            throw;
        }
        pending_.clear();
    }
    static const LocalOp& nth_collective(const std::vector<LocalOp>& ops, size_t n) {
        size_t i = 0;
        for (const LocalOp& o : ops)
            if (o.kind == LocalOp::BCAST || o.kind == LocalOp::ALLREDUCE) {
                if (i == n) return o;
                ++i;
            }
        throw std::runtime_error("inter-GPU exchange: collective sequences differ between ranks");
    }

    std::shared_ptr<LocalGroup> g_;
    int rank_, device_;
    bool grouping_ = false;
    std::vector<LocalOp> pending_;
    uint64_t bytes_ = 0;
};

}  // namespace

namespace {
class PlanComm : public DistComm {
public:
    PlanComm(int rank, int world) : rank_(rank), world_(world) {}
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    const char* transport() const override { return "plan"; }
    void group_begin() override { no(); }
    void group_end() override { no(); }
    void broadcast(double*, size_t, int) override { no(); }
    void all_reduce_sum(double*, size_t) override { no(); }
    void send(const double*, size_t, int) override { no(); }
    void recv(double*, size_t, int) override { no(); }
    void wait() override { no(); }
    void broadcast_parts_on(hipStream_t, int, double* const*, const size_t*) override { no(); }
    uint64_t bytes_moved() const override { return 0; }

private:
    static void no() { throw std::runtime_error("inter-GPU exchange: a plan has no transport"); }
    int rank_, world_;
};
}  // namespace

std::shared_ptr<DistComm> plan_comm_create(int rank, int world) { return std::make_shared<PlanComm>(rank, world); }

bool rccl_available(std::string* why) {
    RcclApi& api = rccl();
    if (!api.handle && why) *why = api.error;
    return api.handle != nullptr;
}

void rccl_unique_id(unsigned char id[DIST_UNIQUE_ID_BYTES]) {
    RcclApi& api = rccl();
    if (!api.handle) throw std::runtime_error(api.error);
    ncclUniqueId uid;
    nccl_check(api.GetUniqueId(&uid), "ncclGetUniqueId");
    memcpy(id, uid.internal, DIST_UNIQUE_ID_BYTES);
}

std::shared_ptr<DistComm> rccl_comm_create(int rank, int world, const unsigned char id[DIST_UNIQUE_ID_BYTES], int device) {
    return std::make_shared<RcclComm>(rank, world, id, device);
}

std::vector<std::shared_ptr<DistComm>> local_comm_create(int world, const std::vector<int>& devices) {
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    g->ops.resize(world);
    g->contrib.resize(world);
    std::vector<std::shared_ptr<DistComm>> out;
    for (int r = 0; r < world; ++r) out.push_back(std::make_shared<LocalComm>(g, r, devices.at(r)));
    return out;
}

// ---- the 128 bytes, from rank 0 to everybody, over TCP ----------------------------------------------------------------------
namespace {
void write_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
        ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
        if (w <= 0) throw std::runtime_error("unique-id exchange: send failed");
        c += w;
        n -= (size_t)w;
    }
}
void read_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n) {
        ssize_t r = ::recv(fd, c, n, 0);
        if (r <= 0) throw std::runtime_error("unique-id exchange: receive failed");
        c += r;
        n -= (size_t)r;
    }
}
}  // namespace

constexpr int32_t UNIQUE_ID_HELLO = 0x444e4131;     // "DNA1": what a rank says first

void tcp_share_unique_id(int rank, int world, unsigned char id[DIST_UNIQUE_ID_BYTES], const char* addr, int port, double timeout_s) {
    if (world <= 1) return;
    std::string host = addr && *addr ? addr : (getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1");
    if (port <= 0) {
        const char* e = getenv("DNAGPU_MASTER_PORT");
        if (e && atoi(e) > 0)
            port = atoi(e);
        else
            port = (getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) : 29500) + 17;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
    if (rank == 0) {
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) throw std::runtime_error("unique-id exchange: socket()");
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in sa;
        memset(&sa, 0, sizeof(sa));
        sa.sin_family = AF_INET;
        // the rendezvous address names the interface rank 0 listens on when it is a literal address (the launcher's
        // MASTER_ADDR=127.0.0.1 keeps the exchange inside the host) or a name that resolves to a routable local interface.  A NAME that
        // resolves to loopback says nothing about the other ranks' view of it -- Debian maps a node's own hostname to 127.0.1.1 in
        // /etc/hosts while the other nodes resolve the same name to its real address -- so it listens on every interface, as does a
        // name that does not resolve to a local address at all.
        sa.sin_addr.s_addr = htonl(INADDR_ANY);
        {
            in_addr literal;
            const bool is_literal = inet_pton(AF_INET, host.c_str(), &literal) == 1;
            addrinfo hints, *res = nullptr;
            memset(&hints, 0, sizeof(hints));
            hints.ai_family = AF_INET;
            hints.ai_socktype = SOCK_STREAM;
            if (getaddrinfo(host.c_str(), nullptr, &hints, &res) == 0 && res) {
                const in_addr got = ((sockaddr_in*)res->ai_addr)->sin_addr;
                const bool loopback = (ntohl(got.s_addr) >> 24) == 127;
                if (is_literal || !loopback) sa.sin_addr = got;
                freeaddrinfo(res);
            }
        }
        sa.sin_port = htons((uint16_t)port);
        if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0) {
            sa.sin_addr.s_addr = htonl(INADDR_ANY);
            if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0) {
                ::close(ls);
                throw std::runtime_error("unique-id exchange: cannot bind port " + std::to_string(port));
            }
        }
        if (::listen(ls, world) != 0) {
            ::close(ls);
            throw std::runtime_error("unique-id exchange: cannot listen on port " + std::to_string(port));
        }
        // every rank 1 .. world - 1 is served (a rank that asks again after a failed read is answered again and counted once);
        // a connection that does not introduce itself as one of them -- a port scan, a health probe -- gets nothing and counts for nothing
        std::vector<char> served((size_t)world, 0);
        for (int got = 0; got < world - 1;) {
            timeval tv;
            double left = std::chrono::duration<double>(deadline - std::chrono::steady_clock::now()).count();
            if (left <= 0) {
                ::close(ls);
                throw std::runtime_error("unique-id exchange: timed out waiting for the other ranks");
            }
            tv.tv_sec = (long)left;
            tv.tv_usec = (long)((left - (double)tv.tv_sec) * 1e6);
            fd_set fds;
            FD_ZERO(&fds);
            FD_SET(ls, &fds);
            if (::select(ls + 1, &fds, nullptr, nullptr, &tv) <= 0) continue;
            int c = ::accept(ls, nullptr, nullptr);
            if (c < 0) continue;
            timeval io = {5, 0};                 // (a silent connection may hold the loop for five seconds, not for good)
            setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &io, sizeof(io));
            setsockopt(c, SOL_SOCKET, SO_SNDTIMEO, &io, sizeof(io));
            int32_t hello[2] = {0, -1};
            try {
                read_all(c, hello, sizeof(hello));
                if (hello[0] == UNIQUE_ID_HELLO && hello[1] > 0 && hello[1] < world) {
                    write_all(c, id, DIST_UNIQUE_ID_BYTES);
                    if (!served[(size_t)hello[1]]) {
                        served[(size_t)hello[1]] = 1;
                        ++got;
                    }
                }
            } catch (...) {
            }
            ::close(c);
        }
        ::close(ls);
        return;
    }
    addrinfo hints, *res = nullptr;
    memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res)
        throw std::runtime_error("unique-id exchange: cannot resolve " + host);
    for (;;) {
        int s = ::socket(AF_INET, SOCK_STREAM, 0);
        if (s >= 0 && ::connect(s, res->ai_addr, res->ai_addrlen) == 0) {
            try {
                const int32_t hello[2] = {UNIQUE_ID_HELLO, rank};
                write_all(s, hello, sizeof(hello));
                read_all(s, id, DIST_UNIQUE_ID_BYTES);
                ::close(s);
                freeaddrinfo(res);
                return;
            } catch (...) {
            }
        }
        if (s >= 0) ::close(s);
        if (std::chrono::steady_clock::now() > deadline) {
            freeaddrinfo(res);
            throw std::runtime_error("unique-id exchange: rank 0 not reachable at " + host + ":" + std::to_string(port));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
}

}  // namespace networkadjust
}  // namespace dynadjust
