// The "shared" transport of the inter-GPU exchange (dist_comm.hpp): ranks are PROCESSES that cannot (or need not) use RCCL between them --
// several processes on one GPU (RCCL refuses two ranks on one device), or GPUs without a fabric between them.  Payloads are staged through
// host memory and travel over TCP sockets, one per pair of ranks; every rank drives all its transfers of a group together (poll), so the
// order in which two ranks post their sends and receives cannot deadlock them.  Slow by design (PCIe + loopback): it exists so that every
// multi-PROCESS path of the driver -- the rendezvous, the agreement after every phase, the two-level chains, the variance matrices that
// travel to rank 0, the result files, a rank that dies -- runs on a box with ONE GPU, and as the fallback where RCCL has no path.
// The reference's threads share an address space (dnaadjust-multi.cpp:92-244); this is the same hand-over between address spaces.
#include "dist_comm.hpp"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace dynadjust {
namespace networkadjust {

namespace {

constexpr int32_t SHARED_HELLO = 0x444e4133;     // "DNA3": first word of the greeting both sides of a new connection exchange

// The greeting: {SHARED_HELLO, rank, world size, job}.  The connecting rank sends its own, the accepting rank checks it and answers with
// its own, which the connecting rank checks in turn -- a listener of another job (another world size, another rendezvous) on the same
// port is recognised on BOTH sides instead of being talked to.  `job` is what every rank of one job can derive alone: the base port, the
// world size and the launcher's rendezvous (MASTER_ADDR / MASTER_PORT).  Ports base .. base + world - 2 must be free per communicator:
// two communicators created at the same time on one host need different bases (DNAGPU_MASTER_PORT, or the launcher's MASTER_PORT).
int32_t job_id(int world, int base_port) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) {
        for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char*)p)[i]) * 1099511628211ull;
    };
    mix(&world, sizeof(world));
    mix(&base_port, sizeof(base_port));
    for (const char* name : {"MASTER_ADDR", "MASTER_PORT"})
        if (const char* e = getenv(name)) mix(e, strlen(e) + 1);
    return (int32_t)(h ^ (h >> 32));
}

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("inter-GPU exchange (shared): ") + what + ": " + hipGetErrorString(e));
}

struct Xfer {                 // one message on one pair's socket
    int peer;
    bool send;
    char* p;
    size_t left;
};

class SharedComm : public DistComm {
public:
    SharedComm(int rank, int world, int device, const std::string& host, int base_port, double timeout_s) : rank_(rank), world_(world), device_(device) {
        fd_.assign((size_t)world, -1);
        connect_all(host, base_port, timeout_s);
    }
    ~SharedComm() override { close_all(); }
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    const char* transport() const override { return "shared"; }
    void group_begin() override { grouping_ = true; }
    void group_end() override {
        grouping_ = false;
        execute();
    }
    void broadcast(double* buf, size_t count, int root) override { post({Op::BCAST, buf, count, root}); }
    void all_reduce_sum(double* buf, size_t count) override { post({Op::ALLREDUCE, buf, count, 0}); }
    void send(const double* buf, size_t count, int peer) override { post({Op::SEND, const_cast<double*>(buf), count, peer}); }
    void recv(double* buf, size_t count, int peer) override { post({Op::RECV, buf, count, peer}); }
    void wait() override {}
    void broadcast_parts_on(hipStream_t stream, int nparts, double* const* bufs, const size_t* counts) override {
        hip_ok(hipSetDevice(device_), "hipSetDevice");
        hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
        group_begin();
        for (int q = 0; q < nparts; ++q)
            if (counts[q]) broadcast(bufs[q], counts[q], q);
        group_end();
    }
    uint64_t bytes_moved() const override { return bytes_; }

private:
    struct Op {
        enum Kind { BCAST, ALLREDUCE, SEND, RECV } kind;
        double* buf;
        size_t count;
        int peer;
    };

    void post(const Op& op) {
        pending_.push_back(op);
        if (!grouping_) execute();
    }

    // every pair of ranks gets one connection: the higher rank connects to the lower one's port (base + rank)
    void connect_all(const std::string& host, int base_port, double timeout_s) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
        int ls = -1;
        if (rank_ < world_ - 1) {
            ls = ::socket(AF_INET, SOCK_STREAM, 0);
            if (ls < 0) throw std::runtime_error("inter-GPU exchange (shared): socket()");
            int one = 1;
            setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
            sockaddr_in sa;
            memset(&sa, 0, sizeof(sa));
            sa.sin_family = AF_INET;
            sa.sin_addr.s_addr = htonl(INADDR_ANY);
            in_addr literal;
            if (inet_pton(AF_INET, host.c_str(), &literal) == 1 && rank_ == 0) sa.sin_addr = literal;     // (rank 0 sits at the rendezvous address)
            sa.sin_port = htons((uint16_t)(base_port + rank_));
            if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0) {
                sa.sin_addr.s_addr = htonl(INADDR_ANY);
                if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0) {
                    ::close(ls);
                    throw std::runtime_error("inter-GPU exchange (shared): cannot bind port " + std::to_string(base_port + rank_));
                }
            }
            if (::listen(ls, world_) != 0) {
                ::close(ls);
                throw std::runtime_error("inter-GPU exchange (shared): cannot listen");
            }
        }
        try {
            // connect to every lower rank (all ranks of a shared-GPU run live on the rendezvous host)
            const int32_t job = job_id(world_, base_port);
            auto read_all = [](int fd, void* buf, size_t n) {
                size_t have = 0;
                while (have < n) {
                    ssize_t r = ::recv(fd, (char*)buf + have, n - have, 0);
                    if (r <= 0) return false;
                    have += (size_t)r;
                }
                return true;
            };
            const timeval io = {1, 0};               // (a silent peer holds a greeting up for a second, not the whole rendezvous)
            for (int q = 0; q < rank_; ++q) {
                addrinfo hints, *res = nullptr;
                memset(&hints, 0, sizeof(hints));
                hints.ai_family = AF_INET;
                hints.ai_socktype = SOCK_STREAM;
                if (getaddrinfo(host.c_str(), std::to_string(base_port + q).c_str(), &hints, &res) != 0 || !res)
                    throw std::runtime_error("inter-GPU exchange (shared): cannot resolve " + host);
                bool connected = false;
                std::string refused;
                while (!connected) {
                    for (addrinfo* ai = res; ai && !connected; ai = ai->ai_next) {       // (every address the name resolves to)
                        int s = ::socket(AF_INET, SOCK_STREAM, 0);
                        if (s < 0) continue;
                        if (::connect(s, ai->ai_addr, ai->ai_addrlen) == 0) {
                            setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &io, sizeof(io));
                            const int32_t hello[4] = {SHARED_HELLO, rank_, world_, job};
                            int32_t back[4] = {0, -1, 0, 0};
                            if (::send(s, hello, sizeof(hello), MSG_NOSIGNAL) == (ssize_t)sizeof(hello) && read_all(s, back, sizeof(back))) {
                                if (back[0] == SHARED_HELLO && back[1] == q && back[2] == world_ && back[3] == job) {
                                    fd_[(size_t)q] = s;
                                    connected = true;
                                    break;
                                }
                                refused = " (a listener of another job answered there)";
                            }
                        }
                        ::close(s);
                    }
                    if (connected) break;
                    if (std::chrono::steady_clock::now() > deadline) {
                        freeaddrinfo(res);
                        throw std::runtime_error("inter-GPU exchange (shared): rank " + std::to_string(q) + " not reachable at " + host + ":" +
                                                 std::to_string(base_port + q) + refused);
                    }
                    std::this_thread::sleep_for(std::chrono::milliseconds(50));
                }
                freeaddrinfo(res);
            }
            // accept every higher rank
            for (int got = 0; got < world_ - 1 - rank_;) {
                const double left = std::chrono::duration<double>(deadline - std::chrono::steady_clock::now()).count();
                if (left <= 0) throw std::runtime_error("inter-GPU exchange (shared): timed out waiting for the other ranks");
                pollfd pf{ls, POLLIN, 0};
                if (::poll(&pf, 1, (int)std::min(1000.0, left * 1e3)) <= 0) continue;
                int c = ::accept(ls, nullptr, nullptr);
                if (c < 0) continue;
                setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &io, sizeof(io));
                int32_t hello[4] = {0, -1, 0, 0};
                const int32_t mine[4] = {SHARED_HELLO, rank_, world_, job};
                if (read_all(c, hello, sizeof(hello)) && hello[0] == SHARED_HELLO && hello[1] > rank_ && hello[1] < world_ && hello[2] == world_ &&
                    hello[3] == job && fd_[(size_t)hello[1]] < 0 && ::send(c, mine, sizeof(mine), MSG_NOSIGNAL) == (ssize_t)sizeof(mine)) {
                    fd_[(size_t)hello[1]] = c;
                    ++got;
                } else {
                    ::close(c);          // (not one of ours)
                }
            }
        } catch (...) {
            if (ls >= 0) ::close(ls);
            close_all();
            throw;
        }
        if (ls >= 0) ::close(ls);
        for (int q = 0; q < world_; ++q) {
            if (fd_[(size_t)q] < 0) continue;
            int one = 1;
            setsockopt(fd_[(size_t)q], IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            timeval none = {0, 0};
            setsockopt(fd_[(size_t)q], SOL_SOCKET, SO_RCVTIMEO, &none, sizeof(none));
        }
    }

    void close_all() {
        for (int& f : fd_)
            if (f >= 0) {
                ::shutdown(f, SHUT_RDWR);
                ::close(f);
                f = -1;
            }
    }

    // all transfers progress together; per pair and direction they complete in the order given (one stream per socket and direction)
    void run(std::vector<Xfer>& xs, const char* where) {
        if (broken_) throw std::runtime_error("inter-GPU exchange (shared): the connection to the other ranks was given up earlier");
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(dist_collective_timeout());
        for (;;) {
            // the first unfinished transfer per (peer, direction)
            std::vector<pollfd> pfs;
            std::vector<std::pair<int, int>> who;       // (index of the send, index of the receive) per polled socket
            bool any = false;
            for (int q = 0; q < world_; ++q) {
                int si = -1, ri = -1;
                for (size_t i = 0; i < xs.size(); ++i)
                    if (xs[i].peer == q && xs[i].left) {
                        if (xs[i].send && si < 0) si = (int)i;
                        if (!xs[i].send && ri < 0) ri = (int)i;
                    }
                if (si < 0 && ri < 0) continue;
                any = true;
                pollfd pf{fd_[(size_t)q], (short)((si >= 0 ? POLLOUT : 0) | (ri >= 0 ? POLLIN : 0)), 0};
                pfs.push_back(pf);
                who.emplace_back(si, ri);
            }
            if (!any) return;
            const double left = std::chrono::duration<double>(deadline - std::chrono::steady_clock::now()).count();
            if (left <= 0) give_up(std::string("no answer from the other GPUs within ") + std::to_string((int)dist_collective_timeout()) + " s (" + where + "): a rank has failed or left the schedule.");
            const int n = ::poll(pfs.data(), (nfds_t)pfs.size(), (int)std::min(200.0, left * 1e3));
            if (n < 0 && errno != EINTR) give_up("poll() failed");
            for (size_t k = 0; k < pfs.size(); ++k) {
                if (pfs[k].revents & (POLLERR | POLLNVAL)) give_up("a rank has gone (connection error): another rank failed or was stopped.");
                if ((pfs[k].revents & POLLIN) && who[k].second >= 0) {
                    Xfer& x = xs[(size_t)who[k].second];
                    const ssize_t r = ::recv(pfs[k].fd, x.p, std::min<size_t>(x.left, (size_t)4 << 20), MSG_DONTWAIT);
                    if (r == 0) give_up("a rank has gone (connection closed): another rank failed or was stopped.");
                    if (r < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) give_up("a rank has gone (receive failed): another rank failed or was stopped.");
                    if (r > 0) {
                        x.p += r;
                        x.left -= (size_t)r;
                        bytes_ += (uint64_t)r;
                    }
                } else if ((pfs[k].revents & POLLHUP) && who[k].second >= 0) {
                    give_up("a rank has gone (connection closed): another rank failed or was stopped.");
                }
                if ((pfs[k].revents & POLLOUT) && who[k].first >= 0) {
                    Xfer& x = xs[(size_t)who[k].first];
                    const ssize_t w = ::send(pfs[k].fd, x.p, std::min<size_t>(x.left, (size_t)4 << 20), MSG_DONTWAIT | MSG_NOSIGNAL);
                    if (w < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) give_up("a rank has gone (send failed): another rank failed or was stopped.");
                    if (w > 0) {
                        x.p += w;
                        x.left -= (size_t)w;
                        bytes_ += (uint64_t)w;
                    }
                }
            }
        }
    }
    [[noreturn]] void give_up(const std::string& why) {
        broken_ = true;
        close_all();
        throw std::runtime_error("inter-GPU exchange: " + why);
    }

    void execute() {
        hip_ok(hipSetDevice(device_), "hipSetDevice");
        std::vector<Op> ops;
        ops.swap(pending_);
        size_t i = 0;
        while (i < ops.size()) {
            if (ops[i].kind == Op::SEND || ops[i].kind == Op::RECV) {
                // a run of point-to-point messages: staged, then all in flight together
                size_t j = i;
                while (j < ops.size() && (ops[j].kind == Op::SEND || ops[j].kind == Op::RECV)) ++j;
                std::vector<std::vector<double>> host(j - i);
                std::vector<Xfer> xs;
                for (size_t k = i; k < j; ++k) {
                    if (ops[k].peer < 0 || ops[k].peer >= world_ || ops[k].peer == rank_) throw std::runtime_error("inter-GPU exchange (shared): bad peer");
                    host[k - i].resize(ops[k].count);
                    if (ops[k].kind == Op::SEND)
                        hip_ok(hipMemcpy(host[k - i].data(), ops[k].buf, ops[k].count * sizeof(double), hipMemcpyDeviceToHost), "download");
                    xs.push_back({ops[k].peer, ops[k].kind == Op::SEND, (char*)host[k - i].data(), ops[k].count * sizeof(double)});
                }
                run(xs, "point-to-point messages");
                for (size_t k = i; k < j; ++k)
                    if (ops[k].kind == Op::RECV)
                        hip_ok(hipMemcpy(ops[k].buf, host[k - i].data(), ops[k].count * sizeof(double), hipMemcpyHostToDevice), "upload");
                i = j;
                continue;
            }
            const Op& op = ops[i++];
            std::vector<double> h(op.count);
            if (op.kind == Op::BCAST) {
                std::vector<Xfer> xs;
                if (op.peer == rank_) {
                    hip_ok(hipMemcpy(h.data(), op.buf, op.count * sizeof(double), hipMemcpyDeviceToHost), "download");
                    for (int q = 0; q < world_; ++q)
                        if (q != rank_) xs.push_back({q, true, (char*)h.data(), op.count * sizeof(double)});
                    run(xs, "broadcast");
                } else {
                    xs.push_back({op.peer, false, (char*)h.data(), op.count * sizeof(double)});
                    run(xs, "broadcast");
                    hip_ok(hipMemcpy(op.buf, h.data(), op.count * sizeof(double), hipMemcpyHostToDevice), "upload");
                }
                continue;
            }
            // all-reduce: every rank's input to rank 0, summed there in rank order (the same bits everywhere), the sum back to everybody
            hip_ok(hipMemcpy(h.data(), op.buf, op.count * sizeof(double), hipMemcpyDeviceToHost), "download");
            if (rank_ == 0) {
                std::vector<std::vector<double>> in((size_t)world_);
                std::vector<Xfer> xs;
                for (int q = 1; q < world_; ++q) {
                    in[(size_t)q].resize(op.count);
                    xs.push_back({q, false, (char*)in[(size_t)q].data(), op.count * sizeof(double)});
                }
                run(xs, "all-reduce");
                for (int q = 1; q < world_; ++q)
                    for (size_t e = 0; e < op.count; ++e) h[e] += in[(size_t)q][e];
                xs.clear();
                for (int q = 1; q < world_; ++q) xs.push_back({q, true, (char*)h.data(), op.count * sizeof(double)});
                run(xs, "all-reduce");
            } else {
                std::vector<Xfer> xs{{0, true, (char*)h.data(), op.count * sizeof(double)}};
                run(xs, "all-reduce");
                xs.assign(1, {0, false, (char*)h.data(), op.count * sizeof(double)});
                run(xs, "all-reduce");
            }
            hip_ok(hipMemcpy(op.buf, h.data(), op.count * sizeof(double), hipMemcpyHostToDevice), "upload");
        }
    }

    int rank_, world_, device_;
    std::vector<int> fd_;
    bool grouping_ = false, broken_ = false;
    std::vector<Op> pending_;
    uint64_t bytes_ = 0;
};

}  // namespace

std::shared_ptr<DistComm> shared_comm_create(int rank, int world, int device, const char* addr, int port, double timeout_s) {
    std::string host = addr && *addr ? addr : (getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1");
    if (port <= 0) {
        const char* e = getenv("DNAGPU_MASTER_PORT");
        port = (e && atoi(e) > 0 ? atoi(e) : (getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) : 29500) + 17) + 1;    // (+ 17: the unique-id hand-off)
    }
    return std::make_shared<SharedComm>(rank, world, device, host, port, timeout_s);
}

}  // namespace networkadjust
}  // namespace dynadjust
