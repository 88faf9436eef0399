// ptnn_comm.hpp -- transports of the sharded ladder (host side only): RCCL over xGMI, loaded on demand, and a host-staged
// transport behind two callbacks.  What moves and when is decided in ptnn.hip (comm_swap_round); this file only moves bytes.
//
// Stands in for the reference's multiprocessing.Queue / Event star between the parent and its forked chains
// (multicore-pt-regression/pt_timeseries_regression.py:427-437 <-> 694-759).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library itself is dlopen'ed by the first ptnn_comm_init
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../../include/ptnn.h"

namespace ptnn {

// ---- progress stamps and bounded waits --------------------------------------------------------------------------------
// Every step of bringing a communicator up (dlopen of the 570 MB librccl, ncclGetUniqueId, ncclCommInitRank) and every wait
// behind a collective is stamped and bounded, so that a stall names its stage instead of hanging the caller: the reference's
// parent at least polls is_alive() every round (REG:721-727).  $PTNN_COMM_TRACE=1 prints the stamps to stderr as they happen;
// the last one is always kept and goes into the error text of a timeout (and ptnn_comm_last_stage).
inline double comm_clock() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
struct CommStageLog {
    std::mutex mu;
    char last[192] = "none";
    double t_last = 0.0;
};
inline CommStageLog& comm_stage_log() { static CommStageLog l; return l; }
inline void comm_stage(const char* fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    const double t = comm_clock();
    static const bool trace = [] { const char* e = std::getenv("PTNN_COMM_TRACE"); return e && *e && *e != '0'; }();
    CommStageLog& l = comm_stage_log();
    {
        std::lock_guard<std::mutex> lock(l.mu);
        snprintf(l.last, sizeof l.last, "%s", buf);
        l.t_last = t;
    }
    if (trace) { fprintf(stderr, "[ptnn comm %9.3f s pid %d] %s\n", t, (int)getpid(), buf); fflush(stderr); }
}
inline std::string comm_last_stage() {
    CommStageLog& l = comm_stage_log();
    std::lock_guard<std::mutex> lock(l.mu);
    char buf[256];
    snprintf(buf, sizeof buf, "%s (entered %.1f s ago)", l.last, comm_clock() - l.t_last);
    return buf;
}
// seconds a communicator stage may take before the call gives up with error -7: $PTNN_COMM_TIMEOUT_S, default 120 (a cold
// dlopen + ncclCommInitRank takes 2 - 6 s on the MI355X boxes, profiles/r03_rccl_init_timeline.txt)
inline double comm_timeout_s() {
    if (const char* e = std::getenv("PTNN_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0.0) return v; }
    return 120.0;
}
// Runs f() on a helper thread and waits for it at most timeout_s.  Returns true when f finished (its value in *result).
// On a timeout the helper is left behind (a blocked dlopen / ncclCommInitRank cannot be cancelled from outside) and the
// caller reports the stage; the process is expected to end soon after such an error.
template <class F>
inline bool run_bounded(F f, double timeout_s, int* result) {
    struct St { std::mutex mu; std::condition_variable cv; bool done = false; int value = 0; };
    auto st = std::make_shared<St>();
    std::thread t([st, f]() mutable {
        const int v = f();
        std::lock_guard<std::mutex> lock(st->mu);
        st->value = v; st->done = true;
        st->cv.notify_all();
    });
    std::unique_lock<std::mutex> lock(st->mu);
    const bool ok = st->cv.wait_for(lock, std::chrono::duration<double>(timeout_s), [&] { return st->done; });
    if (ok) { *result = st->value; lock.unlock(); t.join(); return true; }
    lock.unlock();
    t.detach();
    return false;
}

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;          // optional (ptnn_comm_info): null in a table that does not offer it
};

// A test may install its own function table (tests/native/comm_mock.cpp fills one with in-process fakes that record the call
// order): every later rccl_api() returns it and librccl is never opened.
inline const RcclApi*& rccl_api_override() { static const RcclApi* o = nullptr; return o; }

// The ladder is sharded over the GPUs of ONE node (north_star), so the bootstrap of a communicator never needs a routable
// interface or an InfiniBand probe: NCCL_SOCKET_IFNAME=lo and NCCL_IB_DISABLE=1 take out two stages of ncclGetUniqueId /
// ncclCommInitRank whose duration depends on the box's network set-up rather than on anything this library does.  The LIBRARY
// does not touch the environment (setenv is process-wide and not safe against a concurrent getenv of another thread; a later
// multi-node NCCL user of the same process would silently be pinned to loopback): the callers that own their process set these
// defaults before they start a thread -- distributed.single_node_rccl_env() for LadderGroup, bench.py at start-up -- and
// INTEGRATION.md lists them.  Here the choice is only recorded in the stage log.
inline void rccl_single_node_env() {
    const char* ifn = std::getenv("NCCL_SOCKET_IFNAME");
    const char* ib = std::getenv("NCCL_IB_DISABLE");
    comm_stage("RCCL bootstrap environment: NCCL_SOCKET_IFNAME=%s NCCL_IB_DISABLE=%s", ifn ? ifn : "(unset: RCCL picks an interface)",
               ib ? ib : "(unset)");
}

// dlopen once per process; returns nullptr and fills `why` when the library or a symbol is missing or the load does not finish
// within comm_timeout_s()
inline const RcclApi* rccl_api(std::string& why) {
    if (const RcclApi* o = rccl_api_override()) return o;
    static std::mutex mu;
    static RcclApi api;
    static bool tried = false;
    static std::string err;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        rccl_single_node_env();
        const char* env = std::getenv("PTNN_RCCL_LIBRARY");
        const char* names[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            comm_stage("dlopen(%s)", n);
            int rc = 0;
            // static storage: the helper may outlive this call when the load is abandoned
            static void* loaded = nullptr;
            static std::string dl_err;
            const std::string name = n;
            const bool finished = run_bounded([name]() -> int {
                void* l = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
                if (!l) { const char* e = dlerror(); dl_err = e ? e : "dlopen failed"; return 1; }
                loaded = l;
                return 0;
            }, comm_timeout_s(), &rc);
            if (!finished) { err = "dlopen(" + name + ") did not return within " + std::to_string((int)comm_timeout_s()) + " s"; break; }
            if (rc == 0) { api.lib = loaded; comm_stage("dlopen(%s) done", n); break; }
            err = dl_err;
        }
        if (api.lib) {
            bool ok = true;
#define PTNN_RCCL_SYM(field, name)                                                   \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name));         \
    if (!api.field) { ok = false; err = std::string("librccl has no symbol ") + name; }
            PTNN_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
            PTNN_RCCL_SYM(CommInitRank, "ncclCommInitRank")
            PTNN_RCCL_SYM(CommDestroy, "ncclCommDestroy")
            PTNN_RCCL_SYM(CommAbort, "ncclCommAbort")
            PTNN_RCCL_SYM(AllGather, "ncclAllGather")
            PTNN_RCCL_SYM(Send, "ncclSend")
            PTNN_RCCL_SYM(Recv, "ncclRecv")
            PTNN_RCCL_SYM(GroupStart, "ncclGroupStart")
            PTNN_RCCL_SYM(GroupEnd, "ncclGroupEnd")
            PTNN_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PTNN_RCCL_SYM
            if (ok) api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
            if (!ok) { dlclose(api.lib); api.lib = nullptr; }
        }
    }
    if (!api.lib) { why = err.empty() ? "librccl.so not found" : err; return nullptr; }
    return &api;
}

struct RowMsg {          // one (w, eta) row crossing a GPU boundary
    int is_send, peer, local_row, global_dst;
};

// Which rows `rank` receives and sends for the permutation src (slot k receives the state of slot src[k]); ascending global
// destination slot, so both ends of every pair enumerate their messages in the same order.
inline void route_rows(const int32_t* src, int n_global, int n_local, int rank, std::vector<RowMsg>& out) {
    out.clear();
    const int first = rank * n_local;
    for (int kg = 0; kg < n_global; ++kg) {
        const int s = src[kg];
        const int dst_owner = kg / n_local, src_owner = s / n_local;
        if (dst_owner == src_owner) continue;
        if (dst_owner == rank) out.push_back({0, src_owner, kg - first, kg});
        else if (src_owner == rank) out.push_back({1, dst_owner, s - first, kg});
    }
}

enum { COMM_NONE = 0, COMM_RCCL = 1, COMM_HOST = 2 };

struct Comm {
    int kind = COMM_NONE;
    int rank = 0, nranks = 1;
    int mode = PTNN_XCHG_AUTO;              // as requested; ptnn.hip resolves AUTO per handle
    const RcclApi* api = nullptr;
    ncclComm_t nccl = nullptr;
    ptnn_all_gather_fn h_all_gather = nullptr;
    ptnn_send_recv_fn h_send_recv = nullptr;
    void* h_ctx = nullptr;
    char* stage = nullptr;                  // pinned host staging (host transport)
    size_t stage_bytes = 0;
    int64_t bytes_sent = 0, bytes_received = 0, rounds = 0, calls = 0;
    std::string err;

    bool fail(const std::string& m) { err = m; return false; }
    bool hip_ok(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        return fail(std::string(what) + " failed: " + hipGetErrorString(e));
    }
    bool nccl_ok(ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return true;
        return fail(std::string(what) + " failed: " + (api ? api->GetErrorString(r) : "?"));
    }
    bool need_stage(size_t bytes) {
        if (bytes <= stage_bytes) return true;
        if (stage) (void)hipHostFree(stage);
        stage = nullptr; stage_bytes = 0;
        if (!hip_ok(hipHostMalloc(reinterpret_cast<void**>(&stage), bytes, hipHostMallocDefault), "hipHostMalloc")) return false;
        stage_bytes = bytes;
        return true;
    }

    // buf: device, nranks blocks of bytes_per_rank, this rank's block filled by work already queued on `stream`
    bool all_gather(void* buf, size_t bytes_per_rank, hipStream_t stream) {
        char* base = static_cast<char*>(buf);
        if (kind == COMM_RCCL) {
            comm_stage("ncclAllGather #%lld (%zu B per rank, rank %d of %d)", (long long)calls++, bytes_per_rank, rank, nranks);
            if (!nccl_ok(api->AllGather(base + (size_t)rank * bytes_per_rank, base, bytes_per_rank, ncclChar, nccl, stream), "ncclAllGather"))
                return false;
        } else if (kind == COMM_HOST) {
            const size_t total = bytes_per_rank * (size_t)nranks, mine = (size_t)rank * bytes_per_rank;
            if (!need_stage(total)) return false;
            if (!hip_ok(hipMemcpyAsync(stage + mine, base + mine, bytes_per_rank, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync")) return false;
            if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
            if (h_all_gather(h_ctx, stage, (int64_t)bytes_per_rank) != 0) return fail("the all_gather callback of the host transport failed");
            if (!hip_ok(hipMemcpyAsync(base, stage, total, hipMemcpyHostToDevice, stream), "hipMemcpyAsync")) return false;
            // the staging buffer is reused by the next call: the upload must have left it
            if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
        } else return fail("no communicator");
        bytes_sent += (int64_t)bytes_per_rank * (nranks - 1);
        bytes_received += (int64_t)bytes_per_rank * (nranks - 1);
        return true;
    }

    // msgs in route order; send_ptr(local_row) / recv_ptr(local_row) are device rows of row_bytes bytes
    template <class SendPtr, class RecvPtr>
    bool exchange_rows(const std::vector<RowMsg>& msgs, size_t row_bytes, hipStream_t stream, SendPtr send_ptr, RecvPtr recv_ptr) {
        if (msgs.empty()) return true;
        if (kind == COMM_RCCL) {
            comm_stage("ncclGroupStart/Send/Recv/GroupEnd #%lld (%zu rows of %zu B, rank %d of %d)", (long long)calls++, msgs.size(), row_bytes, rank, nranks);
            if (!nccl_ok(api->GroupStart(), "ncclGroupStart")) return false;
            bool ok = true;
            for (const RowMsg& m : msgs) {
                if (m.is_send) ok = ok && nccl_ok(api->Send(send_ptr(m.local_row), row_bytes, ncclChar, m.peer, nccl, stream), "ncclSend");
                else ok = ok && nccl_ok(api->Recv(recv_ptr(m.local_row), row_bytes, ncclChar, m.peer, nccl, stream), "ncclRecv");
            }
            const std::string first_err = err;
            if (!nccl_ok(api->GroupEnd(), "ncclGroupEnd")) return false;
            if (!ok) return fail(first_err);
        } else if (kind == COMM_HOST) {
            const size_t n = msgs.size();
            if (!need_stage(n * row_bytes)) return false;
            std::vector<int32_t> peer(n), is_send(n);
            std::vector<void*> bufs(n);
            for (size_t k = 0; k < n; ++k) {
                peer[k] = msgs[k].peer; is_send[k] = msgs[k].is_send; bufs[k] = stage + k * row_bytes;
                if (msgs[k].is_send &&
                    !hip_ok(hipMemcpyAsync(bufs[k], send_ptr(msgs[k].local_row), row_bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync"))
                    return false;
            }
            if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
            if (h_send_recv(h_ctx, (int)n, peer.data(), is_send.data(), bufs.data(), (int64_t)row_bytes) != 0)
                return fail("the send_recv callback of the host transport failed");
            for (size_t k = 0; k < n; ++k)
                if (!msgs[k].is_send &&
                    !hip_ok(hipMemcpyAsync(recv_ptr(msgs[k].local_row), bufs[k], row_bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync"))
                    return false;
            if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
        } else return fail("no communicator");
        for (const RowMsg& m : msgs) (m.is_send ? bytes_sent : bytes_received) += (int64_t)row_bytes;
        return true;
    }

    bool failed = false;                    // a collective did not complete: the communicator is aborted, not drained
    void release() {
        if (kind == COMM_RCCL && nccl && api) {
            comm_stage(failed ? "ncclCommAbort" : "ncclCommDestroy");
            if (failed && api->CommAbort) (void)api->CommAbort(nccl);
            else (void)api->CommDestroy(nccl);
            comm_stage("communicator released");
        }
        nccl = nullptr;
        if (stage) (void)hipHostFree(stage);
        stage = nullptr; stage_bytes = 0;
        kind = COMM_NONE;
    }
};

}  // namespace ptnn
