// ptnn_comm.hpp -- transports of the sharded ladder (host side only): RCCL over xGMI, loaded on demand, and a host-staged
// transport behind two callbacks.  What moves and when is decided in ptnn.hip (comm_swap_round); this file only moves bytes.
//
// Stands in for the reference's multiprocessing.Queue / Event star between the parent and its forked chains
// (multicore-pt-regression/pt_timeseries_regression.py:427-437 <-> 694-759).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library itself is dlopen'ed by the first ptnn_comm_init
#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptnn.h"

namespace ptnn {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// dlopen once per process; returns nullptr and fills `why` when the library or a symbol is missing
inline const RcclApi* rccl_api(std::string& why) {
    static std::mutex mu;
    static RcclApi api;
    static bool tried = false;
    static std::string err;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        const char* env = std::getenv("PTNN_RCCL_LIBRARY");
        const char* names[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
            err = dlerror();
        }
        if (api.lib) {
            bool ok = true;
#define PTNN_RCCL_SYM(field, name)                                                   \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name));         \
    if (!api.field) { ok = false; err = std::string("librccl has no symbol ") + name; }
            PTNN_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
            PTNN_RCCL_SYM(CommInitRank, "ncclCommInitRank")
            PTNN_RCCL_SYM(CommDestroy, "ncclCommDestroy")
            PTNN_RCCL_SYM(AllGather, "ncclAllGather")
            PTNN_RCCL_SYM(Send, "ncclSend")
            PTNN_RCCL_SYM(Recv, "ncclRecv")
            PTNN_RCCL_SYM(GroupStart, "ncclGroupStart")
            PTNN_RCCL_SYM(GroupEnd, "ncclGroupEnd")
            PTNN_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PTNN_RCCL_SYM
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
    int64_t bytes_sent = 0, bytes_received = 0, rounds = 0;
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

    void release() {
        if (kind == COMM_RCCL && nccl && api) (void)api->CommDestroy(nccl);
        nccl = nullptr;
        if (stage) (void)hipHostFree(stage);
        stage = nullptr; stage_bytes = 0;
        kind = COMM_NONE;
    }
};

}  // namespace ptnn
