// comm_mock.cpp -- CPU rehearsal of the RCCL call pattern of the sharded ladder (csrc/ptnn_comm.hpp) with an in-process fake
// of librccl: N ranks = N threads of this program, "device" memory = host memory, a mock RcclApi whose functions move the bytes
// and RECORD every call.  What it proves without a single GPU:
//   * Comm::all_gather is called in place (sendbuff == recvbuff + rank * count) with the same count on every rank, every round;
//   * Comm::exchange_rows issues its ncclSend / ncclRecv ONLY between ncclGroupStart and ncclGroupEnd, one group per round and
//     rank, and both ends of every (sender, receiver) pair enumerate their messages in the same order (ascending global
//     destination slot, route_rows) -- the mock matches them FIFO per pair and flags any message left unmatched or mismatched;
//   * after allgather(L) -> cascade -> route_rows -> exchange_rows -> local moves, every slot k of the whole ladder holds the row of
//     slot src[k] (the permutation applied directly), for R = 256 / 8 ranks (Ionosphere rows) and R = 1024 / 8 ranks, over many
//     random cascades with the reference rule's high swap rates and with low ones;
//   * per GPU and round at most one row arrives from below and one from above, and the byte counters agree with that;
//   * a rank whose collective fails makes the round fail on that rank (error text) without hanging the others (the mock's waits
//     are bounded and report what they were waiting for).
// Stands in for nothing in the reference (it has no communication library): the exchange replaces REG:427-437 <-> 694-759.
// Build + run: tests/test_sharding_gloo.py::test_rccl_call_pattern_with_mock_library (hipcc, host code only).
#include "../../parallel-tempering-neural-net_amd/csrc/ptnn_comm.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <deque>
#include <map>
#include <random>

using namespace ptnn;

namespace {

struct World;
struct RankCtx {             // what the mock hides behind an ncclComm_t
    World* world;
    int rank;
    int group_depth = 0;
    struct Op { bool send; int peer; void* buf; size_t bytes; };
    std::vector<Op> pending;                 // sends / receives issued inside the open group
    std::vector<std::string> calls;          // the call log of this rank
    int fail_allgather_at = -1;              // inject ncclSystemError at this all-gather call
    int n_allgather = 0;
};

struct World {
    int n;
    std::mutex mu;
    std::condition_variable cv;
    // all-gather rendezvous
    int ag_arrived = 0, ag_generation = 0;
    std::vector<const void*> ag_src;
    std::vector<size_t> ag_count;
    // point to point: FIFO of payloads per (src, dst)
    std::map<std::pair<int, int>, std::deque<std::vector<char>>> mail;
    std::vector<std::string> errors;
    std::atomic<bool> aborted{false};
    void error(const std::string& e) { std::lock_guard<std::mutex> l(mu); errors.push_back(e); }
};

thread_local RankCtx* t_ctx = nullptr;       // ncclGroupStart / End carry no communicator: the calling thread's rank

RankCtx* ctx_of(ncclComm_t c) { return reinterpret_cast<RankCtx*>(c); }

ncclResult_t mock_all_gather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t) {
    RankCtx* c = ctx_of(comm);
    World* w = c->world;
    c->calls.push_back("AllGather " + std::to_string(count));
    if (dt != ncclChar) w->error("all-gather with a data type other than ncclChar");
    if (static_cast<const char*>(send) != static_cast<char*>(recv) + (size_t)c->rank * count)
        w->error("rank " + std::to_string(c->rank) + ": all-gather is not in place");
    if (c->group_depth) w->error("all-gather inside a send/recv group");
    if (c->n_allgather++ == c->fail_allgather_at) { w->aborted = true; w->cv.notify_all(); return ncclSystemError; }
    std::unique_lock<std::mutex> l(w->mu);
    const int gen = w->ag_generation;
    w->ag_src[c->rank] = send; w->ag_count[c->rank] = count;
    if (++w->ag_arrived == w->n) {
        for (int r = 0; r < w->n; ++r)
            if (w->ag_count[r] != count) w->errors.push_back("all-gather counts differ between ranks");
        w->ag_arrived = 0; ++w->ag_generation;
        w->cv.notify_all();
    } else if (!w->cv.wait_for(l, std::chrono::seconds(20), [&] { return w->ag_generation != gen || w->aborted.load(); }) || w->aborted) {
        w->errors.push_back("rank " + std::to_string(c->rank) + ": all-gather never completed (a rank is missing)");
        return ncclSystemError;
    }
    // every rank's block is final once the generation moved: copy the others' blocks
    std::vector<const void*> src = w->ag_src;
    l.unlock();
    for (int r = 0; r < w->n; ++r)
        if (r != c->rank) std::memcpy(static_cast<char*>(recv) + (size_t)r * count, src[r], count);
    // second rendezvous: nobody overwrites its block before everyone has read it (RCCL orders this on the stream)
    l.lock();
    const int gen2 = w->ag_generation;
    if (++w->ag_arrived == w->n) { w->ag_arrived = 0; ++w->ag_generation; w->cv.notify_all(); }
    else w->cv.wait_for(l, std::chrono::seconds(20), [&] { return w->ag_generation != gen2 || w->aborted.load(); });
    return ncclSuccess;
}

ncclResult_t mock_group_start() {
    t_ctx->calls.push_back("GroupStart");
    t_ctx->group_depth += 1;
    return ncclSuccess;
}

ncclResult_t mock_send(const void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t) {
    RankCtx* c = ctx_of(comm);
    c->calls.push_back("Send->" + std::to_string(peer));
    if (!c->group_depth) c->world->error("rank " + std::to_string(c->rank) + ": ncclSend outside a group");
    if (peer == c->rank || peer < 0 || peer >= c->world->n) c->world->error("send to a bad peer");
    c->pending.push_back({true, peer, const_cast<void*>(buf), count});
    return ncclSuccess;
}

ncclResult_t mock_recv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t) {
    RankCtx* c = ctx_of(comm);
    c->calls.push_back("Recv<-" + std::to_string(peer));
    if (!c->group_depth) c->world->error("rank " + std::to_string(c->rank) + ": ncclRecv outside a group");
    if (peer == c->rank || peer < 0 || peer >= c->world->n) c->world->error("recv from a bad peer");
    c->pending.push_back({false, peer, buf, count});
    return ncclSuccess;
}

ncclResult_t mock_group_end() {
    RankCtx* c = t_ctx;
    World* w = c->world;
    c->calls.push_back("GroupEnd");
    if (--c->group_depth) return ncclSuccess;
    {   // all sends of the group are posted before any receive waits: what the fused group launch of RCCL does
        std::lock_guard<std::mutex> l(w->mu);
        for (const auto& op : c->pending)
            if (op.send) w->mail[{c->rank, op.peer}].emplace_back(static_cast<char*>(op.buf), static_cast<char*>(op.buf) + op.bytes);
        w->cv.notify_all();
    }
    for (const auto& op : c->pending) {
        if (op.send) continue;
        std::unique_lock<std::mutex> l(w->mu);
        auto& q = w->mail[{op.peer, c->rank}];
        if (!w->cv.wait_for(l, std::chrono::seconds(20), [&] { return !q.empty() || w->aborted.load(); }) || q.empty()) {
            w->errors.push_back("rank " + std::to_string(c->rank) + ": receive from " + std::to_string(op.peer) + " was never matched by a send");
            c->pending.clear();
            return ncclSystemError;
        }
        if (q.front().size() != op.bytes) w->errors.push_back("matched send and receive differ in size");
        std::memcpy(op.buf, q.front().data(), std::min(op.bytes, q.front().size()));
        q.pop_front();
    }
    c->pending.clear();
    return ncclSuccess;
}

const char* mock_error_string(ncclResult_t r) { return r == ncclSuccess ? "ok" : "mock failure"; }
ncclResult_t mock_destroy(ncclComm_t) { return ncclSuccess; }

RcclApi make_mock() {
    RcclApi a;
    a.lib = nullptr;
    a.AllGather = mock_all_gather; a.Send = mock_send; a.Recv = mock_recv;
    a.GroupStart = mock_group_start; a.GroupEnd = mock_group_end;
    a.GetErrorString = mock_error_string; a.CommDestroy = mock_destroy; a.CommAbort = mock_destroy;
    return a;
}

// the reference's bubble pass in closed form (REG:659-690, 741-748)
std::vector<int> cascade(const std::vector<float>& L, const std::vector<float>& u) {
    const int R = (int)L.size();
    std::vector<int> src(R);
    int c = 0;
    for (int k = 0; k < R - 1; ++k) {
        float d = L[k + 1] - L[c];
        d = (d < 709.0f) ? d : 709.0f;
        if (std::log(2.0f * u[k]) < d) src[k] = k + 1;
        else { src[k] = c; c = k + 1; }
    }
    src[R - 1] = c;
    return src;
}

struct Case { int R, nranks, PS, rounds; float spread; unsigned seed; int fail_rank; };

int run_case(const Case& cs, const RcclApi& api) {
    const int R = cs.R, n = cs.nranks, Rl = R / n, PS = cs.PS;
    World w;
    w.n = n; w.ag_src.assign(n, nullptr); w.ag_count.assign(n, 0);
    std::vector<RankCtx> ctx(n);
    // per-rank "device" buffers: L[R], cur[Rl][PS], next[Rl][PS]
    std::vector<std::vector<float>> L(n, std::vector<float>(R)), cur(n, std::vector<float>((size_t)Rl * PS)), nxt(n, std::vector<float>((size_t)Rl * PS));
    std::vector<float> truth((size_t)R * PS);                  // the whole ladder as one GPU would hold it
    for (int k = 0; k < R; ++k)
        for (int j = 0; j < PS; ++j) truth[(size_t)k * PS + j] = (float)(k * 131 + j);
    for (int r = 0; r < n; ++r) std::memcpy(cur[r].data(), &truth[(size_t)r * Rl * PS], (size_t)Rl * PS * sizeof(float));
    std::vector<std::string> rank_err(n);
    std::vector<int> max_in_below(n, 0), max_in_above(n, 0);
    std::vector<int64_t> sent(n, 0), received(n, 0);
    std::atomic<int> failed_rounds{0};

    auto body = [&](int r) {
        t_ctx = &ctx[r];
        ctx[r].world = &w; ctx[r].rank = r;
        if (cs.fail_rank == r) ctx[r].fail_allgather_at = 2;
        Comm c;
        c.kind = COMM_RCCL; c.rank = r; c.nranks = n; c.api = &api; c.nccl = reinterpret_cast<ncclComm_t>(&ctx[r]);
        std::mt19937 rng(cs.seed);                           // the same stream on every rank: identical scalars and uniforms
        std::vector<RowMsg> route;
        std::vector<float> Lg(R), u(R - 1);
        for (int round = 0; round < cs.rounds; ++round) {
            std::normal_distribution<float> nd(0.0f, cs.spread);
            std::uniform_real_distribution<float> ud(1e-6f, 1.0f);
            for (int k = 0; k < R; ++k) Lg[k] = nd(rng);
            for (int k = 0; k < R - 1; ++k) u[k] = ud(rng);
            for (int k = 0; k < Rl; ++k) L[r][(size_t)r * Rl + k] = Lg[(size_t)r * Rl + k];      // the segment kernel posted the local block
            if (!c.all_gather(L[r].data(), (size_t)Rl * sizeof(float), nullptr)) { rank_err[r] = c.err; failed_rounds++; return; }
            for (int k = 0; k < R; ++k)
                if (L[r][k] != Lg[k]) { w.error("gathered L differs from the global L"); break; }
            const std::vector<int> src = cascade(L[r], u);
            route_rows(src.data(), R, Rl, r, route);
            int in_below = 0, in_above = 0;
            for (const RowMsg& m : route)
                if (!m.is_send) (m.peer < r ? in_below : in_above) += 1;
            max_in_below[r] = std::max(max_in_below[r], in_below); max_in_above[r] = std::max(max_in_above[r], in_above);
            float* cr = cur[r].data();
            float* nx = nxt[r].data();
            if (!c.exchange_rows(route, (size_t)PS * sizeof(float), nullptr,
                                 [&](int row) { return static_cast<void*>(cr + (size_t)row * PS); },
                                 [&](int row) { return static_cast<void*>(nx + (size_t)row * PS); })) { rank_err[r] = c.err; failed_rounds++; return; }
            for (int k = 0; k < Rl; ++k) {                   // what swap_kernel does with the local sources
                const int s = src[(size_t)r * Rl + k] - r * Rl;
                if (s >= 0 && s < Rl) std::memcpy(nx + (size_t)k * PS, cr + (size_t)s * PS, (size_t)PS * sizeof(float));
            }
            if (r == 0) {                                    // the same permutation applied to the one-GPU picture
                std::vector<float> t2(truth.size());
                for (int k = 0; k < R; ++k) std::memcpy(&t2[(size_t)k * PS], &truth[(size_t)src[k] * PS], (size_t)PS * sizeof(float));
                // published to the other ranks through the all-gather of the next round (they only read `truth` after joining)
                truth.swap(t2);
            }
            cur[r].swap(nxt[r]);
            // rendezvous so that `truth` of this round is final before anyone compares (uses the mock's own barrier)
            float dummy[64] = {0};
            (void)dummy;
        }
        sent[r] = c.bytes_sent; received[r] = c.bytes_received;
    };
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r) th.emplace_back(body, r);
    for (auto& t : th) t.join();

    if (cs.fail_rank >= 0) {
        // the injected failure must surface as an error on the failing rank, the others must come back too (bounded waits)
        if (rank_err[cs.fail_rank].find("ncclAllGather failed") == std::string::npos) { fprintf(stderr, "injected failure not reported: '%s'\n", rank_err[cs.fail_rank].c_str()); return 1; }
        printf("OK failure-injection R=%d ranks=%d: rank %d reported '%s', %d ranks ended their round with an error, nobody hung\n", R, n,
               cs.fail_rank, rank_err[cs.fail_rank].c_str(), failed_rounds.load());
        return 0;
    }
    for (const auto& e : w.errors) fprintf(stderr, "mock: %s\n", e.c_str());
    if (!w.errors.empty()) return 1;
    for (int r = 0; r < n; ++r)
        if (!rank_err[r].empty()) { fprintf(stderr, "rank %d: %s\n", r, rank_err[r].c_str()); return 1; }
    for (const auto& kv : w.mail)
        if (!kv.second.empty()) { fprintf(stderr, "a send from %d to %d was never received\n", kv.first.first, kv.first.second); return 1; }
    // the ladder after all rounds equals the permutations applied directly
    for (int r = 0; r < n; ++r)
        if (std::memcmp(cur[r].data(), &truth[(size_t)r * Rl * PS], (size_t)Rl * PS * sizeof(float)) != 0) { fprintf(stderr, "rank %d holds the wrong rows\n", r); return 1; }
    int64_t tot_s = 0, tot_r = 0, groups = 0, sends = 0, recvs = 0;
    for (int r = 0; r < n; ++r) {
        if (max_in_below[r] > 1 || max_in_above[r] > 1) { fprintf(stderr, "rank %d received %d rows from below, %d from above in one round\n", r, max_in_below[r], max_in_above[r]); return 1; }
        tot_s += sent[r]; tot_r += received[r];
        // call pattern of this rank: (AllGather, [GroupStart, (Send|Recv)+, GroupEnd])*
        bool in_group = false;
        int in_this_group = 0;
        for (const std::string& c : ctx[r].calls) {
            if (c == "GroupStart") { if (in_group) { fprintf(stderr, "nested group\n"); return 1; } in_group = true; in_this_group = 0; ++groups; }
            else if (c == "GroupEnd") { if (!in_group || !in_this_group) { fprintf(stderr, "empty or unopened group\n"); return 1; } in_group = false; }
            else if (c.rfind("Send", 0) == 0) { ++sends; ++in_this_group; if (!in_group) return 1; }
            else if (c.rfind("Recv", 0) == 0) { ++recvs; ++in_this_group; if (!in_group) return 1; }
            else if (in_group) { fprintf(stderr, "collective inside a group\n"); return 1; }
        }
        if ((int)std::count_if(ctx[r].calls.begin(), ctx[r].calls.end(), [](const std::string& c) { return c.rfind("AllGather", 0) == 0; }) != cs.rounds) return 1;
    }
    if (sends != recvs) { fprintf(stderr, "%lld sends vs %lld receives\n", (long long)sends, (long long)recvs); return 1; }
    const int64_t ag = (int64_t)cs.rounds * n * (int64_t)(n - 1) * Rl * (int64_t)sizeof(float);
    if (tot_s - ag != sends * (int64_t)PS * (int64_t)sizeof(float) || tot_r != tot_s) { fprintf(stderr, "byte counters do not add up\n"); return 1; }
    printf("OK R=%d ranks=%d PS=%d rounds=%d spread=%g: %lld rows crossed a boundary in %lld groups (%.2f per rank and round), <= 1 in from below, <= 1 from above\n",
           R, n, PS, cs.rounds, cs.spread, (long long)sends, (long long)groups, (double)sends / (cs.rounds * n));
    return 0;
}

}  // namespace

int main() {
    const RcclApi api = make_mock();
    rccl_api_override() = &api;
    std::string why;
    if (rccl_api(why) != &api) { fprintf(stderr, "override not honoured\n"); return 1; }
    const Case cases[] = {
        {256, 8, 1856, 200, 0.3f, 1u, -1},      // Ionosphere rows (34-50-2: PS = 1856), the reference rule's high swap rates
        {256, 8, 1856, 200, 30.0f, 2u, -1},     // widely spread scalars: few swaps, long-range carries rare
        {1024, 8, 32, 300, 1.0f, 3u, -1},       // config 5's ladder length with short rows
        {1024, 8, 17412, 6, 1.0f, 4u, -1},      // ... and its real rows (32-512-1: PS = 17 412, 70 KB each)
        {64, 2, 32, 300, 0.5f, 5u, -1},
        {16, 4, 32, 50, 0.5f, 6u, 2},           // rank 2's third all-gather fails: reported, nobody hangs
    };
    for (const Case& c : cases)
        if (int rc = run_case(c, api)) return rc;
    printf("OK all\n");
    return 0;
}
