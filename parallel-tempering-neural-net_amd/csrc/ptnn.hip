// ptnn.hip -- host side of libptnn.so: the C ABI of include/ptnn.h over the gfx950 kernels of ptnn_device.hpp.
//
// Replaces, for the hot path only, what the reference does with one forked ptReplica process per chain plus the
// parent's swap loop (REG = multicore-pt-regression/pt_timeseries_regression.py:223-485, 659-771;
// CLS = multicore-pt-classification/pt_classification.py:232-494, 668-776).
#include "ptnn_shapes.hpp"
#include "ptnn_comm.hpp"
#include "ptnn_text.hpp"
#include "../../include/ptnn.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sched.h>

using namespace ptnn;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            (void)hipGetLastError(); /* reported here: must not surface again at the next launch check */   \
            return fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
        }                                                                                                  \
    } while (0)

}  // namespace

// one table per shape, each defined in its own translation unit (ptnn_shape.hip)
#define X_DECL(T, I, O) extern "C" const ptnn::Shape ptnn_shape_##T##_##I##_##O;
PTNN_SHAPES(X_DECL)
#undef X_DECL

namespace {

#define X_ENTRY(T, I, O) &ptnn_shape_##T##_##I##_##O,
const Shape* const g_shapes[] = {PTNN_SHAPES(X_ENTRY)};
#undef X_ENTRY
constexpr int MAX_HIDDEN = MAX_WAVES * WAVE;    // wide nets: one thread per hidden unit

const Shape* find_shape(int task, int I, int O) {
    for (const Shape* s : g_shapes)
        if (s->task == task && s->I == I && s->O == O) return s;
    return nullptr;
}

inline int round_up4(int v) { return (v + 3) & ~3; }

}  // namespace

struct ptnn_handle {
    ptnn_config cfg{};
    const Shape* shape = nullptr;
    hipStream_t stream = nullptr;
    int P = 0, PS = 0, PW = 0, IPY = 0, FWS = 0, Ntr = 0, Nte = 0, nthreads = 64;
    size_t seg_lds = 0, model_lds = 0;
    int model_threads = 64;
    bool speculative = false;
    bool wide = false;              // 64 < H: vectors in HBM, one thread per hidden unit
    bool wide_res = false;          // ... with the state and the proposal resident in LDS (matrix-core layout, 2 vectors fit)
    bool compact = false;           // wide nets with all trace rows resident: rejected steps record a row index, no pos_w row
    bool persistent = false;        // all work-groups of the grid are resident: ptnn_run queues ONE launch, swap rounds inside
    unsigned* d_barrier = nullptr;  // grid barrier of the persistent launch: one slot per work-group
    int barrier_slots = 0;
    bool packed = false;            // H <= 16: packed speculative schedule on one CU
    bool tree = false;              // prefetching tree schedule: groups = 2^depth - 1 work-groups per replica
    bool tree_ahead = false;        // ... with room in LDS for two sets of tapes
    int pk_nred = 3;                // its lane-group width: 2^3 (H <= 8) or 2^4 hidden units
    float* d_wide_scratch = nullptr;
    float* d_xt = nullptr;          // transposed data image for the MFMA forward pass
    void* d_xs = nullptr;           // wide nets: the data image split into three bf16 levels (split-operand forward pass)
    int Npad = 0;
    int fw_mfma = 0;                // cooperative / tree schedule: forward pass on the matrix cores (24 <= H <= 64, I >= 6): 1 exact fp32, 2 split bf16 operands (cooperative only)
    bool xy_global = false;         // split forward pass: no room for the row-major data image in LDS, its rare readers go to global memory
    int groups = 1;                 // work-groups (CUs) per replica in the speculative schedule
    int blocks_per_cu = 0;          // occupancy of the segment kernel as the runtime reports it (0 = not queried)
    unsigned epoch_base = 0;
    int num_cus = 0;
    unsigned long long *d_xslots = nullptr, *d_xw = nullptr, *d_xverdict = nullptr, *d_xswap = nullptr;
    int* d_error = nullptr;
    float *h_stage = nullptr, *d_stage = nullptr;   // initial weights + temperatures on their way to the device (ptnn_set_state)
    int* h_progress = nullptr;      // pinned host word: swap rounds the device has completed (swap_kernel stores it)
    bool failed = false;            // a run on this handle ended in an error (-5 / -7): results are refused until the chains restart
    std::string failure;
    unsigned long long* d_stamps = nullptr;
    bool have_data = false, have_state = false, finalized = false;
    int cap = 0;            // trace ring rows per replica
    int drained = 0;        // rows [0, drained] have been fetched by the caller (streaming mode)
    int first_row = 0;      // trace rows below this one are not on this device (chains restored from a checkpoint)
    int cur = 0;            // next MH step index
    int rounds_done = 0;    // swap rounds counted (including the phantom one)
    int max_rounds = 0;
    int flip = 0;           // which state buffer is current
    // device memory
    float* d_data = nullptr;
    float* d_state[2] = {nullptr, nullptr};
    float *d_rec_w = nullptr, *d_st_f = nullptr, *d_temps = nullptr;
    float* d_gd_w[2] = {nullptr, nullptr};
    int* d_gd_valid[2] = {nullptr, nullptr};
    int* d_st_i = nullptr;
    float *d_L_handoff = nullptr, *d_L_final = nullptr;
    float *d_L_raw = nullptr, *d_prior_post = nullptr, *d_temps_global = nullptr;   // swap_rule 1
    bool have_ladder = false;
    int *d_label[2] = {nullptr, nullptr}, *d_slot_of[2] = {nullptr, nullptr};   // label_swap: slot <-> temperature maps, ping-pong
    int lflip = 0;
    float *d_pos_w = nullptr;       // [Rl][cap][PW]
    float *d_scal = nullptr;        // [Rl][cap][TR_COUNT] scalar trace rows
    int *d_src = nullptr, *d_src_log = nullptr;
    int* h_src = nullptr;
    float* d_xchg = nullptr;                                // [R_global][XS] exchange rows of the gathered sharding mode                                   // pinned staging for the permutation of a round (sharded ladder)
    long long* d_counters = nullptr;
    // sharded ladder: transport and what a swap round moves through it
    Comm comm;
    std::vector<RowMsg> route;
    // trace images on the host (ptnn_trace_image*): pinned copies of d_pos_w / d_scal that a second stream fills while the chains
    // go on sampling
    hipStream_t copy_stream = nullptr;
    float *h_img_pos = nullptr, *h_img_rows = nullptr;
    std::vector<hipEvent_t> img_events;                     // ticket k: the copy of its rows has landed
    // kernel timing (HIP events on our stream)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing;
    size_t timing_used = 0;
    int timing_stride = 1;          // PTNN_TIMING_STRIDE: 0 = never, n = every n-th segment launch
    long long launch_count = 0;
    int64_t timed_launches = 0;
    double timed_ms = 0.0;

    SegParams seg_params() const {
        SegParams p{};
        p.H = cfg.n_hidden; p.P = P; p.PS = PS;
        p.Ntr = Ntr; p.Nte = Nte; p.IPY = IPY; p.FWS = FWS;
        p.S = cfg.n_samples; p.switch_step = cfg.pt_switch_step; p.use_lg = cfg.use_langevin;
        p.trace_cap = cap;
        p.first_global = cfg.first_global_replica;
        p.l_prob = cfg.l_prob; p.lr = cfg.learn_rate; p.step_w = cfg.step_w; p.step_eta = cfg.step_eta;
        p.inv_2sig2 = 1.0f / (2.0f * cfg.sigma_squared);
        const int I = cfg.n_in, H = cfg.n_hidden, O = cfg.n_out;
        // part1 of prior_likelihood: REG uses d*h + h + 2 (REG:218), CLS d*h + h + o + h*o (CLS:227)
        const double cnt = (cfg.task == PTNN_TASK_REG) ? (double)(I * H + H + 2) : (double)(I * H + H + O + H * O);
        p.prior_c = (float)(-1.0 * (cnt / 2.0) * std::log((double)cfg.sigma_squared));
        p.nu1 = cfg.nu_1; p.nu2 = cfg.nu_2;
        p.seed_lo = (uint32_t)(cfg.seed & 0xffffffffull); p.seed_hi = (uint32_t)(cfg.seed >> 32);
        p.data = d_data; p.w_state = d_state[flip]; p.rec_w = d_rec_w; p.gd_w = d_gd_w[flip]; p.gd_valid = d_gd_valid[flip];
        p.st_f = d_st_f; p.st_i = d_st_i; p.temps = d_temps;
        p.L_handoff = d_L_handoff; p.L_final = d_L_final;
        p.L_raw = (cfg.swap_rule == 1) ? d_L_raw : nullptr; p.prior_post = d_prior_post;
        p.tr_pos_w = d_pos_w; p.tr_scal = d_scal; p.PW = PW;
        p.G = groups; p.epoch_base = epoch_base; p.xslots = d_xslots; p.xw = d_xw; p.xverdict = d_xverdict; p.error_flag = d_error; p.stamps = d_stamps; p.wide_scratch = d_wide_scratch; p.noise_shared = cfg.shared_noise ? 1 : 0; p.pk_nred = pk_nred; p.xt = d_xt; p.xs = reinterpret_cast<const uint4*>(d_xs); p.Npad = Npad; p.fw_mfma = fw_mfma; p.xy_global = xy_global ? 1 : 0; p.forward_bf16 = cfg.forward_bf16 == 1 ? 1 : 0; p.tree_ahead = tree_ahead ? 1 : 0; p.compact = compact ? 1 : 0;
        {
            // records through the XCD's L2: asked for only where xcd_block (ptnn_device.hpp) can put a replica's work-groups on one XCD --
            // a grid of 8 k blocks with k a multiple of the groups per replica; elsewhere the in-kernel handshake could only time out
            const char* e = std::getenv("PTNN_XCD_GRANULES");
            const int grid_ = cfg.n_replicas_local * groups;
            const bool can = groups > 1 && (grid_ & 7) == 0 && ((grid_ >> 3) % groups) == 0;
            p.xcd_granules = ((e && e[0] == '0') || cfg.shared_device || !can) ? 0 : 1;
        }
        p.xswap = d_xswap;
        // wide nets over several work-groups: a window of 16 steps lets the groups balance Langevin (10 units) against random-walk
        // (1) steps (measured on config 5: 8 steps 0.680 M, 12: 0.692 M, 16: 0.698 M samples/s; wide nets accept 1 - 5 %, so little of
        // a window is thrown away); random-walk-only runs have nothing to balance and a longer window only wastes what follows an accept
        p.wide_window = cfg.use_langevin ? 16 : groups;
        if (const char* e = std::getenv("PTNN_WIDE_WINDOW")) p.wide_window = std::atoi(e);   // experiments
        return p;
    }
};

namespace {

int wait_stream(ptnn_handle* h);

inline int tree_depth(int groups) { int d = 0; while ((1 << (d + 1)) - 1 <= groups) ++d; return d; }   // groups = 2^d - 1

// Q10: REG hands off after step i when i % si == 0 and i != 0 (REG:427); CLS when (i+1) % si == 0 (CLS:438)
inline bool swap_trigger(const ptnn_config& c, int i) {
    if (c.task == PTNN_TASK_REG) return (i % c.swap_interval == 0) && i != 0;
    return ((i + 1) % c.swap_interval) == 0;
}

// the segment kernel the schedule resolved to
seg_fn segment_function(const ptnn_handle* h) {
    const Shape* sh = h->shape;
    if (h->wide) return h->wide_res ? sh->seg_wide_res : sh->seg_wide;
    if (h->tree) return sh->tree;
    if (h->packed) return h->groups > 1 ? sh->packm : sh->pack;
    return h->speculative ? sh->spec : sh->seg;
}

void collect_timing(ptnn_handle* h) {
    for (size_t k = 0; k < h->timing_used; ++k) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->timing[k].first, h->timing[k].second) == hipSuccess) {
            h->timed_ms += ms;
            h->timed_launches += 1;
        }
    }
    h->timing_used = 0;
}

void fill_swap_params(ptnn_handle* h, bool phantom, SwapParams& sp) {
    sp.R = h->cfg.n_replicas_global; sp.Rl = h->cfg.n_replicas_local; sp.first_global = h->cfg.first_global_replica;
    sp.PS = h->PS;
    sp.seed_lo = (uint32_t)(h->cfg.seed & 0xffffffffull); sp.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    sp.L = phantom ? h->d_L_final : h->d_L_handoff;
    sp.cur = h->d_state[h->flip]; sp.next = h->d_state[h->flip ^ 1];
    sp.gd_cur = h->d_gd_w[h->flip]; sp.gd_next = h->d_gd_w[h->flip ^ 1];
    sp.gd_valid_cur = h->d_gd_valid[h->flip]; sp.gd_valid_next = h->d_gd_valid[h->flip ^ 1];
    sp.src_out = nullptr;
    sp.counters = h->d_counters; sp.src_log = h->d_src_log; sp.log_capacity = h->max_rounds;
    sp.rule = h->cfg.swap_rule; sp.L_raw = h->d_L_raw; sp.prior_post = h->d_prior_post; sp.temps_global = h->d_temps_global;
    sp.st_f = h->d_st_f;
    sp.canonical = (h->cfg.pt_switch_step >= 0 && h->cur - 1 >= h->cfg.pt_switch_step) ? 1 : 0;
    sp.xchg = h->d_xchg; sp.XS = xchg_row_floats(h->PS); sp.L_stride = 1;
    sp.label_mode = h->cfg.label_swap ? 1 : 0;
    sp.label_cur = h->d_label[h->lflip]; sp.slot_cur = h->d_slot_of[h->lflip];
    sp.label_next = h->d_label[h->lflip ^ 1]; sp.slot_next = h->d_slot_of[h->lflip ^ 1];
    sp.temps_local = h->d_temps;
    sp.progress = nullptr;
}

// MH steps [begin, end) in one launch.  swap_inside: the swap rounds between the intervals run inside it (persistent launch:
// every work-group resident, grid barriers); otherwise [begin, end) is one interval and the caller queues swap_kernel behind it.
int launch_segment(ptnn_handle* h, int begin, int end, bool swap_inside = false, bool round_follows = false) {
    if (end <= begin) return 0;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    SegParams p = h->seg_params();
    if (round_follows && h->comm.kind == COMM_RCCL) { p.seg_progress = h->h_progress + 1; p.seg_ordinal = h->rounds_done + 1; }
    PersistParams pp{};
    pp.end = end; pp.swap_inside = swap_inside ? 1 : 0; pp.task = h->cfg.task; pp.si = h->cfg.swap_interval;
    pp.round0 = h->rounds_done; pp.flip0 = h->flip; pp.lflip0 = h->lflip;
    const int grid = h->cfg.n_replicas_local * ((h->speculative || h->tree || h->wide) ? h->groups : 1);   // packed: groups > 1 only for the multi-CU variant
    pp.nblocks = grid; pp.barrier = h->d_barrier;
    for (int b = 0; b < 2; ++b) {
        pp.state[b] = h->d_state[b]; pp.gd[b] = h->d_gd_w[b]; pp.gd_valid[b] = h->d_gd_valid[b];
        pp.label[b] = h->d_label[b]; pp.slot_of[b] = h->d_slot_of[b];
    }
    fill_swap_params(h, false, pp.sp);
    if (swap_inside) {
        if (grid > h->barrier_slots) {
            if (h->d_barrier) HIP_TRY(hipFree(h->d_barrier));
            h->d_barrier = nullptr;
            HIP_TRY(hipMalloc(&h->d_barrier, (size_t)grid * sizeof(unsigned)));
            h->barrier_slots = grid;
            pp.barrier = h->d_barrier;
        }
        HIP_TRY(hipMemsetAsync(h->d_barrier, 0, (size_t)grid * sizeof(unsigned), h->stream));
    }
    // event pairs around a launch cost a pipeline bubble each; time every timing_stride-th launch only
    const bool timed = h->timing_stride > 0 && (h->launch_count++ % h->timing_stride) == 0;
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (timed) {
        if (h->timing_used == h->timing.size()) {
            if (h->timing.size() >= 4096) {               // keep the pool bounded: drain it (synchronises)
                if (int rc = wait_stream(h)) return rc;
                collect_timing(h);
            } else {
                hipEvent_t a, b;
                HIP_TRY(hipEventCreate(&a));
                HIP_TRY(hipEventCreate(&b));
                h->timing.emplace_back(a, b);
            }
        }
        ev = &h->timing[h->timing_used++];
        HIP_TRY(hipEventRecord(ev->first, h->stream));
    }
    hipLaunchKernelGGL(segment_function(h), dim3(grid), dim3(h->nthreads), h->seg_lds, h->stream, p, pp, begin);
    h->epoch_base += (unsigned)(end - begin) + 1u + (swap_inside ? (unsigned)((end - begin) / h->cfg.swap_interval + 2) : 0u);   // granule tags never repeat across launches
    HIP_TRY(hipGetLastError());
    if (ev) HIP_TRY(hipEventRecord(ev->second, h->stream));
    return 0;
}

// mode bit 0 = apply moves (and flip), bit 1 = count + log, bit 2 = L and the source rows come from the gathered exchange
// buffer; src_out optional.  mode -1 = pack the exchange rows of the local replicas.
int launch_swap(ptnn_handle* h, bool phantom, int mode, bool want_src) {
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    SwapParams sp{};
    fill_swap_params(h, phantom, sp);
    sp.src_out = want_src ? h->d_src : nullptr;
    sp.progress = (mode >= 0 && (mode & 2)) ? h->h_progress : nullptr;     // the counting pass of a round is its last kernel
    if (mode == -1) {
        hipLaunchKernelGGL(xchg_pack_kernel, dim3(sp.Rl), dim3(64), 0, h->stream, sp);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (mode & 4) {
        sp.L = h->d_xchg + 2 * h->PS + 1; sp.L_stride = sp.XS;
        sp.L_raw = h->d_xchg + 2 * h->PS + 2; sp.prior_post = h->d_xchg + 2 * h->PS + 3;
    }
    const size_t lds = (size_t)(3 * sp.R + 1) * sizeof(float);        // L, ln 2u, src (+ count)
    const int swap_threads = (sp.R <= 512 && sp.PS <= 256) ? 64 : 256;     // more threads for long ladders and long rows
    hipLaunchKernelGGL(swap_kernel, dim3(sp.Rl), dim3(swap_threads), lds, h->stream, sp, h->rounds_done, mode);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The dynamic-LDS ceiling of a kernel is a property of the function, shared by every handle of the process: only ever raise
// it, so that a handle created later with a smaller data set does not pull it below what an earlier one launches with.
int raise_lds_limit(const void* func, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> limit;
    if (bytes <= 64 * 1024) return 0;
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = limit[func];
    if (bytes > cur) {
        HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        cur = bytes;
    }
    return 0;
}

// Wait for everything queued on the handle's stream.  With an RCCL communicator attached the wait is bounded: a collective
// whose peer never arrives would otherwise block the host for ever (the reference's parent at least polls is_alive() every
// round, REG:721-727).  "No progress" = the stream is busy and the device has not completed a swap round (swap_kernel stores the
// round count into a pinned host word) for comm_timeout_s() seconds; a long segment between two rounds is far below that.
int wait_stream(ptnn_handle* h) {
    if (h->comm.kind != COMM_RCCL) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        return 0;
    }
    const double limit = comm_timeout_s();
    volatile int* prog = h->h_progress;      // [0] swap rounds completed, [1] index + 1 of the round whose segment has ended
    int seen0 = prog[0], seen1 = prog[1];
    double t_seen = comm_clock();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail(-2, "hipStreamQuery failed: %s", hipGetErrorString(q));
        const int now0 = prog[0], now1 = prog[1];
        // the clock runs only while a collective is at the head of the stream: the segment before round k has ended (prog[1] ==
        // k + 1) and the round has not (prog[0] == k).  A segment, however long, is bounded by its own kernel spins.
        if (now0 != seen0 || now1 != seen1 || now1 <= now0) { seen0 = now0; seen1 = now1; t_seen = comm_clock(); }
        else if (comm_clock() - t_seen > limit) {
            h->failed = true; h->comm.failed = true;
            h->failure = "no progress on the handle's stream for " + std::to_string((int)limit) + " s inside swap round " + std::to_string(now0) +
                         " (" + std::to_string(h->rounds_done) + " queued); last communicator stage: " + comm_last_stage();
            return fail(-7, "%s", h->failure.c_str());
        }
        if (spins < 4096) sched_yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

// wait_stream + the device's error flag: a bounded spin that expired inside a segment kernel invalidates the run
int finish_stream(ptnn_handle* h) {
    if (h->failed) return fail(h->failure.find("no progress") == 0 ? -7 : -5, "%s", h->failure.c_str());
    if (int rc = wait_stream(h)) return rc;
    int err = 0;
    HIP_TRY(hipMemcpy(&err, h->d_error, sizeof(int), hipMemcpyDeviceToHost));
    if (err) {
        h->failed = true;
        h->failure = "a cross-work-group hand-off timed out inside the segment kernel (" + std::to_string(err) +
                     " work-groups gave up); the run is invalid -- restart the chains (ptnn_set_state / ptnn_checkpoint_load); schedules "
                     "with several work-groups per replica expect the GPU to themselves";
        return fail(-5, "%s", h->failure.c_str());
    }
    return 0;
}

// One launch per run needs every work-group of the grid resident at once (they meet at grid barriers) and room in LDS for the
// cascade of a swap round.  Decided per handle once the schedule is known.  Taken by default where it is measured to pay: one
// work-group per replica (packed, cooperative, one-group wide: one barrier per round; Sunspot + 2.6 %, Ionosphere + 0.2 %); with
// several work-groups per replica a round needs a second rendezvous and the launch boundary it replaces is cheaper (Iris tree
// - 5 %, Mackey-Glass - 2 %, profiles/r03_persistent_ab.json).  $PTNN_PERSISTENT=0: never; =1: wherever resident.
// Granules of the swap rounds a multi-group launch runs by itself (ptnn_device.hpp: segment_tree_body, segment_pack_body<MULTI>): two
// parities of swap_xchg_granules(R, row); only when the whole ladder is on this handle (a sharded ladder exchanges through its communicator)
int alloc_swap_granules(ptnn_handle* h, int row) {
    if (h->d_xswap) { HIP_TRY(hipFree(h->d_xswap)); h->d_xswap = nullptr; }
    if (h->cfg.n_replicas_local != h->cfg.n_replicas_global) return 0;
    const size_t n = 2 * swap_xchg_granules(h->cfg.n_replicas_global, row);
    HIP_TRY(hipMalloc(&h->d_xswap, n * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->d_xswap, 0, n * sizeof(unsigned long long)));
    return 0;
}

int resolve_persistent(ptnn_handle* h) {
    h->persistent = false;
    const char* e = std::getenv("PTNN_PERSISTENT");
    if ((e && e[0] == '0') || h->cfg.shared_device) return 0;     // grid barriers want every work-group resident: not on a shared GPU
    const int G = ((h->speculative || h->tree || h->wide) ? h->groups : 1);
    // The prefetching tree runs its swap rounds inside the launch by itself (segment_tree_body: the root groups exchange scalars and
    // state rows as granules, no grid barrier): the reference's cascade without label swapping, a ladder that is not sharded, and as
    // many replicas as the cascade has room for in the record area of LDS.  One launch per run then, unless $PTNN_PERSISTENT=0.
    const bool whole = h->cfg.swap_rule == 0 && !h->cfg.label_swap && h->d_xswap != nullptr && h->cfg.n_replicas_local == h->cfg.n_replicas_global;
    const bool tree_inside = h->tree && whole && h->cfg.n_replicas_global <= TREE_PERSIST_MAX_R;
    // ... and so does the packed round over several CUs (segment_pack_body<MULTI>; the cascade's 3 R + 1 floats live in the slots' area)
    const bool packm_inside = h->packed && h->groups > 1 && whole &&
                              (size_t)(3 * h->cfg.n_replicas_global + 1) <= (size_t)pack_slots(h->pk_nred) * pack_slot_floats(h->PS);
    const bool own_rounds = tree_inside || packm_inside;
    if (G > 1 && !own_rounds && !(e && e[0] == '1')) return 0;
    // One barrier per round (G == 1) leaves the posted scalars single-buffered: a work-group that has left the barrier reads all R of
    // them into LDS at once (cascade_lds), and the next write to any of them comes a whole swap interval later, at the end of the
    // writer's next interval.  The invariant "no resident work-group falls a whole interval behind between leaving a barrier and its
    // next few loads" holds with orders of magnitude to spare for intervals of tens of microseconds; for intervals of a few MH steps
    // of a small net it is not worth relying on: those runs take one launch per interval (a kernel boundary orders everything).
    if (h->cfg.swap_interval < 8 && !own_rounds && !(e && e[0] == '1')) return 0;
    // kernels compiled without the interval loop (ptnn_device.hpp: persistent_loop<false>)
    if ((h->speculative && !h->packed) || (h->tree && !tree_inside) || (h->packed && h->groups > 1 && !packm_inside)) return 0;
    if (h->packed && h->groups == 1 && !(h->shape->loops & 2)) return 0;
    if (!h->wide && !h->packed && !h->speculative && !h->tree && !(h->shape->loops & 1)) return 0;
    const size_t swap_lds = (size_t)(3 * h->cfg.n_replicas_global + 1) * sizeof(float);
    if (swap_lds > h->seg_lds) {
        if (swap_lds > 152 * 1024) return 0;
        h->seg_lds = swap_lds;
    }
    const void* fn = reinterpret_cast<const void*>(segment_function(h));
    if (int rc = raise_lds_limit(fn, h->seg_lds)) return rc;
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, h->nthreads, h->seg_lds));
    const long long grid = (long long)h->cfg.n_replicas_local * ((h->speculative || h->tree || h->wide) ? h->groups : 1);
    h->persistent = grid <= (long long)per_cu * h->num_cus;
    return 0;
}

int check_ready(ptnn_handle* h) {
    if (!h) return fail(-1, "null handle");
    if (!h->have_data) return fail(-1, "ptnn_set_data has not been called");
    if (!h->have_state) return fail(-1, "ptnn_set_state has not been called");
    return 0;
}

}  // namespace

namespace {
// everything of ptnn_create that can fail after the handle exists: the caller destroys the handle on a non-zero return
int create_buffers(ptnn_handle* h, const ptnn_config* cfg, const Shape* sh, const hipDeviceProp_t& prop) {
    h->cfg = *cfg;
    h->shape = sh;
    h->num_cus = prop.multiProcessorCount;
    if (const char* ts = std::getenv("PTNN_TIMING_STRIDE")) h->timing_stride = std::atoi(ts);
    const int I = cfg->n_in, H = cfg->n_hidden, O = cfg->n_out;
    h->P = I * H + H * O + H + O;
    h->PS = round_up4(h->P + 1);
    h->PW = (h->P + 15) & ~15;                             // pos_w trace rows: whole 64-byte sectors
    h->IPY = round_up4(I + 2);                             // x[I], y, then 1 + x[n].x[n-1] for the pipelined SGD epoch
    h->FWS = round_up4(I + 1 + O);
    h->max_rounds = cfg->n_samples / cfg->swap_interval + 2;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->cap = (cfg->trace_capacity > 0 && cfg->trace_capacity < cfg->n_samples) ? cfg->trace_capacity : cfg->n_samples;
    const size_t Rl = cfg->n_replicas_local, R = cfg->n_replicas_global, S = h->cap;
    HIP_TRY(hipMalloc(&h->d_state[0], Rl * h->PS * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_state[1], Rl * h->PS * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_rec_w, Rl * h->PS * sizeof(float)));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipMalloc(&h->d_gd_w[b], Rl * h->PS * sizeof(float)));
        HIP_TRY(hipMalloc(&h->d_gd_valid[b], Rl * sizeof(int)));
    }
    HIP_TRY(hipMalloc(&h->d_st_f, Rl * SF_COUNT * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_st_i, Rl * SI_COUNT * sizeof(int)));
    HIP_TRY(hipMalloc(&h->d_temps, Rl * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_L_handoff, R * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_L_final, R * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_L_raw, R * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_prior_post, R * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_temps_global, R * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_pos_w, Rl * S * h->PW * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_scal, Rl * S * TR_COUNT * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_src, R * sizeof(int)));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipMalloc(&h->d_label[b], R * sizeof(int)));
        HIP_TRY(hipMalloc(&h->d_slot_of[b], R * sizeof(int)));
    }
    HIP_TRY(hipHostMalloc(&h->h_src, R * sizeof(int), hipHostMallocDefault));
    HIP_TRY(hipMalloc(&h->d_xchg, (size_t)R * xchg_row_floats(h->PS) * sizeof(float)));
    HIP_TRY(hipMemsetAsync(h->d_xchg, 0, (size_t)R * xchg_row_floats(h->PS) * sizeof(float), h->stream));
    HIP_TRY(hipMalloc(&h->d_src_log, (size_t)h->max_rounds * R * sizeof(int)));
    HIP_TRY(hipMalloc(&h->d_counters, 2 * sizeof(long long)));
    HIP_TRY(hipMalloc(&h->d_error, sizeof(int)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), (Rl * h->P + Rl) * sizeof(float), hipHostMallocDefault));
    HIP_TRY(hipMalloc(&h->d_stage, (Rl * h->P + Rl) * sizeof(float)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_progress), 2 * sizeof(int), hipHostMallocDefault));
    h->h_progress[0] = h->h_progress[1] = 0;
    HIP_TRY(hipMalloc(&h->d_stamps, 160 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(h->d_stamps, 0, 160 * sizeof(unsigned long long), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_error, 0, sizeof(int), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_counters, 0, 2 * sizeof(long long), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_L_handoff, 0, R * sizeof(float), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_L_final, 0, R * sizeof(float), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_src_log, 0xff, (size_t)h->max_rounds * R * sizeof(int), h->stream));
    return 0;
}
}  // namespace

extern "C" {

int ptnn_abi_version(void) { return PTNN_ABI_VERSION; }
const char* ptnn_last_error(void) { return g_err.c_str(); }

int ptnn_supports(int task, int n_in, int n_hidden, int n_out) {
    return (find_shape(task, n_in, n_out) != nullptr && n_hidden >= 1 && n_hidden <= MAX_HIDDEN) ? 1 : 0;
}

int ptnn_create(const ptnn_config* cfg, ptnn_handle** out) {
    if (!cfg || !out) return fail(-1, "null argument");
    if (cfg->struct_bytes != (int32_t)sizeof(ptnn_config))
        return fail(-1, "ptnn_config size mismatch: caller %d, library %d", cfg->struct_bytes, (int)sizeof(ptnn_config));
    if (cfg->task != PTNN_TASK_REG && cfg->task != PTNN_TASK_CLS) return fail(-1, "unknown task %d", cfg->task);
    if (cfg->n_in < 1 || cfg->n_hidden < 1 || cfg->n_out < 1) return fail(-1, "bad topology");
    if (cfg->task == PTNN_TASK_REG && cfg->n_out != 1) return fail(-1, "regression needs n_out == 1");
    const Shape* sh = find_shape(cfg->task, cfg->n_in, cfg->n_out);
    if (!sh)
        return fail(-3, "no gfx950 kernel compiled for task=%d n_in=%d n_out=%d: add it to PTNN_SHAPES and rebuild",
                    cfg->task, cfg->n_in, cfg->n_out);
    if (cfg->n_hidden > MAX_HIDDEN)
        return fail(-3, "n_hidden=%d > %d: the SGD sweep holds one hidden unit per thread of one work-group", cfg->n_hidden, MAX_HIDDEN);
    if (cfg->n_replicas_local < 1 || cfg->n_replicas_global < 2 || cfg->first_global_replica < 0 ||
        cfg->first_global_replica + cfg->n_replicas_local > cfg->n_replicas_global)
        return fail(-1, "bad replica partition: local=%d global=%d first=%d", cfg->n_replicas_local,
                    cfg->n_replicas_global, cfg->first_global_replica);
    if (cfg->n_samples < 2) return fail(-1, "n_samples must be >= 2");
    if (cfg->swap_interval < 1) return fail(-1, "swap_interval must be >= 1 (the reference divides by it, REG:427)");
    if (cfg->swap_rule != 0 && cfg->swap_rule != 1) return fail(-1, "swap_rule must be 0 (reference cascade) or 1 (even/odd Metropolis)");
    if (cfg->label_swap != 0 && cfg->label_swap != 1) return fail(-1, "label_swap must be 0 or 1");
    if (cfg->shared_device != 0 && cfg->shared_device != 1) return fail(-1, "shared_device must be 0 or 1");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(-2, "device %d not present (%d devices)", cfg->device_id, ndev);
    HIP_TRY(hipSetDevice(cfg->device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(-2, "device %d is %s; libptnn is built for gfx950 only", cfg->device_id, prop.gcnArchName);

    if (cfg->trace_capacity < 0 || (cfg->trace_capacity > 0 && cfg->trace_capacity < 2))
        return fail(-1, "trace_capacity must be 0 (= n_samples) or >= 2");
    ptnn_handle* h = new ptnn_handle();
    if (int rc = create_buffers(h, cfg, sh, prop)) {       // frees whatever was allocated (stream included); g_err keeps the cause
        const std::string why = g_err;
        ptnn_destroy(h);
        g_err = why;
        return rc;
    }
    *out = h;
    return 0;
}

int ptnn_destroy(ptnn_handle* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device_id);
    if (h->stream && !h->comm.failed) (void)wait_stream(h);
    h->comm.release();                                       // a failed communicator is aborted: its kernels leave the stream
    if (h->comm.failed && h->stream) {
        // give the aborted collective a few seconds to drain; if the stream still does not empty, leak the handle rather than
        // block in hipFree for ever
        const double t0 = comm_clock();
        while (hipStreamQuery(h->stream) == hipErrorNotReady && comm_clock() - t0 < 5.0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (hipStreamQuery(h->stream) == hipErrorNotReady) return fail(-7, "the stream of a failed communicator did not drain: handle leaked");
    }
    void* ptrs[] = {h->d_data, h->d_state[0], h->d_state[1], h->d_rec_w, h->d_gd_w[0], h->d_gd_w[1], h->d_gd_valid[0], h->d_gd_valid[1], h->d_st_f, h->d_st_i, h->d_temps,
                    h->d_L_handoff, h->d_L_final, h->d_L_raw, h->d_prior_post, h->d_temps_global, h->d_pos_w, h->d_scal, h->d_src, h->d_label[0], h->d_label[1], h->d_slot_of[0], h->d_slot_of[1], h->d_src_log, h->d_counters, h->d_error, h->d_xslots, h->d_xw, h->d_xverdict, h->d_xswap, h->d_stamps, h->d_wide_scratch, h->d_xt, h->d_xs, h->d_barrier};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->h_src) (void)hipHostFree(h->h_src);
    if (h->h_progress) (void)hipHostFree(h->h_progress);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->d_xchg) (void)hipFree(h->d_xchg);
    for (auto& ev : h->timing) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    for (auto& ev : h->img_events) (void)hipEventDestroy(ev);
    if (h->h_img_pos) (void)hipHostFree(h->h_img_pos);
    if (h->h_img_rows) (void)hipHostFree(h->h_img_rows);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int ptnn_set_data(ptnn_handle* h, const float* train, int ntr, const float* test, int nte, int ncols) {
    if (!h || !train || !test) return fail(-1, "null argument");
    const int I = h->cfg.n_in;
    if (ncols < I + 1) return fail(-1, "data needs at least n_in + 1 = %d columns, got %d", I + 1, ncols);
    if (ntr < 1 || nte < 1) return fail(-1, "empty data set");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    const int Nall = ntr + nte;
    const int IPY = h->IPY;
    std::vector<float> packed((size_t)(Nall + 2) * IPY, 0.0f);      // two zero rows: look-ahead of the SGD sweep
    for (int n = 0; n < Nall; ++n) {
        const float* row = (n < ntr) ? train + (size_t)n * ncols : test + (size_t)(n - ntr) * ncols;
        for (int c = 0; c <= I; ++c) packed[(size_t)n * IPY + c] = row[c];
        if (n > 0) {                                                 // see sgd_sweep: z[n] = zpart + lhd[n-1] * (1 + x[n].x[n-1])
            float d = 1.0f;
            for (int c = 0; c < I; ++c) d = std::fmaf(row[c], packed[(size_t)(n - 1) * IPY + c], d);
            packed[(size_t)n * IPY + I + 1] = d;
        }
        if (h->cfg.task == PTNN_TASK_CLS) {
            const float y = row[I];
            if (!(y >= 0.0f) || y >= (float)h->cfg.n_out || y != std::floor(y))
                return fail(-1, "class label %g in row %d is not an integer in [0, %d)", (double)y, n, h->cfg.n_out);
        }
    }
    const int H = h->cfg.n_hidden;
    // transposed image Xt[k][Npad] for the MFMA forward passes (rows = data rows are the lanes of the B operand)
    h->Npad = (Nall + 31) & ~31;
    {
        std::vector<float> xt((size_t)I * h->Npad, 0.0f);
        for (int n = 0; n < Nall; ++n)
            for (int k = 0; k < I; ++k) xt[(size_t)k * h->Npad + n] = packed[(size_t)n * IPY + k];
        if (h->d_xt) { HIP_TRY(hipFree(h->d_xt)); h->d_xt = nullptr; }
        HIP_TRY(hipMalloc(&h->d_xt, xt.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->d_xt, xt.data(), xt.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    h->fw_mfma = 0; h->xy_global = false;
    if (h->d_xs) { HIP_TRY(hipFree(h->d_xs)); h->d_xs = nullptr; }
    if (H > WAVE && H % 32 == 0 && h->shape->split_ch > 0 && h->cfg.forward_bf16 == 0) {
        // split-operand forward pass of the wide kernels (ptnn_device.hpp, SplitK / eval_rows_mfma_wsplit): x = hi + mid + lo, three
        // bf16 roundings (nearest even; x - hi and x - hi - mid are exact in fp32), rows of 8 * CH bf16, k contiguous
        const char* fs = std::getenv("PTNN_FW_SPLIT");
        if (!(fs && fs[0] == '0')) {
            const int CH = h->shape->split_ch, KBF = 8 * CH;
            auto bf16 = [](float f) -> uint32_t { uint32_t u; std::memcpy(&u, &f, 4); return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; };
            auto back = [](uint32_t b) -> float { const uint32_t u = b << 16; float f; std::memcpy(&f, &u, 4); return f; };
            std::vector<uint16_t> xs((size_t)3 * h->Npad * KBF, 0);
            for (int n = 0; n < Nall; ++n)
                for (int k = 0; k < I && k < KBF; ++k) {
                    const float x = packed[(size_t)n * IPY + k];
                    const uint32_t hi = bf16(x);
                    const float r1 = x - back(hi);
                    const uint32_t mid = bf16(r1);
                    const float r2 = r1 - back(mid);
                    const uint32_t lo = bf16(r2);
                    xs[((size_t)0 * h->Npad + n) * KBF + k] = (uint16_t)hi;
                    xs[((size_t)1 * h->Npad + n) * KBF + k] = (uint16_t)mid;
                    xs[((size_t)2 * h->Npad + n) * KBF + k] = (uint16_t)lo;
                }
            HIP_TRY(hipMalloc(&h->d_xs, xs.size() * sizeof(uint16_t)));
            HIP_TRY(hipMemcpy(h->d_xs, xs.data(), xs.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            h->fw_mfma = 2;
        }
    }
    if (H > WAVE) {
        // wide net: one thread per hidden unit, vectors in HBM, only the packed forward image + scratch in LDS
        // matrix-core layout (H a multiple of 32): the state vector joins the proposal in LDS when both fit (ceilings just below
        // 160 KiB are refused by the runtime); $PTNN_WIDE_RES=0 keeps the streaming kernel (A/B measurements)
        const size_t lds_res = wide_lds_floats(H, h->FWS, h->cfg.n_out, h->PS, true) * sizeof(float);
        const char* res_env = std::getenv("PTNN_WIDE_RES");
        h->wide_res = (H % 32 == 0) && lds_res <= 152 * 1024 && !(res_env && res_env[0] == '0');
        const size_t lds = h->wide_res ? lds_res : wide_lds_floats(H, h->FWS, h->cfg.n_out, h->PS) * sizeof(float);
        if (lds > 160 * 1024) return fail(-3, "wide net needs %zu B of LDS (> 160 KiB)", lds);
        {
            const char* ce = std::getenv("PTNN_COMPACT_TRACES");
            h->compact = h->cap == h->cfg.n_samples && !(ce && ce[0] == '0');
        }
        if (h->cfg.schedule == PTNN_SCHED_SPECULATIVE || h->cfg.schedule == PTNN_SCHED_PACKED || h->cfg.schedule == PTNN_SCHED_TREE)
            return fail(-3, "schedules 2-4 are built for n_hidden <= 64; a wide net speculates over work-groups through groups_per_replica");
        h->wide = true; h->speculative = false; h->groups = 1;
        h->nthreads = ((H + WAVE - 1) / WAVE) * WAVE;
        h->model_threads = h->nthreads;
        h->seg_lds = h->model_lds = lds;
        h->Ntr = ntr; h->Nte = nte;
        if (h->d_data) { HIP_TRY(hipFree(h->d_data)); h->d_data = nullptr; }
        HIP_TRY(hipMalloc(&h->d_data, packed.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->d_data, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(segment_function(h)), lds)) return rc;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(h->shape->model_wide), lds)) return rc;
        {
            // Speculation over work-groups (one per CU): group g computes step i + g; all Rl x G groups must be resident (they wait
            // for each other's verdicts).  groups_per_replica 1, 2 or 4; 0 = as many of 4, 2 as are resident, else 1.
            const int Rl = h->cfg.n_replicas_local;
            const int want = h->cfg.groups_per_replica;
            if (want != 0 && want != 1 && want != 2 && want != 4) return fail(-1, "wide nets: groups_per_replica must be 0 (auto), 1, 2 or 4");
            int per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(segment_function(h)), h->nthreads, lds));
            h->blocks_per_cu = per_cu;
            const long long cap = (long long)per_cu * h->num_cus;
            int G = 1;
            if (want > 1) {
                if ((long long)Rl * want > cap)
                    return fail(-3, "%d replicas x %d work-groups cannot all be resident: %d work-group(s) of %d threads with %zu B of LDS fit "
                                    "on each of the %d CUs", Rl, want, per_cu, h->nthreads, lds, h->num_cus);
                G = want;
            } else if (want == 0 && !h->cfg.shared_device) {     // (work-groups that wait for each other want the GPU to themselves)
                if ((long long)Rl * 4 <= cap) G = 4;
                else if ((long long)Rl * 2 <= cap) G = 2;
            }
            h->groups = G;
            if (h->d_wide_scratch) { HIP_TRY(hipFree(h->d_wide_scratch)); h->d_wide_scratch = nullptr; }
            HIP_TRY(hipMalloc(&h->d_wide_scratch, (size_t)Rl * G * 5 * h->PS * sizeof(float)));
            if (h->d_xslots) { HIP_TRY(hipFree(h->d_xslots)); h->d_xslots = nullptr; }
            if (h->d_xw) { HIP_TRY(hipFree(h->d_xw)); h->d_xw = nullptr; }
            if (h->d_xverdict) { HIP_TRY(hipFree(h->d_xverdict)); h->d_xverdict = nullptr; }
            if (G > 1) {
                const size_t ns = (size_t)Rl * 2 * MAX_SLOTS * SL_COUNT, nx = (size_t)Rl * 2 * G * 2 * h->PS, nvd = (size_t)Rl * 2 * MAX_SLOTS;
                HIP_TRY(hipMalloc(&h->d_xslots, ns * sizeof(unsigned long long)));
                HIP_TRY(hipMalloc(&h->d_xw, nx * sizeof(unsigned long long)));
                HIP_TRY(hipMalloc(&h->d_xverdict, nvd * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xslots, 0, ns * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xw, 0, nx * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xverdict, 0, nvd * sizeof(unsigned long long)));
                h->epoch_base = 1;                              // tag 0 = never written
            }
        }
        if (int rc = resolve_persistent(h)) return rc;
        h->have_data = true;
        return 0;
    }
    // LDS budget: the data set, the state vectors and the packed forward weights live in LDS for the whole launch
    const size_t coop_lds = lds_floats(Nall, IPY, h->PS, H, h->FWS) * sizeof(float);
    const size_t LDS_MAX = 160 * 1024;
    if (coop_lds > LDS_MAX)
        return fail(-3, "replica working set needs %zu B of LDS (> 160 KiB): data %d rows x %d floats, P = %d", coop_lds, Nall,
                    IPY, h->P);
    h->model_lds = coop_lds;
    // schedule: speculative pays when MH acceptance is low (regression chains: 1-15 %) or the step is dominated by the
    // sequential SGD sweep; cooperative when one step's row-parallel forward pass is the bulk of the work
    int sched = h->cfg.schedule;
    if (sched == PTNN_SCHED_AUTO)
        sched = (h->cfg.task == PTNN_TASK_REG || h->cfg.use_langevin) ? PTNN_SCHED_SPECULATIVE : PTNN_SCHED_COOPERATIVE;
    if (sched != PTNN_SCHED_COOPERATIVE && sched != PTNN_SCHED_SPECULATIVE && sched != PTNN_SCHED_PACKED && sched != PTNN_SCHED_TREE)
        return fail(-1, "unknown schedule %d", sched);
    if (sched == PTNN_SCHED_TREE && (h->cfg.task != PTNN_TASK_CLS || h->cfg.use_langevin))
        return fail(-3, "the prefetching tree schedule is built for random-walk classification runs");
    h->packed = false; h->tree = false;
    {
        // packed speculative: all slots of a round on one CU, the SGD epochs of the slots in the lane groups of two waves:
        // 16 slots in groups of 8 lanes for n_hidden <= 8, 8 slots in groups of 16 lanes for n_hidden <= 16.  Taken
        // automatically for Langevin runs of such nets (faster than 4 CUs per replica on a quarter of the GPU, and more than
        // twice the throughput once there are more replicas than CUs); random-walk-only runs have no epochs to pack and keep
        // the multi-CU speculative schedule.
        h->pk_nred = (H <= 8) ? 3 : 4;
        const size_t pk = pack_lds_floats(Nall, IPY, h->PS, H, h->FWS, pack_slots(h->pk_nred)) * sizeof(float);
        const bool fits = H <= 16 && pk <= LDS_MAX;
        if (sched == PTNN_SCHED_PACKED && !fits)
            return fail(-3, "the packed schedule needs n_hidden <= 16 and %zu B of LDS <= 160 KiB", pk);
        // 16-lane groups give 8 slots per round on one CU.  With CUs to spare the packed round runs on 2 or 4 CUs per replica (16 / 32
        // slots per round, segment_packm_kernel): Mackey-Glass 4-10-1, 64 replicas needs 18.7 / 13.6 / 11.6 rounds per swap interval
        // with 8 / 16 / 32 slots (profiles/r03_window_sim.jsonl) and a packed round is shorter than the multi-CU speculative one (the
        // forward passes run beside the epochs).  groups_per_replica = 1, 2, 4 decides otherwise; $PTNN_PACK_MULTI=0 keeps one CU.
        int pack_groups = 1;
        const size_t pkm = pack_multi_lds_floats(Nall, IPY, h->PS, H, h->FWS, pack_slots(h->pk_nred)) * sizeof(float);
        const char* pm_env = std::getenv("PTNN_PACK_MULTI");
        if (fits && pkm <= LDS_MAX && h->cfg.use_langevin && !h->cfg.shared_device && !(pm_env && pm_env[0] == '0') &&
            (sched == PTNN_SCHED_PACKED || h->cfg.schedule == PTNN_SCHED_AUTO) && h->cfg.waves_per_replica == 0) {
            const int Rl_ = h->cfg.n_replicas_local, want = h->cfg.groups_per_replica;
            if (want == 2 || want == 4) pack_groups = want;                       // (8-lane groups: on request only -- 16 slots on one CU already)
            else if (want == 0 && h->pk_nred == 4) { if (Rl_ * 4 <= h->num_cus) pack_groups = 4; else if (Rl_ * 2 <= h->num_cus) pack_groups = 2; }
        }
        const bool pays = (H <= 8) || pack_groups > 1 || h->cfg.n_replicas_local * 4 > h->num_cus;
        if (sched == PTNN_SCHED_PACKED ||
            (h->cfg.schedule == PTNN_SCHED_AUTO && sched == PTNN_SCHED_SPECULATIVE && fits && pays && h->cfg.use_langevin &&
             h->cfg.waves_per_replica == 0 && (h->cfg.groups_per_replica == 0 || pack_groups > 1))) {
            h->packed = true; h->speculative = true; h->groups = pack_groups;
            // eight waves (forward passes two to a SIMD) while every replica has a CU to itself, four beyond that; an explicit
            // waves_per_replica of 4 or 8 decides otherwise
            const int pkw = (h->cfg.waves_per_replica == 4 || h->cfg.waves_per_replica == 8) ? h->cfg.waves_per_replica
                            : (h->cfg.n_replicas_local <= h->num_cus ? PK_WAVES : 4);
            h->nthreads = (pack_groups > 1) ? PK_WAVES * WAVE : pkw * WAVE;
            h->seg_lds = (pack_groups > 1) ? pkm : pk;
            sched = PTNN_SCHED_PACKED;
            if (pack_groups > 1) {                              // exchange buffers of the multi-CU variant (segment_spec_body's layout)
                const int Rl_ = h->cfg.n_replicas_local;
                if (h->d_xslots) { HIP_TRY(hipFree(h->d_xslots)); h->d_xslots = nullptr; }
                if (h->d_xw) { HIP_TRY(hipFree(h->d_xw)); h->d_xw = nullptr; }
                if (h->d_xverdict) { HIP_TRY(hipFree(h->d_xverdict)); h->d_xverdict = nullptr; }
                const size_t ns = (size_t)Rl_ * 2 * MAX_SLOTS * SL_COUNT, nx = (size_t)Rl_ * 2 * pack_groups * 2 * h->PS, nvd = (size_t)Rl_ * 2 * MAX_SLOTS;
                HIP_TRY(hipMalloc(&h->d_xslots, ns * sizeof(unsigned long long)));
                HIP_TRY(hipMalloc(&h->d_xw, nx * sizeof(unsigned long long)));
                HIP_TRY(hipMalloc(&h->d_xverdict, nvd * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xslots, 0, ns * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xw, 0, nx * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xverdict, 0, nvd * sizeof(unsigned long long)));
                if (int rc = alloc_swap_granules(h, 2 * h->PS + 8)) return rc;     // in-launch swap rounds: state + cached gradient + flag
                h->epoch_base = 1;                              // tag 0 = never written
            }
        }
    }
    int nw = h->cfg.waves_per_replica;
    if (nw != 0 && nw != 1 && nw != 2 && nw != 4 && nw != 8)
        return fail(-1, "waves_per_replica must be 0 (auto), 1, 2, 4 or 8");
    int coop_nw = (Nall + 63) / 64, pow2 = 1;
    while (pow2 < coop_nw) pow2 <<= 1;
    coop_nw = std::min(pow2, 8);
    h->model_threads = coop_nw * 64;
    if (sched == PTNN_SCHED_SPECULATIVE) {
        h->packed = false;
        // Two waves on one SIMD slow each other ~1.65x (the SGD sweep is VALU-issue bound), so speculation depth comes
        // from more CUs first: G work-groups of 4 waves (one per SIMD) per replica while R*G <= number of CUs, and
        // 8 waves on a single CU otherwise.
        const int Rl = h->cfg.n_replicas_local;
        int G = 1;
        if (h->cfg.groups_per_replica > 0) G = h->cfg.groups_per_replica;
        else if (!h->cfg.shared_device) { while (G < 4 && Rl * (G * 2) <= h->num_cus) G *= 2; }
        if (G != 1 && G != 2 && G != 4 && G != 8) return fail(-1, "groups_per_replica must be 0 (auto), 1, 2, 4 or 8");
        int k = nw ? nw : (G > 1 ? 4 : 8);
        while (k > 1 && spec_lds_floats(Nall, IPY, h->PS, H, h->FWS, k, G) * sizeof(float) > LDS_MAX) k >>= 1;
        if (nw && k != nw) {
            if (h->cfg.schedule == PTNN_SCHED_AUTO) k = 0;      // auto: fall back to the cooperative schedule below
            else return fail(-3, "speculative schedule with %d waves needs more than 160 KiB of LDS", nw);
        }
        if (k * G > MAX_SLOTS) return fail(-1, "waves x groups must not exceed %d", MAX_SLOTS);
        if (k == 0 || spec_lds_floats(Nall, IPY, h->PS, H, h->FWS, k, G) * sizeof(float) > LDS_MAX) sched = PTNN_SCHED_COOPERATIVE;
        else {
            h->speculative = true; h->nthreads = k * 64; h->groups = G;
            h->seg_lds = spec_lds_floats(Nall, IPY, h->PS, H, h->FWS, k, G) * sizeof(float);
            if (G > 1) {
                // The work-groups of one replica wait for each other inside the kernel, so all Rl x G of them must be
                // resident at once: ask the runtime how many blocks of THIS kernel (its registers, this LDS size, this block
                // size) fit on a CU instead of guessing, and refuse the configuration otherwise (a non-resident partner would
                // be a bounded spin and an error from ptnn_sync).  The count is for a GPU this handle has to itself: other
                // handles or processes on the same device take CUs this query does not see.
                const void* fn = reinterpret_cast<const void*>(h->shape->spec);
                if (int rc = raise_lds_limit(fn, h->seg_lds)) return rc;
                int per_cu = 0;
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, h->nthreads, h->seg_lds));
                h->blocks_per_cu = per_cu;
                if ((long long)Rl * G > (long long)per_cu * h->num_cus)
                    return fail(-3, "%d replicas x %d work-groups cannot all be resident: %d work-group(s) of %d threads with %zu B "
                                    "of LDS fit on each of the %d CUs; use fewer groups_per_replica or the packed schedule",
                                Rl, G, per_cu, h->nthreads, h->seg_lds, h->num_cus);
            }
            if (h->d_xslots) { HIP_TRY(hipFree(h->d_xslots)); h->d_xslots = nullptr; }
            if (h->d_xw) { HIP_TRY(hipFree(h->d_xw)); h->d_xw = nullptr; }
            if (h->d_xverdict) { HIP_TRY(hipFree(h->d_xverdict)); h->d_xverdict = nullptr; }
            if (G > 1) {
                const size_t ns = (size_t)Rl * 2 * MAX_SLOTS * SL_COUNT, nx = (size_t)Rl * 2 * MAX_SLOTS * 2 * h->PS;
                HIP_TRY(hipMalloc(&h->d_xslots, ns * sizeof(unsigned long long)));
                HIP_TRY(hipMalloc(&h->d_xw, nx * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xslots, 0, ns * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xw, 0, nx * sizeof(unsigned long long)));
                HIP_TRY(hipMalloc(&h->d_xverdict, (size_t)Rl * 2 * MAX_SLOTS * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(h->d_xverdict, 0, (size_t)Rl * 2 * MAX_SLOTS * sizeof(unsigned long long)));
                h->epoch_base = 1;                              // tag 0 = never written
            }
        }
    }
    if (sched == PTNN_SCHED_COOPERATIVE) {
        h->speculative = false;
        h->nthreads = (nw ? nw : coop_nw) * 64;
        const size_t seg_lds = lds_floats(Nall, IPY, h->PS, H, h->FWS, h->cfg.use_langevin != 0) * sizeof(float);
        h->seg_lds = seg_lds;
        // a hidden layer that fills most of a 32-unit tile and at least three k-steps: forward pass on the matrix cores
        // (the VALU pass re-reads the weights from LDS with broadcast reads and is bound by the LDS pipe at this size)
        const size_t extra = (mfma_coop_lds_floats(I, h->cfg.n_out, H, h->Npad) + 4) * sizeof(float);
        h->fw_mfma = (H >= 24 && I >= 6 && seg_lds + extra <= LDS_MAX) ? 1 : 0;
        if (h->fw_mfma) h->seg_lds = seg_lds + extra;
        // Split bf16 operands (ptnn_device.hpp, SplitK): the default where the matrix cores are used, unless the caller asked for
        // the exact fp32 instruction (forward_bf16 = 2: bit-identical to the VALU schedules) or the images do not fit.  They take
        // more LDS than the transposed fp32 image; a random-walk launch may give up the row-major data image for them (its only
        // readers left are the chain start and the SGD epochs of a Langevin launch, which therefore keeps it).
        const char* fs = std::getenv("PTNN_FW_SPLIT");
        if (h->fw_mfma && h->shape->split_ch > 0 && h->cfg.forward_bf16 != 2 && !(fs && fs[0] == '0')) {
            const int CH = h->shape->split_ch, KR = h->shape->split_kr, Hpad = ((H + 31) >> 5) << 5;
            const size_t sfl = (size_t)3 * h->Npad * CH * 4 + (size_t)2 * KR * h->Npad + (size_t)h->Npad + (size_t)3 * Hpad * CH * 4 +
                               (size_t)(Hpad >> 5) * h->Npad * h->cfg.n_out + 4;
            const size_t with_xy = seg_lds + sfl * sizeof(float);
            const size_t without_xy = lds_floats(Nall, IPY, h->PS, H, h->FWS, h->cfg.use_langevin != 0, false) * sizeof(float) + sfl * sizeof(float);
            if (with_xy <= LDS_MAX) { h->fw_mfma = 2; h->xy_global = false; h->seg_lds = with_xy; }
            else if (!h->cfg.use_langevin && without_xy <= LDS_MAX) { h->fw_mfma = 2; h->xy_global = true; h->seg_lds = without_xy; }
        }
    }
    if (sched == PTNN_SCHED_TREE || (sched == PTNN_SCHED_COOPERATIVE && h->cfg.schedule == PTNN_SCHED_AUTO && h->cfg.task == PTNN_TASK_CLS &&
                                     !h->cfg.use_langevin && h->cfg.groups_per_replica == 0 && h->cfg.waves_per_replica == 0 &&
                                     !h->cfg.shared_device)) {
        // Prefetching tree: 2^D - 1 work-groups per replica, all of them resident (they wait for each other's records).
        // Explicit: groups_per_replica = 3, 7, 15 or 31 (0: deepest that fits); auto: deepest of 31 / 15 / 7 / 3 that fits (31 nodes:
        // five steps per round; Iris 16 x 31 = 496 work-groups, two to a CU: 11.8 M against 11.4 M samples/s with 15),
        // none -> the cooperative schedule stays.
        const bool explicit_tree = sched == PTNN_SCHED_TREE;
        const int Rl = h->cfg.n_replicas_local;
        const int want = h->cfg.groups_per_replica;
        if (explicit_tree && want != 0 && want != 3 && want != 7 && want != 15 && want != 31)
            return fail(-1, "tree schedule: groups_per_replica must be 0 (auto), 3, 7, 15 or 31");
        h->nthreads = (nw ? nw : coop_nw) * 64;
        size_t extra = (mfma_coop_lds_floats(I, h->cfg.n_out, H, h->Npad) + 4) * sizeof(float);
        const bool coop_mfma = H >= 24 && I >= 6 && lds_floats(Nall, IPY, h->PS, H, h->FWS, false) * sizeof(float) + extra <= LDS_MAX;
        // the split-operand forward pass of the cooperative kernel (same arithmetic: the tree commits the cooperative chain bit for
        // bit either way), when its images fit next to the shallowest tree
        bool tree_split = false;
        {
            const char* fs = std::getenv("PTNN_FW_SPLIT");
            if (coop_mfma && h->shape->split_ch > 0 && h->cfg.forward_bf16 != 2 && !(fs && fs[0] == '0')) {
                const int CH = h->shape->split_ch, KR = h->shape->split_kr, Hpad = ((H + 31) >> 5) << 5;
                const size_t sfl = (size_t)3 * h->Npad * CH * 4 + (size_t)2 * KR * h->Npad + (size_t)h->Npad + (size_t)3 * Hpad * CH * 4 +
                                   (size_t)(Hpad >> 5) * h->Npad * h->cfg.n_out + 4;
                if (tree_lds_floats(Nall, IPY, h->PS, H, h->FWS, 2, false, true) * sizeof(float) + sfl * sizeof(float) <= 152 * 1024) {
                    tree_split = true;
                    extra = sfl * sizeof(float);
                }
            }
        }
        int chosen = 0;
        size_t chosen_lds = 0;
        bool chosen_mfma = false;
        const void* fn = reinterpret_cast<const void*>(h->shape->tree);
        for (int G = (explicit_tree && want) ? want : TREE_MAX_NODES; G >= 3; G = (G - 1) / 2) {
            const int Dp = tree_depth(G);
            // two sets of tapes (the next round's drawn while the records travel) when they fit
            const size_t ceiling = 152 * 1024;                  // dynamic-LDS ceilings just below 160 KiB are refused by the runtime
            bool ah = tree_lds_floats(Nall, IPY, h->PS, H, h->FWS, Dp, true, coop_mfma) * sizeof(float) + (coop_mfma ? extra : 0) <= ceiling;
            size_t lds = tree_lds_floats(Nall, IPY, h->PS, H, h->FWS, Dp, ah, coop_mfma) * sizeof(float);
            // same forward pass as the cooperative schedule would run (matrix cores or not): a deeper tree that has no room
            // for the transposed data image is not taken
            const bool mf = coop_mfma;
            if (mf) lds += extra;
            // (the runtime may refuse a dynamic-LDS ceiling just below 160 KiB: such a depth does not fit either)
            const bool fits = lds <= LDS_MAX && raise_lds_limit(fn, lds) == 0;
            if (!fits) (void)hipGetLastError();                 // a refused ceiling must not surface at the next launch
            if (fits) {
                int per_cu = 0;
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, h->nthreads, lds));
                if ((long long)Rl * G <= (long long)per_cu * h->num_cus) { chosen = G; chosen_lds = lds; chosen_mfma = mf; h->tree_ahead = ah; h->blocks_per_cu = per_cu; break; }
            }
            if (explicit_tree && want) break;
        }
        if (chosen) {
            h->tree = true; h->speculative = false; h->groups = chosen; h->seg_lds = chosen_lds; h->fw_mfma = chosen_mfma ? (tree_split ? 2 : 1) : 0; h->xy_global = false;
            if (h->d_xslots) { HIP_TRY(hipFree(h->d_xslots)); h->d_xslots = nullptr; }
            const size_t ng = (size_t)Rl * 2 * (TREE_MAX_NODES + 1) * TREE_REC;
            HIP_TRY(hipMalloc(&h->d_xslots, ng * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(h->d_xslots, 0, ng * sizeof(unsigned long long)));
            // granules of the in-launch swap rounds (two parities), when the whole ladder is on this handle
            if (int rc = alloc_swap_granules(h, h->PS)) return rc;
            h->epoch_base = 1;                                  // tag 0 = never written
        } else if (explicit_tree) {
            return fail(-3, "tree schedule: %d replicas x %d work-groups of %d threads cannot all be resident on %d CUs (or need more than "
                            "160 KiB of LDS)", Rl, want ? want : 3, h->nthreads, h->num_cus);
        }
    }
    h->Ntr = ntr; h->Nte = nte;
    if (h->d_data) { HIP_TRY(hipFree(h->d_data)); h->d_data = nullptr; }
    HIP_TRY(hipMalloc(&h->d_data, packed.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(h->d_data, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(segment_function(h)), h->seg_lds)) return rc;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(h->shape->model), h->model_lds)) return rc;
    if (int rc = resolve_persistent(h)) return rc;
    h->have_data = true;
    return 0;
}

int ptnn_set_state(ptnn_handle* h, const float* w0, const float* temperatures) {
    if (!h || !w0 || !temperatures) return fail(-1, "null argument");
    if (!h->have_data) return fail(-1, "call ptnn_set_data before ptnn_set_state");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = wait_stream(h)) return rc;                // a restart must not overtake a run still in flight (nor the staging buffer)
    if (h->copy_stream) {                                   // ... nor the trace rows of the previous run on their way to the host
        HIP_TRY(hipStreamSynchronize(h->copy_stream));
        for (auto& ev : h->img_events) (void)hipEventDestroy(ev);
        h->img_events.clear();
    }
    const int Rl = h->cfg.n_replicas_local, P = h->P;
    // one pinned staging buffer, one asynchronous copy, one kernel -- all on the handle's stream (a whole-run restart is part of
    // the benchmark's timed region).  Trace rows 1 .. S-1 need no clearing: every one of them is written by the MH step it
    // belongs to before ptnn_get_traces lets anybody read it.
    std::memcpy(h->h_stage, w0, (size_t)Rl * P * sizeof(float));
    std::memcpy(h->h_stage + (size_t)Rl * P, temperatures, (size_t)Rl * sizeof(float));
    HIP_TRY(hipMemcpyAsync(h->d_stage, h->h_stage, ((size_t)Rl * P + Rl) * sizeof(float), hipMemcpyHostToDevice, h->stream));
    ResetParams q{};
    q.R = h->cfg.n_replicas_global; q.Rl = Rl; q.P = P; q.PS = h->PS; q.PW = h->PW; q.cap = (size_t)h->cap;
    q.w0 = h->d_stage; q.temps_in = h->d_stage + (size_t)Rl * P;
    q.state0 = h->d_state[0]; q.state1 = h->d_state[1]; q.rec_w = h->d_rec_w; q.gd0 = h->d_gd_w[0]; q.gd1 = h->d_gd_w[1];
    q.st_f = h->d_st_f; q.temps = h->d_temps; q.pos_w = h->d_pos_w; q.scal = h->d_scal;
    q.gd_valid0 = h->d_gd_valid[0]; q.gd_valid1 = h->d_gd_valid[1]; q.st_i = h->d_st_i; q.error = h->d_error;
    q.label0 = h->d_label[0]; q.label1 = h->d_label[1]; q.slot0 = h->d_slot_of[0]; q.slot1 = h->d_slot_of[1];
    q.counters = h->d_counters;
    hipLaunchKernelGGL(chain_reset_kernel, dim3(Rl), dim3(256), 0, h->stream, q);
    HIP_TRY(hipGetLastError());
    h->flip = 0; h->cur = 0; h->rounds_done = 0; h->finalized = false; h->drained = 0; h->first_row = 0; h->lflip = 0;
    if (!h->comm.failed) { h->failed = false; h->failure.clear(); }   // a restart clears a failed run (a failed communicator stays failed)
    h->h_progress[0] = h->h_progress[1] = 0;
    h->have_state = true;
    return 0;
}

int ptnn_set_ladder(ptnn_handle* h, const float* temperatures_global) {
    if (!h || !temperatures_global) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipMemcpy(h->d_temps_global, temperatures_global, h->cfg.n_replicas_global * sizeof(float), hipMemcpyHostToDevice));
    h->have_ladder = true;
    return 0;
}

int ptnn_steps_done(ptnn_handle* h) { return h ? h->cur : -1; }

// what a swap round moves between GPUs for this handle: AUTO = gather while the gathered buffer stays small
static int resolved_xchg_mode(const ptnn_handle* h) {
    if (h->cfg.swap_rule == 1) return PTNN_XCHG_GATHER;       // the moved state brings likelihood and prior along: rows only
    if (h->comm.mode != PTNN_XCHG_AUTO) return h->comm.mode;
    const size_t gathered = (size_t)h->cfg.n_replicas_global * xchg_row_floats(h->PS) * sizeof(float);
    return gathered <= (size_t)4 << 20 ? PTNN_XCHG_GATHER : PTNN_XCHG_BOUNDARY;
}

// bookkeeping after the kernels of a round are queued: which buffers are current now
static void round_queued(ptnn_handle* h, bool phantom) {
    if (!phantom) {
        if (h->cfg.label_swap) h->lflip ^= 1;                // the maps changed, the chains stayed where they are
        else h->flip ^= 1;
    }
    h->rounds_done += 1;
}

// One swap round of a sharded ladder (the handle owns a block of it), everything queued on the handle's stream.
static int comm_swap_round(ptnn_handle* h, bool phantom) {
    Comm& c = h->comm;
    const int Rl = h->cfg.n_replicas_local, R = h->cfg.n_replicas_global;
    if (h->cfg.label_swap) {
        // zero payload: only the posted scalars travel (4 R bytes; the untempered ones too under swap_rule 1)
        if (!c.all_gather(phantom ? h->d_L_final : h->d_L_handoff, (size_t)Rl * sizeof(float), h->stream)) return fail(-7, "%s", c.err.c_str());
        if (h->cfg.swap_rule == 1 && !c.all_gather(h->d_L_raw, (size_t)Rl * sizeof(float), h->stream)) return fail(-7, "%s", c.err.c_str());
        if (int rc = launch_swap(h, phantom, phantom ? 2 : 3, false)) return rc;
        round_queued(h, phantom);
        c.rounds += 1;
        return 0;
    }
    if (resolved_xchg_mode(h) == PTNN_XCHG_GATHER) {
        if (int rc = launch_swap(h, phantom, -1, false)) return rc;                     // exchange rows of the local replicas
        if (!c.all_gather(h->d_xchg, (size_t)Rl * xchg_row_floats(h->PS) * sizeof(float), h->stream)) return fail(-7, "%s", c.err.c_str());
        if (int rc = launch_swap(h, phantom, phantom ? (2 | 4) : (3 | 4), false)) return rc;   // identical cascade + source rows
    } else {
        if (!c.all_gather(phantom ? h->d_L_final : h->d_L_handoff, (size_t)Rl * sizeof(float), h->stream)) return fail(-7, "%s", c.err.c_str());
        if (phantom) {                                                                  // counted, result discarded (Q13)
            if (int rc = launch_swap(h, true, 2, false)) return rc;
        } else {
            if (int rc = launch_swap(h, false, 0, true)) return rc;                     // cascade only: src[R]
            HIP_TRY(hipMemcpyAsync(h->h_src, h->d_src, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            if (int rc = wait_stream(h)) return rc;                                   // the one host wait of this mode
            route_rows(h->h_src, R, Rl, c.rank, h->route);
            float* cur = h->d_state[h->flip];
            float* next = h->d_state[h->flip ^ 1];
            const size_t PS = h->PS;
            if (!c.exchange_rows(h->route, PS * sizeof(float), h->stream,
                                 [&](int row) { return static_cast<void*>(cur + (size_t)row * PS); },
                                 [&](int row) { return static_cast<void*>(next + (size_t)row * PS); }))
                return fail(-7, "%s", c.err.c_str());
            if (int rc = launch_swap(h, false, 3, false)) return rc;                    // local moves (arrived rows stay), count, log
        }
    }
    round_queued(h, phantom);
    c.rounds += 1;
    return 0;
}

int ptnn_run(ptnn_handle* h, int n_steps) {
    if (int rc = check_ready(h)) return rc;
    // with a communicator attached the swap rounds go through it, also when it has a single rank (rehearsal of the path)
    const bool sharded = h->comm.kind != COMM_NONE;
    if (h->cfg.n_replicas_local != h->cfg.n_replicas_global && !sharded)
        return fail(-1, "this handle owns replicas %d..%d of %d: attach a communicator first (ptnn_comm_init / ptnn_comm_init_host), "
                        "or drive the pieces yourself with ptnn_run_segment + ptnn_swap_*", h->cfg.first_global_replica,
                    h->cfg.first_global_replica + h->cfg.n_replicas_local - 1, h->cfg.n_replicas_global);
    if ((h->cfg.swap_rule == 1 || h->cfg.label_swap) && !h->have_ladder)
        return fail(-1, "swap_rule 1 and label_swap need ptnn_set_ladder (all temperatures)");
    const int S = h->cfg.n_samples;
    const int last = S - 1;                                  // steps are i = 0 .. S-2
    int end = (n_steps < 0) ? last : std::min(last, h->cur + n_steps);
    if (h->cap < S && end - h->drained > h->cap - 1)
        return fail(-6, "trace ring of %d rows would overflow: rows from %d on have not been fetched; call ptnn_get_traces "
                        "first or run fewer steps", h->cap, h->drained + 1);
    if (h->persistent && !sharded && h->cur < end) {
        // every work-group of the grid is resident: ONE launch runs all the intervals up to `end`, the swap rounds between them
        // inside the kernel (persistent_loop in ptnn_device.hpp); what is left to do here is the bookkeeping of those rounds
        int n_ho = 0;
        for (int c = h->cur; c < end;) {
            int seg_end = c;
            while (seg_end < end && !swap_trigger(h->cfg, seg_end)) ++seg_end;
            if (seg_end >= end) break;
            ++n_ho;
            c = seg_end + 1;
        }
        if (int rc = launch_segment(h, h->cur, end, true)) return rc;
        h->cur = end;
        for (int k = 0; k < n_ho; ++k) round_queued(h, false);
    }
    while (h->cur < end) {
        int seg_end = h->cur;
        while (seg_end < end && !swap_trigger(h->cfg, seg_end)) ++seg_end;
        const bool handoff = seg_end < end;                  // step seg_end triggers a hand-off
        const int stop = handoff ? seg_end + 1 : end;
        const bool phantom_next = !handoff && stop == last && !h->finalized && h->cfg.swap_rule == 0 && S / h->cfg.swap_interval > h->rounds_done;
        if (int rc = launch_segment(h, h->cur, stop, false, sharded && (handoff || phantom_next))) return rc;
        h->cur = stop;
        if (handoff) {
            if (sharded) {
                if (int rc = comm_swap_round(h, false)) return rc;
            } else {
                if (int rc = launch_swap(h, false, 3, false)) return rc;
                round_queued(h, false);
            }
        }
    }
    if (h->cur == last && !h->finalized) {
        // Q13: the parent loops int(S/si) rounds; a round beyond the replicas' hand-offs consumes the end-of-chain
        // vectors, is counted in swap_perc and its result is discarded
        if (h->cfg.swap_rule == 0 && S / h->cfg.swap_interval > h->rounds_done) {
            if (sharded) {
                if (int rc = comm_swap_round(h, true)) return rc;
            } else {
                if (int rc = launch_swap(h, true, 2, false)) return rc;
                h->rounds_done += 1;
            }
        }
        h->finalized = true;
    }
    return 0;
}

// ---- communicators of the sharded ladder ----
static int comm_check_partition(ptnn_handle* h, int rank, int nranks) {
    if (!h) return fail(-1, "null handle");
    if (h->comm.kind != COMM_NONE) return fail(-1, "this handle already has a communicator");
    const int Rl = h->cfg.n_replicas_local, R = h->cfg.n_replicas_global;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "bad rank %d of %d", rank, nranks);
    if (Rl * nranks != R || h->cfg.first_global_replica != rank * Rl)
        return fail(-1, "the ladder must be cut into equal contiguous blocks: rank %d of %d owns replicas %d..%d of %d, expected "
                        "first_global_replica == rank * n_replicas_local and n_replicas_local * nranks == n_replicas_global",
                    rank, nranks, h->cfg.first_global_replica, h->cfg.first_global_replica + Rl - 1, R);
    return 0;
}

int ptnn_comm_unique_id(void* id_out, int nbytes) {
    if (!id_out || nbytes < (int)sizeof(ncclUniqueId)) return fail(-1, "the unique id needs a buffer of %d bytes", (int)sizeof(ncclUniqueId));
    // fault injection for the fall-back tests ($PTNN_COMM_FAULT=ncclGetUniqueId): fails before RCCL is loaded
    if (const char* f = std::getenv("PTNN_COMM_FAULT"))
        if (std::strstr(f, "ncclGetUniqueId")) return fail(-7, "ncclGetUniqueId failed: injected by $PTNN_COMM_FAULT");
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return fail(-7, "cannot load RCCL: %s", why.c_str());
    // static storage: the helper thread may outlive this call when the stage is abandoned
    static ncclUniqueId id;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    comm_stage("ncclGetUniqueId");
    int r = 0;
    if (!run_bounded([api]() -> int { return (int)api->GetUniqueId(&id); }, comm_timeout_s(), &r))
        return fail(-7, "ncclGetUniqueId did not return within %d s (it opens the bootstrap listener: NCCL_SOCKET_IFNAME=%s)",
                    (int)comm_timeout_s(), std::getenv("NCCL_SOCKET_IFNAME") ? std::getenv("NCCL_SOCKET_IFNAME") : "unset");
    if (r != (int)ncclSuccess) return fail(-7, "ncclGetUniqueId failed: %s", api->GetErrorString((ncclResult_t)r));
    comm_stage("ncclGetUniqueId done");
    std::memcpy(id_out, &id, sizeof id);
    return (int)sizeof id;
}

int ptnn_comm_init(ptnn_handle* h, const void* unique_id, int nbytes, int rank, int nranks) {
    if (int rc = comm_check_partition(h, rank, nranks)) return rc;
    if (!unique_id || nbytes != (int)sizeof(ncclUniqueId)) return fail(-1, "the unique id must be the %d bytes ptnn_comm_unique_id wrote", (int)sizeof(ncclUniqueId));
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return fail(-7, "cannot load RCCL: %s", why.c_str());
    // fault injection for the fall-back tests ($PTNN_COMM_FAULT=ncclCommInitRank): fails where a refused bring-up would, RCCL not called
    if (const char* f = std::getenv("PTNN_COMM_FAULT"))
        if (std::strstr(f, "ncclCommInitRank"))
            return fail(-7, "ncclCommInitRank(rank %d of %d, device %d) failed: injected by $PTNN_COMM_FAULT", rank, nranks, h->cfg.device_id);
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    // ncclCommInitRank is a collective: it returns when all nranks have joined.  A peer that never does (it failed before, or was
    // never started) would block this thread for ever, so the call runs on a helper that is abandoned after comm_timeout_s().
    struct Job { ncclUniqueId id; ncclComm_t comm = nullptr; };
    auto job = std::make_shared<Job>();
    std::memcpy(&job->id, unique_id, sizeof job->id);
    const int dev = h->cfg.device_id;
    comm_stage("ncclCommInitRank(rank %d of %d, device %d)", rank, nranks, dev);
    int r = 0;
    const bool finished = run_bounded([api, job, dev, rank, nranks]() -> int {
        if (hipSetDevice(dev) != hipSuccess) return (int)ncclUnhandledCudaError;
        return (int)api->CommInitRank(&job->comm, nranks, job->id, rank);
    }, comm_timeout_s(), &r);
    if (!finished)
        return fail(-7, "ncclCommInitRank(rank %d of %d, device %d) did not return within %d s: a rank never joined, or the bring-up "
                        "stalled ($PTNN_COMM_TRACE=1 and NCCL_DEBUG=INFO show where)", rank, nranks, dev, (int)comm_timeout_s());
    if (r != (int)ncclSuccess) return fail(-7, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, dev, api->GetErrorString((ncclResult_t)r));
    comm_stage("ncclCommInitRank done (rank %d of %d)", rank, nranks);
    h->comm.api = api; h->comm.nccl = job->comm; h->comm.rank = rank; h->comm.nranks = nranks; h->comm.kind = COMM_RCCL;
    return 0;
}

// One bounded round trip of RCCL among `devices` from THIS process: unique id, one ncclCommInitRank per device (a thread each),
// a 4-byte all-gather, destroy.  Meant to be run in a fresh CHILD process before the long-lived one touches RCCL (Python:
// distributed.rccl_probe): a bring-up that stalls or fails is then the child's, which is killed -- an RCCL left half
// initialised in the caller's own process (a unique id that no ncclCommInitRank follows, helpers abandoned inside
// ncclCommInitRank) can keep that process from exiting.  Honours $PTNN_COMM_FAULT like the real bring-up.
int ptnn_comm_probe(const int32_t* devices, int n, double* seconds) {
    if (!devices || n < 1 || n > 64) return fail(-1, "bad argument");
    const double t0 = comm_clock();
    const char* fault = std::getenv("PTNN_COMM_FAULT");
    if (fault && std::strstr(fault, "ncclGetUniqueId")) return fail(-7, "ncclGetUniqueId failed: injected by $PTNN_COMM_FAULT");
    char id[sizeof(ncclUniqueId)];
    if (int rc = ptnn_comm_unique_id(id, (int)sizeof id); rc < 0) return rc;
    if (fault && std::strstr(fault, "ncclCommInitRank")) return fail(-7, "ncclCommInitRank failed: injected by $PTNN_COMM_FAULT");
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (devices[a] == devices[b]) return fail(-1, "RCCL needs one distinct device per rank (device %d appears twice)", devices[a]);
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return fail(-7, "cannot load RCCL: %s", why.c_str());
    struct Shared { ncclUniqueId id; std::vector<int> rc; std::vector<std::string> msg; };
    auto sh = std::make_shared<Shared>();
    std::memcpy(&sh->id, id, sizeof sh->id);
    sh->rc.assign((size_t)n, -1); sh->msg.resize((size_t)n);
    std::vector<int> devs(devices, devices + n);
    comm_stage("probe: ncclCommInitRank x %d + one all-gather", n);
    int r = 0;
    const bool finished = run_bounded([api, sh, devs, n]() -> int {
        std::vector<std::thread> th;
        for (int k = 0; k < n; ++k)
            th.emplace_back([api, sh, devs, n, k]() {
                auto bad = [&](const char* what, const char* detail) { sh->msg[(size_t)k] = std::string(what) + ": " + detail; sh->rc[(size_t)k] = 1; };
                if (hipSetDevice(devs[(size_t)k]) != hipSuccess) return bad("hipSetDevice", "failed");
                ncclComm_t comm = nullptr;
                ncclResult_t e = api->CommInitRank(&comm, n, sh->id, k);
                if (e != ncclSuccess) return bad("ncclCommInitRank", api->GetErrorString(e));
                hipStream_t st = nullptr;
                int32_t* buf = nullptr;
                bool ok = hipStreamCreate(&st) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&buf), sizeof(int32_t) * (size_t)n) == hipSuccess;
                const int32_t mine = 1000 + k;
                ok = ok && hipMemcpyAsync(buf + k, &mine, sizeof mine, hipMemcpyHostToDevice, st) == hipSuccess;
                if (ok) {
                    e = api->AllGather(buf + k, buf, sizeof(int32_t), ncclChar, comm, st);
                    std::vector<int32_t> got((size_t)n, 0);
                    ok = e == ncclSuccess && hipMemcpyAsync(got.data(), buf, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipStreamSynchronize(st) == hipSuccess;
                    for (int j = 0; ok && j < n; ++j) ok = got[(size_t)j] == 1000 + j;
                    if (!ok) bad("ncclAllGather", e == ncclSuccess ? "wrong or missing data" : api->GetErrorString(e));
                } else bad("hip", "stream / buffer set-up failed");
                if (buf) (void)hipFree(buf);
                if (st) (void)hipStreamDestroy(st);
                (void)api->CommDestroy(comm);
                if (ok) sh->rc[(size_t)k] = 0;
            });
        for (auto& t : th) t.join();
        for (int k = 0; k < n; ++k) if (sh->rc[(size_t)k] != 0) return 1 + k;
        return 0;
    }, comm_timeout_s(), &r);
    if (seconds) *seconds = comm_clock() - t0;
    if (!finished) return fail(-7, "the RCCL probe over %d devices did not finish within %d s (last stage: %s)", n, (int)comm_timeout_s(), comm_last_stage().c_str());
    if (r != 0) return fail(-7, "the RCCL probe failed on rank %d: %s", r - 1, sh->msg[(size_t)(r - 1)].c_str());
    comm_stage("probe done");
    return 0;
}

int ptnn_comm_info(ptnn_handle* h, int32_t* transport, int32_t* rank, int32_t* nranks, int32_t* device) {
    if (!h) return fail(-1, "null handle");
    if (transport) *transport = h->comm.kind;
    if (rank) *rank = h->comm.kind == COMM_NONE ? 0 : h->comm.rank;
    if (device) *device = h->cfg.device_id;
    if (nranks) {
        *nranks = h->comm.kind == COMM_NONE ? 1 : h->comm.nranks;
        if (h->comm.kind == COMM_RCCL && h->comm.api && h->comm.api->CommCount && h->comm.nccl) {
            int c = 0;                                         // what the communicator itself says, not what the caller passed in
            if (h->comm.api->CommCount(h->comm.nccl, &c) != ncclSuccess) return fail(-7, "ncclCommCount failed");
            *nranks = c;
        }
    }
    return 0;
}

int ptnn_comm_last_stage(char* buf, int nbytes) {
    if (!buf || nbytes < 1) return fail(-1, "bad argument");
    const std::string s = comm_last_stage();
    std::snprintf(buf, (size_t)nbytes, "%s", s.c_str());
    return (int)std::min<size_t>(s.size(), (size_t)nbytes - 1);
}

int ptnn_comm_init_host(ptnn_handle* h, int rank, int nranks, ptnn_all_gather_fn all_gather, ptnn_send_recv_fn send_recv, void* ctx) {
    if (int rc = comm_check_partition(h, rank, nranks)) return rc;
    if (!all_gather || !send_recv) return fail(-1, "null callback");
    h->comm.h_all_gather = all_gather; h->comm.h_send_recv = send_recv; h->comm.h_ctx = ctx;
    h->comm.rank = rank; h->comm.nranks = nranks; h->comm.kind = COMM_HOST;
    return 0;
}

int ptnn_comm_set_mode(ptnn_handle* h, int mode) {
    if (!h) return fail(-1, "null handle");
    if (mode != PTNN_XCHG_AUTO && mode != PTNN_XCHG_GATHER && mode != PTNN_XCHG_BOUNDARY) return fail(-1, "unknown exchange mode %d", mode);
    if (mode == PTNN_XCHG_BOUNDARY && h->cfg.swap_rule != 0)
        return fail(-3, "the boundary exchange implements the reference's cascade (swap_rule 0) only: use the gathered exchange");
    h->comm.mode = mode;
    return 0;
}

int ptnn_comm_stats(ptnn_handle* h, int64_t* bytes_sent, int64_t* bytes_received, int64_t* rounds, int32_t* mode) {
    if (!h) return fail(-1, "null handle");
    if (bytes_sent) *bytes_sent = h->comm.bytes_sent;
    if (bytes_received) *bytes_received = h->comm.bytes_received;
    if (rounds) *rounds = h->comm.rounds;
    if (mode) *mode = h->comm.kind == COMM_NONE ? 0 : resolved_xchg_mode(h);
    return 0;
}

int ptnn_comm_finalize(ptnn_handle* h) {
    if (!h) return fail(-1, "null handle");
    (void)hipSetDevice(h->cfg.device_id);
    int rc = 0;
    if (h->stream && !h->comm.failed) rc = wait_stream(h);
    h->comm.release();
    return rc;
}

int ptnn_route(const int32_t* src, int n_global, int n_local, int rank, int32_t* msg, int max_msgs) {
    if (!src || !msg || n_local < 1 || n_global < n_local || n_global % n_local != 0 || rank < 0 || rank >= n_global / n_local)
        return fail(-1, "bad argument");
    for (int k = 0; k < n_global; ++k)
        if (src[k] < 0 || src[k] >= n_global) return fail(-1, "src[%d] = %d is not a slot", k, src[k]);
    std::vector<RowMsg> r;
    route_rows(src, n_global, n_local, rank, r);
    if ((int)r.size() > max_msgs) return fail(-1, "%d messages, room for %d", (int)r.size(), max_msgs);
    for (size_t m = 0; m < r.size(); ++m) {
        msg[4 * m] = r[m].is_send; msg[4 * m + 1] = r[m].peer; msg[4 * m + 2] = r[m].local_row; msg[4 * m + 3] = r[m].global_dst;
    }
    return (int)r.size();
}

int ptnn_sync(ptnn_handle* h) {
    if (!h) return fail(-1, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    collect_timing(h);
    return 0;
}

int ptnn_run_segment(ptnn_handle* h, int* handoff) {
    if (int rc = check_ready(h)) return rc;
    if (!handoff) return fail(-1, "null argument");
    const int S = h->cfg.n_samples, last = S - 1;
    *handoff = 0;
    if (h->cur < last) {
        int seg_end = h->cur;
        while (seg_end < last && !swap_trigger(h->cfg, seg_end)) ++seg_end;
        const bool ho = seg_end < last;
        const int stop = ho ? seg_end + 1 : last;
        if (h->cap < S && stop - h->drained > h->cap - 1)
            return fail(-6, "trace ring of %d rows would overflow: fetch rows from %d on with ptnn_get_traces first", h->cap,
                        h->drained + 1);
        if (int rc = launch_segment(h, h->cur, stop)) return rc;
        h->cur = stop;
        if (ho) { *handoff = 1; return 0; }
    }
    if (h->cur == last && !h->finalized) {
        h->finalized = true;
        if (h->cfg.swap_rule == 0 && S / h->cfg.swap_interval > h->rounds_done) *handoff = 2;   // no phantom round in rule 1
    }
    return 0;
}

int ptnn_swap_L_ptr(ptnn_handle* h, int phantom, void** dev_ptr) {
    if (!h || !dev_ptr) return fail(-1, "null argument");
    *dev_ptr = phantom ? h->d_L_final : h->d_L_handoff;
    return 0;
}

int ptnn_swap_set_L(ptnn_handle* h, int phantom, const float* L_host) {
    if (!h || !L_host) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = wait_stream(h)) return rc;
    HIP_TRY(hipMemcpy(phantom ? h->d_L_final : h->d_L_handoff, L_host, h->cfg.n_replicas_global * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int ptnn_swap_cascade(ptnn_handle* h, int phantom, int32_t* src_host) {
    if (!h || !src_host) return fail(-1, "null argument");
    if (h->cfg.swap_rule != 0)
        return fail(-3, "the point-to-point exchange implements the reference's cascade (swap_rule 0) only: use the gathered mode");
    if (int rc = launch_swap(h, phantom != 0, 0, true)) return rc;
    const size_t bytes = h->cfg.n_replicas_global * sizeof(int);
    HIP_TRY(hipMemcpyAsync(h->h_src, h->d_src, bytes, hipMemcpyDeviceToHost, h->stream));   // pinned: no staging copy
    if (int rc = wait_stream(h)) return rc;
    std::memcpy(src_host, h->h_src, bytes);
    return 0;
}

int ptnn_state_row_floats(ptnn_handle* h) { return h ? h->PS : -1; }

int ptnn_stream(ptnn_handle* h, void** hip_stream) {
    if (!h || !hip_stream) return fail(-1, "null argument");
    *hip_stream = reinterpret_cast<void*>(h->stream);
    return 0;
}

int ptnn_swap_row_ptr(ptnn_handle* h, int local_replica, void** cur_row, void** next_row) {
    if (!h) return fail(-1, "null handle");
    if (local_replica < 0 || local_replica >= h->cfg.n_replicas_local) return fail(-1, "replica %d out of range", local_replica);
    if (cur_row) *cur_row = h->d_state[h->flip] + (size_t)local_replica * h->PS;
    if (next_row) *next_row = h->d_state[h->flip ^ 1] + (size_t)local_replica * h->PS;
    return 0;
}

int ptnn_swap_apply(ptnn_handle* h, const int32_t* src_host, int phantom) {
    if (int rc = check_ready(h)) return rc;
    if (h->cfg.label_swap) return fail(-3, "label swapping moves no rows: drive it with ptnn_run");
    (void)src_host;   // the device recomputes the identical cascade; the host copy only routed the remote rows
    if (int rc = launch_swap(h, phantom != 0, phantom ? 2 : 3, false)) return rc;
    if (!phantom) h->flip ^= 1;
    h->rounds_done += 1;
    return 0;
}

int ptnn_xchg_ptr(ptnn_handle* h, void** base, int* row_floats) {
    if (!h || !base || !row_floats) return fail(-1, "null argument");
    *base = h->d_xchg;
    *row_floats = xchg_row_floats(h->PS);
    return 0;
}

int ptnn_swap_pack(ptnn_handle* h, int phantom) {
    if (int rc = check_ready(h)) return rc;
    if (h->cfg.label_swap) return fail(-3, "label swapping moves no rows: drive it with ptnn_run");
    if (h->cfg.swap_rule == 1 && !h->have_ladder) return fail(-1, "swap_rule 1 needs ptnn_set_ladder (all temperatures)");
    return launch_swap(h, phantom != 0, -1, false);
}

int ptnn_swap_apply_gathered(ptnn_handle* h, int phantom) {
    if (int rc = check_ready(h)) return rc;
    if (h->cfg.label_swap) return fail(-3, "label swapping moves no rows: drive it with ptnn_run");
    if (int rc = launch_swap(h, phantom != 0, phantom ? (2 | 4) : (3 | 4), false)) return rc;
    if (!phantom) h->flip ^= 1;
    h->rounds_done += 1;
    return 0;
}

int ptnn_get_traces(ptnn_handle* h, int step0, int nsteps, float* pos_w, float* likeh, float* rmse_train,
                    float* rmse_test, float* acc_train, float* acc_test, int32_t* accept_count) {
    if (int rc = check_ready(h)) return rc;
    const int S = h->cfg.n_samples, Rl = h->cfg.n_replicas_local, P = h->P, cap = h->cap;
    if (step0 < 0 || nsteps < 0 || step0 + nsteps > S) return fail(-1, "trace range [%d, %d) outside [0, %d)", step0, step0 + nsteps, S);
    if (nsteps == 0) return 0;
    if (step0 + nsteps > h->cur + 1) return fail(-1, "rows up to %d requested but only %d MH steps have been queued", step0 + nsteps - 1, h->cur);
    if (step0 < h->first_row) return fail(-1, "rows below %d were produced before the checkpoint these chains were restored from", h->first_row);
    if (step0 < h->cur + 1 - cap) return fail(-1, "row %d has already been overwritten in the trace ring (capacity %d, %d steps done)", step0, cap, h->cur);
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    collect_timing(h);
    // the range may wrap around the ring: at most two contiguous pieces
    auto copy2d = [&](void* dst, const void* src_base, size_t elem_bytes, size_t per_step) -> hipError_t {
        if (!dst) return hipSuccess;
        const size_t dpitch = (size_t)nsteps * per_step * elem_bytes;
        const size_t spitch = (size_t)cap * per_step * elem_bytes;
        int done = 0;
        while (done < nsteps) {
            const int slot = (step0 + done) % cap;
            const int n = std::min(nsteps - done, cap - slot);
            const size_t width = (size_t)n * per_step * elem_bytes;
            const char* src = static_cast<const char*>(src_base) + (size_t)slot * per_step * elem_bytes;
            char* d = static_cast<char*>(dst) + (size_t)done * per_step * elem_bytes;
            hipError_t e = hipMemcpy2D(d, dpitch, src, spitch, width, Rl, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return e;
            done += n;
        }
        return hipSuccess;
    };
    std::vector<float> rows;
    const bool want_scalars = likeh || rmse_train || rmse_test || acc_train || acc_test || accept_count;
    if (want_scalars || (pos_w && h->compact)) {
        // the scalars of a step sit in one 32-byte row on the device (one sector per step instead of seven); the per-file
        // arrays of the reference's layout (REG:454-481) are split out below
        rows.resize((size_t)Rl * nsteps * TR_COUNT);
        HIP_TRY(copy2d(rows.data(), h->d_scal, sizeof(float), TR_COUNT));
    }
    if (pos_w && h->compact) {
        // compact traces (wide nets, every row resident): a rejected step wrote no pos_w row, only the index of the row it
        // repeats (pos_w[i+1] = pos_w[i], REG:417).  Fetch every distinct source row once and fill the repeats in on the host.
        const size_t PW = h->PW;
        for (int r = 0; r < Rl; ++r) {
            int prev = -1;
            for (int t = 0; t < nsteps; ++t) {
                int32_t src;
                std::memcpy(&src, &rows[((size_t)r * nsteps + t) * TR_COUNT + TR_SRC], sizeof src);
                if (src < 0 || src > step0 + t || src < h->first_row - 1)
                    return fail(-2, "trace row %d of replica %d refers to row %d (internal error)", step0 + t, r, src);
                float* dst = pos_w + ((size_t)r * nsteps + t) * P;
                if (t > 0 && src == prev) std::memcpy(dst, dst - P, (size_t)P * sizeof(float));
                else HIP_TRY(hipMemcpy(dst, h->d_pos_w + ((size_t)r * cap + (size_t)(src % cap)) * PW, (size_t)P * sizeof(float), hipMemcpyDeviceToHost));
                prev = src;
            }
        }
    } else if (pos_w) {
        // device rows are padded to PW floats (whole sectors); the caller's array is dense: one strided copy per replica
        // and ring piece
        const size_t PW = h->PW;
        for (int r = 0; r < Rl; ++r) {
            int done = 0;
            while (done < nsteps) {
                const int slot = (step0 + done) % cap;
                const int n = std::min(nsteps - done, cap - slot);
                HIP_TRY(hipMemcpy2D(pos_w + ((size_t)r * nsteps + done) * P, (size_t)P * sizeof(float),
                                    h->d_pos_w + ((size_t)r * cap + slot) * PW, PW * sizeof(float), (size_t)P * sizeof(float), n,
                                    hipMemcpyDeviceToHost));
                done += n;
            }
        }
    }
    if (want_scalars) {
        float* outs[5] = {likeh, rmse_train, rmse_test, acc_train, acc_test};
        const int cols[5] = {TR_LIKEH, TR_RMSE_TR, TR_RMSE_TE, TR_ACC_TR, TR_ACC_TE};
        const size_t n = (size_t)Rl * nsteps;
        for (int c = 0; c < 5; ++c)
            if (outs[c])
                for (size_t k = 0; k < n; ++k) outs[c][k] = rows[k * TR_COUNT + cols[c]];
        // a regression has no accuracy (acc_train[i+1] = 0 on every step, REG:403): its TR_ACC_TR slot records eta (ptnn_device.hpp:
        // finish_eval<TASK, true>; ptnn_get_trace_rows shows it), the array of the reference's layout is zeros
        if (acc_train && h->cfg.task == PTNN_TASK_REG) std::fill(acc_train, acc_train + n, 0.0f);
        if (accept_count)
            for (size_t k = 0; k < n; ++k) std::memcpy(&accept_count[k], &rows[k * TR_COUNT + TR_ACCEPT], sizeof(int32_t));
    }
    h->drained = std::max(h->drained, step0 + nsteps - 1);
    return 0;
}

int ptnn_get_trace_rows(ptnn_handle* h, int step0, int nsteps, float* rows) {
    if (int rc = check_ready(h)) return rc;
    if (!rows) return fail(-1, "null argument");
    const int S = h->cfg.n_samples, Rl = h->cfg.n_replicas_local, cap = h->cap;
    if (step0 < 0 || nsteps < 1 || step0 + nsteps > S) return fail(-1, "trace range [%d, %d) outside [0, %d)", step0, step0 + nsteps, S);
    if (step0 + nsteps > h->cur + 1) return fail(-1, "rows up to %d requested but only %d MH steps have been queued", step0 + nsteps - 1, h->cur);
    if (step0 < h->first_row || step0 < h->cur + 1 - cap) return fail(-1, "row %d is no longer (or was never) on this device", step0);
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    const size_t rowb = TR_COUNT * sizeof(float);
    int done = 0;
    while (done < nsteps) {                                  // the range may wrap around the ring
        const int slot = (step0 + done) % cap;
        const int n = std::min(nsteps - done, cap - slot);
        HIP_TRY(hipMemcpy2D(reinterpret_cast<char*>(rows) + (size_t)done * rowb, (size_t)nsteps * rowb,
                            reinterpret_cast<const char*>(h->d_scal) + (size_t)slot * rowb, (size_t)cap * rowb, (size_t)n * rowb, Rl,
                            hipMemcpyDeviceToHost));
        done += n;
    }
    return 0;
}

// ---- trace images: the download of the trace rows overlapped with sampling ----
// The reference's chains write their files after their last step and the parent reads them back (REG:454-481, 775-871); here the
// rows of the steps queued so far can leave for the host while the steps queued behind them are being sampled: a pinned host copy
// of the device's trace arrays (same layout, so a range of rows is one strided copy per array) filled by a second stream behind an
// event of the handle's stream.
int ptnn_trace_image(ptnn_handle* h, float** pos_w, int32_t* row_floats, float** rows) {
    if (int rc = check_ready(h)) return rc;
    if (!pos_w || !row_floats || !rows) return fail(-1, "null argument");
    const int S = h->cfg.n_samples, Rl = h->cfg.n_replicas_local;
    if (h->cap != S) return fail(-1, "trace images need every row resident (trace_capacity 0 or >= n_samples; this handle keeps a ring of %d)", h->cap);
    if (h->compact) return fail(-1, "trace images are not available with compact traces (wide nets): fetch with ptnn_get_traces");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (!h->h_img_pos) {
        // all three or none: a half-made set must not be handed out by the next call
        float *pos = nullptr, *rws = nullptr;
        hipStream_t st = nullptr;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&pos), (size_t)Rl * S * h->PW * sizeof(float), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&rws), (size_t)Rl * S * TR_COUNT * sizeof(float), hipHostMallocDefault);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) {
            if (pos) (void)hipHostFree(pos);
            if (rws) (void)hipHostFree(rws);
            (void)hipGetLastError();
            return fail(-2, "pinned trace images of %zu MB: %s", ((size_t)Rl * S * (h->PW + TR_COUNT) * sizeof(float)) >> 20, hipGetErrorString(e));
        }
        h->h_img_pos = pos; h->h_img_rows = rws; h->copy_stream = st;
    }
    *pos_w = h->h_img_pos; *row_floats = h->PW; *rows = h->h_img_rows;
    return 0;
}

int ptnn_trace_image_fetch(ptnn_handle* h, int step0, int nsteps) {
    if (int rc = check_ready(h)) return rc;
    if (!h->h_img_pos) return fail(-1, "call ptnn_trace_image first");
    if (h->failed) return fail(-5, "%s", h->failure.c_str());
    const int S = h->cfg.n_samples, Rl = h->cfg.n_replicas_local;
    if (step0 < 0 || nsteps < 1 || step0 + nsteps > S) return fail(-1, "trace range [%d, %d) outside [0, %d)", step0, step0 + nsteps, S);
    if (step0 + nsteps > h->cur + 1) return fail(-1, "rows up to %d requested but only %d MH steps have been queued", step0 + nsteps - 1, h->cur);
    if (step0 < h->first_row) return fail(-1, "rows below %d were produced before the checkpoint these chains were restored from", h->first_row);
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    hipEvent_t queued = nullptr, landed = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&queued, hipEventDisableTiming));
    hipError_t e = hipEventRecord(queued, h->stream);                         // everything queued so far: the steps that write these rows
    if (e == hipSuccess) e = hipStreamWaitEvent(h->copy_stream, queued, 0);
    (void)hipEventDestroy(queued);                                           // released once it has completed
    HIP_TRY(e);
    // one strided copy per array (a replica's rows [step0, step0 + nsteps) are contiguous; replicas lie S rows apart in both layouts).
    // Measured against one linear copy per replica and array (128 enqueues per window for 64 replicas): the rows land 1 ms earlier,
    // and neither shape takes time from the kernels -- Mackey-Glass, every CU full of work-groups that wait for each other: 28.0 ms
    // of kernel time with the copies in flight against 27.9 ms without (profiles/tools/chunk_probe.py)
    const size_t PWb = (size_t)h->PW * sizeof(float), RWb = (size_t)TR_COUNT * sizeof(float);
    HIP_TRY(hipMemcpy2DAsync(reinterpret_cast<char*>(h->h_img_pos) + (size_t)step0 * PWb, (size_t)S * PWb,
                             reinterpret_cast<const char*>(h->d_pos_w) + (size_t)step0 * PWb, (size_t)S * PWb, (size_t)nsteps * PWb, Rl,
                             hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(hipMemcpy2DAsync(reinterpret_cast<char*>(h->h_img_rows) + (size_t)step0 * RWb, (size_t)S * RWb,
                             reinterpret_cast<const char*>(h->d_scal) + (size_t)step0 * RWb, (size_t)S * RWb, (size_t)nsteps * RWb, Rl,
                             hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(hipEventCreateWithFlags(&landed, hipEventDisableTiming));
    e = hipEventRecord(landed, h->copy_stream);
    if (e != hipSuccess) { (void)hipEventDestroy(landed); HIP_TRY(e); }
    h->img_events.push_back(landed);
    h->drained = std::max(h->drained, step0 + nsteps - 1);
    return (int)h->img_events.size() - 1;
}

int ptnn_trace_image_wait(ptnn_handle* h, int ticket) {
    if (!h) return fail(-1, "null handle");
    if (ticket < 0 || ticket >= (int)h->img_events.size() || !h->img_events[(size_t)ticket]) return fail(-1, "no such ticket");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    HIP_TRY(hipEventSynchronize(h->img_events[(size_t)ticket]));
    return 0;
}

int ptnn_get_swap_stats(ptnn_handle* h, int64_t* num_swap, int64_t* total_proposals, int32_t* rounds_done) {
    if (!h) return fail(-1, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    long long c[2];
    HIP_TRY(hipMemcpy(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost));
    if (num_swap) *num_swap = c[0];
    if (total_proposals) *total_proposals = c[1];
    if (rounds_done) *rounds_done = h->rounds_done;
    return 0;
}

int ptnn_get_labels(ptnn_handle* h, int32_t* label) {
    if (!h || !label) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    HIP_TRY(hipMemcpy(label, h->d_label[h->lflip], (size_t)h->cfg.n_replicas_global * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

int ptnn_get_swap_log(ptnn_handle* h, int32_t* src, int max_rounds) {
    if (!h || !src) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    const int n = std::min(max_rounds, std::min(h->rounds_done, h->max_rounds));
    if (n > 0) HIP_TRY(hipMemcpy(src, h->d_src_log, (size_t)n * h->cfg.n_replicas_global * sizeof(int), hipMemcpyDeviceToHost));
    return n;
}

int ptnn_get_state(ptnn_handle* h, float* w, float* eta, float* likelihood, float* prior, int32_t* num_accepted,
                   int32_t* langevin_count, int32_t* langevin_accepted) {
    if (int rc = check_ready(h)) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    const int Rl = h->cfg.n_replicas_local, P = h->P, PS = h->PS;
    std::vector<float> st((size_t)Rl * PS), sf((size_t)Rl * SF_COUNT);
    std::vector<int> si((size_t)Rl * SI_COUNT);
    HIP_TRY(hipMemcpy(st.data(), h->d_state[h->flip], st.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sf.data(), h->d_st_f, sf.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(si.data(), h->d_st_i, si.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int r = 0; r < Rl; ++r) {
        if (w) std::memcpy(w + (size_t)r * P, &st[(size_t)r * PS], P * sizeof(float));
        if (eta) eta[r] = st[(size_t)r * PS + P];
        if (likelihood) likelihood[r] = sf[(size_t)r * SF_COUNT + SF_LIK];
        if (prior) prior[r] = sf[(size_t)r * SF_COUNT + SF_PRIOR];
        if (num_accepted) num_accepted[r] = si[(size_t)r * SI_COUNT + SI_NACC];
        if (langevin_count) langevin_count[r] = si[(size_t)r * SI_COUNT + SI_LG_COUNT];
        if (langevin_accepted) langevin_accepted[r] = si[(size_t)r * SI_COUNT + SI_LG_ACC];
    }
    return 0;
}

// ---- checkpoint / resume: the RNG is counter based, so the chain state is small and a restored handle
// continues the chains bit for bit.  Traces are not part of it: the caller keeps the rows it has fetched. ----
namespace {
struct CkHeader {
    uint32_t magic, version;
    ptnn_config cfg;
    int32_t P, PS, cur, rounds_done, finalized, have_ladder, log_rounds, reserved;
    long long counters[2];
};
constexpr uint32_t CK_MAGIC = 0x4b435450u;      // "PTCK"

size_t ck_bytes(const ptnn_handle* h) {
    const size_t Rl = h->cfg.n_replicas_local, R = h->cfg.n_replicas_global, PS = h->PS;
    const size_t logr = (size_t)std::min(h->rounds_done, h->max_rounds);
    return sizeof(CkHeader) + sizeof(float) * (3 * Rl * PS + Rl * SF_COUNT + Rl + 5 * R) + sizeof(int) * (Rl + Rl * SI_COUNT + logr * R + 2 * R);
}

bool same_chain(const ptnn_config& a, const ptnn_config& b) {
    return a.task == b.task && a.n_in == b.n_in && a.n_hidden == b.n_hidden && a.n_out == b.n_out &&
           a.n_replicas_local == b.n_replicas_local && a.n_replicas_global == b.n_replicas_global &&
           a.first_global_replica == b.first_global_replica && a.n_samples == b.n_samples && a.swap_interval == b.swap_interval &&
           a.pt_switch_step == b.pt_switch_step && a.use_langevin == b.use_langevin && a.swap_rule == b.swap_rule &&
           a.shared_noise == b.shared_noise && a.label_swap == b.label_swap && a.forward_bf16 == b.forward_bf16 && a.l_prob == b.l_prob &&
           a.learn_rate == b.learn_rate && a.step_w == b.step_w && a.step_eta == b.step_eta && a.sigma_squared == b.sigma_squared &&
           a.nu_1 == b.nu_1 && a.nu_2 == b.nu_2 && a.seed == b.seed;
}
}  // namespace

int ptnn_checkpoint_size(ptnn_handle* h, int64_t* bytes) {
    if (int rc = check_ready(h)) return rc;
    if (!bytes) return fail(-1, "null argument");
    *bytes = (int64_t)ck_bytes(h);
    return 0;
}

int ptnn_checkpoint_save(ptnn_handle* h, void* buf, int64_t bytes) {
    if (int rc = check_ready(h)) return rc;
    if (!buf || bytes < (int64_t)ck_bytes(h)) return fail(-1, "checkpoint buffer too small: %lld < %zu", (long long)bytes, ck_bytes(h));
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = finish_stream(h)) return rc;
    const size_t Rl = h->cfg.n_replicas_local, R = h->cfg.n_replicas_global, PS = h->PS;
    CkHeader hd{};
    hd.magic = CK_MAGIC; hd.version = 3; hd.cfg = h->cfg; hd.P = h->P; hd.PS = h->PS; hd.cur = h->cur;
    hd.rounds_done = h->rounds_done; hd.finalized = h->finalized ? 1 : 0; hd.have_ladder = h->have_ladder ? 1 : 0;
    hd.log_rounds = std::min(h->rounds_done, h->max_rounds);
    HIP_TRY(hipMemcpy(hd.counters, h->d_counters, sizeof(hd.counters), hipMemcpyDeviceToHost));
    char* q = static_cast<char*>(buf);
    std::memcpy(q, &hd, sizeof(hd)); q += sizeof(hd);
    auto get = [&](const void* dev, size_t n) -> int {
        if (n) HIP_TRY(hipMemcpy(q, dev, n, hipMemcpyDeviceToHost));
        q += n;
        return 0;
    };
    if (int rc = get(h->d_state[h->flip], sizeof(float) * Rl * PS)) return rc;
    if (int rc = get(h->d_gd_w[h->flip], sizeof(float) * Rl * PS)) return rc;
    if (int rc = get(h->d_rec_w, sizeof(float) * Rl * PS)) return rc;
    if (int rc = get(h->d_st_f, sizeof(float) * Rl * SF_COUNT)) return rc;
    if (int rc = get(h->d_temps, sizeof(float) * Rl)) return rc;
    if (int rc = get(h->d_L_handoff, sizeof(float) * R)) return rc;
    if (int rc = get(h->d_L_final, sizeof(float) * R)) return rc;
    if (int rc = get(h->d_L_raw, sizeof(float) * R)) return rc;
    if (int rc = get(h->d_prior_post, sizeof(float) * R)) return rc;
    if (int rc = get(h->d_temps_global, sizeof(float) * R)) return rc;
    if (int rc = get(h->d_gd_valid[h->flip], sizeof(int) * Rl)) return rc;
    if (int rc = get(h->d_st_i, sizeof(int) * Rl * SI_COUNT)) return rc;
    if (int rc = get(h->d_src_log, sizeof(int) * (size_t)hd.log_rounds * R)) return rc;
    if (int rc = get(h->d_label[h->lflip], sizeof(int) * R)) return rc;          // slot <-> temperature maps (identity unless label_swap)
    if (int rc = get(h->d_slot_of[h->lflip], sizeof(int) * R)) return rc;
    return 0;
}

int ptnn_checkpoint_load(ptnn_handle* h, const void* buf, int64_t bytes) {
    if (!h || !buf) return fail(-1, "null argument");
    if (!h->have_data) return fail(-1, "call ptnn_set_data before ptnn_checkpoint_load");
    if (bytes < (int64_t)sizeof(CkHeader)) return fail(-1, "not a checkpoint (too short)");
    CkHeader hd;
    std::memcpy(&hd, buf, sizeof(hd));
    if (hd.magic != CK_MAGIC || hd.version != 3) return fail(-1, "not a libptnn checkpoint (magic %08x version %u)", hd.magic, hd.version);
    if (!same_chain(hd.cfg, h->cfg) || hd.P != h->P || hd.PS != h->PS)
        return fail(-1, "the checkpoint was written by chains with a different configuration (topology, replicas, samples, seed ...)");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = wait_stream(h)) return rc;
    const size_t Rl = h->cfg.n_replicas_local, R = h->cfg.n_replicas_global, PS = h->PS;
    const size_t need = sizeof(CkHeader) + sizeof(float) * (3 * Rl * PS + Rl * SF_COUNT + Rl + 5 * R) +
                        sizeof(int) * (Rl + Rl * SI_COUNT + (size_t)hd.log_rounds * R + 2 * R);
    if ((size_t)bytes < need) return fail(-1, "truncated checkpoint: %lld < %zu bytes", (long long)bytes, need);
    if (hd.log_rounds > h->max_rounds) return fail(-1, "checkpoint holds more swap rounds than this handle can log");
    const char* q = static_cast<const char*>(buf) + sizeof(CkHeader);
    auto put = [&](void* dev, size_t n) -> int {
        if (n) HIP_TRY(hipMemcpy(dev, q, n, hipMemcpyHostToDevice));
        q += n;
        return 0;
    };
    h->flip = 0;
    if (int rc = put(h->d_state[0], sizeof(float) * Rl * PS)) return rc;
    if (int rc = put(h->d_gd_w[0], sizeof(float) * Rl * PS)) return rc;
    if (int rc = put(h->d_rec_w, sizeof(float) * Rl * PS)) return rc;
    if (int rc = put(h->d_st_f, sizeof(float) * Rl * SF_COUNT)) return rc;
    if (int rc = put(h->d_temps, sizeof(float) * Rl)) return rc;
    if (int rc = put(h->d_L_handoff, sizeof(float) * R)) return rc;
    if (int rc = put(h->d_L_final, sizeof(float) * R)) return rc;
    if (int rc = put(h->d_L_raw, sizeof(float) * R)) return rc;
    if (int rc = put(h->d_prior_post, sizeof(float) * R)) return rc;
    if (int rc = put(h->d_temps_global, sizeof(float) * R)) return rc;
    if (int rc = put(h->d_gd_valid[0], sizeof(int) * Rl)) return rc;
    if (int rc = put(h->d_st_i, sizeof(int) * Rl * SI_COUNT)) return rc;
    if (int rc = put(h->d_src_log, sizeof(int) * (size_t)hd.log_rounds * R)) return rc;
    h->lflip = 0;
    if (int rc = put(h->d_label[0], sizeof(int) * R)) return rc;
    if (int rc = put(h->d_slot_of[0], sizeof(int) * R)) return rc;
    HIP_TRY(hipMemcpy(h->d_state[1], h->d_state[0], sizeof(float) * Rl * PS, hipMemcpyDeviceToDevice));
    if (h->compact) {
        // compact traces: the rows a later rejected step may repeat are not on this device -- put the recorded row of every chain
        // into trace row hd.cur (the last one before the checkpoint) and point the chains at it
        HIP_TRY(hipMemcpy2D(h->d_pos_w + (size_t)(hd.cur % h->cap) * h->PW, (size_t)h->cap * h->PW * sizeof(float), h->d_rec_w,
                            PS * sizeof(float), (size_t)h->P * sizeof(float), Rl, hipMemcpyDeviceToDevice));
        std::vector<int> si(Rl * SI_COUNT);
        HIP_TRY(hipMemcpy(si.data(), h->d_st_i, si.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t r = 0; r < Rl; ++r) si[r * SI_COUNT + SI_REC_ROW] = hd.cur;
        HIP_TRY(hipMemcpy(h->d_st_i, si.data(), si.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(h->d_counters, hd.counters, sizeof(hd.counters), hipMemcpyHostToDevice));
    h->cur = hd.cur; h->rounds_done = hd.rounds_done; h->finalized = hd.finalized != 0; h->have_ladder = hd.have_ladder != 0;
    h->drained = hd.cur; h->first_row = hd.cur + 1;
    HIP_TRY(hipMemset(h->d_error, 0, sizeof(int)));
    if (!h->comm.failed) { h->failed = false; h->failure.clear(); }
    h->h_progress[0] = h->h_progress[1] = hd.rounds_done;
    h->have_state = true;
    return 0;
}

static int run_model(ptnn_handle* h, int mode, const float* w_in, const float* tau_sq, int n, float* out, size_t out_floats,
                     int a0, int a1) {
    if (!h) return fail(-1, "null handle");
    if (!h->have_data) return fail(-1, "ptnn_set_data has not been called");
    if (n < 1) return fail(-1, "n must be >= 1");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    float *d_w = nullptr, *d_tau = nullptr, *d_out = nullptr;
    const int P = h->P;
    if (w_in) {
        HIP_TRY(hipMalloc(&d_w, (size_t)n * P * sizeof(float)));
        HIP_TRY(hipMemcpy(d_w, w_in, (size_t)n * P * sizeof(float), hipMemcpyHostToDevice));
    }
    if (tau_sq) {
        HIP_TRY(hipMalloc(&d_tau, (size_t)n * sizeof(float)));
        HIP_TRY(hipMemcpy(d_tau, tau_sq, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&d_out, out_floats * sizeof(float)));
    HIP_TRY(hipMemsetAsync(d_out, 0, out_floats * sizeof(float), h->stream));
    const SegParams p = h->seg_params();
    // mode 4 (ptnn_time_tree_round): ONE input row, but 9 blocks -- blocks 0 and 8 share an XCD under the round-robin dispatch
    hipLaunchKernelGGL(h->wide ? h->shape->model_wide : h->shape->model, dim3(mode == 4 ? 9 : n), dim3(h->model_threads), h->model_lds, h->stream, p,
                       mode, d_w, d_tau, d_out, a0, a1);
    HIP_TRY(hipGetLastError());
    if (int rc = wait_stream(h)) return rc;
    HIP_TRY(hipMemcpy(out, d_out, out_floats * sizeof(float), hipMemcpyDeviceToHost));
    if (d_w) (void)hipFree(d_w);
    if (d_tau) (void)hipFree(d_tau);
    (void)hipFree(d_out);
    return 0;
}

int ptnn_evaluate(ptnn_handle* h, const float* w, const float* tau_sq, int n, float* out) {
    if (!w || !out) return fail(-1, "null argument");
    if (h && h->cfg.task == PTNN_TASK_REG && !tau_sq) return fail(-1, "regression needs tau_sq");
    return run_model(h, 0, w, tau_sq, n, out, (size_t)n * 8, 0, 0);
}

int ptnn_langevin_gradient(ptnn_handle* h, const float* w_in, int n, float* w_out) {
    if (!w_in || !w_out) return fail(-1, "null argument");
    return run_model(h, 1, w_in, nullptr, n, w_out, (size_t)n * (h ? h->P : 0), 0, 0);
}

int ptnn_time_sgd_epoch(ptnn_handle* h, const float* w, int reps, double* ms_per_epoch) {
    if (!h || !w || !ms_per_epoch || reps < 1) return fail(-1, "bad argument");
    float out[4] = {0.f, 0.f, 0.f, 0.f};
    if (int rc = run_model(h, 3, w, nullptr, 1, out, 4, reps, 0)) return rc;
    int khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id));
    if (khz <= 0) return fail(-2, "the device reports no wall clock rate");
    auto ticks = [&](int k) {
        unsigned lo, hi;
        std::memcpy(&lo, &out[k], 4); std::memcpy(&hi, &out[k + 1], 4);
        return (double)(((unsigned long long)hi << 32) | lo);
    };
    ms_per_epoch[0] = ticks(0) / (double)khz / (double)reps;
    // wide nets (n_hidden > 64): [1] = a PAIR of epochs through one row loop (sgd_sweep_wide_pair); narrow nets: 0
    ms_per_epoch[1] = h->wide ? ticks(2) / (double)khz / (double)reps : 0.0;
    return 0;
}

int ptnn_time_tree_round(ptnn_handle* h, const float* w, int reps, int xcd_local, double* ms) {
    if (!h || !w || !ms || reps < 1) return fail(-1, "bad argument");
    if (h->wide) return fail(-3, "the prefetching tree runs nets of up to 64 hidden units");
    float out[64] = {0.f};
    if (int rc = run_model(h, 4, w, nullptr, 1, out, 64, reps, xcd_local ? 1 : 0)) return rc;
    int khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id));
    if (khz <= 0) return fail(-2, "the device reports no wall clock rate");
    if (out[4] != 1.0f) return fail(-2, "the granule round trips between two work-groups timed out");
    auto ticks = [&](int k) {
        unsigned lo, hi;
        std::memcpy(&lo, &out[k], 4); std::memcpy(&hi, &out[k + 1], 4);
        return (double)(((unsigned long long)hi << 32) | lo);
    };
    ms[0] = ticks(0) / (double)khz / (double)reps;             // one forward pass + likelihood + prior sums of a whole work-group
    ms[1] = ticks(2) / (double)khz / (double)reps / 2.0;       // one granule, one way (half a round trip)
    ms[2] = (double)out[5];                                    // 1: the round trips went through the XCD's L2
    return 0;
}

int ptnn_tape(ptnn_handle* h, int replica, int step, float* noise, float* scal) {
    if (!h || !noise || !scal) return fail(-1, "null argument");
    // narrow nets return {noise[P], scal[3]}; the wide kernel writes whole float4s: {noise[PS], scal[3]}
    const size_t off = h->wide ? (size_t)h->PS : (size_t)h->P;
    std::vector<float> buf(off + 3);
    if (int rc = run_model(h, 2, nullptr, nullptr, 1, buf.data(), buf.size(), replica, step)) return rc;
    std::memcpy(noise, buf.data(), h->P * sizeof(float));
    std::memcpy(scal, buf.data() + off, 3 * sizeof(float));
    return 0;
}

int ptnn_describe(ptnn_handle* h, char* buf, int nbytes) {
    if (!h || !buf || nbytes < 1) return fail(-1, "bad argument");
    if (!h->have_data) return fail(-1, "ptnn_set_data has not been called (the schedule depends on the data set)");
    const char* kern = h->wide ? (h->wide_res ? "segment_wide_res_kernel" : "segment_wide_kernel") : (h->tree ? "segment_tree_kernel" : (h->packed ? (h->groups > 1 ? "segment_packm_kernel" : "segment_pack_kernel") : (h->speculative ? "segment_spec_kernel" : "segment_kernel")));
    const void* fn = reinterpret_cast<const void*>(segment_function(h));
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, h->nthreads, h->seg_lds));
    hipFuncAttributes fa{};
    HIP_TRY(hipFuncGetAttributes(&fa, fn));
    const int grid = h->cfg.n_replicas_local * ((h->speculative || h->tree || h->wide) ? h->groups : 1);
    // tree: the steps committed per round (its depth)
    const int slots = h->tree ? tree_depth(h->groups) : (h->wide ? (h->groups > 1 && h->cfg.use_langevin ? 8 : h->groups) : !h->speculative ? 1 : (h->packed ? pack_slots(h->pk_nred) * h->groups : h->groups * (h->nthreads / WAVE)));
    const int n = std::snprintf(buf, (size_t)nbytes,
                                "{\"kernel\": \"ptnn::%s<%d,%d,%d>\", \"schedule\": \"%s\", \"grid_blocks\": %d, \"block_threads\": %d, "
                                "\"lds_bytes\": %zu, \"groups_per_replica\": %d, \"slots_per_round\": %d, \"num_cus\": %d, "
                                "\"blocks_per_cu\": %d, \"vgprs\": %d, \"scratch_bytes\": %zu, \"forward_mfma\": %d, \"exchange\": \"%s\", \"lds_resident_state\": %d, \"compact_traces\": %d, \"launches\": \"%s\"}",
                                kern, h->cfg.task, h->cfg.n_in, h->cfg.n_out,
                                h->wide ? (h->groups > 1 ? "speculative-wide" : "cooperative-wide") : (h->tree ? "prefetching-tree" : (h->packed ? "packed-speculative" : (h->speculative ? "speculative" : "cooperative"))),
                                grid, h->nthreads, h->seg_lds, h->groups, slots, h->num_cus, per_cu, fa.numRegs, (size_t)fa.localSizeBytes,
                                h->fw_mfma ? h->fw_mfma : ((h->wide && h->cfg.n_hidden % 32 == 0) ? 1 : 0),
                                h->comm.kind == COMM_NONE ? "none" : (h->cfg.label_swap ? "labels" : (resolved_xchg_mode(h) == PTNN_XCHG_GATHER ? "gather" : "boundary")),
                                h->wide_res ? 1 : 0, h->compact ? 1 : 0,
                                (h->persistent && h->comm.kind == COMM_NONE) ? "one per ptnn_run (swap rounds inside)" : "one per swap interval");
    if (n < 0 || n >= nbytes) return fail(-1, "buffer of %d bytes is too small for the description", nbytes);
    return n;
}

int ptnn_kernel_time(ptnn_handle* h, int reset, int64_t* launches, double* total_ms) {
    if (!h) return fail(-1, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = wait_stream(h)) return rc;
    collect_timing(h);
    if (launches) *launches = h->timed_launches;
    if (total_ms) *total_ms = h->timed_ms;
    if (reset) { h->timed_launches = 0; h->timed_ms = 0.0; }
    return 0;
}

int ptnn_debug_stamps(ptnn_handle* h, uint64_t* out16) {   // 160 entries: 16 phase sums + 64 x (cycles, rounds)
    if (!h || !out16) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = wait_stream(h)) return rc;
    HIP_TRY(hipMemcpy(out16, h->d_stamps, 160 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(h->d_stamps, 0, 160 * sizeof(unsigned long long)));
    const unsigned long long big = ~0ull;
    HIP_TRY(hipMemcpy(h->d_stamps + 12, &big, sizeof big, hipMemcpyHostToDevice));
    return 0;
}

// exactly one floating conversion: % [flags] [width] [.precision] (e|E|f|F|g|G)
static bool float_format_ok(const char* fmt) {
    const size_t fl = std::strlen(fmt);
    bool ok = fl >= 2 && fl < 16 && fmt[0] == '%' && std::strchr("eEfFgG", fmt[fl - 1]) != nullptr;
    for (size_t k = 1; ok && k + 1 < fl; ++k) ok = std::strchr("0123456789.+- #", fmt[k]) != nullptr;
    // width and precision stay far inside the 400-byte slot ptnn_savetxt formats a value into
    for (size_t k = 1; ok && k + 1 < fl;) {
        if (fmt[k] >= '0' && fmt[k] <= '9') {
            long v = 0;
            while (k + 1 < fl && fmt[k] >= '0' && fmt[k] <= '9') v = v * 10 + (fmt[k++] - '0');
            ok = v <= 40;
        } else ++k;
    }
    return ok;
}

int ptnn_text_round(double* values, int64_t n, const char* fmt) {
    if (!values || !fmt || n < 0) return fail(-1, "bad argument");
    if (!float_format_ok(fmt)) return fail(-1, "unsupported format '%s'", fmt);
    const ptnn_text::Format f = ptnn_text::parse_format(fmt);
    for (int64_t k = 0; k < n; ++k) values[k] = ptnn_text::round_trip(values[k], f);
    return 0;
}

int ptnn_text_round_f32(const float* in, double* out, int64_t n, const char* fmt) {
    if (!in || !out || !fmt || n < 0) return fail(-1, "bad argument");
    if (!float_format_ok(fmt)) return fail(-1, "unsupported format '%s'", fmt);
    const ptnn_text::Format f = ptnn_text::parse_format(fmt);
    for (int64_t k = 0; k < n; ++k) out[k] = ptnn_text::round_trip((double)in[k], f);
    return 0;
}

}  // extern "C" (the row writer below is a template)

// rows [0, rows) of a matrix as np.savetxt writes them; value(r, c) yields the double to print, same_as_prev(r) whether row r
// repeats row r - 1 bit for bit (its text is then copied, not formatted again)
template <class Value, class SameAsPrev>
static int write_text_rows(const char* path, int64_t rows, int64_t cols, const char* fmt, bool append, Value value, SameAsPrev same_as_prev) {
    if (!float_format_ok(fmt)) return fail(-1, "unsupported format '%s'", fmt);
    const ptnn_text::Format f = ptnn_text::parse_format(fmt);
    FILE* fp = std::fopen(path, append ? "a" : "w");
    if (!fp) return fail(-4, "cannot open %s for writing", path);
    std::setvbuf(fp, nullptr, _IONBF, 0);                      // the block below is the buffer
    const size_t line_cap = (size_t)cols * 401 + 2;
    // no larger than the file can get, and not value-initialised: most of a run's files are a few KB
    const size_t buf_size = std::max<size_t>(std::min<size_t>(4u << 20, (size_t)std::max<int64_t>(rows, 1) * line_cap), 2 * line_cap);
    const std::unique_ptr<char[]> buf_mem(new char[buf_size]), line_mem(new char[line_cap]);
    struct Span { char* p; size_t n; char* data() const { return p; } size_t size() const { return n; } };
    const Span buf{buf_mem.get(), buf_size}, line{line_mem.get(), line_cap};
    size_t used = 0, line_len = 0;
    for (int64_t r = 0; r < rows; ++r) {
        if (r == 0 || !same_as_prev(r)) {
            char* o = line.data();
            for (int64_t c = 0; c < cols; ++c) {
                if (c) *o++ = ' ';
                o = ptnn_text::put_value(o, value(r, c), f);
            }
            *o++ = '\n';
            line_len = (size_t)(o - line.data());
        }
        if (buf.size() - used < line_len) {
            if (std::fwrite(buf.data(), 1, used, fp) != used) { std::fclose(fp); return fail(-4, "write to %s failed", path); }
            used = 0;
        }
        std::memcpy(buf.data() + used, line.data(), line_len);
        used += line_len;
    }
    const bool wrote = std::fwrite(buf.data(), 1, used, fp) == used;
    if (std::fclose(fp) != 0 || !wrote) return fail(-4, "write to %s failed", path);
    return 0;
}

extern "C" {

int ptnn_savetxt(const char* path, const double* data, int64_t rows, int64_t cols, const char* fmt) {
    if (!path || !data || !fmt) return fail(-1, "null argument");
    if (rows < 0 || cols < 1) return fail(-1, "bad shape %lld x %lld", (long long)rows, (long long)cols);
    return write_text_rows(path, rows, cols, fmt, false, [&](int64_t r, int64_t c) { return data[r * cols + c]; },
                           [&](int64_t r) { return std::memcmp(data + r * cols, data + (r - 1) * cols, (size_t)cols * sizeof(double)) == 0; });
}

int ptnn_savetxt_f32(const char* path, const float* data, int64_t rows, int64_t cols, int64_t row_stride, const char* fmt, int append) {
    if (!path || !data || !fmt) return fail(-1, "null argument");
    if (rows < 0 || cols < 1 || row_stride < cols) return fail(-1, "bad shape %lld x %lld (row stride %lld)", (long long)rows, (long long)cols, (long long)row_stride);
    return write_text_rows(path, rows, cols, fmt, append != 0, [&](int64_t r, int64_t c) { return (double)data[r * row_stride + c]; },
                           [&](int64_t r) { return std::memcmp(data + r * row_stride, data + (r - 1) * row_stride, (size_t)cols * sizeof(float)) == 0; });
}

int ptnn_savetxt_f32_batch(int n_files, const char* const* paths, const float* const* data, const int64_t* rows, const int64_t* cols,
                           const int64_t* row_stride, const char* const* fmts, int append, int threads) {
    if (n_files < 0 || (n_files && (!paths || !data || !rows || !cols || !row_stride || !fmts))) return fail(-1, "null argument");
    const int T = std::max(1, std::min(threads, n_files));
    std::atomic<int> next{0}, bad{-1};
    std::mutex mu;
    std::string why;
    auto work = [&]() {
        for (int k = next.fetch_add(1); k < n_files; k = next.fetch_add(1)) {
            if (ptnn_savetxt_f32(paths[k], data[k], rows[k], cols[k], row_stride[k], fmts[k], append) < 0) {
                std::lock_guard<std::mutex> lock(mu);
                if (bad.load() < 0) { bad.store(k); why = g_err; }     // g_err is per thread: carry the first cause to the caller's
            }
        }
    };
    if (T == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work);
        for (auto& x : th) x.join();
    }
    if (bad.load() >= 0) return fail(-4, "%s", why.c_str());
    return 0;
}

int ptnn_posterior_matrix(const float* pos_w, int64_t n_chains, int64_t n_rows, int64_t n_param, int64_t row_floats, int64_t first_row, double* out, int threads) {
    // out[p][c * m + t] = pos_w[c][first_row + t][p], m = n_rows - first_row: the (P, R (S - b)) float64 matrix show_results
    // returns (REG:795-797, 848: np.loadtxt of every chain's pos_w file, burn-in cut, chains side by side, transposed)
    if (!pos_w || !out || n_chains < 1 || n_param < 1 || row_floats < n_param || first_row < 0 || first_row > n_rows) return fail(-1, "bad argument");
    const int64_t m = n_rows - first_row;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_chains));
    auto work = [&](int t) {
        for (int64_t c = t; c < n_chains; c += T) {
            const float* src = pos_w + (c * n_rows + first_row) * row_floats;
            // blocks of rows: the block's source (bt x P floats) stays in cache while it is read P times with stride P
            for (int64_t t0 = 0; t0 < m; t0 += 256) {
                const int64_t bt = std::min<int64_t>(256, m - t0);
                for (int64_t p = 0; p < n_param; ++p) {
                    double* dst = out + p * (n_chains * m) + c * m + t0;
                    const float* s = src + t0 * row_floats + p;
                    for (int64_t k = 0; k < bt; ++k) dst[k] = (double)s[k * row_floats];
                }
            }
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    return 0;
}

}  // extern "C"
