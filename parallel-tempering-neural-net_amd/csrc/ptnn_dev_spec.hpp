// ptnn_dev_spec.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// speculative schedules: slots, {tag, value} granules (agent scope / through an XCD's L2), the swap cascade, PersistParams, segment_spec_body.

// ------------------------------------------------------------------------------------------------
// Speculative schedule ("prefetching" Metropolis-Hastings).  The MH acceptance rate of these chains is low (1-15 %),
// and every random draw is a pure function of (seed, replica, step), so wave v of the work-group computes step i+v
// under the assumption that steps i .. i+v-1 are rejected -- the chain state (w, eta, likelihood, prior) they all start
// from is then the same.  After one round the prefix up to and including the first accepted step is committed and the
// rest is thrown away: the committed chain is exactly the sequential one, only the wall time per committed step drops
// by (1 - (1-a)^k) / a for acceptance rate a and k waves.  Each wave runs its whole step alone (wave-local LDS
// scratch, no work-group barrier inside a step), so the result does not depend on the number of waves.
// ------------------------------------------------------------------------------------------------
enum { SL_ACCEPT = 0, SL_LIKPROP, SL_PRIORPROP, SL_ETAPRO, SL_RM_TR, SL_RM_TE, SL_AC_TR, SL_AC_TE, SL_LG, SL_ADAPT, SL_LOGALPHA, SL_D2, SL_COUNT = 16 };
constexpr int MAX_SLOTS = 64;          // speculative steps per round: work-groups per replica x waves per work-group
constexpr unsigned SPIN_LIMIT = 1u << 22;   // x (s_sleep 2 + one L2 round trip) = a few seconds, then the launch gives up

__host__ __device__ inline size_t spec_wave_floats(int PS, int H, int FWS) { return 3 * (size_t)PS + fw_floats(H, FWS) + 8; }
__host__ __device__ inline size_t spec_lds_floats(int Nall, int IPY, int PS, int H, int FWS, int NW, int G) {
    return (size_t)(Nall + 2) * IPY + 3 * (size_t)PS + MAX_WAVES * 8 + 32 + (size_t)NW * G * SL_COUNT + (size_t)NW * spec_wave_floats(PS, H, FWS);
}

// 8-byte {tag, value} granule written by ONE agent-scope relaxed atomic store (sc1, write-through) and polled with
// agent-scope relaxed atomic loads (sc1, L1 bypass): the data is its own flag, no fence on either side
// (cdna_hip_programming.md Guideline 16, form R2).  Every spin is bounded.
typedef unsigned long long granule_t;
__device__ __forceinline__ void granule_store(granule_t* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((granule_t)epoch << 32) | (granule_t)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool granule_wait(const granule_t* g, unsigned epoch, float& v) {
    for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        const granule_t x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == epoch) { v = __builtin_bit_cast(float, (unsigned)x); return true; }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}
// The same granule between work-groups that are KNOWN to sit on one XCD (they share its L2): a plain store (into that L2, not
// written through to memory like the agent-scope store) and a non-temporal load (`nt`: not served from the reader's L1 like a plain or
// sc0 load).  Measured (profiles/tools/micro/granule_pingpong.hip, profiles/r04_granule_pingpong.txt): one way 241 ns instead of
// 508 - 588 ns, and no fabric traffic (the agent-scope pair costs ~32 B written + ~64 B fetched per message); between two XCDs such a
// store never arrives.
// So: only after the work-groups have compared their XCC ids through the agent-scope path (xcc_id below; the tree does it in the
// first round of every launch).
__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}
__device__ __forceinline__ void granule_store_xcd(granule_t* g, unsigned epoch, float v) {
    const granule_t x = ((granule_t)epoch << 32) | (granule_t)__builtin_bit_cast(unsigned, v);
    // a plain store: write-through the CU's L1 into the XCD's L2, where the line stays (write-back, ordinary replacement) and is
    // rewritten two rounds later.  With `nt` on the STORE the line is marked evict-first: under the stream of trace rows every
    // granule went out to memory on its own (Iris, one launch per run: 131 MB written per run against 77 MB of trace rows; the
    // agent-scope path 118 MB) -- profiles/README.md, r04d vs r04e.  The polling LOAD keeps `nt` (it must not be served from L1).
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g), "v"(x) : "memory");
}
__device__ __forceinline__ bool granule_wait_xcd(const granule_t* g, unsigned epoch, float& v, unsigned limit = SPIN_LIMIT) {
    for (unsigned spins = 0; spins < limit; ++spins) {
        granule_t x;
        asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(g) : "memory");
        if ((unsigned)(x >> 32) == epoch) { v = __builtin_bit_cast(float, (unsigned)x); return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
// Do the G work-groups of a replica share an XCD?  Asked once per launch, before the first round: every group stores its XCC id
// with an AGENT-scope store (write-through: it reaches memory, and it is in the writer's L2) into its own granule of `ids`, and
// reads the others' with `nt` loads.  On one XCD those loads hit the common L2: everybody sees G equal ids -- without a byte of
// fabric traffic for the polling (agent-scope polling of the same granules was most of what the tree still fetched per launch).
// A group on ANOTHER XCD reads a different id, or a line its own L2 fetched too early and keeps (then the bounded wait runs out):
// either way it answers no, and so does everybody who waits for it -- the verdict is the same in every group, because "all G ids
// equal mine, seen through L2" can only hold for all of them at once.  `tmp`: G floats of LDS; ends with a work-group barrier.
constexpr unsigned XCD_HANDSHAKE_SPINS = 1u << 13;          // about a millisecond: the groups of one launch start microseconds apart
__device__ __forceinline__ bool xcd_handshake_strided(granule_t* ids, int stride, int G, int grp, unsigned tag, float* tmp) {
    const int my_xcc = xcc_id();
    if (threadIdx.x == 0) granule_store(ids + (size_t)grp * stride, tag, (float)my_xcc);
    if ((int)threadIdx.x < G) {
        float v = -1.0f;
        if (!granule_wait_xcd(ids + (size_t)threadIdx.x * stride, tag, v, XCD_HANDSHAKE_SPINS)) v = -2.0f;
        tmp[threadIdx.x] = v;
    }
    __syncthreads();
    bool same = true;
    for (int g_ = 0; g_ < G; ++g_) same = same && (tmp[g_] == (float)my_xcc);
    __syncthreads();
    return same;
}
__device__ __forceinline__ bool xcd_handshake(granule_t* ids, int G, int grp, unsigned tag, float* tmp) {
    return xcd_handshake_strided(ids, 1, G, grp, tag, tmp);
}


// Work-groups are dispatched round-robin to the 8 XCDs (each with its own L2).  The work-groups of ONE replica exchange records
// every round, so each XCD is given a contiguous range of logical blocks: a replica's groups then share an L2 (when the grid is a
// multiple of 8 and the replicas per XCD come out whole; otherwise the plain order).
__device__ __forceinline__ int xcd_block(int G) {
    const int b = blockIdx.x, nb = gridDim.x;
#if defined(PTNN_NO_XCD_MAP)
    return b;
#else
    return (G > 1 && (nb & 7) == 0 && ((nb >> 3) % G) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;   // G == 1: block = replica, as the one-group bodies take it
#endif
}

// ------------------------------------------------------------------------------------------------
// R12 swap cascade (REG:659-690, 741-748): one sequential bubble pass over the ladder.  Every block recomputes it
// (R <= a few thousand scalars), then block b moves the (w, eta) row for local replica b.
// ------------------------------------------------------------------------------------------------
struct SwapParams {
    int R, Rl, first_global, PS;
    uint32_t seed_lo, seed_hi;
    const float* L;            // [R] posted scalars
    const float* cur;          // [Rl][PS]
    float* next;               // [Rl][PS]
    const float* gd_cur;       // [Rl][PS] cached langevin_gradient(w) rows, travel with w inside one GPU
    float* gd_next;            // [Rl][PS]
    const int* gd_valid_cur;   // [Rl]
    int* gd_valid_next;        // [Rl]
    int* src_out;              // [R] (may be null)
    long long* counters;       // [0] num_swap, [1] total_swap_proposals
    int* src_log;              // [max_rounds][R] (may be null)
    int log_capacity;
    // swap_rule 1: even/odd Metropolis exchange exp((1/T_k - 1/T_k+1)(L_k+1 - L_k)) on untempered log-likelihoods; the
    // moved state brings its likelihood and prior along (no stale values), no phantom round
    int rule, canonical;       // canonical: the chains are past the temperature switch (likelihoods untempered)
    const float* L_raw;        // [R]
    const float* prior_post;   // [R]
    const float* temps_global; // [R]
    float* st_f;               // [Rl][SF_COUNT]
    // gathered exchange (ladder sharded over GPUs): every rank holds, after one all-gather, the exchange rows of ALL replicas
    // xchg[R][XS] = { state row (PS) | cached-gradient row (PS) | gradient valid | posted L | pad }
    float* xchg;               // null: single-GPU / point-to-point modes
    int XS;
    int L_stride;              // 1, or XS when L (and, swap_rule 1, L_raw / prior_post) are read from the exchange rows
    // label swapping (not in the reference): the chains stay where they are and the TEMPERATURES move.  label[slot] =
    // temperature index the chain in that slot holds, slot_of[temperature] = its inverse, both over the whole ladder and replicated
    // on every rank; the cascade runs over temperature indices, the round only rewrites the maps, the temperature of the local
    // slots and (before the temperature switch) the tempering of their likelihoods.  Nothing but L crosses a GPU boundary.
    int label_mode;
    const int* label_cur;      // [R]
    const int* slot_cur;       // [R]
    int* label_next;           // [R]
    int* slot_next;            // [R]
    float* temps_local;        // [Rl] temperature of the local slots (what the segment kernels read)
    int* progress;             // pinned host word (or null): block 0 stores round + 1 when the round is through -- what the bounded
                               // waits of a communicator handle watch (ptnn.hip: wait_stream)
};
__host__ __device__ inline int xchg_row_floats(int PS) { return (2 * PS + 4 + 3) & ~3; }

// sSrc has R + 1 ints: the last one carries the number of accepted swaps
// have_L: sL already holds the R posted scalars (the tree's in-launch swap round reads them from granules)
__device__ __forceinline__ int cascade_lds(const SwapParams& sp, int round, float* sL, float* sU, int* sSrc, bool have_L = false) {
    const int R = sp.R;
    for (int k = threadIdx.x; k < R; k += blockDim.x) {
        if (!have_L) sL[k] = sp.L[(size_t)(sp.label_mode ? sp.slot_cur[k] : k) * sp.L_stride];    // k is a temperature index
        if (k < R - 1) {
            uint32_t x[4];
            philox4x32_10((uint32_t)k, (uint32_t)round, 0u, STREAM_SWAP, sp.seed_lo, sp.seed_hi, x);
            // rule 0 compares in the log domain (below): ln(2 u) is computed here, by all threads at once, instead of an exp inside
            // the sequential chain
            sU[k] = (sp.rule == 1) ? u23(x[0]) : logf_fast(2.0f * u23(x[0]));
        }
    }
    __syncthreads();
    if (sp.rule == 1) {
        // independent pairs (k, k+1), k of the round's parity
        for (int k = threadIdx.x; k < R; k += blockDim.x) sSrc[k] = k;
        if (threadIdx.x == 0) sSrc[R] = 0;
        __syncthreads();
        for (int k = (round & 1) + 2 * threadIdx.x; k < R - 1; k += 2 * blockDim.x) {
            const int s0 = sp.label_mode ? sp.slot_cur[k] : k, s1 = sp.label_mode ? sp.slot_cur[k + 1] : k + 1;
            const float d = (1.0f / sp.temps_global[k] - 1.0f / sp.temps_global[k + 1]) *
                            (sp.L_raw[(size_t)s1 * sp.L_stride] - sp.L_raw[(size_t)s0 * sp.L_stride]);
            const float pr = (d != d) ? 1.0f : fminf(1.0f, expf_fast(fminf(d, 80.0f)));
            if (sU[k] < pr) { sSrc[k] = k + 1; sSrc[k + 1] = k; atomicAdd(&sSrc[R], 1); }
        }
        __syncthreads();
        return sSrc[R];
    }
    // REG:674-679: swap iff u < min(1, 0.5 exp(min(709, L[k+1] - L[c]))).  u < 1 always, so the outer min never binds, and with
    // u > 0 the test is ln(2 u) < min(709, L[k+1] - L[c]) -- subtract, clamp, compare, no transcendental.
    //
    // The bubble pass is sequential only through WHICH state is being carried: while the carried state is c, the tests of the
    // pairs ahead are all against the same L[c], i.e. independent.  Wave 0 takes the pairs 64 at a time (lane = pair), tests
    // all of them against the current carried L with one compare, and a ballot finds the first pair where it fails: the
    // carried state is dropped there (src[k] = c), the next state is picked up (its L comes from that lane's register) and
    // the lanes behind it are re-tested -- one iteration per DROP, not per pair, plus one per 64 pairs.  Wave-uniform control
    // throughout.  (Round 1 walked the pairs one by one in thread 0 of every block, with an exp, a branch and an LDS store per
    // pair: 95 ns per pair -- 6.1 us per round at R = 64, 40 us at 256, 72 us at 1024; independent forward scans from every
    // start + pointer doubling were tried and are worse, because the reference's rule accepts 60 - 98 % of the swaps and the runs
    // are long.)
    if (threadIdx.x < WAVE) {
        const int lane = threadIdx.x;
        int c = 0, nsw = 0;
        float Lc = sL[0];
        for (int k0 = 0; k0 < R - 1; k0 += WAVE) {
            const int k = k0 + lane;
            const bool valid = k < R - 1;
            const float Ln = valid ? sL[k + 1] : 0.0f;
            const float tk = valid ? sU[k] : 0.0f;
            unsigned long long todo = __ballot(valid);
            unsigned long long swapped = 0ull;
            while (todo) {
                float d = Ln - Lc;
                d = (d < 709.0f) ? d : 709.0f;              // python min(709, nan) == 709
                const unsigned long long fail = __ballot(!(tk < d)) & todo;
                if (!fail) { swapped |= todo; break; }       // the carried state passes every remaining pair of this window
                const int j = __ffsll((long long)fail) - 1;  // first pair where it is dropped
                swapped |= todo & ((1ull << j) - 1ull);
                if (lane == j) sSrc[k] = c;                  // slot k0 + j receives the carried state ...
                c = k0 + j + 1;                              // ... and the state of the next slot is picked up
                Lc = __shfl(Ln, j);
                todo &= (j == 63) ? 0ull : ~((2ull << j) - 1ull);
            }
            if (valid && ((swapped >> lane) & 1ull)) sSrc[k] = k + 1;
            nsw += __popcll(swapped);
        }
        if (lane == 0) { sSrc[R - 1] = c; sSrc[R] = nsw; }
    }
    __syncthreads();
    return sSrc[R];
}

// What a launch that spans several swap intervals needs to know (persistent_loop at the end of this file; the tree body runs its
// own swap rounds and reads it too)
struct PersistParams {
    int end;                 // MH steps are run up to here (exclusive)
    int swap_inside;         // 1: the swap rounds between the intervals run inside this launch
    int task, si;            // hand-off rule (Q10): REG after step i when i % si == 0 and i != 0; CLS when (i + 1) % si == 0
    int round0;              // index of the first swap round of this launch
    int flip0, lflip0;       // which state / label-map buffers are current at entry
    int nblocks;             // work-groups of the grid
    unsigned* barrier;       // [nblocks] phase every work-group has reached, zero at launch
    float* state[2];
    float* gd[2];
    int* gd_valid[2];
    int* label[2];
    int* slot_of[2];
    SwapParams sp;           // everything of a round that does not flip
};

// The PersistParams of the launch, read from the kernel-argument segment where it lies (second argument, behind SegParams) through
// a pointer the optimiser cannot see through: every use re-loads the few words it needs (scalar loads from the constant cache)
// instead of keeping ~60 words of it live across the interval body -- hoisted out of the loop they were spilled into vector
// registers and, in the two kernels closest to the register ceiling, on into scratch.
__device__ __forceinline__ persist_cptr persist_args() {
    constexpr size_t off = (sizeof(SegParams) + alignof(PersistParams) - 1) & ~(alignof(PersistParams) - 1);
    unsigned long long a = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + off;
    asm volatile("" : "+s"(a));
    return (persist_cptr)(uintptr_t)a;
}

typedef __attribute__((address_space(4))) const SegParams* seg_cptr;
__device__ __forceinline__ seg_cptr seg_args() {
    unsigned long long a = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return (seg_cptr)(uintptr_t)a;
}

// Granules of a swap round that a launch with several work-groups per replica runs by itself (tree, packed multi-CU), per parity:
// R posted scalars (padded to 8), R rows of `row` floats for the other replicas' root groups (the state; with a cached gradient:
// state, gradient, its valid flag), R rows {flag, the same row} from every root to its own siblings
__host__ __device__ inline size_t swap_xchg_granules(int R, int row) { return (size_t)((R + 7) & ~7) + (size_t)R * row + (size_t)R * (row + 8); }

// p.G work-groups (one per CU) cooperate on one replica: work-group g, wave v owns speculative slot g*NW + v.
// Every work-group keeps its own LDS copy of the chain state and applies the same commits, so the copies never
// diverge; only the per-slot results (and the accepted proposal) cross CUs.
template <int TASK, int I, int O>
__device__ __forceinline__ void segment_spec_body(const SegParams& p, const SegDyn& dyn, const int step_begin, const int n_steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    PTNN_DIAG(spec_entry);
    const int G = p.G;
    const int lb = xcd_block(G);
    const int r = lb / G, grp = lb - r * G;
    const int gid = p.first_global + r;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63, NW = nthr >> 6;
    const int K = NW * G;                                  // speculative slots per round
    const int sidx = grp * NW + wave;                      // my slot
    const int Nall = p.Ntr + p.Nte;
    const int P = p.P, PS = p.PS, H = p.H;
    // shared part
    float* q = smem;
    float* xy = q; q += (Nall + 2) * p.IPY;
    float* w_cur = q; q += PS;
    float* w_gd = q; q += PS;
    float* rec_w = q; q += PS;
    float* red = q; q += MAX_WAVES * 8;
    // forward passes of Langevin slots are taken over by waves of the work-group that have nothing left to do (below): per wave
    // {proposal ready, forward claimed, forward done}, each the tag of the round it refers to
    unsigned* pready = reinterpret_cast<unsigned*>(q); unsigned* fclaim = pready + MAX_WAVES; unsigned* fdone = fclaim + MAX_WAVES; q += 32;
    float* slots = q; q += K * SL_COUNT;
    // private part of every wave: proposal, its SGD epoch, noise, packed forward image, scalars
    const size_t wfl = spec_wave_floats(PS, H, p.FWS);
    float* priv0 = q;
    float* mine = priv0 + (size_t)wave * wfl;
    float* my_prop = mine;
    float* my_pgd = mine + PS;
    float* my_noise = mine + 2 * PS;
    float* my_fw = mine + 3 * PS;
    float* my_scal = my_fw + fw_floats(H, p.FWS);
    // exchange areas of this replica (G > 1): [parity][slot][16] result granules, [parity][slot][2 PS] proposal granules
    granule_t* xs = p.xslots + (size_t)r * 2 * MAX_SLOTS * SL_COUNT;
    granule_t* xw = p.xw + (size_t)r * 2 * MAX_SLOTS * 2 * PS;
    granule_t* xv = p.xverdict + (size_t)r * 2 * MAX_SLOTS;

    {
        const float4* src = reinterpret_cast<const float4*>(p.data);
        float4* dst = reinterpret_cast<float4*>(xy);
        for (int e = tid; e < ((Nall + 2) * p.IPY) >> 2; e += nthr) dst[e] = src[e];
    }
    float* gw = dyn.w_state + (size_t)r * PS;
    for (int j = tid; j < PS; j += nthr) {
        w_cur[j] = gw[j];
        rec_w[j] = p.rec_w[(size_t)r * PS + j];
        w_gd[j] = dyn.gd_w[(size_t)r * PS + j];
    }
    if (tid < 3 * MAX_WAVES) pready[tid] = 0u;                 // tag 0 is never a round's tag
    __syncthreads();

    const float T = uni_f(p.temps[r]);
    float eta = (TASK == TASK_REG) ? uni_f(w_cur[P]) : 0.0f;
    float* sf = p.st_f + (size_t)r * SF_COUNT;
    int* si = p.st_i + (size_t)r * SI_COUNT;
    float lik, prior_cur, tau_eta_last, rec_rmse_tr, rec_rmse_te, rec_acc_tr, rec_acc_te;
    int nacc, gd_valid, lg_count, lg_acc;
    if (step_begin == 0) {
        lg_acc = 0;
        if (wave == 0) {                                       // one wave alone: independent of wave and group count
            chain_startup<TASK, I, O, true>(p, xy, w_cur, my_fw, red, T, eta, lik, prior_cur);
            if (lane == 0) { red[0] = eta; red[1] = lik; red[2] = prior_cur; }
        }
        __syncthreads();
        eta = uni_f(red[0]); lik = uni_f(red[1]); prior_cur = uni_f(red[2]);
        tau_eta_last = eta;
        rec_rmse_tr = rec_rmse_te = rec_acc_tr = rec_acc_te = 0.f;
        nacc = 0; gd_valid = 0; lg_count = 0;
        __syncthreads();
    } else {
        lik = uni_f(sf[SF_LIK]); prior_cur = uni_f(sf[SF_PRIOR]); tau_eta_last = uni_f(sf[SF_TAU_LAST]);
        rec_rmse_tr = uni_f(sf[SF_REC_RMSE_TR]); rec_rmse_te = uni_f(sf[SF_REC_RMSE_TE]);
        rec_acc_tr = uni_f(sf[SF_REC_ACC_TR]); rec_acc_te = uni_f(sf[SF_REC_ACC_TE]);
        nacc = uni_i(si[SI_NACC]); gd_valid = uni_i(dyn.gd_valid[r]); lg_count = uni_i(si[SI_LG_COUNT]); lg_acc = uni_i(si[SI_LG_ACC]);
    }

    PTNN_DIAG(spec_begin);
    const size_t trow = (size_t)r * p.trace_cap;        // traces are rings of trace_cap rows per replica (== S unless streaming)
    const int end = step_begin + n_steps;
    int i = step_begin;
    unsigned epoch = dyn.epoch_base;
    int par = 0;
    bool failed = false;
    STAMP(0);                                             // launch prologue: staging, start-up
    while (i < end) {
        epoch += 1;
        PTNN_DIAG(count_round);
        if (i == p.switch_step) {
            // R10 (REG:320-324): canonical from here on; re-evaluate the current w with the LAST PROPOSED tau (Q9)
            if (wave == 0) {
                build_fw<I, O, true>(w_cur, my_fw, H, p.FWS);
                gsync<true>();
                const EvalSums sc = eval_rows<TASK, I, O, true>(my_fw, xy, p.IPY, p.FWS, H, p.Ntr, Nall, nullptr);
                float l2, r1, r2, a1, a2;
                finish_eval<TASK>(sc, p.Ntr, p.Nte, tau_eta_last, l2, r1, r2, a1, a2);
                if (lane == 0) red[0] = l2;
            }
            __syncthreads();
            lik = uni_f(red[0]);
            __syncthreads();
        }
        int k = min(K, end - i);
        if (p.switch_step > i) k = min(k, p.switch_step - i);     // a round never straddles the temperature switch
        const int j = i + sidx;
        const bool active = sidx < k;
        bool lg = false;
        float u = 0.f, n_eta = 0.f;
        if (active) {
            tape_step<true>(p, gid, j, my_noise, my_scal);
            gsync<true>();
            const float lx = my_scal[0];
            u = my_scal[1];
            n_eta = my_scal[2];
            lg = p.use_lg && (lx < p.l_prob);
        }
        STAMP(1);                                         // random tape
        if (p.use_lg && !gd_valid) {
            // w_gd = langevin_gradient(w) is missing (chain start, or w arrived from another GPU): every work-group
            // recomputes it for itself.  Decided from state all groups share, so they all take this branch together.
            if (wave == 0) sgd_sweep_dispatch<TASK, I, O>(w_cur, w_gd, xy, p.data, p.Ntr, H, p.lr);
            gd_valid = 1;
            __syncthreads();
        }
        float* sl = slots + sidx * SL_COUNT;
        bool acc_mine = false;
        STAMP(2);                                         // langevin_gradient(w) recompute (rare)
        // forward pass + likelihood + prior of a proposal of THIS work-group (own or another wave's), on this wave's image scratch
        auto forward_of = [&](const float* prop, float eta_x, float adapt_x, float& lik_prop, float& prior_prop, float& rm_tr,
                              float& rm_te, float& ac_tr, float& ac_te) {
            build_fw<I, O, true>(prop, my_fw, H, p.FWS);
            gsync<true>();
            const EvalSums es = eval_rows<TASK, I, O, true>(my_fw, xy, p.IPY, p.FWS, H, p.Ntr, Nall, nullptr);
            float ll;
            finish_eval<TASK, true>(es, p.Ntr, p.Nte, eta_x, ll, rm_tr, rm_te, ac_tr, ac_te);
            lik_prop = ll / adapt_x;
            const float ssq = block_sumsq<true>(prop, P, nullptr);
            prior_prop = prior_value<TASK>(p, ssq, eta_x);
        };
        if (active) {
            const float adapttemp = (p.switch_step >= 0 && j >= p.switch_step) ? 1.0f : T;
            float diff_prop = 0.0f;
            float eta_pro = eta;
            if (TASK == TASK_REG) eta_pro = fmaf(p.step_eta, n_eta, eta);
            float lik_prop = 0.f, prior_prop = 0.f, rm_tr = 0.f, rm_te = 0.f, ac_tr = 0.f, ac_te = 0.f;
            bool have_forward = false;
            if (lg) {
                for (int e = lane; e < P; e += WAVE) my_prop[e] = fmaf(p.step_w, my_noise[e], w_gd[e]);
                // The forward pass of the proposal does not depend on its SGD epoch: announce the proposal, so that a wave of
                // this work-group with nothing left to do (a random-walk slot, an idle slot at the end of an interval) runs it
                // while this wave sweeps.  Whoever sets the claim word to the round's tag first does the pass.
                if (lane == 0) { sl[SL_ETAPRO] = eta_pro; sl[SL_ADAPT] = adapttemp; }
                gsync<true>();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(pready + wave, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                sgd_sweep_dispatch<TASK, I, O>(my_prop, my_pgd, xy, p.data, p.Ntr, H, p.lr);
                gsync<true>();
                const float d1 = block_sumsq_diff<true>(w_cur, my_pgd, P, nullptr);
                const float d2 = block_sumsq<true>(my_noise, P, nullptr);
                diff_prop = langevin_ratio(d1, d2, p.step_w, adapttemp);
                unsigned prev = 0u;
                if (lane == 0) prev = __hip_atomic_exchange(fclaim + wave, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                prev = (unsigned)uni_i((int)prev);
                if (prev == epoch) {                            // a helper took it: wait for its results (it is far ahead of us)
                    bool done = false;
                    for (unsigned spins = 0; spins < (1u << 20) && !done; ++spins) {
                        done = uni_i((int)__hip_atomic_load(fdone + wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == (int)epoch;
                        if (!done) __builtin_amdgcn_s_sleep(1);
                    }
                    if (done) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        lik_prop = uni_f(sl[SL_LIKPROP]); prior_prop = uni_f(sl[SL_PRIORPROP]);
                        rm_tr = uni_f(sl[SL_RM_TR]); rm_te = uni_f(sl[SL_RM_TE]); ac_tr = uni_f(sl[SL_AC_TR]); ac_te = uni_f(sl[SL_AC_TE]);
                        have_forward = true;
                    }                                           // (never seen: the pass is then simply done here, same values)
                }
            } else {
                for (int e = lane; e < P; e += WAVE) my_prop[e] = fmaf(p.step_w, my_noise[e], w_cur[e]);
                gsync<true>();
            }
            STAMP(3);                                     // proposal (+ SGD epoch for a Langevin slot)
            if (!have_forward) forward_of(my_prop, eta_pro, adapttemp, lik_prop, prior_prop, rm_tr, rm_te, ac_tr, ac_te);
            const float logalpha = (lik_prop - lik) + (prior_prop - prior_cur) + diff_prop;
            const float mh = (logalpha != logalpha) ? 1.0f : fminf(1.0f, expf_fast(logalpha));
            acc_mine = u < mh;
            STAMP(4);                                     // forward pass, likelihood, prior, MH ratio
            if (p.use_lg && !lg && acc_mine) {
                // an accepted random-walk step: run the SGD epoch from its proposal now, so langevin_gradient(new w)
                // is already there when the step is committed (the Langevin waves of this round are sweeping anyway)
                sgd_sweep_dispatch<TASK, I, O>(my_prop, my_pgd, xy, p.data, p.Ntr, H, p.lr);
                gsync<true>();
            }
            if (lane == 0) {
                sl[SL_ACCEPT] = acc_mine ? 1.0f : 0.0f;
                sl[SL_LIKPROP] = lik_prop; sl[SL_PRIORPROP] = prior_prop; sl[SL_ETAPRO] = eta_pro;
                sl[SL_RM_TR] = rm_tr; sl[SL_RM_TE] = rm_te; sl[SL_AC_TR] = ac_tr; sl[SL_AC_TE] = ac_te;
                sl[SL_LG] = lg ? 1.0f : 0.0f; sl[SL_ADAPT] = adapttemp; sl[SL_LOGALPHA] = logalpha;
            }
            if (G > 1 && acc_mine) {
                // an accepted slot publishes its record (one 128-byte wave store), the proposal and its SGD epoch: the other
                // groups read them at commit, and only then
                gsync<true>();
                if (lane < SL_COUNT) granule_store(xs + ((size_t)par * MAX_SLOTS + sidx) * SL_COUNT + lane, epoch, sl[lane]);
                granule_t* xo = xw + ((size_t)par * MAX_SLOTS + sidx) * 2 * PS;
                for (int e = lane; e < 2 * PS; e += WAVE) granule_store(xo + e, epoch, mine[e]);   // my_prop ++ my_pgd
            }
        }
        // Every slot, every round: ONE 8-byte verdict granule {tag, accepted?}, published the moment the slot is decided; the
        // verdicts of a replica's round are one contiguous row, which is all a foreign group polls.  Everything else it needs of a
        // rejected foreign slot -- the Langevin coin, eta_pro -- follows from the tape and the shared chain state; records and
        // proposals are published by accepted slots only and read at commit.  rocprofv3, Mackey-Glass 64 replicas x 4 groups, HBM
        // bytes per launch: 4.2 MB (1.8 MB of it trace rows); 7.6 MB when every slot published a 16-granule record every round
        // and every group polled all of them.  (One granule per GROUP, carrying the accept bits of its slots and published after
        // the group's barrier, moves 6 % fewer bytes and was 4 % slower: the other groups see a decision later.)
        // A wave without a step this round (k < K: the last rounds of an interval, or before the temperature switch) publishes
        // too: every group waits for EVERY slot's tag below, which keeps the groups within one round of each other, so two-deep
        // buffers suffice.
        if (G > 1 && lane == 0) granule_store(xv + (size_t)par * MAX_SLOTS + sidx, epoch, (active && acc_mine) ? 1.0f : 0.0f);
        STAMP(5);                                         // publish
        // With its own step decided (or none to do), a wave takes over forward passes of Langevin slots of its work-group that are
        // still sweeping: same code on the same proposal, so the values are those the owner would compute.
        if (p.use_lg) {
            for (int t_ = 0; t_ < NW; ++t_) {
                if (t_ == wave) continue;
                const bool ready = uni_i((int)__hip_atomic_load(pready + t_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == (int)epoch;
                if (!ready) continue;
                unsigned prev = epoch;
                if (lane == 0) prev = __hip_atomic_exchange(fclaim + t_, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                prev = (unsigned)uni_i((int)prev);
                if (prev == epoch) continue;                    // its owner or another helper has it
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float* slt = slots + (grp * NW + t_) * SL_COUNT;
                float lp, pp, r1, r2, a1, a2;
                forward_of(priv0 + (size_t)t_ * wfl, uni_f(slt[SL_ETAPRO]), uni_f(slt[SL_ADAPT]), lp, pp, r1, r2, a1, a2);
                if (lane == 0) {
                    slt[SL_LIKPROP] = lp; slt[SL_PRIORPROP] = pp; slt[SL_RM_TR] = r1; slt[SL_RM_TE] = r2; slt[SL_AC_TR] = a1; slt[SL_AC_TE] = a2;
                }
                gsync<true>();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(fdone + t_, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        STAMP(6);                                         // waiting for the slowest wave of this work-group
        if (G > 1) {
            // verdicts of the other work-groups' slots: one thread per slot (K <= 64: the lanes of wave 0 read one row)
            bool ok = true;
            if (tid < K && tid / NW != grp) {
                const int s_ = tid;
                float v = 0.0f;
                ok = granule_wait(xv + (size_t)par * MAX_SLOTS + s_, epoch, v);
                if (s_ < k) {
                    uint32_t x[4];
                    philox4x32_10(0u, (uint32_t)(i + s_), p.noise_shared ? 0u : (uint32_t)gid, STREAM_STEP, p.seed_lo, p.seed_hi, x);
                    float n2, n3;
                    box_muller(x[2], x[3], n2, n3);          // the same scalars tape_step hands the slot's owner
                    float* fs = slots + s_ * SL_COUNT;
                    fs[SL_ACCEPT] = v;
                    fs[SL_LG] = (p.use_lg && u23(x[0]) < p.l_prob) ? 1.0f : 0.0f;
                    fs[SL_ETAPRO] = (TASK == TASK_REG) ? fmaf(p.step_eta, n2, eta) : eta;
                }
            }
            if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
        }
        STAMP(7);                                         // gather from the other work-groups (incl. waiting for them)
        // commit the prefix up to and including the first accepted step
        const bool my_flag_acc = (lane < k) && (slots[lane * SL_COUNT + SL_ACCEPT] != 0.0f);
        const bool my_flag_lg = (lane < k) && (slots[lane * SL_COUNT + SL_LG] != 0.0f);
        const unsigned long long bal_acc = __ballot(my_flag_acc), bal_lg = __ballot(my_flag_lg);
        const int m = bal_acc ? (__ffsll((long long)bal_acc) - 1) : k;
        const int ncommit = (m < k) ? m + 1 : k;
        if (sidx < ncommit) {
            const bool acc_me = (sidx == m);
            const float* srcw = acc_me ? my_prop : rec_w;
            const size_t tpos = trow + (size_t)((j + 1) % p.trace_cap);
            float* prow = p.tr_pos_w + tpos * (size_t)p.PW;
            for (int e = lane; e < p.PW; e += WAVE) prow[e] = (e < P) ? srcw[e] : 0.0f;
            if (lane == 0) {
                store_trace_row(p.tr_scal + tpos * TR_COUNT, (TASK == TASK_REG) ? sl[SL_LIKPROP] : sl[SL_LIKPROP] * sl[SL_ADAPT],
                                acc_me ? sl[SL_RM_TR] : rec_rmse_tr, acc_me ? sl[SL_RM_TE] : rec_rmse_te,
                                acc_me ? sl[SL_AC_TR] : rec_acc_tr, acc_me ? sl[SL_AC_TE] : rec_acc_te,
                                nacc /* count BEFORE this step (REG:380) */, sl[SL_LOGALPHA]);
            }
        }
        lg_count += __popcll(bal_lg & ((ncommit >= 64) ? ~0ull : ((1ull << ncommit) - 1ull)));
        if (TASK == TASK_REG) tau_eta_last = uni_f(slots[(ncommit - 1) * SL_COUNT + SL_ETAPRO]);
        __syncthreads();                                    // every reader of rec_w is done
        if (m < k) {
            const int owner = m / NW;
            bool ok = true;
            if (G > 1 && owner != grp) {                      // the accepted slot ran elsewhere: fetch its record
                if (tid < SL_COUNT) {
                    float v = 0.0f;
                    ok = granule_wait(xs + ((size_t)par * MAX_SLOTS + m) * SL_COUNT + tid, epoch, v);
                    slots[m * SL_COUNT + tid] = v;
                }
                if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
            }
            const float* sm = slots + m * SL_COUNT;
            nacc += 1;
            lik = uni_f(sm[SL_LIKPROP]); prior_cur = uni_f(sm[SL_PRIORPROP]); eta = uni_f(sm[SL_ETAPRO]);
            rec_rmse_tr = uni_f(sm[SL_RM_TR]); rec_rmse_te = uni_f(sm[SL_RM_TE]);
            rec_acc_tr = uni_f(sm[SL_AC_TR]); rec_acc_te = uni_f(sm[SL_AC_TE]);
            gd_valid = p.use_lg ? 1 : 0;
            lg_acc += (sm[SL_LG] != 0.0f) ? 1 : 0;
            if (owner == grp) {
                const float* wacc = priv0 + (size_t)(m - grp * NW) * wfl;
                for (int e = tid; e < P; e += nthr) {
                    const float v = wacc[e];
                    w_cur[e] = v; rec_w[e] = v;
                    if (p.use_lg) w_gd[e] = wacc[PS + e];
                }
            } else {
                const granule_t* xo = xw + ((size_t)par * MAX_SLOTS + m) * 2 * PS;
                for (int e = tid; e < P; e += nthr) {
                    float v = 0.f, gv = 0.f;
                    ok = granule_wait(xo + e, epoch, v) && ok;
                    if (p.use_lg) ok = granule_wait(xo + PS + e, epoch, gv) && ok;
                    w_cur[e] = v; rec_w[e] = v;
                    if (p.use_lg) w_gd[e] = gv;
                }
            }
            if (G > 1 && __syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
        }
        __syncthreads();
        i += ncommit;
        par ^= 1;
        STAMP(8);                                         // commit: trace rows, state update
    }
    PTNN_DIAG(spec_flush);

    if (failed) {
        if (tid == 0) atomicAdd(p.error_flag, 1);           // a bounded spin ran out: the host reports it
        return;
    }
    if (grp == 0) {
        for (int j = tid; j < PS; j += nthr) {
            gw[j] = (j == P) ? eta : w_cur[j];
            p.rec_w[(size_t)r * PS + j] = rec_w[j];
            dyn.gd_w[(size_t)r * PS + j] = w_gd[j];
        }
        if (tid == 0) {
            sf[SF_LIK] = lik; sf[SF_PRIOR] = prior_cur; sf[SF_TAU_LAST] = tau_eta_last;
            sf[SF_REC_RMSE_TR] = rec_rmse_tr; sf[SF_REC_RMSE_TE] = rec_rmse_te;
            sf[SF_REC_ACC_TR] = rec_acc_tr; sf[SF_REC_ACC_TE] = rec_acc_te;
            si[SI_NACC] = nacc; dyn.gd_valid[r] = gd_valid; si[SI_LG_COUNT] = lg_count; si[SI_LG_ACC] = lg_acc;
            p.L_handoff[gid] = (TASK == TASK_REG) ? lik * T : lik;
            p.L_final[gid] = lik;
            post_raw(p, gid, lik, prior_cur, T, step_begin + n_steps - 1);
        }
    }
}
