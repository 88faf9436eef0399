// ptnn_dev_pack.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// packed speculative schedule on one CU and over several (segment_pack_body), with its own swap rounds inside a launch.

// ------------------------------------------------------------------------------------------------
// Packed speculative schedule (H <= 8): the whole round of PK_SLOTS speculative steps lives on ONE CU.  A net with <= 8
// hidden units uses 8 lanes of a wave, so waves 0 and 1 run the SGD epochs of all 16 slots at once (8 lane groups each,
// see sgd_sweep) -- for every slot, random-walk ones included, so an accepted step always brings langevin_gradient(new w)
// with it -- while waves 2 and 3 run the 16 forward passes in the meantime; no work-group of another CU is involved, hence
// no cross-CU exchange on the critical path.  Per-slot arithmetic is the same code as in segment_spec_kernel, so the
// committed chain is bit-identical to it (tested).
// ------------------------------------------------------------------------------------------------
// Eight waves, two per SIMD (waves w and w + 4 share one): waves 0,1 run the SGD epochs and have their SIMDs to themselves
// (waves 4,5 only take part in the phases between the barriers), waves 2,3,6,7 run the forward passes two to a SIMD -- forward
// passes are full of transcendental and LDS latency, and two waves on a SIMD get 1.2 x the work done per cycle.  With four
// waves the 16 forward passes took as long as the 16 epochs once the tape and the proposals had moved into them.
constexpr int PK_WAVES = 8, PK_SWEEP_WAVES = 2, PK_FWD_WAVES = 4;
// lane groups of 2^nred hidden units: 8 (n_hidden <= 8: 16 slots per round) or 16 (n_hidden <= 16: 8 slots per round)
__host__ __device__ constexpr int pack_slots(int nred) { return PK_SWEEP_WAVES * (WAVE >> nred); }

__host__ __device__ inline size_t pack_slot_floats(int PS) { return 2 * (size_t)PS; }      // proposal, its SGD epoch
// the random tape lives in a ring of 2 x nslots steps: {noise[PS], lx, u, n_eta, -} per step
__host__ __device__ inline size_t pack_ring_floats(int PS, int nslots) { return (size_t)2 * nslots * ((size_t)PS + 4); }
__host__ __device__ inline size_t pack_lds_floats(int Nall, int IPY, int PS, int H, int FWS, int nslots) {
    return (size_t)(Nall + 2) * IPY + 4 * (size_t)PS + MAX_WAVES * 8 + (size_t)nslots * SL_COUNT +
           (size_t)nslots * pack_slot_floats(PS) + pack_ring_floats(PS, nslots) + (size_t)PK_WAVES * fw_floats(H, FWS);
}
// Several CUs per replica (MULTI, 16-lane groups only): every work-group runs a packed round over ITS PK_SLOTS slots of a window of
// G x PK_SLOTS steps; the ring holds twice the widest window (G <= PK_MULTI_MAXG), plus a staging area for the accepted step of
// another work-group {proposal, its epoch, slot scalars} and the groups' verdicts.
constexpr int PK_MULTI_MAXG = 4;
__host__ __device__ inline size_t pack_multi_lds_floats(int Nall, int IPY, int PS, int H, int FWS, int nslots) {
    return pack_lds_floats(Nall, IPY, PS, H, FWS, nslots) + pack_ring_floats(PS, (PK_MULTI_MAXG - 1) * nslots) + 2 * (size_t)PS + SL_COUNT + 8;
}

template <int TASK, int I, int O, int PK_NRED, bool MULTI = false>
__device__ __forceinline__ void segment_pack_body(const SegParams& p, const SegDyn& dyn, const int step_begin, const int n_steps) {
    constexpr int PK_NG = WAVE >> PK_NRED, PK_SLOTS = pack_slots(PK_NRED);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // MULTI: p.G work-groups (CUs) per replica, work-group grp owns slots [grp PK_SLOTS, (grp + 1) PK_SLOTS) of a window of KT steps;
    // all of them keep the chain state and apply the same commits (segment_spec_body's protocol: verdicts and the accepted step
    // cross CUs as {tag, value} granules)
    const int G = MULTI ? p.G : 1;
    const int lb = MULTI ? xcd_block(G) : (int)blockIdx.x;
    const int r = MULTI ? lb / G : lb, grp = MULTI ? lb - r * G : 0;
    const int KT = G * PK_SLOTS, s0 = grp * PK_SLOTS;
    const int gid = p.first_global + r;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int Nall = p.Ntr + p.Nte;
    const int P = p.P, PS = p.PS, H = p.H;
    float* q = smem;
    float* xy = q; q += (Nall + 2) * p.IPY;
    float* w_cur = q; q += PS;
    float* w_gd = q; q += PS;
    float* rec_w = q; q += PS;                              // last recorded pos_w row ...
    float* rec_alt = q; q += PS;                            // ... and where the next one goes: the two swap on every accepted step
    float* red = q; q += MAX_WAVES * 8;
    float* slots = q; q += PK_SLOTS * SL_COUNT;
    const size_t SLF = pack_slot_floats(PS);                // per slot: proposal, its SGD epoch
    float* sl0 = q; q += PK_SLOTS * SLF;
    // The random tape of a step depends on (seed, replica, step) only, so it is generated AHEAD of the rounds into a ring of
    // RING = 2 PK_SLOTS steps (step j lives in entry j mod RING): at the start of a round the ring holds steps [i, i + RING)
    // minus what the previous round committed, and the forward-pass waves refill it while the sweep waves are still sweeping.
    // The tape is never on the critical path of a round.
    constexpr int RING = (MULTI ? 2 * PK_MULTI_MAXG : 2) * PK_SLOTS;
    float* ring_n = q; q += (size_t)RING * PS;
    float* ring_s = q; q += (size_t)RING * 4;
    float* my_fw = q + (size_t)wave * fw_floats(H, p.FWS);
    q += (size_t)PK_WAVES * fw_floats(H, p.FWS);
    float* win_v = q; q += MULTI ? 2 * (size_t)PS : 0;      // MULTI: the accepted step of another work-group {proposal, its SGD epoch}
    float* win_s = q; q += MULTI ? SL_COUNT : 0;            //        its slot scalars
    float* gverd = q;                                       //        the groups' verdicts of this round (first accepted local slot, or -1)
    granule_t* const xv = MULTI ? p.xverdict + (size_t)r * 2 * MAX_SLOTS : nullptr;
    granule_t* const xsl = MULTI ? p.xslots + (size_t)r * 2 * MAX_SLOTS * SL_COUNT : nullptr;
    granule_t* const xwv = MULTI ? p.xw + (size_t)r * 2 * G * 2 * PS : nullptr;
    unsigned epoch = dyn.epoch_base;
    int xpar = 0;
    bool failed = false;
    auto s_prop = [&](int s_) { return sl0 + (size_t)s_ * SLF; };
    auto s_pgd = [&](int s_) { return sl0 + (size_t)s_ * SLF + PS; };

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
    __syncthreads();

    const float T = uni_f(p.temps[r]);
    float eta = (TASK == TASK_REG) ? uni_f(w_cur[P]) : 0.0f;
    float* sf = p.st_f + (size_t)r * SF_COUNT;
    int* si = p.st_i + (size_t)r * SI_COUNT;
    float lik, prior_cur, tau_eta_last, rec_rmse_tr, rec_rmse_te, rec_acc_tr, rec_acc_te;
    int nacc, gd_valid, lg_count, lg_acc;
    if (step_begin == 0) {
        lg_acc = 0;
        if (wave == 0) {
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

    PTNN_DIAG(pack_begin);
    const size_t trow = (size_t)r * p.trace_cap;
    const int end = step_begin + n_steps;
    const bool sweeping = p.use_lg != 0;
    // forward passes: waves 2,3(,6,7) while waves 0,1 sweep; all waves when there is nothing to sweep.  The host launches eight
    // waves while every replica has a CU to itself and four beyond that (with 232 VGPRs two waves fit on a SIMD: an eight-wave
    // work-group has the CU to itself, of four-wave ones two are resident -- 1024 replicas on one GPU: 105 M vs 91 M samples/s)
    const int nwaves = nthr >> 6;
    const int ev_n = sweeping ? (nwaves == PK_WAVES ? PK_FWD_WAVES : 2) : nwaves;
    const int ev_i = !sweeping ? wave : ((wave & 3) >= 2 ? (wave & 1) + ((wave >> 2) << 1) : -1);   // my index among them, or -1
    int i = step_begin;
    int tpos0 = (step_begin + 1) % p.trace_cap;            // ring position of the trace row of step i
    const float inv_PW = 1.0f / (float)p.PW;
    // tape of steps [lo, hi) into the ring, by the waves [w0, w0 + nw): one Philox counter quad per lane, a wave covers
    // 64 / (quads per step) steps in one pass
    const int nq1 = ((P + 3) >> 2) + 1;                     // noise quads + the scalar quad
    auto fill_ring = [&](int lo, int hi, int wi, int nw) {    // this wave is number wi of the nw waves that share the job
        if (nq1 <= WAVE) {
            const int per_pass = WAVE / nq1, ls = lane / nq1, q_ = lane - ls * nq1;
            for (int sb = lo + wi * per_pass; sb < hi; sb += nw * per_pass) {
                const int j = sb + ls;
                if (ls < per_pass && j < hi) {
                    const int slot = j % RING;
                    const bool sc = (q_ == nq1 - 1);
                    uint32_t x[4];
                    philox4x32_10(sc ? 0u : (uint32_t)q_, (uint32_t)j, p.noise_shared ? 0u : (uint32_t)gid, sc ? STREAM_STEP : STREAM_WNOISE,
                                  p.seed_lo, p.seed_hi, x);
                    float n0, n1, n2, n3;
                    box_muller(x[0], x[1], n0, n1);
                    box_muller(x[2], x[3], n2, n3);
                    if (sc) { float* sc_ = ring_s + slot * 4; sc_[0] = u23(x[0]); sc_[1] = u23(x[1]); sc_[2] = n2; }
                    else *reinterpret_cast<float4*>(ring_n + (size_t)slot * PS + 4 * q_) = make_float4(n0, n1, n2, n3);
                }
            }
        } else {
            for (int j = lo + wi; j < hi; j += nw) tape_step<true>(p, gid, j, ring_n + (size_t)(j % RING) * PS, ring_s + (j % RING) * 4);
        }
    };
    fill_ring(step_begin, step_begin + RING, wave, nwaves);
    int ring_hi = step_begin + RING;                        // first step whose tape is not in the ring yet
    // MULTI: when the work-groups of this replica share an XCD, the rounds' verdicts and the accepted step travel through that XCD's L2
    // (granule_*_xcd); asked once per launch (xcd_handshake; tag epoch_base: the rounds use epoch_base + 1 ...; granules 32 .. 32 + G of
    // the verdict row, which the rounds never touch)
    bool xcd_local = false;
    if constexpr (MULTI) {
        if (p.xcd_granules) xcd_local = xcd_handshake(xv + 32, G, grp, dyn.epoch_base, gverd);
    }
    __syncthreads();
    // MULTI, one launch for several swap intervals (PersistParams::swap_inside): the swap round after a hand-off step runs inside the
    // launch, at the end of the round loop below -- the same protocol as the tree's (segment_tree_body), with the cached gradient and
    // its flag travelling beside the state as swap_block moves them
    persist_cptr const pp = dyn.pp;
    const bool swap_inside = MULTI && pp->swap_inside != 0;
    const int si_ = pp->si;
    auto next_handoff = [&](int cur) {
        if (TASK == TASK_REG) { const int c1 = cur > 1 ? cur : 1; return ((c1 + si_ - 1) / si_) * si_; }
        return ((cur + si_) / si_) * si_ - 1;
    };
    int ho_next = swap_inside ? next_handoff(step_begin) : 0x7fffffff;
    int nx = 0;                                             // swap rounds done inside this launch
    while (i < end && !failed) {
        if (MULTI) { epoch += 1; }
        if (i == p.switch_step) {
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
        int kt = min(KT, end - i);                          // steps of this round's window (all work-groups of the replica)
        if (p.switch_step > i) kt = min(kt, p.switch_step - i);
        if (swap_inside && ho_next >= i) kt = min(kt, ho_next - i + 1);     // a window never crosses a hand-off
        const int k = MULTI ? max(0, min(PK_SLOTS, kt - s0)) : kt;   // ... of which this work-group computes slots s0 .. s0 + k - 1
        PTNN_DIAG(count_round);
        STAMP(0);
        if (sweeping && !gd_valid) {                       // chain start, or w arrived from another GPU
            if (wave == 0) sgd_sweep_dispatch<TASK, I, O>(w_cur, w_gd, xy, p.data, p.Ntr, H, p.lr);
            gd_valid = 1;
            __syncthreads();
        }
        const int rpos0 = (i + s0) % RING;                  // ring entry of this work-group's slot 0 (= step i + s0)
        auto s_noise = [&](int s_) { int e_ = rpos0 + s_; if (e_ >= RING) e_ -= RING; return ring_n + (size_t)e_ * PS; };
        auto s_scal = [&](int s_) { int e_ = rpos0 + s_; if (e_ >= RING) e_ -= RING; return ring_s + e_ * 4; };
        STAMP(1);
        STAMP(2);
        // SGD epochs of all slots in lane groups (waves 0,1) || forward passes (the other waves).  Nobody waits for a proposal
        // phase: a proposal is base + step_w * noise with base = w_gd or w_cur by the step's Langevin coin, and each consumer
        // forms the elements it needs -- the sweep lanes their own weights, the forward wave of a slot the whole vector, which
        // it also writes out (the commit and an accepted step need it).
        if (sweeping && wave < PK_SWEEP_WAVES) {
            const int ng = min(PK_NG, k - wave * PK_NG);
            if (ng > 0) {
                SweepProposals pp;
                pp.noise = ring_n; pp.scal = ring_s; pp.w_cur = w_cur; pp.w_gd = w_gd;
                pp.pos0 = rpos0 + wave * PK_NG; if (pp.pos0 >= RING) pp.pos0 -= RING;
                pp.ring = RING; pp.nstride = PS; pp.step_w = p.step_w; pp.l_prob = p.l_prob; pp.use_lg = 1;
                sgd_sweep<TASK, I, O, PK_NRED, true>(nullptr, s_pgd(wave * PK_NG), xy, p.data, p.Ntr, H, p.lr, ng, (int)SLF, &pp);
            }
        }
        STAMP(3);                                           // sweep
        PTNN_DIAG(pack_eval_begin);
        if (ev_i >= 0) {
            for (int s_ = ev_i; s_ < k; s_ += ev_n) {
                const int j = i + s0 + s_;
                const float adapttemp = (p.switch_step >= 0 && j >= p.switch_step) ? 1.0f : T;
                const float* sc_ = s_scal(s_);
                const bool lg = sweeping && (sc_[0] < p.l_prob);
                float eta_pro = eta;
                if (TASK == TASK_REG) eta_pro = fmaf(p.step_eta, sc_[2], eta);
                {
                    const float* nz = s_noise(s_);
                    const float* base = lg ? w_gd : w_cur;
                    float* pr = s_prop(s_);
                    for (int e = lane; e < P; e += WAVE) pr[e] = fmaf(p.step_w, nz[e], base[e]);
                }
                gsync<true>();
                build_fw<I, O, true>(s_prop(s_), my_fw, H, p.FWS);
                gsync<true>();
                const EvalSums es = eval_rows<TASK, I, O, true>(my_fw, xy, p.IPY, p.FWS, H, p.Ntr, Nall, nullptr);
                float ll, rm_tr, rm_te, ac_tr, ac_te;
                finish_eval<TASK, true>(es, p.Ntr, p.Nte, eta_pro, ll, rm_tr, rm_te, ac_tr, ac_te);
                const float ssq = block_sumsq<true>(s_prop(s_), P, nullptr);
                const float prior_prop = prior_value<TASK>(p, ssq, eta_pro);
                // |noise|^2 of the Langevin ratio does not wait for the epoch: taken here, off the critical path (same
                // association order as the one-wave-per-slot schedule's block_sumsq)
                const float d2 = lg ? block_sumsq<true>(s_noise(s_), P, nullptr) : 0.0f;
                if (lane == 0) {
                    float* sl = slots + s_ * SL_COUNT;
                    sl[SL_LIKPROP] = ll / adapttemp; sl[SL_PRIORPROP] = prior_prop; sl[SL_ETAPRO] = eta_pro;
                    sl[SL_RM_TR] = rm_tr; sl[SL_RM_TE] = rm_te; sl[SL_AC_TR] = ac_tr; sl[SL_AC_TE] = ac_te;
                    sl[SL_ADAPT] = adapttemp; sl[SL_LG] = lg ? 1.0f : 0.0f; sl[SL_D2] = d2;
                }
                gsync<true>();
            }
            // tape of the steps the next round may reach and the ring does not hold yet: at most as many as the previous round
            // committed.  They overwrite entries of steps below i.
            fill_ring(ring_hi, i + RING, ev_i, ev_n);
        }
        ring_hi = i + RING;
        PTNN_DIAG(pack_eval_end);
        __syncthreads();
        STAMP(4);                                           // waiting for the forward passes
        // phase 3: Metropolis-Hastings ratio of every slot
        // one 16-lane row per slot, all 16 slots at once (waves 0-3).  |w - epoch(proposal)|^2 is taken in the association order
        // of block_sumsq_diff<true> (element j in lane j of a wave: rows of 16, then (row0 + row1) + (row2 + row3)), so the
        // decision is bit-identical to the one-wave-per-slot schedule.
        {
            const int s_ = wave * (WAVE / 16) + (lane >> 4), l16 = lane & 15;
            const bool on = s_ < k;
            float* sl = slots + (on ? s_ : 0) * SL_COUNT;
            float diff_prop = 0.0f;
            if (sweeping && wave < 4) {
                const float* pg = s_pgd(on ? s_ : 0);
                float r1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {                 // row c of the wave layout: elements 16 c + l16 (+ 64 t)
                    float a1 = 0.0f;
                    for (int e = 16 * c + l16; e < P; e += WAVE) {
                        const float d = w_cur[e] - pg[e];
                        a1 = fmaf(d, d, a1);
                    }
                    r1[c] = (16 * c < P) ? group_allsum<4>(a1) : 0.0f;     // a row beyond P sums zeros: exactly 0 either way
                }
                const float d1 = (r1[0] + r1[1]) + (r1[2] + r1[3]);
                if (sl[SL_LG] != 0.0f) diff_prop = langevin_ratio(d1, sl[SL_D2], p.step_w, sl[SL_ADAPT]);
            }
            const float logalpha = (sl[SL_LIKPROP] - lik) + (sl[SL_PRIORPROP] - prior_cur) + diff_prop;
            const float mh = (logalpha != logalpha) ? 1.0f : fminf(1.0f, expf_fast(logalpha));
            if (on && l16 == 0) { sl[SL_ACCEPT] = (s_scal(s_)[1] < mh) ? 1.0f : 0.0f; sl[SL_LOGALPHA] = logalpha; }
        }
        __syncthreads();
        STAMP(5);                                           // MH
        // commit the prefix up to and including the first accepted step
        const bool f_acc = (lane < k) && (slots[lane * SL_COUNT + SL_ACCEPT] != 0.0f);
        const unsigned long long bal_acc = __ballot(f_acc);
        const int ml = bal_acc ? (__ffsll((long long)bal_acc) - 1) : -1;          // first accepted LOCAL slot, or none
        int m = (ml >= 0) ? s0 + ml : kt;                                           // first accepted step of the window (slot index)
        const float* sm = slots + (ml >= 0 ? ml : 0) * SL_COUNT;                    // its scalars ...
        const float* wacc = s_prop(ml >= 0 ? ml : 0);                               // ... and vectors {proposal, its epoch}
        if constexpr (MULTI) {
            // every work-group posts its verdict; one with an accepted slot posts that slot's scalars and vectors with it (it may not
            // be the window's first: the readers take the winner's only)
            granule_t* const xv_r = xv + (size_t)xpar * MAX_SLOTS;
            auto gstore = [&](granule_t* g_, float v_) { if (xcd_local) granule_store_xcd(g_, epoch, v_); else granule_store(g_, epoch, v_); };
            auto gwait = [&](const granule_t* g_, float& v_) { return xcd_local ? granule_wait_xcd(g_, epoch, v_) : granule_wait(g_, epoch, v_); };
            if (tid == 0) gstore(xv_r + grp, (float)ml);
            if (ml >= 0) {
                if (tid < SL_COUNT) gstore(xsl + ((size_t)xpar * MAX_SLOTS + grp) * SL_COUNT + tid, sm[tid]);
                granule_t* const xo = xwv + ((size_t)xpar * G + grp) * 2 * PS;
                for (int e = tid; e < 2 * PS; e += nthr) gstore(xo + e, wacc[e]);
            }
            if (tid < G) {
                float v = 0.0f;
                if (!gwait(xv_r + tid, v)) v = -2.0f;
                gverd[tid] = v;
            }
            __syncthreads();
            int win = -1;
            m = kt;
            for (int g_ = 0; g_ < G; ++g_) {
                const float v = gverd[g_];
                if (v == -2.0f) failed = true;
                const int cand = (v >= 0.0f) ? g_ * PK_SLOTS + (int)v : kt;
                if (cand < m) { m = cand; win = g_; }
            }
            if (failed) break;
            if (win >= 0 && win != grp) {                       // the accepted step was computed elsewhere: fetch it
                bool ok = true;
                if (tid < SL_COUNT) {
                    float v;
                    ok = gwait(xsl + ((size_t)xpar * MAX_SLOTS + win) * SL_COUNT + tid, v);
                    win_s[tid] = v;
                }
                const granule_t* const xi = xwv + ((size_t)xpar * G + win) * 2 * PS;
                for (int e = tid; e < 2 * PS; e += nthr) {
                    float v;
                    ok = gwait(xi + e, v) && ok;
                    win_v[e] = v;
                }
                failed = __syncthreads_or(ok ? 0 : 1) != 0;
                if (failed) break;
                sm = win_s; wacc = win_v;
            }
            xpar ^= 1;
        }
        const int ncommit = (m < kt) ? m + 1 : kt;
        const int nloc = max(0, min(k, ncommit - s0));          // of them this work-group's own slots: it writes their trace rows
        for (int item = tid; item < nloc * p.PW; item += nthr) { // trace rows: all (slot, element) pairs at once
            const int s_ = (int)(((float)item + 0.5f) * inv_PW), e = item - s_ * p.PW;
            int tp = tpos0 + s0 + s_;
            if (tp >= p.trace_cap) tp -= p.trace_cap;
            p.tr_pos_w[(trow + (size_t)tp) * (size_t)p.PW + e] = (e < P) ? ((s0 + s_ == m) ? s_prop(s_) : rec_w)[e] : 0.0f;
        }
        if (tid < nloc) {
            const int s_ = tid;
            const bool acc_me = (s0 + s_ == m);
            const float* sl = slots + s_ * SL_COUNT;
            int tp = tpos0 + s0 + s_;
            if (tp >= p.trace_cap) tp -= p.trace_cap;
            const size_t tpos = trow + (size_t)tp;
            store_trace_row(p.tr_scal + tpos * TR_COUNT, (TASK == TASK_REG) ? sl[SL_LIKPROP] : sl[SL_LIKPROP] * sl[SL_ADAPT],
                            acc_me ? sl[SL_RM_TR] : rec_rmse_tr, acc_me ? sl[SL_RM_TE] : rec_rmse_te,
                            acc_me ? sl[SL_AC_TR] : rec_acc_tr, acc_me ? sl[SL_AC_TE] : rec_acc_te, nacc, sl[SL_LOGALPHA]);
        }
        if constexpr (MULTI) {
            // Langevin coins and the last proposed eta of the committed steps, from the tape (every work-group holds the whole
            // window's): the same expressions the forward waves evaluate for their own slots
            const int e_ = (i + lane) % RING;
            const bool f_lg = sweeping && (lane < ncommit) && (ring_s[e_ * 4] < p.l_prob);
            lg_count += __popcll(__ballot(f_lg));
            if (TASK == TASK_REG) tau_eta_last = uni_f(fmaf(p.step_eta, ring_s[((i + ncommit - 1) % RING) * 4 + 2], eta));
        } else {
            const bool f_lg = (lane < k) && (slots[lane * SL_COUNT + SL_LG] != 0.0f);
            const unsigned long long bal_lg = __ballot(f_lg);
            lg_count += __popcll(bal_lg & ((1ull << ncommit) - 1ull));
            if (TASK == TASK_REG) tau_eta_last = uni_f(slots[(ncommit - 1) * SL_COUNT + SL_ETAPRO]);
        }
        if (m < kt) {
            // no barrier between the trace rows above and this update: they read rec_w, the new recorded row goes to rec_alt
            nacc += 1;
            lik = uni_f(sm[SL_LIKPROP]); prior_cur = uni_f(sm[SL_PRIORPROP]); eta = uni_f(sm[SL_ETAPRO]);
            rec_rmse_tr = uni_f(sm[SL_RM_TR]); rec_rmse_te = uni_f(sm[SL_RM_TE]);
            rec_acc_tr = uni_f(sm[SL_AC_TR]); rec_acc_te = uni_f(sm[SL_AC_TE]);
            lg_acc += (sm[SL_LG] != 0.0f) ? 1 : 0;
            gd_valid = sweeping ? 1 : 0;
            for (int e = tid; e < P; e += nthr) {
                const float v = wacc[e];
                w_cur[e] = v; rec_alt[e] = v;
                if (sweeping) w_gd[e] = wacc[PS + e];
            }
            float* t_ = rec_w; rec_w = rec_alt; rec_alt = t_;
        }
        __syncthreads();
        i += ncommit;
        tpos0 += ncommit;
        if (tpos0 >= p.trace_cap) tpos0 -= p.trace_cap;
        STAMP(6);                                           // commit
        if constexpr (MULTI) {
            if (swap_inside && i == ho_next + 1 && ho_next < end) {
                // ---- the swap round of this hand-off (REG:427-437 <-> 719-752) inside the launch: see segment_tree_body.  A row is
                // {state (w, eta): PS | cached langevin_gradient(w): PS | its valid flag}
                const int Rg = pp->sp.R, ROW = 2 * PS + 8;
                const unsigned xtag = dyn.epoch_base + (unsigned)nx + 1u;
                granule_t* const xl = p.xswap + (size_t)(nx & 1) * swap_xchg_granules(Rg, ROW);
                granule_t* const xst = xl + ((Rg + 7) & ~7);
                granule_t* const xsb = xst + (size_t)Rg * ROW + (size_t)gid * (ROW + 8);
                auto gstore = [&](granule_t* g_, float v_) { if (xcd_local) granule_store_xcd(g_, xtag, v_); else granule_store(g_, xtag, v_); };
                auto gwait = [&](const granule_t* g_, float& v_) { return xcd_local ? granule_wait_xcd(g_, xtag, v_) : granule_wait(g_, xtag, v_); };
                auto row_value = [&](int j) { return (j < PS) ? ((j == P) ? eta : ((j < P) ? w_cur[j] : 0.0f)) : ((j < 2 * PS) ? w_gd[j - PS] : ((j == 2 * PS) ? (gd_valid ? 1.0f : 0.0f) : 0.0f)); };
                auto row_take = [&](int j, float v) { if (j < PS) w_cur[j] = v; else if (j < 2 * PS) w_gd[j - PS] = v; else if (j == 2 * PS) red[1] = v; };
                bool ok = true;
                bool moved = false;
                if (grp == 0) {
                    if (tid == 0) granule_store(xl + gid, xtag, (TASK == TASK_REG) ? lik * T : lik);
                    for (int j = tid; j <= 2 * PS; j += nthr) granule_store(xst + (size_t)gid * ROW + j, xtag, row_value(j));
                    float* const sL = sl0;                   // the slots' proposals are dead between two rounds
                    float* const sU = sl0 + Rg;
                    int* const sSrc = reinterpret_cast<int*>(sl0 + 2 * Rg);
                    for (int k_ = tid; k_ < Rg; k_ += nthr) { float v = 0.0f; ok = granule_wait(xl + k_, xtag, v) && ok; sL[k_] = v; }
                    if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
#if defined(__HIP_DEVICE_COMPILE__)
                    const SwapParams sp = pp->sp;
#else
                    const SwapParams sp{};
#endif
                    const int round = pp->round0 + nx;
                    const int nsw = cascade_lds(sp, round, sL, sU, sSrc, true);
                    const int src = sSrc[gid];
                    if (gid == sp.first_global) {            // replica 0's root keeps the books (swap_block: b == 0)
                        if (sp.src_log && round < sp.log_capacity)
                            for (int k_ = tid; k_ < Rg; k_ += nthr) sp.src_log[(size_t)round * Rg + k_] = sSrc[k_];
                        if (tid == 0) { sp.counters[0] += nsw; sp.counters[1] += Rg - 1; }
                    }
                    __syncthreads();
                    moved = src != gid;
                    if (moved) {
                        for (int j = tid; j <= 2 * PS; j += nthr) { float v = 0.0f; ok = granule_wait(xst + (size_t)src * ROW + j, xtag, v) && ok; row_take(j, v); }
                        if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
                        for (int j = tid; j <= 2 * PS; j += nthr) gstore(xsb + 8 + j, (j == 2 * PS) ? red[1] : ((j < PS) ? w_cur[j] : w_gd[j - PS]));
                    }
                    if (tid == 0) gstore(xsb, moved ? 1.0f : 0.0f);
                } else {
                    if (tid == 0) { float mv = 0.0f; ok = gwait(xsb, mv); red[0] = mv; }
                    if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
                    moved = red[0] != 0.0f;
                    if (moved) {
                        for (int j = tid; j <= 2 * PS; j += nthr) { float v = 0.0f; ok = gwait(xsb + 8 + j, v) && ok; row_take(j, v); }
                        if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
                    }
                }
                __syncthreads();
                if (moved) {                                 // what arrived: eta and the gradient's flag travel with the state (REG:436-437)
                    if (TASK == TASK_REG) eta = uni_f(w_cur[P]);
                    gd_valid = (uni_f(red[1]) != 0.0f) ? 1 : 0;
                }
                __syncthreads();
                nx += 1;
                ho_next = next_handoff(i);
            }
        }
    }
    PTNN_DIAG(pack_flush);

    if (failed) {
        if (tid == 0) atomicAdd(p.error_flag, 1);           // a bounded spin ran out: the host reports it
        return;
    }
    if (grp != 0) return;                                   // every work-group holds the same state: the first one writes it back
    const int fl_end = (pp->flip0 + nx) & 1;               // every in-launch round flips the host's buffers
    float* const gw_end = swap_inside ? pp->state[fl_end] + (size_t)r * PS : gw;
    float* const gd_end = swap_inside ? pp->gd[fl_end] + (size_t)r * PS : dyn.gd_w + (size_t)r * PS;
    int* const gdv_end = swap_inside ? pp->gd_valid[fl_end] : dyn.gd_valid;
    for (int j = tid; j < PS; j += nthr) {
        gw_end[j] = (j == P) ? eta : w_cur[j];
        p.rec_w[(size_t)r * PS + j] = rec_w[j];
        gd_end[j] = w_gd[j];
    }
    if (tid == 0) {
        sf[SF_LIK] = lik; sf[SF_PRIOR] = prior_cur; sf[SF_TAU_LAST] = tau_eta_last;
        sf[SF_REC_RMSE_TR] = rec_rmse_tr; sf[SF_REC_RMSE_TE] = rec_rmse_te;
        sf[SF_REC_ACC_TR] = rec_acc_tr; sf[SF_REC_ACC_TE] = rec_acc_te;
        si[SI_NACC] = nacc; gdv_end[r] = gd_valid; si[SI_LG_COUNT] = lg_count; si[SI_LG_ACC] = lg_acc;
        p.L_handoff[gid] = (TASK == TASK_REG) ? lik * T : lik;
        p.L_final[gid] = lik;
        post_raw(p, gid, lik, prior_cur, T, step_begin + n_steps - 1);
    }
}

// ------------------------------------------------------------------------------------------------
