// ptnn_dev_tree.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// prefetching tree schedule (segment_tree_body), with its own swap rounds inside a launch.

// ------------------------------------------------------------------------------------------------
// Prefetching ("tree") schedule for random-walk classification chains, where half of the proposals are accepted and
// speculating on rejections alone gains nothing: G = 2^D - 1 work-groups (all resident; several may share a CU) evaluate, at the same time, the
// proposals of ALL 2^D - 1 outcomes of the next D accept/reject decisions.  Work-group g is node g + 1 of a binary heap:
// the root proposes step i from the current state; the left child of a node proposes the next step assuming the node's
// proposal was rejected (same state), the right child assuming it was accepted (state = the node's proposal).  A random-walk
// proposal is state + step * noise and the noise of a step depends on the step number only, so a node forms its proposal
// from the shared state and the tapes of the steps on its path -- the same fused multiply-adds, in the same order, as the
// sequential chain -- runs the cooperative forward pass on it and publishes ONE record {likelihood, prior, scores}.  Every
// work-group then reads all records, walks the D decisions (each against the likelihood / prior of the state the walk has
// reached), and rebuilds the new state locally from the tapes: D steps are committed per round, whatever the decisions,
// and no vector ever crosses CUs.  Bit-identical to the cooperative schedule at the same block size (tested).
// Records are 8-byte {tag, value} granules (granule_store / granule_wait), two-deep by round parity: a work-group needs
// every record of round n before it can publish round n + 1, so nobody is more than one round ahead.
// ------------------------------------------------------------------------------------------------
constexpr int TREE_MAX_DEPTH = 5;
constexpr int TREE_MAX_NODES = 31;
constexpr int TREE_REC = 8;            // row stride of a record
constexpr int TREE_FIELDS = 6;         // lik_prop, prior_prop, rmse_tr, rmse_te, acc_tr, acc_te: what is published and polled
// mfma: the forward pass reads the transposed data image (behind this block), so only the labels of the row-major image are kept
__host__ __device__ inline size_t tree_lds_floats(int Nall, int IPY, int PS, int H, int FWS, int D, bool ahead, bool mfma) {
    size_t tapes = (size_t)(ahead ? 2 : 1) * D * (PS + 8);
    if (mfma && tapes < fw_floats(H, FWS)) tapes = fw_floats(H, FWS);      // the start-up builds its forward image there
    return (mfma ? (size_t)((Nall + 3) & ~3) : (size_t)(Nall + 2) * IPY) + 3 * (size_t)PS + tapes + (mfma ? 0 : fw_floats(H, FWS)) +
           MAX_WAVES * 8 + (size_t)(TREE_MAX_NODES + 1) * TREE_REC;
}

constexpr int TREE_PERSIST_MAX_R = ((TREE_MAX_NODES + 1) * TREE_REC - 1) / 3;      // the cascade's 3 R + 1 floats live in the record area of LDS

template <int TASK, int I, int O>
__device__ __forceinline__ void segment_tree_body(const SegParams& p, const SegDyn& dyn, const int step_begin, const int n_steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int G = p.G;                                     // 2^D - 1
    const int D = 31 - __clz(G + 1);
    const int lb = xcd_block(G);
    const int r = lb / G, g = lb - r * G;
    const int node = g + 1, depth = 31 - __clz(node);      // heap index, level (root: 0)
    const int gid = p.first_global + r;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int Nall = p.Ntr + p.Nte;
    const int P = p.P, PS = p.PS, H = p.H;
    const bool mfma = p.fw_mfma != 0;
    float* xy = smem;                                      // row-major data image, or (matrix-core forward) just its labels
    float* w_cur = xy + (mfma ? (size_t)((Nall + 3) & ~3) : (size_t)(Nall + 2) * p.IPY);
    float* w_prop = w_cur + PS;
    float* rec_w = w_prop + PS;
    float* const tapes0 = rec_w + PS;                      // D x {noise[PS], scal[8]}, twice when the tapes are drawn ahead
    const bool ahead = p.tree_ahead != 0;
    size_t tape_floats = (size_t)(ahead ? 2 : 1) * D * (PS + 8);
    if (mfma && tape_floats < fw_floats(H, p.FWS)) tape_floats = fw_floats(H, p.FWS);
    float* fw = tapes0 + tape_floats;
    float* red = fw + (mfma ? 0 : fw_floats(H, p.FWS));
    float* recs = red + MAX_WAVES * 8;                     // [nodes][TREE_REC]
    float* xt_l = smem + ((tree_lds_floats(Nall, p.IPY, PS, H, p.FWS, D, ahead, mfma) + 3) & ~(size_t)3);
    float* part_l = xt_l + (size_t)I * p.Npad;
    const bool split = SplitK<I>::OK && p.fw_mfma == 2;     // split-operand forward pass: the cooperative kernel's (same arithmetic, same chain)
    SplitLds sl = {};
    // what the scoring reads as xy[n * stride + I]
    const float* const ysrc = mfma ? xy - I : xy;
    const int ystride = mfma ? 1 : p.IPY;
    float* gw = dyn.w_state + (size_t)r * PS;
    auto stage = [&]() {                                   // the launch's working set: global memory -> LDS
        if (mfma) {
            for (int n = tid; n < Nall; n += nthr) xy[n] = p.data[(size_t)n * p.IPY + I];
        } else {
            const float4* src = reinterpret_cast<const float4*>(p.data);
            float4* dst = reinterpret_cast<float4*>(xy);
            for (int e = tid; e < ((Nall + 2) * p.IPY) >> 2; e += nthr) dst[e] = src[e];
        }
        for (int j = tid; j < PS; j += nthr) {
            w_cur[j] = gw[j];
            rec_w[j] = p.rec_w[(size_t)r * PS + j];
        }
        if constexpr (SplitK<I>::OK) {
            if (split) {
                sl = carve_split<I>(xt_l, O, H, p.Npad);
                stage_split_data<I>(sl, p.data, p.IPY, Nall, p.Npad);
            }
        }
        if (p.fw_mfma == 1)
            for (int e = tid; e < I * p.Npad; e += nthr) xt_l[e] = p.xt[e];
    };
    stage();
    __syncthreads();
    // When all groups of the replica sit on one XCD (xcd_block arranges that wherever the grid allows; asked once per launch:
    // xcd_handshake, granule 7 of every node's record row, which the records do not use), the records travel through that XCD's L2
    // alone (granule_store_xcd / granule_wait_xcd), else through the agent-scope path.  (Letting the root group stage first, so
    // that the other 14 hit the lines it brought into the L2, was measured: no less fetched, 1 % slower -- what this kernel still
    // fetches per launch is its own code and the data image once per XCD, profiles/README.md.)
    granule_t* const xrec = reinterpret_cast<granule_t*>(p.xslots) + (size_t)r * 2 * (TREE_MAX_NODES + 1) * TREE_REC;
    const bool xcd_local = p.xcd_granules != 0 && xcd_handshake_strided(xrec + 7, TREE_REC, G, g, dyn.epoch_base, recs);

    const float T = p.temps[r];
    float eta = 0.0f;                                      // classification: no noise parameter
    float* sf = p.st_f + (size_t)r * SF_COUNT;
    int* si = p.st_i + (size_t)r * SI_COUNT;
    float lik, prior_cur, rec_rmse_tr, rec_rmse_te, rec_acc_tr, rec_acc_te;
    int nacc;
    if (step_begin == 0) {                                 // every group of the replica computes the same start-up
        // matrix-core mode keeps neither the row-major image nor the packed forward image in LDS: the one-off start-up reads
        // the image from global memory and builds its forward image in the (still unused) tape area
        chain_startup<TASK, I, O>(p, mfma ? p.data : xy, w_cur, mfma ? tapes0 : fw, red, T, eta, lik, prior_cur);
        rec_rmse_tr = rec_rmse_te = rec_acc_tr = rec_acc_te = 0.f;
        nacc = 0;
        __syncthreads();
    } else {
        lik = sf[SF_LIK]; prior_cur = sf[SF_PRIOR];
        rec_rmse_tr = sf[SF_REC_RMSE_TR]; rec_rmse_te = sf[SF_REC_RMSE_TE];
        rec_acc_tr = sf[SF_REC_ACC_TR]; rec_acc_te = sf[SF_REC_ACC_TE];
        nacc = si[SI_NACC];
    }
    lik = uni_f(lik); prior_cur = uni_f(prior_cur); nacc = uni_i(nacc);
    rec_rmse_tr = uni_f(rec_rmse_tr); rec_rmse_te = uni_f(rec_rmse_te); rec_acc_tr = uni_f(rec_acc_tr); rec_acc_te = uni_f(rec_acc_te);

    const size_t trow = (size_t)r * p.trace_cap;
    const int step_end = step_begin + n_steps;
    const int nq1 = ((P + 3) >> 2) + 1;
    unsigned epoch = dyn.epoch_base;
    int par = 0;
    bool failed = false;
    int i = step_begin;
    // One launch for several swap intervals (PersistParams::swap_inside; the host takes this shape when the whole grid is resident):
    // the swap round after a hand-off step runs inside the launch, below.  ho_next = the next hand-off step (Q10: REG after step i
    // when i % si == 0 and i != 0, CLS when (i + 1) % si == 0), as persistent_loop and ptnn_run find it.
    persist_cptr const pp = dyn.pp;
    const bool swap_inside = pp->swap_inside != 0;
    const int si_ = pp->si;
    auto next_handoff = [&](int cur) {
        if (TASK == TASK_REG) { const int c1 = cur > 1 ? cur : 1; return ((c1 + si_ - 1) / si_) * si_; }
        return ((cur + si_) / si_) * si_ - 1;
    };
    int ho_next = swap_inside ? next_handoff(step_begin) : 0x7fffffff;
    int nx = 0;                                             // swap rounds done inside this launch
    // steps of the round that starts at step `first`: a round never crosses the temperature switch (its re-evaluation opens one)
    // nor a hand-off
    auto round_steps = [&](int first) {
        int n = min(D, step_end - first);
        if (p.switch_step > first) n = min(n, p.switch_step - first);
        if (swap_inside && ho_next >= first) n = min(n, ho_next - first + 1);
        return n;
    };
    // the random tapes of `count` steps from `first` (tape_step's body, flattened over (step, counter quad))
    auto draw_tapes = [&](float* base, int first, int count) {
        for (int e = tid; e < count * nq1; e += nthr) {
            const int l = e / nq1, q = e - l * nq1;
            const bool sc = (q == nq1 - 1);
            float* tp = base + (size_t)l * (PS + 8);
            uint32_t x[4];
            philox4x32_10(sc ? 0u : (uint32_t)q, (uint32_t)(first + l), p.noise_shared ? 0u : (uint32_t)gid, sc ? STREAM_STEP : STREAM_WNOISE,
                          p.seed_lo, p.seed_hi, x);
            float n0, n1, n2, n3;
            box_muller(x[0], x[1], n0, n1);
            box_muller(x[2], x[3], n2, n3);
            if (sc) { tp[PS] = u23(x[0]); tp[PS + 1] = u23(x[1]); tp[PS + 2] = n2; }
            else *reinterpret_cast<float4*>(tp + 4 * q) = make_float4(n0, n1, n2, n3);
        }
    };
    int tpar = 0;
    if (ahead && i < step_end) { draw_tapes(tapes0, i, round_steps(i)); __syncthreads(); }
    PTNN_DIAG(tree_begin);
    while (i < step_end) {
        const int dr = round_steps(i);
        float* const tapes = tapes0 + (size_t)tpar * D * (PS + 8);
        const float adapttemp = (p.switch_step >= 0 && i >= p.switch_step) ? 1.0f : T;
        if (i == p.switch_step) {                            // re-evaluate the current w untempered (Q9, REG:322 / CLS)
            EvalSums sc;
            float none = 0.0f;
            if (split) {
                if constexpr (SplitK<I>::OK) {
                    split_weights<I>(sl.as, H, [&](int idx) { return w_cur[idx]; });
                    __syncthreads();
                    sc = eval_rows_mfma_split<TASK, I, O>(w_cur, sl, H, p.Ntr, Nall, p.Npad, red, none);
                }
            } else if (p.fw_mfma) {
                sc = eval_rows_mfma_coop<TASK, I, O>(w_cur, xt_l, part_l, ysrc, ystride, H, p.Ntr, Nall, p.Npad, red, none);
            } else {
                build_fw<I, O>(w_cur, fw, H, p.FWS);
                __syncthreads();
                sc = eval_rows<TASK, I, O>(fw, xy, p.IPY, p.FWS, H, p.Ntr, Nall, red);
            }
            lik = uni_f(finish_loglik<TASK>(sc, p.Ntr, eta));
            __syncthreads();
        }
        STAMP(0);
        // 1. the random tapes of the dr steps (drawn during the previous round's exchange when LDS has room for two sets)
        if (!ahead) { draw_tapes(tapes, i, dr); __syncthreads(); }
        STAMP(1);
        // 2. this node's proposal: the state after the accepted ancestors on its path, plus its own step
        const bool active = depth < dr;
        auto path_value = [&](int idx) {
            float v = w_cur[idx];
            for (int l = 0; l < depth; ++l)
                if ((node >> (depth - l - 1)) & 1) v = fmaf(p.step_w, tapes[(size_t)l * (PS + 8) + idx], v);
            return fmaf(p.step_w, tapes[(size_t)depth * (PS + 8) + idx], v);
        };
        if (active) {
            if (p.fw_mfma) {
                for (int j = tid; j < P; j += nthr) w_prop[j] = path_value(j);
                if constexpr (SplitK<I>::OK) { if (split) split_weights<I>(sl.as, H, [&](int idx) { return path_value(idx); }); }
            } else {
                const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
                constexpr int K = I + 1 + O;
                const bool pairs = FwLayout<I>::pairs(H);
                const int HR = pairs ? 2 * fw_pairs(H) : H;
                for (int e = tid; e < HR * K; e += nthr) {
                    const int h = e / K, c = e - h * K;
                    float v = 0.0f;
                    if (h < H) {
                        const int idx = (c < I) ? c * H + h : (c == I) ? oB1 + h : oW2 + h * O + (c - I - 1);
                        v = path_value(idx);
                        w_prop[idx] = v;
                    }
                    fw[pairs ? (h >> 1) * 2 * p.FWS + 2 * c + (h & 1) : h * p.FWS + c] = v;
                }
                if (tid < O) {
                    const float v = path_value(oB2 + tid);
                    w_prop[oB2 + tid] = v;
                    fw[HR * p.FWS + tid] = v;
                }
            }
        }
        __syncthreads();
        STAMP(2);
        // 3. forward pass of the node's proposal (the cooperative kernel's phase B)
        float rv[TREE_FIELDS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (active) {
            float ssq = 0.0f;
            for (int j = tid; j < P; j += nthr) ssq = fmaf(w_prop[j], w_prop[j], ssq);
            EvalSums es;
            if (split) { if constexpr (SplitK<I>::OK) es = eval_rows_mfma_split<TASK, I, O, true>(w_prop, sl, H, p.Ntr, Nall, p.Npad, red, ssq); }
            else if (p.fw_mfma) es = eval_rows_mfma_coop<TASK, I, O, true>(w_prop, xt_l, part_l, ysrc, ystride, H, p.Ntr, Nall, p.Npad, red, ssq);
            else es = eval_rows<TASK, I, O, false, true>(fw, xy, p.IPY, p.FWS, H, p.Ntr, Nall, red, ssq);
            rv[0] = finish_loglik<TASK>(es, p.Ntr, eta) / adapttemp;
            rv[1] = prior_value<TASK>(p, ssq, eta);
            finish_scores<TASK>(es, p.Ntr, p.Nte, rv[2], rv[3], rv[4], rv[5]);
            if (TASK == TASK_REG) rv[4] = eta;                // as finish_eval<TASK, true>
        }
        STAMP(3);
        // 4. publish the record (idle nodes publish their tag too: everybody waits for everybody, which keeps the groups
        //    within one round of each other)
        granule_t* const xr = xrec + (size_t)par * (TREE_MAX_NODES + 1) * TREE_REC;
        if (tid < TREE_FIELDS) {
            float v = rv[0];
#pragma unroll
            for (int f = 1; f < TREE_FIELDS; ++f) v = (tid == f) ? rv[f] : v;
            if (xcd_local) granule_store_xcd(xr + (size_t)g * TREE_REC + tid, epoch, v);
            else granule_store(xr + (size_t)g * TREE_REC + tid, epoch, v);
        }
        // ... and while the records travel, the tapes of the next round (they depend on step numbers only)
        if (ahead && i + dr < step_end) draw_tapes(tapes0 + (size_t)(tpar ^ 1) * D * (PS + 8), i + dr, round_steps(i + dr));
        STAMP(4);
        // 5. all records of the round
        bool ok = true;
        for (int q = tid; q < G * TREE_FIELDS; q += nthr) {
            const int nd_ = q / TREE_FIELDS, f_ = q - nd_ * TREE_FIELDS;
            float v = 0.0f;
            ok = (xcd_local ? granule_wait_xcd(xr + (size_t)nd_ * TREE_REC + f_, epoch, v) : granule_wait(xr + (size_t)nd_ * TREE_REC + f_, epoch, v)) && ok;
            recs[nd_ * TREE_REC + f_] = v;
        }
        if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }

        STAMP(5);
        // 6. the dr decisions, by every thread alike: each against the likelihood / prior of the state the walk has reached
        unsigned accmask = 0, my_mask = 0;
        int my_acc_before = 0;
        float my_likprop = 0.f, my_logalpha = 0.f, my_s0 = 0.f, my_s1 = 0.f, my_s2 = 0.f, my_s3 = 0.f;
        int nd = 1;
        for (int l = 0; l < dr; ++l) {
            const float* rc = recs + (size_t)(nd - 1) * TREE_REC;
            const float u = tapes[(size_t)l * (PS + 8) + PS + 1];
            const float lik_prop = rc[0], prior_prop = rc[1];
            // R9 Metropolis-Hastings (REG:372-423): NaN -> accept (Q8), overflow -> 1; random walk: no proposal ratio
            const float logalpha = (lik_prop - lik) + (prior_prop - prior_cur) + 0.0f;
            const float mh = (logalpha != logalpha) ? 1.0f : fminf(1.0f, expf_fast(logalpha));
            const bool accept = uni_i((u < mh) ? 1 : 0) != 0;
            const int acc_before = nacc;
            if (accept) {
                nacc += 1;
                lik = uni_f(lik_prop); prior_cur = uni_f(prior_prop);
                rec_rmse_tr = uni_f(rc[2]); rec_rmse_te = uni_f(rc[3]); rec_acc_tr = uni_f(rc[4]); rec_acc_te = uni_f(rc[5]);
                accmask |= 1u << l;
            }
            if (l == g) {                                    // group l writes the trace row of step i + l
                my_mask = accmask; my_acc_before = acc_before; my_likprop = lik_prop; my_logalpha = logalpha;
                my_s0 = rec_rmse_tr; my_s1 = rec_rmse_te; my_s2 = rec_acc_tr; my_s3 = rec_acc_te;
            }
            nd = 2 * nd + (accept ? 1 : 0);
        }
        STAMP(6);
        STAMP(6);
        // 7. the new state, rebuilt from the tapes, and the trace rows (row of step i + l: the recorded vector after that
        //    step's decision = the state after it if anything was accepted up to there, else the old recorded vector)
        const bool write_row = g < dr;
        float* prow = nullptr;
        if (write_row) {
            const size_t tpos = trow + (size_t)((i + g + 1) % p.trace_cap);
            prow = p.tr_pos_w + tpos * (size_t)p.PW;
            if (tid == 0)
                store_trace_row(p.tr_scal + tpos * TR_COUNT, (TASK == TASK_REG) ? my_likprop : my_likprop * adapttemp, my_s0, my_s1, my_s2,
                                my_s3, my_acc_before, my_logalpha);
        }
        for (int j = tid; j < p.PW; j += nthr) {
            float rowv = 0.0f;
            if (j < P) {
                float v = w_cur[j];
                float vrow = v;
                for (int l = 0; l < dr; ++l) {
                    if ((accmask >> l) & 1u) v = fmaf(p.step_w, tapes[(size_t)l * (PS + 8) + j], v);
                    if (l == g) vrow = v;
                }
                rowv = my_mask ? vrow : rec_w[j];
                if (accmask) { w_cur[j] = v; rec_w[j] = v; }
            }
            if (write_row) prow[j] = rowv;
        }
        __syncthreads();                                    // the next round's tapes and proposals read what was just written
        STAMP(7);
        PTNN_DIAG(count_round);
        i += dr;
        epoch += 1;
        par ^= 1;
        if (ahead) tpar ^= 1;
        if (swap_inside && i == ho_next + 1 && ho_next < step_end) {
            // ---- the swap round of this hand-off (REG:427-437 <-> 719-752), inside the launch.  The ROOT group of every replica
            // posts its scalar (Q11) and its state row as granules (agent scope: the other replicas sit on other XCDs), reads all R
            // scalars, computes the cascade (the code swap_kernel runs: cascade_lds on the same uniforms), fetches the row of its
            // source replica and hands {moved?, new state} to its own siblings -- through the XCD's L2 when they share it.  The
            // likelihood and prior stay the ones of the state that left (Q12).  Granules are two-deep by round parity: a root posts
            // round k + 2 only after it has read every scalar of round k + 1, which their owners post after reading round k.
            const int Rg = pp->sp.R;
            const unsigned xtag = dyn.epoch_base + (unsigned)nx + 1u;
            granule_t* const xl = p.xswap + (size_t)(nx & 1) * swap_xchg_granules(Rg, PS);
            granule_t* const xst = xl + ((Rg + 7) & ~7);
            granule_t* const xsb = xst + (size_t)Rg * PS + (size_t)gid * (PS + 8);
            auto gstore = [&](granule_t* g_, float v_) { if (xcd_local) granule_store_xcd(g_, xtag, v_); else granule_store(g_, xtag, v_); };
            auto gwait = [&](const granule_t* g_, float& v_) { return xcd_local ? granule_wait_xcd(g_, xtag, v_) : granule_wait(g_, xtag, v_); };
            bool ok = true;
            if (g == 0) {
                if (tid == 0) granule_store(xl + gid, xtag, (TASK == TASK_REG) ? lik * T : lik);
                for (int j = tid; j < PS; j += nthr) granule_store(xst + (size_t)gid * PS + j, xtag, (j == P) ? eta : ((j < P) ? w_cur[j] : 0.0f));
                float* const sL = recs;
                float* const sU = recs + Rg;
                int* const sSrc = reinterpret_cast<int*>(recs + 2 * Rg);
                for (int k = tid; k < Rg; k += nthr) { float v = 0.0f; ok = granule_wait(xl + k, xtag, v) && ok; sL[k] = v; }
                if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
#if defined(__HIP_DEVICE_COMPILE__)
                const SwapParams sp = pp->sp;
#else
                const SwapParams sp{};
#endif
                const int round = pp->round0 + nx;
                const int nsw = cascade_lds(sp, round, sL, sU, sSrc, true);
                const int src = sSrc[gid];
                if (gid == sp.first_global) {                // replica 0's root keeps the books (swap_block: b == 0)
                    if (sp.src_log && round < sp.log_capacity)
                        for (int k = tid; k < Rg; k += nthr) sp.src_log[(size_t)round * Rg + k] = sSrc[k];
                    if (tid == 0) { sp.counters[0] += nsw; sp.counters[1] += Rg - 1; }
                }
                __syncthreads();                            // sSrc has been read: the record area is free again
                if (src != gid) {
                    for (int j = tid; j < PS; j += nthr) { float v = 0.0f; ok = granule_wait(xst + (size_t)src * PS + j, xtag, v) && ok; w_cur[j] = v; }
                    if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
                    for (int j = tid; j < PS; j += nthr) gstore(xsb + 8 + j, w_cur[j]);
                }
                if (tid == 0) gstore(xsb, (src != gid) ? 1.0f : 0.0f);
            } else {
                if (tid == 0) { float mv = 0.0f; ok = gwait(xsb, mv); red[0] = mv; }
                if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
                if (red[0] != 0.0f) {
                    for (int j = tid; j < PS; j += nthr) { float v = 0.0f; ok = gwait(xsb + 8 + j, v) && ok; w_cur[j] = v; }
                    if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
                }
            }
            __syncthreads();
            if (TASK == TASK_REG) eta = uni_f(w_cur[P]);    // eta travels with the state (REG:436-437)
            nx += 1;
            ho_next = next_handoff(i);
            // the tapes drawn ahead for the next round were cut at the hand-off like this one: nothing to redo
        }
    }
    PTNN_DIAG(tree_flush);
    if (failed) {
        if (tid == 0) atomicAdd(p.error_flag, 1);           // a bounded spin ran out: the host reports it
        return;
    }
    if (g == 0) {
        float* const gw_end = swap_inside ? pp->state[(pp->flip0 + nx) & 1] + (size_t)r * PS : gw;     // every in-launch round flips the host's buffers
        for (int j = tid; j < PS; j += nthr) {
            gw_end[j] = (j == P) ? eta : w_cur[j];
            p.rec_w[(size_t)r * PS + j] = rec_w[j];
        }
        if (tid == 0) {
            sf[SF_LIK] = lik; sf[SF_PRIOR] = prior_cur;
            sf[SF_REC_RMSE_TR] = rec_rmse_tr; sf[SF_REC_RMSE_TE] = rec_rmse_te;
            sf[SF_REC_ACC_TR] = rec_acc_tr; sf[SF_REC_ACC_TE] = rec_acc_te;
            si[SI_NACC] = nacc;
            if (step_begin == 0) { sf[SF_TAU_LAST] = eta; si[SI_LG_COUNT] = 0; dyn.gd_valid[r] = 0; }   // what the other schedules leave
            p.L_handoff[gid] = (TASK == TASK_REG) ? lik * T : lik;      // Q11
            p.L_final[gid] = lik;
            post_raw(p, gid, lik, prior_cur, T, step_begin + n_steps - 1);
        }
    }
}
