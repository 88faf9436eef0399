// ptnn_dev_kernels.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// model_kernel, persistent_loop, the __global__ segment kernels, the per-shape table.

// ------------------------------------------------------------------------------------------------
// stand-alone model functions (same device code): mode 0 = evaluate, 1 = langevin_gradient, 2 = tape
// ------------------------------------------------------------------------------------------------
template <int TASK, int I, int O>
__global__ void __launch_bounds__(MAX_THREADS) model_kernel(const SegParams p, const int mode, const float* __restrict__ w_in,
                             const float* __restrict__ tau_sq, float* __restrict__ out, int a0, int a1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int Nall = p.Ntr + p.Nte;
    const Lds l = carve(smem, Nall, p.IPY, p.PS, p.H, p.FWS);
    if (mode == 2) {
        tape_step(p, a0, a1, l.noise, l.scal);
        __syncthreads();
        for (int j = tid; j < p.P; j += nthr) out[j] = l.noise[j];
        if (tid < 3) out[p.P + tid] = l.scal[tid];
        return;
    }
    {
        const float4* src = reinterpret_cast<const float4*>(p.data);
        float4* dst = reinterpret_cast<float4*>(l.xy);
        for (int e = tid; e < ((Nall + 2) * p.IPY) >> 2; e += nthr) dst[e] = src[e];
    }
    for (int j = tid; j < p.P; j += nthr) l.w_cur[j] = w_in[(size_t)(mode == 4 ? 0 : b) * p.P + j];
    __syncthreads();
    if (mode == 4) {
        // The two things a round of the prefetching tree cannot do without, timed with the constant-rate counter (ptnn_time_tree_round;
        // bench.py's roofline.tree): (a) block 0: a0 times what a node does between its proposal and its record -- the packed
        // forward image, the forward pass over all rows, the likelihood and the prior's sum of squares in the same reduction;
        // (b) blocks 0 and 8 -- one XCD under the round-robin dispatch -- a0 round trips of one granule through the path the
        // tree's records take (a1 = 1: through the XCD's L2 when both blocks report the same XCC id, else agent scope).
        granule_t* const ga = reinterpret_cast<granule_t*>(out + 16);
        granule_t* const gb = ga + 8;                          // its own 64-byte line
        if (b == 0) {
            unsigned long long t_fw = 0, t_rt = 0;
            float keep = 0.0f;
            const unsigned long long t0 = wall_clock64();
            for (int rep = 0; rep < a0; ++rep) {
                build_fw<I, O>(l.w_cur, l.fw, p.H, p.FWS);
                __syncthreads();
                float ssq = 0.0f;
                for (int j = tid; j < p.P; j += nthr) ssq = fmaf(l.w_cur[j], l.w_cur[j], ssq);
                const EvalSums es = eval_rows<TASK, I, O, false, true>(l.fw, l.xy, p.IPY, p.FWS, p.H, p.Ntr, Nall, l.red, ssq);
                keep += finish_loglik<TASK>(es, p.Ntr, 0.0f) + prior_value<TASK>(p, ssq, 0.0f);
                __syncthreads();
            }
            t_fw = wall_clock64() - t0;
            int ok = 1, local = 0;
            if (tid == 0) {
                float v = 0.0f;
                granule_store(ga, 0x51000000u, (float)xcc_id());
                ok = granule_wait(gb, 0x51000000u, v) ? 1 : 0;
                local = (a1 != 0 && ok && (int)v == xcc_id()) ? 1 : 0;
                const unsigned long long t1 = wall_clock64();
                for (int k = 1; k <= a0 && ok; ++k) {
                    if (local) { granule_store_xcd(ga, 0x51000000u + k, 1.0f); ok = granule_wait_xcd(gb, 0x51000000u + k, v) ? 1 : 0; }
                    else { granule_store(ga, 0x51000000u + k, 1.0f); ok = granule_wait(gb, 0x51000000u + k, v) ? 1 : 0; }
                }
                t_rt = wall_clock64() - t1;
                out[0] = __uint_as_float((unsigned)(t_fw & 0xffffffffull)); out[1] = __uint_as_float((unsigned)(t_fw >> 32));
                out[2] = __uint_as_float((unsigned)(t_rt & 0xffffffffull)); out[3] = __uint_as_float((unsigned)(t_rt >> 32));
                out[4] = (float)ok; out[5] = (float)local; out[6] = keep;
            }
        } else if (b == 8 && tid == 0) {
            float v = 0.0f;
            int ok = granule_wait(ga, 0x51000000u, v) ? 1 : 0;
            const int local = (a1 != 0 && ok && (int)v == xcc_id()) ? 1 : 0;
            granule_store(gb, 0x51000000u, (float)xcc_id());
            for (int k = 1; k <= a0 && ok; ++k) {
                if (local) { ok = granule_wait_xcd(ga, 0x51000000u + k, v) ? 1 : 0; granule_store_xcd(gb, 0x51000000u + k, 1.0f); }
                else { ok = granule_wait(ga, 0x51000000u + k, v) ? 1 : 0; granule_store(gb, 0x51000000u + k, 1.0f); }
            }
        }
        return;
    }
    if (mode == 1) {
        if (tid < WAVE) sgd_sweep_dispatch<TASK, I, O>(l.w_cur, l.w_gd, l.xy, p.data, p.Ntr, p.H, p.lr);
        __syncthreads();
        for (int j = tid; j < p.P; j += nthr) out[(size_t)b * p.P + j] = l.w_gd[j];
        return;
    }
    if (mode == 3) {
        // a0 SGD epochs back to back on wave 0, timed with the constant-rate counter (s_memrealtime): what ONE sequential epoch of
        // this net on this data costs on this device -- the unit of the dependent-chain floor bench.py reports (ptnn_time_sgd_epoch)
        unsigned long long ticks = 0;
        if (tid < WAVE) {
            const unsigned long long t0 = wall_clock64();
            for (int rep = 0; rep < a0; ++rep) {
                sgd_sweep_dispatch<TASK, I, O>(l.w_cur, l.w_gd, l.xy, p.data, p.Ntr, p.H, p.lr);
                gsync<true>();
            }
            ticks = wall_clock64() - t0;
        }
        if (tid == 0) { out[0] = __uint_as_float((unsigned)(ticks & 0xffffffffull)); out[1] = __uint_as_float((unsigned)(ticks >> 32)); out[2] = l.w_gd[0]; }
        return;
    }
    build_fw<I, O>(l.w_cur, l.fw, p.H, p.FWS);
    __syncthreads();
    const EvalSums s = eval_rows<TASK, I, O>(l.fw, l.xy, p.IPY, p.FWS, p.H, p.Ntr, Nall, l.red);
    const float eta = (TASK == TASK_REG) ? logf_fast(tau_sq[b]) : 0.0f;
    float ll, r1, r2, a_tr, a_te;
    finish_eval<TASK>(s, p.Ntr, p.Nte, eta, ll, r1, r2, a_tr, a_te);
    const float ss = block_sumsq(l.w_cur, p.P, l.red);
    const float pr = prior_value<TASK>(p, ss, eta);
    float ll_te = s.a_te;
    if (TASK == TASK_REG) ll_te = -0.5f * (float)p.Nte * (LOG_2PI + eta) - 0.5f * s.a_te * expf_fast(-eta);
    if (tid == 0) {
        float* o = out + (size_t)b * 8;
        o[0] = ll; o[1] = r1; o[2] = r2; o[3] = a_tr; o[4] = a_te; o[5] = pr; o[6] = ll_te; o[7] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// One launch for a whole run (the parent's round loop REG:719-752 inside the kernel).  Every segment
// kernel runs the MH steps [step_begin, pp.end): swap interval after swap interval, and between two of them -- when pp.swap_inside
// -- the swap round itself: a grid-wide barrier (every replica has posted its scalar and written its state row back), the cascade
// + this replica's row move by the work-group that owns the replica (swap_block: the same code swap_kernel runs), a second
// barrier (the other work-groups of a replica re-stage the moved row), and the next interval.  All R x G work-groups must be
// resident (the host checks the occupancy of THIS kernel and otherwise launches one interval at a time with pp.swap_inside = 0,
// pp.end = the end of the interval, followed by swap_kernel: the round-2 shape); the barrier spins are bounded like every other
// cross-work-group wait and a timeout surfaces through the error flag.  Identical to the per-interval launches because an interval
// runs the same body from the same global state and the round runs the same swap_block -- under ONE timing invariant where a round
// has a single barrier (one work-group per replica): the posted scalars (L_handoff / L_final / L_raw) are single-buffered, every
// work-group copies all of them into LDS right after the barrier (cascade_lds), and the next write to any of them is a whole swap
// interval away.  The host only takes this shape for intervals of 8 MH steps or more (ptnn.hip: resolve_persistent).
// ------------------------------------------------------------------------------------------------
// Grid barrier without a read-modify-write: every work-group owns one slot and stores the phase it has reached (distinct
// addresses: nothing serialises -- 256 agent-scope atomic adds on ONE counter cost more than the launch boundary this replaces),
// wave 0 polls all slots (lane = slot) until every one has reached the phase.  Bounded like every cross-work-group wait.
__device__ __forceinline__ bool grid_barrier(unsigned* slots, int nblocks, unsigned phase, int* error_flag) {
    __syncthreads();
    int bad = 0;
    if (threadIdx.x < WAVE) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // this work-group's rows and scalars, visible on every XCD
            __hip_atomic_store(slots + blockIdx.x, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        for (;;) {
            bool here = true;
            for (int j = threadIdx.x; j < nblocks; j += WAVE)
                here = here && (__hip_atomic_load(slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase);
            if (__all(here)) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT || __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { bad = 1; break; }
        }
    }
    bad = __syncthreads_or(bad);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");              // nothing cached from before the others arrived
    return bad == 0;
}

// LOOP = false: one interval per launch and nothing else (the two schedules with several work-groups per replica and the most
// registers -- multi-CU speculative, prefetching tree -- where the loop around the body cost 30 - 40 vector registers, i.e. scratch,
// and where a persistent launch is measured to lose against the launch boundary anyway).
template <bool LOOP, class Body>
__device__ __forceinline__ void persistent_loop(const SegParams& p0, const int step_begin, Body body) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (!LOOP) {
        const SegParams& p = p0;
        persist_cptr pp = persist_args();
        SegDyn dyn;
        const int flip = pp->flip0;
        dyn.pp = pp;
        dyn.w_state = pp->state[flip]; dyn.gd_w = pp->gd[flip]; dyn.gd_valid = pp->gd_valid[flip];
        dyn.epoch_base = p.epoch_base;
        body(p, dyn, step_begin, pp->end - step_begin);
        if (p.seg_progress && blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store(p.seg_progress, p.seg_ordinal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // the interval body sees the kernel arguments through an opaque pointer, re-read every interval: nothing the body derives
    // from them (LDS carving, per-thread addresses) is hoisted out of the loop and kept alive across it -- left to itself the
    // optimiser did exactly that, at 30 - 45 vector registers per kernel
    int flip, lflip, round, cur = step_begin;
    unsigned phase = 0, epoch_add = 0;
    {
        persist_cptr pp = persist_args();
        flip = pp->flip0; lflip = pp->lflip0; round = pp->round0;
    }
    for (;;) {
        persist_cptr pp = persist_args();
        const int end = pp->end, si = pp->si;
        if (cur >= end) break;
        // the step that hands off next (Q10), as ptnn_run finds it on the host
        int seg_end;
        if (pp->task == TASK_REG) { const int c1 = cur > 1 ? cur : 1; seg_end = ((c1 + si - 1) / si) * si; }
        else seg_end = ((cur + si) / si) * si - 1;
        const bool handoff = seg_end < end;
        const int stop = handoff ? seg_end + 1 : end;
        const bool swap_inside = pp->swap_inside != 0;
#if defined(__HIP_DEVICE_COMPILE__)
        const SegParams p = *seg_args();
#else
        const SegParams p = p0;                                       // host pass of the compiler: never executed
#endif
        {
            SegDyn dyn;
            dyn.pp = pp;
            dyn.w_state = pp->state[flip]; dyn.gd_w = pp->gd[flip]; dyn.gd_valid = pp->gd_valid[flip];
            dyn.epoch_base = p.epoch_base + epoch_add;
            body(p, dyn, cur, stop - cur);
        }
        epoch_add += (unsigned)(stop - cur) + 1u;                     // granule tags never repeat across intervals
        cur = stop;
        if (!handoff || !swap_inside) break;
        pp = persist_args();
        if (!grid_barrier(pp->barrier, pp->nblocks, ++phase, p.error_flag)) { if (threadIdx.x == 0) atomicAdd(p.error_flag, 1); return; }
        if (xcd_block(p.G) % p.G == 0) {                              // the bodies' own (replica, group) of this work-group
#if defined(__HIP_DEVICE_COMPILE__)
            SwapParams sp = pp->sp;
#else
            SwapParams sp{};
#endif
            sp.cur = pp->state[flip]; sp.next = pp->state[flip ^ 1];
            sp.gd_cur = pp->gd[flip]; sp.gd_next = pp->gd[flip ^ 1];
            sp.gd_valid_cur = pp->gd_valid[flip]; sp.gd_valid_next = pp->gd_valid[flip ^ 1];
            sp.label_cur = pp->label[lflip]; sp.slot_cur = pp->slot_of[lflip];
            sp.label_next = pp->label[lflip ^ 1]; sp.slot_next = pp->slot_of[lflip ^ 1];
            sp.canonical = (p.switch_step >= 0 && cur - 1 >= p.switch_step) ? 1 : 0;
            swap_block(sp, round, 3, xcd_block(p.G) / p.G, smem);
        }
        if (pp->sp.label_mode) lflip ^= 1; else flip ^= 1;
        round += 1;
        // the other work-groups of a replica re-stage the row its owner has just moved; a replica of one work-group reads its own
        // writes (same CU, write-through L1) and needs no second rendezvous
        if (p.G > 1 && !grid_barrier(pp->barrier, pp->nblocks, ++phase, p.error_flag)) { if (threadIdx.x == 0) atomicAdd(p.error_flag, 1); return; }
        if (p.G == 1) __syncthreads();
    }
    if (p0.seg_progress && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(p0.seg_progress, p0.seg_ordinal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Which kernels carry the interval loop.  The loop costs 10 - 40 vector registers (values the optimiser keeps alive around the
// body); it is compiled in where the kernel stays free of scratch with it -- every shape of the BASELINE configurations that
// takes one work-group per replica -- and left out where it would add spills (many-class heads, wide-input packed nets); those
// run one launch per interval as before (Shape::loops tells the host).
template <int TASK, int I, int O> constexpr bool coop_has_loop() { return O <= 3; }
template <int TASK, int I, int O> constexpr bool pack_has_loop() { return TASK == TASK_REG && I <= 8; }

template <int TASK, int I, int O>
__global__ void __launch_bounds__(MAX_THREADS) segment_kernel(const SegParams p, const PersistParams pp, const int step_begin) {
    persistent_loop<coop_has_loop<TASK, I, O>()>(p, step_begin, [](const SegParams& q, const SegDyn& d, int b, int n) { segment_body<TASK, I, O>(q, d, b, n); });
}
template <int TASK, int I, int O>
__global__ void __launch_bounds__(MAX_THREADS) segment_spec_kernel(const SegParams p, const PersistParams pp, const int step_begin) {
    persistent_loop<false>(p, step_begin, [](const SegParams& q, const SegDyn& d, int b, int n) { segment_spec_body<TASK, I, O>(q, d, b, n); });
}
template <int TASK, int I, int O>
__global__ void __launch_bounds__(PK_WAVES * WAVE) segment_pack_kernel(const SegParams p, const PersistParams pp, const int step_begin) {
    persistent_loop<pack_has_loop<TASK, I, O>()>(p, step_begin, [](const SegParams& q, const SegDyn& d, int b, int n) {
        if (q.pk_nred == 4) segment_pack_body<TASK, I, O, 4>(q, d, b, n);
        else segment_pack_body<TASK, I, O, 3>(q, d, b, n);
    });
}
// the packed schedule over several CUs per replica (16-lane groups: 9 <= n_hidden <= 16): its own kernel, so that the one-CU kernel
// (the benchmark's) keeps its registers and its code
template <int TASK, int I, int O>
__global__ void __launch_bounds__(PK_WAVES * WAVE) segment_packm_kernel(const SegParams p, const PersistParams pp, const int step_begin) {
    persistent_loop<false>(p, step_begin, [](const SegParams& q, const SegDyn& d, int b, int n) {
        if (q.pk_nred == 4) segment_pack_body<TASK, I, O, 4, true>(q, d, b, n);
        else segment_pack_body<TASK, I, O, 3, true>(q, d, b, n);
    });
}
template <int TASK, int I, int O>
__global__ void __launch_bounds__(MAX_THREADS) segment_tree_kernel(const SegParams p, const PersistParams pp, const int step_begin) {
    persistent_loop<false>(p, step_begin, [](const SegParams& q, const SegDyn& d, int b, int n) { segment_tree_body<TASK, I, O>(q, d, b, n); });
}
#pragma clang fp contract(fast)     // the wide-net section's policy (see above model_wide_kernel)
template <int TASK, int I, int O>
__global__ void __launch_bounds__(MAX_THREADS) segment_wide_kernel(const SegParams p, const PersistParams pp, const int step_begin) {
    persistent_loop<true>(p, step_begin, [](const SegParams& q, const SegDyn& d, int b, int n) { segment_wide_body<TASK, I, O, false>(q, d, b, n); });
}
// state and proposal resident in LDS (host: matrix-core layout and 2 vectors + scratch fit in 160 KB)
template <int TASK, int I, int O>
__global__ void __launch_bounds__(MAX_THREADS) segment_wide_res_kernel(const SegParams p, const PersistParams pp, const int step_begin) {
    persistent_loop<true>(p, step_begin, [](const SegParams& q, const SegDyn& d, int b, int n) { segment_wide_body<TASK, I, O, true>(q, d, b, n); });
}
#pragma clang fp contract(off)
