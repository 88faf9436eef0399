// ptnn_dev_wide.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// wide nets (64 < H <= 512): sgd_sweep_wide, matrix-core forward, segment_wide_body, model_wide_kernel; then swap_block and the non-template kernels.

// The wide-net section is compared with nothing but itself and the float64 oracle: here the compiler may fuse as it likes
// (5 % on config 5); the policy of the top of the file returns after model_wide_kernel.
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
// Wide hidden layers (64 < H <= 512, e.g. BASELINE config 5: 32-512-1, P = 17 409): one THREAD per hidden unit over up to
// 8 waves of the work-group.  The per-replica vectors (w, proposal, SGD results, noise: 70 KB each) no longer fit in LDS
// next to each other, so they live in HBM/L2 and are streamed with coalesced accesses; LDS holds the packed forward image of
// the proposal and the reduction scratch; the data set is read through the scalar cache (wave-uniform rows) in the sweep
// and through L2 in the forward pass.  All waves of a work-group share one MH step; several work-groups per replica speculate over steps (segment_wide_kernel).
// ------------------------------------------------------------------------------------------------
// res: the current state vector is kept in LDS next to the proposal (matrix-core layout only: there the flat image IS the proposal)
__host__ __device__ inline size_t wide_lds_floats(int H, int FWS, int O, int PS, bool res = false) {
    const size_t img = (fw_floats(H, FWS) > (size_t)PS) ? fw_floats(H, FWS) : (size_t)PS;   // packed or flat image
    return img + (res ? (size_t)PS : 0) + MAX_WAVES * 8 + 4 * MAX_WAVES * (size_t)((O + 3) & ~3) + 16 + 6 * 16;   // partial sums of up to two epochs; + per-slot scalars of a window (WIDE_WINDOW = 16)
}
__host__ __device__ inline size_t wide_img_floats(int H, int FWS, int PS) {
    return (fw_floats(H, FWS) > (size_t)PS) ? fw_floats(H, FWS) : (size_t)PS;
}

// R5 for H > 64: thread h owns hidden unit h; the output pre-activation is a two-level sum (DPP inside the wave, then the
// per-wave partials through LDS, summed in a fixed order by every thread); ONE work-group barrier per data row (the
// partial buffers alternate between rows).
//
// A row costs what its dependent chain costs: z -> sigmoid -> hid W2 -> wave sum -> LDS -> barrier -> sum of the partials ->
// sigmoid -> deltas -> lhd.  The 2 I multiply-adds of a row (the W1 update and the next row's x . W1) are kept OFF that chain,
// the way the narrow sweep does it (deferred update): the update of row n-1 is applied during row n, and the pre-activation
// of row n+1 starts from the weights of row n-1,
//     z[n+1] = (x[n+1] . W1[n-1] - B1[n-1]) + lhd[n] (1 + x[n+1] . x[n]),
// the last factor being column I+1 of the data image.  Both run as packed v_pk_fma_f32 on input PAIRS while the wave waits
// for the reduction and the barrier of row n.  The rows are wave-uniform and come through the scalar cache (s_load from the
// constant address space, a row ahead), so the data values are SGPR operands of the multiply-adds: no vector loads, no
// register copies (the plain chain spent 136 issue slots per row: 32 dependent v_fmac, 16 v_pk_fma, 16 v_mov_b64 of row
// buffers, 10 flat loads).  Called out of line: inlined twice into the segment kernel next to the two MFMA forward variants
// it pushed 312 VGPRs of the kernel into scratch (1236 B per lane for the 32-H-1 shape).
// w_ref (optional): returns this thread's share of |w_ref - w_out|^2, summed from the registers the result is written from -- the
// first term of the Langevin proposal ratio (REG:336-340) without reading the 70 KB result back.
//
// NE = 2: TWO independent epochs (two Langevin proposals of one speculative window, both made on the assumption that the steps
// before them reject) run through the SAME row loop: one barrier, one set of data rows in scalar registers and one trip through
// the reduction latency per row serve both; every epoch performs exactly the operations of the NE = 1 code in the same order.
// MEASURED AND NOT USED by the sampler: a pair costs 1.69 x one epoch (ptnn_time_sgd_epoch: 436 us vs 737 us for 32-512-1; with
// two waves per SIMD the row loop is mostly issue-bound, not latency-bound), and pairing the Langevin steps of a window in
// segment_wide_body bought 3 % against 4 % lost to the extra registers of the step loop (profiles/r03_wide_pair.json).  Kept
// for the timer (model_wide_kernel mode 3), as the record of that experiment.
// The epoch of a Langevin PROPOSAL is needed for two things: |w - epoch(w')|^2 in the proposal ratio (REG:336-340), always, and the
// vector itself -- the cached langevin_gradient of the next state -- only if the step is accepted (1 - 5 % of the steps of a wide net).
// With a SweepDecide the sweep takes the Metropolis-Hastings decision itself, from the registers its result lives in, and writes the
// 70 KB vector out only for an accepted step: everything else the decision needs (likelihood and prior of the proposal, the step's
// uniform) is known before the epoch starts when the forward pass runs first.  Same operations in the same order as the caller's
// own decision (segment_wide_body), so the chain does not change.
struct SweepDecide {
    float base;                      // (lik_prop - lik) + (prior_prop - prior_cur)
    float d2, step_w, adapttemp, u;  // |noise|^2, proposal step, temperature of the step, its uniform
    float* red;                      // scratch of block_sum
    float logalpha;                  // out (uniform)
    bool accept;                     // out (uniform)
};

template <int TASK, int I, int O, int NE, bool DECIDE = false>
__device__ __forceinline__ void sgd_sweep_wide_n(const float* const (&w_in)[NE], float* const (&w_out)[NE], const float* __restrict__ data,
                                                 int IPY, int Ntr, int H, float lr, float* __restrict__ part, const float* w_ref,
                                                 float (&d1_out)[NE], SweepDecide* dec = nullptr) {
    constexpr float C = -LOG2E, IC = -LN2;
    constexpr int OP = (O + 3) & ~3;
    constexpr int IP = (I + 1) / 2;                                   // input pairs (an odd I is padded with a zero weight)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool act = t < H;
    const int hl = act ? t : 0;
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    const float clr = C * lr;
    f32x2 w1[NE][IP];
    float w2[NE][O], cl[NE][O], nb1[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const float* wi = w_in[e];
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            w1[e][i][0] = act ? C * wi[(2 * i) * H + hl] : 0.0f;
            w1[e][i][1] = (act && 2 * i + 1 < I) ? C * wi[(2 * i + 1) * H + hl] : 0.0f;
        }
#pragma unroll
        for (int o = 0; o < O; ++o) w2[e][o] = act ? C * wi[oW2 + hl * O + o] : 0.0f;
        nb1[e] = act ? -C * wi[oB1 + hl] : 1.0e30f;                   // -B1'; inactive threads: exponent +1e30 -> hid == 0 exactly
#pragma unroll
        for (int o = 0; o < O; ++o) cl[e][o] = -C * wi[oB2 + o];      // replicated in every thread, updated identically
    }
    int par = 0;
    static_assert(MAX_WAVES == 8, "the partial sums are read as two float4");
    for (int e = t; e < 2 * NE * MAX_WAVES * OP; e += blockDim.x) part[e] = 0.0f;
    __syncthreads();
    // rows through the scalar cache: the address is wave-uniform and the image is never written while a kernel runs
    // (a device function receives its arguments in VGPRs: the address is made scalar by hand, or the loads would be vector loads)
    const unsigned long long da = (unsigned long long)(uintptr_t)data;
    const cfloat* cdata = (const cfloat*)(uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(da >> 32)) << 32) |
                                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)da));
    const int ipy = __builtin_amdgcn_readfirstlane(IPY);
    const int ntr = __builtin_amdgcn_readfirstlane(Ntr);
    typedef __attribute__((address_space(3))) float lfloat;
    lfloat* lpart = (lfloat*)part;                                    // the partial sums live in LDS: ds_ instead of flat_ accesses
    // a whole row of input pairs into SGPRs (the pad column of an odd I meets a zero weight)
    auto load_row = [&](int n, f32x2 (&x)[IP]) {
        const cfloat* row = cdata + (size_t)n * ipy;
#pragma unroll
        for (int i = 0; i < IP; ++i) { x[i][0] = row[2 * i]; x[i][1] = row[2 * i + 1]; }
    };
    auto zpart = [&](const f32x2 (&x)[IP], int e) {                   // x . W1 - B1 with the weights as they are now
        f32x2 a0 = {nb1[e], 0.0f}, a1 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < IP; i += 2) {
            a0 = __builtin_elementwise_fma(x[i], w1[e][i], a0);
            if (i + 1 < IP) a1 = __builtin_elementwise_fma(x[i + 1], w1[e][i + 1], a1);
        }
        const f32x2 s_ = a0 + a1;
        return s_[0] + s_[1];
    };
    // Two rows live in scalar registers: xu = the row whose update is pending (row n-1 during iteration n), xz = the row whose
    // pre-activation is started next (row n+1).  Both are consumed BEFORE the barrier of an iteration and reloaded right AFTER
    // it (into the same registers: the old rows are dead by then), a whole iteration ahead of the next barrier -- the only place
    // that has to wait for them, because the LDS traffic of the reduction shares the scalar loads' counter (lgkmcnt).
    f32x2 xu[IP], xz[IP];
    load_row(0, xz);
    float lhd_p[NE], zp[NE];                                          // lhd of the previous row: its update is still pending
#pragma unroll
    for (int e = 0; e < NE; ++e) { lhd_p[e] = 0.0f; zp[e] = zpart(xz, e); }
    load_row(0, xu);
    load_row(1, xz);
    // the row loop is one latency-bound dependent chain with three synchronisation points per row: where its head falls in an
    // instruction-cache line decided 8 % of the epoch (472 vs 512 us for the same instructions), so it is pinned
    // (the pin sits some forty instructions ahead of the loop's top, which the compiler places: the top of the deciding variant
    // comes out 8 bytes earlier than the plain one's -- 24 instead of 32 bytes into its 64-byte line, 4 % of the epoch -- and is put
    // back where the plain one's is; checked in the code object: llvm-objdump, the s_or_b64 behind the loop's entry branch)
    if constexpr (DECIDE) asm volatile(".p2align 8\n\ts_nop 0\n\ts_nop 0");
    else asm volatile(".p2align 8");
    for (int n = 0; n < ntr; ++n) {
        const cfloat* row = cdata + (size_t)n * ipy;
        const float yn = row[I], dn = row[I + 1];
        float hid[NE], ldh[NE];
        lfloat* mypart = lpart + par * NE * MAX_WAVES * OP;           // [epoch][o][wave]: the partials of one output are contiguous
        // stage by stage over the epochs (NE = 2): the wave issues in order, so the second epoch's instruction of a stage fills the
        // latency of the first one's -- for NE = 1 this is the plain sequence
        float ex[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float z = fmaf(lhd_p[e], dn, zp[e]);                // + lhd[n-1] (1 + x[n] . x[n-1])   (row 0: lhd_p == 0)
            ex[e] = __builtin_amdgcn_exp2f(z);
        }
        // off the chain: apply the update of row n-1 (xu; a no-op for n = 0, where lhd_p = 0), then start row n+1 (xz) from the
        // updated weights; two zero rows follow the image, so the look-ahead never leaves it
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const f32x2 l2 = {lhd_p[e], lhd_p[e]};
#pragma unroll
            for (int i = 0; i < IP; ++i) w1[e][i] = __builtin_elementwise_fma(l2, xu[i], w1[e][i]);
            if constexpr (I & 1) w1[e][IP - 1][1] = 0.0f;             // the pad weight of an odd input count stays zero
            nb1[e] += lhd_p[e];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) zp[e] = zpart(xz, e);
#pragma unroll
        for (int e = 0; e < NE; ++e) hid[e] = __builtin_amdgcn_rcpf(1.0f + ex[e]);
#pragma unroll
        for (int e = 0; e < NE; ++e) ldh[e] = lr * fmaf(-hid[e], hid[e], hid[e]);
        {
            float sums[NE * O];
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int o = 0; o < O; ++o) sums[e * O + o] = hid[e] * w2[e][o];
            wave_allsum_n<NE * O>(sums);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < NE * O; ++q) mypart[q * MAX_WAVES + wave] = sums[q];
            }
        }
        __syncthreads();
        // the partials first, THEN the scalar loads: a wait for the LDS reads is a wait for everything on lgkmcnt
        float4 pa[NE][O], pb[NE][O];
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const volatile lfloat* pq = mypart + (e * O + o) * MAX_WAVES;   // volatile: the reads stay above the wait below
                pa[e][o] = make_float4(pq[0], pq[1], pq[2], pq[3]);
                pb[e][o] = make_float4(pq[4], pq[5], pq[6], pq[7]);
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): the partials are here
        __builtin_amdgcn_sched_barrier(0);
        load_row(n, xu);                                              // pending update of the next iteration
        load_row(n + 2, xz);                                          // pre-activation started in the next iteration
        __builtin_amdgcn_sched_barrier(0);
        float zo[NE][O], eo[NE][O], out[NE][O];
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int o = 0; o < O; ++o)       // all MAX_WAVES partials (entries of absent waves are zero), summed in a fixed order
                zo[e][o] = cl[e][o] + (((pa[e][o].x + pa[e][o].y) + (pa[e][o].z + pa[e][o].w)) + ((pb[e][o].x + pb[e][o].y) + (pb[e][o].z + pb[e][o].w)));
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int o = 0; o < O; ++o) eo[e][o] = __builtin_amdgcn_exp2f(zo[e][o]);
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int o = 0; o < O; ++o) out[e][o] = __builtin_amdgcn_rcpf(1.0f + eo[e][o]);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            float g = 0.0f;
            float lod[O];
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float tt;
                if (TASK == TASK_CLS) tt = ((int)yn == o) ? 1.0f : 0.0f;
                else tt = yn;
                const float od = (tt - out[e][o]) * fmaf(-out[e][o], out[e][o], out[e][o]);
                g = fmaf(od, w2[e][o], g);                            // pre-update W2 (Q4)
                lod[o] = clr * od;
            }
            lhd_p[e] = g * ldh[e];
#pragma unroll
            for (int o = 0; o < O; ++o) {
                w2[e][o] = fmaf(lod[o], hid[e], w2[e][o]);
                cl[e][o] += lod[o];
            }
        }
        par ^= 1;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (ntr > 0) {                                                // the update of the last row is still pending (xu = row ntr-1)
            const f32x2 l2 = {lhd_p[e], lhd_p[e]};
#pragma unroll
            for (int i = 0; i < IP; ++i) w1[e][i] = __builtin_elementwise_fma(l2, xu[i], w1[e][i]);
            nb1[e] += lhd_p[e];
        }
        float d1 = 0.0f;
        float* wo = w_out[e];
        // every element of the result, in the order the squared distance is summed in: f(index, value)
        auto each = [&](auto f) {
            if (act) {
#pragma unroll
                for (int i = 0; i < I; ++i) f(i * H + t, IC * w1[e][i >> 1][i & 1]);
#pragma unroll
                for (int o = 0; o < O; ++o) f(oW2 + t * O + o, IC * w2[e][o]);
                f(oB1 + t, -IC * nb1[e]);
            }
            if (t == 0) {
#pragma unroll
                for (int o = 0; o < O; ++o) f(oB2 + o, -IC * cl[e][o]);
            }
        };
        if (dec == nullptr) {
            each([&](int idx, float v) {
                wo[idx] = v;
                if (w_ref) { const float d = w_ref[idx] - v; d1 = fmaf(d, d, d1); }
            });
        } else {
            // (in groups of eight: left to itself the scheduler forms all 34 addresses and issues all 34 loads at once -- 248 VGPRs
            // clobbered by this function instead of ~100, and the kernel spills what it holds around the call)
            int k8 = 0;
            each([&](int idx, float v) {
                const float d = w_ref[idx] - v;
                d1 = fmaf(d, d, d1);
                if ((++k8 & 7) == 0) __builtin_amdgcn_sched_barrier(0);
            });
            const float d1_all = block_sum(d1, dec->red);
            const float diff_prop = langevin_ratio(d1_all, dec->d2, dec->step_w, dec->adapttemp);
            dec->logalpha = dec->base + diff_prop;
            const float mh = (dec->logalpha != dec->logalpha) ? 1.0f : fminf(1.0f, expf_fast(dec->logalpha));
            dec->accept = dec->u < mh;
            if (dec->accept) {
                // the destination as a value the compiler knows nothing about: the addresses of this (rare) pass are formed here, not
                // ahead of the reduction above, where 34 of them per thread stayed live across it (248 VGPRs clobbered by this
                // function instead of ~100, and the kernel spilled everything it holds around the call)
                float* wo2 = wo;
                asm volatile("" : "+v"(wo2));
                int j8 = 0;
                each([&](int idx, float v) {
                    wo2[idx] = v;
                    if ((++j8 & 7) == 0) __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
        d1_out[e] = d1;
    }
    __syncthreads();
}

template <int TASK, int I, int O>
__device__ __attribute__((noinline, aligned(256))) float sgd_sweep_wide(const float* __restrict__ w_in, float* __restrict__ w_out,
                                                          const float* __restrict__ data, int IPY, int Ntr, int H, float lr,
                                                          float* __restrict__ part, const float* w_ref = nullptr) {
    const float* const wi[1] = {w_in};
    float* const wo[1] = {w_out};
    float d1[1];
    sgd_sweep_wide_n<TASK, I, O, 1>(wi, wo, data, IPY, Ntr, H, lr, part, w_ref, d1);
    return d1[0];
}
// the epoch of a Langevin proposal that decides its own step (SweepDecide): everything by value, the verdict in registers
struct SweepVerdict { float logalpha; int accept; };
template <int TASK, int I, int O>
__device__ __attribute__((noinline, aligned(256))) SweepVerdict sgd_sweep_wide_decide(const float* __restrict__ w_in, float* __restrict__ w_out,
                                                                       const float* __restrict__ data, int IPY, int Ntr, int H, float lr,
                                                                       float* __restrict__ part, const float* w_ref, float* red,
                                                                       float base, float d2, float step_w, float adapttemp, float u) {
    const float* const wi[1] = {w_in};
    float* const wo[1] = {w_out};
    float d1[1];
    SweepDecide dec;
    dec.base = base; dec.d2 = d2; dec.step_w = step_w; dec.adapttemp = adapttemp; dec.u = u; dec.red = red;
    dec.logalpha = 0.0f; dec.accept = false;
    sgd_sweep_wide_n<TASK, I, O, 1, true>(wi, wo, data, IPY, Ntr, H, lr, part, w_ref, d1, &dec);
    return SweepVerdict{dec.logalpha, dec.accept ? 1 : 0};
}
// the pair: epoch A from w_inA (the proposal in LDS), epoch B from w_inB (a proposal parked in global memory)
template <int TASK, int I, int O>
__device__ __attribute__((noinline, aligned(256))) void sgd_sweep_wide_pair(const float* w_inA, const float* w_inB, float* w_outA, float* w_outB,
                                                                              const float* __restrict__ data, int IPY, int Ntr, int H, float lr,
                                                                              float* __restrict__ part, const float* w_ref, float* d1_ab) {
    const float* const wi[2] = {w_inA, w_inB};
    float* const wo[2] = {w_outA, w_outB};
    float d1[2];
    sgd_sweep_wide_n<TASK, I, O, 2>(wi, wo, data, IPY, Ntr, H, lr, part, w_ref, d1);
    d1_ab[0] = d1[0]; d1_ab[1] = d1[1];
}

// ------------------------------------------------------------------------------------------------
// R2/R3/R6 for wide nets on the matrix cores.  The product is taken transposed, Z^T[h][n] = sum_i W1[i][h] X[n][i], so
// that in the 32x32 accumulator tile a LANE is a data row (column n = lane & 31) and the 16 REGISTERS are hidden units
// (h = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): bias, sigmoid and the product with W2 are applied in place and the sum
// over hidden units stays in the lane across all H/32 tiles; one v_permlane32_swap at the end joins the two lane halves.
// Operands: A[h][k] = W1[k][h] is read from the flat copy of w in LDS (lanes = consecutive h: conflict-free), B[k][n] from
// the transposed data image Xt[k][n] in L2 (lanes = consecutive rows: coalesced), 2 k-values per v_mfma_f32_32x32x2_f32
// (exact fp32, k-ordered fma chain) or 16 per v_mfma_f32_32x32x16_bf16 (BF16 = true: operands rounded to bf16,
// fp32 accumulation; the tolerance study of BASELINE config 5).  Needs H % 32 == 0; I is zero-padded to IK.
// ------------------------------------------------------------------------------------------------

template <int TASK, int I, int O, bool BF16>
__device__ __forceinline__ EvalSums eval_rows_mfma(const float* __restrict__ wl, const float* __restrict__ xt,
                                                   const float* __restrict__ data, int IPY, int H, int Ntr, int Nall,
                                                   int Npad, float* __restrict__ red) {
    constexpr int IK = BF16 ? ((I + 15) & ~15) : ((I + 1) & ~1);      // k extent actually multiplied
    constexpr int KS = BF16 ? IK / 16 : IK / 2;                        // MFMA instructions per tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    float a_tr = 0.f, b_tr = 0.f, c_tr = 0.f, a_te = 0.f, b_te = 0.f, c_te = 0.f;
    float b2[O];
#pragma unroll
    for (int o = 0; o < O; ++o) b2[o] = wl[oB2 + o];
    const int ntiles = H >> 5;
    for (int rb = wave; rb * 32 < Nall; rb += nw) {
        const int n = rb * 32 + col;                                   // this lane's data row
        // B fragments of this row block stay in registers for all hidden tiles
        float bf[BF16 ? 1 : KS];
        bf16x8 bh[BF16 ? KS : 1];
        if (BF16) {
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s_ + 8 * half + j;
                    bh[s_][j] = (k < I) ? f32_to_bf16(xt[(size_t)k * Npad + n]) : (short)0;
                }
        } else {
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                const int k = 2 * s_ + half;
                bf[s_] = (k < I) ? xt[(size_t)k * Npad + n] : 0.0f;
            }
        }
        float sum[O];
#pragma unroll
        for (int o = 0; o < O; ++o) sum[o] = 0.0f;
        for (int t = 0; t < ntiles; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[r_] = 0.0f;
            const int hbase = t * 32;
            if (BF16) {
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    bf16x8 ah;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = 16 * s_ + 8 * half + j;
                        ah[j] = (k < I) ? f32_to_bf16(wl[k * H + hbase + col]) : (short)0;
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s_], acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    const int k = 2 * s_ + half;
                    const float a = (k < I) ? wl[k * H + hbase + col] : 0.0f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[s_], acc, 0, 0, 0);
                }
            }
            // epilogue in place: register r_ is hidden unit hbase + (r_ & 3) + 8 (r_ >> 2) + 4 half
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int h0 = hbase + 8 * q + 4 * half;
                const float4 b1v = *reinterpret_cast<const float4*>(wl + oB1 + h0);
                const float b1a[4] = {b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hid = sigmoidf_fast(acc[4 * q + e] - b1a[e]);
#pragma unroll
                    for (int o = 0; o < O; ++o) sum[o] = fmaf(hid, wl[oW2 + (h0 + e) * O + o], sum[o]);
                }
            }
        }
        // join the two lane halves (hidden units 4..7, 12..15, ... live in lanes 32..63)
        float tot[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            const unsigned u = __builtin_bit_cast(unsigned, sum[o]);
            auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            tot[o] = __builtin_bit_cast(float, (unsigned)r2[0]) + __builtin_bit_cast(float, (unsigned)r2[1]) - b2[o];
        }
        if (half == 0 && n < Nall) {
            const float y = data[(size_t)n * IPY + I];
            float a, bb = 0.f, c = 0.f;
            if (TASK == TASK_REG) {
                const float d = y - sigmoidf_fast(tot[0]);
                a = d * d;
            } else {
                ArgKey best = argmax_key(tot[0]);
                float se = 0.0f, oy = 0.0f;
                int arg = 0;
                const int yi = (int)y;
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const float out = sigmoidf_fast(tot[o]);
                    const ArgKey key = argmax_key(tot[o]);
                    if (argkey_greater(key, best)) { best = key; arg = o; }
                    se += expf_fast(out);
                    oy = (o == yi) ? out : oy;
                }
                a = oy - logf_fast(se);
                const float dd = (float)arg - y;
                bb = dd * dd;
                c = ((float)arg == y) ? 1.0f : 0.0f;
            }
            if (n < Ntr) { a_tr += a; b_tr += bb; c_tr += c; }
            else { a_te += a; b_te += bb; c_te += c; }
        }
    }
    a_tr = wave_allsum(a_tr);
    a_te = wave_allsum(a_te);
    if (TASK == TASK_CLS) {
        b_tr = wave_allsum(b_tr); c_tr = wave_allsum(c_tr);
        b_te = wave_allsum(b_te); c_te = wave_allsum(c_te);
    }
    EvalSums s;
    __syncthreads();
    if (lane == 0) {
        float* r = red + wave * 8;
        r[0] = a_tr; r[1] = b_tr; r[2] = c_tr; r[3] = a_te; r[4] = b_te; r[5] = c_te;
    }
    __syncthreads();
    s.a_tr = s.b_tr = s.c_tr = s.a_te = s.b_te = s.c_te = 0.f;
    for (int k = 0; k < nw; ++k) {
        const float* r = red + k * 8;
        s.a_tr += r[0]; s.b_tr += r[1]; s.c_tr += r[2]; s.a_te += r[3]; s.b_te += r[4]; s.c_te += r[5];
    }
    return s;
}

// Split-operand forward pass of a wide net (see SplitK above: on gfx950 the fp32 matrix instruction runs at VALU rate and blocks the
// VALU; six bf16 partial products per k-step of 16 take 198 pipe cycles instead of 520 and leave the vector issue free).  B: the
// split data image is made once by ptnn_set_data and read from L2 (16 bytes per lane, operand and k-step); a wave holds the
// operands of TWO row blocks for all hidden tiles.  A: the weights of a tile are read from the flat fp32 proposal in LDS and
// split in registers, once per pair of row blocks (a split image of W1 would take 96 KB of LDS next to the resident state).
// Tiles are software-pipelined: the matrix instructions of tile t + 1 are interleaved with the sigmoid / W2 epilogue of tile t.
template <int TASK, int I, int O>
__device__ __forceinline__ EvalSums eval_rows_mfma_wsplit(const float* __restrict__ wl, const uint4* __restrict__ xs,
                                                          const float* __restrict__ xt, const float* __restrict__ data, int IPY, int H,
                                                          int Ntr, int Nall, int Npad, float* __restrict__ red) {
    typedef SplitK<I> K;
    constexpr int KB = K::KB, KR = K::KR, CH = K::CH, NB = 2;
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    float a_tr = 0.f, b_tr = 0.f, c_tr = 0.f, a_te = 0.f, b_te = 0.f, c_te = 0.f;
    float b2[O];
#pragma unroll
    for (int o = 0; o < O; ++o) b2[o] = wl[oB2 + o];
    const int ntiles = H >> 5, nrb = Npad >> 5;
    struct AFrag { bf16x8 h[KB], m[KB], l[KB]; float r[KR > 0 ? KR : 1]; };
    auto make_a = [&](int t, AFrag& a) {
        const float* pa = wl + t * 32 + col;
#pragma unroll
        for (int s_ = 0; s_ < KB; ++s_) {
            unsigned hh[4], mm[4], ll[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * s_ + 8 * half + 2 * j;
                const float v0 = pa[min(k, I - 1) * H], v1 = pa[min(k + 1, I - 1) * H];
                const float x0 = (k < I) ? v0 : 0.0f, x1 = (k + 1 < I) ? v1 : 0.0f;
                const unsigned h = pack_bf16(x0, x1);
                const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
                const unsigned m = pack_bf16(r0, r1);
                const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
                hh[j] = h; mm[j] = m; ll[j] = pack_bf16(q0, q1);
            }
            a.h[s_] = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
            a.m[s_] = __builtin_bit_cast(bf16x8, make_uint4(mm[0], mm[1], mm[2], mm[3]));
            a.l[s_] = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
        }
#pragma unroll
        for (int s_ = 0; s_ < KR; ++s_) {
            const int k = K::KBF + 2 * s_ + half;
            const float v = pa[min(k, I - 1) * H];
            a.r[s_] = (k < I) ? v : 0.0f;
        }
    };
    for (int g = wave; g * NB < nrb; g += nw) {
        // B operands of this group's row blocks stay in registers for all hidden tiles (a block past the end repeats the last one
        // and is not scored)
        bf16x8 b_h[NB][KB], b_m[NB][KB], b_l[NB][KB];
        float b_r[NB][KR > 0 ? KR : 1];
#pragma unroll
        for (int b_ = 0; b_ < NB; ++b_) {
            const int rb = min(g * NB + b_, nrb - 1), n = rb * 32 + col;
            const uint4* base = xs + (size_t)n * CH;
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) {
                b_h[b_][s_] = __builtin_bit_cast(bf16x8, base[2 * s_ + half]);
                b_m[b_][s_] = __builtin_bit_cast(bf16x8, base[(size_t)Npad * CH + 2 * s_ + half]);
                b_l[b_][s_] = __builtin_bit_cast(bf16x8, base[(size_t)2 * Npad * CH + 2 * s_ + half]);
            }
#pragma unroll
            for (int s_ = 0; s_ < KR; ++s_) b_r[b_][s_] = xt[(size_t)(K::KBF + 2 * s_ + half) * Npad + n];
        }
        auto chain = [&](const AFrag& a, int b_) {
            f32x16 acc;
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[r_] = 0.0f;
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[s_], b_h[b_][s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s_], b_l[b_][s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s_], b_m[b_][s_], acc, 0, 0, 0);
            }
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s_], b_h[b_][s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s_], b_m[b_][s_], acc, 0, 0, 0);
            }
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s_], b_h[b_][s_], acc, 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < KR; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.r[s_], b_r[b_][s_], acc, 0, 0, 0);
            return acc;
        };
        f32x2 sum2[NB][O];                                              // partial W2 sums over the even / odd registers
#pragma unroll
        for (int b_ = 0; b_ < NB; ++b_)
#pragma unroll
            for (int o = 0; o < O; ++o) sum2[b_][o] = f32x2{0.0f, 0.0f};
        // epilogue in place: register r_ is hidden unit hbase + (r_ & 3) + 8 (r_ >> 2) + 4 half; bias and W2 rows of a tile are read
        // once for both blocks, BEFORE the interleaved region (the scheduler hints place matrix and vector instructions only: a
        // vector instruction that waits for an LDS read inside the region would drag the whole epilogue behind the matrix block)
        struct Epi { float b1[16]; float w2[16][O]; };                  // b1 pre-scaled by log2e
        auto load_epi = [&](int t, Epi& e) {
            const int hbase = t * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int h0 = hbase + 8 * q + 4 * half;
                const float4 b1v = *reinterpret_cast<const float4*>(wl + oB1 + h0);
                e.b1[4 * q] = LOG2E * b1v.x; e.b1[4 * q + 1] = LOG2E * b1v.y; e.b1[4 * q + 2] = LOG2E * b1v.z; e.b1[4 * q + 3] = LOG2E * b1v.w;
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_)
#pragma unroll
                    for (int o = 0; o < O; ++o) e.w2[4 * q + i_][o] = wl[oW2 + (h0 + i_) * O + o];
            }
        };
        // on register pairs (packed fp32: two elements per issue slot; this pass is VALU-bound), see eval_rows_mfma_split
        auto finish = [&](const f32x16 (&acc)[NB], const Epi& e) {
#pragma unroll
            for (int r_ = 0; r_ < 16; r_ += 2)
#pragma unroll
                for (int b_ = 0; b_ < NB; ++b_) {
                    const f32x2 zz = __builtin_elementwise_fma(f32x2{acc[b_][r_], acc[b_][r_ + 1]}, f32x2{-LOG2E, -LOG2E}, f32x2{e.b1[r_], e.b1[r_ + 1]});
                    const f32x2 ee = f32x2{__builtin_amdgcn_exp2f(zz.x), __builtin_amdgcn_exp2f(zz.y)} + f32x2{1.0f, 1.0f};
                    const f32x2 hid = f32x2{__builtin_amdgcn_rcpf(ee.x), __builtin_amdgcn_rcpf(ee.y)};
#pragma unroll
                    for (int o = 0; o < O; ++o) sum2[b_][o] = __builtin_elementwise_fma(hid, f32x2{e.w2[r_][o], e.w2[r_ + 1][o]}, sum2[b_][o]);
                }
        };
        // one pipeline stage: the matrix instructions of tile t + 1 into `nxt`, interleaved with the epilogue of tile t in `cur` (ONE
        // basic block: the scheduler hints only reach what sits in the same block)
        auto stage = [&](const f32x16 (&cur)[NB], f32x16 (&nxt)[NB], int t) {
            AFrag a;
            Epi e;
            make_a(t + 1, a);
            load_epi(t, e);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b_ = 0; b_ < NB; ++b_) nxt[b_] = chain(a, b_);
            finish(cur, e);
#pragma unroll
            for (int q_ = 0; q_ < NB * (6 * KB + KR); ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (8 * (2 + O) + 6 * KB + KR - 1) / (6 * KB + KR), 0);
                __builtin_amdgcn_sched_group_barrier(0x400, (32 + 6 * KB + KR - 1) / (6 * KB + KR), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto last = [&](const f32x16 (&cur)[NB], int t) { Epi e; load_epi(t, e); finish(cur, e); };
        f32x16 accA[NB], accB[NB];
        {
            AFrag a;
            make_a(0, a);
#pragma unroll
            for (int b_ = 0; b_ < NB; ++b_) accA[b_] = chain(a, b_);
        }
        int t = 0;
        for (; t + 2 < ntiles; t += 2) { stage(accA, accB, t); stage(accB, accA, t + 1); }   // two stages a trip: no accumulator is ever copied
        if (t + 1 < ntiles) { stage(accA, accB, t); last(accB, t + 1); }
        else last(accA, t);
#pragma unroll
        for (int b_ = 0; b_ < NB; ++b_) {
            const int rb = g * NB + b_, n = rb * 32 + col;
            // join the two lane halves (hidden units 4..7, 12..15, ... live in lanes 32..63)
            float tot[O];
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const unsigned u = __builtin_bit_cast(unsigned, sum2[b_][o].x + sum2[b_][o].y);
                auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                tot[o] = __builtin_bit_cast(float, (unsigned)r2[0]) + __builtin_bit_cast(float, (unsigned)r2[1]) - b2[o];
            }
            if (half == 0 && rb < nrb && n < Nall) {
                const float y = data[(size_t)n * IPY + I];
                float a_, bb = 0.f, c = 0.f;
                if (TASK == TASK_REG) {
                    const float d = y - sigmoidf_fast(tot[0]);
                    a_ = d * d;
                } else {
                    ArgKey best = argmax_key(tot[0]);
                    float se = 0.0f, oy = 0.0f;
                    int arg = 0;
                    const int yi = (int)y;
#pragma unroll
                    for (int o = 0; o < O; ++o) {
                        const float out = sigmoidf_fast(tot[o]);
                        const ArgKey key = argmax_key(tot[o]);
                        if (argkey_greater(key, best)) { best = key; arg = o; }
                        se += expf_fast(out);
                        oy = (o == yi) ? out : oy;
                    }
                    a_ = oy - logf_fast(se);
                    const float dd = (float)arg - y;
                    bb = dd * dd;
                    c = ((float)arg == y) ? 1.0f : 0.0f;
                }
                if (n < Ntr) { a_tr += a_; b_tr += bb; c_tr += c; }
                else { a_te += a_; b_te += bb; c_te += c; }
            }
        }
    }
    a_tr = wave_allsum(a_tr);
    a_te = wave_allsum(a_te);
    if (TASK == TASK_CLS) {
        b_tr = wave_allsum(b_tr); c_tr = wave_allsum(c_tr);
        b_te = wave_allsum(b_te); c_te = wave_allsum(c_te);
    }
    EvalSums s;
    __syncthreads();
    if (lane == 0) {
        float* r = red + wave * 8;
        r[0] = a_tr; r[1] = b_tr; r[2] = c_tr; r[3] = a_te; r[4] = b_te; r[5] = c_te;
    }
    __syncthreads();
    s.a_tr = s.b_tr = s.c_tr = s.a_te = s.b_te = s.c_te = 0.f;
    for (int k = 0; k < nw; ++k) {
        const float* r = red + k * 8;
        s.a_tr += r[0]; s.b_tr += r[1]; s.c_tr += r[2]; s.a_te += r[3]; s.b_te += r[4]; s.c_te += r[5];
    }
    return s;
}

// forward pass of a wide net under weight vector w (global): MFMA when the hidden layer tiles (H % 32 == 0), else the
// lane-per-row VALU path on the packed image.  `img` is the LDS image area (max of both layouts).
__device__ __forceinline__ bool wide_mfma(const SegParams& p) { return (p.H & 31) == 0 && p.xt != nullptr; }

// img_ready: the caller has already put the flat copy of w into img (MFMA layout only) and passed a barrier
template <int TASK, int I, int O>
__device__ __forceinline__ EvalSums wide_forward(const SegParams& p, const float* __restrict__ w, float* __restrict__ img,
                                                 float* __restrict__ red, bool img_ready = false) {
    const int Nall = p.Ntr + p.Nte;
    if (wide_mfma(p)) {
        if (!img_ready) {
            for (int j = threadIdx.x; j < p.P; j += blockDim.x) img[j] = w[j];    // flat copy: the layout IS [k][h]
            __syncthreads();
        }
        if (p.forward_bf16) return eval_rows_mfma<TASK, I, O, true>(img, p.xt, p.data, p.IPY, p.H, p.Ntr, Nall, p.Npad, red);
        if constexpr (SplitK<I>::OK) {
            if (p.fw_mfma == 2) return eval_rows_mfma_wsplit<TASK, I, O>(img, p.xs, p.xt, p.data, p.IPY, p.H, p.Ntr, Nall, p.Npad, red);
        }
        return eval_rows_mfma<TASK, I, O, false>(img, p.xt, p.data, p.IPY, p.H, p.Ntr, Nall, p.Npad, red);
    }
    build_fw<I, O>(w, img, p.H, p.FWS);
    __syncthreads();
    return eval_rows<TASK, I, O>(img, p.data, p.IPY, p.FWS, p.H, p.Ntr, Nall, red);
}

// One work-group per replica (p.G == 1), or the speculative schedule over p.G work-groups (one per CU): a round covers a WINDOW
// of up to WIDE_WINDOW steps, every one computed on the assumption that the steps before it reject, and the prefix up to and
// including the first accepted step is committed.  Wide nets accept 1 - 5 % of their proposals, so almost whole windows are
// committed.  A Langevin step costs five times a random-walk step here (its SGD epoch), and which step is which is on the tape:
// every group replays the same greedy list scheduling of the window (next step to the group with the least work so far), so the
// groups finish together instead of one sweeping while the other waits.  A group stops at its first accepted step (what it would
// compute after it can never be committed) and, before each step, looks whether an earlier step of another group has been accepted.
// Every group keeps its own copy of the chain vectors (group 0 the canonical rows, the others rows of the scratch buffer) and
// applies the same commits; what crosses CUs are {tag, value} granules: one verdict per step, and -- only from the group whose step
// was accepted -- its record and its vectors (proposal, SGD epoch).
//
// RES (matrix-core forward only, where the flat LDS image the MFMAs read IS the proposal): the CURRENT state lives in LDS too
// (2 x 70 KB of the 160 KB for the 32-512-1 net), the proposal is never written to global memory, the SGD epoch of a Langevin
// step reads it from LDS and hands back its share of |w - w_prop_gd|^2 from registers.  With compact traces (p.compact) a
// rejected step moves no vector at all: per step a random-walk proposal touches global memory for nothing but the shared data
// image, a Langevin one reads the cached epoch (70 KB) and writes its own (70 KB).  It was 280 - 560 KB per step and group.
constexpr int WIDE_WINDOW = 16;
template <int TASK, int I, int O, bool RES>
__device__ __forceinline__ void segment_wide_body(const SegParams& p, const SegDyn& dyn, const int step_begin, const int n_steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int OP = (O + 3) & ~3;
    const int G = p.G;
    const int lb = xcd_block(G);
    const int r = lb / G, grp = lb - r * G;
    const int gid = p.first_global + r;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int P = p.P, PS = p.PS, H = p.H;
    float* fw = smem;
    float* wc = fw + wide_img_floats(H, p.FWS, PS);             // RES: the current state (w, eta)
    float* red = wc + (RES ? PS : 0);
    float* part = red + MAX_WAVES * 8;
    float* scal = part + 4 * MAX_WAVES * OP;                    // 16 floats: staging of an accepted foreign record; then per window slot:
    float* s_u = scal + 16;                                     // the step's uniform
    float* s_ne = s_u + WIDE_WINDOW;                            // its eta noise
    float* s_lg = s_ne + WIDE_WINDOW;                           // its Langevin coin (0 / 1)
    float* s_lik = s_lg + WIDE_WINDOW;                          // proposal likelihood of a step this group computed
    float* s_la = s_lik + WIDE_WINDOW;                          // its log alpha
    float* s_v = s_la + WIDE_WINDOW;                            // verdicts of the window: 0 rejected, 1 accepted, 2 never computed
    const float* xy = p.data;                                   // global (L2 / scalar cache)
    float* const mine = p.wide_scratch + (size_t)(r * G + grp) * 5 * PS;
    float* w_prop = RES ? fw : mine;                            // RES: the flat LDS image is the proposal
    float* w_pgd = mine + PS;
    float* w_cur = RES ? wc : ((grp == 0) ? dyn.w_state + (size_t)r * PS : mine + 2 * (size_t)PS);   // chain state row (group 0: the canonical one)
    float* w_gd = (grp == 0) ? dyn.gd_w + (size_t)r * PS : mine + 3 * (size_t)PS;
    float* rec_w = (grp == 0) ? p.rec_w + (size_t)r * PS : mine + 4 * (size_t)PS;
    const bool compact = p.compact != 0;
    if (grp > 0 || RES) {
        // (group 0 touches the canonical rows at its first commit, which needs this group's first verdicts)
        for (int q = tid; q < PS / 4; q += nthr) {
            if (grp > 0 || RES) reinterpret_cast<float4*>(w_cur)[q] = reinterpret_cast<const float4*>(dyn.w_state + (size_t)r * PS)[q];
            if (grp > 0) {
                reinterpret_cast<float4*>(w_gd)[q] = reinterpret_cast<const float4*>(dyn.gd_w + (size_t)r * PS)[q];
                reinterpret_cast<float4*>(rec_w)[q] = reinterpret_cast<const float4*>(p.rec_w + (size_t)r * PS)[q];
            }
        }
        __syncthreads();
    }
    granule_t* const xs = p.xslots + (size_t)r * 2 * MAX_SLOTS * SL_COUNT;
    granule_t* const xv = p.xverdict + (size_t)r * 2 * MAX_SLOTS;
    granule_t* const xw = p.xw + (size_t)r * 2 * G * 2 * PS;    // [parity][group][proposal ++ its SGD epoch]

    const float T = p.temps[r];
    float eta = (TASK == TASK_REG) ? w_cur[P] : 0.0f;
    float* sf = p.st_f + (size_t)r * SF_COUNT;
    int* si = p.st_i + (size_t)r * SI_COUNT;
    float lik, prior_cur, tau_eta_last, rec_rmse_tr, rec_rmse_te, rec_acc_tr, rec_acc_te;
    int nacc, gd_valid, lg_count, rec_row;
    if (step_begin == 0) {
        chain_startup<TASK, I, O>(p, xy, w_cur, fw, red, T, eta, lik, prior_cur);
        tau_eta_last = eta;
        rec_rmse_tr = rec_rmse_te = rec_acc_tr = rec_acc_te = 0.f;
        nacc = 0; gd_valid = 0; lg_count = 0; rec_row = 0;
        __syncthreads();
    } else {
        lik = sf[SF_LIK]; prior_cur = sf[SF_PRIOR]; tau_eta_last = sf[SF_TAU_LAST];
        rec_rmse_tr = sf[SF_REC_RMSE_TR]; rec_rmse_te = sf[SF_REC_RMSE_TE];
        rec_acc_tr = sf[SF_REC_ACC_TR]; rec_acc_te = sf[SF_REC_ACC_TE];
        nacc = si[SI_NACC]; gd_valid = dyn.gd_valid[r]; lg_count = si[SI_LG_COUNT]; rec_row = si[SI_REC_ROW];
    }

    const size_t trow = (size_t)r * p.trace_cap;        // traces are rings of trace_cap rows per replica (== S unless streaming)
    const int end = step_begin + n_steps;
    const int nv = (P + 3) >> 2;
    const int W = (G > 1) ? min(max(p.wide_window, G), WIDE_WINDOW) : 1;
    unsigned epoch = dyn.epoch_base;
    int par = 0;
    bool failed = false;
    int i = step_begin;
    while (i < end && !failed) {
        epoch += 1;
        int k = min(W, end - i);
        if (p.switch_step > i) k = min(k, p.switch_step - i);   // a round never straddles the temperature switch
        if (i == p.switch_step) {
            const EvalSums sc = wide_forward<TASK, I, O>(p, w_cur, fw, red);
            float ll, r1, r2, a1, a2;
            finish_eval<TASK>(sc, p.Ntr, p.Nte, tau_eta_last, ll, r1, r2, a1, a2);
            lik = ll;
            __syncthreads();
        }
        const float adapttemp = (p.switch_step >= 0 && i >= p.switch_step) ? 1.0f : T;
        // the scalars {lx, u, n_eta} of the window's steps: one Philox call each
        if (tid < k) {
            uint32_t x[4];
            philox4x32_10(0u, (uint32_t)(i + tid), p.noise_shared ? 0u : (uint32_t)gid, STREAM_STEP, p.seed_lo, p.seed_hi, x);
            float n2, n3;
            box_muller(x[2], x[3], n2, n3);
            s_u[tid] = u23(x[1]); s_ne[tid] = n2;
            s_lg[tid] = (p.use_lg && u23(x[0]) < p.l_prob) ? 1.0f : 0.0f;
        }
        __syncthreads();
        // who computes which step: greedy list scheduling on the known costs, replayed identically by every group
        unsigned my_steps = 0;
        {
            int load[4] = {0, 0, 0, 0};
            for (int s_ = 0; s_ < k; ++s_) {
                int g_ = 0;
                for (int c = 1; c < G; ++c)
                    if (load[c] < load[g_]) g_ = c;
                load[g_] += (s_lg[s_] != 0.0f) ? 5 : 1;
                if (g_ == grp) my_steps |= 1u << s_;
            }
        }
        bool stopped = false;
        int my_acc = -1;                                        // my accepted step of this window, if any (then my last one)
        bool a_lg = false;
        float a_lik = 0.f, a_prior = 0.f, a_eta = 0.f, a_rm_tr = 0.f, a_rm_te = 0.f, a_ac_tr = 0.f, a_ac_te = 0.f;
        for (int s_ = 0; s_ < k && !failed; ++s_) {
            if (!((my_steps >> s_) & 1u)) continue;
            if (G > 1 && !stopped) {                            // has an earlier step of another group been accepted in the meantime?
                bool hit = false;
                if (tid < s_) {
                    const granule_t x = __hip_atomic_load(xv + (size_t)par * MAX_SLOTS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hit = ((unsigned)(x >> 32) == epoch) && (__builtin_bit_cast(float, (unsigned)x) == 1.0f);
                }
                if (__syncthreads_or(hit ? 1 : 0)) stopped = true;
            }
            if (stopped) {
                if (tid == 0) granule_store(xv + (size_t)par * MAX_SLOTS + s_, epoch, 2.0f);
                continue;
            }
            const int j = i + s_;
            const bool lg = s_lg[s_] != 0.0f;
            const float u = s_u[s_], n_eta = s_ne[s_];
            float diff_prop = 0.0f;
            if (lg && !gd_valid) {
                // (RES: through the one out-of-line copy of the row loop this kernel has -- a uniform of -1 makes it write its result
                // whatever the ratio)
                if constexpr (RES) (void)sgd_sweep_wide_decide<TASK, I, O>(w_cur, w_gd, xy, p.IPY, p.Ntr, H, p.lr, part, w_cur, red, 0.0f, 0.0f, p.step_w, 1.0f, -1.0f);
                else sgd_sweep_wide<TASK, I, O>(w_cur, w_gd, xy, p.IPY, p.Ntr, H, p.lr, part);
                gd_valid = 1;
            }
            // ONE pass over the weights: draw the noise (4 normals per Philox call), form the proposal from w (random walk) or
            // from the cached SGD epoch (Langevin), put it where the forward pass reads it (the flat LDS image of the MFMA
            // layout) and in its global row, and add up |proposal|^2 (prior) and |noise|^2 (Langevin ratio) on the way.  The
            // noise itself is never stored.  (It used to be five passes through global memory: tape, proposal, image copy and
            // the two norms.)
            const bool img_direct = RES || wide_mfma(p);
            float ssq_part = 0.0f, nsq_part = 0.0f;
            auto propose = [&](const float* __restrict__ base) {
                for (int q = tid; q < nv; q += nthr) {
                    uint32_t x[4];
                    philox4x32_10((uint32_t)q, (uint32_t)j, p.noise_shared ? 0u : (uint32_t)gid, STREAM_WNOISE, p.seed_lo, p.seed_hi, x);
                    float n[4];
                    box_muller(x[0], x[1], n[0], n[1]);
                    box_muller(x[2], x[3], n[2], n[3]);
                    const int j0 = 4 * q;
                    if (j0 + 3 < P) {
                        const float4 b = *reinterpret_cast<const float4*>(base + j0);
                        const float4 v = make_float4(fmaf(p.step_w, n[0], b.x), fmaf(p.step_w, n[1], b.y), fmaf(p.step_w, n[2], b.z),
                                                     fmaf(p.step_w, n[3], b.w));
                        if (!RES) *reinterpret_cast<float4*>(w_prop + j0) = v;
                        if (img_direct) *reinterpret_cast<float4*>(fw + j0) = v;
                        ssq_part = fmaf(v.x, v.x, ssq_part); ssq_part = fmaf(v.y, v.y, ssq_part);
                        ssq_part = fmaf(v.z, v.z, ssq_part); ssq_part = fmaf(v.w, v.w, ssq_part);
                        nsq_part = fmaf(n[0], n[0], nsq_part); nsq_part = fmaf(n[1], n[1], nsq_part);
                        nsq_part = fmaf(n[2], n[2], nsq_part); nsq_part = fmaf(n[3], n[3], nsq_part);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (j0 + e < P) {
                                const float v = fmaf(p.step_w, n[e], base[j0 + e]);
                                if (!RES) w_prop[j0 + e] = v;
                                if (img_direct) fw[j0 + e] = v;
                                ssq_part = fmaf(v, v, ssq_part);
                                nsq_part = fmaf(n[e], n[e], nsq_part);
                            }
                    }
                }
            };
            // (two calls, not one pointer picked at run time: with RES the random-walk base is in LDS and the cached epoch in
            // global memory, and a pointer that may be either costs flat accesses in the one loop every step runs)
            if (lg) propose(w_gd);
            else propose(w_cur);
            __syncthreads();
            if (lg && !RES) {
                sgd_sweep_wide<TASK, I, O>(w_prop, w_pgd, xy, p.IPY, p.Ntr, H, p.lr, part);
                const float d1 = block_sumsq_diff(w_cur, w_pgd, P, red);
                const float d2 = block_sum(nsq_part, red);
                diff_prop = langevin_ratio(d1, d2, p.step_w, adapttemp);
            }
            float eta_pro = eta;
            if (TASK == TASK_REG) eta_pro = fmaf(p.step_eta, n_eta, eta);
            const EvalSums es = wide_forward<TASK, I, O>(p, w_prop, fw, red, img_direct);
            float ll, rm_tr, rm_te, ac_tr, ac_te;
            finish_eval<TASK, true>(es, p.Ntr, p.Nte, eta_pro, ll, rm_tr, rm_te, ac_tr, ac_te);
            const float lik_prop = ll / adapttemp;
            const float ssq = block_sum(ssq_part, red);
            const float prior_prop = prior_value<TASK>(p, ssq, eta_pro);
            float logalpha;
            bool accept;
            if (lg && RES) {
                // the proposal is still in LDS (the forward pass only read it): its epoch runs last, decides, and writes its 70 KB
                // out only when the step is accepted (SweepDecide)
                const float d2 = block_sum(nsq_part, red);
                const SweepVerdict v = sgd_sweep_wide_decide<TASK, I, O>(w_prop, w_pgd, xy, p.IPY, p.Ntr, H, p.lr, part, w_cur, red,
                                                                         (lik_prop - lik) + (prior_prop - prior_cur), d2, p.step_w, adapttemp, u);
                logalpha = v.logalpha; accept = v.accept != 0;
            } else {
                logalpha = (lik_prop - lik) + (prior_prop - prior_cur) + diff_prop;
                const float mh = (logalpha != logalpha) ? 1.0f : fminf(1.0f, expf_fast(logalpha));
                accept = u < mh;
            }
            if (tid == 0) { s_lik[s_] = lik_prop; s_la[s_] = logalpha; }
            if (accept) {
                my_acc = s_; stopped = true;
                a_lg = lg; a_lik = lik_prop; a_prior = prior_prop; a_eta = eta_pro;
                a_rm_tr = rm_tr; a_rm_te = rm_te; a_ac_tr = ac_tr; a_ac_te = ac_te;
                if (G > 1) {
                    // an accepted step publishes its record and its vectors; the other groups read them at commit, and only then
                    if (tid == 0) {
                        granule_t* xr = xs + ((size_t)par * MAX_SLOTS + s_) * SL_COUNT;
                        granule_store(xr + SL_LIKPROP, epoch, lik_prop); granule_store(xr + SL_PRIORPROP, epoch, prior_prop);
                        granule_store(xr + SL_ETAPRO, epoch, eta_pro);
                        granule_store(xr + SL_RM_TR, epoch, rm_tr); granule_store(xr + SL_RM_TE, epoch, rm_te);
                        granule_store(xr + SL_AC_TR, epoch, ac_tr); granule_store(xr + SL_AC_TE, epoch, ac_te);
                    }
                    granule_t* xo = xw + ((size_t)par * G + grp) * 2 * PS;
                    for (int e = tid; e < P; e += nthr) {
                        granule_store(xo + e, epoch, w_prop[e]);
                        if (lg) granule_store(xo + PS + e, epoch, w_pgd[e]);
                    }
                }
            }
            if (G > 1 && tid == 0) granule_store(xv + (size_t)par * MAX_SLOTS + s_, epoch, accept ? 1.0f : 0.0f);
        }
        // the first accepted step of the window
        int m = k;
        if (G == 1) {
            if (my_acc == 0) m = 0;
        } else {
            bool ok = true;
            if (tid < k) {
                float v = 0.0f;
                ok = granule_wait(xv + (size_t)par * MAX_SLOTS + tid, epoch, v);
                s_v[tid] = v;
            }
            if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
            for (int s_ = k - 1; s_ >= 0; --s_)
                if (s_v[s_] == 1.0f) m = s_;
        }
        const int ncommit = (m < k) ? m + 1 : k;
        // what the committed steps leave behind apart from the state: the Langevin count and the last PROPOSED eta
        for (int s_ = 0; s_ < ncommit; ++s_) lg_count += (s_lg[s_] != 0.0f) ? 1 : 0;
        if (TASK == TASK_REG) tau_eta_last = fmaf(p.step_eta, s_ne[ncommit - 1], eta);
        const bool lg_m = (m < k) && (s_lg[m] != 0.0f);
        const bool acc_me = (m < k) && (my_acc == m);
        const int acc_before = nacc;
        // new chain scalars of an accepted foreign step: from its record
        if (m < k && !acc_me) {
            bool ok = true;
            if (tid < SL_COUNT && (tid == SL_LIKPROP || tid == SL_PRIORPROP || tid == SL_ETAPRO || tid == SL_RM_TR || tid == SL_RM_TE ||
                                   tid == SL_AC_TR || tid == SL_AC_TE)) {
                float v = 0.0f;
                ok = granule_wait(xs + ((size_t)par * MAX_SLOTS + m) * SL_COUNT + tid, epoch, v);
                scal[tid] = v;
            }
            if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
        }
        __syncthreads();                                    // every reader of w_cur / w_gd of this round is done
        // trace rows of my committed steps: rows of PS / PW floats are 16-byte aligned and padded: whole float4s, the tail past P
        // rewritten as it must be (trace row: zeros; state rows: the element at P is eta, restored at the end of the launch)
        for (int s_ = 0; s_ < ncommit; ++s_) {
            if (!((my_steps >> s_) & 1u)) continue;
            const bool acc_row = acc_me && (s_ == m);
            const size_t tpos = trow + (size_t)((i + s_ + 1) % p.trace_cap);
            if (acc_row || !compact) {                          // compact traces: a rejected step repeats row rec_row, no vector moves
                float* prow = p.tr_pos_w + tpos * (size_t)p.PW;
                for (int e = 4 * nv + tid; e < p.PW; e += nthr) prow[e] = 0.0f;
                auto copy_row = [&](const float* __restrict__ src) {
                    for (int q = tid; q < nv; q += nthr) {
                        float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
                        if (4 * q + 3 >= P) {                    // the last, partial quad: nothing past P
                            if (4 * q + 1 >= P) v.y = 0.0f;
                            if (4 * q + 2 >= P) v.z = 0.0f;
                            v.w = 0.0f;
                        }
                        *reinterpret_cast<float4*>(prow + 4 * q) = v;
                    }
                };
                if (acc_row) copy_row(w_prop);
                else copy_row(rec_w);
            }
            if (tid == 0) {
                const float lp = s_lik[s_];
                store_trace_row(p.tr_scal + tpos * TR_COUNT, (TASK == TASK_REG) ? lp : lp * adapttemp,
                                acc_row ? a_rm_tr : rec_rmse_tr, acc_row ? a_rm_te : rec_rmse_te, acc_row ? a_ac_tr : rec_acc_tr,
                                acc_row ? a_ac_te : rec_acc_te, acc_before, s_la[s_], compact ? (acc_row ? i + s_ + 1 : rec_row) : 0);
            }
        }
        if (m < k) {
            nacc += 1;
            rec_row = i + m + 1;
            gd_valid = lg_m ? 1 : 0;
            __syncthreads();                                    // the trace rows above have read rec_w
            if (acc_me) {
                lik = a_lik; prior_cur = a_prior; eta = a_eta;
                rec_rmse_tr = a_rm_tr; rec_rmse_te = a_rm_te; rec_acc_tr = a_ac_tr; rec_acc_te = a_ac_te;
                for (int q = tid; q < nv; q += nthr) {
                    float4 v = *reinterpret_cast<const float4*>(w_prop + 4 * q);
                    if (4 * q + 3 >= P) {
                        if (4 * q + 1 >= P) v.y = 0.0f;
                        if (4 * q + 2 >= P) v.z = 0.0f;
                        v.w = 0.0f;
                    }
                    *reinterpret_cast<float4*>(w_cur + 4 * q) = v;
                    *reinterpret_cast<float4*>(rec_w + 4 * q) = v;
                    if (a_lg) *reinterpret_cast<float4*>(w_gd + 4 * q) = *reinterpret_cast<const float4*>(w_pgd + 4 * q);
                }
            } else {
                lik = scal[SL_LIKPROP]; prior_cur = scal[SL_PRIORPROP]; eta = scal[SL_ETAPRO];
                rec_rmse_tr = scal[SL_RM_TR]; rec_rmse_te = scal[SL_RM_TE]; rec_acc_tr = scal[SL_AC_TR]; rec_acc_te = scal[SL_AC_TE];
                // the accepted step's group: replay the assignment
                int owner = 0;
                {
                    int load[4] = {0, 0, 0, 0};
                    for (int s_ = 0; s_ <= m; ++s_) {
                        int g_ = 0;
                        for (int c = 1; c < G; ++c)
                            if (load[c] < load[g_]) g_ = c;
                        load[g_] += (s_lg[s_] != 0.0f) ? 5 : 1;
                        owner = g_;
                    }
                }
                const granule_t* xo = xw + ((size_t)par * G + owner) * 2 * PS;
                bool ok = true;
                for (int e = tid; e < 4 * nv; e += nthr) {
                    float v = 0.0f, g_ = 0.0f;
                    if (e < P) {
                        ok = granule_wait(xo + e, epoch, v) && ok;
                        if (lg_m) ok = granule_wait(xo + PS + e, epoch, g_) && ok;
                    }
                    w_cur[e] = v; rec_w[e] = v;
                    if (lg_m) w_gd[e] = g_;
                }
                if (__syncthreads_or(ok ? 0 : 1)) { failed = true; break; }
            }
        }
        __syncthreads();
        i += ncommit;
        par ^= 1;
    }
    if (failed) {
        if (tid == 0) atomicAdd(p.error_flag, 1);           // a bounded spin ran out: the host reports it
        return;
    }
    if (RES && grp == 0) {                                      // the state back to its canonical row: swap rounds and the next launch read it
        __syncthreads();
        float* row = dyn.w_state + (size_t)r * PS;
        for (int q = tid; q < PS / 4; q += nthr) reinterpret_cast<float4*>(row)[q] = reinterpret_cast<const float4*>(w_cur)[q];
        __syncthreads();
        w_cur = row;
    }
    if (grp == 0 && tid == 0) {
        w_cur[P] = eta;
        sf[SF_LIK] = lik; sf[SF_PRIOR] = prior_cur; sf[SF_TAU_LAST] = tau_eta_last;
        sf[SF_REC_RMSE_TR] = rec_rmse_tr; sf[SF_REC_RMSE_TE] = rec_rmse_te;
        sf[SF_REC_ACC_TR] = rec_acc_tr; sf[SF_REC_ACC_TE] = rec_acc_te;
        si[SI_NACC] = nacc; dyn.gd_valid[r] = gd_valid; si[SI_LG_COUNT] = lg_count; si[SI_REC_ROW] = rec_row;
        p.L_handoff[gid] = (TASK == TASK_REG) ? lik * T : lik;
        p.L_final[gid] = lik;
        post_raw(p, gid, lik, prior_cur, T, step_begin + n_steps - 1);
    }
}


// stand-alone model functions for wide nets: mode 0 = evaluate, 1 = langevin_gradient (mode 2, the tape, is shape
// independent and served by model_kernel)
template <int TASK, int I, int O>
__global__ void __launch_bounds__(MAX_THREADS) model_wide_kernel(const SegParams p, const int mode, const float* __restrict__ w_in,
                                                                  const float* __restrict__ tau_sq, float* __restrict__ out, int a0,
                                                                  int a1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int OP = (O + 3) & ~3;
    const int b = blockIdx.x, tid = threadIdx.x;
    float* fw = smem;
    float* red = fw + wide_img_floats(p.H, p.FWS, p.PS);
    float* part = red + MAX_WAVES * 8;
    float* scal = part + 4 * MAX_WAVES * OP;
    if (mode == 2) {
        tape_step(p, a0, a1, out, scal);                  // noise straight to the output buffer (16-byte aligned)
        __syncthreads();
        if (tid < 3) out[p.PS + tid] = scal[tid];
        return;
    }
    const float* w = w_in + (size_t)b * p.P;
    if (mode == 1) {
        sgd_sweep_wide<TASK, I, O>(w, out + (size_t)b * p.P, p.data, p.IPY, p.Ntr, p.H, p.lr, part);
        return;
    }
    if (mode == 3) {
        // timing of the epoch (constant-rate counter): a0 single epochs, then a0 PAIRS through one row loop; out = ticks of each
        // (ptnn_time_sgd_epoch for wide nets; results go to p.wide_scratch rows, which no chain uses while this runs)
        float* oa = p.wide_scratch;
        float* ob = p.wide_scratch + p.PS;
        unsigned long long t0 = wall_clock64();
        for (int rep = 0; rep < a0; ++rep) sgd_sweep_wide<TASK, I, O>(w, oa, p.data, p.IPY, p.Ntr, p.H, p.lr, part);
        const unsigned long long t1 = wall_clock64();
        float dd[2];
        for (int rep = 0; rep < a0; ++rep) sgd_sweep_wide_pair<TASK, I, O>(w, w, oa, ob, p.data, p.IPY, p.Ntr, p.H, p.lr, part, nullptr, dd);
        const unsigned long long t2 = wall_clock64();
        if (tid == 0) {
            out[0] = __uint_as_float((unsigned)((t1 - t0) & 0xffffffffull)); out[1] = __uint_as_float((unsigned)((t1 - t0) >> 32));
            out[2] = __uint_as_float((unsigned)((t2 - t1) & 0xffffffffull)); out[3] = __uint_as_float((unsigned)((t2 - t1) >> 32));
        }
        return;
    }
    const EvalSums s = wide_forward<TASK, I, O>(p, w, fw, red);
    const float eta = (TASK == TASK_REG) ? logf_fast(tau_sq[b]) : 0.0f;
    float ll, r1, r2, a_tr, a_te;
    finish_eval<TASK>(s, p.Ntr, p.Nte, eta, ll, r1, r2, a_tr, a_te);
    const float ss = block_sumsq(w, p.P, red);
    const float pr = prior_value<TASK>(p, ss, eta);
    float ll_te = s.a_te;
    if (TASK == TASK_REG) ll_te = -0.5f * (float)p.Nte * (LOG_2PI + eta) - 0.5f * s.a_te * expf_fast(-eta);
    if (tid == 0) {
        float* o = out + (size_t)b * 8;
        o[0] = ll; o[1] = r1; o[2] = r2; o[3] = a_tr; o[4] = a_te; o[5] = pr; o[6] = ll_te; o[7] = 0.f;
    }
}

#pragma clang fp contract(off)

// One block's share of a swap round: the cascade (every block recomputes it in LDS), then block b's row.  mode bit 0: apply the
// local moves; bit 1: count the round and log it (block 0); bit 2: the source rows come from the gathered exchange buffer.
__device__ __forceinline__ void swap_block(const SwapParams& sp, const int round, const int mode, const int b, float* smem) {
    float* sL = smem;
    float* sU = smem + sp.R;
    int* sSrc = reinterpret_cast<int*>(smem + 2 * sp.R);
    const int nsw = cascade_lds(sp, round, sL, sU, sSrc);
    if (sp.label_mode) {
        // temperature t is handed to the chain that held temperature src[t]: only the maps change
        if (mode & 1) {
            const int g = sp.first_global + b;
            const int t_old = sp.label_cur[g];
            int t_new = t_old;
            for (int t = threadIdx.x; t < sp.R; t += blockDim.x)
                if (sSrc[t] == t_old) sL[0] = __int_as_float(t);           // exactly one t has src[t] == t_old (a permutation)
            __syncthreads();
            t_new = __float_as_int(sL[0]);
            if (threadIdx.x == 0 && t_new != t_old) {
                const float To = sp.temps_global[t_old], Tn = sp.temps_global[t_new];
                sp.temps_local[b] = Tn;
                // the chain keeps its own likelihood; while the chains are tempered it is re-tempered for the new temperature
                if (!sp.canonical) sp.st_f[(size_t)b * SF_COUNT + SF_LIK] *= To / Tn;
            }
            if (b == 0)
                for (int t = threadIdx.x; t < sp.R; t += blockDim.x) {
                    const int slot = sp.slot_cur[sSrc[t]];
                    sp.slot_next[t] = slot;
                    sp.label_next[slot] = t;
                }
        }
    } else if (mode & 1) {
        const int k = sp.first_global + b;
        const int s = sSrc[k];
        const int sl = s - sp.first_global;
        // gd_valid of the destination = gd_valid of the source when the source is local; a row that arrives from
        // another GPU comes without its cached gradient
        int valid = 0;
        if (mode & 4) {                                     // the source row is in the gathered exchange buffer, wherever it ran
            const float* from = sp.xchg + (size_t)s * sp.XS;
            float* to = sp.next + (size_t)b * sp.PS;
            float* gto = sp.gd_next + (size_t)b * sp.PS;
            // rows are multiples of 4 floats and 16-byte aligned (PS = round4(P + 1), XS = round4(2 PS + 4))
            for (int j = threadIdx.x; j < sp.PS / 4; j += blockDim.x) {
                reinterpret_cast<float4*>(to)[j] = reinterpret_cast<const float4*>(from)[j];
                reinterpret_cast<float4*>(gto)[j] = reinterpret_cast<const float4*>(from + sp.PS)[j];
            }
            valid = (from[2 * sp.PS] != 0.0f) ? 1 : 0;
        } else if (sl >= 0 && sl < sp.Rl) {
            const float* from = sp.cur + (size_t)sl * sp.PS;
            float* to = sp.next + (size_t)b * sp.PS;
            const float* gfrom = sp.gd_cur + (size_t)sl * sp.PS;
            float* gto = sp.gd_next + (size_t)b * sp.PS;
            for (int j = threadIdx.x; j < sp.PS / 4; j += blockDim.x) {
                reinterpret_cast<float4*>(to)[j] = reinterpret_cast<const float4*>(from)[j];
                reinterpret_cast<float4*>(gto)[j] = reinterpret_cast<const float4*>(gfrom)[j];
            }
            valid = sp.gd_valid_cur[sl];
        }
        if (threadIdx.x == 0) sp.gd_valid_next[b] = valid;
        if (sp.rule == 1 && s != k && threadIdx.x == 0) {
            // the arriving state brings its own likelihood (re-tempered for this slot) and prior
            const float lraw = sp.L_raw[(size_t)s * sp.L_stride];
            sp.st_f[(size_t)b * SF_COUNT + SF_LIK] = sp.canonical ? lraw : lraw / sp.temps_global[k];
            sp.st_f[(size_t)b * SF_COUNT + SF_PRIOR] = sp.prior_post[(size_t)s * sp.L_stride];
        }
    }
    if (b == 0) {
        if (sp.src_out) for (int k = threadIdx.x; k < sp.R; k += blockDim.x) sp.src_out[k] = sSrc[k];
        if (mode & 2) {
            if (sp.src_log && round < sp.log_capacity)
                for (int k = threadIdx.x; k < sp.R; k += blockDim.x) sp.src_log[(size_t)round * sp.R + k] = sSrc[k];
            if (threadIdx.x == 0) {
                sp.counters[0] += nsw;
                sp.counters[1] += (sp.rule == 1) ? (sp.R - 1 - (round & 1) + 1) / 2 : sp.R - 1;      // pairs proposed
            }
        }
        if (threadIdx.x == 0 && sp.progress) __hip_atomic_store(sp.progress, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


#ifndef PTNN_SHAPE_TU      // non-template kernels: defined in the main translation unit only
// mode bit 0: apply the local moves; bit 1: count the round and log it
// exchange row of every local replica: state, cached gradient, its valid flag and the posted scalar, ready for the all-gather
__global__ void xchg_pack_kernel(const SwapParams sp) {
    const int b = blockIdx.x;
    float* row = sp.xchg + (size_t)(sp.first_global + b) * sp.XS;
    const float* from = sp.cur + (size_t)b * sp.PS;
    const float* gfrom = sp.gd_cur + (size_t)b * sp.PS;
    for (int j = threadIdx.x; j < sp.PS; j += blockDim.x) { row[j] = from[j]; row[sp.PS + j] = gfrom[j]; }
    if (threadIdx.x == 0) {
        row[2 * sp.PS] = sp.gd_valid_cur[b] ? 1.0f : 0.0f;
        row[2 * sp.PS + 1] = sp.L[sp.first_global + b];
        if (sp.rule == 1) {
            row[2 * sp.PS + 2] = sp.L_raw[sp.first_global + b];
            row[2 * sp.PS + 3] = sp.prior_post[sp.first_global + b];
        }
    }
}

// Restart of the chains (ptnn_set_state), one block per local replica, everything a run starts from in ONE kernel on the handle's
// stream: the initial weights into both state buffers (REG:649), the recorded row = ones and row 0 of every trace (Q7: pos_w =
// ones, REG:240; likeh = -100, REG:292-293; the rest zero), the cached-gradient rows and flags, the per-chain scalars and
// counters, the temperatures, the error flag, the swap counters and the identity slot <-> temperature maps.  (It was some twenty
// blocking copies and fills on the null stream, two of them hipMemcpy2D calls with the trace ring's pitch -- 74 MB for Ionosphere,
// where a restart cost 25 ms: a fifth of a whole 256-replica run, profiles/r03a_gap_probe_before.json.)
struct ResetParams {
    int R, Rl, P, PS, PW;
    size_t cap;
    const float* w0;          // [Rl][P]  staged initial weights
    const float* temps_in;    // [Rl]
    float *state0, *state1, *rec_w, *gd0, *gd1, *st_f, *temps, *pos_w, *scal;
    int *gd_valid0, *gd_valid1, *st_i, *error, *label0, *label1, *slot0, *slot1;
    long long* counters;
};
__global__ void chain_reset_kernel(const ResetParams q) {
    const int r = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const size_t row = (size_t)r * q.PS;
    for (int j = tid; j < q.PS; j += nthr) {
        const float v = (j < q.P) ? q.w0[(size_t)r * q.P + j] : 0.0f;
        q.state0[row + j] = v; q.state1[row + j] = v;
        q.rec_w[row + j] = 1.0f;
        q.gd0[row + j] = 0.0f; q.gd1[row + j] = 0.0f;
    }
    float* prow = q.pos_w + (size_t)r * q.cap * q.PW;
    for (int j = tid; j < q.PW; j += nthr) prow[j] = (j < q.P) ? 1.0f : 0.0f;
    if (tid == 0) {
        store_trace_row(q.scal + (size_t)r * q.cap * TR_COUNT, -100.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0, 0.0f);
        q.gd_valid0[r] = 0; q.gd_valid1[r] = 0;
        q.temps[r] = q.temps_in[r];
    }
    if (tid < SF_COUNT) q.st_f[(size_t)r * SF_COUNT + tid] = 0.0f;
    if (tid < SI_COUNT) q.st_i[(size_t)r * SI_COUNT + tid] = 0;
    if (r == 0) {
        if (tid == 0) { q.counters[0] = 0; q.counters[1] = 0; *q.error = 0; }
        for (int k = tid; k < q.R; k += nthr) { q.label0[k] = k; q.label1[k] = k; q.slot0[k] = k; q.slot1[k] = k; }
    }
}

__global__ void swap_kernel(const SwapParams sp, const int round, const int mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    swap_block(sp, round, mode, blockIdx.x, smem);
}

#endif  // PTNN_SHAPE_TU
