// ptnn_dev_sweep_forward.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// R4/R5 the SGD epoch (sgd_sweep), R1-R3/R6/R7 forward image, eval_rows, likelihood / prior / proposal ratio, chain start-up.

// ------------------------------------------------------------------------------------------------
// R5  Network.langevin_gradient (REG:99-118 / CLS:114-132) on wave 0: lane h owns hidden unit h
// (column h of W1, row h of W2, B1[h]); B2 and the outputs are replicated in every lane.  Rows are visited
// in file order, each row is a dependent chain; the next row's inputs are fetched while this one computes.
// Lanes >= H carry B1 = +1e30 so their hidden activation is exactly 0 and they never contribute or update.
// ------------------------------------------------------------------------------------------------
// Where the lane groups of a packed sweep find their input vectors when these are PROPOSALS that nobody has written out yet:
// group g sweeps from  base + step_w * noise  with the noise row of ring slot (pos0 + g) mod ring and base = w_gd when the
// step's Langevin coin (first scalar of the same ring slot) came up, else w_cur -- the same fmaf the proposal is written with.
struct SweepProposals {
    const float* noise;      // ring[ring][nstride]
    const float* scal;       // ring[ring][4]: {lx, u, n_eta, -}
    const float* w_cur;
    const float* w_gd;
    int pos0, ring, nstride;
    float step_w, l_prob;
    int use_lg;
};

template <int TASK, int I, int O, int NRED, bool PROP = false>
__device__ __forceinline__ void sgd_sweep(const float* __restrict__ w_in, float* __restrict__ w_out,
                                          const float* __restrict__ xy, const float* __restrict__ gdata, int Ntr, int H,
                                          float lr, int ngroups = 1, int gstride = 0, const SweepProposals* pp = nullptr) {
    // All weights are kept pre-multiplied by c = -log2(e): the pre-activation then IS the exponent of
    // sigmoid(z) = 1 / (1 + 2^(c z)), and every update rule keeps its shape with lr folded into two constants:
    //   W1' += lr (g' dh) x,  B1' -= lr g' dh      with g' = sum_o od W2'[.,o]  (= c g)
    //   W2' += (c lr) od hid, B2' -= (c lr) od
    // B2' lives negated in lane 0 of a per-lane constant that rides along in the wave reduction of hid * W2'.
    // Lane groups: the wave is cut into aligned groups of 2^NRED lanes; group g < ngroups runs its OWN sweep on the
    // vectors at w_in + g gstride -> w_out + g gstride (same data rows for all: the loads stay wave-uniform), so a
    // 5-unit net fills a wave with 8 independent sweeps at the cost of one.
    constexpr float C = -LOG2E, IC = -LN2;
    const int wlane = threadIdx.x & (WAVE - 1);
    const int lane = wlane & ((1 << NRED) - 1);                // lane inside its group = hidden unit
    const int grp = wlane >> NRED;
    const bool gact = grp < ngroups;
    const bool act = gact && (lane < H);
    const int hl = act ? lane : 0;
    if constexpr (!PROP) w_in += (size_t)(gact ? grp : 0) * gstride;
    w_out += (size_t)(gact ? grp : 0) * gstride;
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    const float clr = C * lr;
    const float m0 = (lane == 0) ? 1.0f : 0.0f;
    float w1[I], w2[O], cl[O];
    const float* pnz = nullptr;
    const float* pbase = nullptr;
    float pstep = 0.0f;
    if constexpr (PROP) {
        int slot = pp->pos0 + (gact ? grp : 0);
        if (slot >= pp->ring) slot -= pp->ring;
        pnz = pp->noise + (size_t)slot * pp->nstride;
        pbase = (pp->use_lg && pp->scal[slot * 4] < pp->l_prob) ? pp->w_gd : pp->w_cur;
        pstep = pp->step_w;
    }
    auto win = [&](int e) -> float {
        if constexpr (PROP) return fmaf(pstep, pnz[e], pbase[e]);
        else return w_in[e];
    };
#pragma unroll
    for (int i = 0; i < I; ++i) w1[i] = act ? C * win(i * H + hl) : 0.0f;
#pragma unroll
    for (int o = 0; o < O; ++o) w2[o] = act ? C * win(oW2 + hl * O + o) : 0.0f;
    float b1 = act ? C * win(oB1 + hl) : -1.0e30f;           // inactive lanes: exponent +1e30 -> hid == 0 exactly
#pragma unroll
    for (int o = 0; o < O; ++o) cl[o] = (lane == 0) ? -C * win(oB2 + o) : 0.0f;

    // A lone wave issues one instruction (of any kind) per 4 cycles, so the epoch costs (instructions per row) x 4 cycles
    // and every hazard slot (VALU -> DPP needs two, transcendental -> use one) that holds no useful instruction is lost.
    // The row chain is therefore software-pipelined so that it carries independent work: the W1/B1 update of row n-1 is
    // applied DURING row n, and the pre-activation of row n+1 is started from the weights of row n-1,
    //     z[n+1] = (x[n+1] . W1[n-1] - B1[n-1])  +  lhd[n] (x[n+1] . x[n] + 1),
    // the second factor being a property of the data (column I+1 of the image, filled by the host).  Exact algebra; the
    // rounding differs from the plain chain by O(eps).
    constexpr int RW = I + 2;
    float nb1 = -b1;
    float lhd_p = 0.0f, zp;
    auto zpart = [&](const float (&x)[RW]) {
        float z = fmaf(x[0], w1[0], nb1);
#pragma unroll
        for (int i = 1; i < I; ++i) z = fmaf(x[i], w1[i], z);
        return z;
    };
    constexpr int IPY = sweep_row_stride(I);
    int n = 0;
    if constexpr (TASK == TASK_REG && I == 4 && O == 1 && (NRED == 3 || NRED == 4)) {
        // the reference's time-series nets (4 lags -> <= 8 hidden units -> 1 output): rows 0 .. 4 floor(Ntr/4) - 1 in a
        // hand-scheduled loop (sweep_rows_reg41), whatever is left by the generic code below
        const int iters = Ntr / 4;
        if (iters > 0) {
            sweep_rows_reg41<NRED>(w1, nb1, w2[0], cl[0], m0, lr, clr, gdata, iters);
            n = 4 * iters;
        }
    }
    if constexpr (I <= 8) {
        auto row_step = [&](const float (&xprev)[RW], const float (&x)[RW], const float (&xnext)[RW]) {
            const float z = fmaf(lhd_p, x[I + 1], zp);
            const float e = __builtin_amdgcn_exp2f(z);
    #pragma unroll
            for (int i = 0; i < I; ++i) w1[i] = fmaf(lhd_p, xprev[i], w1[i]);     // row n-1's update
            nb1 += lhd_p;
            zp = zpart(xnext);
            const float hid = __builtin_amdgcn_rcpf(1.0f + e);
            const float dh = fmaf(-hid, hid, hid);                 // hid (1 - hid)
            const float ldh = lr * dh;
            float g = 0.0f;
            float lod[O];
    #pragma unroll
            for (int o = 0; o < O; ++o) {
                const float zo = group_allsum<NRED>(fmaf(hid, w2[o], cl[o]));
                const float out = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zo));
                float t;
                if (TASK == TASK_CLS) t = ((int)x[I] == o) ? 1.0f : 0.0f;   // one-hot(int(y)) (CLS:73-75)
                else t = x[I];                                               // REG: O == 1
                const float od = (t - out) * fmaf(-out, out, out);
                g = fmaf(od, w2[o], g);                                      // pre-update W2 (Q4)
                lod[o] = clr * od;
            }
            lhd_p = g * ldh;
    #pragma unroll
            for (int o = 0; o < O; ++o) {
                w2[o] = fmaf(lod[o], hid, w2[o]);
                cl[o] = fmaf(lod[o], m0, cl[o]);
            }
        };

        // ring of four row buffers: previous, current, next, and the one being fetched (row n+2).  The data image carries
        // two padding rows, so the look-ahead never leaves it.
        float xa[RW], xb[RW], xc[RW], xd[RW];
        const float* pr = xy + (size_t)n * IPY;
        lds_load<RW>(pr, xb);
        lds_load<RW>(pr + IPY, xc);
    #pragma unroll
        for (int i = 0; i < RW; ++i) xa[i] = 0.0f;
        zp = zpart(xb);
        pr += 2 * IPY;
        for (; n + 3 < Ntr; n += 4) {
            lds_load<RW>(pr, xd);
            row_step(xa, xb, xc);
            lds_load<RW>(pr + IPY, xa);
            row_step(xb, xc, xd);
            lds_load<RW>(pr + 2 * IPY, xb);
            row_step(xc, xd, xa);
            lds_load<RW>(pr + 3 * IPY, xc);
            row_step(xd, xa, xb);
            pr += 4 * IPY;
        }
        // tail: up to three rows; afterwards the update of the very last row is still pending
        float xl[RW];
        const int rem = Ntr - n;
        if (rem == 0) {
    #pragma unroll
            for (int i = 0; i < RW; ++i) xl[i] = xa[i];
        } else if (rem == 1) {
            row_step(xa, xb, xc);
    #pragma unroll
            for (int i = 0; i < RW; ++i) xl[i] = xb[i];
        } else if (rem == 2) {
            lds_load<RW>(pr, xd);
            row_step(xa, xb, xc);
            row_step(xb, xc, xd);
    #pragma unroll
            for (int i = 0; i < RW; ++i) xl[i] = xc[i];
        } else {
            lds_load<RW>(pr, xd);
            row_step(xa, xb, xc);
            lds_load<RW>(pr + IPY, xa);
            row_step(xb, xc, xd);
            row_step(xc, xd, xa);
    #pragma unroll
            for (int i = 0; i < RW; ++i) xl[i] = xd[i];
        }
    #pragma unroll
        for (int i = 0; i < I; ++i) w1[i] = fmaf(lhd_p, xl[i], w1[i]);
        b1 = -(nb1 + lhd_p);
    } else {
        // wide input layers: the 2 I independent FMAs of a row already fill the hazard slots, and a ring of four I-wide
        // rows would cost more registers than the deferral saves -- plain chain, two rows in flight
        b1 = -nb1;
        auto row_plain = [&](const float (&x)[I + 1]) {
            float z = fmaf(x[0], w1[0], -b1);
#pragma unroll
            for (int i = 1; i < I; ++i) z = fmaf(x[i], w1[i], z);
            const float hid = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
            const float ldh = lr * fmaf(-hid, hid, hid);
            float g = 0.0f;
            float lod[O];
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const float zo = group_allsum<NRED>(fmaf(hid, w2[o], cl[o]));
                const float out = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zo));
                float t;
                if (TASK == TASK_CLS) t = ((int)x[I] == o) ? 1.0f : 0.0f;   // one-hot(int(y)) (CLS:73-75)
                else t = x[I];
                const float od = (t - out) * fmaf(-out, out, out);
                g = fmaf(od, w2[o], g);                                      // pre-update W2 (Q4)
                lod[o] = clr * od;
            }
            const float lhd = g * ldh;
#pragma unroll
            for (int o = 0; o < O; ++o) {
                w2[o] = fmaf(lod[o], hid, w2[o]);
                cl[o] = fmaf(lod[o], m0, cl[o]);
            }
#pragma unroll
            for (int i = 0; i < I; ++i) w1[i] = fmaf(lhd, x[i], w1[i]);
            b1 -= lhd;
        };
        float ya[I + 1], yb[I + 1];
        lds_load<I + 1>(xy, ya);
        lds_load<I + 1>(xy + IPY, yb);
        const float* pr = xy + 2 * IPY;
        int m = 0;
        for (; m + 1 < Ntr; m += 2) {
            row_plain(ya);
            lds_load<I + 1>(pr, ya);
            row_plain(yb);
            lds_load<I + 1>(pr + IPY, yb);
            pr += 2 * IPY;
        }
        if (m < Ntr) row_plain(ya);
    }

    if (act) {
#pragma unroll
        for (int i = 0; i < I; ++i) w_out[i * H + lane] = IC * w1[i];
#pragma unroll
        for (int o = 0; o < O; ++o) w_out[oW2 + lane * O + o] = IC * w2[o];
        w_out[oB1 + lane] = IC * b1;
    }
    if (lane == 0 && gact) {
#pragma unroll
        for (int o = 0; o < O; ++o) w_out[oB2 + o] = -IC * cl[o];
    }
}

template <int TASK, int I, int O>
__device__ __forceinline__ void sgd_sweep_select(const float* w_in, float* w_out, const float* xy, const float* gdata, int Ntr,
                                                 int H, float lr) {
    // (no 4-lane variant: the packed schedule runs nets of <= 8 hidden units in 8-lane groups, and every schedule must
    // commit the same chain bit for bit)
    if (H <= 8) sgd_sweep<TASK, I, O, 3>(w_in, w_out, xy, gdata, Ntr, H, lr);
    else if (H <= 16) sgd_sweep<TASK, I, O, 4>(w_in, w_out, xy, gdata, Ntr, H, lr);
    else if (H <= 32) sgd_sweep<TASK, I, O, 5>(w_in, w_out, xy, gdata, Ntr, H, lr);
    else sgd_sweep<TASK, I, O, 6>(w_in, w_out, xy, gdata, Ntr, H, lr);
}
// The epoch is called out of line from the cooperative and the multi-CU speculative kernels: inlined (twice, four lane-group
// variants each) its registers pushed the kernels' own loop state into scratch even in runs that never take a Langevin step --
// rocprofv3 WRITE_SIZE: 2.9x the algorithmic bytes on the Ionosphere workload, 2.1x on Iris (51 VGPRs spilled, 208 B of
// scratch per lane written back every launch), 24 spilled VGPRs in the Mackey-Glass kernel.  An epoch is 10^4..10^5 cycles: a
// real call costs nothing.  The 4-H-1 time-series nets with H <= 16 are the exception: their row loop is the hand-scheduled
// asm with fixed physical registers, some of them callee-saved in the AMDGPU calling convention (v40-v47, v56-v63, s36-s72),
// which a callee would have to save to a stack frame -- that loop stays inline (it needs 27 VGPRs, no spill comes from it), only
// the wider lane groups (H > 16) go through the call.  The packed kernel always inlines its lane-group variant.
template <int TASK, int I, int O>
__device__ __attribute__((noinline)) void sgd_sweep_call(const float* w_in, float* w_out, const float* xy, const float* gdata,
                                                         int Ntr, int H, float lr) {
    if constexpr (TASK == TASK_REG && I == 4 && O == 1) {
        if (H <= 32) sgd_sweep<TASK, I, O, 5>(w_in, w_out, xy, gdata, Ntr, H, lr);
        else sgd_sweep<TASK, I, O, 6>(w_in, w_out, xy, gdata, Ntr, H, lr);
    } else {
        sgd_sweep_select<TASK, I, O>(w_in, w_out, xy, gdata, Ntr, H, lr);
    }
}
template <int TASK, int I, int O>
__device__ __forceinline__ void sgd_sweep_dispatch(const float* w_in, float* w_out, const float* xy, const float* gdata, int Ntr,
                                                   int H, float lr) {
    if constexpr (TASK == TASK_REG && I == 4 && O == 1) {
        if (H <= 8) sgd_sweep<TASK, I, O, 3>(w_in, w_out, xy, gdata, Ntr, H, lr);
        else if (H <= 16) sgd_sweep<TASK, I, O, 4>(w_in, w_out, xy, gdata, Ntr, H, lr);
        else sgd_sweep_call<TASK, I, O>(w_in, w_out, xy, gdata, Ntr, H, lr);
    } else {
        sgd_sweep_call<TASK, I, O>(w_in, w_out, xy, gdata, Ntr, H, lr);
    }
}

// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// packed forward layout, hidden units in PAIRS so that the forward pass runs on v_pk_fma_f32: pair hp = units (2hp, 2hp+1)
// occupies 2 FWS floats, element c of unit j at [2c + j] with c = 0..I-1: W1[c][h], c = I: B1[h], c = I+1+o: W2[h][o];
// an absent odd unit is all zeros (its hid = 0.5 meets W2 = 0).  B2[0..O-1] follows at fw[fw_pairs(H) * 2 FWS].
// Built by all threads from a flat w.
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int fw_pairs(int H) { return (H + 1) >> 1; }
__host__ __device__ inline size_t fw_floats(int H, int FWS) { return ((size_t)2 * fw_pairs(H) + 1) * FWS; }

// Nets with few inputs (I < 8) and an odd or tiny hidden layer keep one unit per row of FWS floats (fw[h] = { W1[0..I-1][h], B1[h], W2[h][0..O-1] }, B2 at
// fw[H*FWS]): their forward pass is sigmoid-bound and padding an odd H to a pair costs more than the packed FMAs save.
// An even hidden layer of at least 8 units takes the pairs too (Iris 4-12-3).  The rule is a function of (I, H) only, so
// build_fw and its readers agree without passing a flag around.
template <int I>
struct FwLayout {
    static __host__ __device__ constexpr bool pairs(int H) { return I >= 8 || (H >= 8 && (H & 1) == 0); }
};

template <int I, int O, bool WL = false>
__device__ __forceinline__ void build_fw(const float* __restrict__ w, float* __restrict__ fw, int H, int FWS) {
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    constexpr int K = I + 1 + O;
    if (FwLayout<I>::pairs(H)) {
        const int HP2 = 2 * fw_pairs(H);
        for (int e = gtid<WL>(); e < HP2 * K; e += gsize<WL>()) {
            const int h = e / K, c = e - h * K;
            float v = 0.0f;
            if (h < H) {
                if (c < I) v = w[c * H + h];
                else if (c == I) v = w[oB1 + h];
                else v = w[oW2 + h * O + (c - I - 1)];
            }
            fw[(h >> 1) * 2 * FWS + 2 * c + (h & 1)] = v;
        }
        if (gtid<WL>() < O) fw[HP2 * FWS + gtid<WL>()] = w[oB2 + gtid<WL>()];
    } else {
        for (int e = gtid<WL>(); e < H * K; e += gsize<WL>()) {
            const int h = e / K, c = e - h * K;
            float v;
            if (c < I) v = w[c * H + h];
            else if (c == I) v = w[oB1 + h];
            else v = w[oW2 + h * O + (c - I - 1)];
            fw[h * FWS + c] = v;
        }
        if (gtid<WL>() < O) fw[H * FWS + gtid<WL>()] = w[oB2 + gtid<WL>()];
    }
}

// Random-walk proposal and its forward image in one pass: every weight appears in the image exactly once, so the thread that
// places element e also forms w_prop[idx] = w[idx] + step * noise[idx] (the value build_fw would read back).
template <int I, int O>
__device__ __forceinline__ void propose_build_fw(const float* __restrict__ w, const float* __restrict__ noise, float step,
                                                 float* __restrict__ w_prop, float* __restrict__ fw, int H, int FWS) {
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    constexpr int K = I + 1 + O;
    const bool pairs = FwLayout<I>::pairs(H);
    const int HR = pairs ? 2 * fw_pairs(H) : H;
    for (int e = threadIdx.x; e < HR * K; e += blockDim.x) {
        const int h = e / K, c = e - h * K;
        float v = 0.0f;
        if (h < H) {
            const int idx = (c < I) ? c * H + h : (c == I) ? oB1 + h : oW2 + h * O + (c - I - 1);
            v = fmaf(step, noise[idx], w[idx]);
            w_prop[idx] = v;
        }
        fw[pairs ? (h >> 1) * 2 * FWS + 2 * c + (h & 1) : h * FWS + c] = v;
    }
    if (threadIdx.x < O) {
        const float v = fmaf(step, noise[oB2 + threadIdx.x], w[oB2 + threadIdx.x]);
        w_prop[oB2 + threadIdx.x] = v;
        fw[HR * FWS + threadIdx.x] = v;
    }
}

// Ordering key of np.argmax over the reference's FLOAT64 sigmoid outputs, computed from the fp32 pre-activation z.
// sigmoid is monotone, so below z = 30 the key is z itself (fp32 outputs saturating to 1.0f must not tie where float64
// outputs still differ).  From z = 30 on, float64 itself quantises: 1 + e^-z is rounded to a multiple of 2^-52, outputs
// tie exactly when that multiple k = rint(e^-z 2^52) ties, and for z >= 53 ln 2 = 36.74 every output is exactly 1.0
// (k = 0): np.argmax then returns the FIRST such class.  Below z = -709.78 np.exp(-z) overflows and the output is 0.0.
// Returns (regime, value): compared lexicographically, full fp32 resolution of z inside the ordinary regime.
struct ArgKey { int hi; float lo; };
__device__ __forceinline__ ArgKey argmax_key(float z) {
    ArgKey k;
    if (z >= 30.0f) { k.hi = 2; k.lo = -rintf(__builtin_amdgcn_exp2f(fmaf(-LOG2E, z, 52.0f))); }
    else if (z < -709.78f) { k.hi = 0; k.lo = 0.0f; }
    else { k.hi = 1; k.lo = z; }
    return k;
}
__device__ __forceinline__ bool argkey_greater(const ArgKey& a, const ArgKey& b) {
    return (a.hi > b.hi) || (a.hi == b.hi && a.lo > b.lo);
}

// sums produced by one evaluation of (train ++ test) under a weight vector
struct EvalSums {
    float a_tr, b_tr, c_tr;   // REG: SSE, -, -      CLS: sum log p(y), sum (pred-y)^2, #correct   (train rows)
    float a_te, b_te, c_te;   // same for test rows
};

// Work-group sums of the per-lane row scores, returned in every thread: wave DPP reduction, then a fixed-order sum of the
// per-wave partials through LDS.  LEAN (the cooperative kernel's step loop): the caller guarantees that nobody still reads
// red[] (a barrier separates the previous readers from this call), and `extra` -- one more per-thread partial, the sum of
// squares of the proposal for the prior -- rides along in the same rows, so a step has ONE reduction instead of three.
template <int TASK, bool WL, bool LEAN>
__device__ __forceinline__ EvalSums reduce_eval(float a_tr, float b_tr, float c_tr, float a_te, float b_te, float c_te,
                                                float* __restrict__ red, float& extra) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    // all wave sums stage by stage (wave_allsum_n: the same operations per value as wave_allsum, the DPP latencies shared)
    if (TASK == TASK_CLS) {
        float v[LEAN ? 7 : 6];
        v[0] = a_tr; v[1] = a_te; v[2] = b_tr; v[3] = c_tr; v[4] = b_te; v[5] = c_te;
        if (LEAN) v[LEAN ? 6 : 0] = extra;
        wave_allsum_n(v);
        a_tr = v[0]; a_te = v[1]; b_tr = v[2]; c_tr = v[3]; b_te = v[4]; c_te = v[5];
        if (LEAN) extra = v[LEAN ? 6 : 0];
    } else {
        float v[LEAN ? 3 : 2];
        v[0] = a_tr; v[1] = a_te;
        if (LEAN) v[LEAN ? 2 : 0] = extra;
        wave_allsum_n(v);
        a_tr = v[0]; a_te = v[1];
        if (LEAN) extra = v[LEAN ? 2 : 0];
    }
    EvalSums s;
    if (WL || nw == 1) {
        s.a_tr = a_tr; s.b_tr = b_tr; s.c_tr = c_tr; s.a_te = a_te; s.b_te = b_te; s.c_te = c_te;
        return s;
    }
    if (!LEAN) __syncthreads();                            // red[] may still be read from the previous use
    if (lane == 0) {
        float* r = red + wave * 8;
        *reinterpret_cast<float4*>(r) = make_float4(a_tr, b_tr, c_tr, a_te);
        *reinterpret_cast<float4*>(r + 4) = make_float4(b_te, c_te, LEAN ? extra : 0.0f, 0.0f);
    }
    __syncthreads();
    s.a_tr = s.b_tr = s.c_tr = s.a_te = s.b_te = s.c_te = 0.f;
    float ex = 0.f;
    for (int k = 0; k < nw; ++k) {
        const float4 u = *reinterpret_cast<const float4*>(red + k * 8), v = *reinterpret_cast<const float4*>(red + k * 8 + 4);
        s.a_tr += u.x; s.b_tr += u.y; s.c_tr += u.z; s.a_te += u.w; s.b_te += v.x; s.c_te += v.y; ex += v.z;
    }
    if (LEAN) extra = ex;
    return s;
}

// R2/R3/R6: one lane per data row; the block's threads stride over train ++ test.  Returns block-wide sums in every
// thread (reduce_eval).
template <int TASK, int I, int O, bool WL = false, bool LEAN = false>
__device__ __forceinline__ EvalSums eval_rows(const float* __restrict__ fw, const float* __restrict__ xy, int IPY,
                                              int FWS, int H, int Ntr, int Nall, float* __restrict__ red, float& extra) {
    float a_tr = 0.f, b_tr = 0.f, c_tr = 0.f, a_te = 0.f, b_te = 0.f, c_te = 0.f;
    constexpr int K = I + 1 + O;
    // RB data rows of one lane are kept in registers while the hidden units stream by: the packed weights of unit h are
    // read from LDS once per RB rows, and the RB independent chains hide the exp/rcp latency of each other
    constexpr int RB = ((I + 1) * 8 <= 64) ? 8 : ((I + 1) * 4 <= 64) ? 4 : ((I + 1) * 2 <= 80) ? 2 : 1;
    float b2[O];
#pragma unroll
    for (int o = 0; o < O; ++o) b2[o] = fw[(FwLayout<I>::pairs(H) ? 2 * fw_pairs(H) : H) * FWS + o];
    const int stride = gsize<WL>();
    // one block = RBK rows of this lane (rows tid + (b0 + b) stride); the last rows of a lane take the smaller blockings,
    // so a small data set spread over many lanes costs one row per lane, not RB.  Every lane adds its rows in ascending
    // order whatever the blocking, so the sums do not depend on it.
    auto block = [&](auto rbk, int b0) {
        constexpr int RBK = decltype(rbk)::value;
        constexpr int UNR = LEAN ? 2 : 1;
        const int n0 = gtid<WL>() + b0 * stride;
        const int nc = n0 < Nall ? n0 : 0;
        float x[RBK][I + 1];
        float acc[RBK][O];
#pragma unroll
        for (int b = 0; b < RBK; ++b) {
            const int n = n0 + b * stride;
            lds_load<I + 1>(xy + (n < Nall ? n : nc) * IPY, x[b]);
        }
        if (FwLayout<I>::pairs(H)) {
            f32x2 acc2[RBK][O];                            // {even units, odd units}: joined after the hidden loop
#pragma unroll
            for (int b = 0; b < RBK; ++b)
#pragma unroll
                for (int o = 0; o < O; ++o) acc2[b][o] = f32x2{0.0f, 0.0f};
            const int HP = fw_pairs(H);
            constexpr int CH = 8, NF = I / CH, RQ = K - NF * CH;   // inputs in chunks of 8 pairs: bounded register footprint
#pragma unroll UNR                                        // cooperative step loop: the next pair's weights arrive while this pair computes
            for (int hp = 0; hp < HP; ++hp) {
                const float* row = fw + hp * 2 * FWS;      // wave-uniform address: broadcast reads
                f32x2 z[RBK];
#pragma unroll
                for (int b = 0; b < RBK; ++b) z[b] = f32x2{0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    float f[2 * CH];
                    lds_load<2 * CH>(row + 2 * CH * q, f);
#pragma unroll
                    for (int b = 0; b < RBK; ++b)
#pragma unroll
                        for (int i = 0; i < CH; ++i)
                            z[b] = __builtin_elementwise_fma(f32x2{x[b][CH * q + i], x[b][CH * q + i]}, f32x2{f[2 * i], f[2 * i + 1]}, z[b]);
                }
                float f[2 * RQ];                            // the remaining inputs, B1, W2
                lds_load<2 * RQ>(row + 2 * CH * NF, f);
#pragma unroll
                for (int b = 0; b < RBK; ++b) {
#pragma unroll
                    for (int i = NF * CH; i < I; ++i)
                        z[b] = __builtin_elementwise_fma(f32x2{x[b][i], x[b][i]}, f32x2{f[2 * (i - NF * CH)], f[2 * (i - NF * CH) + 1]}, z[b]);
                    const f32x2 zz = z[b] - f32x2{f[2 * (I - NF * CH)], f[2 * (I - NF * CH) + 1]};
                    const f32x2 hid = f32x2{sigmoidf_fast(zz.x), sigmoidf_fast(zz.y)};
#pragma unroll
                    for (int o = 0; o < O; ++o)
                        acc2[b][o] = __builtin_elementwise_fma(hid, f32x2{f[2 * (I + 1 + o - NF * CH)], f[2 * (I + 1 + o - NF * CH) + 1]}, acc2[b][o]);
                }
            }
#pragma unroll
            for (int b = 0; b < RBK; ++b)
#pragma unroll
                for (int o = 0; o < O; ++o) acc[b][o] = (acc2[b][o].x + acc2[b][o].y) - b2[o];
        } else {
#pragma unroll
            for (int b = 0; b < RBK; ++b)
#pragma unroll
                for (int o = 0; o < O; ++o) acc[b][o] = -b2[o];
#pragma unroll UNR
            for (int h = 0; h < H; ++h) {
                float f[K];
                lds_load<K>(fw + h * FWS, f);              // wave-uniform address: broadcast reads
#pragma unroll
                for (int b = 0; b < RBK; ++b) {
                    float z = -f[I];
#pragma unroll
                    for (int i = 0; i < I; ++i) z = fmaf(x[b][i], f[i], z);
                    const float hid = sigmoidf_fast(z);
#pragma unroll
                    for (int o = 0; o < O; ++o) acc[b][o] = fmaf(hid, f[I + 1 + o], acc[b][o]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < RBK; ++b) {
            const int n = n0 + b * stride;
            if (n >= Nall) continue;
            const float y = x[b][I];
            float a, bb = 0.f, c = 0.f;
            if (TASK == TASK_REG) {
                const float d = y - sigmoidf_fast(acc[b][0]);
                a = d * d;
            } else {
                ArgKey best = argmax_key(acc[b][0]);
                float se = 0.0f, oy = 0.0f;
                int arg = 0;
                const int yi = (int)y;
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const float out = sigmoidf_fast(acc[b][o]);
                    const ArgKey key = argmax_key(acc[b][o]);
                    if (argkey_greater(key, best)) { best = key; arg = o; }   // np.argmax(out): first maximum (CLS:55)
                    se += expf_fast(out);                              // softmax of the sigmoid outputs (Q3)
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
    };
    const int cnt = (Nall + stride - 1) / stride;          // rows of the busiest lane (uniform over the group)
    int b0 = 0;
    for (; cnt - b0 >= RB; b0 += RB) block(std::integral_constant<int, RB>{}, b0);
    if (RB >= 8 && cnt - b0 >= 4) { block(std::integral_constant<int, 4>{}, b0); b0 += 4; }
    if (RB >= 4 && cnt - b0 >= 2) { block(std::integral_constant<int, 2>{}, b0); b0 += 2; }
    if (RB >= 2 && cnt - b0 >= 1) { block(std::integral_constant<int, 1>{}, b0); b0 += 1; }
    return reduce_eval<TASK, WL, LEAN>(a_tr, b_tr, c_tr, a_te, b_te, c_te, red, extra);
}

template <int TASK, int I, int O, bool WL = false>
__device__ __forceinline__ EvalSums eval_rows(const float* __restrict__ fw, const float* __restrict__ xy, int IPY,
                                              int FWS, int H, int Ntr, int Nall, float* __restrict__ red) {
    float none = 0.0f;
    return eval_rows<TASK, I, O, WL, false>(fw, xy, IPY, FWS, H, Ntr, Nall, red, none);
}

// block-wide sum of one value per thread, returned in every thread
template <bool WL = false>
__device__ __forceinline__ float block_sum(float s, float* __restrict__ red) {
    s = wave_allsum(s);
    const int nw = blockDim.x >> 6;
    if (WL || nw == 1) return s;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 8] = s;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < nw; ++k) t += red[k * 8];
    return t;
}

// fx - y of one row for the O == 1 regression net (chain start-up, REG:266-270)
template <int I, int O>
__device__ __forceinline__ float reg_residual(const float* __restrict__ row, const float* __restrict__ fw, int FWS,
                                              int H) {
    constexpr int K = I + 1 + O;
    float x[I + 1];
    lds_load<I + 1>(row, x);
    float acc;
    if (FwLayout<I>::pairs(H)) {
        f32x2 acc2 = f32x2{0.0f, 0.0f};
        for (int hp = 0; hp < fw_pairs(H); ++hp) {
            float f[2 * K];
            lds_load<2 * K>(fw + hp * 2 * FWS, f);
            f32x2 z = f32x2{-f[2 * I], -f[2 * I + 1]};
#pragma unroll
            for (int i = 0; i < I; ++i) z = __builtin_elementwise_fma(f32x2{x[i], x[i]}, f32x2{f[2 * i], f[2 * i + 1]}, z);
            acc2 = __builtin_elementwise_fma(f32x2{sigmoidf_fast(z.x), sigmoidf_fast(z.y)}, f32x2{f[2 * (I + 1)], f[2 * (I + 1) + 1]}, acc2);
        }
        acc = (acc2.x + acc2.y) - fw[2 * fw_pairs(H) * FWS];
    } else {
        acc = -fw[H * FWS];
        for (int h = 0; h < H; ++h) {
            float f[K];
            lds_load<K>(fw + h * FWS, f);
            float z = -f[I];
#pragma unroll
            for (int i = 0; i < I; ++i) z = fmaf(x[i], f[i], z);
            acc = fmaf(sigmoidf_fast(z), f[I + 1], acc);
        }
    }
    return sigmoidf_fast(acc) - x[I];
}

// block-wide sum of squares of a vector in LDS (prior, REG:219)
template <bool WL = false>
__device__ __forceinline__ float block_sumsq(const float* __restrict__ v, int n, float* __restrict__ red) {
    float s = 0.f;
    for (int j = gtid<WL>(); j < n; j += gsize<WL>()) s = fmaf(v[j], v[j], s);
    return block_sum<WL>(s, red);
}

// block-wide sum of squared differences (Langevin proposal ratio, REG:336-346)
template <bool WL = false>
__device__ __forceinline__ float block_sumsq_diff(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                  float* __restrict__ red) {
    float s = 0.f;
    for (int j = gtid<WL>(); j < n; j += gsize<WL>()) { const float d = a[j] - b[j]; s = fmaf(d, d, s); }
    return block_sum<WL>(s, red);
}

// likelihood / rmse / accuracy from the sums (R6: REG:200-205, CLS:209-222, 200-207); untempered log-likelihood.  In two
// parts because only the likelihood feeds the MH test: the cooperative step loop forms the scores after the decision, and
// only in the wave that writes them.
template <int TASK>
__device__ __forceinline__ float finish_loglik(const EvalSums& s, int Ntr, float eta) {
    if (TASK == TASK_REG) {
        // sum_n [-0.5 log(2 pi tau^2) - 0.5 (y-fx)^2 / tau^2], tau^2 = exp(eta)
        // explicit fused operation (as in prior_value): every kernel must form the same bits whatever surrounds the call
        return fmaf(-0.5f * s.a_tr, expf_fast(-eta), -0.5f * (float)Ntr * (LOG_2PI + eta));
    }
    return s.a_tr;
}
template <int TASK>
__device__ __forceinline__ void finish_scores(const EvalSums& s, int Ntr, int Nte, float& rmse_tr, float& rmse_te, float& acc_tr,
                                              float& acc_te) {
    if (TASK == TASK_REG) {
        rmse_tr = __builtin_amdgcn_sqrtf(s.a_tr / (float)Ntr);
        rmse_te = __builtin_amdgcn_sqrtf(s.a_te / (float)Nte);
        acc_tr = 0.f;
        acc_te = 0.f;
    } else {
        rmse_tr = __builtin_amdgcn_sqrtf(s.b_tr / (float)Ntr);
        rmse_te = __builtin_amdgcn_sqrtf(s.b_te / (float)Nte);
        acc_tr = 100.0f * (s.c_tr / (float)Ntr);
        acc_te = 100.0f * (s.c_te / (float)Nte);
    }
}
// REC (the step loops): regression has no accuracy -- acc_train is identically 0 in the reference (REG:403) -- so the slot
// that carries it through the commit into the scalar trace row (TR_ACC_TR) carries the eta the proposal was evaluated with
// instead: the recorded row of an accepted step then holds the chain's new eta, a rejected step repeats the recorded one.
// ptnn_get_traces hands out zeros for a regression's acc_train; ptnn_get_trace_rows shows the raw row (the parity tests set
// the oracle's state from it after every accepted step, tests/parity.py: follow_device_run).
template <int TASK, bool REC = false>
__device__ __forceinline__ void finish_eval(const EvalSums& s, int Ntr, int Nte, float eta, float& loglik,
                                            float& rmse_tr, float& rmse_te, float& acc_tr, float& acc_te) {
    loglik = finish_loglik<TASK>(s, Ntr, eta);
    finish_scores<TASK>(s, Ntr, Nte, rmse_tr, rmse_te, acc_tr, acc_te);
    if (REC && TASK == TASK_REG) acc_tr = eta;
}

// Langevin proposal ratio (REG:336-347, Q6): [-0.5 d1 / step^2 + 0.5 d2] / adapttemp with d1 = |w - w_prop_gd|^2 and
// d2 = |noise|^2; one explicit fused operation, the same bits in every kernel
__device__ __forceinline__ float langevin_ratio(float d1, float d2, float step_w, float adapttemp) {
    return fmaf(0.5f, d2, -0.5f * d1 / (step_w * step_w)) / adapttemp;
}

// R7 prior_likelihood (REG:215-221 / CLS:224-230); prior_c = part1, log tau^2 = eta
template <int TASK>
__device__ __forceinline__ float prior_value(const SegParams& p, float sumsq, float eta) {
    // explicit fused operations: the value must not depend on how the compiler contracts the expression in each kernel
    float v = fmaf(-p.inv_2sig2, sumsq, p.prior_c);
    if (TASK == TASK_REG) v = fmaf(-p.nu2, expf_fast(-eta), fmaf(-(1.0f + p.nu1), eta, v));
    return v;
}

// LDS carve-up shared by the kernels
struct Lds {
    float* xy; float* w_cur; float* w_prop; float* w_gd; float* w_pgd; float* rec_w; float* noise; float* fw;
    float* red; float* scal;
};
// lg = false (a launch without Langevin proposals): the two cached SGD epochs are not carved at all
__device__ __forceinline__ Lds carve(float* base, int Nall, int IPY, int PS, int H, int FWS, bool lg = true, bool xy = true) {
    Lds l;
    float* q = base;
    l.xy = q; q += xy ? (Nall + 2) * IPY : 0;
    l.w_cur = q; q += PS;                                  // w_cur, w_prop, rec_w, w_gd, w_pgd in this order: the cooperative
    l.w_prop = q; q += PS;                                 // step loop rotates them by offset
    l.rec_w = q; q += PS;
    l.w_gd = q; l.w_pgd = q + (lg ? PS : 0); q += lg ? 2 * PS : 0;
    l.noise = q; l.scal = q + PS; q += 2 * (PS + 8);        // two tapes {noise[PS], scal[8]}: a step draws the next one's
    l.fw = q; q += fw_floats(H, FWS);
    l.red = q; q += MAX_WAVES * 8;
    return l;
}
__host__ __device__ inline size_t lds_floats(int Nall, int IPY, int PS, int H, int FWS, bool lg = true, bool xy = true) {
    return (xy ? (size_t)(Nall + 2) * IPY : 0) + (lg ? 7 : 5) * (size_t)PS + fw_floats(H, FWS) + MAX_WAVES * 8 + 16;
}

// random tape of one step: noise[0..P) and scal[0..2] = {lx, u, n_eta}
template <bool WL = false>
__device__ __forceinline__ void tape_step(const SegParams& p, int gid, int step, float* __restrict__ noise,
                                          float* __restrict__ scal) {
    const int nq = (p.P + 3) >> 2;
    for (int q = gtid<WL>(); q <= nq; q += gsize<WL>()) {
        const bool sc = (q == nq);
        uint32_t x[4];
        philox4x32_10(sc ? 0u : (uint32_t)q, (uint32_t)step, p.noise_shared ? 0u : (uint32_t)gid, sc ? STREAM_STEP : STREAM_WNOISE, p.seed_lo,
                      p.seed_hi, x);
        float n0, n1, n2, n3;
        box_muller(x[0], x[1], n0, n1);
        box_muller(x[2], x[3], n2, n3);
        if (sc) {
            scal[0] = u23(x[0]);
            scal[1] = u23(x[1]);
            scal[2] = n2;
        } else {
            *reinterpret_cast<float4*>(noise + 4 * q) = make_float4(n0, n1, n2, n3);
        }
    }
}

// swap_rule 1 (textbook exchange) works on untempered quantities: the likelihood held by the chain is tempered
// by the adapttemp of the last executed step (T before the switch step, 1 from it on)
__device__ __forceinline__ void post_raw(const SegParams& p, int gid, float lik, float prior_cur, float T, int last_step) {
    if (p.L_raw == nullptr) return;
    const float a = (p.switch_step >= 0 && last_step >= p.switch_step) ? 1.0f : T;
    p.L_raw[gid] = lik * a;
    p.prior_post[gid] = prior_cur;
}

// R14 chain start-up: eta0 = log var(fx_train(w0) - y) (REG:270), prior (REG:280), tempered likelihood (REG:284).
// WL = false: the whole work-group shares the rows; WL = true: the calling wave does it alone (the speculative
// schedule uses wave 0 so that the result does not depend on the number of waves).
template <int TASK, int I, int O, bool WL = false>
__device__ __forceinline__ void chain_startup(const SegParams& p, const float* xy, const float* w_cur, float* fw, float* red,
                                              float T, float& eta, float& lik, float& prior_cur) {
    const int tid = gtid<WL>(), nthr = gsize<WL>(), H = p.H, Nall = p.Ntr + p.Nte;
    build_fw<I, O, WL>(w_cur, fw, H, p.FWS);
    gsync<WL>();
    if (TASK == TASK_REG) {
        // population variance of the residuals (np.var), two passes over the train rows
        float s1 = 0.f;
        for (int n = tid; n < p.Ntr; n += nthr) s1 += reg_residual<I, O>(xy + n * p.IPY, fw, p.FWS, H);
        const float mean = block_sum<WL>(s1, red) / (float)p.Ntr;
        float s2 = 0.f;
        for (int n = tid; n < p.Ntr; n += nthr) {
            const float d = reg_residual<I, O>(xy + n * p.IPY, fw, p.FWS, H) - mean;
            s2 = fmaf(d, d, s2);
        }
        eta = logf_fast(block_sum<WL>(s2, red) / (float)p.Ntr);
    }
    const EvalSums s0 = eval_rows<TASK, I, O, WL>(fw, xy, p.IPY, p.FWS, H, p.Ntr, Nall, red);
    float ll, r1, r2, a1, a2;
    finish_eval<TASK>(s0, p.Ntr, p.Nte, eta, ll, r1, r2, a1, a2);
    lik = ll / T;
    const float ss = block_sumsq<WL>(w_cur, p.P, red);
    prior_cur = prior_value<TASK>(p, ss, eta);
}
