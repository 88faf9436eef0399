// ptnn_dev_coop.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// cooperative schedule: matrix-core forward passes (exact fp32 / split bf16 operands) and segment_body.

// ------------------------------------------------------------------------------------------------
// The segment kernel: MH steps [step_begin, step_begin + n_steps) of every local replica; block = replica.
// step_begin == 0 also performs the chain start-up (REG:266-285).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Forward pass of a mid-sized net (24 <= H <= 64) on the matrix cores, cooperative schedule.  The lane-per-row VALU pass
// re-reads every weight from LDS with all 64 lanes on one address (one 16-byte broadcast read per 4 weights and wave): at
// 34 inputs x 50 hidden units the LDS pipe, not the VALU, bounds it.  Here the product is taken transposed,
// Z^T[h][n] = sum_k W1[k][h] X[n][k] with v_mfma_f32_32x32x2_f32: A = W1 straight from the flat proposal in LDS (lane = hidden
// unit: conflict-free 4-byte reads, all k-steps of a tile fetched in one batch), B = the transposed data image from L2
// (lane = data row), two hidden tiles with independent accumulators in flight; in the 32x32 accumulator a lane is a data
// row and the 16 registers are hidden units, so bias, sigmoid and the W2 product are applied in place (same epilogue as
// eval_rows_mfma).  Exact fp32 (k-ordered fma chains).  A partial last tile is masked: absent units get W1 = W2 = 0.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS floats the MFMA forward pass of the cooperative schedule adds: the transposed data image and the per-tile partial
// output sums of every row
__host__ __device__ inline size_t mfma_coop_lds_floats(int I, int O, int H, int Npad) {
    return (size_t)I * Npad + (size_t)((H + 31) >> 5) * Npad * O;
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ short f32_to_bf16(float f) {            // round to nearest even; inputs are finite
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// ------------------------------------------------------------------------------------------------
// Split-operand forward pass (cooperative schedule, fw_mfma == 2).  Measured on gfx950 (profiles/r03_micro_mfma_valu_overlap.txt):
// v_mfma_f32_32x32x2_f32 runs at the packed-fp32 VALU rate AND keeps the SIMD's vector issue to itself for its 64 cycles -- the
// sigmoid / W2 epilogue cannot hide behind it, the two add up.  v_mfma_f32_32x32x16_bf16 covers 8 x the k extent in half the
// cycles and holds the vector issue for 8 of its 32.  So every fp32 operand is split into three bf16 terms, x = hi + mid + lo
// (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 24 significant bits, the residual is below 2^-26 |x|), and a
// product keeps the six terms down to 2^-24: hi hi, hi mid, mid hi, mid mid, hi lo, lo hi, accumulated in fp32 by the matrix
// core, small terms first.  The k values a multiple of 16 leaves over (34 = 2 x 16 + 2) go through one exact fp32 instruction
// per pair.  The result is a sum of the same products in another order with errors of the size of fp32 rounding: the same
// accuracy against the float64 oracle as the exact pass, NOT bit-identical to the VALU pass (forward_bf16 = 2 keeps that one).
// Data image: split once per launch into LDS, rows of 16 KB bf16 (k contiguous: one ds_read_b128 per operand and k-step), the
// 16-byte chunks XOR-swizzled by the row so that 16 consecutive rows hit 16 different bank groups.  Weights: split once per
// step by the pass that forms the proposal.
// ------------------------------------------------------------------------------------------------
template <int I> struct SplitK {
    static constexpr int KB0 = I / 16, REM0 = I - 16 * KB0;
    static constexpr bool PADLAST = REM0 >= 7;                 // a zero-padded bf16 k-step (6 instructions) beats >= 4 fp32 ones
    static constexpr int KB = KB0 + (PADLAST ? 1 : 0);          // bf16 k-steps of 16
    static constexpr int KBF = 16 * KB;                         // k extent they cover
    static constexpr int REM = PADLAST ? 0 : REM0;              // k values left to the exact fp32 instruction
    static constexpr int KR = (REM + 1) / 2;                    // its k-steps of 2
    static constexpr int CH = 2 * KB;                           // 16-byte chunks per image row
    static constexpr bool OK = (KB == 1 || KB == 2 || KB == 4);
};
// LDS floats of the split images: data {3 levels x Npad rows}, remainder columns (fp32, transposed), labels, weights
// {3 levels x Hpad rows}, per-tile partial sums
template <int I>
__host__ __device__ inline size_t mfma_split_lds_floats(int O, int H, int Npad) {
    typedef SplitK<I> K;
    const int Hpad = ((H + 31) >> 5) << 5;
    return (size_t)3 * Npad * K::CH * 4 + (size_t)2 * K::KR * Npad + (size_t)Npad + (size_t)3 * Hpad * K::CH * 4 + (size_t)(Hpad >> 5) * Npad * O;
}
struct SplitLds { uint4* xs; float* xr; float* ylab; uint4* as; float* part; };
template <int I>
__device__ __forceinline__ SplitLds carve_split(float* base, int O, int H, int Npad) {
    typedef SplitK<I> K;
    const int Hpad = ((H + 31) >> 5) << 5;
    SplitLds l;
    float* q = base;
    l.xs = reinterpret_cast<uint4*>(q); q += (size_t)3 * Npad * K::CH * 4;
    l.xr = q; q += (size_t)2 * K::KR * Npad;
    l.ylab = q; q += Npad;
    l.as = reinterpret_cast<uint4*>(q); q += (size_t)3 * Hpad * K::CH * 4;
    l.part = q;
    return l;
}
template <int CH> __device__ __forceinline__ int split_chunk(int row, int c) { return c ^ ((row / (16 / CH)) & (CH - 1)); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two floats -> their bf16 roundings (nearest even) packed {lo16 = first, hi16 = second}: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// image[level][row][k]: elements (row, k) and (row, k + 1), k even, of all three levels.  x - hi and (x - hi) - mid are exact in fp32.
template <int CH>
__device__ __forceinline__ void split_store_pair(uint4* img, int rows, int row, int k, float x0, float x1) {
    const unsigned h = pack_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = pack_bf16(r0, r1);
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    const unsigned l = pack_bf16(q0, q1);
    unsigned* w = reinterpret_cast<unsigned*>(img);
    const size_t at = ((size_t)row * CH + split_chunk<CH>(row, k >> 3)) * 4 + ((k & 7) >> 1);
    const size_t lvl = (size_t)rows * CH * 4;
    w[at] = h;
    w[at + lvl] = m;
    w[at + 2 * lvl] = l;
}
// once per launch: the data set (global, row-major with IPY floats per row) into the split image, the remainder columns and the labels
template <int I>
__device__ __forceinline__ void stage_split_data(const SplitLds& l, const float* __restrict__ data, int IPY, int Nall, int Npad) {
    typedef SplitK<I> K;
    for (int e = threadIdx.x; e < Npad * (K::KBF / 2); e += blockDim.x) {
        const int n = e / (K::KBF / 2), k = 2 * (e - n * (K::KBF / 2));
        const float x0 = (n < Nall && k < I) ? data[(size_t)n * IPY + k] : 0.0f;
        const float x1 = (n < Nall && k + 1 < I) ? data[(size_t)n * IPY + k + 1] : 0.0f;
        split_store_pair<K::CH>(l.xs, Npad, n, k, x0, x1);
    }
    for (int e = threadIdx.x; e < 2 * K::KR * Npad; e += blockDim.x) {
        const int s2 = e / Npad, n = e - s2 * Npad, k = K::KBF + s2;
        l.xr[e] = (n < Nall && k < I) ? data[(size_t)n * IPY + k] : 0.0f;
    }
    for (int n = threadIdx.x; n < Npad; n += blockDim.x) l.ylab[n] = (n < Nall) ? data[(size_t)n * IPY + I] : 0.0f;
}
// once per weight vector: W1 (k < 16 KB) into the split image; wval(idx) yields element idx of the flat vector
template <int I, class F>
__device__ __forceinline__ void split_weights(uint4* as, int H, F wval) {
    typedef SplitK<I> K;
    const int Hpad = ((H + 31) >> 5) << 5, hs = 31 - __clz(Hpad);       // H <= 64 here: 32 or 64
    for (int e = threadIdx.x; e < Hpad * (K::KBF / 2); e += blockDim.x) {
        const int kp = e >> hs, hid = e & (Hpad - 1), k = 2 * kp;      // consecutive threads: consecutive hidden units (w is [k][h])
        const float x0 = (hid < H && k < I) ? wval(k * H + hid) : 0.0f;
        const float x1 = (hid < H && k + 1 < I) ? wval((k + 1) * H + hid) : 0.0f;
        split_store_pair<K::CH>(as, Hpad, hid, k, x0, x1);
    }
}

template <int TASK, int I, int O, bool LEAN = false>
__device__ __forceinline__ EvalSums eval_rows_mfma_coop(const float* __restrict__ wl, const float* __restrict__ xt,
                                                        float* __restrict__ part, const float* __restrict__ xy, int IPY,
                                                        int H, int Ntr, int Nall, int Npad, float* __restrict__ red, float& extra) {
    constexpr int IK = (I + 1) & ~1, KS = IK / 2;
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: the unit loops branch on the SALU
    const int col = lane & 31, half = lane >> 5;
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    const int ntiles = (H + 31) >> 5;
    PTNN_DIAG(fw_begin);
    // phase 1: one unit = 32 data rows x 32 hidden units (11 row blocks x 2 tiles for Ionosphere).  With two tiles the waves
    // split between them, so a wave keeps ONE tile's operands -- the weights (A), biases and W2 rows of its 32 hidden units --
    // in registers for all of its row blocks, and only the data columns (B) change.  Up to three row blocks run at once:
    // the k-steps of one accumulator depend on each other (a 32x32x2 MFMA issues in 64 cycles but its result returns later),
    // three independent accumulators keep the matrix pipe busy.  Every accumulator still adds its k-steps in ascending
    // order.  Straight-line address arithmetic, no divergent control flow: absent units of a partial tile read the last
    // real unit again / whatever follows in LDS (finite weights) and meet W2 = 0.
    const int nrb = Npad >> 5;
    const int tsplit = (ntiles == 2 && nw >= 2) ? 2 : 1;
    const int t0 = (tsplit == 2) ? (wave & 1) : 0, tcount = (tsplit == 2) ? 1 : ntiles;
    const int rb0 = (tsplit == 2) ? (wave >> 1) : wave, rbstride = nw / tsplit;
    for (int tt = 0; tt < tcount; ++tt) {
        const int t = t0 + tt;
        const int hbase = t * 32;
        const float* pa = wl + half * H + min(hbase + col, H - 1);
        float aa[KS];
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const bool pad = (IK != I) && (s_ == KS - 1) && half;      // odd I: k = I of the upper lane half is padding
            const float va = pa[s_ * 2 * H - (pad ? H : 0)];
            aa[s_] = pad ? 0.0f : va;
        }
        // bias and W2 rows of this lane's 16 hidden units (absent units: both zero, so their sigmoid is a finite 0.5 that
        // meets W2 = 0)
        const int hq = hbase + 4 * half;
        const float* pb1 = wl + oB1 + hq;
        const float* pw2 = wl + oW2 + hq * O;
        float b1r[16], w2r[16][O];
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) {
            const int dh = 8 * (r_ >> 2) + (r_ & 3);
            const bool in = hq + dh < H;
            const float bv = pb1[dh];
            b1r[r_] = in ? bv : 0.0f;                                  // past H the read lands in padding: never let a NaN in
#pragma unroll
            for (int o = 0; o < O; ++o) { const float v = pw2[dh * O + o]; w2r[r_][o] = in ? v : 0.0f; }
        }
        auto batch = [&](auto nb_, int rb) {
            constexpr int NB = decltype(nb_)::value;
            float bf[NB][KS];
            f32x16 acc[NB];
#pragma unroll
            for (int b_ = 0; b_ < NB; ++b_) {
                const float* pb = xt + half * Npad + (rb + b_ * rbstride) * 32 + col;   // this lane's data row (Npad covers the last block)
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {                      // all k-steps in one batch of conflict-free LDS reads
                    const bool pad = (IK != I) && (s_ == KS - 1) && half;
                    const float vb = pb[(size_t)s_ * 2 * Npad - (pad ? Npad : 0)];
                    bf[b_][s_] = pad ? 0.0f : vb;
                }
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) acc[b_][r_] = 0.0f;
            }
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int b_ = 0; b_ < NB; ++b_) acc[b_] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[s_], bf[b_][s_], acc[b_], 0, 0, 0);
#pragma unroll
            for (int b_ = 0; b_ < NB; ++b_) {
                const int n = (rb + b_ * rbstride) * 32 + col;
                float sum[O];
#pragma unroll
                for (int o = 0; o < O; ++o) sum[o] = 0.0f;
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) {
                    const float hid = sigmoidf_fast(acc[b_][r_] - b1r[r_]);
#pragma unroll
                    for (int o = 0; o < O; ++o) sum[o] = fmaf(hid, w2r[r_][o], sum[o]);
                }
#pragma unroll
                for (int o = 0; o < O; ++o) {                          // hidden units 4..7, 12..15, ... live in lanes 32..63
                    const unsigned uu = __builtin_bit_cast(unsigned, sum[o]);
                    auto r2 = __builtin_amdgcn_permlane32_swap(uu, uu, false, false);
                    const float tot = __builtin_bit_cast(float, (unsigned)r2[0]) + __builtin_bit_cast(float, (unsigned)r2[1]);
                    if (half == 0) part[((size_t)t * Npad + n) * O + o] = tot;
                }
            }
        };
        int rb = rb0;
        for (; rb + 2 * rbstride < nrb; rb += 3 * rbstride) batch(std::integral_constant<int, 3>{}, rb);
        if (rb + rbstride < nrb) { batch(std::integral_constant<int, 2>{}, rb); rb += 2 * rbstride; }
        if (rb < nrb) batch(std::integral_constant<int, 1>{}, rb);
    }
    FW_DBG(0);                                                 // matrix products + epilogues of wave 0
    __syncthreads();
    FW_DBG(1);                                                 // waiting for the other waves
    // phase 2: one lane per data row joins the tiles (ascending) and scores the row
    float a_tr = 0.f, b_tr = 0.f, c_tr = 0.f, a_te = 0.f, b_te = 0.f, c_te = 0.f;
    float b2[O];
#pragma unroll
    for (int o = 0; o < O; ++o) b2[o] = wl[oB2 + o];
    for (int n = threadIdx.x; n < Nall; n += blockDim.x) {
        float tot[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float v = part[(size_t)n * O + o];
            if (ntiles > 1) v += part[((size_t)Npad + n) * O + o];
            tot[o] = v - b2[o];
        }
        const float y = xy[(size_t)n * IPY + I];
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
    FW_DBG(2);                                                 // scoring the rows
    const EvalSums es_ = reduce_eval<TASK, false, LEAN>(a_tr, b_tr, c_tr, a_te, b_te, c_te, red, extra);
    FW_DBG(3);                                                 // work-group reduction
    return es_;
}


// The forward pass itself: same tiling, work split and epilogue as eval_rows_mfma_coop (one unit = 32 data rows x 32 hidden
// units, a wave keeps one hidden tile's operands in registers for all its row blocks); per unit 6 KB bf16 matrix instructions
// + KR exact fp32 ones on ONE accumulator (a dependent chain runs at the full pipe rate), software-pipelined against the VALU
// epilogue of the previous row block (the bf16 instruction leaves the vector issue free for 24 of its 32 cycles).
template <int TASK, int I, int O, bool LEAN = false>
__device__ __forceinline__ EvalSums eval_rows_mfma_split(const float* __restrict__ wl, const SplitLds& sl, int H, int Ntr, int Nall,
                                                         int Npad, float* __restrict__ red, float& extra) {
    typedef SplitK<I> K;
    constexpr int KB = K::KB, KR = K::KR, CH = K::CH;
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int oW2 = I * H, oB1 = oW2 + H * O, oB2 = oB1 + H;
    const int ntiles = (H + 31) >> 5, Hpad = ntiles << 5;
    float* __restrict__ part = sl.part;
    PTNN_DIAG(fw_begin);
    const int nrb = Npad >> 5;
    const int tsplit = (ntiles == 2 && nw >= 2) ? 2 : 1;
    const int t0 = (tsplit == 2) ? (wave & 1) : 0, tcount = (tsplit == 2) ? 1 : ntiles;
    const int rb0 = (tsplit == 2) ? (wave >> 1) : wave, rbstride = nw / tsplit;
    for (int tt = 0; tt < tcount; ++tt) {
        const int t = t0 + tt;
        const int hbase = t * 32;
        // A: this lane's hidden unit, k = 16 s + 8 half .. + 7 of every level; the fp32 remainder straight from the flat vector
        bf16x8 a_h[KB], a_m[KB], a_l[KB];
        {
            const int row = hbase + col;
            const uint4* base = sl.as + (size_t)row * CH;
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) {
                const int c = split_chunk<CH>(row, 2 * s_ + half);
                a_h[s_] = __builtin_bit_cast(bf16x8, base[c]);
                a_m[s_] = __builtin_bit_cast(bf16x8, base[(size_t)Hpad * CH + c]);
                a_l[s_] = __builtin_bit_cast(bf16x8, base[(size_t)2 * Hpad * CH + c]);
            }
        }
        float a_r[KR > 0 ? KR : 1];
#pragma unroll
        for (int s_ = 0; s_ < KR; ++s_) {
            const int k = K::KBF + 2 * s_ + half;
            const float va = wl[min(k, I - 1) * H + min(hbase + col, H - 1)];
            a_r[s_] = (k < I && hbase + col < H) ? va : 0.0f;
        }
        // bias and W2 rows of this lane's 16 hidden units (absent units: both zero: sigmoid(0) = 0.5 meets W2 = 0)
        const int hq = hbase + 4 * half;
        const float* pb1 = wl + oB1 + hq;
        const float* pw2 = wl + oW2 + hq * O;
        float b1r[16], w2r[16][O];
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) {
            const int dh = 8 * (r_ >> 2) + (r_ & 3);
            const bool in = hq + dh < H;
            const float bv = pb1[dh];
            b1r[r_] = in ? bv : 0.0f;
#pragma unroll
            for (int o = 0; o < O; ++o) { const float v = pw2[dh * O + o]; w2r[r_][o] = in ? v : 0.0f; }
        }
        float b1s[16];                                                  // log2e b1: exp2(-log2e z + log2e b1) = exp(-(z - b1))
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) b1s[r_] = LOG2E * b1r[r_];
        struct BFrag { bf16x8 h[KB], m[KB], l[KB]; float r[KR > 0 ? KR : 1]; };
        auto load_b = [&](int rb, BFrag& b) {
            const int row = rb * 32 + col;                              // this lane's data row (Npad covers the last block)
            const uint4* base = sl.xs + (size_t)row * CH;
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) {
                const int c = split_chunk<CH>(row, 2 * s_ + half);
                b.h[s_] = __builtin_bit_cast(bf16x8, base[c]);
                b.m[s_] = __builtin_bit_cast(bf16x8, base[(size_t)Npad * CH + c]);
                b.l[s_] = __builtin_bit_cast(bf16x8, base[(size_t)2 * Npad * CH + c]);
            }
#pragma unroll
            for (int s_ = 0; s_ < KR; ++s_) b.r[s_] = sl.xr[(size_t)(2 * s_ + half) * Npad + row];
        };
        auto chain = [&](const BFrag& b) {
            f32x16 acc;
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[r_] = 0.0f;
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) {                           // the 2^-16 terms
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l[s_], b.h[s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h[s_], b.l[s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m[s_], b.m[s_], acc, 0, 0, 0);
            }
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) {                           // the 2^-8 terms
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m[s_], b.h[s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h[s_], b.m[s_], acc, 0, 0, 0);
            }
#pragma unroll
            for (int s_ = 0; s_ < KB; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h[s_], b.h[s_], acc, 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < KR; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_r[s_], b.r[s_], acc, 0, 0, 0);
            return acc;
        };
        // epilogue on register PAIRS (v_pk_fma_f32 / v_pk_add_f32: two elements per issue slot; this pass is VALU-bound):
        // exp2(-log2e (z - b1)) as ONE packed fma with the bias pre-scaled, 1 + e packed, two partial W2 sums (even / odd registers)
        auto finish = [&](const f32x16& acc, int rb) {
            const int n = rb * 32 + col;
            f32x2 sum2[O];
#pragma unroll
            for (int o = 0; o < O; ++o) sum2[o] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int r_ = 0; r_ < 16; r_ += 2) {
                const f32x2 zz = __builtin_elementwise_fma(f32x2{acc[r_], acc[r_ + 1]}, f32x2{-LOG2E, -LOG2E}, f32x2{b1s[r_], b1s[r_ + 1]});
                const f32x2 ee = f32x2{__builtin_amdgcn_exp2f(zz.x), __builtin_amdgcn_exp2f(zz.y)} + f32x2{1.0f, 1.0f};
                const f32x2 hid = f32x2{__builtin_amdgcn_rcpf(ee.x), __builtin_amdgcn_rcpf(ee.y)};
#pragma unroll
                for (int o = 0; o < O; ++o) sum2[o] = __builtin_elementwise_fma(hid, f32x2{w2r[r_][o], w2r[r_ + 1][o]}, sum2[o]);
            }
#pragma unroll
            for (int o = 0; o < O; ++o) {                               // hidden units 4..7, 12..15, ... live in lanes 32..63
                const unsigned uu = __builtin_bit_cast(unsigned, sum2[o].x + sum2[o].y);
                auto r2 = __builtin_amdgcn_permlane32_swap(uu, uu, false, false);
                const float tot = __builtin_bit_cast(float, (unsigned)r2[0]) + __builtin_bit_cast(float, (unsigned)r2[1]);
                if (half == 0) part[((size_t)t * Npad + n) * O + o] = tot;
            }
        };
        if (rb0 < nrb) {
            BFrag bcur, bnxt;
            load_b(rb0, bcur);
            int rb = rb0, nx = rb0 + rbstride;
            if (nx < nrb) load_b(nx, bnxt);
            f32x16 acc = chain(bcur);
            while (nx < nrb) {
                bcur = bnxt;
                const int nn = nx + rbstride;
                if (nn < nrb) load_b(nn, bnxt);
                const f32x16 acc2 = chain(bcur);
                finish(acc, rb);
                // one matrix instruction, then its share of the previous block's epilogue (16 elements x {4 VALU + 2
                // transcendental + O fma} over 6 KB + KR instructions)
#pragma unroll
                for (int q_ = 0; q_ < 6 * KB + KR; ++q_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (8 * (2 + O) + 8 + 6 * KB + KR - 1) / (6 * KB + KR), 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, (32 + 6 * KB + KR - 1) / (6 * KB + KR), 0);
                }
                acc = acc2; rb = nx; nx = nn;
            }
            finish(acc, rb);
        }
    }
    FW_DBG(0);
    __syncthreads();
    FW_DBG(1);
    // phase 2: one lane per data row joins the tiles (ascending) and scores the row
    float a_tr = 0.f, b_tr = 0.f, c_tr = 0.f, a_te = 0.f, b_te = 0.f, c_te = 0.f;
    float b2[O];
#pragma unroll
    for (int o = 0; o < O; ++o) b2[o] = wl[oB2 + o];
    for (int n = threadIdx.x; n < Nall; n += blockDim.x) {
        float tot[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float v = part[(size_t)n * O + o];
            for (int t = 1; t < ntiles; ++t) v += part[((size_t)t * Npad + n) * O + o];
            tot[o] = v - b2[o];
        }
        const float y = sl.ylab[n];
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
    FW_DBG(2);
    const EvalSums es_ = reduce_eval<TASK, false, LEAN>(a_tr, b_tr, c_tr, a_te, b_te, c_te, red, extra);
    FW_DBG(3);
    return es_;
}



template <int TASK, int I, int O>
__device__ __forceinline__ void segment_body(const SegParams& p, const SegDyn& dyn, const int step_begin, const int n_steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int r = blockIdx.x;
    const int gid = p.first_global + r;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int Nall = p.Ntr + p.Nte;
    Lds l = carve(smem, Nall, p.IPY, p.PS, p.H, p.FWS, p.use_lg != 0, p.xy_global == 0);
    const int P = p.P, PS = p.PS, H = p.H;
    const bool split = SplitK<I>::OK && p.fw_mfma == 2;

    // stage the data set and this replica's vectors (coalesced)
    if (p.xy_global) {
        l.xy = const_cast<float*>(p.data);                    // read-only; the rare row-major readers (chain start) go to L2
    } else {
        const float4* src = reinterpret_cast<const float4*>(p.data);
        float4* dst = reinterpret_cast<float4*>(l.xy);
        for (int e = tid; e < ((Nall + 2) * p.IPY) >> 2; e += nthr) dst[e] = src[e];
    }
    float* gw = dyn.w_state + (size_t)r * PS;
    for (int j = tid; j < PS; j += nthr) {
        l.w_cur[j] = gw[j];
        l.rec_w[j] = p.rec_w[(size_t)r * PS + j];
        if (p.use_lg) l.w_gd[j] = dyn.gd_w[(size_t)r * PS + j];
    }
    // MFMA forward pass (host decides): transposed data image and per-tile partial sums behind the common LDS block
    float* xt_l = smem + ((lds_floats(Nall, p.IPY, p.PS, p.H, p.FWS, p.use_lg != 0, p.xy_global == 0) + 3) & ~(size_t)3);
    float* part_l = xt_l + (size_t)I * p.Npad;
    SplitLds sl = {};
    if constexpr (SplitK<I>::OK) {
        if (split) {
            sl = carve_split<I>(xt_l, O, H, p.Npad);
            stage_split_data<I>(sl, p.data, p.IPY, Nall, p.Npad);
        }
    }
    if (p.fw_mfma == 1)
        for (int e = tid; e < I * p.Npad; e += nthr) xt_l[e] = p.xt[e];
    __syncthreads();

    const float T = p.temps[r];
    float eta = (TASK == TASK_REG) ? l.w_cur[P] : 0.0f;
    float* sf = p.st_f + (size_t)r * SF_COUNT;
    int* si = p.st_i + (size_t)r * SI_COUNT;
    float lik, prior_cur, tau_eta_last, rec_rmse_tr, rec_rmse_te, rec_acc_tr, rec_acc_te;
    int nacc, gd_valid, lg_count;

    if (step_begin == 0) {
        chain_startup<TASK, I, O>(p, l.xy, l.w_cur, l.fw, l.red, T, eta, lik, prior_cur);
        tau_eta_last = eta;
        rec_rmse_tr = rec_rmse_te = rec_acc_tr = rec_acc_te = 0.f;
        nacc = 0; gd_valid = 0; lg_count = 0;
        __syncthreads();
    } else {
        lik = sf[SF_LIK]; prior_cur = sf[SF_PRIOR]; tau_eta_last = sf[SF_TAU_LAST];
        rec_rmse_tr = sf[SF_REC_RMSE_TR]; rec_rmse_te = sf[SF_REC_RMSE_TE];
        rec_acc_tr = sf[SF_REC_ACC_TR]; rec_acc_te = sf[SF_REC_ACC_TE];
        nacc = si[SI_NACC]; gd_valid = dyn.gd_valid[r]; lg_count = si[SI_LG_COUNT];
    }

    PTNN_DIAG(coop_begin);
    const size_t trow = (size_t)r * p.trace_cap;        // traces are rings of trace_cap rows per replica (== S unless streaming)
    // A step is three phases between two work-group barriers (a random-walk step; a Langevin step adds its SGD epochs):
    //   A  proposal and its packed forward image in one pass over the weights
    //   B  forward pass over all rows; |proposal|^2 for the prior rides in the same reduction; the random tape of the NEXT
    //      step is drawn here too (it depends on nothing the chain computes), by the last wave alone when it fits one wave
    //      (that wave has the fewest data rows, for Iris none)
    //   C  every thread finishes likelihood, prior and the MH test from the reduced sums; trace row
    // The weight vectors are not copied on an accept: current, recorded and proposed vector rotate through three buffers
    // (the recorded vector differs from the current one only until the first accept after a swap), and so do the two
    // cached SGD epochs.  Double-buffered tape, one reduction array: nothing a slow thread still reads in C is written
    // before the next barrier.
    const int step_end = step_begin + n_steps;
    const int wave = uni_i(tid >> 6), nwaves = nthr >> 6;
    const bool tape_one_wave = ((P + 3) >> 2) + 1 <= WAVE && nwaves > 1;
    float* const wbuf = l.w_cur;                          // w_cur, w_prop, rec_w, w_gd, w_pgd: PS floats each, in this order
    int o_cur = 0, o_prop = PS, o_rec = 2 * PS, o_gd = 3 * PS, o_pgd = 4 * PS;
    int par = 0;
    int ring_pos = (step_begin + 1) % p.trace_cap;       // trace row of step i is row (i + 1) mod trace_cap of the replica's ring
    if (n_steps > 0) tape_step(p, gid, step_begin, l.noise, l.scal);
    __syncthreads();
    for (int i = step_begin; i < step_end; ++i) {
        STAMP(0);
        float* const w_cur = wbuf + o_cur;
        float* const w_prop = wbuf + o_prop;
        const float* const noise = l.noise + par * (PS + 8);
        const float* const scal = l.scal + par * (PS + 8);
        // R10 temperature schedule (REG:317-324): tempered until the switch step, canonical afterwards
        float adapttemp = (p.switch_step >= 0 && i >= p.switch_step) ? 1.0f : T;
        if (i == p.switch_step) {
            // re-evaluate the current w with the LAST PROPOSED tau (Q9, REG:322)
            EvalSums sc;
            float none = 0.0f;
            __syncthreads();                                // the previous step's readers of red[]
            if (split) {
                if constexpr (SplitK<I>::OK) {
                    split_weights<I>(sl.as, H, [&](int idx) { return w_cur[idx]; });
                    __syncthreads();
                    sc = eval_rows_mfma_split<TASK, I, O>(w_cur, sl, H, p.Ntr, Nall, p.Npad, l.red, none);
                }
            } else if (p.fw_mfma) {
                sc = eval_rows_mfma_coop<TASK, I, O>(w_cur, xt_l, part_l, l.xy, p.IPY, H, p.Ntr, Nall, p.Npad, l.red, none);
            } else {
                build_fw<I, O>(w_cur, l.fw, H, p.FWS);
                __syncthreads();
                sc = eval_rows<TASK, I, O>(l.fw, l.xy, p.IPY, p.FWS, H, p.Ntr, Nall, l.red);
            }
            float ll, r1, r2, a1, a2;
            finish_eval<TASK>(sc, p.Ntr, p.Nte, tau_eta_last, ll, r1, r2, a1, a2);
            lik = ll;                                       // adapttemp == 1
            __syncthreads();
        }
        STAMP(1);
        const float lx = scal[0], u = scal[1], n_eta = scal[2];
        float diff_prop = 0.0f;
        const bool lg = p.use_lg && (lx < p.l_prob);
        if (lg) {
            // Langevin proposal (REG:329-347): w_gd = SGD epoch from w (cached while w is unchanged),
            // w_proposal = w_gd + step_w * noise, w_prop_gd = SGD epoch from w_proposal
            float* const w_gd = wbuf + o_gd;
            float* const w_pgd = wbuf + o_pgd;
            if (!gd_valid) {
                if (tid < WAVE) sgd_sweep_dispatch<TASK, I, O>(w_cur, w_gd, l.xy, p.data, p.Ntr, H, p.lr);
                gd_valid = 1;
                __syncthreads();
            }
            for (int j = tid; j < P; j += nthr) w_prop[j] = fmaf(p.step_w, noise[j], w_gd[j]);
            __syncthreads();
            if (tid < WAVE) sgd_sweep_dispatch<TASK, I, O>(w_prop, w_pgd, l.xy, p.data, p.Ntr, H, p.lr);
            __syncthreads();
            // first - second = [-0.5 |w - w_prop_gd|^2 + 0.5 |w_proposal - w_gd|^2] / step_w^2; the second norm is
            // step_w^2 |noise|^2 exactly in real arithmetic
            const float d1 = block_sumsq_diff(w_cur, w_pgd, P, l.red);
            const float d2 = block_sumsq(noise, P, l.red);
            diff_prop = langevin_ratio(d1, d2, p.step_w, adapttemp);   // Q6
            lg_count += 1;
            if (!p.fw_mfma) build_fw<I, O>(w_prop, l.fw, H, p.FWS);
            if constexpr (SplitK<I>::OK) { if (split) split_weights<I>(sl.as, H, [&](int idx) { return w_prop[idx]; }); }   // w_prop: behind the barrier above
        } else if (p.fw_mfma) {
            for (int j = tid; j < P; j += nthr) w_prop[j] = fmaf(p.step_w, noise[j], w_cur[j]);
            // the split image of W1 is formed from the same expression, so nobody waits for w_prop
            if constexpr (SplitK<I>::OK) { if (split) split_weights<I>(sl.as, H, [&](int idx) { return fmaf(p.step_w, noise[idx], w_cur[idx]); }); }
        } else {
            propose_build_fw<I, O>(w_cur, noise, p.step_w, w_prop, l.fw, H, p.FWS);
        }
        __syncthreads();
        float eta_pro = eta;
        if (TASK == TASK_REG) { eta_pro = fmaf(p.step_eta, n_eta, eta); tau_eta_last = eta_pro; }
        STAMP(2);                                         // proposal (+ SGD epochs) and packed forward image

        float ssq = 0.0f;                                 // this thread's part of |proposal|^2 (block_sumsq's partition)
        for (int j = tid; j < P; j += nthr) ssq = fmaf(w_prop[j], w_prop[j], ssq);
        if (i + 1 < step_end) {
            float* const nn = l.noise + (par ^ 1) * (PS + 8);
            float* const ns = l.scal + (par ^ 1) * (PS + 8);
            {
            if (!tape_one_wave) tape_step(p, gid, i + 1, nn, ns);
            else if (wave == nwaves - 1) tape_step<true>(p, gid, i + 1, nn, ns);
            }
        }
        STAMP(3);                                         // next step's tape
        EvalSums es;
        if (split) { if constexpr (SplitK<I>::OK) es = eval_rows_mfma_split<TASK, I, O, true>(w_prop, sl, H, p.Ntr, Nall, p.Npad, l.red, ssq); }
        else if (p.fw_mfma) es = eval_rows_mfma_coop<TASK, I, O, true>(w_prop, xt_l, part_l, l.xy, p.IPY, H, p.Ntr, Nall, p.Npad, l.red, ssq);
        else es = eval_rows<TASK, I, O, false, true>(l.fw, l.xy, p.IPY, p.FWS, H, p.Ntr, Nall, l.red, ssq);
        const float lik_prop = finish_loglik<TASK>(es, p.Ntr, eta_pro) / adapttemp;
        STAMP(4);                                         // forward pass over all rows + likelihood
        const float prior_prop = prior_value<TASK>(p, ssq, eta_pro);

        // R9 Metropolis-Hastings (REG:372-423): NaN -> accept (Q8), overflow -> 1
        const float logalpha = (lik_prop - lik) + (prior_prop - prior_cur) + diff_prop;
        const float mh = (logalpha != logalpha) ? 1.0f : fminf(1.0f, expf_fast(logalpha));
        const bool accept = uni_i((u < mh) ? 1 : 0) != 0;  // the same value in every thread: a scalar branch
        const int acc_before = nacc;
        if (accept) {
            nacc += 1;
            lik = lik_prop;
            prior_cur = prior_prop;
            eta = eta_pro;
            // the recorded scores live in wave 0 only: thread 0 writes them (trace row, state write-back)
            if (wave == 0) {
                finish_scores<TASK>(es, p.Ntr, p.Nte, rec_rmse_tr, rec_rmse_te, rec_acc_tr, rec_acc_te);   // REG: acc 0 (REG:403-404); CLS: accuracy (CLS:414-415)
                if (TASK == TASK_REG) rec_acc_tr = eta;       // the regression's acc_train slot records eta (finish_eval<TASK, true>)
            }
            gd_valid = lg ? 1 : 0;                        // w_prop_gd is langevin_gradient(new w): keep it as the cache
            const int old_cur = o_cur;
            o_cur = o_prop; o_rec = o_prop; o_prop = old_cur;   // old_cur is neither the new current nor the new recorded vector
            if (lg) { const int t_ = o_gd; o_gd = o_pgd; o_pgd = t_; }
        }
        par ^= 1;
        STAMP(5);                                         // prior, MH, state update
        // trace row i+1 (the only HBM traffic of a step)
        const size_t tpos = trow + (size_t)ring_pos;
        ring_pos = (ring_pos + 1 == p.trace_cap) ? 0 : ring_pos + 1;
        float* prow = p.tr_pos_w + tpos * (size_t)p.PW;
        const float* const w_rec = wbuf + o_rec;
        for (int j = tid; j < p.PW; j += nthr) prow[j] = (j < P) ? w_rec[j] : 0.0f;
        if (tid == 0) {
            store_trace_row(p.tr_scal + tpos * TR_COUNT, (TASK == TASK_REG) ? lik_prop : lik_prop * adapttemp /* REG:391 / CLS:404 */,
                            rec_rmse_tr, rec_rmse_te, rec_acc_tr, rec_acc_te, acc_before /* REG:380 */, logalpha);
        }
        STAMP(6);                                         // trace row
    }
    PTNN_DIAG(coop_flush);

    // write the chain state back and post the swap scalars
    __syncthreads();
    for (int j = tid; j < PS; j += nthr) {
        gw[j] = (j == P) ? eta : wbuf[o_cur + j];
        p.rec_w[(size_t)r * PS + j] = wbuf[o_rec + j];
        if (p.use_lg) dyn.gd_w[(size_t)r * PS + j] = wbuf[o_gd + j];
    }
    if (tid == 0) {
        sf[SF_LIK] = lik; sf[SF_PRIOR] = prior_cur; sf[SF_TAU_LAST] = tau_eta_last;
        sf[SF_REC_RMSE_TR] = rec_rmse_tr; sf[SF_REC_RMSE_TE] = rec_rmse_te;
        sf[SF_REC_ACC_TR] = rec_acc_tr; sf[SF_REC_ACC_TE] = rec_acc_te;
        si[SI_NACC] = nacc; dyn.gd_valid[r] = gd_valid; si[SI_LG_COUNT] = lg_count;
        p.L_handoff[gid] = (TASK == TASK_REG) ? lik * T : lik;      // Q11
        p.L_final[gid] = lik;
        post_raw(p, gid, lik, prior_cur, T, step_begin + n_steps - 1);
    }
}
