// ptnn_dev_math.hpp -- part of ptnn_device.hpp (textually included there, inside namespace ptnn; not a stand-alone header):
// scalar math, the Philox tape, wave reductions, LDS helpers, the hand-scheduled SGD rows of the 4-H-1 nets.

// ------------------------------------------------------------------------------------------------
// scalar math on the hardware transcendental units
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_fast(float z) {
    // 1 / (1 + exp(-z)) as v_mul, v_exp_f32, v_add, v_rcp_f32
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * z));
}
__device__ __forceinline__ float expf_fast(float x) { return __builtin_amdgcn_exp2f(LOG2E * x); }
// a value every lane of the wave holds alike, moved to a scalar register (the chain state of a replica -- likelihood, prior,
// counters -- is such a value; loaded from memory or LDS it would occupy a VGPR each for the whole launch)
__device__ __forceinline__ float uni_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float logf_fast(float x) { return LN2 * __builtin_amdgcn_logf(x); }

// ------------------------------------------------------------------------------------------------
// Philox4x32-10
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&x)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}
// 23-bit uniform in (0,1): ((x >> 9) + 0.5) * 2^-23, exact in fp32
__device__ __forceinline__ float u23(uint32_t x) { return fmaf((float)(x >> 9), 1.1920928955078125e-07f, 5.9604644775390625e-08f); }
// Box-Muller: r = sqrt(-2 ln u1); (r cos 2 pi u2, r sin 2 pi u2).  v_sin/v_cos take revolutions.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float r = __builtin_amdgcn_sqrtf(-2.0f * LN2 * __builtin_amdgcn_logf(u23(a)));
    const float t = u23(b);
    n0 = r * __builtin_amdgcn_cosf(t);
    n1 = r * __builtin_amdgcn_sinf(t);
}

// ------------------------------------------------------------------------------------------------
// wave-wide all-lanes sum over the first 2^NRED lanes' groups: DPP inside a row of 16, permlane swaps across rows
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// after the call every lane of each aligned group of 2^NRED lanes holds that group's sum
template <int NRED>
__device__ __forceinline__ float group_allsum(float v) {
    if (NRED >= 1) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]   : lane ^ 1
    if (NRED >= 2) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]   : lane ^ 2
    if (NRED >= 3) v += dpp_mov<0x141>(v);   // row_half_mirror       : 7 - lane within 8
    if (NRED >= 4) v += dpp_mov<0x140>(v);   // row_mirror            : 15 - lane within 16
    if (NRED >= 5) {                         // rows 0<->1, 2<->3
        const unsigned u = __builtin_bit_cast(unsigned, v);
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    if (NRED >= 6) {                         // lanes 0-31 <-> 32-63
        const unsigned u = __builtin_bit_cast(unsigned, v);
        auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    return v;
}
__device__ __forceinline__ float wave_allsum(float v) { return group_allsum<6>(v); }
// N independent wave sums, stage by stage: an in-order wave then always has the other values' stage to issue while one value's
// DPP result is in flight (same operations per value as wave_allsum)
template <int N>
__device__ __forceinline__ void wave_allsum_n(float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov<0xB1>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov<0x4E>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov<0x141>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_mov<0x140>(v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const unsigned u = __builtin_bit_cast(unsigned, v[k]);
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v[k] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const unsigned u = __builtin_bit_cast(unsigned, v[k]);
        auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v[k] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
}

// A "group" is either the whole work-group (WL = false: cooperative schedule, all waves work on one MH step) or one
// wavefront (WL = true: speculative schedule, every wave works on its own MH step).  LDS traffic inside one wave is
// ordered by the hardware; the fence only stops the compiler from moving accesses across it.
template <bool WL> __device__ __forceinline__ int gtid() { return WL ? (int)(threadIdx.x & 63) : (int)threadIdx.x; }
template <bool WL> __device__ __forceinline__ int gsize() { return WL ? WAVE : (int)blockDim.x; }
template <bool WL> __device__ __forceinline__ void gsync() {
    // WL: the LDS executes one wave's instructions in issue order, so a later ds_read of any lane sees an earlier
    // ds_write of any lane; only the compiler has to be stopped from reordering (no s_waitcnt vmcnt: a work-group
    // scope fence would also wait for the trace stores still in flight)
    if (WL) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    else __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// LDS helpers: N floats from a 16-byte aligned address as ds_read_b128s
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void lds_load(const float* __restrict__ p, float (&v)[N]) {
    constexpr int N4 = N / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int k = 0; k < N4; ++k) {
        const float4 t = p4[k];
        v[4 * k + 0] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int k = 4 * N4; k < N; ++k) v[k] = p[k];
}

// ------------------------------------------------------------------------------------------------
// Hand-scheduled SGD rows for the reference's time-series nets (TASK_REG, I = 4, O = 1, lane groups of 8 or 16 hidden
// units: Sunspot/Lazer 4-5-1, Mackey-Glass 4-10-1).
// A lone wave issues ONE instruction of any kind per 4 cycles, so the cost of a row is its instruction count; the
// compiler's version of the loop carries ~38-44 instructions per row (address arithmetic, register copies of the row
// ring, s_nop in the VALU->DPP and transcendental->use hazard slots).  Here a row is 25 VALU + 1 s_load + 1 s_waitcnt
// with every hazard slot holding useful work, and 4 scalar instructions of loop control per 4 rows:
//   * rows come from the global copy of the data image through the scalar cache (wave-uniform address): s_load_dwordx8
//     puts x0..x3, y, d = 1 + x[n].x[n-1] into SGPRs that feed the VALU directly; ring of four rows A..D (previous,
//     current, next, arriving); row n+3 is requested during row n into the buffer of row n-1, right after the wait for
//     row n+2, so a request has a whole row of time;
//   * the W1/B1 update of row n-1 and the partial pre-activation of row n+1 (two v_pk_fma_f32 + one add) fill the
//     hazard slots of row n (deferred update, see sgd_sweep);
//   * {B2' (lane 0), W} and {W1'[0],W1'[1]}, {W1'[2],W1'[3]} are updated with v_pk_fma_f32;
//   * scaling that removes two multiplies: with a = (lr log2 e)^-1/2 the loop keeps W = a W2' and computes
//     HN = -hid / a = rcp(-a (1 + 2^z)) (the "+1" of the sigmoid becomes an fma), so that
//         hid W2' = -HN W,    W += od HN  (is  W2' += (c lr) od hid),    lhd = (od W) HN fma(HN, -lr a, -lr).
// Physical registers are fixed (v40-v66, s36-s72) and declared as clobbers; the state enters and leaves through
// operands.  Processes rows 0 .. 4 iters - 1 and applies the pending update of the last one.
// Hazards honoured by construction (gfx950): transcendental result -> 1 slot before a non-transcendental use,
// VALU result -> 2 slots before a DPP read, SMEM result -> s_waitcnt lgkmcnt(0) before use and before the block ends.
// ------------------------------------------------------------------------------------------------
#define PTNN_SW_STEP(P01, P23, PALL, XY, XD, N01, N23, ZP, ZN, OFF, DPP4)                                               \
    "v_fmac_f32_e32 " ZP ", " XD ", v50\n"                             /*  z = zp + lhd d                       */ \
    "v_exp_f32_e32 v57, " ZP "\n"                                                                                  \
    "v_pk_fma_f32 v[40:41], v[50:51], " P01 ", v[40:41] op_sel_hi:[0,1,1]\n" /* W1[0:1] += lhd x[n-1]      */ \
    "v_fma_f32 v57, v57, s71, s71\n"                                  /*  -a (1 + 2^z)                         */ \
    "v_rcp_f32_e32 v47, v57\n"                                        /*  HN = -hid / a                        */ \
    "v_pk_fma_f32 v[42:43], v[50:51], " P23 ", v[42:43] op_sel_hi:[0,1,1]\n"                                  \
    "v_fma_f32 v58, -v47, v45, v44\n"                               /*  hid W2' + B2'(lane 0)                */ \
    "s_waitcnt lgkmcnt(0)\n"                                            /*  row n+2 has arrived                  */ \
    "s_load_dwordx8 " PALL ", s[68:69], " OFF "\n"                      /*  row n+3 -> buffer of row n-1         */ \
    "v_add_f32_dpp v58, v58, v58 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                  \
    "v_add_f32_e32 v48, v48, v50\n"                                  /*  -B1' += lhd                          */ \
    "v_fma_f32 v59, v47, s72, v66\n"                                 /*  -lr a HN - lr                        */ \
    "v_add_f32_dpp v58, v58, v58 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                  \
    "v_pk_fma_f32 v[52:53], " N01 ", v[40:41], v[48:49]\n"        /*  partial z of row n+1 ...             */ \
    "v_mul_f32_e32 v60, v47, v59\n"                                  /*  lr hid (1 - hid) / a                 */ \
    "v_add_f32_dpp v58, v58, v58 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                      \
    DPP4                                                                /*  16-lane groups: one more stage       */ \
    "v_exp_f32_e32 v57, v58\n"                                                                                    \
    "v_pk_fma_f32 v[52:53], " N23 ", v[42:43], v[52:53]\n"                                                    \
    "v_add_f32_e32 v57, 1.0, v57\n"                                                                               \
    "v_rcp_f32_e32 v61, v57\n"                                        /*  out                                  */ \
    "v_add_f32_e32 " ZN ", v52, v53\n"                                                                            \
    "v_sub_f32_e32 v62, " XY ", v61\n"                                /*  y - out                              */ \
    "v_fma_f32 v63, -v61, v61, v61\n"                                                                           \
    "v_mul_f32_e32 v64, v62, v63\n"                                  /*  od                                   */ \
    "v_mul_f32_e32 v65, v64, v45\n"                                  /*  od W (pre-update)                    */ \
    "v_mul_f32_e32 v50, v65, v60\n"                                  /*  lhd                                  */ \
    "v_pk_fma_f32 v[44:45], v[64:65], v[46:47], v[44:45] op_sel_hi:[0,1,1]\n" /* {B2',W} += od {c lr m0, HN} */

#define PTNN_SW_ASM(DPP4) \
    asm volatile( \
        "s_mov_b64 s[68:69], %[gp]\n" \
        "s_mov_b32 s70, %[endlo]\n" \
        "s_mov_b32 s71, %[kb]\n" \
        "s_mov_b32 s72, %[k1]\n" \
        "s_load_dwordx8 s[44:51], s[68:69], 0x0\n" \
        "s_load_dwordx8 s[52:59], s[68:69], 0x20\n" \
        "s_load_dwordx8 s[60:67], s[68:69], 0x40\n" \
        "s_mov_b64 s[36:37], 0\n" \
        "s_mov_b64 s[38:39], 0\n" \
        "v_mov_b32_e32 v40, %[w0]\n" \
        "v_mov_b32_e32 v41, %[w1]\n" \
        "v_mov_b32_e32 v42, %[w2]\n" \
        "v_mov_b32_e32 v43, %[w3]\n" \
        "v_mov_b32_e32 v44, %[cl]\n" \
        "v_mov_b32_e32 v45, %[v2]\n" \
        "v_mov_b32_e32 v46, %[cm0]\n" \
        "v_mov_b32_e32 v47, 0\n" \
        "v_mov_b32_e32 v48, %[nb]\n" \
        "v_mov_b32_e32 v49, 0\n" \
        "v_mov_b32_e32 v50, 0\n" \
        "v_mov_b32_e32 v51, 0\n" \
        "v_mov_b32_e32 v66, %[k2]\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "v_pk_fma_f32 v[52:53], s[44:45], v[40:41], v[48:49]\n" \
        "s_nop 1\n" \
        "v_pk_fma_f32 v[52:53], s[46:47], v[42:43], v[52:53]\n" \
        "s_nop 1\n" \
        "v_add_f32_e32 v54, v52, v53\n" \
        "L_ptnn_sweep_%=:\n" \
 \
        PTNN_SW_STEP("s[36:37]", "s[38:39]", "s[36:43]", "s48", "s49", "s[52:53]", "s[54:55]", "v54", "v55", "0x60", DPP4) \
        PTNN_SW_STEP("s[44:45]", "s[46:47]", "s[44:51]", "s56", "s57", "s[60:61]", "s[62:63]", "v55", "v54", "0x80", DPP4) \
        PTNN_SW_STEP("s[52:53]", "s[54:55]", "s[52:59]", "s64", "s65", "s[36:37]", "s[38:39]", "v54", "v55", "0xa0", DPP4) \
        PTNN_SW_STEP("s[60:61]", "s[62:63]", "s[60:67]", "s40", "s41", "s[44:45]", "s[46:47]", "v55", "v54", "0xc0", DPP4) \
        "s_add_u32 s68, s68, 0x80\n" \
        "s_addc_u32 s69, s69, 0\n" \
        "s_cmp_lg_u32 s68, s70\n" \
        "s_cbranch_scc1 L_ptnn_sweep_%=\n" \
 \
        "s_waitcnt lgkmcnt(0)\n" \
        "v_pk_fma_f32 v[40:41], v[50:51], s[36:37], v[40:41] op_sel_hi:[0,1,1]\n" \
        "v_pk_fma_f32 v[42:43], v[50:51], s[38:39], v[42:43] op_sel_hi:[0,1,1]\n" \
        "v_add_f32_e32 v48, v48, v50\n" \
        "s_nop 1\n" \
        "v_mov_b32_e32 %[o0], v40\n" \
        "v_mov_b32_e32 %[o1], v41\n" \
        "v_mov_b32_e32 %[o2], v42\n" \
        "v_mov_b32_e32 %[o3], v43\n" \
        "v_mov_b32_e32 %[ocl], v44\n" \
        "v_mov_b32_e32 %[ow2], v45\n" \
        "v_mov_b32_e32 %[onb], v48\n" \
        : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [onb] "=&v"(onb), [ow2] "=&v"(ow2), [ocl] "=&v"(ocl) \
        : [gp] "s"(gp), [endlo] "s"(end_lo), [kb] "s"(kb), [k1] "s"(k1), [k2] "v"(k2), [w0] "v"(w1[0]), [w1] "v"(w1[1]), \
          [w2] "v"(w1[2]), [w3] "v"(w1[3]), [cl] "v"(cl), [v2] "v"(w2 * sa), [cm0] "v"(clr * m0), [nb] "v"(nb1) \
        : "memory", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", \
          "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", \
          "v66", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", \
          "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", \
          "s67", "s68", "s69", "s70", "s71", "s72");

template <int NRED>
__device__ __forceinline__ void sweep_rows_reg41(float (&w1)[4], float& nb1, float& w2, float& cl, float m0, float lr,
                                                 float clr, const float* gdata, int iters) {
    const unsigned long long gp = (unsigned long long)(uintptr_t)gdata;
    const unsigned end_lo = (unsigned)gp + (unsigned)iters * 128u;     // low word of the running pointer after the last pass
    const float lr_u = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lr)));
    const float sa = __builtin_amdgcn_rsqf(LOG2E * lr_u), sb = __builtin_amdgcn_sqrtf(LOG2E * lr_u);   // a, 1 / a
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float kb = uni(-sa), k1 = uni(-lr_u * sa), k2 = -lr_u;
    float o0, o1, o2, o3, onb, ow2, ocl;
    if constexpr (NRED == 3) {
        PTNN_SW_ASM("")
    } else {
        PTNN_SW_ASM("s_nop 1\n"
                    "v_add_f32_dpp v58, v58, v58 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
    }
    w1[0] = o0; w1[1] = o1; w1[2] = o2; w1[3] = o3; nb1 = onb; w2 = ow2 * sb; cl = ocl;
}

// row stride of the data image in floats: x[0..I-1], y, 1 + x[n].x[n-1] (see sgd_sweep), padded to a multiple of 4
__host__ __device__ constexpr int sweep_row_stride(int I) { return (I + 2 + 3) & ~3; }
typedef __attribute__((address_space(4))) float cfloat;    // constant address space: uniform loads become s_load
