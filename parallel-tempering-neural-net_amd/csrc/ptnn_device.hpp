// ptnn_device.hpp -- gfx950 (CDNA4, wave64) device code of the parallel-tempering FNN sampler.
//
// One work-group owns one replica (one temperature of the ladder).  Wave 0 of the group runs the
// inherently sequential parts (the row-by-row SGD sweep of Network.langevin_gradient, REG:99-118);
// all NW waves share the row-parallel parts (evaluate_proposal over train+test rows, REG:120-134,
// the Philox noise pass and the trace-row store).  Weights, the data set and every per-step
// vector live in LDS; HBM sees only the trace rows the result-file layout requires.
//
// Layout of this file: the interval BODIES of the five schedules (segment_body, segment_spec_body, segment_pack_body,
// segment_wide_body, segment_tree_body: all MH steps of one swap interval), swap_block (one work-group's share of a swap round),
// and at the end persistent_loop + the __global__ kernels: body, grid barrier, swap_block, next interval -- one launch per run.
//
// Written for gfx950 only: wave size 64, DPP row operations, v_permlane{16,32}_swap.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

// No implicit fused multiply-adds anywhere in the device code: whether `a * b + c` becomes one v_fma or a multiply and an add
// would otherwise be decided per inlined copy by what surrounds it, and the same function could round differently in two kernels
// (found by the stress tool: the bias update `b1 -= g * ldh` of the SGD epoch differed by one ulp between the packed and the
// multi-CU speculative kernel on classification nets).  Every fused operation in this file is written as fmaf / v_pk_fma.
#pragma clang fp contract(off)

namespace ptnn {

constexpr int TASK_REG = 0;
constexpr int TASK_CLS = 1;
constexpr int WAVE = 64;
// Diagnostic hooks.  The product build contains none of the diagnostic code: STAMP / FW_DBG / PTNN_DIAG expand to nothing.  A
// diagnostic build (-DPTNN_STAMPS, profiles/tools/build_stamps.sh; never the product) includes ptnn_diag.hpp, which holds the
// bodies: in-kernel cycle stamps per phase of a round, summed into SegParams::stamps and read back by ptnn_debug_stamps.
#ifdef PTNN_STAMPS
#include "ptnn_diag.hpp"
#else
#define STAMP(slot) do { } while (0)
#define FW_DBG(q_) do { } while (0)
#define PTNN_DIAG(name)
#endif

constexpr int MAX_WAVES = 8;            // waves per work-group: 2 per SIMD, 256 VGPRs each
constexpr int MAX_THREADS = MAX_WAVES * WAVE;

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float LOG_2PI = 1.8378770664093453f;

// Philox4x32-10 streams (specification shared with oracle/ptnn_oracle.py)
constexpr uint32_t STREAM_STEP = 0, STREAM_WNOISE = 1, STREAM_SWAP = 2, STREAM_INIT = 3;

// per-replica float state (st_f) and int state (st_i) slots
enum { SF_LIK = 0, SF_PRIOR, SF_TAU_LAST, SF_REC_RMSE_TR, SF_REC_RMSE_TE, SF_REC_ACC_TR, SF_REC_ACC_TE, SF_COUNT = 8 };
enum { SI_NACC = 0, SI_REC_ROW, SI_LG_COUNT, SI_LG_ACC, SI_COUNT = 4 };   // accepted steps, trace row holding the last recorded pos_w (compact traces), Langevin steps proposed / accepted
// scalar trace row of MH step i (row i + 1): what the per-chain result files need beside pos_w (REG:454-481) -- likeh_list
// column 0 (REG:391 / CLS:404), rmse_train, rmse_test, acc_train, acc_test, accept_list (the count BEFORE the step, REG:380, as
// int bits) -- plus the step's log alpha as the kernel computed it (diagnostic: the parity tests measure its fp32 error with it)
// TR_ACC_TR of a REGRESSION (whose acc_train is identically 0, REG:403): eta of the recorded state (finish_eval<TASK, true>)
// TR_SRC (int bits): compact traces only -- the trace row that holds this step's pos_w values (the step's own row when it was
// accepted, else the row of the last accepted step: pos_w[i+1] = pos_w[i] on a reject, REG:417); 0 in the full layout
enum { TR_LIKEH = 0, TR_RMSE_TR, TR_RMSE_TE, TR_ACC_TR, TR_ACC_TE, TR_ACCEPT, TR_LOGALPHA, TR_SRC, TR_COUNT = 8 };

// one lane writes the whole row: two 16-byte stores into one 32-byte sector
__device__ __forceinline__ void store_trace_row(float* __restrict__ row, float likeh, float rm_tr, float rm_te, float ac_tr, float ac_te,
                                                int accept_before, float logalpha, int src_row = 0) {
    float4* q = reinterpret_cast<float4*>(row);
    q[0] = make_float4(likeh, rm_tr, rm_te, ac_tr);
    q[1] = make_float4(ac_te, __int_as_float(accept_before), logalpha, __int_as_float(src_row));
}

struct SegParams {
    int H, P, PS;            // hidden units, parameters, state row length (P + 1 rounded up to 4)
    int Ntr, Nte, IPY, FWS;  // rows, data row stride (floats), packed forward row stride (floats)
    int pk_nred;             // packed schedule: log2 of the lane-group width (3: n_hidden <= 8, 4: n_hidden <= 16)
    int noise_shared;        // Q14: every replica reads the step tape of replica 0
    int fw_mfma;             // cooperative schedule: forward pass on the matrix cores (host decides: 24 <= H <= 64, I >= 6): 1 exact fp32
                             // instruction, 2 split bf16 operands (SplitK; cooperative kernel only)
    int xy_global;           // split forward pass without room for the row-major data image in LDS: rows are read from p.data
    int S, switch_step, use_lg;
    int trace_cap;           // rows per replica in the trace rings
    int first_global;
    float l_prob, lr, step_w, step_eta;
    float inv_2sig2, prior_c, nu1, nu2;
    uint32_t seed_lo, seed_hi;
    const float* data;       // [Ntr+Nte][IPY]: x_0..x_{I-1}, y, pad
    float* w_state;          // current (w, eta) rows [Rl][PS]
    float* rec_w;            // last recorded pos_w row [Rl][PS]
    float* gd_w;             // cached langevin_gradient(w) [Rl][PS]
    float* st_f;             // [Rl][SF_COUNT]
    int* st_i;               // [Rl][SI_COUNT]
    int* gd_valid;           // [Rl] 1 when gd_w holds langevin_gradient of the current w
    const float* temps;      // [Rl]
    float* L_handoff;        // [Rglobal] posted scalar at a hand-off (REG:430 / CLS:439)
    float* L_final;          // [Rglobal] end-of-chain scalar (REG:442 / CLS:451)
    float* L_raw;            // [Rglobal] untempered log-likelihood of the current state (swap_rule 1 only)
    float* prior_post;       // [Rglobal] prior of the current state (swap_rule 1 only)
    float* tr_pos_w;         // [Rl][S][PW]: pos_w rows padded to whole 64-byte sectors (pad written as zeros: no partial-sector writes)
    int PW;                  // P rounded up to 16 floats
    float* tr_scal;          // [Rl][S][TR_COUNT]: the scalars of a step in ONE 32-byte row (one sector instead of seven)
    // speculative schedule across G work-groups (CUs) per replica
    int G;                   // work-groups per replica (1 = no cross-CU exchange)
    unsigned epoch_base;     // granule tags of this launch are epoch_base + round
    int tree_ahead;          // tree schedule: LDS holds two sets of tapes, the next round's are drawn while the records travel
    int wide_window;         // wide nets, several work-groups: steps per round (<= WIDE_WINDOW)
    unsigned long long* xverdict; // [Rl][2][MAX_SLOTS] one {tag, accepted?} granule per slot and round: all a foreign group polls
    unsigned long long* xslots;   // [Rl][2][MAX_SLOTS][16] result granules of an ACCEPTED slot (read by the other groups at commit)
    unsigned long long* xw;       // [Rl][2][MAX_SLOTS][2 PS] accepted-proposal granules
    int* error_flag;         // != 0 after a launch: a bounded spin expired
    unsigned long long* stamps;   // diagnostic build only (PTNN_STAMPS): cycle sums per phase, else unused
    float* wide_scratch;     // [Rl][3][PS] proposal, its SGD epoch, noise (wide nets only: these do not fit in LDS)
    const float* xt;         // wide nets: transposed data image Xt[k][Npad] (B operand of the MFMA forward pass), or null
    const uint4* xs;         // wide nets, split-operand forward pass (fw_mfma == 2): [3 levels][Npad rows][SplitK::CH chunks of 8 bf16], or null
    int Npad;                // rows of Xt, Nall rounded up to 32
    int forward_bf16;        // 1: forward GEMM operands rounded to bf16 (fp32 accumulate); 0: exact fp32 MFMA
    int* seg_progress;       // pinned host word or null (RCCL communicator attached): block 0 stores seg_ordinal when this launch ends
    int seg_ordinal;         // with a swap round due -- "the collective of round seg_ordinal - 1 is next on the stream" (ptnn.hip: wait_stream)
    int xcd_granules;        // 1: work-groups of a replica that find themselves on one XCD exchange through its L2 (granule_*_xcd); 0: always agent scope
    unsigned long long* xswap;    // granules of the swap rounds a multi-group launch runs by itself (tree, packed multi-CU): 2 x swap_xchg_granules, or null
    int compact;             // wide nets, all rows resident (trace_cap == S): a REJECTED step writes no pos_w row, only the index of the
                             // row it repeats (TR_SRC); ptnn_get_traces fills the rows in.  A 70 KB copy per rejected step otherwise.
};

// What changes from one swap interval to the next inside one launch (persistent_loop): which of the two state buffers is
// current, and where the granule tags continue.  Kept apart from SegParams so that the kernel argument itself stays constant (a
// modified copy of it, live across the whole interval, cost ~60 scalar registers and pushed two kernels into scratch).
struct PersistParams;
typedef __attribute__((address_space(4))) const PersistParams* persist_cptr;
struct SegDyn {
    persist_cptr pp;         // the launch's PersistParams (kernel-argument segment), found by the KERNEL and handed down: a body that the
                             // compiler does not inline (many-class heads) must not look for the kernel arguments itself
    float* w_state;          // current (w, eta) rows [Rl][PS]
    float* gd_w;             // cached langevin_gradient(w) [Rl][PS]
    int* gd_valid;           // [Rl]
    unsigned epoch_base;     // granule tags of this interval are epoch_base + round
};

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
template <int TASK, int I, int O, int NE>
__device__ __forceinline__ void sgd_sweep_wide_n(const float* const (&w_in)[NE], float* const (&w_out)[NE], const float* __restrict__ data,
                                                 int IPY, int Ntr, int H, float lr, float* __restrict__ part, const float* w_ref,
                                                 float (&d1_out)[NE]) {
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
    asm volatile(".p2align 8");
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
        auto put = [&](int idx, float v) {
            wo[idx] = v;
            if (w_ref) { const float d = w_ref[idx] - v; d1 = fmaf(d, d, d1); }
        };
        if (act) {
#pragma unroll
            for (int i = 0; i < I; ++i) put(i * H + t, IC * w1[e][i >> 1][i & 1]);
#pragma unroll
            for (int o = 0; o < O; ++o) put(oW2 + t * O + o, IC * w2[e][o]);
            put(oB1 + t, -IC * nb1[e]);
        }
        if (t == 0) {
#pragma unroll
            for (int o = 0; o < O; ++o) put(oB2 + o, -IC * cl[e][o]);
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
                sgd_sweep_wide<TASK, I, O>(w_cur, w_gd, xy, p.IPY, p.Ntr, H, p.lr, part);
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
            if (lg) {
                float d1;
                if constexpr (RES) d1 = block_sum(sgd_sweep_wide<TASK, I, O>(w_prop, w_pgd, xy, p.IPY, p.Ntr, H, p.lr, part, w_cur), red);
                else {
                    sgd_sweep_wide<TASK, I, O>(w_prop, w_pgd, xy, p.IPY, p.Ntr, H, p.lr, part);
                    d1 = block_sumsq_diff(w_cur, w_pgd, P, red);
                }
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
            const float logalpha = (lik_prop - lik) + (prior_prop - prior_cur) + diff_prop;
            const float mh = (logalpha != logalpha) ? 1.0f : fminf(1.0f, expf_fast(logalpha));
            const bool accept = u < mh;
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

// ------------------------------------------------------------------------------------------------
// Prefetching ("tree") schedule for random-walk classification chains, where half of the proposals are accepted and
// speculating on rejections alone gains nothing: G = 2^D - 1 work-groups (one per CU) evaluate, at the same time, the
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

}  // namespace ptnn
